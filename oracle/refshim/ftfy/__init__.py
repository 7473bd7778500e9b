"""Stub of ftfy (reference open_clip/tokenizer.py:11): identity is exact for ASCII prompts."""


def fix_text(s):
    return s
