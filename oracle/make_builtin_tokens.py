"""TEST/BUILD INFRASTRUCTURE. Regenerates diffbir_amd/model/clip_builtin_tokens.json (token ids of the reference's
default prompts, inference.py:42-52, and the benchmark negative prompt) with the reference tokenizer.
Run here only: python -m oracle.make_builtin_tokens"""
import json
import os
import runpy

from .cases import NEG_PROMPT
from .ref_import import REFERENCE_ROOT, load_reference

if __name__ == "__main__":
    load_reference()
    from diffbir.model.open_clip import tokenize
    src = open(os.path.join(REFERENCE_ROOT, "inference.py")).read()
    ns = {}
    start = src.index("DEFAULT_POS_PROMPT = (")
    end = src.index("def parse_args")
    exec(src[start:end], ns)
    prompts = [ns["DEFAULT_POS_PROMPT"], ns["DEFAULT_NEG_PROMPT"], NEG_PROMPT]
    tab = {}
    for p in prompts:
        ids = tokenize([p])[0].tolist()
        tab[p] = ids[1:ids.index(49407)]
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "diffbir_amd", "model",
                       "clip_builtin_tokens.json")
    json.dump(tab, open(out, "w"))
    print("wrote", out, [len(v) for v in tab.values()])
