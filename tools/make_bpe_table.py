#!/usr/bin/env python
"""Build diffbir_amd/model/clip_bpe_merges.txt.gz from the published CLIP byte-pair vocabulary.

Source: `bpe_simple_vocab_16e6.txt.gz` of OpenAI CLIP (MIT licence; redistributed by open_clip and vendored by the
reference under diffbir/model/open_clip/).  It is DATA the SD-2.1 text encoder weights are tied to (token ids index the
checkpoint's `token_embedding.weight`), like checkpoint key names: category "unavoidable for compatibility".  Only the
48 894 merge rules the tokenizer uses (lines 1 .. 49152-256-2 of the file) are kept, one `a b` pair per line.

    python tools/make_bpe_table.py /path/to/bpe_simple_vocab_16e6.txt.gz
"""
import gzip
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = sys.argv[1]
    lines = gzip.open(src).read().decode("utf-8").split("\n")[1:49152 - 256 - 2 + 1]
    out = os.path.join(ROOT, "diffbir_amd", "model", "clip_bpe_merges.txt.gz")
    with gzip.GzipFile(out, "wb", mtime=0) as f:
        f.write(("\n".join(lines) + "\n").encode("utf-8"))
    print(out, len(lines), "merges", os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
