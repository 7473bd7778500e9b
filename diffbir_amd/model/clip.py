"""OpenCLIP text tower (FrozenOpenCLIPEmbedder, reference clip.py:8-61) — COLD path.

Two calls per pipeline run, 0.05 % of the FLOPs: kept on plain PyTorch-ROCm ops by design; only
the text tower is built (the reference builds and deletes a ViT-H vision tower, clip.py:21-22).  State-dict keys
are those below `cond_stage_model.` in SD checkpoints (`model.transformer.resblocks.N...`).
"""
import gzip
import html
import json
import os
from functools import lru_cache
from typing import Dict, List

import torch
import torch.nn.functional as F

from .base import NativeModule
from .specs import clip_text_spec

T = torch.Tensor
SOT, EOT = 49406, 49407
_BUILTIN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "clip_builtin_tokens.json")


@lru_cache()
def _byte_unicode() -> Dict[int, str]:
    keep = list(range(33, 127)) + list(range(161, 173)) + list(range(174, 256))
    chars, extra = keep[:], 0
    for b in range(256):
        if b not in keep:
            keep.append(b)
            chars.append(256 + extra)
            extra += 1
    return {b: chr(c) for b, c in zip(keep, chars)}


_MERGES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "clip_bpe_merges.txt.gz")


class BPETokenizer:
    """CLIP byte-pair tokenizer (same published algorithm as reference open_clip/tokenizer.py:72-186).  The merge rules
    ship with the package (`clip_bpe_merges.txt.gz`: the 48 894 rules of OpenAI CLIP's bpe_simple_vocab_16e6, see
    tools/make_bpe_table.py); `path` / DIFFBIR_BPE_VOCAB may point at the original `bpe_simple_vocab_16e6.txt.gz`."""

    def __init__(self, path: str = None):
        import regex
        path = path or _MERGES
        raw = gzip.open(path).read().decode("utf-8").split("\n")
        if raw and raw[0].startswith("\"bpe_simple_vocab") or (raw and raw[0].startswith("#version")):
            raw = raw[1:]   # the original file has a header line
        lines = [l for l in raw if l][:49152 - 256 - 2]
        merges = [tuple(m.split()) for m in lines]
        base = list(_byte_unicode().values())
        vocab = base + [v + "</w>" for v in base] + ["".join(m) for m in merges] + ["<start_of_text>", "<end_of_text>"]
        self.enc = {t: i for i, t in enumerate(vocab)}
        self.rank = {m: i for i, m in enumerate(merges)}
        self.pat = regex.compile(r"<start_of_text>|<end_of_text>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+",
                                 regex.IGNORECASE)
        self._ws = regex.compile(r"\s+")
        # literal special tokens in a prompt map to themselves (reference tokenizer.py:87 seeds its cache the same way)
        self.cache: Dict[str, List[str]] = {"<start_of_text>": ["<start_of_text>"], "<end_of_text>": ["<end_of_text>"]}

    def _bpe(self, tok: str) -> List[str]:
        if tok in self.cache:
            return self.cache[tok]
        word = list(tok[:-1]) + [tok[-1] + "</w>"]
        while len(word) > 1:
            pairs = {(a, b) for a, b in zip(word, word[1:])}
            best = min(pairs, key=lambda p: self.rank.get(p, float("inf")))
            if best not in self.rank:
                break
            out, i = [], 0
            while i < len(word):
                if i < len(word) - 1 and (word[i], word[i + 1]) == best:
                    out.append(word[i] + word[i + 1])
                    i += 2
                else:
                    out.append(word[i])
                    i += 1
            word = out
        self.cache[tok] = word
        return word

    def encode(self, text: str) -> List[int]:
        text = html.unescape(html.unescape(text)).strip()
        text = self._ws.sub(" ", text).strip().lower()
        ids: List[int] = []
        bu = _byte_unicode()
        for tok in self.pat.findall(text):
            tok = "".join(bu[b] for b in tok.encode("utf-8"))
            ids.extend(self.enc[t] for t in self._bpe(tok))
        return ids


_tokenizer = None


def tokenize(texts: List[str], context_length: int = 77) -> T:
    """reference open_clip/tokenizer.py:159-186. Uses the BPE vocab if available, else a built-in table of the
    reference's default prompts (inference.py:42-52) — any other prompt then raises."""
    global _tokenizer
    if isinstance(texts, str):
        texts = [texts]
    if _tokenizer is None:
        path = os.environ.get("DIFFBIR_BPE_VOCAB")
        if path and os.path.exists(path):
            _tokenizer = BPETokenizer(path)
        elif os.path.exists(_MERGES):
            _tokenizer = BPETokenizer(_MERGES)
    table = None
    out = torch.zeros(len(texts), context_length, dtype=torch.long)
    for i, t in enumerate(texts):
        if _tokenizer is not None:
            ids = _tokenizer.encode(t)
        elif t == "":
            ids = []
        else:
            if table is None:
                with open(_BUILTIN) as f:
                    table = json.load(f)
            if t not in table:
                raise RuntimeError("CLIP BPE vocabulary not found: set DIFFBIR_BPE_VOCAB to bpe_simple_vocab_16e6.txt.gz "
                                   f"(only the built-in default prompts can be tokenized without it); prompt={t!r}")
            ids = table[t]
        ids = [SOT] + list(ids) + [EOT]
        if len(ids) > context_length:
            ids = ids[:context_length]
            ids[-1] = EOT
        out[i, : len(ids)] = torch.tensor(ids)
    return out


class FrozenOpenCLIPEmbedder(NativeModule):
    def __init__(self, embed_dim, vision_cfg=None, text_cfg=None, layer="last"):
        cfg = dict(embed_dim=embed_dim, text_cfg=dict(text_cfg), layer=layer)
        super().__init__(clip_text_spec(cfg))
        self.cfg = cfg
        self.layer_idx = {"last": 0, "penultimate": 1}[layer]

    def _pack(self):
        self.P = {k: v.to(self._device, torch.float32) for k, v in self._sd.items()}

    def set_dtype(self, dtype):  # CLIP stays f32 (reference keeps fp32 weights; loop.py:82)
        return self

    def forward(self, tokens: T) -> T:
        """reference clip.py:37-54: 23 of 24 pre-LN blocks with causal mask, then ln_final. -> f32 [B,77,W]."""
        self._ensure_packed()
        P, t = self.P, self.cfg["text_cfg"]
        heads, L = t["heads"], t["layers"]
        tokens = tokens.to(self._device)
        x = P["model.token_embedding.weight"][tokens] + P["model.positional_embedding"]
        n = x.shape[1]
        mask = torch.full((n, n), float("-inf"), device=x.device).triu_(1)

        def ln(p, v):
            return F.layer_norm(v, (v.shape[-1],), P[p + ".weight"], P[p + ".bias"], 1e-5)

        for i in range(L - self.layer_idx):
            p = f"model.transformer.resblocks.{i}"
            qkv = F.linear(ln(p + ".ln_1", x), P[p + ".attn.in_proj_weight"], P[p + ".attn.in_proj_bias"])
            B, N, C3 = qkv.shape
            q, k, v = qkv.reshape(B, N, 3, heads, C3 // 3 // heads).permute(2, 0, 3, 1, 4)
            o = F.scaled_dot_product_attention(q, k, v, attn_mask=mask).permute(0, 2, 1, 3).reshape(B, N, C3 // 3)
            x = x + F.linear(o, P[p + ".attn.out_proj.weight"], P[p + ".attn.out_proj.bias"])
            hdn = F.gelu(F.linear(ln(p + ".ln_2", x), P[p + ".mlp.c_fc.weight"], P[p + ".mlp.c_fc.bias"]))
            x = x + F.linear(hdn, P[p + ".mlp.c_proj.weight"], P[p + ".mlp.c_proj.bias"])
        return ln("model.ln_final", x)

    __call__ = forward

    def encode(self, text: List[str]) -> T:
        return self(tokenize(text))
