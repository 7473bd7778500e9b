"""OpenCLIP text tower (FrozenOpenCLIPEmbedder, reference clip.py:8-61).

Two calls per pipeline run, 0.05 % of the FLOPs.  Since round 2 the tower runs on the engine's
own HIP kernels (csrc/clip.hip + the MFMA GEMM) instead of PyTorch-ROCm library ops; only the text tower is built
(the reference builds and deletes a ViT-H vision tower, clip.py:21-22).  State-dict keys are those below
`cond_stage_model.` in SD checkpoints (`model.transformer.resblocks.N...`).  The BPE tokenizer is host code.
"""
import gzip
import html
import json
import os
from functools import lru_cache
from typing import Dict, List

import torch

from .. import ops
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
        """Weights of the blocks actually evaluated, packed for the MFMA GEMM in the engine dtype; embeddings and
        LayerNorm affine parameters stay f32 (they feed the f32 residual stream)."""
        t = self.cfg["text_cfg"]
        f32 = lambda k: self._sd[k].to(self._device, torch.float32).contiguous()
        lin = lambda w, b: ops.pack_linear(self._sd[w], self._sd[b], self._dtype, self._device)
        self.tok_emb, self.pos = f32("model.token_embedding.weight"), f32("model.positional_embedding")
        self.blocks = []
        for i in range(t["layers"] - self.layer_idx):
            p = f"model.transformer.resblocks.{i}"
            self.blocks.append(dict(
                ln1=(f32(p + ".ln_1.weight"), f32(p + ".ln_1.bias")), ln2=(f32(p + ".ln_2.weight"), f32(p + ".ln_2.bias")),
                in_proj=lin(p + ".attn.in_proj_weight", p + ".attn.in_proj_bias"),
                out_proj=lin(p + ".attn.out_proj.weight", p + ".attn.out_proj.bias"),
                c_fc=lin(p + ".mlp.c_fc.weight", p + ".mlp.c_fc.bias"),
                c_proj=lin(p + ".mlp.c_proj.weight", p + ".mlp.c_proj.bias")))
        self.ln_final = (f32("model.ln_final.weight"), f32("model.ln_final.bias"))

    def set_dtype(self, dtype):
        """The reference keeps CLIP's weights f32 (loop.py:82) and runs it under the pipeline's autocast
        (loop.py:180): nn.Linear / attention in the 16-bit type, LayerNorm and the residual sums in f32.  Same split
        here: GEMM operands in the engine dtype, f32 accumulation, f32 residual stream."""
        return super().set_dtype(dtype)

    def forward(self, tokens: T) -> T:
        """reference clip.py:37-54: token + positional embedding, 23 of 24 pre-LN blocks with the causal mask, then
        ln_final. tokens int64 [B, 77] -> f32 [B, 77, W].  Every step is a HIP kernel of this engine: dbir_clip_embed,
        dbir_add_layernorm_f32 (residual add + LayerNorm fused), dbir_gemm (in_proj / out_proj / c_fc + GELU /
        c_proj), dbir_causal_attention."""
        self._ensure_packed()
        t = self.cfg["text_cfg"]
        heads, W = t["heads"], t["width"]
        assert W == heads * 64, "engine supports head_dim 64 text towers"
        tokens = tokens.to(self._device, torch.int64).contiguous()
        B, L = tokens.shape
        x = ops.clip_embed(tokens, self.tok_emb, self.pos)          # f32 residual stream [B, L, W]
        y = None
        for blk in self.blocks:
            n = ops.add_layernorm_f32(x, y, blk["ln1"][0], blk["ln1"][1], self._dtype)
            qkv = ops.linear(n.reshape(B * L, W), blk["in_proj"]).reshape(B, L, 3 * W)
            o = ops.causal_attention(qkv, heads, 64 ** -0.5)
            y = ops.linear(o.reshape(B * L, W), blk["out_proj"], out_f32=True).reshape(B, L, W)
            n = ops.add_layernorm_f32(x, y, blk["ln2"][0], blk["ln2"][1], self._dtype)
            hdn = ops.linear(n.reshape(B * L, W), blk["c_fc"], act=ops.ACT_GELU)
            y = ops.linear(hdn, blk["c_proj"], out_f32=True).reshape(B, L, W)
        return ops.add_layernorm_f32(x, y, self.ln_final[0], self.ln_final[1], torch.float32)

    __call__ = forward

    def encode(self, text: List[str]) -> T:
        return self(tokenize(text))
