"""AutoencoderKL encode / decode over the HIP kernels (reference vae.py:306-582, distributions.py:24-62).

NHWC 16-bit; every conv is a fused implicit-GEMM launch, GroupNorm(eps 1e-6)+swish is one fused kernel chain,
the (0,1,0,1)-padded stride-2 downsample (vae.py:50-54) and the nearest-x2 upsample (vae.py:34) are address math
inside the conv gather.  The mid-block attention is single-head with d = C = 512: it runs as two batched MFMA
GEMMs (Q K^T, P V) around a row-softmax kernel, over QUERY CHUNKS of at most `ATTN_CHUNK_BYTES` of scores: exact
attention (every query row sees all keys), but never an [L, L] matrix — L = 4096 at 512x512 is one chunk, L = 262144
at 4096x4096 (137 GB of scores if materialised) runs in 4 GB pieces.
"""
from typing import Optional

import torch

from .. import ops
from .base import NativeModule
from .specs import vae_spec

T = torch.Tensor


ATTN_CHUNK_BYTES = 4 << 30   # upper bound of the per-chunk score buffer [B, Lq_chunk, L] (16-bit)


class _VRes:
    __slots__ = ("gn1", "conv1", "gn2", "conv2", "nin")


class AutoencoderKL(NativeModule):
    def __init__(self, ddconfig: dict, embed_dim: int):
        cfg = dict(ddconfig=dict(ddconfig), embed_dim=embed_dim)
        super().__init__(vae_spec(cfg))
        self.cfg = cfg
        self.embed_dim = embed_dim

    # ------------------------------------------------------------------ packing
    def _n(self, p):
        return (self._f32(p + ".weight"), self._f32(p + ".bias"))

    def _c3(self, p, **kw):
        return ops.pack_conv3x3(self._w(p + ".weight"), self._w(p + ".bias"), self._dtype, self._device, **kw)

    def _c1(self, p, **kw):
        return ops.pack_linear(self._w(p + ".weight"), self._w(p + ".bias"), self._dtype, self._device, **kw)

    def _res(self, p, cin, cout) -> _VRes:
        r = _VRes()
        r.gn1, r.conv1 = self._n(p + ".norm1"), self._c3(p + ".conv1")
        r.gn2, r.conv2 = self._n(p + ".norm2"), self._c3(p + ".conv2")
        r.nin = self._c1(p + ".nin_shortcut") if cin != cout else None
        return r

    def _attn(self, p):
        wqk = torch.cat([self._w(p + ".q.weight"), self._w(p + ".k.weight")], dim=0)
        bqk = torch.cat([self._w(p + ".q.bias"), self._w(p + ".k.bias")], dim=0)
        return dict(gn=self._n(p + ".norm"), qk=ops.pack_linear(wqk, bqk, self._dtype, self._device),
                    v=self._c1(p + ".v"), proj=self._c1(p + ".proj_out"))

    def _pack(self):
        dd = self.cfg["ddconfig"]
        ch, mult, nrb = dd["ch"], list(dd["ch_mult"]), dd["num_res_blocks"]
        nlev = len(mult)
        e = "encoder"
        self.e_conv_in = self._c3(f"{e}.conv_in", cin_pad_to=8)
        self.e_down = []
        in_mult = [1] + mult
        bi = ch
        for l in range(nlev):
            bi, bo = ch * in_mult[l], ch * mult[l]
            blocks = []
            for b in range(nrb):
                blocks.append(self._res(f"{e}.down.{l}.block.{b}", bi, bo))
                bi = bo
            ds = self._c3(f"{e}.down.{l}.downsample.conv") if l != nlev - 1 else None
            self.e_down.append((blocks, ds))
        self.e_mid = (self._res(f"{e}.mid.block_1", bi, bi), self._attn(f"{e}.mid.attn_1"),
                      self._res(f"{e}.mid.block_2", bi, bi))
        self.e_norm_out = self._n(f"{e}.norm_out")
        self.e_conv_out = self._c3(f"{e}.conv_out", n_pad_to=8)
        self.quant = self._c1("quant_conv")
        d = "decoder"
        bi = ch * mult[-1]
        self.post_quant = self._c1("post_quant_conv", n_pad_to=8)
        self.d_conv_in = self._c3(f"{d}.conv_in", cin_pad_to=8)
        self.d_mid = (self._res(f"{d}.mid.block_1", bi, bi), self._attn(f"{d}.mid.attn_1"),
                      self._res(f"{d}.mid.block_2", bi, bi))
        self.d_up = []
        for l in reversed(range(nlev)):
            bo = ch * mult[l]
            blocks = []
            for b in range(nrb + 1):
                blocks.append(self._res(f"{d}.up.{l}.block.{b}", bi, bo))
                bi = bo
            us = self._c3(f"{d}.up.{l}.upsample.conv") if l != 0 else None
            self.d_up.append((blocks, us))
        self.d_norm_out = self._n(f"{d}.norm_out")
        self.d_conv_out = self._c3(f"{d}.conv_out")
        self.z_channels, self.out_ch = dd["z_channels"], dd["out_ch"]

    # ------------------------------------------------------------------ blocks
    def _run_res(self, r: _VRes, x: T) -> T:
        h = ops.groupnorm(x, r.gn1[0], r.gn1[1], 1e-6, True)
        h = ops.conv3x3(h, r.conv1)
        h = ops.groupnorm(h, r.gn2[0], r.gn2[1], 1e-6, True)
        skip = x if r.nin is None else ops.linear(x, r.nin)
        return ops.conv3x3(h, r.conv2, residual=skip)

    def _run_attn(self, a: dict, x: T) -> T:
        B, H, W, C = x.shape
        L = H * W
        Lp = (L + 63) // 64 * 64
        hn = ops.groupnorm(x, a["gn"][0], a["gn"][1], 1e-6, False).reshape(B * L, C)
        qk = ops.linear(hn, a["qk"]).reshape(B, L, 2 * C)
        vt = torch.zeros((B, C, Lp), dtype=x.dtype, device=x.device) if Lp != L else \
            torch.empty((B, C, Lp), dtype=x.dtype, device=x.device)
        ops.linear_t(hn, a["v"], L, vt)
        o = torch.empty((B, L, C), dtype=x.dtype, device=x.device)
        rows = max(64, min(L, (ATTN_CHUNK_BYTES // (2 * B * Lp)) // 64 * 64))   # query rows per chunk
        s = torch.empty((B, rows, Lp), dtype=x.dtype, device=x.device)
        for q0 in range(0, L, rows):
            n = min(rows, L - q0)
            sc = s if n == rows else s.reshape(-1)[: B * n * Lp].view(B, n, Lp)   # dense [B, n, Lp] in the same memory
            ops.bmm_nt(qk[:, q0:q0 + n, :C], qk[..., C:], sc[..., :L] if Lp != L else sc, out_scale=float(C) ** -0.5)
            ops.softmax_rows_(sc, L)
            ops.bmm_nt(sc, vt, o[:, q0:q0 + n])
        return ops.linear(o.reshape(B, H, W, C), a["proj"], residual=x)

    # ------------------------------------------------------------------ API
    def encode_moments(self, x: T, in_scale: float = 1.0, in_shift: float = 0.0) -> T:
        """x f32 NCHW [B,3,H,W] -> moments NHWC f32 [B,h,w,2*embed] (vae.py:401-426, 573-577)."""
        self._ensure_packed()
        h = ops.nchw_to_nhwc(x.float().contiguous(), None, 8, self._dtype, in_scale, in_shift)
        h = ops.conv3x3(h, self.e_conv_in)
        for blocks, ds in self.e_down:
            for r in blocks:
                h = self._run_res(r, h)
            if ds is not None:
                h = ops.conv3x3(h, ds, stride=2, pad=0, out_hw=(h.shape[1] // 2, h.shape[2] // 2))
        h = self._run_res(self.e_mid[0], h)
        h = self._run_attn(self.e_mid[1], h)
        h = self._run_res(self.e_mid[2], h)
        h = ops.groupnorm(h, self.e_norm_out[0], self.e_norm_out[1], 1e-6, True)
        h = ops.conv3x3(h, self.e_conv_out)
        return ops.linear(h, self.quant, out_f32=True)

    def encode_mode(self, x: T, scale_factor: float, in_scale: float = 1.0, in_shift: float = 0.0) -> T:
        """mode() of the diagonal Gaussian = mean = first z_channels of the moments, times scale_factor."""
        m = self.encode_moments(x, in_scale, in_shift)
        return ops.nhwc_to_nchw(m, self.z_channels, scale=scale_factor)

    def decode(self, z: T, in_scale: float = 1.0, out_scale: float = 1.0, out_shift: Optional[T] = None) -> T:
        """z f32 NCHW [B,4,h,w] -> f32 NCHW image (vae.py:579-582, 526-559)."""
        self._ensure_packed()
        h = ops.nchw_to_nhwc(z.float().contiguous(), None, 8, self._dtype, in_scale, 0.0)
        h = ops.linear(h, self.post_quant)
        h = ops.conv3x3(h, self.d_conv_in)
        h = self._run_res(self.d_mid[0], h)
        h = self._run_attn(self.d_mid[1], h)
        h = self._run_res(self.d_mid[2], h)
        for blocks, us in self.d_up:
            for r in blocks:
                h = self._run_res(r, h)
            if us is not None:
                h = ops.conv3x3(h, us, upsample=True)
        h = ops.groupnorm(h, self.d_norm_out[0], self.d_norm_out[1], 1e-6, True)
        o = ops.conv3x3(h, self.d_conv_out, out_f32=True)
        return ops.nhwc_to_nchw(o, self.out_ch, scale=out_scale, shift=out_shift)
