"""AutoencoderKL encode / decode over the HIP kernels (reference vae.py:306-582, distributions.py:24-62).

NHWC 16-bit; every conv is a fused implicit-GEMM launch, GroupNorm(eps 1e-6)+swish is one fused kernel chain,
the (0,1,0,1)-padded stride-2 downsample (vae.py:50-54) and the nearest-x2 upsample (vae.py:34) are address math
inside the conv gather.  The mid-block attention is single-head with d = C = 512: it runs as two batched MFMA
GEMMs (Q K^T, P V) around a row-softmax kernel, over QUERY CHUNKS of at most `ATTN_CHUNK_BYTES` of scores: exact
attention (every query row sees all keys), but never an [L, L] matrix — L = 4096 at 512x512 is one chunk, L = 262144
at 4096x4096 (137 GB of scores if materialised) runs in 4 GB pieces.
"""
import math
from typing import List, Optional

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
        return self._attn_core(a, ops.groupnorm(x, a["gn"][0], a["gn"][1], 1e-6, False), x)

    def _attn_core(self, a: dict, hn: T, x: T) -> T:
        """single-head attention on the normalised input `hn`, residual `x` (vae.py:253-282)."""
        B, H, W, C = x.shape
        L = H * W
        Lp = (L + 63) // 64 * 64
        hn = hn.reshape(B * L, C)
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

    # ------------------------------------------------------------------ tiled execution (reference utils/tilevae/tilevae.py)
    # The reference's VAEHook (non-fast mode, what cldm.py:99-111,127-138 selects) cuts the input into tiles padded by
    # 32 px (encoder) / 11 latent px (decoder), runs the network tile by tile, and keeps the result close to the untiled
    # one by sharing GroupNorm statistics: at every GroupNorm all tiles stop, their per-(sample, group) mean and variance
    # are averaged with weights proportional to the tile's pixel count, and every tile is normalised with the averages.
    # Convolutions see each tile's own zero padding (the padded ring is cropped away at the end) and the mid-block
    # attention runs inside each tile.  The engine reproduces exactly that layer-synchronous algorithm: every layer is
    # applied to all tiles before the next one (same result as the reference's task-queue zig-zag, which only orders
    # host <-> device traffic), so its output tracks the reference's TILED output, not the untiled one.
    @staticmethod
    def _best_tile_size(lower: int, upper: int) -> int:
        """tilevae.py:325-338: the smallest size >= lower that is a multiple of 32 / 16 / ... / 2 and <= upper."""
        div = 32
        while div >= 2:
            rem = lower % div
            if rem == 0:
                return lower
            cand = lower - rem + div
            if cand <= upper:
                return cand
            div //= 2
        return lower

    @classmethod
    def split_tiles(cls, h: int, w: int, tile_size: int, pad: int, is_decoder: bool):
        """tilevae.py:340-398 -> (input boxes, output boxes), boxes = [x1, x2, y1, y2]."""
        nh = max(math.ceil((h - 2 * pad) / tile_size), 1)
        nw = max(math.ceil((w - 2 * pad) / tile_size), 1)
        th = cls._best_tile_size(math.ceil((h - 2 * pad) / nh), tile_size)
        tw = cls._best_tile_size(math.ceil((w - 2 * pad) / nw), tile_size)
        ins, outs = [], []
        for i in range(nh):
            for j in range(nw):
                box = [pad + j * tw, min(pad + (j + 1) * tw, w), pad + i * th, min(pad + (i + 1) * th, h)]
                ob = [box[0] if box[0] > pad else 0, box[1] if box[1] < w - pad else w,
                      box[2] if box[2] > pad else 0, box[3] if box[3] < h - pad else h]
                outs.append([v * 8 if is_decoder else v // 8 for v in ob])
                ins.append([max(0, box[0] - pad), min(w, box[1] + pad), max(0, box[2] - pad), min(h, box[3] + pad)])
        return ins, outs

    # ---- tile sharding (one process per GPU; diffbir_amd.parallel.enable_tile_sharding) --------------------------
    # tile_shard = (rank, world): this rank runs tiles rank::world of every tiled encode / decode.  The only cross-tile
    # couplings of the algorithm are the averaged GroupNorm statistics (2*groups floats per sample and layer) and the
    # final paste: the statistics are summed over ranks with `tile_all_reduce` (a tiny all-reduce per GroupNorm layer),
    # the pasted result (disjoint regions, zeros elsewhere) with one all-reduce at the end.
    tile_shard = None
    tile_all_reduce = None

    class _Tiles:
        """Bookkeeping of one tiled pass: global indices of the tiles held by this rank and the CURRENT (h, w) of every
        tile of the image (all ranks track all shapes: the GroupNorm weights are pixel counts of all tiles)."""

        def __init__(self, hw: List[tuple], shard, reduce):
            self.hw = list(hw)
            n = len(hw)
            self.idx = list(range(n)) if shard is None else list(range(shard[0], n, shard[1]))
            self.reduce = reduce if shard is not None else None

        def scale(self, f):
            self.hw = [f(h, w) for h, w in self.hw]

    def _gn_tiles(self, tiles: List[T], gn, silu: bool, tc: "AutoencoderKL._Tiles") -> List[T]:
        """GroupNorm over a list of tiles with pixel-weighted averaged statistics (tilevae.py:241-279)."""
        if len(tc.hw) == 1 and tc.reduce is None:
            return [ops.groupnorm(tiles[0], gn[0], gn[1], 1e-6, silu)]
        dev = self._device
        px = torch.tensor([h * w for h, w in tc.hw], dtype=torch.float32, device=dev)
        wgt = px / px.max()
        wgt = wgt / wgt.sum()
        if tiles:
            assert all(tuple(t.shape[1:3]) == tuple(tc.hw[i]) for t, i in zip(tiles, tc.idx))
            stats = torch.stack([ops.groupnorm_stats(t) for t in tiles], dim=0)
            mv = (stats * wgt[tc.idx][:, None, None]).sum(dim=0).contiguous()
        else:   # more ranks than tiles: this rank only takes part in the reductions
            mv = torch.zeros((self._tile_batch, 64), dtype=torch.float32, device=dev)
        if tc.reduce is not None:
            mv = tc.reduce(mv)
        return [ops.groupnorm_apply(t, gn[0], gn[1], mv, 1e-6, silu) for t in tiles]

    def _res_tiles(self, r: _VRes, tiles: List[T], tc) -> List[T]:
        h = self._gn_tiles(tiles, r.gn1, True, tc)
        h = [ops.conv3x3(t, r.conv1) for t in h]
        h = self._gn_tiles(h, r.gn2, True, tc)
        skip = tiles if r.nin is None else [ops.linear(t, r.nin) for t in tiles]
        return [ops.conv3x3(t, r.conv2, residual=sk) for t, sk in zip(h, skip)]

    def _attn_tiles(self, a: dict, tiles: List[T], tc) -> List[T]:
        hn = self._gn_tiles(tiles, a["gn"], False, tc)
        return [self._attn_core(a, n, x) for n, x in zip(hn, tiles)]

    def _paste(self, tiles: List[T], ins, outs, is_decoder: bool, B: int, H: int, W: int, tc, channels: int, dtype) -> T:
        """crop_valid_region + write into the result (tilevae.py:218-229, 545-547); NHWC.  Sharded: every rank pastes
        its tiles into zeros and the (disjoint) partial results are summed."""
        res = torch.zeros((B, H, W, channels), dtype=torch.float32 if tc.reduce is not None else dtype, device=self._device)
        for t, i in zip(tiles, tc.idx):
            ib, ob = ins[i], outs[i]
            pb = [v * 8 if is_decoder else v // 8 for v in ib]
            m = [ob[k] - pb[k] for k in range(4)]
            res[:, ob[2]:ob[3], ob[0]:ob[1]] = t[:, m[2]:t.shape[1] + m[3], m[0]:t.shape[2] + m[1]]
        if tc.reduce is not None:
            res = tc.reduce(res).to(dtype)
        return res

    def encode_moments_tiled(self, x: T, tile_size: int, in_scale: float = 1.0, in_shift: float = 0.0) -> T:
        """Tiled encoder -> moments NHWC f32 (VAEHook(encoder) then quant_conv, cldm.py:99-111)."""
        self._ensure_packed()
        B, _, H, W = x.shape
        pad = 32
        if max(H, W) <= pad * 2 + tile_size:
            print("[Tiled VAE]: the input size is tiny and unnecessary to tile.")
            return self.encode_moments(x, in_scale, in_shift)
        ins, outs = self.split_tiles(H, W, tile_size, pad, False)
        tc = self._Tiles([(b[3] - b[2], b[1] - b[0]) for b in ins], self.tile_shard, self.tile_all_reduce)
        self._tile_batch = B
        x = x.float()
        t = [ops.nchw_to_nhwc(x[:, :, ins[i][2]:ins[i][3], ins[i][0]:ins[i][1]].contiguous(), None, 8, self._dtype, in_scale,
                              in_shift) for i in tc.idx]
        t = [ops.conv3x3(v, self.e_conv_in) for v in t]
        for blocks, ds in self.e_down:
            for r in blocks:
                t = self._res_tiles(r, t, tc)
            if ds is not None:
                t = [ops.conv3x3(v, ds, stride=2, pad=0, out_hw=(v.shape[1] // 2, v.shape[2] // 2)) for v in t]
                tc.scale(lambda h, w: (h // 2, w // 2))
        t = self._res_tiles(self.e_mid[0], t, tc)
        t = self._attn_tiles(self.e_mid[1], t, tc)
        t = self._res_tiles(self.e_mid[2], t, tc)
        t = self._gn_tiles(t, self.e_norm_out, True, tc)
        t = [ops.conv3x3(v, self.e_conv_out) for v in t]
        h = self._paste(t, ins, outs, False, B, H // 8, W // 8, tc, self.e_conv_out.n_out, self._dtype)
        return ops.linear(h, self.quant, out_f32=True)

    def decode_tiled(self, z: T, tile_size: int, in_scale: float = 1.0) -> T:
        """post_quant_conv then the tiled decoder (cldm.py:127-138); tile_size in latent pixels."""
        self._ensure_packed()
        B, _, H, W = z.shape
        pad = 11
        if max(H, W) <= pad * 2 + tile_size:
            print("[Tiled VAE]: the input size is tiny and unnecessary to tile.")
            return self.decode(z, in_scale=in_scale)
        h = ops.nchw_to_nhwc(z.float().contiguous(), None, 8, self._dtype, in_scale, 0.0)
        h = ops.linear(h, self.post_quant)
        ins, outs = self.split_tiles(H, W, tile_size, pad, True)
        tc = self._Tiles([(b[3] - b[2], b[1] - b[0]) for b in ins], self.tile_shard, self.tile_all_reduce)
        self._tile_batch = B
        t = [h[:, ins[i][2]:ins[i][3], ins[i][0]:ins[i][1]].contiguous() for i in tc.idx]
        t = [ops.conv3x3(v, self.d_conv_in) for v in t]
        t = self._res_tiles(self.d_mid[0], t, tc)
        t = self._attn_tiles(self.d_mid[1], t, tc)
        t = self._res_tiles(self.d_mid[2], t, tc)
        for blocks, us in self.d_up:
            for r in blocks:
                t = self._res_tiles(r, t, tc)
            if us is not None:
                t = [ops.conv3x3(v, us, upsample=True) for v in t]
                tc.scale(lambda h, w: (2 * h, 2 * w))
        t = self._gn_tiles(t, self.d_norm_out, True, tc)
        t = [ops.conv3x3(v, self.d_conv_out, out_f32=True) for v in t]
        o = self._paste(t, ins, outs, True, B, H * 8, W * 8, tc, self.d_conv_out.n_out, torch.float32)
        return ops.nhwc_to_nchw(o, self.out_ch)

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

    def encode_mode(self, x: T, scale_factor: float, in_scale: float = 1.0, in_shift: float = 0.0,
                    tile_size: int = 0) -> T:
        """mode() of the diagonal Gaussian = mean = first z_channels of the moments, times scale_factor.
        tile_size > 0: the reference's tiled encoder."""
        m = (self.encode_moments_tiled(x, tile_size, in_scale, in_shift) if tile_size > 0
             else self.encode_moments(x, in_scale, in_shift))
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
