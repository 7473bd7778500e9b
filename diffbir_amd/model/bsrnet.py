"""BSRNet stage-1 cleaner (RRDBNet from BSRGAN; reference diffbir/model/bsrnet.py:36-104) over the HIP kernels.

Every layer is a 3x3 convolution -> one fused MFMA implicit-GEMM launch (bias, LeakyReLU(0.2), `* 0.2 + x` residual,
nearest-x2 upsample folded into the gather).  The dense connections of ResidualDenseBlock_5C (`torch.cat((x, x1, ..))`,
bsrnet.py:50-55) never materialise: one [B, H, W, nf + 4*gc] buffer per block holds x | x1 | x2 | x3 | x4, conv_k
writes its gc output channels into its column slice, and every conv reads the whole buffer with its weight zero-padded
over the channels that are not produced yet (those columns hold finite stale values, multiplied by exact zeros) —
Cin = 192 keeps every launch on the direct-to-LDS 64-channel-slice kernels.
"""
import torch

from .. import ops
from .base import NativeModule
from .specs import bsrnet_spec

T = torch.Tensor


class RRDBNet(NativeModule):
    def __init__(self, in_nc=3, out_nc=3, nf=64, nb=23, gc=32, sf=4):
        self.cfg = dict(in_nc=in_nc, out_nc=out_nc, nf=nf, nb=nb, gc=gc, sf=sf)
        super().__init__(bsrnet_spec(self.cfg))
        self.sf = sf
        if nf % 8 or gc % 8:
            raise NotImplementedError("RRDBNet: nf and gc must be multiples of 8 (16-byte channel slices)")

    def _pack(self):
        c = self.cfg
        nf, gc = c["nf"], c["gc"]
        self.wide = nf + 4 * gc
        dt, dev = self._dtype, self._device

        def c3(p, **kw):
            return ops.pack_conv3x3(self._w(p + ".weight"), self._w(p + ".bias"), dt, dev, **kw)

        self.conv_first = c3("conv_first", cin_pad_to=8)
        self.trunk = []
        for i in range(c["nb"]):
            rdbs = []
            for r in (1, 2, 3):
                p = f"RRDB_trunk.{i}.RDB{r}"
                rdbs.append([c3(f"{p}.conv{k}", cin_pad_to=self.wide) for k in (1, 2, 3, 4, 5)])
            self.trunk.append(rdbs)
        # trunk_conv reads the nf leading channels of a block buffer (the rest is weighted by zeros)
        self.trunk_conv = c3("trunk_conv", cin_pad_to=self.wide)
        self.ups = [c3("upconv1")] + ([c3("upconv2")] if c["sf"] == 4 else [])
        self.hr, self.last = c3("HRconv"), c3("conv_last")

    def forward(self, x: T) -> T:
        """x: f32 NCHW [B, in_nc, H, W] -> f32 NCHW [B, out_nc, sf*H, sf*W]   (bsrnet.py:94-104)."""
        self._ensure_packed()
        c = self.cfg
        nf, gc = c["nf"], c["gc"]
        B, _, H, W = x.shape
        t = ops.nchw_to_nhwc(x.float().contiguous(), None, (c["in_nc"] + 7) // 8 * 8, self._dtype)
        fea = ops.conv3x3(t, self.conv_first)
        bufs = [torch.zeros((B, H, W, self.wide), dtype=self._dtype, device=x.device) for _ in range(4)]
        cur = 0
        bufs[cur][..., :nf].copy_(fea)
        tmp = torch.empty((B, H, W, nf), dtype=self._dtype, device=x.device)
        for rdbs in self.trunk:
            rrdb_in = bufs[cur]                    # its first nf columns stay untouched until this RRDB is done
            b = rrdb_in
            for r, convs in enumerate(rdbs):
                for k in range(4):                 # x_k = lrelu(conv_k(cat(x, x1..x_{k-1})))
                    ops.conv3x3(b, convs[k], act=ops.ACT_LRELU, act_param=0.2, out=b[..., nf + k * gc: nf + (k + 1) * gc])
                nxt = bufs[(cur + 1 + r) % 4]
                if r < 2:                          # x5 * 0.2 + x  -> input of the next dense block
                    ops.conv3x3(b, convs[4], out_scale=0.2, residual=b[..., :nf], out=nxt[..., :nf])
                else:                              # (x5 * 0.2 + x) * 0.2 + rrdb_in   (bsrnet.py:71)
                    ops.conv3x3(b, convs[4], out_scale=0.2, residual=b[..., :nf], out=tmp)
                    ops.add_scaled(rrdb_in[..., :nf], tmp, 0.2, out=nxt[..., :nf])
                b = nxt
            cur = (cur + 3) % 4
        fea = ops.conv3x3(bufs[cur], self.trunk_conv, residual=fea)
        for u in self.ups:
            fea = ops.conv3x3(fea, u, upsample=True, act=ops.ACT_LRELU, act_param=0.2)
        fea = ops.conv3x3(fea, self.hr, act=ops.ACT_LRELU, act_param=0.2)
        o = ops.conv3x3(fea, self.last, out_f32=True)
        return ops.nhwc_to_nchw(o, c["out_nc"])

    __call__ = forward
