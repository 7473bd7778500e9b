"""SwinIR stage-1 cleaner over the HIP kernels (reference swinir.py:856-894 'nearest+conv' branch, 841-854,
487-488, 245-285, 120-151).

embed_dim 180 is carried as 192 channels (zero padded) so every row is 16-byte aligned for the MFMA GEMM's
vector loads; weights are zero padded accordingly, LayerNorm normalises over the real 180 columns and writes
zeros to the pad.  Window attention (cyclic roll, partition, bias, mask, softmax, PV, reverse) is one fused
kernel; qkv / proj / MLP are MFMA GEMMs with GELU / residual epilogues; the three nearest-x2 upsamples are folded
into the following conv's gather; LeakyReLU into the conv epilogues.
"""
import torch
import torch.nn.functional as F

from .. import ops
from .base import NativeModule
from .specs import swinir_spec

T = torch.Tensor


def _rup(x, m):
    return (x + m - 1) // m * m


class SwinIR(NativeModule):
    def __init__(self, **cfg):
        super().__init__(swinir_spec(cfg))
        self.cfg = dict(cfg)
        self.window_size = cfg["window_size"]
        self.upscale = cfg["sf"]
        self.img_range = cfg["img_range"]
        self.mean = torch.tensor((0.4488, 0.4371, 0.4040)) if cfg["in_chans"] == 3 else torch.zeros(1)

    def _pack(self):
        c = self.cfg
        C = c["embed_dim"]
        Cp = _rup(C, 16)
        self.C, self.Cp = C, Cp
        dt, dev = self._dtype, self._device

        def c3(p, **kw):
            return ops.pack_conv3x3(self._w(p + ".weight"), self._w(p + ".bias"), dt, dev, **kw)

        def lin(p, **kw):
            return ops.pack_linear(self._w(p + ".weight"), self._w(p + ".bias"), dt, dev, **kw)

        def norm(p):
            return (self._f32(p + ".weight"), self._f32(p + ".bias"))

        self.conv_first = c3("conv_first.1", n_pad_to=Cp)
        self.pe_norm = norm("patch_embed.norm")
        self.layers = []
        hidden = int(C * c["mlp_ratio"])
        self.Hp = _rup(hidden, 8)
        for i, (depth, heads) in enumerate(zip(c["depths"], c["num_heads"])):
            blocks = []
            for j in range(depth):
                p = f"layers.{i}.residual_group.blocks.{j}"
                blocks.append(dict(
                    n1=norm(p + ".norm1"), n2=norm(p + ".norm2"),
                    qkv=lin(p + ".attn.qkv", k_pad_to=Cp, n_pad_to=8),
                    proj=lin(p + ".attn.proj", k_pad_to=Cp, n_pad_to=Cp),
                    fc1=lin(p + ".mlp.fc1", k_pad_to=Cp, n_pad_to=8),
                    fc2=lin(p + ".mlp.fc2", k_pad_to=self.Hp, n_pad_to=Cp),
                    table=self._f32(p + ".attn.relative_position_bias_table"),
                    heads=heads, shift=0 if j % 2 == 0 else self.window_size // 2))
            self.layers.append((blocks, c3(f"layers.{i}.conv", cin_pad_to=Cp, n_pad_to=Cp)))
        self.norm = norm("norm")
        self.conv_after_body = c3("conv_after_body", cin_pad_to=Cp, n_pad_to=Cp)
        self.conv_before_upsample = c3("conv_before_upsample.0", cin_pad_to=Cp)
        self.ups = [c3(n) for n in ("conv_up1", "conv_up2", "conv_up3")[: {2: 1, 4: 2, 8: 3}[self.upscale]]]
        self.conv_hr = c3("conv_hr")
        self.conv_last = c3("conv_last")
        self.mean_dev = self.mean.to(dev, torch.float32)

    def _block(self, b: dict, x: T, ao: T) -> T:
        """x: [B,h,w,Cp]; ao: zero-padded scratch [B,h,w,Cp] for the attention output (pad columns stay 0)."""
        C, ws = self.C, self.window_size
        n = ops.layernorm(x, b["n1"][0], b["n1"][1], C)
        qkv = ops.linear(n, b["qkv"])
        ops.window_attention(qkv, ao, b["table"], C, b["heads"], ws, b["shift"], (C // b["heads"]) ** -0.5)
        x = ops.linear(ao, b["proj"], residual=x)
        n = ops.layernorm(x, b["n2"][0], b["n2"][1], C)
        m = ops.linear(n, b["fc1"], act=ops.ACT_GELU)
        return ops.linear(m, b["fc2"], residual=x)

    def forward(self, x: T) -> T:
        """x: f32 NCHW [B,3,H,W] in [0,1] -> f32 NCHW [B,3,H,W]."""
        self._ensure_packed()
        ws, sf = self.window_size, self.upscale
        H0, W0 = x.shape[2:]
        ph, pw = (ws - H0 % ws) % ws, (ws - W0 % ws) % ws
        x = x.float()
        if ph or pw:  # reference swinir.py:834-839; cold path kept on PyTorch
            x = F.pad(x, (0, pw, 0, ph), "reflect")
        B, _, H, W = x.shape
        if (H // sf) % ws or (W // sf) % ws or H % sf or W % sf:
            raise ValueError(f"SwinIR input {H}x{W} must be a multiple of {sf * ws} (the pipeline pads to 64)")
        t0 = ops.pixel_unshuffle(x.contiguous(), sf, _rup(3 * sf * sf, 8), self.mean_dev, self.img_range, self._dtype)
        x0 = ops.conv3x3(t0, self.conv_first)
        t = ops.layernorm(x0, self.pe_norm[0], self.pe_norm[1], self.C)
        ao = torch.zeros_like(t)
        for blocks, conv in self.layers:
            r = t
            for b in blocks:
                r = self._block(b, r, ao)
            t = ops.conv3x3(r, conv, residual=t)
        t = ops.layernorm(t, self.norm[0], self.norm[1], self.C)
        h = ops.conv3x3(t, self.conv_after_body, residual=x0)
        h = ops.conv3x3(h, self.conv_before_upsample, act=ops.ACT_LRELU, act_param=0.01)
        for u in self.ups:
            h = ops.conv3x3(h, u, upsample=True, act=ops.ACT_LRELU, act_param=0.2)
        h = ops.conv3x3(h, self.conv_hr, act=ops.ACT_LRELU, act_param=0.2)
        o = ops.conv3x3(h, self.conv_last, out_f32=True)
        out = ops.nhwc_to_nchw(o, 3, scale=1.0 / self.img_range, shift=self.mean_dev)
        return out[:, :, : H0 * sf, : W0 * sf]

    __call__ = forward
