"""SCUNet stage-1 cleaner (Swin-Conv-UNet; reference diffbir/model/scunet.py:9-264) over the HIP kernels.

NHWC 16-bit activations.  Per ConvTransBlock (scunet.py:136-160):
  * conv1_1 (1x1) is split by output rows into two GEMMs so the conv half and the transformer half are separate dense
    tensors (`torch.split`, scunet.py:153) — no strided conv operand;
  * conv half: conv3x3 -> ReLU -> conv3x3 (+ conv_x) as two fused implicit-GEMM launches (ReLU = LeakyReLU slope 0);
  * transformer half: LayerNorm, qkv GEMM, the engine's fused (shifted-)window attention kernel (window 8, head_dim 32;
    the relative-position parameters [heads, 15, 15] are re-laid out as the [(2ws-1)^2, heads] table it reads, the
    'SW' mask of WMSA.generate_mask is the Swin shift mask at shift = ws/2), projection (+ residual), LayerNorm, MLP
    with GELU in the first GEMM's epilogue (+ residual);
  * both halves land in the two column halves of one buffer (`torch.cat`, scunet.py:157), conv1_2 (1x1) + x.
Down / up sampling: Conv2d(k=2, s=2) = space-to-depth + GEMM (K = 4C); ConvTranspose2d(k=2, s=2) = GEMM (N = 4C) +
depth-to-space (dbir_block2x2).  The U-Net skip sums (`x + x4` ...) are add_scaled launches.
"""
import torch
import torch.nn.functional as F

from .. import ops
from .base import NativeModule
from .specs import scunet_spec, scunet_stages

T = torch.Tensor
WS, HD = 8, 32   # SCUNet hard-codes window_size 8 and head_dim 32 (scunet.py:169-170)


class SCUNet(NativeModule):
    def __init__(self, in_nc=3, config=(2, 2, 2, 2, 2, 2, 2), dim=64, drop_path_rate=0.0, input_resolution=256):
        self.cfg = dict(in_nc=in_nc, config=list(config), dim=dim)
        super().__init__(scunet_spec(self.cfg))
        if dim % 64:
            raise NotImplementedError("SCUNet: dim must be a multiple of 64 (head_dim 32 on dim / 2 channels)")
        # reference Block: type falls back to 'W' when input_resolution <= window_size (scunet.py:110-111) — never for
        # the shipped configuration (256, 128, 64, 32 > 8)
        self.shift_ok = [input_resolution // d > WS for d in (1, 2, 4, 8, 4, 2, 1)]

    def _pack(self):
        dt, dev = self._dtype, self._device
        w = self._w
        lin = lambda wt, b=None: ops.pack_linear(wt, b, dt, dev)
        self.head = ops.pack_conv3x3(w("m_head.0.weight"), None, dt, dev, cin_pad_to=8)
        self.tail = ops.pack_conv3x3(w("m_tail.0.weight"), None, dt, dev)
        self.stages = []
        for si, (name, c, nblk, first) in enumerate(scunet_stages(self.cfg)):
            st = dict(c=c, blocks=[], up=None, down=None)
            if first == 1:   # ConvTranspose2d weight [in, out, ky, kx] -> rows (ky, kx, out)
                wt = w(f"{name}.0.weight")
                st["up"] = lin(wt.permute(2, 3, 1, 0).reshape(4 * wt.shape[1], wt.shape[0]))
            for i in range(nblk):
                p = f"{name}.{i + first}"
                t = f"{p}.trans_block"
                w11, b11 = w(f"{p}.conv1_1.weight").reshape(2 * c, 2 * c), w(f"{p}.conv1_1.bias")
                rel = w(f"{t}.msa.relative_position_params")          # [heads, 2ws-1, 2ws-1]
                st["blocks"].append(dict(
                    conv_in=lin(w11[:c], b11[:c]), trans_in=lin(w11[c:], b11[c:]),
                    cb0=ops.pack_conv3x3(w(f"{p}.conv_block.0.weight"), None, dt, dev),
                    cb2=ops.pack_conv3x3(w(f"{p}.conv_block.2.weight"), None, dt, dev),
                    ln1=(self._f32(f"{t}.ln1.weight"), self._f32(f"{t}.ln1.bias")),
                    ln2=(self._f32(f"{t}.ln2.weight"), self._f32(f"{t}.ln2.bias")),
                    qkv=lin(w(f"{t}.msa.embedding_layer.weight"), w(f"{t}.msa.embedding_layer.bias")),
                    proj=lin(w(f"{t}.msa.linear.weight"), w(f"{t}.msa.linear.bias")),
                    table=rel.float().permute(1, 2, 0).reshape((2 * WS - 1) ** 2, -1).contiguous().to(dev),
                    fc1=lin(w(f"{t}.mlp.0.weight"), w(f"{t}.mlp.0.bias")),
                    fc2=lin(w(f"{t}.mlp.2.weight"), w(f"{t}.mlp.2.bias")),
                    out=lin(w(f"{p}.conv1_2.weight").reshape(2 * c, 2 * c), w(f"{p}.conv1_2.bias")),
                    shift=WS // 2 if (i % 2 == 1 and self.shift_ok[si]) else 0))
            if name.startswith("m_down"):   # Conv2d weight [out, in, ky, kx] -> columns (ky, kx, in)
                wd = w(f"{name}.{nblk}.weight")
                st["down"] = lin(wd.permute(0, 2, 3, 1).reshape(wd.shape[0], 4 * wd.shape[1]))
            self.stages.append(st)

    def _block(self, b: dict, x: T, c: int) -> T:
        B, h, w, _ = x.shape
        conv_x = ops.linear(x, b["conv_in"])
        trans_x = ops.linear(x, b["trans_in"])
        z = torch.empty_like(x)
        t = ops.conv3x3(conv_x, b["cb0"], act=ops.ACT_LRELU, act_param=0.0)
        ops.conv3x3(t, b["cb2"], residual=conv_x, out=z[..., :c])
        n = ops.layernorm(trans_x, b["ln1"][0], b["ln1"][1])
        qkv = ops.linear(n, b["qkv"])
        ao = torch.empty_like(trans_x)
        ops.window_attention(qkv, ao, b["table"], c, c // HD, WS, b["shift"], HD ** -0.5)
        tx = ops.linear(ao, b["proj"], residual=trans_x)
        n = ops.layernorm(tx, b["ln2"][0], b["ln2"][1])
        m = ops.linear(n, b["fc1"], act=ops.ACT_GELU)
        ops.linear(m, b["fc2"], residual=tx, out=z[..., c:])
        return ops.linear(z, b["out"], residual=x)

    def _stage(self, st: dict, x: T) -> T:
        if st["up"] is not None:
            x = ops.depth_to_space2(ops.linear(x, st["up"]))
        for b in st["blocks"]:
            x = self._block(b, x, st["c"])
        if st["down"] is not None:
            x = ops.linear(ops.space_to_depth2(x), st["down"])
        return x

    def forward(self, x0: T) -> T:
        """x0: f32 NCHW [B, in_nc, H, W] -> f32 NCHW same size (scunet.py:229-248)."""
        self._ensure_packed()
        h, w = x0.shape[-2:]
        pb, pr = (64 - h % 64) % 64, (64 - w % 64) % 64
        x0 = x0.float()
        if pb or pr:   # nn.ReplicationPad2d on the 3-channel boundary image (host-side layout plumbing)
            x0 = F.pad(x0, (0, pr, 0, pb), mode="replicate")
        t = ops.nchw_to_nhwc(x0.contiguous(), None, (self.cfg["in_nc"] + 7) // 8 * 8, self._dtype)
        x1 = ops.conv3x3(t, self.head)
        s = self.stages
        x2 = self._stage(s[0], x1)
        x3 = self._stage(s[1], x2)
        x4 = self._stage(s[2], x3)
        x = self._stage(s[3], x4)
        x = self._stage(s[4], ops.add_scaled(x, x4, 1.0))
        x = self._stage(s[5], ops.add_scaled(x, x3, 1.0))
        x = self._stage(s[6], ops.add_scaled(x, x2, 1.0))
        o = ops.conv3x3(ops.add_scaled(x, x1, 1.0), self.tail, out_f32=True)
        return ops.nhwc_to_nchw(o, self.cfg["in_nc"])[..., :h, :w]

    __call__ = forward
