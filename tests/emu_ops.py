"""TEST DOUBLE for `diffbir_amd.ops` — plain PyTorch (f32 math) implementations with identical signatures,
layouts and rounding points (results are cast to the 16-bit activation dtype where the kernels store 16-bit).

Two uses (tests only, never imported by the product):
  * `-m gpu` tests compare every HIP kernel against the function of the same name here on the same inputs;
  * `-m "not gpu"` tests monkeypatch `diffbir_amd.ops` with this module (``install()``) so that the host-side
    orchestration (weight packing, layouts, fusion bookkeeping) is validated on CPU against the oracle.
"""
import math
from typing import Optional, Tuple

import torch
import torch.nn.functional as F

from diffbir_amd import ops as real_ops
from diffbir_amd.ops import (ACT_GEGLU, ACT_GELU, ACT_LRELU, ACT_NONE, ACT_SILU, PackedWeight,  # noqa: F401
                             pack_conv3x3, pack_geglu, pack_linear)

T = torch.Tensor


def _act(v: T, act: int, p: float) -> T:
    if act == ACT_SILU:
        return F.silu(v)
    if act == ACT_GELU:
        return F.gelu(v)
    if act == ACT_LRELU:
        return F.leaky_relu(v, p)
    return v


def _epilogue(acc: T, pw: Optional[PackedWeight], act, act_param, out_scale, residual, rowvec, rows_per_batch):
    """acc: f32 [M, N_packed]"""
    M = acc.shape[0]
    if pw is not None and pw.bias is not None:
        acc = acc + pw.bias.float()[None, : acc.shape[1]]
    if rowvec is not None:
        idx = torch.arange(M, device=acc.device) // rows_per_batch
        acc = acc + rowvec.float()[idx][:, : acc.shape[1]]
    if act == ACT_GEGLU:
        n2 = acc.shape[1]
        blk = acc.reshape(M, n2 // 64, 2, 32)
        acc = (blk[:, :, 0] * F.gelu(blk[:, :, 1])).reshape(M, n2 // 2)
    else:
        acc = _act(acc, act, act_param)
    acc = acc * out_scale
    if residual is not None:
        acc = acc + residual.reshape(M, -1).float()[:, : acc.shape[1]]
    return acc


def linear(x, pw, out=None, act=ACT_NONE, act_param=0.0, out_scale=1.0, residual=None, rowvec=None,
           rows_per_batch=0, out_f32=False, tile=0, stats=False):
    if stats:  # the test double has no epilogue statistics: consumers then compute them from the tensor (None)
        return linear(x, pw, out, act, act_param, out_scale, residual, rowvec, rows_per_batch, out_f32, tile), None
    K = x.shape[-1]
    assert K == pw.K
    if pw.geglu:
        act = ACT_GEGLU
    M = x.numel() // K if x.is_contiguous() else real_ops._rows(x)
    acc = x.reshape(-1, K).float() @ pw.w[: pw.N, :K].float().t()
    res = _epilogue(acc, pw, act, act_param, out_scale, residual, rowvec, rows_per_batch)[:, : pw.n_out]
    dt = torch.float32 if out_f32 else x.dtype
    if out is None:
        return res.to(dt).reshape(x.shape[:-1] + (pw.n_out,))
    out.copy_(res.to(dt).reshape(out.shape))
    return out


def linear_t(x, pw, L, out_t, tile=0):
    K = x.shape[-1]
    acc = x.reshape(-1, K).float() @ pw.w[: pw.N, :K].float().t()
    if pw.bias is not None:
        acc = acc + pw.bias.float()[None]
    Bz = acc.shape[0] // L
    out_t[:, :, :L] = acc.reshape(Bz, L, pw.N).permute(0, 2, 1).to(out_t.dtype)
    return out_t


def conv3x3(x, pw, stride=1, pad=1, upsample=False, out=None, act=ACT_NONE, act_param=0.0, out_scale=1.0,
            residual=None, rowvec=None, out_f32=False, out_hw=None, tile=0, stats=False):
    if stats:
        return conv3x3(x, pw, stride, pad, upsample, out, act, act_param, out_scale, residual, rowvec, out_f32, out_hw,
                       tile), None
    B, Hi, Wi, Cin = x.shape
    assert Cin == pw.cin
    xi = x.float().permute(0, 3, 1, 2)
    if upsample:
        xi = F.interpolate(xi, scale_factor=2, mode="nearest")
    Hv, Wv = xi.shape[2:]
    if out_hw is None:
        Ho, Wo = (Hv + 2 * pad - 3) // stride + 1, (Wv + 2 * pad - 3) // stride + 1
    else:
        Ho, Wo = out_hw
    # pad right/bottom enough for the requested output extent (zeros), left/top by `pad`
    need_h = (Ho - 1) * stride + 3 - Hv - pad
    need_w = (Wo - 1) * stride + 3 - Wv - pad
    xi = F.pad(xi, (pad, max(need_w, 0), pad, max(need_h, 0)))
    w = pw.w[: pw.N, : 9 * Cin].float().reshape(pw.N, 3, 3, Cin).permute(0, 3, 1, 2)
    y = F.conv2d(xi, w, None, stride=stride)[:, :, :Ho, :Wo]
    acc = y.permute(0, 2, 3, 1).reshape(B * Ho * Wo, pw.N)
    res = _epilogue(acc, pw, act, act_param, out_scale, residual, rowvec, Ho * Wo)[:, : pw.n_out]
    dt = torch.float32 if out_f32 else x.dtype
    res = res.to(dt).reshape(B, Ho, Wo, pw.n_out)
    if out is None:
        return res
    out.copy_(res)
    return out


def conv3x3_up4(x, pw, out=None, act=ACT_NONE, act_param=0.0, out_scale=1.0):
    """The parity-collapsed statement of nearest-x2 upsample + conv3x3 on the PACKED up4 weights (ops.pack_conv3x3_up4):
    out[b, 2i+a, 2j+c] = sum_{ty,tx} x[b, i+a-1+ty, j+c-1+tx] . W[2a+c][:, ty, tx]  (zero outside the input)."""
    u = pw.up4
    B, Hi, Wi, Cin = x.shape
    N, Wrows = u.N, u.w.shape[0] // 4
    w4 = u.w.float().reshape(4, Wrows, 2, 2, Cin)[:, :N].permute(0, 1, 4, 2, 3)          # [4, N, Cin, 2, 2]
    xi = F.pad(x.float().permute(0, 3, 1, 2), (1, 1, 1, 1))                               # low-res pixel (i, j) at (i+1, j+1)
    y = torch.zeros((B, N, 2 * Hi, 2 * Wi), dtype=torch.float32, device=x.device)
    for a in range(2):
        for c in range(2):
            full = F.conv2d(xi, w4[2 * a + c])                                            # [B, N, Hi+1, Wi+1], (r, s) <-> taps (r.., s..)
            y[:, :, a::2, c::2] = full[:, :, a:a + Hi, c:c + Wi]
    acc = y.permute(0, 2, 3, 1).reshape(B * 4 * Hi * Wi, N)
    res = _epilogue(acc, pw, act, act_param, out_scale, None, None, 4 * Hi * Wi)[:, : pw.n_out]
    res = res.to(x.dtype).reshape(B, 2 * Hi, 2 * Wi, pw.n_out)
    if out is None:
        return res
    out.copy_(res)
    return out


def bmm_nt(a, b, out, out_scale=1.0):
    out.copy_((torch.bmm(a.float(), b.float().transpose(1, 2)) * out_scale).to(out.dtype))
    return out


def attention(q, k, vt, out, heads, Lk, scale):
    B, Lq = q.shape[:2]
    d = 64
    qh = q[..., : heads * d].float().reshape(B, Lq, heads, d).permute(0, 2, 1, 3)
    kh = k[:, :Lk, : heads * d].float().reshape(B, Lk, heads, d).permute(0, 2, 1, 3)
    vh = vt[:, : heads * d, :Lk].float().reshape(B, heads, d, Lk).permute(0, 1, 3, 2)
    p = torch.softmax(qh @ kh.transpose(-1, -2) * scale, dim=-1)
    o = (p @ vh).permute(0, 2, 1, 3).reshape(B, Lq, heads * d)
    out[..., : heads * d] = o.to(out.dtype)
    return out


def window_attention(qkv, out, bias_table, C, heads, ws, shift, scale):
    B, H, W = qkv.shape[:3]
    hd = C // heads
    x = qkv[..., : 3 * C].float()
    if shift:
        x = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2))
    win = x.view(B, H // ws, ws, W // ws, ws, 3 * C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, 3, heads, hd)
    q, k, v = (win[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    attn = (q * scale) @ k.transpose(-1, -2)
    co = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)
    rel = (co[:, :, None] - co[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    idx = rel.sum(-1).to(qkv.device)
    attn = attn + bias_table.float()[idx.view(-1)].view(ws * ws, ws * ws, heads).permute(2, 0, 1)[None]
    if shift:
        img = torch.zeros((1, H, W, 1))
        cnt = 0
        for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
                img[:, hs, wsl, :] = cnt
                cnt += 1
        mw = img.view(1, H // ws, ws, W // ws, ws, 1).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws)
        am = mw.unsqueeze(1) - mw.unsqueeze(2)
        am = am.masked_fill(am != 0, -100.0).to(qkv.device)
        nW = am.shape[0]
        attn = (attn.view(B, nW, heads, ws * ws, ws * ws) + am[None, :, None]).view(-1, heads, ws * ws, ws * ws)
    o = (attn.softmax(-1) @ v).permute(0, 2, 1, 3).reshape(-1, ws * ws, C)
    o = o.view(B, H // ws, W // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, C)
    if shift:
        o = torch.roll(o, shifts=(shift, shift), dims=(1, 2))
    out[..., :C] = o.to(out.dtype)
    return out


def groupnorm(x, gamma, beta, eps, silu, out=None, groups=32, stats=None):
    C = x.shape[-1]
    B = x.shape[0]
    xf = x.float().reshape(B, -1, C).permute(0, 2, 1)
    y = F.group_norm(xf, groups, gamma.float(), beta.float(), eps)
    if silu:
        y = F.silu(y)
    y = y.permute(0, 2, 1).reshape(x.shape).to(x.dtype)
    if out is None:
        return y
    out.copy_(y)
    return out


def groupnorm_stats(x, groups=32):
    B, C = x.shape[0], x.shape[-1]
    xf = x.float().reshape(B, -1, groups, C // groups)
    var, mean = torch.var_mean(xf, dim=(1, 3), unbiased=False)
    return torch.cat([mean, var], dim=1).contiguous()


def groupnorm_apply(x, gamma, beta, mean_var, eps, silu, out=None, groups=32):
    B, C = x.shape[0], x.shape[-1]
    cpg = C // groups
    mean = mean_var[:, :groups].repeat_interleave(cpg, dim=1)
    var = mean_var[:, groups:].repeat_interleave(cpg, dim=1)
    shp = (B,) + (1,) * (x.dim() - 2) + (C,)
    y = (x.float() - mean.reshape(shp)) * torch.rsqrt(var.reshape(shp) + eps) * gamma.float() + beta.float()
    if silu:
        y = F.silu(y)
    y = y.to(x.dtype)
    if out is None:
        return y
    out.copy_(y)
    return out


# ---- fused transformer block (csrc/xformer.hip): the same math from the block's UNPACKED weights -------------------
XfBlock = real_ops.XfBlock
pack_xf_block = real_ops.pack_xf_block
xf_supported = real_ops.xf_supported


def pack_context_frags(k, vt, Lk, heads):
    """The test double keeps the text context in its natural layout (the kernel's fragment order is checked on the GPU)."""
    return k[:, :Lk].contiguous(), vt[:, :, :Lk].contiguous()


def groupnorm_affine(x, gamma, beta, eps, groups=32, stats=None):
    B, C = x.shape[0], x.shape[-1]
    xf = x.float().reshape(B, -1, groups, C // groups)
    mean = xf.mean(dim=(1, 3))
    var = xf.var(dim=(1, 3), unbiased=False)
    rstd = torch.rsqrt(var + eps).repeat_interleave(C // groups, dim=1)
    a = rstd * gamma.float()[None]
    s = beta.float()[None] - mean.repeat_interleave(C // groups, dim=1) * a
    return torch.stack([a, s], dim=1)


def _ln(v, g, b):
    return F.layer_norm(v, (v.shape[-1],), g.float(), b.float(), 1e-5)


def xf_head(x, ab, blk, L):
    w, dt = blk.logical, x.dtype
    C = x.shape[-1]
    xr = x.reshape(-1, C).float()
    B = xr.shape[0] // L
    xn = (xr.reshape(B, L, C) * ab[:, 0][:, None] + ab[:, 1][:, None]).to(dt).reshape(-1, C)
    h = (xn.float() @ w["proj_in.w"].float().t() + w["proj_in.b"].float()).to(dt)
    n = _ln(h.float(), w["norm1.w"], w["norm1.b"]).to(dt).float()
    q = (n @ w["q1.w"].float().t()).to(dt)
    k = (n @ w["k1.w"].float().t()).to(dt)
    v = (n @ w["v1.w"].float().t()).to(dt)
    return h, torch.cat([q, k], dim=1).reshape(B, L, 2 * C), v.reshape(B, L, C).permute(0, 2, 1).contiguous()


def xf_tail(attn, h, x, blk, kf, vf, Lk, scale, L, out=None, pair_bs=0, stop_after=0):
    """kf / vf here are the test double's (k [B, Lk, C], v^T [B, C, Lk]) from pack_context_frags above."""
    w, dt = blk.logical, attn.dtype
    C, heads = attn.shape[-1], blk.heads
    a, hh, xx = attn.reshape(-1, C), h.reshape(-1, C), x.reshape(-1, C)
    if pair_bs:
        def expand(t):
            G = t.shape[0] // (L * pair_bs)
            return t.reshape(G, 1, pair_bs * L, C).expand(G, 2, pair_bs * L, C).reshape(-1, C)
        a, hh, xx = expand(a), expand(hh), expand(xx)
    B = a.shape[0] // L
    h1 = (a.float() @ w["out1.w"].float().t() + w["out1.b"].float() + hh.float()).to(dt)
    n2 = _ln(h1.float(), w["norm2.w"], w["norm2.b"]).to(dt)
    q = (n2.float() @ w["q2.w"].float().t()).to(dt)
    qh = q.float().reshape(B, L, heads, 64).permute(0, 2, 1, 3)
    kh = kf.float().reshape(B, Lk, heads, 64).permute(0, 2, 1, 3)
    vh = vf.float().reshape(B, heads, 64, Lk).permute(0, 1, 3, 2)
    ca = (torch.softmax(qh @ kh.transpose(-1, -2) * scale, dim=-1) @ vh).permute(0, 2, 1, 3).reshape(B * L, C).to(dt)
    h2 = (ca.float() @ w["out2.w"].float().t() + w["out2.b"].float() + h1.float()).to(dt)
    n3 = _ln(h2.float(), w["norm3.w"], w["norm3.b"]).to(dt)
    u = n3.float() @ w["ff1.w"].float().t() + w["ff1.b"].float()
    g = (u[:, : 4 * C] * F.gelu(u[:, 4 * C:])).to(dt)
    h3 = (g.float() @ w["ff2.w"].float().t() + w["ff2.b"].float() + h2.float()).to(dt)
    res = (h3.float() @ w["proj_out.w"].float().t() + w["proj_out.b"].float() + xx.float()).to(dt)
    if stop_after in (1, 4):  # the kernel's operand image holds the normalised rows; the affine map lives in its weights
        res = F.layer_norm((h1 if stop_after == 1 else h2).float(), (C,)).to(dt)
    else:
        res = {0: res, 11: h1, 2: q, 3: ca, 14: h2, 5: h3}[stop_after]
    if out is None:
        shp = (x.shape[0] * (2 if pair_bs else 1),) + tuple(x.shape[1:])
        return res.reshape(shp)
    out.copy_(res.reshape(out.shape))
    return out


def layernorm(x, gamma, beta, C=None, eps=1e-5, out=None):
    Cpad = x.shape[-1]
    C = Cpad if C is None else C
    y = torch.zeros(x.shape, dtype=torch.float32, device=x.device)
    y[..., :C] = F.layer_norm(x[..., :C].float(), (C,), gamma.float()[:C], beta.float()[:C], eps)
    y = y.to(x.dtype)
    if out is None:
        return y
    out.copy_(y)
    return out


def clip_embed(tokens, tok_emb, pos):
    return tok_emb[tokens] + pos[: tokens.shape[1]]


def add_layernorm_f32(x, y, gamma, beta, out_dtype, eps=1e-5):
    if y is not None:
        x.add_(y)
    return F.layer_norm(x, (x.shape[-1],), gamma.float(), beta.float(), eps).to(out_dtype)


def causal_attention(qkv, heads, scale):
    B, L = qkv.shape[:2]
    q, k, v = qkv[..., : 3 * heads * 64].float().reshape(B, L, 3, heads, 64).permute(2, 0, 3, 1, 4)
    mask = torch.full((L, L), float("-inf"), device=qkv.device).triu_(1)
    p = torch.softmax(q @ k.transpose(-1, -2) * scale + mask, dim=-1)
    return (p @ v).permute(0, 2, 1, 3).reshape(B, L, heads * 64).to(qkv.dtype)


def softmax_rows_(x, L):
    y = torch.zeros_like(x, dtype=torch.float32)
    y[..., :L] = torch.softmax(x[..., :L].float(), dim=-1)
    x.copy_(y.to(x.dtype))
    return x


def add_scaled(a, b, s, out=None):
    y = (a.float() + s * b.float()).to(a.dtype)
    if out is None:
        return y
    out.copy_(y)
    return out


def space_to_depth2(x):
    B, H, W, C = x.shape
    return x.reshape(B, H // 2, 2, W // 2, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(B, H // 2, W // 2, 4 * C).contiguous()


def depth_to_space2(x, out=None):
    B, h, w, C4 = x.shape
    C = C4 // 4
    y = x.reshape(B, h, w, 2, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(B, 2 * h, 2 * w, C)
    if out is None:
        return y.contiguous()
    out.copy_(y)
    return out


def nchw_to_nhwc(src0, src1, cpad, dtype, scale=1.0, shift=0.0):
    src = src0 if src1 is None else torch.cat([src0, src1], dim=1)
    B, C, H, W = src.shape
    out = torch.zeros((B, H, W, cpad), dtype=dtype, device=src.device)
    out[..., :C] = (src * scale + shift).permute(0, 2, 3, 1).to(dtype)
    return out


def nhwc_to_nchw(src, C, scale=1.0, shift=None):
    y = src[..., :C].float().permute(0, 3, 1, 2) * scale
    if shift is not None:
        y = y + shift.float().view(1, C, 1, 1)
    return y.contiguous()


def pixel_unshuffle(src, r, cpad, mean, rng, dtype):
    B, C, H, W = src.shape
    y = F.pixel_unshuffle((src - mean.view(1, C, 1, 1)) * rng, r)
    out = torch.zeros((B, H // r, W // r, cpad), dtype=dtype, device=src.device)
    out[..., : C * r * r] = y.permute(0, 2, 3, 1).to(dtype)
    return out


def timestep_embedding(t, dim, dtype, max_period=10000.0):
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half).to(t.device)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb.to(dtype)


def lincomb4(x, ca, y=None, cb=None, z=None, cc=None, w=None, cd=None):
    sh = (-1,) + (1,) * (x.dim() - 1)
    out = ca.view(sh) * x
    for t, c in ((y, cb), (z, cc), (w, cd)):
        if t is not None:
            out = out + c.view(sh) * t
    return out


def spaced_step(x, oc, ou, noise, s, k_x, k_o, c1, c2, sd):
    sh = (-1,) + (1,) * (x.dim() - 1)
    o = oc if ou is None else ou + s * (oc - ou)
    x0 = k_x.view(sh) * x - k_o.view(sh) * o
    return c1.view(sh) * x0 + c2.view(sh) * x + sd.view(sh) * noise


def tile_gather(x, coords, ts):
    tiles = [x[..., int(h):int(h) + ts, int(w):int(w) + ts] for h, w in coords.tolist()]
    return torch.cat(tiles, dim=0).contiguous()


def tile_accumulate(tiles, weights, coords, B, H, W):
    C, ts = tiles.shape[1], tiles.shape[2]
    out = torch.zeros((B, C, H, W), dtype=torch.float32, device=tiles.device)
    cnt = torch.zeros_like(out)
    for t, (h, w) in enumerate(coords.tolist()):
        out[..., h:h + ts, w:w + ts] += tiles[t * B:(t + 1) * B] * weights
        cnt[..., h:h + ts, w:w + ts] += weights
    return out / cnt


def tile_accumulate_partial(tiles, weights, coords, B, C, H, W):
    ts = weights.shape[-1]
    out = torch.zeros((B, C, H, W), dtype=torch.float32, device=weights.device)
    for t, (h, w) in enumerate(coords.tolist()):
        out[..., h:h + ts, w:w + ts] += weights if tiles is None else tiles[t * B:(t + 1) * B] * weights
    return out


def tile_normalize(num, den):
    return num / den.reshape(num.shape[-2:])


def u8_to_f32_nchw(src):
    return src.float().div(255).clamp(0, 1).permute(0, 3, 1, 2).contiguous()


def wavelet_blur(src, radius):
    sh = src.shape
    x = src.reshape(-1, 1, sh[-2], sh[-1])
    k = torch.tensor([[0.0625, 0.125, 0.0625], [0.125, 0.25, 0.125], [0.0625, 0.125, 0.0625]],
                     dtype=src.dtype, device=src.device)[None, None]
    x = F.pad(x, (radius,) * 4, mode="replicate")
    return F.conv2d(x, k, dilation=radius).reshape(sh)


def colorfix(content, content_low, style_low):
    return (content - content_low) + style_low


def f32_nchw_to_u8_nhwc(src):
    return (src * 255.0).clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()


_NAMES = ["linear", "linear_t", "conv3x3", "bmm_nt", "attention", "window_attention", "groupnorm", "groupnorm_stats",
          "groupnorm_apply", "layernorm", "clip_embed", "add_layernorm_f32", "causal_attention",
          "space_to_depth2", "depth_to_space2", "pack_context_frags", "groupnorm_affine", "xf_head", "xf_tail",
          "softmax_rows_", "add_scaled", "nchw_to_nhwc", "nhwc_to_nchw", "pixel_unshuffle", "timestep_embedding",
          "lincomb4", "spaced_step", "tile_gather", "tile_accumulate", "tile_accumulate_partial", "tile_normalize",
          "u8_to_f32_nchw", "wavelet_blur", "colorfix",
          "f32_nchw_to_u8_nhwc"]


def install(monkeypatch):
    """Replace the kernel-backed ops with the emulation (CPU wiring tests only)."""
    g = globals()
    for n in _NAMES:
        monkeypatch.setattr(real_ops, n, g[n])
