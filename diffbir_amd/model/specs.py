"""Parameter-tree specifications (names + shapes) of the networks on the hot path.

The reference loads checkpoints with ``load_state_dict(strict=True)`` (reference cldm.py:66,
inference/bsr_loop.py:32), so the *key names* of its module tree are part of the drop-in boundary
(SURVEY.md §8b B5).  The engine does not mirror the reference's nn.Module classes; instead these
functions enumerate, from the YAML config values alone, every ``(key, shape, kind)`` the reference
module tree would hold:

* UNet / ControlNet   — reference unet.py:391-685, controlnet.py:52-312, attention.py:219-332
* AutoencoderKL       — reference vae.py:306-399 (Encoder), 429-524 (Decoder), 562-571
* SwinIR              — reference swinir.py:622-812
* OpenCLIP text tower — reference open_clip/transformer.py:199-254, 517-566, model.py:160-180

``kind`` drives synthetic initialisation (utils/synth.py) and packing:
  "w" dense weight (conv / linear), "b" bias, "g" norm gain, "e" embedding/table, "buf" non-learned buffer.
"""
from collections import OrderedDict
from typing import Dict, List

Spec = "OrderedDict[str, Tuple[Tuple[int, ...], str]]"


def _conv(sp, name, cin, cout, k):
    sp[f"{name}.weight"] = ((cout, cin, k, k), "w")
    sp[f"{name}.bias"] = ((cout,), "b")


def _lin(sp, name, cin, cout, bias=True):
    sp[f"{name}.weight"] = ((cout, cin), "w")
    if bias:
        sp[f"{name}.bias"] = ((cout,), "b")


def _norm(sp, name, c):
    sp[f"{name}.weight"] = ((c,), "g")
    sp[f"{name}.bias"] = ((c,), "b")


# --------------------------------------------------------------------------------------------
# UNet / ControlNet (SD-2.1 layout)
# --------------------------------------------------------------------------------------------
def _resblock(sp, p, cin, cout, emb):
    _norm(sp, f"{p}.in_layers.0", cin)
    _conv(sp, f"{p}.in_layers.2", cin, cout, 3)
    _lin(sp, f"{p}.emb_layers.1", emb, cout)
    _norm(sp, f"{p}.out_layers.0", cout)
    _conv(sp, f"{p}.out_layers.3", cout, cout, 3)
    if cin != cout:
        _conv(sp, f"{p}.skip_connection", cin, cout, 1)


def _spatial_transformer(sp, p, ch, ctx_dim, depth=1):
    _norm(sp, f"{p}.norm", ch)
    _lin(sp, f"{p}.proj_in", ch, ch)
    for d in range(depth):
        q = f"{p}.transformer_blocks.{d}"
        for nm, kdim in (("attn1", ch), ("attn2", ctx_dim)):
            _lin(sp, f"{q}.{nm}.to_q", ch, ch, bias=False)
            _lin(sp, f"{q}.{nm}.to_k", kdim, ch, bias=False)
            _lin(sp, f"{q}.{nm}.to_v", kdim, ch, bias=False)
            _lin(sp, f"{q}.{nm}.to_out.0", ch, ch)
        _lin(sp, f"{q}.ff.net.0.proj", ch, ch * 8)
        _lin(sp, f"{q}.ff.net.2", ch * 4, ch)
        for n in ("norm1", "norm2", "norm3"):
            _norm(sp, f"{q}.{n}", ch)
    _lin(sp, f"{p}.proj_out", ch, ch)


class UNetPlan:
    """Static description of the encoder / middle / decoder block sequence (shared by UNet & ControlNet).

    blocks: list of dicts with keys
      kind: "conv_in" | "res" | "down" | "up"
      (res) cin, cout, attn(bool), and for decoder blocks `skip` = channels of the popped skip tensor,
      `up`(bool) if the block ends with nearest-x2 + conv.
    """

    def __init__(self, cfg: dict, hint_channels: int = 0):
        mc = cfg["model_channels"]
        mult = list(cfg["channel_mult"])
        nrb = cfg["num_res_blocks"]
        nrb = [nrb] * len(mult) if isinstance(nrb, int) else list(nrb)
        attn_res = set(cfg["attention_resolutions"])
        self.mc, self.emb = mc, 4 * mc
        self.in_ch = cfg["in_channels"] + hint_channels
        self.out_ch = cfg.get("out_channels", cfg["in_channels"])
        self.ctx_dim = cfg["context_dim"]
        self.head_dim = cfg["num_head_channels"]
        self.depth = cfg.get("transformer_depth", 1)
        assert cfg.get("use_spatial_transformer", True) and cfg.get("use_linear_in_transformer", True)
        assert not cfg.get("resblock_updown", False) and not cfg.get("use_scale_shift_norm", False)
        self.input: List[dict] = [dict(kind="conv_in", cin=self.in_ch, cout=mc, ds=1)]
        chans = [mc]
        ch, ds = mc, 1
        for level, m in enumerate(mult):
            for _ in range(nrb[level]):
                self.input.append(dict(kind="res", cin=ch, cout=m * mc, attn=ds in attn_res, ds=ds))
                ch = m * mc
                chans.append(ch)
            if level != len(mult) - 1:
                self.input.append(dict(kind="down", cin=ch, cout=ch, ds=ds))
                chans.append(ch)
                ds *= 2
        self.mid_ch, self.mid_ds = ch, ds
        self.skip_chans = list(chans)
        self.output: List[dict] = []
        for level, m in list(enumerate(mult))[::-1]:
            for i in range(nrb[level] + 1):
                ich = chans.pop()
                blk = dict(kind="res", cin=ch + ich, skip=ich, cout=mc * m, attn=ds in attn_res, ds=ds, up=False)
                ch = mc * m
                if level and i == nrb[level]:
                    blk["up"] = True
                    ds //= 2
                self.output.append(blk)
        self.final_ch = ch


def unet_spec(cfg: dict) -> Spec:
    """Keys of reference ``ControlledUnetModel`` (= UNetModel, unet.py:391-685)."""
    plan = UNetPlan(cfg)
    sp = OrderedDict()
    _lin(sp, "time_embed.0", plan.mc, plan.emb)
    _lin(sp, "time_embed.2", plan.emb, plan.emb)
    _encoder_spec(sp, plan)
    for i, b in enumerate(plan.output):
        p = f"output_blocks.{i}"
        _resblock(sp, f"{p}.0", b["cin"], b["cout"], plan.emb)
        j = 1
        if b["attn"]:
            _spatial_transformer(sp, f"{p}.1", b["cout"], plan.ctx_dim, plan.depth)
            j = 2
        if b["up"]:
            _conv(sp, f"{p}.{j}.conv", b["cout"], b["cout"], 3)
    _norm(sp, "out.0", plan.final_ch)
    _conv(sp, "out.2", plan.mc, plan.out_ch, 3)
    return sp


def _encoder_spec(sp, plan: UNetPlan):
    for i, b in enumerate(plan.input):
        p = f"input_blocks.{i}"
        if b["kind"] == "conv_in":
            _conv(sp, f"{p}.0", b["cin"], b["cout"], 3)
        elif b["kind"] == "res":
            _resblock(sp, f"{p}.0", b["cin"], b["cout"], plan.emb)
            if b["attn"]:
                _spatial_transformer(sp, f"{p}.1", b["cout"], plan.ctx_dim, plan.depth)
        else:
            _conv(sp, f"{p}.0.op", b["cin"], b["cout"], 3)
    c = plan.mid_ch
    _resblock(sp, "middle_block.0", c, c, plan.emb)
    _spatial_transformer(sp, "middle_block.1", c, plan.ctx_dim, plan.depth)
    _resblock(sp, "middle_block.2", c, c, plan.emb)


def controlnet_spec(cfg: dict) -> Spec:
    """Keys of reference ``ControlNet`` (controlnet.py:52-312): encoder copy + 13 zero-convs."""
    plan = UNetPlan(cfg, hint_channels=cfg["hint_channels"])
    sp = OrderedDict()
    _lin(sp, "time_embed.0", plan.mc, plan.emb)
    _lin(sp, "time_embed.2", plan.emb, plan.emb)
    _encoder_spec(sp, plan)
    for i, b in enumerate(plan.input):
        _conv(sp, f"zero_convs.{i}.0", b["cout"], b["cout"], 1)
    _conv(sp, "middle_block_out.0", plan.mid_ch, plan.mid_ch, 1)
    return sp


# --------------------------------------------------------------------------------------------
# AutoencoderKL
# --------------------------------------------------------------------------------------------
def _vae_res(sp, p, cin, cout):
    _norm(sp, f"{p}.norm1", cin)
    _conv(sp, f"{p}.conv1", cin, cout, 3)
    _norm(sp, f"{p}.norm2", cout)
    _conv(sp, f"{p}.conv2", cout, cout, 3)
    if cin != cout:
        _conv(sp, f"{p}.nin_shortcut", cin, cout, 1)


def _vae_attn(sp, p, c):
    _norm(sp, f"{p}.norm", c)
    for n in ("q", "k", "v", "proj_out"):
        _conv(sp, f"{p}.{n}", c, c, 1)


def vae_spec(cfg: dict) -> Spec:
    dd = cfg["ddconfig"]
    ch, mult, nrb = dd["ch"], list(dd["ch_mult"]), dd["num_res_blocks"]
    zc, embed = dd["z_channels"], cfg["embed_dim"]
    assert not dd.get("attn_resolutions"), "attn_resolutions other than [] not on the hot path"
    sp = OrderedDict()
    # encoder (vae.py:306-399)
    _conv(sp, "encoder.conv_in", dd["in_channels"], ch, 3)
    in_mult = [1] + mult
    bi = ch
    for l in range(len(mult)):
        bi, bo = ch * in_mult[l], ch * mult[l]
        for b in range(nrb):
            _vae_res(sp, f"encoder.down.{l}.block.{b}", bi, bo)
            bi = bo
        if l != len(mult) - 1:
            _conv(sp, f"encoder.down.{l}.downsample.conv", bi, bi, 3)
    _vae_res(sp, "encoder.mid.block_1", bi, bi)
    _vae_attn(sp, "encoder.mid.attn_1", bi)
    _vae_res(sp, "encoder.mid.block_2", bi, bi)
    _norm(sp, "encoder.norm_out", bi)
    _conv(sp, "encoder.conv_out", bi, 2 * zc if dd.get("double_z", True) else zc, 3)
    # decoder (vae.py:429-524); note reference inserts levels at index 0 => key index == level
    bi = ch * mult[-1]
    _conv(sp, "decoder.conv_in", zc, bi, 3)
    _vae_res(sp, "decoder.mid.block_1", bi, bi)
    _vae_attn(sp, "decoder.mid.attn_1", bi)
    _vae_res(sp, "decoder.mid.block_2", bi, bi)
    for l in reversed(range(len(mult))):
        bo = ch * mult[l]
        for b in range(nrb + 1):
            _vae_res(sp, f"decoder.up.{l}.block.{b}", bi, bo)
            bi = bo
        if l != 0:
            _conv(sp, f"decoder.up.{l}.upsample.conv", bi, bi, 3)
    _norm(sp, "decoder.norm_out", bi)
    _conv(sp, "decoder.conv_out", bi, dd["out_ch"], 3)
    _conv(sp, "quant_conv", 2 * zc, 2 * embed, 1)
    _conv(sp, "post_quant_conv", embed, zc, 1)
    return sp


# --------------------------------------------------------------------------------------------
# SwinIR (nearest+conv upsampler, 1conv residual connection, unshuffle front-end)
# --------------------------------------------------------------------------------------------
def swinir_spec(cfg: dict) -> Spec:
    assert cfg.get("upsampler") == "nearest+conv" and cfg.get("resi_connection", "1conv") == "1conv"
    assert cfg.get("unshuffle", False) and cfg.get("patch_size", 1) == 1
    C = cfg["embed_dim"]
    ws = cfg["window_size"]
    nin = cfg["in_chans"] * cfg["unshuffle_scale"] ** 2
    hidden = int(C * cfg["mlp_ratio"])
    sf = cfg["sf"]
    sp = OrderedDict()
    _conv(sp, "conv_first.1", nin, C, 3)
    _norm(sp, "patch_embed.norm", C)
    res = cfg["img_size"]
    for i, (depth, heads) in enumerate(zip(cfg["depths"], cfg["num_heads"])):
        for j in range(depth):
            p = f"layers.{i}.residual_group.blocks.{j}"
            if j % 2 == 1 and res > ws:
                sp[f"{p}.attn_mask"] = (((res // ws) ** 2, ws * ws, ws * ws), "buf")
            _norm(sp, f"{p}.norm1", C)
            sp[f"{p}.attn.relative_position_bias_table"] = (((2 * ws - 1) ** 2, heads), "e")
            sp[f"{p}.attn.relative_position_index"] = ((ws * ws, ws * ws), "buf")
            _lin(sp, f"{p}.attn.qkv", C, 3 * C)
            _lin(sp, f"{p}.attn.proj", C, C)
            _norm(sp, f"{p}.norm2", C)
            _lin(sp, f"{p}.mlp.fc1", C, hidden)
            _lin(sp, f"{p}.mlp.fc2", hidden, C)
        _conv(sp, f"layers.{i}.conv", C, C, 3)
    _norm(sp, "norm", C)
    _conv(sp, "conv_after_body", C, C, 3)
    _conv(sp, "conv_before_upsample.0", C, 64, 3)
    _conv(sp, "conv_up1", 64, 64, 3)
    if sf in (4, 8):
        _conv(sp, "conv_up2", 64, 64, 3)
    if sf == 8:
        _conv(sp, "conv_up3", 64, 64, 3)
    _conv(sp, "conv_hr", 64, 64, 3)
    _conv(sp, "conv_last", 64, cfg["in_chans"], 3)
    return sp


# --------------------------------------------------------------------------------------------
# BSRNet (RRDBNet, reference bsrnet.py:75-104) and SCUNet (scunet.py:163-264) stage-1 cleaners
# --------------------------------------------------------------------------------------------
def bsrnet_spec(cfg: dict) -> Spec:
    nf, gc, nb, sf = cfg.get("nf", 64), cfg.get("gc", 32), cfg.get("nb", 23), cfg.get("sf", 4)
    assert sf in (2, 4), "RRDBNet: sf 2 or 4 (bsrnet.py:87-88, 99-100)"
    sp = OrderedDict()
    _conv(sp, "conv_first", cfg.get("in_nc", 3), nf, 3)
    for i in range(nb):
        for r in (1, 2, 3):
            p = f"RRDB_trunk.{i}.RDB{r}"
            for k in range(4):
                _conv(sp, f"{p}.conv{k + 1}", nf + k * gc, gc, 3)
            _conv(sp, f"{p}.conv5", nf + 4 * gc, nf, 3)
    _conv(sp, "trunk_conv", nf, nf, 3)
    _conv(sp, "upconv1", nf, nf, 3)
    if sf == 4:
        _conv(sp, "upconv2", nf, nf, 3)
    _conv(sp, "HRconv", nf, nf, 3)
    _conv(sp, "conv_last", nf, cfg.get("out_nc", 3), 3)
    return sp


def scunet_stages(cfg: dict):
    """(state-dict prefix, block channels (conv_dim == trans_dim == ch // 2 ... see below), #blocks, first index) per
    stage, in forward order.  ConvTransBlock(conv_dim=c, trans_dim=c) works on 2c channels (scunet.py:173-206)."""
    dim, n = cfg.get("dim", 64), list(cfg.get("config", [2] * 7))
    return [("m_down1", dim // 2, n[0], 0), ("m_down2", dim, n[1], 0), ("m_down3", 2 * dim, n[2], 0),
            ("m_body", 4 * dim, n[3], 0), ("m_up3", 2 * dim, n[4], 1), ("m_up2", dim, n[5], 1),
            ("m_up1", dim // 2, n[6], 1)]


def scunet_spec(cfg: dict) -> Spec:
    dim, in_nc, hd, ws = cfg.get("dim", 64), cfg.get("in_nc", 3), 32, 8
    sp = OrderedDict()
    sp["m_head.0.weight"] = ((dim, in_nc, 3, 3), "w")
    for name, c, nblk, first in scunet_stages(cfg):
        if first == 1:  # ConvTranspose2d(4c, 2c, 2, 2): weight [in, out, 2, 2]
            sp[f"{name}.0.weight"] = ((4 * c, 2 * c, 2, 2), "w")
        for i in range(nblk):
            p = f"{name}.{i + first}"
            t = f"{p}.trans_block"
            _norm(sp, f"{t}.ln1", c)
            sp[f"{t}.msa.relative_position_params"] = ((c // hd, 2 * ws - 1, 2 * ws - 1), "e")
            _lin(sp, f"{t}.msa.embedding_layer", c, 3 * c)
            _lin(sp, f"{t}.msa.linear", c, c)
            _norm(sp, f"{t}.ln2", c)
            _lin(sp, f"{t}.mlp.0", c, 4 * c)
            _lin(sp, f"{t}.mlp.2", 4 * c, c)
            _conv(sp, f"{p}.conv1_1", 2 * c, 2 * c, 1)
            _conv(sp, f"{p}.conv1_2", 2 * c, 2 * c, 1)
            sp[f"{p}.conv_block.0.weight"] = ((c, c, 3, 3), "w")
            sp[f"{p}.conv_block.2.weight"] = ((c, c, 3, 3), "w")
        if name.startswith("m_down"):  # Conv2d(2c, 4c, 2, 2)
            sp[f"{name}.{nblk}.weight"] = ((4 * c, 2 * c, 2, 2), "w")
    sp["m_tail.0.weight"] = ((in_nc, dim, 3, 3), "w")
    return sp


# --------------------------------------------------------------------------------------------
# OpenCLIP text tower (keys below FrozenOpenCLIPEmbedder, i.e. prefixed "model.")
# --------------------------------------------------------------------------------------------
def clip_text_spec(cfg: dict) -> Spec:
    t = cfg["text_cfg"]
    W, L = t["width"], t["layers"]
    sp = OrderedDict()
    sp["model.positional_embedding"] = ((t["context_length"], W), "e")
    sp["model.text_projection"] = ((W, cfg["embed_dim"]), "w")
    sp["model.logit_scale"] = ((), "e")
    for i in range(L):
        p = f"model.transformer.resblocks.{i}"
        _norm(sp, f"{p}.ln_1", W)
        sp[f"{p}.attn.in_proj_weight"] = ((3 * W, W), "w")
        sp[f"{p}.attn.in_proj_bias"] = ((3 * W,), "b")
        _lin(sp, f"{p}.attn.out_proj", W, W)
        _norm(sp, f"{p}.ln_2", W)
        _lin(sp, f"{p}.mlp.c_fc", W, 4 * W)
        _lin(sp, f"{p}.mlp.c_proj", 4 * W, W)
    sp["model.token_embedding.weight"] = ((t["vocab_size"], W), "e")
    _norm(sp, "model.ln_final", W)
    return sp


def cldm_spec(cfg: dict) -> Dict[str, Spec]:
    """Specs of the four sub-networks of ControlLDM (reference cldm.py:22-32)."""
    return dict(
        unet=unet_spec(cfg["unet_cfg"]),
        controlnet=controlnet_spec(cfg["controlnet_cfg"]),
        vae=vae_spec(cfg["vae_cfg"]),
        clip=clip_text_spec(cfg["clip_cfg"]),
    )
