"""Built-in network configurations.

`FULL_*` reproduce the values of the reference's configs/inference/{cldm,swinir,bsrnet,scunet,diffusion,diffusion_v2.1}.yaml
(the config surface); `TINY_*` are structurally identical, narrow networks used by the
parity tests and golden fixtures (they run through the reference on CPU in seconds).
"""
import copy


def _unet(mc, ctx, in_ch=4, hint=None, mult=(1, 2, 4, 4), attn=(4, 2, 1), nrb=2):
    d = dict(use_checkpoint=True, image_size=32, in_channels=in_ch, model_channels=mc,
             attention_resolutions=list(attn), num_res_blocks=nrb, channel_mult=list(mult),
             num_head_channels=64, use_spatial_transformer=True, use_linear_in_transformer=True,
             transformer_depth=1, context_dim=ctx, legacy=False)
    if hint is None:
        d["out_channels"] = 4
    else:
        d["hint_channels"] = hint
    return d


def _vae(ch, mult=(1, 2, 4, 4)):
    return dict(embed_dim=4, ddconfig=dict(double_z=True, z_channels=4, resolution=256, in_channels=3,
                                          out_ch=3, ch=ch, ch_mult=list(mult), num_res_blocks=2,
                                          attn_resolutions=[], dropout=0.0))


def _clip(width, heads, layers, vocab=49408):
    return dict(embed_dim=width,
                vision_cfg=dict(image_size=224, layers=32, width=1280, head_width=80, patch_size=14),
                text_cfg=dict(context_length=77, vocab_size=vocab, width=width, heads=heads, layers=layers),
                layer="penultimate")


FULL_CLDM = dict(latent_scale_factor=0.18215, unet_cfg=_unet(320, 1024), vae_cfg=_vae(128),
                 clip_cfg=_clip(1024, 16, 24), controlnet_cfg=_unet(320, 1024, hint=4))

FULL_SWINIR = dict(img_size=64, patch_size=1, in_chans=3, embed_dim=180, depths=[6] * 8, num_heads=[6] * 8,
                   window_size=8, mlp_ratio=2, sf=8, img_range=1.0, upsampler="nearest+conv",
                   resi_connection="1conv", unshuffle=True, unshuffle_scale=8)

FULL_BSRNET = dict(in_nc=3, out_nc=3, nf=64, nb=23, gc=32, sf=4)          # configs/inference/bsrnet.yaml
FULL_SCUNET = dict(in_nc=3, config=[4, 4, 4, 4, 4, 4, 4], dim=64)        # configs/inference/scunet.yaml
TINY_BSRNET = dict(in_nc=3, out_nc=3, nf=16, nb=2, gc=8, sf=4)
TINY_SCUNET = dict(in_nc=3, config=[2, 1, 2, 1, 2, 1, 2], dim=64)        # dim/2 must stay a multiple of head_dim 32

DIFFUSION_V2 = dict(linear_start=0.00085, linear_end=0.0120, timesteps=1000)
DIFFUSION_V21 = dict(linear_start=0.00085, linear_end=0.0120, timesteps=1000, zero_snr=True,
                     parameterization="v")

# Narrow twins (same topology: 4 levels, attention at ds 1/2/4, 13 control tensors).
TINY_CLDM = dict(latent_scale_factor=0.18215, unet_cfg=_unet(64, 128), vae_cfg=_vae(32),
                 clip_cfg=_clip(128, 2, 3, vocab=49408), controlnet_cfg=_unet(64, 128, hint=4))
TINY_SWINIR = dict(img_size=64, patch_size=1, in_chans=3, embed_dim=60, depths=[2, 2], num_heads=[6, 6],
                   window_size=8, mlp_ratio=2, sf=8, img_range=1.0, upsampler="nearest+conv",
                   resi_connection="1conv", unshuffle=True, unshuffle_scale=8)


def get(name: str) -> dict:
    return copy.deepcopy(globals()[name])


_YAML = {
    "cldm": ("diffbir.model.ControlLDM", "FULL_CLDM"),
    "swinir": ("diffbir.model.SwinIR", "FULL_SWINIR"),
    "bsrnet": ("diffbir.model.RRDBNet", "FULL_BSRNET"),
    "scunet": ("diffbir.model.SCUNet", "FULL_SCUNET"),
    "diffusion": ("diffbir.model.Diffusion", "DIFFUSION_V2"),
    "diffusion_v2.1": ("diffbir.model.Diffusion", "DIFFUSION_V21"),
}


def yaml_config(name: str) -> dict:
    """The `{target, params}` tree of the reference's configs/inference/<name>.yaml (targets in the reference's
    `diffbir.model.*` namespace, which `instantiate_from_config` resolves to this package)."""
    target, key = _YAML[name]
    return dict(target=target, params=get(key))
