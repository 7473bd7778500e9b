#!/usr/bin/env python
"""Command line of the restoration engine — the flag surface of the reference CLI (reference inference.py:55-305:
same names, types, defaults and choices), driving the same loop classes (`diffbir.inference`).

    python inference.py --task sr --upscale 4 --version v2.1 --sampler spaced --steps 50 --captioner none \
        --cfg_scale 4 --input inputs/demo/bsr --output results/demo_bsr --precision fp16 --device cuda

Checkpoints are looked up under ./weights/<file name of the reference's download URL>.  `--task sr | face | denoise`
and `--version v1 | v2 | v2.1 | custom` run on the engine (SwinIR / BSRNet / SCUNet stage-1 models); `unaligned_face`
(RetinaFace detection) and the LLaVA / RAM captioners are outside it and raise with a pointer to DESIGN.md §7.
"""
import os
from argparse import ArgumentParser, Namespace

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")   # a default for an UNSET variable, before torch initialises HIP (diffbir_amd/__init__.py)

import torch  # noqa: E402

DEFAULT_POS_PROMPT = (
    "Cinematic, High Contrast, highly detailed, taken using a Canon EOS R camera, "
    "hyper detailed photo - realistic maximum detail, 32k, Color Grading, ultra HD, extreme meticulous detailing, "
    "skin pore detailing, hyper sharpness, perfect without deformations.")
DEFAULT_NEG_PROMPT = (
    "painting, oil painting, illustration, drawing, art, sketch, oil painting, cartoon, "
    "CG Style, 3D render, unreal engine, blurring, dirty, messy, worst quality, low quality, frames, watermark, "
    "signature, jpeg artifacts, deformed, lowres, over-smooth.")
SAMPLERS = ["dpm++_m2", "spaced", "ddim", "edm_euler", "edm_euler_a", "edm_heun", "edm_dpm_2", "edm_dpm_2_a", "edm_lms",
            "edm_dpm++_2s_a", "edm_dpm++_sde", "edm_dpm++_2m", "edm_dpm++_2m_sde", "edm_dpm++_3m_sde"]


def check_device(device: str) -> str:
    """ROCm PyTorch reports the MI355X as `cuda`; the engine has no CPU / MPS compute path."""
    if device == "cuda" and not torch.cuda.is_available():
        raise SystemExit("no GPU visible: this engine runs its HIP kernels on an MI355X (device 'cuda' under ROCm)")
    if device != "cuda":
        raise SystemExit(f"device '{device}' is not supported by the MI355X engine (use --device cuda)")
    print(f"using device {device}")
    return device


def build_parser() -> ArgumentParser:
    p = ArgumentParser()
    # model
    p.add_argument("--task", type=str, default="sr", choices=["sr", "face", "denoise", "unaligned_face"])
    p.add_argument("--upscale", type=float, default=4)
    p.add_argument("--version", type=str, default="v2.1", choices=["v1", "v2", "v2.1", "custom"])
    p.add_argument("--train_cfg", type=str, default="")
    p.add_argument("--ckpt", type=str, default="")
    # sampling
    p.add_argument("--sampler", type=str, default="edm_dpm++_3m_sde", choices=SAMPLERS)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--start_point_type", type=str, choices=["noise", "cond"], default="noise")
    p.add_argument("--cleaner_tiled", action="store_true")
    p.add_argument("--cleaner_tile_size", type=int, default=512)
    p.add_argument("--cleaner_tile_stride", type=int, default=256)
    p.add_argument("--vae_encoder_tiled", action="store_true")
    p.add_argument("--vae_encoder_tile_size", type=int, default=256)
    p.add_argument("--vae_decoder_tiled", action="store_true")
    p.add_argument("--vae_decoder_tile_size", type=int, default=256)
    p.add_argument("--cldm_tiled", action="store_true")
    p.add_argument("--cldm_tile_size", type=int, default=512)
    p.add_argument("--cldm_tile_stride", type=int, default=256)
    p.add_argument("--captioner", type=str, choices=["none", "llava", "ram"], default="llava")
    p.add_argument("--pos_prompt", type=str, default=DEFAULT_POS_PROMPT)
    p.add_argument("--neg_prompt", type=str, default=DEFAULT_NEG_PROMPT)
    p.add_argument("--cfg_scale", type=float, default=6.0)
    p.add_argument("--rescale_cfg", action="store_true")
    p.add_argument("--noise_aug", type=int, default=0)
    p.add_argument("--s_churn", type=float, default=0)
    p.add_argument("--s_tmin", type=float, default=0)
    p.add_argument("--s_tmax", type=float, default=300)
    p.add_argument("--s_noise", type=float, default=1)
    p.add_argument("--eta", type=float, default=1)
    p.add_argument("--order", type=int, default=1)
    p.add_argument("--strength", type=float, default=1)
    p.add_argument("--batch_size", type=int, default=1)
    # guidance (accepted for CLI compatibility; refused by the loop when enabled)
    p.add_argument("--guidance", action="store_true")
    p.add_argument("--g_loss", type=str, default="w_mse", choices=["mse", "w_mse"])
    p.add_argument("--g_scale", type=float, default=0.0)
    # common
    p.add_argument("--input", type=str, required=True)
    p.add_argument("--n_samples", type=int, default=1)
    p.add_argument("--output", type=str, required=True)
    p.add_argument("--seed", type=int, default=231)
    p.add_argument("--device", type=str, default="cuda", choices=["cpu", "cuda", "mps"])
    p.add_argument("--precision", type=str, default="fp16", choices=["fp32", "fp16", "bf16"])
    p.add_argument("--llava_bit", type=str, default="4", choices=["16", "8", "4"])
    return p


def parse_args(argv=None) -> Namespace:
    return build_parser().parse_args(argv)


def set_seed(seed: int) -> None:
    """what accelerate.utils.set_seed does (reference inference.py:293): python, numpy and torch (all devices)."""
    import random

    import numpy as np
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def main(argv=None):
    args = parse_args(argv)
    args.device = check_device(args.device)
    set_seed(args.seed)
    from diffbir.inference import BFRInferenceLoop, BIDInferenceLoop, BSRInferenceLoop, CustomInferenceLoop
    if args.version == "custom":
        CustomInferenceLoop(args).run()
        print("done!")
        return
    loops = {"sr": BSRInferenceLoop, "face": BFRInferenceLoop, "denoise": BIDInferenceLoop}
    if args.task not in loops:
        raise SystemExit(f"--task {args.task}: the RetinaFace detection / alignment front-end is outside this engine's "
                         "scope; supported: sr, face, denoise")
    loops[args.task](args).run()
    print("done!")


if __name__ == "__main__":
    main()
