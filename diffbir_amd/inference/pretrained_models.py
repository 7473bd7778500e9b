"""Checkpoint registry of the reference CLI (reference diffbir/inference/pretrained_models.py:34-51): the same keys and
URLs, because `load_model_from_url` caches / looks files up under `weights/<basename of the URL path>` exactly like the
reference does — a `weights/` directory populated for the reference works here unchanged."""
_HF1, _HF2 = "https://huggingface.co/lxq007/DiffBIR/resolve/main/", "https://huggingface.co/lxq007/DiffBIR-v2/resolve/main/"
_KAIR = "https://github.com/cszn/KAIR/releases/download/v1.0/"
MODELS = {
    # stage-1 cleaners
    "bsrnet": _KAIR + "BSRNet.pth",
    "swinir_face": _HF1 + "face_swinir_v1.ckpt",
    "scunet_psnr": _KAIR + "scunet_color_real_psnr.pth",
    "swinir_general": _HF1 + "general_swinir_v1.ckpt",
    "swinir_realesrgan": _HF2 + "realesrgan_s4_swinir_100k.pth",
    # Stable Diffusion 2.1 base (eps) / zero-terminal-SNR v-prediction fine-tune
    "sd_v2.1": "https://huggingface.co/stabilityai/stable-diffusion-2-1-base/resolve/main/v2-1_512-ema-pruned.ckpt",
    "sd_v2.1_zsnr": _HF2 + "sd2.1-base-zsnr-laionaes5.ckpt",
    # IRControlNet
    "v1_face": _HF2 + "v1_face.pth",
    "v1_general": _HF2 + "v1_general.pth",
    "v2": _HF2 + "v2.pth",
    "v2.1": _HF2 + "DiffBIR_v2.1.pt",
}
