"""Host-side helpers mirroring reference diffbir/utils/common.py (names / semantics kept: SURVEY.md §8b)."""
import importlib
import os
from typing import Any, Callable, List, Mapping, Optional, Tuple

import numpy as np
import torch

from .. import ops

T = torch.Tensor


def get_obj_from_str(string: str) -> Any:
    module, cls = string.rsplit(".", 1)
    if module == "diffbir.model" or module.startswith("diffbir.model."):  # reference YAMLs load unchanged
        module = "diffbir_amd.model" + module[len("diffbir.model"):]
    return getattr(importlib.import_module(module), cls)


def instantiate_from_config(config: Mapping[str, Any]) -> Any:
    """reference utils/common.py:23-26 (YAML `target:` / `params:` trees)."""
    if "target" not in config:
        raise KeyError("Expected key `target` to instantiate.")
    return get_obj_from_str(config["target"])(**config.get("params", dict()))


def load_file_from_url(url: str, model_dir: str = "weights", file_name: Optional[str] = None) -> str:
    """reference utils/common.py:92-110: cache location `<model_dir>/<basename of the URL path>`; downloads when the
    file is missing (needs network access — the target environment has none, so a clear error is raised instead)."""
    from urllib.parse import urlparse
    os.makedirs(model_dir, exist_ok=True)
    filename = file_name if file_name is not None else os.path.basename(urlparse(url).path)
    cached_file = os.path.abspath(os.path.join(model_dir, filename))
    if not os.path.exists(cached_file):
        print(f'Downloading: "{url}" to {cached_file}\n')
        try:
            torch.hub.download_url_to_file(url, cached_file, hash_prefix=None, progress=True)
        except Exception as e:
            raise FileNotFoundError(f"{cached_file} is missing and {url} could not be downloaded ({e!r}); place the "
                                    f"checkpoint at that path") from e
    return cached_file


def load_model_from_url(url_or_path: str) -> dict:
    """reference utils/common.py:113-120: URL (cached under `weights/`) or, as an engine extension, a local path;
    unwraps `state_dict` and strips a `module.` prefix."""
    is_url = "://" in url_or_path
    sd_path = load_file_from_url(url_or_path, model_dir="weights") if is_url else url_or_path
    if not os.path.exists(sd_path):
        raise FileNotFoundError(f"{sd_path}: checkpoint not found")
    sd = torch.load(sd_path, map_location="cpu", weights_only=False)
    if "state_dict" in sd:
        sd = sd["state_dict"]
    if list(sd.keys())[0].startswith("module"):
        sd = {k[len("module."):]: v for k, v in sd.items()}
    return sd


def sliding_windows(h: int, w: int, tile_size: int, tile_stride: int) -> List[Tuple[int, int, int, int]]:
    """reference utils/common.py:123-138: regular grid plus an edge-flush window when not aligned."""
    his = list(range(0, h - tile_size + 1, tile_stride))
    if (h - tile_size) % tile_stride != 0:
        his.append(h - tile_size)
    wis = list(range(0, w - tile_size + 1, tile_stride))
    if (w - tile_size) % tile_stride != 0:
        wis.append(w - tile_size)
    return [(hi, hi + tile_size, wi, wi + tile_size) for hi in his for wi in wis]


def gaussian_weights(tile_width: int, tile_height: int) -> np.ndarray:
    """reference utils/common.py:142-169 — x midpoint (w-1)/2 but y midpoint h/2 (asymmetric; reproduced)."""
    var = 0.01
    xm = (tile_width - 1) / 2
    xs = np.array([np.exp(-(x - xm) * (x - xm) / (tile_width * tile_width) / (2 * var)) / np.sqrt(2 * np.pi * var)
                   for x in range(tile_width)])
    ym = tile_height / 2
    ys = np.array([np.exp(-(y - ym) * (y - ym) / (tile_height * tile_height) / (2 * var)) / np.sqrt(2 * np.pi * var)
                   for y in range(tile_height)])
    return np.outer(ys, xs)


def make_tiled_fn(fn: Callable, size: int, stride: int, scale_type: str = "up", scale: int = 1,
                  channel: Optional[int] = None, weight: str = "gaussian", dtype=None, device=None,
                  progress: bool = True) -> Callable:
    """reference utils/common.py:172-232 for generic f32 NCHW callables (used for the tiled cleaner).
    All tiles are gathered by one kernel, `fn` runs per tile (its batch is the image batch, as in the reference) and
    one kernel does the weighted accumulate + normalise in the reference's tile order.  The diffusion model uses
    the batched scheduler in utils/tiling.py instead."""
    if scale_type != "up" or int(scale) != scale or scale < 1:
        raise NotImplementedError("tiling with an integer up-scale (1: SwinIR / SCUNet / ControlLDM, 4: BSRNet) is on "
                                  "the hot path; the reference never uses scale_type='down'")
    scale = int(scale)

    def tiled_fn(x: T, *args, **kwargs) -> T:
        b, c, h, w = x.shape
        wins = sliding_windows(h, w, size, stride)
        coords = torch.tensor([[hi, wi] for hi, _, wi, _ in wins], dtype=torch.int32, device=x.device)
        so = size * scale   # output tiles, weights and paste positions are those of the up-scaled grid (common.py:183-222)
        wt = gaussian_weights(so, so) if weight == "gaussian" else np.ones((so, so))
        wt = torch.tensor(wt, dtype=torch.float32, device=x.device)
        tiles = ops.tile_gather(x.float().contiguous(), coords, size)
        outs = []
        for t, (hi, he, wi, we) in enumerate(wins):
            kw = dict(kwargs)
            if len(args) or len(kwargs):
                kw.update(dict(hi=hi, hi_end=he, wi=wi, wi_end=we))
            outs.append(fn(tiles[t * b:(t + 1) * b], *args, **kw).float())
        return ops.tile_accumulate(torch.cat(outs, dim=0).contiguous(), wt, coords * scale, b, h * scale, w * scale)

    return tiled_fn


def wavelet_reconstruction(content_feat: T, style_feat: T, levels: int = 5) -> T:
    """reference utils/common.py:29-77: high frequencies of `content` + low frequencies of `style`.
    The reference's running sum  sum_i (img_i - low_i)  telescopes to img_0 - low_last."""
    def low(img: T) -> T:
        for i in range(levels):
            img = ops.wavelet_blur(img, 2 ** i)
        return img

    c = content_feat.float().contiguous()
    s = style_feat.float().contiguous()
    return ops.colorfix(c, low(c), low(s))


# ---- the remaining helpers of reference utils/common.py that user code imports (metrics, wavelet pieces, monitors) ------
def wavelet_blur(image: T, radius: int) -> T:
    """reference utils/common.py:29-48: depthwise [1,2,1]x[1,2,1]/16 blur with dilation `radius`, replicate padding —
    the engine's `dbir_wavelet_blur` kernel (f32 NCHW in / out)."""
    return ops.wavelet_blur(image.float().contiguous(), radius)


def wavelet_decomposition(image: T, levels: int = 5) -> Tuple[T, T]:
    """reference utils/common.py:51-63 -> (high frequencies, low frequencies); the reference's running sum of
    (img_i - low_i) telescopes to image - low_last."""
    img = image.float().contiguous()
    low = img
    for i in range(levels):
        low = ops.wavelet_blur(low, 2 ** i)
    return img - low, low


def to(obj: Any, device) -> Any:
    """reference utils/common.py:300-310: move every tensor of a nested dict / tuple / list to `device`."""
    if torch.is_tensor(obj):
        return obj.to(device)
    if isinstance(obj, dict):
        return {k: to(v, device) for k, v in obj.items()}
    if isinstance(obj, (tuple, list)):
        return type(obj)(to(v, device) for v in obj)
    return obj


_BT601 = ((65.481, -37.797, 112.0), (128.553, -74.203, -93.786), (24.966, 112.0, -18.214))


def rgb2ycbcr_pt(img: T, y_only: bool = False) -> T:
    """reference utils/common.py:314-346 (ITU-R BT.601, studio swing): RGB [n,3,h,w] in [0,1] -> Y or YCbCr in [0,1].
    Metric code, not on the restoration path: plain torch."""
    m = torch.tensor(_BT601, dtype=img.dtype, device=img.device)
    off = torch.tensor((16.0, 128.0, 128.0), dtype=img.dtype, device=img.device)
    if y_only:
        m, off = m[:, :1], off[:1]
    out = torch.einsum("nchw,ck->nkhw", img, m) + off.view(1, -1, 1, 1)
    return out / 255.0


def calculate_psnr_pt(img: T, img2: T, crop_border: int, test_y_channel: bool = False) -> T:
    """reference utils/common.py:350-390: per-image PSNR of [0,1] images (optionally on the Y channel, borders cropped),
    the metric BASELINE.json's parity tolerance is stated in."""
    assert img.shape == img2.shape, f"Image shapes are different: {img.shape}, {img2.shape}."
    if crop_border != 0:
        img = img[:, :, crop_border:-crop_border, crop_border:-crop_border]
        img2 = img2[:, :, crop_border:-crop_border, crop_border:-crop_border]
    if test_y_channel:
        img, img2 = rgb2ycbcr_pt(img, y_only=True), rgb2ycbcr_pt(img2, y_only=True)
    mse = ((img.double() - img2.double()) ** 2).mean(dim=[1, 2, 3])
    return 10.0 * torch.log10(1.0 / (mse + 1e-8))


TRACE_VRAM = int(os.environ.get("TRACE_VRAM", False))


class VRAMPeakMonitor:
    """reference utils/common.py:260-279: context manager printing the HBM peak before / after a block when TRACE_VRAM=1."""

    def __init__(self, tag: str) -> None:
        self.tag = tag

    def __enter__(self):
        self.peak_before = torch.cuda.max_memory_allocated() / 1024 ** 3 if torch.cuda.is_available() else 0.0
        return self

    def __exit__(self, exc_type, exc_value, traceback):
        if torch.cuda.is_available():
            torch.cuda.synchronize()
            peak_after = torch.cuda.max_memory_allocated() / 1024 ** 3
            if TRACE_VRAM:
                print(f"\033[93mVRAM peak before {self.tag}: {self.peak_before:.2f} GB, after: {peak_after:.2f} GB\033[0m")
        return False


def trace_vram_usage(tag: str) -> Callable:
    """reference utils/common.py:236-257: decorator form of VRAMPeakMonitor (identity unless TRACE_VRAM=1)."""
    def deco(func: Callable) -> Callable:
        if not TRACE_VRAM:
            return func

        def wrapped(*args, **kwargs):
            with VRAMPeakMonitor(tag):
                return func(*args, **kwargs)
        return wrapped
    return deco
