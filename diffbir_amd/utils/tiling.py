"""Mixture-of-diffusers tiled-patch scheduler for the diffusion network (reference utils/common.py:172-232 as
used by spaced_sampler.py:204-219 / dpms_sampler.py:56-71).

The reference evaluates the T tiles of one model evaluation sequentially at batch B.  Tiles are independent
(GroupNorm / LayerNorm are per sample), so the engine instead
  1. gathers all windows of the latent with ONE kernel (`dbir_tile_gather`) into a tile-major batch [T*B,4,ts,ts]
     (the overlapping windows are read once from HBM/L2, not once per tile launch),
  2. evaluates them as large batches (chunks of `max_batch` samples) -> MFMA tiles stay full at every UNet level,
  3. blends with ONE kernel (`dbir_tile_accumulate`) that visits tiles in increasing index per output pixel,
     i.e. the reference's sequential f32 accumulation order, then divides by the summed weights.
With `shard=(rank, world)` each rank evaluates tiles rank::world (round-robin: every rank gets tiles from all image
regions, so ragged edge windows spread evenly), accumulates its un-normalised partial sum with
`dbir_tile_accumulate_partial`, and `all_reduce` (RCCL sum over xGMI, one [B,4,H/8,W/8] f32 tensor per evaluation —
classifier-free guidance is batched inside the evaluation, so one reduction per sampler step) combines them before
`dbir_tile_normalize` divides by the (input-independent, cached) summed weights; SURVEY.md §8e,
diffbir_amd/parallel.py.  The parallel reduction changes the f32 summation order with respect to the reference's
sequential loop: parity there is a tolerance (tests), not bit-exactness.
"""
from typing import Callable, Dict, Optional, Tuple

import torch

from .. import ops
from .common import gaussian_weights, sliding_windows

T = torch.Tensor


class TiledModel:
    def __init__(self, forward: Callable, tile_size: int, tile_stride: int, max_batch: int = 32,
                 shard: Optional[Tuple[int, int]] = None, all_reduce: Optional[Callable] = None):
        self.forward, self.ts, self.stride, self.max_batch = forward, tile_size, tile_stride, max_batch
        if shard is not None and shard[1] > 1 and all_reduce is None:
            raise ValueError("tile sharding over >1 ranks needs an all_reduce callable (diffbir_amd.parallel)")
        # (one rank WITH an all_reduce = parallel's forced-collectives mode: the sharded path — partial sum, all-reduce over a
        #  one-rank communicator, normalise — runs as on N GPUs; same f32 operations in the same order as the plain path)
        self.shard = shard if (shard is not None and (shard[1] > 1 or all_reduce is not None)) else None
        self.all_reduce = all_reduce
        self._den: Dict[tuple, T] = {}
        self._coords: Dict[tuple, T] = {}
        self._cimg: Dict[tuple, T] = {}
        self._ctxt: Dict[tuple, T] = {}
        self._weights: Optional[T] = None

    def _get_coords(self, h, w, device) -> T:
        key = (h, w, str(device))
        if key not in self._coords:
            wins = sliding_windows(h, w, self.ts, self.stride)
            self._coords[key] = torch.tensor([[hi, wi] for hi, _, wi, _ in wins], dtype=torch.int32, device=device)
        return self._coords[key]

    def __call__(self, x: T, t: T, cond: Dict[str, T]) -> T:
        B, C, H, W = x.shape
        coords_all = self._get_coords(H, W, x.device)
        coords = coords_all
        if self.shard is not None:
            coords = coords_all[self.shard[0]::self.shard[1]].contiguous()
        Tn = coords.shape[0]
        if self._weights is None:
            self._weights = torch.tensor(gaussian_weights(self.ts, self.ts), dtype=torch.float32, device=x.device)
        if self.shard is not None:
            return self._sharded(x, t, cond, coords, coords_all)
        return ops.tile_accumulate(self._eval_tiles(x, t, cond, coords), self._weights, coords, B, H, W)

    def _sharded(self, x: T, t: T, cond: Dict[str, T], coords: T, coords_all: T) -> T:
        B, C, H, W = x.shape
        if coords.shape[0] > 0:
            num = ops.tile_accumulate_partial(self._eval_tiles(x, t, cond, coords), self._weights, coords, B, C, H, W)
        else:  # more ranks than tiles: this rank only takes part in the reduction
            num = torch.zeros((B, C, H, W), dtype=torch.float32, device=x.device)
        num = self.all_reduce(num)
        kd = (H, W, str(x.device))
        if kd not in self._den:
            self._den[kd] = ops.tile_accumulate_partial(None, self._weights, coords_all, 1, 1, H, W)
        return ops.tile_normalize(num.contiguous(), self._den[kd])

    def _eval_tiles(self, x: T, t: T, cond: Dict[str, T], coords: T) -> T:
        """Network output for the windows `coords` of x, tile-major [T*B, C, ts, ts]."""
        B = x.shape[0]
        Tn = coords.shape[0]
        c_img, c_txt = cond["c_img"], cond["c_txt"]
        kimg = (c_img.data_ptr(), tuple(c_img.shape), c_img._version)
        if kimg not in self._cimg:  # condition latent is constant over the sampling steps: gather its tiles once
            self._cimg = {kimg: ops.tile_gather(c_img.float().contiguous(), coords, self.ts)}
        n = Tn * B
        step = max(B, (self.max_batch // B) * B)
        # every chunk holds whole tiles, i.e. the same [B,77,D] context pattern: ONE repeated tensor serves all chunks
        # (same storage -> one cross-attention K/V cache entry and one captured HIP graph per chunk size)
        ktxt = (c_txt.data_ptr(), tuple(c_txt.shape), c_txt._version, step)
        if ktxt not in self._ctxt:
            self._ctxt = {ktxt: c_txt.repeat(min(step, n) // B, 1, 1).contiguous()}
        cimg_tiles, ctxt_rep = self._cimg[kimg], self._ctxt[ktxt]
        tiles = ops.tile_gather(x.float().contiguous(), coords, self.ts)
        t_rep = t.repeat(Tn)
        outs = []
        pair = cond.get("cfg_pair")  # (1, bs): x = [bs uncond || bs cond] with equal halves -> per tile the same holds
        for i in range(0, n, step):
            j = min(n, i + step)
            c = {"c_txt": ctxt_rep[:j - i], "c_img": cimg_tiles[i:j]}
            if pair is not None:
                c["cfg_pair"] = ((j - i) // B, pair[1])   # chunks hold whole tiles: groups of [bs || bs]
            outs.append(self.forward(tiles[i:j], t_rep[i:j], c))
        eps = outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)
        return eps.contiguous()
