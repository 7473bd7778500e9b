"""Multi-GPU execution of the restoration pipeline: one process per GPU, `torch.distributed` (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" on CPU for the tests).  SURVEY.md §8(e).

The reference is single-device (its only multi-GPU mechanism is `accelerate` in the training scripts); the hot path
shards naturally because every stage is per-sample (GroupNorm / LayerNorm are per-sample, no BatchNorm):

  * **batch sharding** (`run_data_parallel`): rank r restores the contiguous slice `shard_range(B, r, world)` of the
    batch.  No collective inside the sampling loop.  Weights: every rank loads / generates the same state dict, or
    rank 0 does and `broadcast_state_dict` ships it (RCCL broadcast, bucketed flat buffers).  Outputs: uint8
    [b_r, H, W, 3] slices are gathered to rank 0 (`gather_batch`; ragged slices are padded to the largest).
  * **noise parity**: the reference draws `x_T = randn((B,4,h,w))` and one `randn_like(x)` per step for the WHOLE batch
    on one generator (pipeline.py:159, spaced_sampler.py:181).  `ShardedNoise` makes every rank draw the full-batch
    tensor from an identically seeded generator, in the same order, and keep its rows — the restored images do not
    depend on the number of GPUs (64 KB per sample per step: cheaper than a broadcast).
  * **hybrid** (`hybrid_split` / `run_hybrid`): several large images on more GPUs than images — images over rank groups,
    tiles over the ranks of a group (one all-reduce per evaluation inside the group only).
  * **tile sharding** (`enable_tile_sharding`) for tiled sampling of large images: the T tiles of one network
    evaluation are split round-robin over the ranks and the partial weighted sums are combined with ONE all-reduce
    per evaluation (`diffbir_amd.utils.tiling.TiledModel`); every rank then performs the (cheap, deterministic given
    the same noise) sampler update redundantly, so no broadcast of x is needed.

Nothing here touches the kernels' C ABI: collectives are issued from Python on torch tensors.
"""
import os
from dataclasses import dataclass
from typing import Callable, Dict, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist


@dataclass
class DistContext:
    rank: int = 0
    world: int = 1
    device: torch.device = torch.device("cpu")
    group: Optional[object] = None
    _noise: Optional[Callable] = None   # persistent shared-seed noise stream of run_data_parallel
    _tree_seeds: Optional[torch.Generator] = None   # shared-seed source of the SDE solvers' Brownian-tree seeds
    force: bool = False                 # issue the collectives even with ONE rank (see `multi`)

    @property
    def is_main(self) -> bool:
        return self.rank == 0

    @property
    def multi(self) -> bool:
        """Do the collectives run?  world > 1, or a single rank with `force` (DBIR_FORCE_COLLECTIVES=1 /
        init_distributed(force=True)): the 1-GPU box then executes the very RCCL calls an 8-GPU node makes — communicator
        init, bucketed broadcast, all-reduce per evaluation, gather — on a one-rank communicator (library load, stream
        ordering, IPC set-up; results must be bit-identical to the plain single-process run: tests/test_multigpu_gpu.py)."""
        return self.world > 1 or self.force


# collectives issued through this module since import (bench.py reports them: `rccl.calls`)
calls = dict(broadcast=0, all_reduce=0, gather=0, new_group=0)


def init_distributed(backend: Optional[str] = None, device: Optional[torch.device] = None,
                     force: Optional[bool] = None) -> DistContext:
    """Join the process group described by RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT (torchrun).
    backend None -> "nccl" (RCCL) when a GPU is visible, else "gloo".  world == 1 needs no process group — unless `force`
    (default: env DBIR_FORCE_COLLECTIVES=1), which creates the one-rank group and makes every collective below run."""
    if force is None:
        force = os.environ.get("DBIR_FORCE_COLLECTIVES", "0") == "1"
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if device is None:
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
            device = torch.device("cuda", local)
        else:
            device = torch.device("cpu")
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            #  a default port only for the forced ONE-rank group, and a free one (two such jobs on a host must not
            # collide); a multi-rank launch that forgot MASTER_PORT fails loudly instead of meeting strangers on a shared default
            if world > 1:
                raise RuntimeError("init_distributed: WORLD_SIZE > 1 needs MASTER_PORT (launch through torch.distributed.run)")
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        be = backend or ("nccl" if device.type == "cuda" else "gloo")
        kw = dict(device_id=device) if be == "nccl" else {}
        dist.init_process_group(be, rank=rank, world_size=world, **kw)
    return DistContext(rank, world, device, force=bool(force))


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced slice [lo, hi) of n items for `rank` (the first n % world ranks get one extra)."""
    per, extra = divmod(n, world)
    lo = rank * per + min(rank, extra)
    return lo, lo + per + (1 if rank < extra else 0)


class ShardedNoise:
    """Full-batch Gaussian draws (reference order) -> this rank's batch rows.

    `base(shape)` must return identical values on every rank (an identically seeded torch.Generator, e.g.
    oracle.cases.NoiseStream in tests or `seeded(seed, device)` below).  A requested shape [b_r, ...] whose leading
    dim equals this rank's slice length is drawn as [B, ...] and sliced; any other shape passes through."""

    def __init__(self, base: Callable, batch: int, lo: int, hi: int):
        self.base, self.batch, self.lo, self.hi = base, batch, lo, hi

    def __call__(self, shape) -> torch.Tensor:
        shape = tuple(shape)
        if len(shape) >= 1 and shape[0] == self.hi - self.lo:
            return self.base((self.batch,) + shape[1:])[self.lo:self.hi].contiguous()
        return self.base(shape)

    @staticmethod
    def seeded(seed: int, device) -> Callable:
        g = torch.Generator(device=device)
        g.manual_seed(seed)
        return lambda shape: torch.randn(tuple(shape), generator=g, dtype=torch.float32, device=device)


class ShardedBrownianTree:
    """The SDE solvers' Brownian tree (sampler/brownian.py) for a batch-sharded run: the tree is built for the FULL batch
    from a seed every rank shares and each rank keeps its rows, so the restored images do not depend on the number of
    GPUs (the per-node draws have the full-batch shape, as in a 1-GPU run of the whole batch)."""

    def __init__(self, x_local: torch.Tensor, sigma_min: float, sigma_max: float, batch: int, lo: int, hi: int, seed: int,
                 noise_device=None):
        from .sampler.brownian import BrownianTreeNoise
        full = torch.empty((batch,) + tuple(x_local.shape[1:]), dtype=torch.float32, device=x_local.device)
        self.tree, self.lo, self.hi = BrownianTreeNoise(full, sigma_min, sigma_max, seed, noise_device=noise_device), lo, hi

    def __call__(self, sigma: float, sigma_next: float) -> torch.Tensor:
        return self.tree(sigma, sigma_next)[self.lo:self.hi].contiguous()


def all_reduce_sum(ctx: DistContext) -> Callable:
    """tensor -> tensor summed over ranks (in place; RCCL all-reduce on the tensor's device)."""

    def fn(t: torch.Tensor) -> torch.Tensor:
        if ctx.multi:
            calls["all_reduce"] += 1
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=ctx.group)
        return t

    return fn


def broadcast_state_dict(sd: Optional[Dict[str, torch.Tensor]], spec, ctx: DistContext, src: int = 0,
                         bucket_bytes: int = 256 << 20) -> Dict[str, torch.Tensor]:
    """Ship a state dict from `src` to every rank.  `spec`: the module's OrderedDict key -> (shape, kind)
    (model/specs.py) so receivers know names / shapes without a handshake.  Tensors travel as f32 in flat buckets of
    `bucket_bytes` (few large RCCL broadcasts instead of ~3000 small ones: xGMI links are point-to-point, per-message
    latency dominates small transfers)."""
    if not ctx.multi:
        return sd
    keys = [k for k, (_, kind) in spec.items() if kind != "buf"]
    out: Dict[str, torch.Tensor] = {}
    i = 0
    while i < len(keys):
        j, n = i, 0
        while j < len(keys) and (n == 0 or (n + int(np.prod(spec[keys[j]][0]))) * 4 <= bucket_bytes):
            n += int(np.prod(spec[keys[j]][0]))
            j += 1
        flat = torch.empty(n, dtype=torch.float32, device=ctx.device)
        if ctx.rank == src:
            off = 0
            for k in keys[i:j]:
                m = int(np.prod(spec[k][0]))
                flat[off:off + m] = sd[k].reshape(-1).to(device=ctx.device, dtype=torch.float32)
                off += m
        calls["broadcast"] += 1
        dist.broadcast(flat, src=src, group=ctx.group)
        off = 0
        for k in keys[i:j]:
            m = int(np.prod(spec[k][0]))
            out[k] = flat[off:off + m].reshape(tuple(spec[k][0])).clone()
            off += m
        i = j
    return out


def gather_batch(local: np.ndarray, batch: int, ctx: DistContext, dst: int = 0) -> Optional[np.ndarray]:
    """uint8 [b_r, H, W, 3] slices -> [B, H, W, 3] on rank `dst` (None elsewhere)."""
    if not ctx.multi:
        return local
    bmax = shard_range(batch, 0, ctx.world)
    bmax = bmax[1] - bmax[0]
    t = torch.zeros((bmax,) + local.shape[1:], dtype=torch.uint8, device=ctx.device)
    t[: local.shape[0]] = torch.as_tensor(local).to(ctx.device)
    bufs = [torch.empty_like(t) for _ in range(ctx.world)] if ctx.rank == dst else None
    calls["gather"] += 1
    dist.gather(t, bufs, dst=dst, group=ctx.group)
    if ctx.rank != dst:
        return None
    parts = []
    for r, b in enumerate(bufs):
        lo, hi = shard_range(batch, r, ctx.world)
        parts.append(b[: hi - lo].cpu().numpy())
    return np.concatenate(parts, axis=0)


def enable_tile_sharding(pipe, ctx: DistContext, seed: Optional[int] = 231, check_every: int = 0) -> None:
    """Tiled sampling evaluates tiles rank::world on this rank and all-reduces the partial sums; the tiled VAE of
    `pipe.cldm` (model/vae.py `tile_shard`) is sharded over the same ranks.

    Every rank performs the sampler update redundantly, so every rank MUST draw the same x_T and per-step noise: unless
    the caller already installed a shared noise source (`pipe.randn`), an identically seeded device generator is
    installed here (`seed=None` leaves the pipeline alone — then the caller is responsible).  `check_every = n > 0`
    additionally all-reduces a checksum of the blended prediction every n evaluations and raises if the ranks' inputs
    have diverged (debug aid: one extra 8-byte all-reduce)."""
    vae = getattr(getattr(pipe, "cldm", None), "vae", None)
    if not ctx.multi:
        pipe.tile_shard, pipe.tile_all_reduce = None, None
        if vae is not None:
            vae.tile_shard, vae.tile_all_reduce = None, None
        return
    pipe.tile_shard = (ctx.rank, ctx.world)
    reduce = all_reduce_sum(ctx)
    if vae is not None:
        # the tiled VAE (--vae_encoder_tiled / --vae_decoder_tiled) shards its tiles the same way: per GroupNorm layer one
        # all-reduce of the tile-averaged statistics (2 * 32 floats per sample), one all-reduce of the pasted result
        vae.tile_shard, vae.tile_all_reduce = (ctx.rank, ctx.world), reduce
    if check_every > 0:
        state = {"n": 0}

        def checked(t: torch.Tensor) -> torch.Tensor:
            t = reduce(t)
            state["n"] += 1
            if state["n"] % check_every == 0:
                s = t.double().sum().reshape(1)
                lo, hi = s.clone(), s.clone()
                dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=ctx.group)
                dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=ctx.group)
                if not torch.equal(lo, hi):
                    raise RuntimeError("tile sharding: ranks hold different reduced tensors — were x_T / the step noise "
                                       "drawn from differently seeded generators?")
            return t
        pipe.tile_all_reduce = checked
    else:
        pipe.tile_all_reduce = reduce
    if seed is not None and getattr(pipe, "randn", None) is None:
        pipe.randn = ShardedNoise.seeded(seed, ctx.device)
    if seed is not None and getattr(pipe, "brownian", False) is None:
        pipe.brownian = shared_seed_brownian(seed)   # the SDE solvers' tree seed must be the same on every rank too


def shared_seed_brownian(seed: int) -> Callable:
    """Brownian-tree factory (sampler/edm_sampler.py `brownian`) whose tree seeds come from a CPU generator seeded with
    `seed` instead of the process's global one (k_diffusion.py:78-79 draws from the latter): identical on every rank."""
    from .sampler.brownian import BrownianTreeNoise
    g = torch.Generator().manual_seed(seed)
    return lambda x, randn, smin, smax: BrownianTreeNoise(
        x, smin, smax, seed=int(torch.randint(0, 2 ** 63 - 1, [], generator=g).item()))


def hybrid_split(ctx: DistContext, n_images: int) -> Tuple[DistContext, int, int]:
    """Images over rank GROUPS x tiles within a group (BASELINE config C5 = 4 images of 4096x4096 on 8 GPUs
    -> 4 groups of 2 ranks; each group restores one image with its tiles sharded over the group's ranks).

    G = gcd(n_images, world) groups of S = world / G consecutive ranks (neighbours on the xGMI ring); group g owns the
    images [g * n_images / G, (g + 1) * n_images / G).  Returns (sub-context of this rank's group: rank / world / process
    group for `enable_tile_sharding`, first image, one-past-last image).  Every rank must call this (it creates the
    sub-groups collectively).  world == 1 or G == world (one rank per group) need no sub-groups."""
    import math
    G = math.gcd(max(n_images, 1), ctx.world)
    S = ctx.world // G
    g, r = ctx.rank // S, ctx.rank % S
    per = n_images // G
    group = None
    if (ctx.world > 1 and S > 1) or ctx.force:
        for gi in range(G):   # new_group is collective: every rank creates every group, keeps its own
            calls["new_group"] += 1
            h = dist.new_group(list(range(gi * S, (gi + 1) * S)))
            if gi == g:
                group = h
    return DistContext(r, S, ctx.device, group, force=ctx.force), g * per, (g + 1) * per


def gather_group_outputs(local: Optional[np.ndarray], n_images: int, ctx: DistContext, sub: DistContext,
                         dst: int = 0) -> Optional[np.ndarray]:
    """After a hybrid run every rank of a group holds the group's restored images: the group leaders' slices -> rank
    `dst` in image order (RCCL gather over the WORLD group; the other ranks of a group contribute nothing)."""
    if not ctx.multi:
        return local
    G = ctx.world // sub.world
    per = n_images // G
    t = torch.as_tensor(local).to(ctx.device).contiguous()
    bufs = [torch.empty_like(t) for _ in range(ctx.world)] if ctx.rank == dst else None
    calls["gather"] += 1
    dist.gather(t, bufs, dst=dst, group=ctx.group)
    if ctx.rank != dst:
        return None
    return np.concatenate([bufs[gi * sub.world][:per].cpu().numpy() for gi in range(G)], axis=0)


def run_hybrid(pipe, lq: np.ndarray, ctx: DistContext, run_args: tuple, noise_for_image: Optional[Callable] = None,
               gather: bool = True, split=None):
    """Tiled restoration of a batch of large images on `ctx.world` GPUs: images over groups, tiles within a group; a
    group restores its images ONE AT A TIME (a 4096x4096 image alone takes ~25 GB of activations on one MI355X).
    `noise_for_image(i)` -> the noise source (shape -> f32 tensor) of GLOBAL image i, identical on every rank: image i's
    result depends neither on the GPU count nor on which group restores it.  Default: a device generator seeded
    231 + i.  `split`: the (sub, lo, hi) of an earlier `hybrid_split` (sub-groups are created once)."""
    B = lq.shape[0]
    sub, lo, hi = split if split is not None else hybrid_split(ctx, B)
    if noise_for_image is None:
        noise_for_image = lambda i: ShardedNoise.seeded(231 + i, ctx.device)  # noqa: E731
    prev, prev_b = pipe.randn, getattr(pipe, "brownian", None)
    # the sharding state the pipeline / its VAE had before this call comes back afterwards: a later
    # pipe.run or run_data_parallel on the same pipeline must not all-reduce over this call's sub-group
    vae = getattr(getattr(pipe, "cldm", None), "vae", None)
    saved = (getattr(pipe, "tile_shard", None), getattr(pipe, "tile_all_reduce", None),
             getattr(vae, "tile_shard", None), getattr(vae, "tile_all_reduce", None))
    enable_tile_sharding(pipe, sub, seed=None)
    outs = []
    try:
        for i in range(lo, hi):
            pipe.randn = noise_for_image(i)
            if prev_b is None:
                pipe.brownian = shared_seed_brownian(231 + i)   # per GLOBAL image, like its noise
            outs.append(pipe.run(lq[i:i + 1], *run_args))
    finally:
        pipe.randn, pipe.brownian = prev, prev_b
        pipe.tile_shard, pipe.tile_all_reduce = saved[0], saved[1]
        if vae is not None:
            vae.tile_shard, vae.tile_all_reduce = saved[2], saved[3]
    out = np.concatenate(outs, axis=0) if outs else np.zeros((0,) + tuple(lq.shape[1:]), dtype=np.uint8)
    if not gather:
        return out
    return gather_group_outputs(out, B, ctx, sub)


def run_data_parallel(pipe, lq: np.ndarray, ctx: DistContext, run_args: tuple, noise: Optional[Callable] = None,
                      gather: bool = True) -> Optional[np.ndarray]:
    """`pipe.run(lq, *run_args)` with the batch dimension of `lq` sharded over the ranks.

    lq: the FULL uint8 batch [B,H,W,3] on every rank (or only meaningful on this rank's rows).  `noise`: full-batch
    noise source shared by all ranks (identically seeded); None -> a device generator seeded with 231 (the
    reference's default seed).  Returns the full restored batch on rank 0 (None on other ranks) when `gather`, else
    this rank's slice."""
    B = lq.shape[0]
    lo, hi = shard_range(B, ctx.rank, ctx.world)
    if noise is None:
        # ONE generator per context, seeded once: successive batches keep drawing from the same stream (the reference
        # seeds once per process, inference.py:293), instead of replaying the same noise for every batch
        if getattr(ctx, "_noise", None) is None:
            ctx._noise = ShardedNoise.seeded(231, ctx.device)
        noise = ctx._noise
    base = noise
    prev, prev_b = pipe.randn, pipe.brownian
    pipe.randn = ShardedNoise(base, B, lo, hi) if ctx.multi else base
    if prev_b is None:
        # the SDE solvers' Brownian tree: one seed per call from a generator every rank seeds alike (drawn whether or not
        # this call uses an SDE solver or this rank has rows, so the ranks stay in step), full-batch tree, this rank's rows
        if ctx._tree_seeds is None:
            ctx._tree_seeds = torch.Generator().manual_seed(231)
        seed = int(torch.randint(0, 2 ** 63 - 1, [], generator=ctx._tree_seeds).item())
        pipe.brownian = lambda x, randn, smin, smax: ShardedBrownianTree(x, smin, smax, B, lo, hi, seed)
    try:
        if hi > lo:
            out = pipe.run(lq[lo:hi], *run_args)
        else:  # more ranks than images
            out = np.zeros((0,) + tuple(lq.shape[1:]), dtype=np.uint8)
    finally:
        pipe.randn, pipe.brownian = prev, prev_b
    if not gather:
        return out
    return gather_batch(out, B, ctx)
