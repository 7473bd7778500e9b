#!/usr/bin/env python
"""bench.py — restored images/sec @512x512, 50-step SpacedSampler + CFG (BASELINE.json metric / configs[1]).

One "step" = one full pass of the hot path over one batch: uint8 LQ batch [8,512,512,3] (resident in HBM) ->
SwinIR -> VAE encode -> 50 spaced-DDPM steps over ControlNet+UNet with CFG 4.0 -> VAE decode -> wavelet colour
fix -> uint8.  fp16 MFMA compute, synthetic seeded inputs, random-init weights of the real architecture
(no checkpoints / datasets are reachable here).  N GPUs = N data-parallel replicas (one process per GPU, launched
by torch.distributed.run), each restoring its own batch of 8 (weak scaling, no collective on the data path);
timing = barrier + synchronize on both sides, max over ranks.

Prints ONE JSON line (rank 0) with the driver contract fields plus
  roofline     — dominant kernel family (implicit-GEMM MFMA kernel): algorithmic FLOPs / measured launch time
  cpu_baseline — the oracle (CPU fp32 restatement of the reference, "port") timed on a bounded sample on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOPS_PER_IMAGE = 111.2e12        # C2 image, duplicate VAE encode removed (algorithmic minimum)
MFMA_PEAK = 2.5e15                # dense fp16/bf16, MI355X_MICROARCH.md
BATCH = 8
NEG = "low quality, blurry, low-resolution, noisy, unsharp, weird textures"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2, help="timed pipeline passes (batches of 8 images)")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--sampler-steps", type=int, default=50)
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--selftest", action="store_true",
                    help="CPU / gloo dry run of the launch, barrier, max-over-ranks and JSON plumbing with a fake "
                         "workload (tests/test_bench_plumbing_cpu.py); never a measurement")
    return ap.parse_args()


def build_engine(device, dtype):
    """Full-size DiffBIR v2.1 networks with random weights generated directly on the GPU."""
    import torch
    from diffbir_amd import configs
    from diffbir_amd.model import ControlLDM, Diffusion, SwinIR
    from diffbir_amd.model import specs
    from diffbir_amd.pipeline import SwinIRPipeline
    g = torch.Generator(device=device).manual_seed(1234)

    def rand_sd(spec):
        sd = {}
        for k, (shp, kind) in spec.items():
            if kind == "buf":
                continue
            if kind == "w":
                fan = 1
                for s in shp[1:]:
                    fan *= s
                sd[k] = torch.randn(shp, device=device, generator=g) * (max(fan, 1) ** -0.5)
            elif kind == "g":
                sd[k] = 1.0 + 0.05 * torch.randn(shp, device=device, generator=g)
            else:
                sd[k] = 0.02 * torch.randn(shp, device=device, generator=g)
        return sd

    cldm_cfg, swin_cfg = configs.get("FULL_CLDM"), configs.get("FULL_SWINIR")
    cldm = ControlLDM(**cldm_cfg)
    for name, mod in (("unet", cldm.unet), ("controlnet", cldm.controlnet), ("vae", cldm.vae), ("clip", cldm.clip)):
        mod.load_state_dict(rand_sd(mod._spec), strict=True)
    swin = SwinIR(**swin_cfg)
    swin.load_state_dict(rand_sd(swin._spec), strict=True)
    cldm.to(device)
    swin.to(device)
    cldm.cast_dtype(dtype)
    swin.set_dtype(dtype)
    for m in (cldm.unet, cldm.controlnet, cldm.vae, swin, cldm.clip):
        m.release_master()
    diff = Diffusion(**configs.get("DIFFUSION_V21"))
    return SwinIRPipeline(swin, cldm, diff, None, str(device)), cldm, swin


def run_once(pipe, lq, sampler_steps):
    return pipe.run(lq, sampler_steps, 1.0, False, 512, 256, False, 256, False, 256, False, 512, 256, "", NEG, 4.0,
                    "noise", "spaced", 0, False, 0, 0, 300, 1, 1, 1)


def measure_roofline(cldm, device, batch):
    """Per-launch HIP-event timing (events recorded on torch's current stream = the stream the kernels are launched
    on) of every implicit-GEMM launch of ONE batched network evaluation (ControlNet + UNet at batch 2B)."""
    import torch
    from diffbir_amd import ops
    x = torch.randn(2 * batch, 4, 64, 64, device=device)
    c_img = torch.randn(2 * batch, 4, 64, 64, device=device)
    c_txt = torch.randn(2 * batch, 77, 1024, device=device)
    t = torch.full((2 * batch,), 500.0, device=device)
    cond = dict(c_txt=c_txt, c_img=c_img)
    overlap, cldm.overlap_streams = cldm.overlap_streams, False   # per-launch durations are measured un-overlapped
    cldm(x, t, cond)  # warm (context K/V cache, allocator)
    torch.cuda.synchronize()
    prof = ops.start_profile()
    cldm(x, t, cond)
    torch.cuda.synchronize()
    rec = ops.stop_profile()
    cldm.overlap_streams = overlap
    tot = {}
    for kind, flops, e0, e1, _tag, nbytes in rec:
        ms = e0.elapsed_time(e1)
        a = tot.setdefault(kind, [0.0, 0.0, 0, 0.0])
        a[0] += flops
        a[1] += ms * 1e-3
        a[2] += 1
        a[3] += nbytes
    g = tot.get("gemm", [0.0, 1.0, 1, 0.0])
    out = dict(bound="mfma", kernel="implicit-GEMM conv/linear family (gemm_glds_kernel / gemm_ph_kernel / gemm_kernel)",
               achieved=g[0] / g[1] / 1e12, peak=MFMA_PEAK / 1e12, unit="TFLOP/s", frac=g[0] / g[1] / MFMA_PEAK,
               traffic=None, launches=g[2], flops_per_eval=g[0], seconds_per_eval=g[1],
               avg_launch_us=g[1] / g[2] * 1e6, flops_per_launch=g[0] / g[2],
               algorithmic_bytes_per_launch=g[3] / g[2])
    # HBM bytes per GEMM launch from the rocprofv3 PMC passes (tools/pmc_traffic.sh -> profiles/; FETCH_SIZE doubled
    # per the gfx950 correction in MI355X_MICROARCH.md, + WRITE_SIZE), collected on the same network evaluation
    pmc = os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")
    if os.path.exists(pmc):
        with open(pmc) as f:
            t = json.load(f)
        if t.get("gemm_bytes_per_eval"):
            out["traffic"] = t["gemm_bytes_per_eval"] / g[2]   # HBM bytes per logical GEMM launch (incl. split-K slabs)
        out["traffic_source"] = "profiles/r1_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"
    if "attention" in tot:
        a = tot["attention"]
        out["attention_kernel"] = dict(achieved=a[0] / a[1] / 1e12, frac=a[0] / a[1] / MFMA_PEAK, launches=a[2],
                                       seconds_per_eval=a[1])
    return out


def cpu_baseline(batch_unused):
    """Oracle ("port": CPU fp32 restatement of the reference) on a bounded sample of the same workload: one 512x512
    image — SwinIR, VAE encode, ONE CFG sampler step (2 network evals), VAE decode — extrapolated to 50 steps
    (per-step cost is constant)."""
    import torch
    from oracle import cases, nets
    cldm_cfg, swin_cfg = cases.get_cfgs("full")
    W = cases.synth_weights(cldm_cfg, swin_cfg, 0)
    # the GPU box has hundreds of host cores; torch's CPU kernels stop scaling (and then regress) far below that, so
    # the port is timed on at most 32 threads and `cores` reports the threads actually used
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    x = torch.tensor(cases.make_lq(3, 1, 512, 512)).float().div(255).permute(0, 3, 1, 2).contiguous()
    with torch.no_grad():
        t0 = time.time()
        clean = nets.swinir_forward(W["swinir"], swin_cfg, x)
        t_swin = time.time() - t0
        t0 = time.time()
        c_img = nets.vae_encode_mode(W["vae"], cldm_cfg["vae_cfg"], clean * 2 - 1, 0.18215)
        t_enc = time.time() - t0
        c_txt = torch.randn(1, 77, 1024)
        xt = torch.randn(1, 4, 64, 64)
        t0 = time.time()
        for _ in range(2):
            nets.cldm_forward(W, cldm_cfg, xt, torch.tensor([500]), c_txt, c_img, [1.0] * 13)
        t_step = time.time() - t0
        t0 = time.time()
        nets.vae_decode(W["vae"], cldm_cfg["vae_cfg"], xt, 0.18215)
        t_dec = time.time() - t0
    t_img = t_swin + t_enc + 50 * t_step + t_dec
    return dict(value=1.0 / t_img, unit="images/sec", cores=cores, kind="port",
                sample=f"1x512x512: SwinIR {t_swin:.1f}s + VAE-enc {t_enc:.1f}s + 1 of 50 CFG steps {t_step:.1f}s "
                       f"(x50 extrapolated) + VAE-dec {t_dec:.1f}s => {t_img:.0f}s/image")


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.selftest:
        device, sync = torch.device("cpu"), (lambda: None)
        if world > 1:
            dist.init_process_group("gloo")
        dtype = torch.float16
        pipe = cldm = swin = None
        lq_dev = None

        def run_step():
            time.sleep(0.02 * (1 + rank))  # ranks differ: the reported time must be the slowest rank's
            return np.zeros((args.batch, 512, 512, 3), dtype=np.uint8)
    else:
        assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
        torch.cuda.set_device(local)
        device, sync = torch.device("cuda", local), torch.cuda.synchronize
        if world > 1:
            dist.init_process_group("nccl", device_id=device)
        from diffbir_amd import native
        native.lib()
        dtype = torch.float16 if args.dtype == "fp16" else torch.bfloat16
        pipe, cldm, swin = build_engine(device, dtype)
        rs = np.random.RandomState(100 + rank)
        lq = rs.randint(0, 256, (args.batch, 512, 512, 3)).astype(np.uint8)
        lq_dev = torch.as_tensor(lq).to(device)          # inputs resident in HBM before the timed region
        torch.manual_seed(231 + rank)

        def run_step():
            return run_once(pipe, lq_dev, args.sampler_steps)

    def barrier():
        sync()
        if world > 1:
            dist.barrier()
        sync()

    for _ in range(args.warmup):
        run_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = run_step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert out.shape == (args.batch, 512, 512, 3) and out.dtype == np.uint8
    images = args.batch * args.steps * world
    value = images / dt
    res = {
        "metric": "restored images/sec @512x512, 50-step SpacedSampler+CFG", "value": value, "unit": "images/sec",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16" if dtype == torch.float16 else "bf16",
        "data": "synthetic (seeded uint8 noise images, random-init weights of the real architecture)",
        "config": {"workload": f"BSR pipeline, batch {args.batch}x512x512 per GPU, {args.sampler_steps}-step "
                               "SpacedSampler + CFG=4.0, SwinIR+ControlLDM (SD-2.1 v-pred), fp16, data-parallel replicas",
                   "global_batch": args.batch * world, "sampler_steps": args.sampler_steps, "parallelism": f"dp{world}"},
        "mfma_frac_end_to_end": value * FLOPS_PER_IMAGE * (args.sampler_steps / 50.0) / (world * MFMA_PEAK),
    }
    if args.selftest:
        res["data"] = "SELFTEST (fake workload, CPU/gloo) - not a measurement"
    if rank == 0 and not args.no_roofline and not args.selftest:
        res["roofline"] = measure_roofline(cldm, device, args.batch)
    if world > 1:
        dist.barrier()
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.selftest:
        del pipe, cldm, swin
        torch.cuda.empty_cache()
        res["cpu_baseline"] = cpu_baseline(args.batch)
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
