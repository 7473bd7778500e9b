"""-m gpu, needs >= 2 GPUs (skips on the 1-GPU box): bench.py exactly as the driver launches it for N = 2 — one rank per GPU
over RCCL — exercising what the N = 1 run cannot: the bucketed RCCL weight broadcast (`parallel.broadcast_state_dict`), batch
sharding with the gather of the restored uint8 batches to rank 0 (`parallel.gather_batch`), and tile sharding with one RCCL
all-reduce per network evaluation (`parallel.enable_tile_sharding`).  The same plumbing runs on CPU / gloo in
tests/test_bench_plumbing_cpu.py and tests/test_parallel_cpu.py."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
           "--no-roofline"] + extra
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs")
def test_two_ranks_batch_sharded_over_rccl():
    r = _run(["--sampler-steps", "2", "--batch", "2"])
    assert r["n_gpus"] == 2 and r["config"]["global_batch"] == 4 and r["config"]["parallelism"] == "dp2"
    assert r["gathered_batch"] == [4, 512, 512, 3] and "broadcast_state_dict" in r["weights"]
    assert r["rccl"]["ranks"] == 2 and r["rccl"]["backend"] == "nccl"


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs")
def test_two_ranks_tile_sharded_over_rccl():
    r = _run(["--config", "c4", "--sampler-steps", "2"])
    assert r["n_gpus"] == 2 and r["config"]["parallelism"] == "tile-shard2" and r["scaling"] == "strong"


def _parity(n, extra, out):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    base = ["bench.py", "--gpus", str(n), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-roofline", "--parity-out", out]
    if n > 1:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
               "127.0.0.1", "--master-port", str(_port())] + base + extra
    else:
        cmd = [sys.executable] + base + extra
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, p.stderr[-3000:]
    import numpy as np
    return np.load(out)


def _psnr(a, b):
    import numpy as np
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return 999.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs")
def test_gpu_count_parity_over_rccl(tmp_path):
    """The restored images must not depend on the number of GPUs beyond rounding: batch sharding draws the full-batch noise
    from one seed and every rank keeps its rows, so the only difference between 1 and 2 ranks is that a rank's batch (2
    instead of 4) selects other per-shape tiles / epilogue-statistics launches, i.e. other f32 summation orders — NOT
    bit-identical since round 3 (tests/test_pipeline_gpu.py::test_data_parallel_world2_on_one_device is the 1-GPU proxy
    of this comparison and carries the same bar); tile sharding changes the summation order of the tile blend (one RCCL
    all-reduce per evaluation).  Bars: >= 50 dB / >= 55 dB between GPU counts."""
    a = _parity(1, ["--batch", "4", "--sampler-steps", "3"], str(tmp_path / "dp1.npy"))
    b = _parity(2, ["--batch", "4", "--sampler-steps", "3"], str(tmp_path / "dp2.npy"))
    assert a.shape == (4, 512, 512, 3) and a.shape == b.shape
    assert _psnr(a, b) >= 50.0, f"batch-sharded output differs between 1 and 2 GPUs: {_psnr(a, b):.1f} dB"
    a = _parity(1, ["--config", "c4", "--sampler-steps", "2"], str(tmp_path / "t1.npy"))
    b = _parity(2, ["--config", "c4", "--sampler-steps", "2"], str(tmp_path / "t2.npy"))
    assert a.shape == b.shape and _psnr(a, b) >= 55.0


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs")
def test_gpu_count_parity_strict_without_tuned_tiles(tmp_path, monkeypatch):
    """the 50 dB bar above tolerates what tuned per-batch tiles do; a real sharding bug (wrong noise rows)
    could hide under it.  With the tile table and the first-use autotune OFF the C heuristic picks tiles from the problem
    class only where M-independent, and with the epilogue statistics off every GroupNorm reads the stored tensor: the per
    sample arithmetic is then the same for a rank's batch of 2 and the full batch of 4.  EXPECTED: bit-identical outputs.
    ASSERTED: >= 60 dB with the per-image identity list in the message — this test has never run (no 2-GPU box was
    available to any round), so the bar stays at what a correct sharding cannot miss; tighten it to `all(same)` once a
    run has shown the expectation to hold."""
    import numpy as np
    monkeypatch.setenv("DBIR_AUTOTUNE", "0")
    monkeypatch.setenv("DBIR_TUNING", "0")
    monkeypatch.setenv("DBIR_GN_EPILOGUE_STATS", "0")
    monkeypatch.setenv("DBIR_FUSED_XF", "0")
    a = _parity(1, ["--batch", "4", "--sampler-steps", "2"], str(tmp_path / "s1.npy"))
    b = _parity(2, ["--batch", "4", "--sampler-steps", "2"], str(tmp_path / "s2.npy"))
    same = [bool(np.array_equal(a[i], b[i])) for i in range(4)]
    assert a.shape == b.shape and _psnr(a, b) >= 60.0, f"strict GPU-count parity: {_psnr(a, b):.1f} dB, identical images {same}"


def _forced_one_rank(extra, timeout=1200):
    """bench.py under torch.distributed.run --nproc-per-node 1 with --force-collectives: the RCCL code path of an N-GPU run
    on the one GPU of this box."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), "bench.py", "--gpus", "1", "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
           "--no-roofline", "--force-collectives"] + extra
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_rccl_executes_on_one_gpu_batch_sharded():
    """RCCL had never executed on hardware (the 2-GPU tests above skip on the 1-GPU box).  One rank,
    collectives forced: ncclCommInitRank, the bucketed weight broadcast (parallel.broadcast_state_dict), the barrier /
    max-over-ranks all-reduce of the timing and the gather of the restored uint8 batch (parallel.gather_batch) all run on a
    one-rank RCCL communicator, inside the bench job exactly as the driver launches it."""
    r = _forced_one_rank(["--sampler-steps", "2", "--batch", "2"])
    assert r["n_gpus"] == 1 and r["config"]["parallelism"] == "dp1" and r["gathered_batch"] == [2, 512, 512, 3]
    assert r["rccl"]["ranks"] == 1 and r["rccl"]["backend"] == "nccl" and r["rccl"]["forced_single_rank"] is True
    assert r["rccl"]["calls"]["broadcast"] >= 5 and r["rccl"]["calls"]["gather"] == 1, r["rccl"]
    assert "broadcast_state_dict" in r["weights"]


def test_rccl_one_rank_is_bit_identical_to_the_plain_run(tmp_path, monkeypatch):
    """The same restoration with and without the collectives (one rank): weights through the RCCL broadcast, tiled sampling
    through the sharded path (partial weighted sum -> all-reduce -> normalise, one all-reduce per evaluation), outputs through
    the gather.  A one-rank reduction is the identity and the sharded path performs the plain path's f32 operations in the
    same order, so the uint8 results must be IDENTICAL — stream-ordering or staging errors around the collectives would show."""
    import numpy as np
    monkeypatch.setenv("DBIR_AUTOTUNE", "0")   # timing-based tile choices may differ between two processes (other f32 orders)
    a = _parity(1, ["--config", "c4", "--sampler-steps", "2"], str(tmp_path / "plain.npy"))
    out = str(tmp_path / "forced.npy")
    r = _forced_one_rank(["--config", "c4", "--sampler-steps", "2", "--parity-out", out])
    b = np.load(out)
    assert r["rccl"]["calls"]["all_reduce"] >= 2 and r["rccl"]["calls"]["broadcast"] >= 5, r["rccl"]
    assert a.shape == b.shape == (1, 2048, 2048, 3)
    assert np.array_equal(a, b), f"forced-collectives run differs from the plain run: {_psnr(a, b):.1f} dB"


def test_gpus_flag_must_match_world_size():
    """`--gpus N` with a different WORLD_SIZE is an error, not a silent 1-GPU benchmark."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "4", "--selftest"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)
