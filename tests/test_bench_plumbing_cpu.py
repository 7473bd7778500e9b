"""bench.py's multi-process plumbing exactly as the driver launches it (torch.distributed.run, one rank per GPU) — on
CPU with gloo and a fake workload (`--selftest`): rendezvous on 127.0.0.1, barriers, max-over-ranks timing, ONE JSON line
from rank 0 carrying the contract fields."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config"}


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(cmd):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_single_process_contract():
    r = _run([sys.executable, "bench.py", "--selftest", "--steps", "3", "--warmup", "1"])
    assert REQUIRED <= set(r) and r["n_gpus"] == 1 and r["steps"] == 3 and r["warmup"] == 1
    assert r["higher_is_better"] is True and r["scaling"] == "weak" and "workload" in r["config"]
    assert abs(r["value"] - 8 * 3 / (r["ms_per_step"] * 3e-3)) < 1e-6 * r["value"] + 1e-9


def test_two_ranks_as_launched_by_the_driver():
    r = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
              "127.0.0.1", "--master-port", str(_port()), "bench.py", "--gpus", "2", "--steps", "4", "--warmup", "1",
              "--selftest"])
    assert REQUIRED <= set(r) and r["n_gpus"] == 2 and r["config"]["global_batch"] == 16
    # rank 1 sleeps 40 ms per step, rank 0 20 ms: the reported step time is the slowest rank's
    assert r["ms_per_step"] >= 39.0, r["ms_per_step"]
    assert abs(r["value"] - 8 * 4 * 2 / (r["ms_per_step"] * 4e-3)) < 1e-6 * r["value"]


def test_gpus_flag_self_spawns_and_checks_world_size():
    """`python bench.py --gpus 2` without a launcher re-launches itself under torch.distributed.run (one rank per GPU);
    the two-rank selftest also runs the bucketed weight broadcast and the gather of the per-rank batches to rank 0."""
    r = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--selftest"])
    assert r["n_gpus"] == 2 and r.get("broadcast_checked") is True and r["gathered_batch"][0] == 16
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", OMP_NUM_THREADS="1")
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "4", "--selftest"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)


def test_forced_collectives_on_one_rank():
    """`--force-collectives` (the mode tests/test_multigpu_gpu.py runs over RCCL on the 1-GPU box): a ONE-rank process group is
    created and the broadcast / barrier / max-over-ranks all-reduce / gather all execute on it."""
    r = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
              "127.0.0.1", "--master-port", str(_port()), "bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1",
              "--selftest", "--force-collectives"])
    assert REQUIRED <= set(r) and r["n_gpus"] == 1 and r.get("broadcast_checked") is True and r["gathered_batch"][0] == 8
