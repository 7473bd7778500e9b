"""dbir_plan (include/dbir.h "Module-level entry point") without a GPU: the generated dispatch table is the one the binding's
signature table implies, the recorder's encoding round-trips through `dbir_plan_create`'s validation, and malformed plans are
rejected with a message.  (Replaying needs a GPU: tests/test_pipeline_gpu.py::test_plan_replay_*.)"""
import ctypes
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_committed_dispatch_table_is_the_generated_one():
    import gen_plan_dispatch as g
    with open(g.OUT) as f:
        assert f.read() == g.generate(), "diffbir_amd/csrc/plan_dispatch.inc is stale: python tools/gen_plan_dispatch.py"
    names = g.recordable()
    assert "dbir_gemm" in names and "dbir_xf_tail" in names and "dbir_attention" in names and "dbir_copy_rows" in names
    assert "dbir_cldm_forward" not in names and "dbir_plan_run" not in names


def _lib():
    from diffbir_amd import native
    return native, native.lib()


def test_fn_index_matches_the_generator():
    import gen_plan_dispatch as g
    native, lib = _lib()
    for i, n in enumerate(g.recordable()):
        assert lib.dbir_plan_fn_index(n.encode()) == i
    assert lib.dbir_plan_fn_index(b"dbir_abi_version") == -1 and lib.dbir_plan_fn_index(b"nope") == -1


class _FakeStream:
    def __init__(self, h):
        self.cuda_stream = h


def test_recorder_encoding_and_plan_validation():
    from diffbir_amd import plan
    native, lib = _lib()
    rec = plan.Recorder(_FakeStream(7))
    # dbir_add_scaled(dtype, a, lda, b, ldb, s, out, ldo, M, C, stream)
    rec.call("dbir_add_scaled", (0, 4096, 320, 8192, 320, 0.5, 12288, 320, 100, 320, 7))
    d = native.GemmDesc()
    d.M, d.N, d.K = 64, 128, 64
    rec.call("dbir_gemm", (ctypes.byref(d), 7))
    op = rec.ops[0]
    assert op.fn == lib.dbir_plan_fn_index(b"dbir_add_scaled") and op.nargs == 10 and op.stream == 0
    assert op.a[1].p == 4096 and op.a[2].i == 320 and abs(op.a[5].f - 0.5) < 1e-12 and op.a[9].i == 320
    g = rec.ops[1]
    assert g.nargs == 1 and g.a[0].i % 8 == 0 and len(rec.blob) >= g.a[0].i + ctypes.sizeof(native.GemmDesc)
    back = native.GemmDesc.from_buffer_copy(bytes(rec.blob[g.a[0].i:g.a[0].i + ctypes.sizeof(native.GemmDesc)]))
    assert (back.M, back.N, back.K) == (64, 128, 64)
    p = rec.build()                       # one stream, no events: no HIP call is needed to create it
    assert p.n_ops == 2 and p.calls == 2 and lib.dbir_plan_num_ops(p.handle) == 2
    p.close()

    def create(ops, blob=b"", n_streams=1, n_events=0):
        arr = (plan.PlanOp * len(ops))(*ops)
        h = ctypes.c_void_p()
        rc = lib.dbir_plan_create(ctypes.byref(h), arr, len(ops), blob, len(blob), n_streams, n_events)
        if rc == 0:
            lib.dbir_plan_destroy(h)
        return rc, lib.dbir_last_error().decode()

    bad = plan.PlanOp()
    bad.fn, bad.nargs = 9999, 1
    rc, msg = create([bad])
    assert rc != 0 and "function" in msg
    short = plan.PlanOp()
    short.fn, short.nargs = op.fn, 3                       # wrong argument count for dbir_add_scaled
    rc, msg = create([short])
    assert rc != 0 and "arguments" in msg
    wait = plan.PlanOp()
    wait.fn, wait.nargs = plan.STREAM_WAIT, 1
    wait.a[0].i = 0
    rc, msg = create([wait], n_events=0)
    assert rc != 0 and "event" in msg
    gd = plan.PlanOp()
    gd.fn, gd.nargs = lib.dbir_plan_fn_index(b"dbir_gemm"), 1
    gd.a[0].i = 4096                                        # descriptor offset outside the (empty) blob
    rc, msg = create([gd])
    assert rc != 0 and "blob" in msg
    far = plan.PlanOp()
    far.fn, far.nargs, far.stream = op.fn, 10, 3            # stream slot outside the plan
    rc, msg = create([far])
    assert rc != 0 and "stream slot" in msg


def test_wait_on_foreign_event_is_an_error():
    from diffbir_amd import plan
    rec = plan.Recorder(_FakeStream(0))
    with pytest.raises(RuntimeError):
        rec.stream_wait(_FakeStream(0), object())
