"""Recorder and handle for `dbir_plan` (include/dbir.h "Module-level entry point", csrc/plan.hip): one network evaluation —
ControlNet + UNet, ~600 operator calls on two streams — recorded once while it runs eagerly, then replayed from C with ONE
host call per evaluation (`dbir_cldm_forward`).

How a recording works
  * `Recorder` swaps the library handle `native.lib()` returns for a proxy: every operator call is executed as usual AND
    appended to the op list (function index from `dbir_plan_fn_index`, arguments encoded by the binding's own signature table,
    the stream it was issued on as a slot number; `dbir_gemm`'s descriptor is copied by value into the plan's blob).
  * stream ordering: the engine orders its two streams with `record_event` / `wait_event` / `wait_stream` below instead of
    calling torch directly; they do the torch call and, while recording, append the matching plan op.
  * memory: the evaluation's activations are ordinary PyTorch allocations made inside a private `torch.cuda.MemPool`, kept
    alive with the plan — the recorded pointers stay valid and nobody else allocates from it.  PyTorch's caching allocator
    reuses a freed block only on the stream it was allocated on (stream order, which the replay preserves) or, for blocks
    handed across streams with `record_stream`, after the forward has returned: no reuse inside one evaluation can race in a
    replay that issues the same calls in the same per-stream order with the same event waits.
  * what must NOT be recorded: first-use autotuning (re-runs launches), the time-embedding cache (a replay must compute the
    rows from the static `t`) — both check `recording()`; the caller runs one eager warm-up evaluation first.
The C side validates the list (argument counts per function, event waits after their records, descriptor offsets).
"""
import ctypes
from ctypes import Structure, Union, c_double, c_int, c_longlong, c_void_p
from typing import Dict, List, Optional

import torch

from . import native

MAX_ARGS = 24
EVENT_RECORD, STREAM_WAIT = -1, -2


class Arg(Union):
    _fields_ = [("i", c_longlong), ("f", c_double), ("p", c_void_p)]


class PlanOp(Structure):
    """Mirror of `dbir_plan_op`."""
    _fields_ = [("fn", c_int), ("stream", c_int), ("nargs", c_int), ("reserved", c_int), ("a", Arg * MAX_ARGS)]


_REC: Optional["Recorder"] = None


def recording() -> bool:
    return _REC is not None


# ---- stream ordering helpers (the engine calls these instead of torch's, so that a recording sees them) -----------------
def record_event(stream: "torch.cuda.Stream") -> "torch.cuda.Event":
    ev = torch.cuda.Event()
    ev.record(stream)
    if _REC is not None:
        _REC.event_record(ev, stream)
    return ev


def wait_event(stream: "torch.cuda.Stream", ev: "torch.cuda.Event") -> None:
    stream.wait_event(ev)
    if _REC is not None:
        _REC.stream_wait(stream, ev)


def wait_stream(stream: "torch.cuda.Stream", other: "torch.cuda.Stream") -> None:
    """`stream` waits for everything issued on `other` so far."""
    if _REC is None:
        stream.wait_stream(other)
    else:
        wait_event(stream, record_event(other))


class _RecordingLib:
    """Proxy over libdbir_hip.so: executes every call and hands the recordable ones to the recorder."""

    def __init__(self, real, rec: "Recorder"):
        object.__setattr__(self, "_l", real)
        object.__setattr__(self, "_rec", rec)

    def __getattr__(self, name):
        fn = getattr(self._l, name)
        if not callable(fn) or name not in self._rec.fn_index:
            return fn
        rec = self._rec

        def recorded(*a):
            rc = fn(*a)
            if rc == 0:
                rec.call(name, a)
            return rc

        object.__setattr__(self, name, recorded)
        return recorded


class Recorder:
    """with Recorder(main_stream) as rec: <run the evaluation eagerly> ; plan = rec.build()"""

    def __init__(self, main: "torch.cuda.Stream"):
        real = native.lib()
        assert not isinstance(real, _RecordingLib), "nested recordings are not supported"
        self._real = real
        self.fn_index: Dict[str, int] = {}
        for name, sig in native.SIGNATURES.items():
            idx = real.dbir_plan_fn_index(name.encode())
            if idx >= 0:
                self.fn_index[name] = idx
        self.gemm = self.fn_index["dbir_gemm"]
        self.ops: List[PlanOp] = []
        self.blob = bytearray()
        self.slots: Dict[int, int] = {int(main.cuda_stream): 0}
        self.events: Dict[int, int] = {}
        self._keep = []   # torch events stay alive until the recording ends (ids are used as keys)

    # ---- context manager -----------------------------------------------------------------------------------------------
    def __enter__(self):
        global _REC
        _REC = self
        native._lib = _RecordingLib(self._real, self)
        return self

    def __exit__(self, *exc):
        global _REC
        native._lib = self._real
        _REC = None
        return False

    # ---- recording -----------------------------------------------------------------------------------------------------
    def slot(self, handle) -> int:
        h = int(handle or 0)
        if h not in self.slots:
            self.slots[h] = len(self.slots)
        return self.slots[h]

    def call(self, name: str, args) -> None:
        sig = native.SIGNATURES[name]
        assert len(args) == len(sig) and len(sig) - 1 <= MAX_ARGS, (name, len(args))
        op = PlanOp()
        op.fn, op.nargs = self.fn_index[name], len(sig) - 1
        st = args[-1]
        op.stream = self.slot(st.value if isinstance(st, c_void_p) else st)
        for k, (ty, v) in enumerate(zip(sig[:-1], args[:-1])):
            if op.fn == self.gemm and k == 0:   # ctypes.byref(GemmDesc) -> the descriptor by value in the blob
                d = v._obj
                while len(self.blob) % 8:
                    self.blob.append(0)
                op.a[0].i = len(self.blob)
                self.blob += ctypes.string_at(ctypes.addressof(d), ctypes.sizeof(d))
            elif ty is native._F:
                op.a[k].f = float(v)
            elif ty is native._P:
                op.a[k].p = v.value if isinstance(v, c_void_p) else (int(v) if v is not None else None)
            else:
                op.a[k].i = int(v)
        self.ops.append(op)

    def _control(self, fn: int, stream, ev) -> None:
        op = PlanOp()
        op.fn, op.nargs, op.stream = fn, 1, self.slot(stream.cuda_stream)
        op.a[0].i = self.events[id(ev)]
        self.ops.append(op)

    def event_record(self, ev, stream) -> None:
        self.events[id(ev)] = len(self.events)
        self._keep.append(ev)
        self._control(EVENT_RECORD, stream, ev)

    def stream_wait(self, stream, ev) -> None:
        if id(ev) not in self.events:
            raise RuntimeError("plan recording: wait on an event that was not recorded through diffbir_amd.plan.record_event")
        self._control(STREAM_WAIT, stream, ev)

    # ---- result --------------------------------------------------------------------------------------------------------
    def build(self) -> "Plan":
        return Plan(self)


class Plan:
    """Owner of a native `dbir_plan*`."""

    def __init__(self, rec: Recorder):
        n = len(rec.ops)
        if n == 0:
            raise RuntimeError("empty recording")
        arr = (PlanOp * n)(*rec.ops)
        blob = bytes(rec.blob)
        h = c_void_p()
        native.check(native.lib().dbir_plan_create(ctypes.byref(h), arr, n, blob, len(blob), len(rec.slots), len(rec.events)),
                     "dbir_plan_create")
        self.handle = h
        self.n_ops, self.n_streams, self.n_events = n, len(rec.slots), len(rec.events)
        self.calls = sum(1 for o in rec.ops if o.fn >= 0)

    def bind(self, slot: int, t: torch.Tensor) -> None:
        assert t.is_contiguous()
        native.check(native.lib().dbir_plan_bind(self.handle, slot, t.data_ptr(), t.numel() * t.element_size()), "dbir_plan_bind")

    def run(self, stream: Optional[int] = None) -> None:
        s = torch.cuda.current_stream().cuda_stream if stream is None else stream
        native.check(native.lib().dbir_plan_run(self.handle, s), "dbir_plan_run")

    def cldm_forward(self, x: torch.Tensor, t: torch.Tensor, c_img: torch.Tensor, eps: torch.Tensor) -> torch.Tensor:
        for v in (x, t, c_img, eps):
            assert v.is_cuda and v.dtype == torch.float32 and v.is_contiguous()
        native.check(native.lib().dbir_cldm_forward(self.handle, x.data_ptr(), t.data_ptr(), c_img.data_ptr(), eps.data_ptr(),
                                                    torch.cuda.current_stream().cuda_stream), "dbir_cldm_forward")
        return eps

    def close(self) -> None:
        if self.handle:
            native.lib().dbir_plan_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass
