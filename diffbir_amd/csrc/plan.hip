// dbir_plan — a recorded network evaluation replayed from C (include/dbir.h "Module-level entry point").
//
// The host side (diffbir_amd/plan.py) records the ~600 operator calls of one ControlLDM evaluation while it runs them eagerly:
// function index, arguments, stream slot, plus the event record / wait pairs that order the ControlNet's stream against the
// UNet's.  dbir_plan_run walks that list: same kernels, same operands, same per-stream order — one host call per evaluation.
// Nothing here launches a kernel of its own; the operator entry points do (plan_dispatch.inc is generated from the binding's
// signature table by tools/gen_plan_dispatch.py).
#include <stdlib.h>
#include <time.h>

#include <vector>

#include "common.h"
#include "plan_dispatch.inc"

struct dbir_plan {
  std::vector<dbir_plan_op> ops;
  std::vector<char> blob;
  std::vector<hipStream_t> streams;   // slot 0 unused (the caller's stream); slots >= 1 owned
  std::vector<hipEvent_t> events;
  struct {
    void* ptr;
    long long bytes;
  } bind[4];
};

extern "C" int dbir_plan_fn_index(const char* name) {
  if (!name) return -1;
  for (int i = 0; i < kPlanFnCount; ++i)
    if (strcmp(name, kPlanFnNames[i]) == 0) return i;
  return -1;
}

extern "C" int dbir_plan_create(dbir_plan** out, const dbir_plan_op* ops, int n_ops, const void* blob, long long blob_bytes,
                                int n_streams, int n_events) {
  DBIR_CHECK_ARG(out && ops && n_ops > 0 && n_streams >= 1 && n_streams <= 8 && n_events >= 0 && n_events <= 4096 &&
                     blob_bytes >= 0 && (blob || blob_bytes == 0), "dbir_plan_create: bad arguments");
  for (int i = 0; i < n_ops; ++i) {
    const dbir_plan_op& o = ops[i];
    DBIR_CHECK_ARG(o.stream >= 0 && o.stream < n_streams, "dbir_plan_create: op %d: stream slot %d of %d", i, o.stream, n_streams);
    if (o.fn == DBIR_PLAN_EVENT_RECORD || o.fn == DBIR_PLAN_STREAM_WAIT) {
      DBIR_CHECK_ARG(o.nargs == 1 && o.a[0].i >= 0 && o.a[0].i < n_events, "dbir_plan_create: op %d: bad event %lld", i, o.a[0].i);
    } else {
      DBIR_CHECK_ARG(o.fn >= 0 && o.fn < kPlanFnCount && o.nargs == kPlanFnArgs[o.fn],
                     "dbir_plan_create: op %d: function %d with %d arguments", i, o.fn, o.nargs);
      if (o.fn == dbir_plan_fn_index("dbir_gemm"))
        DBIR_CHECK_ARG(o.a[0].i >= 0 && o.a[0].i + (long long)sizeof(dbir_gemm_desc) <= blob_bytes,
                       "dbir_plan_create: op %d: descriptor outside the blob", i);
    }
  }
  // every wait must follow a record of its event (replay order = list order on the host; the device orders the rest)
  {
    std::vector<char> seen(n_events, 0);
    for (int i = 0; i < n_ops; ++i) {
      if (ops[i].fn == DBIR_PLAN_EVENT_RECORD) seen[ops[i].a[0].i] = 1;
      if (ops[i].fn == DBIR_PLAN_STREAM_WAIT)
        DBIR_CHECK_ARG(seen[ops[i].a[0].i], "dbir_plan_create: op %d waits for event %lld before it is recorded", i, ops[i].a[0].i);
    }
  }
  dbir_plan* p = new dbir_plan();
  p->ops.assign(ops, ops + n_ops);
  p->blob.assign(reinterpret_cast<const char*>(blob), reinterpret_cast<const char*>(blob) + blob_bytes);
  p->streams.assign(n_streams, nullptr);
  p->events.assign(n_events, nullptr);
  memset(p->bind, 0, sizeof(p->bind));
  bool ok = true;
  for (int s = 1; s < n_streams && ok; ++s) ok = hipStreamCreateWithFlags(&p->streams[s], hipStreamNonBlocking) == hipSuccess;
  for (int e = 0; e < n_events && ok; ++e) ok = hipEventCreateWithFlags(&p->events[e], hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    dbir_plan_destroy(p);
    dbir_set_error("dbir_plan_create: could not create the plan's streams / events: %s", hipGetErrorString(hipGetLastError()));
    return DBIR_ERR_LAUNCH;
  }
  *out = p;
  return DBIR_OK;
}

extern "C" int dbir_plan_destroy(dbir_plan* p) {
  if (!p) return DBIR_OK;
  for (size_t s = 1; s < p->streams.size(); ++s)
    if (p->streams[s]) {
      (void)hipStreamSynchronize(p->streams[s]);
      (void)hipStreamDestroy(p->streams[s]);
    }
  for (hipEvent_t e : p->events)
    if (e) (void)hipEventDestroy(e);
  delete p;
  return DBIR_OK;
}

extern "C" int dbir_plan_num_ops(const dbir_plan* p) { return p ? (int)p->ops.size() : 0; }

extern "C" int dbir_plan_bind(dbir_plan* p, int slot, void* device_ptr, long long bytes) {
  DBIR_CHECK_ARG(p && slot >= 0 && slot < 4 && device_ptr && bytes > 0, "dbir_plan_bind: bad arguments");
  p->bind[slot].ptr = device_ptr;
  p->bind[slot].bytes = bytes;
  return DBIR_OK;
}

// DIAGNOSTIC (env DBIR_PLAN_PACE_NS, default 0): busy-wait this many nanoseconds between two recorded calls — emulates the
// ~15 - 30 us a Python / ctypes launch takes, to separate "what is launched" from "how fast it is enqueued" when a replay is
// compared with eager launches (profiles/r5_graph_plan_ab.txt)
static long long plan_pace_ns() {
  static const long long v = getenv("DBIR_PLAN_PACE_NS") ? atoll(getenv("DBIR_PLAN_PACE_NS")) : 0;
  return v;
}

extern "C" int dbir_plan_run(dbir_plan* p, void* stream) {
  DBIR_CHECK_ARG(p, "dbir_plan_run: null plan");
  hipStream_t main = reinterpret_cast<hipStream_t>(stream);
  const char* blob = p->blob.data();
  const size_t n = p->ops.size();
  const long long pace = plan_pace_ns();
  for (size_t i = 0; i < n; ++i) {
    if (pace > 0) {
      timespec t0, t1;
      clock_gettime(CLOCK_MONOTONIC, &t0);
      do {
        clock_gettime(CLOCK_MONOTONIC, &t1);
      } while ((t1.tv_sec - t0.tv_sec) * 1000000000LL + (t1.tv_nsec - t0.tv_nsec) < pace);
    }
    const dbir_plan_op& o = p->ops[i];
    hipStream_t s = o.stream == 0 ? main : p->streams[o.stream];
    if (o.fn == DBIR_PLAN_EVENT_RECORD) {
      if (hipEventRecord(p->events[o.a[0].i], s) != hipSuccess) {
        dbir_set_error("dbir_plan_run: op %zu: hipEventRecord failed", i);
        return DBIR_ERR_LAUNCH;
      }
    } else if (o.fn == DBIR_PLAN_STREAM_WAIT) {
      if (hipStreamWaitEvent(s, p->events[o.a[0].i], 0) != hipSuccess) {
        dbir_set_error("dbir_plan_run: op %zu: hipStreamWaitEvent failed", i);
        return DBIR_ERR_LAUNCH;
      }
    } else {
      const int rc = plan_dispatch(o.fn, o.a, blob, s);
      if (rc != DBIR_OK) return rc;   // (dbir_last_error holds the operator's message)
    }
  }
  return DBIR_OK;
}

extern "C" int dbir_cldm_forward(dbir_plan* p, const float* x, const float* t, const float* c_img, float* eps, void* stream) {
  DBIR_CHECK_ARG(p && x && t && c_img && eps, "dbir_cldm_forward: null pointer");
  for (int k = 0; k < 4; ++k) DBIR_CHECK_ARG(p->bind[k].ptr, "dbir_cldm_forward: buffer %d of the plan is not bound (dbir_plan_bind)", k);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const void* src[3] = {x, t, c_img};
  for (int k = 0; k < 3; ++k)
    if (src[k] != p->bind[k].ptr &&
        hipMemcpyAsync(p->bind[k].ptr, src[k], (size_t)p->bind[k].bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) {
      dbir_set_error("dbir_cldm_forward: copying input %d failed", k);
      return DBIR_ERR_LAUNCH;
    }
  const int rc = dbir_plan_run(p, stream);
  if (rc != DBIR_OK) return rc;
  if (eps != p->bind[3].ptr &&
      hipMemcpyAsync(eps, p->bind[3].ptr, (size_t)p->bind[3].bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) {
    dbir_set_error("dbir_cldm_forward: copying the output failed");
    return DBIR_ERR_LAUNCH;
  }
  return DBIR_OK;
}
