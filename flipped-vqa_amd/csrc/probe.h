// Measurement probe behind fvqa_gemm_timing_* (include/fvqa.h): HIP events around each projection-GEMM launch on ITS
// launch stream. One probe object exists between enable(1) and enable(0); launches from ANY host thread record into
// it under a mutex (PyTorch runs the backward of the step on its own autograd thread). With no probe enabled a launch
// reads one atomic pointer and touches nothing else.
#pragma once
#include "common.h"
#include <atomic>
#include <mutex>
#include <vector>

// stride N > 1: only every N-th launch is bracketed by events (the others are recorded with their FLOPs and kind alone), so
// that the measured stream keeps the duty cycle of an un-instrumented one: an event pair after EVERY launch leaves the
// chip ~5 us idle 258 times per step, and a power-limited chip answers lighter duty with a higher clock.
struct FvqaProbeRec { hipEvent_t e0, e1; double flops; int kind; bool timed; };
struct FvqaProbe { std::mutex mu; std::vector<FvqaProbeRec> recs; int stride = 1; std::atomic<unsigned> count{0}; };
FvqaProbe* fvqa_probe_current();            // the enabled probe or nullptr

struct FvqaProbeScope {
  hipStream_t st; FvqaProbe* p; FvqaProbeRec r;
  FvqaProbeScope(hipStream_t s, double flops, int kind) : st(s), p(fvqa_probe_current()) {
    if (!p) return;
    r.flops = flops; r.kind = kind; r.e0 = r.e1 = nullptr;
    r.timed = p->count.fetch_add(1) % (unsigned)p->stride == 0;
    if (!r.timed) return;
    if (hipEventCreate(&r.e0) != hipSuccess) { p = nullptr; return; }
    if (hipEventCreate(&r.e1) != hipSuccess) { (void)hipEventDestroy(r.e0); p = nullptr; return; }
    (void)hipEventRecord(r.e0, st);
  }
  ~FvqaProbeScope() {
    if (!p) return;
    if (r.timed) (void)hipEventRecord(r.e1, st);
    std::lock_guard<std::mutex> g(p->mu);
    p->recs.push_back(r);
  }
};
