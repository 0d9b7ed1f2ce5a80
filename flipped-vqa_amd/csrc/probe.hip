// fvqa_gemm_timing_* (include/fvqa.h): see probe.h.
#include "probe.h"
#include <atomic>

namespace {
std::atomic<FvqaProbe*> g_probe{nullptr};

void drop(FvqaProbe* p) {
  if (!p) return;
  for (auto& r : p->recs)
    if (r.timed) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  delete p;
}
}  // namespace

FvqaProbe* fvqa_probe_current() { return g_probe.load(std::memory_order_acquire); }

// (the caller makes sure no launch is in flight on another host thread while it switches the probe)
extern "C" int fvqa_gemm_timing_enable(int on) {
  FvqaProbe* p = nullptr;
  if (on > 0) { p = new FvqaProbe(); p->stride = on; }
  drop(g_probe.exchange(p));
  return FVQA_OK;
}

extern "C" int fvqa_gemm_timing_read(int max, float* us, double* flops, int* kind) {
  FvqaProbe* p = g_probe.load();
  if (!p) return 0;
  std::lock_guard<std::mutex> g(p->mu);
  const int n = (int)p->recs.size();
  if (max <= 0) return n;                               // size query: record untouched
  for (int i = 0; i < n; ++i) {
    FvqaProbeRec& r = p->recs[i];
    float ms = -2e-3f;                                  // (-2 us: launch counted, not bracketed)
    if (r.timed && (hipEventSynchronize(r.e1) != hipSuccess || hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess))
      ms = -1e-3f;
    if (i < max) {
      if (us) us[i] = ms * 1e3f;
      if (flops) flops[i] = r.flops;
      if (kind) kind[i] = r.kind;
    }
    if (r.timed) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  }
  p->recs.clear();
  return n;
}
