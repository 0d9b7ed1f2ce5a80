// Loss-scaler bookkeeping, gradient norm and AdamW for the ~4.5 M trainable fp32 scalars
// (reference util/misc.py:259-273,282-294 + torch GradScaler/AdamW called from train.py:120-121).
// Everything the optimizer step needs stays on the device: the scale, the found-inf flag and the
// step counter are device scalars, so a training step issues no device->host read here.
#include "common.h"

namespace {

constexpr int NB = 256;  // partial blocks per segment (the largest segment, visual_proj.weight, is 3.1 M scalars)

// grid (NB, n_seg): g *= 1/scale in place; partial sum of squares + non-finite flag per block. 16-byte accesses on the
// aligned body of a segment, scalars on its ragged ends; a block's elements and their order are fixed, so the norm is
// bitwise repeatable.
__global__ __launch_bounds__(256) void unscale_sq_k(float* __restrict__ grad, const int64_t* __restrict__ seg_off,
                                                    const float* __restrict__ scale, float grad_div,
                                                    float* __restrict__ part) {
  __shared__ float red[4];
  const int seg = blockIdx.y;
  const int64_t lo = seg_off[seg], hi = seg_off[seg + 1];
  // grad_div: the number of replicas whose gradients were SUMMED into `grad` (the data-parallel mean rides here instead
  // of in a pass of its own; exact for power-of-two world sizes and loss scales)
  const float inv = 1.f / (scale[0] * grad_div);
  float sq = 0.f, bad = 0.f;
  auto one = [&](int64_t i) {
    const float g = grad[i] * inv;
    grad[i] = g;
    sq += g * g;
    if (!isfinite(g)) bad = 1.f;
  };
  const int64_t a0 = (lo + 3) & ~(int64_t)3, a1 = hi & ~(int64_t)3;       // 16-byte aligned body [a0, a1)
  if (a0 < a1) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t < a0 - lo) one(lo + t);                                         // (< 4 elements)
    if (t < hi - a1) one(a1 + t);
    for (int64_t i = a0 + 4 * t; i < a1; i += (int64_t)NB * 256 * 4) {
      float4 g = *reinterpret_cast<float4*>(grad + i);
      g.x *= inv; g.y *= inv; g.z *= inv; g.w *= inv;
      *reinterpret_cast<float4*>(grad + i) = g;
      sq += g.x * g.x; sq += g.y * g.y; sq += g.z * g.z; sq += g.w * g.w;
      if (!isfinite(g.x) || !isfinite(g.y) || !isfinite(g.z) || !isfinite(g.w)) bad = 1.f;
    }
  } else {
    for (int64_t i = lo + (int64_t)blockIdx.x * 256 + threadIdx.x; i < hi; i += (int64_t)NB * 256) one(i);
  }
  sq = block_sum_256(sq, red);
  bad = block_sum_256(bad, red);
  if (threadIdx.x == 0) {
    part[((size_t)seg * NB + blockIdx.x) * 2] = sq;
    part[((size_t)seg * NB + blockIdx.x) * 2 + 1] = bad;
  }
}

// one wave per segment sums its NB partials in a fixed order; thread 0 then forms the total norm
// gemm_err (may be NULL): the error word of the persistent GEMM's workspace (include/fvqa.h). Non-zero = a split-K
// exchange of some launch of this step timed out and its outputs are garbage: the step is skipped exactly like an
// overflow (found_inf = 2 tells the host which of the two it was) — no extra device->host read on the step's path.
__global__ __launch_bounds__(1024) void norm_finish_k(const float* __restrict__ part, int n_seg,
                                                     float* __restrict__ seg_sq, float* __restrict__ found_inf,
                                                     float* __restrict__ total_norm,
                                                     const unsigned long long* __restrict__ gemm_err,
                                                     const float* __restrict__ err_lane) {
  __shared__ float red[16], redb[16];
  // sixteen waves, a wave per segment at a time: lane l takes partials l, l + 64, ... in order, then a fixed butterfly (round 5: one
  // THREAD per segment walked its NB partials one dependent load after the other: 10.7 us for 67 segments)
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float tot = 0.f, bad = 0.f;
  for (int s = w; s < n_seg; s += 16) {
    float a = 0.f, bd = 0.f;
    for (int b = lane; b < NB; b += 64) {
      a += part[((size_t)s * NB + b) * 2];
      bd += part[((size_t)s * NB + b) * 2 + 1];
    }
    a = wave_sum(a);
    bd = wave_sum(bd);
    if (lane == 0) {
      seg_sq[s] = a;
      tot += a;
      bad += bd;
    }
  }
  if (lane == 0) { red[w] = tot; redb[w] = bad; }
  __syncthreads();
  tot = bad = 0.f;
  if (threadIdx.x == 0)
    for (int i = 0; i < 16; ++i) { tot += red[i]; bad += redb[i]; }       // fixed order
  if (threadIdx.x == 0) {
    total_norm[0] = sqrtf(tot);
    float f = (bad > 0.f || !isfinite(tot)) ? 1.f : 0.f;
    if (gemm_err != nullptr &&
        __hip_atomic_load(gemm_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) f = 2.f;
    if (err_lane != nullptr && err_lane[0] != 0.f) f = 2.f;     // some rank's error word, summed in with the gradients
    found_inf[0] = f;
  }
}

__global__ __launch_bounds__(256) void adamw_k(float* __restrict__ p, const float* __restrict__ g,
                                               float* __restrict__ m, float* __restrict__ v, int64_t n, float lr,
                                               float b1, float b2, float eps, float wd,
                                               const float* __restrict__ step, const float* __restrict__ found_inf) {
  if (found_inf && found_inf[0] != 0.f) return;       // GradScaler.step: skip the whole update
  const float t = step[0] + 1.f;
  const float bc1 = 1.f - powf(b1, t);
  const float bc2s = sqrtf(1.f - powf(b2, t));
  const float step_size = lr / bc1;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float gi = g[i];
    float pi = p[i] * (1.f - lr * wd);
    const float mi = m[i] + (gi - m[i]) * (1.f - b1);
    const float vi = v[i] * b2 + gi * gi * (1.f - b2);
    pi -= step_size * mi / (sqrtf(vi) / bc2s + eps);
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
}

// after every group's adamw_k: advance the step counter and the dynamic loss scale
// (torch.cuda.amp.GradScaler.update: x backoff on overflow, x growth after `interval` clean steps)
__global__ void scaler_update_k(float* __restrict__ step, float* __restrict__ scale, float* __restrict__ tracker,
                                const float* __restrict__ found_inf, float growth, float backoff, float interval) {
  if (threadIdx.x || blockIdx.x) return;
  if (found_inf[0] == 2.f) return;         // a timed-out GEMM exchange is not an overflow: the scale keeps its history
  if (found_inf[0] != 0.f) {
    if (scale) { scale[0] *= backoff; tracker[0] = 0.f; }
  } else {
    if (step) step[0] += 1.f;
    if (scale) {
      const float t = tracker[0] + 1.f;
      if (t >= interval) { scale[0] *= growth; tracker[0] = 0.f; }
      else tracker[0] = t;
    }
  }
}

}  // namespace

extern "C" size_t fvqa_grad_norm_workspace(int n_seg) { return (size_t)(n_seg > 0 ? n_seg : 0) * NB * 2 * sizeof(float); }

extern "C" int fvqa_grad_unscale_norm(float* grad, const int64_t* seg_off, int n_seg, const float* scale,
                                      float grad_div, const void* gemm_err, const float* err_lane, float* seg_sq,
                                      float* found_inf, float* total_norm, void* workspace, size_t workspace_bytes,
                                      void* stream) {
  if (!grad || !seg_off || !scale || !seg_sq || !found_inf || !total_norm || !workspace) return FVQA_EINVAL;
  if (!(grad_div >= 1.f) || ((uintptr_t)gemm_err & 7)) return FVQA_EINVAL;
  if (n_seg <= 0 || n_seg > 65535) return FVQA_ESHAPE;
  if (workspace_bytes < fvqa_grad_norm_workspace(n_seg)) return FVQA_EALIGN;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(unscale_sq_k, dim3(NB, n_seg), dim3(256), 0, st, grad, seg_off, scale, grad_div, (float*)workspace);
  hipLaunchKernelGGL(norm_finish_k, dim3(1), dim3(1024), 0, st, (const float*)workspace, n_seg, seg_sq, found_inf,
                     total_norm, (const unsigned long long*)gemm_err, err_lane);
  FVQA_CHECK_LAUNCH();
  return FVQA_OK;
}

extern "C" int fvqa_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                               float lr, float beta1, float beta2, float eps, float weight_decay, const float* step,
                               const float* found_inf, void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || !step) return FVQA_EINVAL;
  if (n <= 0) return FVQA_ESHAPE;
  int64_t g = (n + 255) / 256;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(adamw_k, dim3((int)g), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n, lr,
                     beta1, beta2, eps, weight_decay, step, found_inf);
  FVQA_CHECK_LAUNCH();
  return FVQA_OK;
}

extern "C" int fvqa_scaler_update(float* step, float* scale, float* growth_tracker, const float* found_inf,
                                  float growth_factor, float backoff_factor, int growth_interval, void* stream) {
  if (!found_inf || (!step && !scale) || (scale && !growth_tracker)) return FVQA_EINVAL;
  hipLaunchKernelGGL(scaler_update_k, dim3(1), dim3(64), 0, (hipStream_t)stream, step, scale, growth_tracker,
                     found_inf, growth_factor, backoff_factor, (float)growth_interval);
  FVQA_CHECK_LAUNCH();
  return FVQA_OK;
}

extern "C" const char* fvqa_arch(void) { return "gfx950"; }
