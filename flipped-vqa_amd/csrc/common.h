// Shared device helpers for the Flipped-VQA gfx950 kernels. CDNA4 only: 64-lane wavefronts.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include "../../include/fvqa.h"

#define FVQA_WAVE 64

// The 16-bit storage type of THIS build of the library. Every kernel source is compiled twice (fvqa/build.py):
//   libfvqa_hip.so      bf16 storage (FVQA_H16; the production build of the BASELINE configs) + the exact-fp32 build
//   libfvqa_hip_f16.so  -DFVQA_H16_F16: IEEE fp16 storage (FVQA_F16) — the reference's own storage type (llama_vqa.py:63 builds
//                       the model under torch.cuda.HalfTensor; util/misc.py:253-273 GradScaler exists for it): the same kernels
//                       with `bf16_t` = _Float16, v_mfma_f32_16x16x32_f16, fp16 pack / unpack. Same C ABI; each library accepts
//                       its own 16-bit dtype code only (FVQA_H16) and the host binds the one the model's storage dtype needs.
#ifdef FVQA_H16_F16
typedef _Float16 bf16_t;                                   // (the name stays: "the 16-bit storage type of this build")
typedef _Float16 h16x8_t __attribute__((ext_vector_type(8)));
#define FVQA_H16 FVQA_F16
#define FVQA_MFMA_H16_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_f16
#define FVQA_MFMA_H16_ASM "v_mfma_f32_16x16x32_f16"
#else
typedef __hip_bfloat16 bf16_t;
typedef __bf16 h16x8_t __attribute__((ext_vector_type(8)));
#define FVQA_H16 FVQA_BF16
#define FVQA_MFMA_H16_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define FVQA_MFMA_H16_ASM "v_mfma_f32_16x16x32_bf16"
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

#define FVQA_CHECK_LAUNCH()                                   \
  do {                                                        \
    hipError_t e__ = hipGetLastError();                       \
    if (e__ != hipSuccess) return -(int)e__ - 1000;           \
  } while (0)

// ---- storage <-> fp32 ------------------------------------------------------------------
// (the bf16-named helpers convert the build's 16-bit storage type: bf16 by shifts, fp16 by v_cvt)
#ifdef FVQA_H16_F16
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) { return (float)__builtin_bit_cast(_Float16, b); }
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
  return __builtin_bit_cast(unsigned short, (_Float16)f);  // v_cvt_f16_f32: round-nearest-even, NaN preserved, overflow -> inf
}
// the two 16-bit values of a 32-bit word as floats
__device__ __forceinline__ float h16_lo(unsigned w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(w & 0xFFFFu)); }
__device__ __forceinline__ float h16_hi(unsigned w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(w >> 16)); }
#else
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) {
  return __uint_as_float(((unsigned)b) << 16);
}
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
  // plain cast: hipcc emits v_cvt_pk_bf16_f32 (round-nearest-even, NaN preserved)
  bf16_t h = __float2bfloat16(f);
  return *reinterpret_cast<unsigned short*>(&h);
}
__device__ __forceinline__ float h16_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float h16_hi(unsigned w) { return __uint_as_float(w & 0xFFFF0000u); }
#endif

template <typename T> struct Vec4;  // 4 consecutive elements
template <> struct Vec4<float> {
  static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
    float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  static __device__ __forceinline__ void store(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
};
template <> struct Vec4<bf16_t> {
  static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[4]) {
    uint2 t = *reinterpret_cast<const uint2*>(p);
    v[0] = h16_lo(t.x); v[1] = h16_hi(t.x);
    v[2] = h16_lo(t.y); v[3] = h16_hi(t.y);
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[4]) {
    uint2 t;
    t.x = (unsigned)f32_to_bf16_bits(v[0]) | ((unsigned)f32_to_bf16_bits(v[1]) << 16);
    t.y = (unsigned)f32_to_bf16_bits(v[2]) | ((unsigned)f32_to_bf16_bits(v[3]) << 16);
    *reinterpret_cast<uint2*>(p) = t;
  }
};

template <typename T> __device__ __forceinline__ float to_f32(T x);
template <> __device__ __forceinline__ float to_f32<float>(float x) { return x; }
#ifdef FVQA_H16_F16
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t x) { return (float)x; }
#else
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t x) { return __bfloat162float(x); }
#endif
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
#ifdef FVQA_H16_F16
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float x) { return (_Float16)x; }
#else
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float x) { return __float2bfloat16(x); }
#endif
// round-trip through the storage type (the reference rounds at these points)
template <typename T> __device__ __forceinline__ float round_to(float x) { return to_f32<T>(from_f32<T>(x)); }

// ---- wave / block reductions -------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// sum over the 4 lanes of an aligned quad (lane ^ 1, lane ^ 2)
__device__ __forceinline__ float quad_sum(float v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  return v;
}
// block-wide sum for blockDim.x == 256 (4 waves); `red` is 4 floats of LDS
__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max_256(float v, float* red) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// hipFuncSetAttribute applies to the CURRENT device: a once-per-process flag would leave every other device of a process that
// drives several (tests, tools; one process per GPU never does) without its dynamic-LDS limit. One bit per device id.
#include <atomic>
static inline bool fvqa_attr_needed(std::atomic<unsigned long long>& done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
  const unsigned long long bit = 1ull << dev;
  if (done.load(std::memory_order_relaxed) & bit) return false;
  done.fetch_or(bit);
  return true;
}

static inline int fvqa_dtype_ok(int dt) { return dt == FVQA_F32 || dt == FVQA_H16; }   // (this build's 16-bit code only)
static inline size_t fvqa_dtype_size(int dt) { return dt == FVQA_F32 ? 4 : 2; }
