// Sustained MFMA rate of the whole chip under its power limit, by instruction shape (no memory traffic: operands stay in
// registers): v_mfma_f32_16x16x32_bf16 against v_mfma_f32_32x32x16_bf16, 8 waves per CU, one workgroup per CU.
// The two shapes have the same nominal FLOP/clk; the 32x32 form reads half as many operand registers per FLOP.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_power_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void k16(const uint4* __restrict__ src, float* __restrict__ out, int iters) {
  uint4 a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = src[(threadIdx.x * 16 + i) & 4095]; b[i] = src[(threadIdx.x * 16 + 8 + i) & 4095]; }
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int h = 0; h < 2; ++h)        // 16 independent accumulators = a 64x64 register tile: 4 A x 4 B fragments, two k-steps of 32
#pragma unroll
      for (int i = 0; i < 16; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a[(i >> 2) + 4 * h]),
                                                         __builtin_bit_cast(bf16x8_t, b[(i & 3) + 4 * h]), acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

__global__ __launch_bounds__(512) void k32(const uint4* __restrict__ src, float* __restrict__ out, int iters) {
  uint4 a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = src[(threadIdx.x * 16 + i) & 4095]; b[i] = src[(threadIdx.x * 16 + 8 + i) & 4095]; }
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int h = 0; h < 4; ++h)        // the same 64x64 register tile: 2 A x 2 B fragments of 32 rows, four k-steps of 16
#pragma unroll
      for (int i = 0; i < 4; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[(i >> 1) + 2 * h]),
                                                         __builtin_bit_cast(bf16x8_t, b[(i & 1) + 2 * h]), acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
  int cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  std::vector<unsigned> h(4096 * 4);
  srand(1);
  for (auto& x : h) {                              // random bf16 pairs of magnitude ~1 (exponent 0x3f / 0xbf)
    unsigned lo = (rand() & 0x80ff) | 0x3f00, hi = (rand() & 0x80ff) | 0x3f00;
    x = lo | (hi << 16);
  }
  uint4* src; float* out;
  hipMalloc(&src, h.size() * 4); hipMalloc(&out, (size_t)cus * 512 * 4);
  hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;                         // ~15-20 ms per launch
  const double flop = (double)cus * 8 * iters * (2.0 * 64 * 64 * 64);   // per wave and iteration: a 64x64 tile, 64 of K
  for (int rep = 0; rep < 3; ++rep)
    for (int which = 0; which < 2; ++which) {
      float best = 1e30f, sum = 0.f; const int n = 40;     // ~0.7 s of continuous load per measurement
      for (int i = 0; i < n; ++i) {
        hipEventRecord(e0);
        if (which == 0) hipLaunchKernelGGL(k16, dim3(cus), dim3(512), 0, 0, src, out, iters);
        else hipLaunchKernelGGL(k32, dim3(cus), dim3(512), 0, 0, src, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (i >= n / 2) { sum += ms; best = ms < best ? ms : best; }
      }
      const double avg = sum / (n / 2);
      printf("%s  %d CUs x 8 waves: %.2f ms/launch (best %.2f)  sustained %.0f TF/s (best %.0f)\n",
             which == 0 ? "16x16x32" : "32x32x16", cus, avg, best, flop / avg / 1e9, flop / best / 1e9);
      fflush(stdout);
    }
  return 0;
}
