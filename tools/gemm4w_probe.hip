// Stand-alone probe of the 4-wave main loop (flipped-vqa_amd/csrc/gemm4w_loop.h) on an MI355X: correctness of whole
// tiles against a host fp64 reference (ragged edges included) and the launch time on the C2 projection shapes for tile
// widths 256 / 192 / 176, next to each other in one process on random data. Tuning aid, not product code: the epilogue
// here is a plain masked store from the accumulator registers.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/gemm4w_probe.hip -o tools/bin/gemm4w_probe
#define FVQA_G4_STAMPS 1
#include "../flipped-vqa_amd/csrc/gemm4w_kernel.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace fvqa_ring4;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ int xcd_chunk(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + (bid >> 3);
}

struct PArgs { const bf16_t* A; const bf16_t* B; bf16_t* C; int M, N, K, lda, ldb, ldc, tm, tn, kdiv; unsigned long long* stamps; };

template <int NBT>
__global__ __launch_bounds__(256) void probe_k(const PArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int wid = xcd_chunk(blockIdx.x, gridDim.x);
  const int ntiles = a.tm * a.tn;
  for (int t = wid; t < ntiles; t += gridDim.x) {
    const int mt = t % a.tm, nt = t / a.tm;                  // consecutive work ids (one XCD) share a weight panel
    const int m0 = mt * TM, n0 = nt * Geo<NBT>::TN;
    f32x16 acc[NBT];
#pragma unroll
    for (int j = 0; j < NBT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    unsigned long long t0 = 0, r0 = 0, t1 = 0, r1 = 0;
    if (a.stamps) asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
    ring4_loop<NBT>(acc, lds0, a.A, a.B, a.M, a.N, a.lda, a.ldb, m0, n0, 0, a.K / 64 / a.kdiv, w, lane, 0);
    if (a.stamps) {
      asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
      if (tid == 0) { a.stamps[blockIdx.x * 4 + 0] = t0; a.stamps[blockIdx.x * 4 + 1] = r0; a.stamps[blockIdx.x * 4 + 2] = t1; a.stamps[blockIdx.x * 4 + 3] = r1; }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + 64 * w + 16 * i + (lane & 15);
#pragma unroll
      for (int j = 0; j < NBT; ++j) {
        const int n = n0 + 16 * j + (lane >> 4) * 4;
        if (m < a.M && n < a.N) {
          float v[4] = {acc[j][4 * i], acc[j][4 * i + 1], acc[j][4 * i + 2], acc[j][4 * i + 3]};
          Vec4<bf16_t>::store(a.C + (size_t)m * a.ldc + n, v);
        }
      }
    }
    __syncthreads();
  }
}

static unsigned short f2bf(float f) {
  unsigned u; memcpy(&u, &f, 4);
  u += 0x7FFF + ((u >> 16) & 1);
  return (unsigned short)(u >> 16);
}
static float bf2f(unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }

template <int NBT>
static float run(const PArgs& a, int reps, int grid_cap, double* clock_ghz, double* loop_us) {
  auto k = probe_k<NBT>;
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, Geo<NBT>::RING_BYTES));
  PArgs b = a;
  b.tm = (a.M + TM - 1) / TM;
  b.tn = (a.N + Geo<NBT>::TN - 1) / Geo<NBT>::TN;
  int grid = b.tm * b.tn;
  if (grid > grid_cap) grid = grid_cap;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  PArgs nb = b; nb.stamps = nullptr;
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k, dim3(grid), dim3(256), Geo<NBT>::RING_BYTES, 0, nb);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  if (clock_ghz && b.stamps) {                                 // one stamped launch right behind the timed ones
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), Geo<NBT>::RING_BYTES, 0, b);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> st(grid * 4);
    CK(hipMemcpy(st.data(), b.stamps, grid * 4 * 8, hipMemcpyDeviceToHost));
    std::vector<double> cl, us;
    for (int g = 0; g < grid; ++g) {
      const double dt = (double)(st[g * 4 + 2] - st[g * 4]), dr = (double)(st[g * 4 + 3] - st[g * 4 + 1]);
      if (dr > 0) { cl.push_back(dt / dr * 0.1); us.push_back(dr / 100.0); }
    }
    std::sort(cl.begin(), cl.end()); std::sort(us.begin(), us.end());
    *clock_ghz = cl.empty() ? 0 : cl[cl.size() / 2];
    *loop_us = us.empty() ? 0 : us[us.size() / 2];
  }
  CK(hipGetLastError());
  return ms * 1e3f / reps;
}

static float dispatch(int nbt, const PArgs& a, int reps, int cap, double* c = nullptr, double* l = nullptr) {
  switch (nbt) {
    case 16: return run<16>(a, reps, cap, c, l);
    case 12: return run<12>(a, reps, cap, c, l);
    case 11: return run<11>(a, reps, cap, c, l);
    default: printf("unsupported NBT %d\n", nbt); exit(1);
  }
}

static void fill(std::vector<unsigned short>& v, float scale, unsigned seed) {
  unsigned s = seed * 2654435761u + 12345u;
  for (auto& x : v) { s = s * 1664525u + 1013904223u; x = f2bf(((float)(s >> 8) / 8388608.0f - 1.0f) * scale); }
}

static int check(int M, int N, int K, int nbt) {
  std::vector<unsigned short> hA((size_t)M * K), hB((size_t)N * K), hC((size_t)M * N);
  fill(hA, 1.f, 1); fill(hB, 1.f / sqrtf((float)K), 2);
  bf16_t *dA, *dB, *dC;
  CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dB, hB.size() * 2)); CK(hipMalloc(&dC, hC.size() * 2));
  CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemset(dC, 0xFF, hC.size() * 2));
  PArgs a{dA, dB, dC, M, N, K, K, K, N, 0, 0, 1, nullptr};
  dispatch(nbt, a, 1, 256);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost));
  double maxerr = 0, maxref = 0;
  const size_t total = (size_t)M * N;
  const size_t stride = total > 400000 ? total / 200003 : 1;
  size_t bad = 0;
  for (size_t idx = 0; idx < total; idx += stride) {
    const int m = (int)(idx / N), n = (int)(idx % N);
    double r = 0;
    for (int k = 0; k < K; ++k) r += (double)bf2f(hA[(size_t)m * K + k]) * (double)bf2f(hB[(size_t)n * K + k]);
    const double e = fabs((double)bf2f(hC[idx]) - r);
    if (e > maxerr) maxerr = e;
    if (fabs(r) > maxref) maxref = fabs(r);
    if (!(e <= 0.02 * fabs(r) + 0.02)) ++bad;
  }
  printf("check NBT=%2d %5d x %5d x %5d: max |err| %.3e (max |ref| %.3f) bad %zu  %s\n", nbt, M, N, K, maxerr, maxref, bad,
         bad == 0 && maxerr < 0.03 * maxref + 0.02 ? "OK" : "FAIL");
  CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
  return bad == 0 ? 0 : 1;
}


// ---- the production kernel (gemm4w_kernel.h) with its epilogues and rider, phase stamps per workgroup
using namespace fvqa_g4;
template <int NBT, int EPI>
static void epi_case(const char* name, G4Args a, int n_cu, bool rider, int reps) {
  auto k = gemm4w_k<NBT, bf16_t, EPI>;
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, Geo<NBT>::RING_BYTES));
  a.tm = (a.M + 255) / 256; a.tn = (a.N + 16 * NBT - 1) / (16 * NBT); a.tiles = a.tm * a.tn;
  a.grid = (rider || a.tiles > n_cu) ? n_cu : a.tiles;
  a.rounds = (a.tiles + a.grid - 1) / a.grid;
  a.rider.on = rider ? 1 : 0;
  const int light = a.grid - (a.tiles - (a.rounds - 1) * a.grid);
  unsigned long long* st = a.stamps;
  a.stamps = nullptr;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int r = 0; r < reps / 4; ++r) hipLaunchKernelGGL(k, dim3(a.grid), dim3(256), Geo<NBT>::RING_BYTES, 0, a);
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k, dim3(a.grid), dim3(256), Geo<NBT>::RING_BYTES, 0, a);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  a.stamps = st;
  CK(hipMemset(st, 0, 256 * 8 * 8));
  hipLaunchKernelGGL(k, dim3(a.grid), dim3(256), Geo<NBT>::RING_BYTES, 0, a);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(256 * 8);
  CK(hipMemcpy(h.data(), st, 256 * 8 * 8, hipMemcpyDeviceToHost));
  unsigned long long t0 = ~0ull;
  for (int g = 0; g < a.grid; ++g) if (h[g * 8]) t0 = std::min(t0, h[g * 8]);
  printf("%-26s NBT=%2d tiles %3d grid %3d light %3d rider %d: %7.1f us/launch |", name, NBT, a.tiles, a.grid, light, (int)rider, ms * 1e3 / reps);
  const char* lab[6] = {"start", "loop0", "stor0", "loop1", "stor1", "rider"};
  for (int sl = 0; sl < 6; ++sl) {
    std::vector<double> v;
    for (int g = 0; g < a.grid; ++g) if (h[g * 8 + sl]) v.push_back((double)(h[g * 8 + sl] - t0) / 100.0);
    if (v.empty()) continue;
    std::sort(v.begin(), v.end());
    printf(" %s n=%zu med %.1f max %.1f |", lab[sl], v.size(), v[v.size() / 2], v.back());
  }
  printf("\n");
  fflush(stdout);
}

static int epi_mode(int reps) {
  const int M = 1024, K = 4096, S = 128;
  unsigned long long* stamps;
  CK(hipMalloc(&stamps, 256 * 8 * 8));
  auto dev_bf = [&](size_t n, float scale, unsigned seed) {
    std::vector<unsigned short> h(n); fill(h, scale, seed);
    bf16_t* d; CK(hipMalloc(&d, n * 2)); CK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice)); return d; };
  bf16_t* A = dev_bf((size_t)M * K, 1.f, 11);
  bf16_t* B = dev_bf((size_t)22016 * K, 1.f / 64.f, 12);
  bf16_t* R = dev_bf((size_t)M * 22016, 1.f, 13);          // residual / (s, t) rows
  bf16_t *C, *C2; CK(hipMalloc(&C, (size_t)M * 22016 * 2)); CK(hipMalloc(&C2, (size_t)M * 11008 * 2));
  std::vector<float> hc(S * 64), hs(S * 64);
  for (int p = 0; p < S; ++p) for (int i = 0; i < 64; ++i) { const float th = p * powf(10000.f, -2.f * i / 128.f); hc[p * 64 + i] = cosf(th); hs[p * 64 + i] = sinf(th); }
  float *dc, *ds; CK(hipMalloc(&dc, hc.size() * 4)); CK(hipMalloc(&ds, hs.size() * 4));
  CK(hipMemcpy(dc, hc.data(), hc.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(ds, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
  bf16_t* rA = dev_bf((size_t)16 * 8192, 1.f, 14);
  bf16_t* rB = dev_bf((size_t)8192 * 8192, 1.f / 64.f, 15);
  bf16_t* rC; CK(hipMalloc(&rC, (size_t)16 * 8192 * 4));
  G4Args a{};
  a.A = A; a.B = B; a.C = C; a.R = R; a.C2 = C2; a.M = M; a.K = K; a.lda = K; a.ldb = K;
  a.rope_cos = dc; a.rope_sin = ds; a.rope_S = S; a.rope_hp = 64; a.rope_hmask = 127; a.stamps = stamps;
  for (int round = 0; round < 2; ++round) {
    // QKV: N = 12288, rider = adapter K/V rows: 10 x 8192 x 4096 (bf16 out)
    a.N = 12288; a.ldc = 12288; a.rope_cols = 8192;
    a.rider = G4Rider{rA, B + (size_t)4096 * K, rC, 10, 8192, 4096, 4096, K, 8192, 0, 0};
    epi_case<16, FVQA_EPI_NONE>("qkv none", a, 256, false, reps);
    epi_case<16, FVQA_EPI_ROPE>("qkv rope", a, 256, false, reps);
    epi_case<16, FVQA_EPI_ROPE>("qkv rope + rider", a, 256, true, reps);
    epi_case<12, FVQA_EPI_NONE>("qkv none", a, 256, false, reps);
    epi_case<12, FVQA_EPI_RESIDUAL>("qkv residual", a, 256, false, reps);
    epi_case<12, FVQA_EPI_ROPE>("qkv rope", a, 256, false, reps);
    // W2^T: N = 11008 hidden, C / R rows of 22016 (AB16); rider = adapter gradient rows: 10 x 4096 x 8192, fp32 +=
    a.N = 11008; a.ldc = 22016;
    a.rider = G4Rider{rA, rB, rC, 10, 4096, 8192, 8192, 8192, 4096, 1, 0};
    epi_case<16, FVQA_EPI_SWIGLU_BWD_ST>("w2t swiglu' ", a, 256, false, reps);
    epi_case<16, FVQA_EPI_SWIGLU_BWD_ST>("w2t swiglu' + rider", a, 256, true, reps);
    epi_case<12, FVQA_EPI_SWIGLU_BWD_ST>("w2t swiglu' ", a, 256, false, reps);
    epi_case<12, FVQA_EPI_SWIGLU_BWD_ST>("w2t swiglu' + rider", a, 256, true, reps);
    epi_case<11, FVQA_EPI_SWIGLU_BWD_ST>("w2t swiglu' ", a, 256, false, reps);
    // W1|W3: N = 22016, z = (M, 11008); rider = adapter K/V rows of the next layer
    a.N = 22016; a.ldc = 22016;
    a.rider = G4Rider{rA, B + (size_t)4096 * K, rC, 10, 8192, 4096, 4096, K, 8192, 0, 0};
    epi_case<12, FVQA_EPI_NONE>("w13 none", a, 256, false, reps);
    epi_case<12, FVQA_EPI_SWIGLU_FWD_ST>("w13 swiglu", a, 256, false, reps);
    epi_case<12, FVQA_EPI_SWIGLU_FWD_ST>("w13 swiglu + rider", a, 256, true, reps);
    epi_case<16, FVQA_EPI_SWIGLU_FWD_ST>("w13 swiglu", a, 256, false, reps);
  }
  return 0;
}

int main(int argc, char** argv) {
  int fails = 0;
  if (argc > 1 && !strcmp(argv[1], "check")) {
    for (int nbt : {16, 12, 11, 8}) {
      fails += check(256, 16 * nbt, 64, nbt);          // one stage
      fails += check(256, 16 * nbt, 128, nbt);         // two
      fails += check(256, 16 * nbt, 192, nbt);         // three
      fails += check(300, 16 * nbt + 40, 512, nbt);    // ragged M and N
      fails += check(1024, 16 * nbt * 5, 1024, nbt);
    }
    printf(fails ? "CHECK FAILED\n" : "all checks passed\n");
    return fails ? 1 : 0;
  }

  if (argc > 1 && !strcmp(argv[1], "epi")) return epi_mode(argc > 2 ? atoi(argv[2]) : 300);
  if (argc > 1 && !strcmp(argv[1], "split")) {
    // loop of a split-K piece on 256 workgroups: 256 x 256 tiles cut 4 ways along K (the shipped partition of the N = 4096
    // outputs) against 256 x 128 tiles cut 2 ways (a third of the exchange bytes per workgroup)
    struct Sp { const char* name; int N, K, kdiv, nbt; };
    const Sp sp[] = {{"wo   256x256 /4", 16384, 4096, 4, 16}, {"wo   256x128 /2", 8192, 4096, 2, 8},
                     {"w2   256x256 /4", 16384, 11008, 4, 16}, {"w2   256x128 /2", 8192, 11008, 2, 8},
                     {"qkvt 256x256 /4", 16384, 12288, 4, 16}, {"qkvt 256x128 /2", 8192, 12288, 2, 8},
                     {"w13t 256x256 /4", 16384, 22016, 4, 16}, {"w13t 256x128 /2", 8192, 22016, 2, 8}};
    const int reps = argc > 2 ? atoi(argv[2]) : 1500;
    unsigned long long* stamps;
    CK(hipMalloc(&stamps, 256 * 4 * 8));
    std::vector<unsigned short> hA((size_t)1024 * 22016), hB((size_t)16384 * 22016);
    fill(hA, 1.f, 3); fill(hB, 1.f / 64.f, 4);
    bf16_t *dA, *dB, *dC;
    CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dB, hB.size() * 2)); CK(hipMalloc(&dC, (size_t)1024 * 16384 * 2));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
    for (int round = 0; round < 3; ++round)
      for (const Sp& s : sp) {
        PArgs a{dA, dB, dC, 1024, s.N, s.K, 22016, 22016, s.N, 0, 0, s.kdiv, stamps};
        dispatch(s.nbt, a, reps / 4, 256);
        double clk = 0, lus = 0;
        const float us = dispatch(s.nbt, a, reps, 256, &clk, &lus);
        const double fl = 2.0 * 1024 * s.N * (s.K / s.kdiv);
        printf("%-16s K %5d: %7.1f us/launch %6.0f TF/s | loop %6.1f us, clock %.3f GHz\n", s.name, s.K, us, fl / us / 1e6, lus, clk);
        fflush(stdout);
      }
    return 0;
  }
  // timing: C2 shapes. name, M, N, K, kdiv (K range per workgroup = K / kdiv: the loop of a split-K piece, no exchange)
  struct Sh { const char* name; int M, N, K, kdiv; };
  const Sh shapes[] = {{"qkv_fwd", 1024, 12288, 4096, 1}, {"w2t_bwd", 1024, 11008, 4096, 1}, {"w13_fwd", 1024, 22016, 4096, 1},
                       {"full256", 1024, 16384, 4096, 1}, {"wo_piece", 1024, 16384, 4096, 4}, {"w13t_piece", 1024, 16384, 22016, 4}};
  const int reps = argc > 1 ? atoi(argv[1]) : 2000;
  unsigned long long* stamps;
  CK(hipMalloc(&stamps, 256 * 4 * 8));
  for (const Sh& s : shapes) {
    std::vector<unsigned short> hA((size_t)s.M * s.K), hB((size_t)s.N * s.K);
    fill(hA, 1.f, 3); fill(hB, 1.f / sqrtf((float)s.K), 4);
    bf16_t *dA, *dB, *dC;
    CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dB, hB.size() * 2)); CK(hipMalloc(&dC, (size_t)s.M * s.N * 2));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
    PArgs a{dA, dB, dC, s.M, s.N, s.K, s.K, s.K, s.N, 0, 0, s.kdiv, stamps};
    const double fl = 2.0 * s.M * s.N * (s.K / s.kdiv);
    for (int round = 0; round < 3; ++round)
      for (int nbt : {16, 12, 11}) {
        const int tiles = ((s.M + 255) / 256) * ((s.N + 16 * nbt - 1) / (16 * nbt));
        dispatch(nbt, a, reps / 4, 256);                 // warm
        double clk = 0, lus = 0;
        const float us = dispatch(nbt, a, reps, 256, &clk, &lus);
        printf("%-10s %5d x %5d x %5d /%d  NBT=%2d tiles %3d  %8.1f us  %7.0f TF/s   in-loop: clock %.3f GHz, first tile's loop %.1f us\n", s.name,
               s.M, s.N, s.K, s.kdiv, nbt, tiles, us, fl / us / 1e6, clk, lus);
        fflush(stdout);
      }
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
  }
  return 0;
}
