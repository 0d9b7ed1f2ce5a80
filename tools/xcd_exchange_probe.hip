// Probe for the next round: what the split-K hand-off of the N = 4096 projections would cost if the four pieces of a tile sat on
// ONE XCD and exchanged their partial blocks through that XCD's L2, instead of on four XCDs through the fabric (what
// csrc/gemm4w_asm.h does today: write-through `sc1` stores to a slab, `sc1` loads by the partners).
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/xcd_exchange_probe tools/xcd_exchange_probe.hip && tools/bin/xcd_exchange_probe
//
// 256 workgroups x 256 threads, teams of 4. Every workgroup publishes three 48 KiB blocks (the three quarters of a 256 x 256 tile of
// 24-bit partials it does not own: 144 KiB, as in the product), raises its flag, waits for its partners' flags, fetches the three
// blocks meant for it and folds them into a checksum that the host verifies (a stale read shows). Modes:
//   team layout  X: workgroups 4t .. 4t+3 (four XCDs: workgroup i runs on XCD i % 8)      S: t, t+8, t+16, t+24 (one XCD)
//   stores       sc1 (write-through, agent scope) | plain
//   loads        sc1 | sc0 (past the CU's vector cache, served by the XCD's L2) | plain
// Only X / sc1 / sc1 is inside the documented memory model for cross-XCD partners; S / plain / sc0 leans on the L2 being the
// coherence point of an XCD. The probe reports microseconds per launch (launch + exchange; nothing else in the kernel) and whether
// every checksum matched.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int WG = 256, TEAM = 4, BLOCK_WORDS = 48 * 1024 / 16;      // 16-byte words per block: 3072 = 12 per thread

template <int ST, int LD>   // ST: 0 plain, 1 sc1. LD: 0 plain, 1 sc1, 2 sc0
__global__ __launch_bounds__(256) void exchange_k(u32x4* slabs, unsigned long long* flags, unsigned long long epoch, int same_xcd,
                                                  unsigned* sums, long long* xcc) {
  const int wg = blockIdx.x, tid = threadIdx.x;
  int team, piece;
  if (same_xcd) { const int blk = wg / 32, r = wg % 32; piece = r / 8; team = blk * 8 + r % 8; }
  else { team = wg / 4; piece = wg % 4; }
  auto member = [&](int t, int p) { return same_xcd ? (t / 8) * 32 + p * 8 + t % 8 : t * 4 + p; };
  if (tid == 0 && xcc) { unsigned id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id)); xcc[wg] = id & 0xf; }
  u32x4* mine = slabs + (size_t)wg * 3 * BLOCK_WORDS;
  // publish: block j (j = 0..2) is meant for partner piece (piece + 1 + j) % 4
  for (int j = 0; j < 3; ++j)
    for (int i = tid; i < BLOCK_WORDS; i += 256) {
      const unsigned v = (unsigned)(epoch * 2654435761u) ^ (unsigned)(wg * 7919 + j * 104729 + i);
      u32x4 q = {v, v + 1, v + 2, v + 3};
      u32x4* p = mine + (size_t)j * BLOCK_WORDS + i;
      if (ST == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(q) : "memory");
      else asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(q) : "memory");
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) __hip_atomic_store(flags + wg, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (tid < 3) {
    const int partner = member(team, (piece + 1 + tid) % 4);
    long long spins = 0;
    while (__hip_atomic_load(flags + partner, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch && ++spins < (1ll << 18)) {}
  }
  __syncthreads();
  unsigned acc = 0;
  for (int j = 0; j < 3; ++j) {
    const int pp = (piece + 1 + j) % 4, partner = member(team, pp);
    const int jb = (piece - pp - 1 + 8) % 4;                // which of the partner's blocks is meant for me
    const u32x4* src = slabs + (size_t)partner * 3 * BLOCK_WORDS + (size_t)jb * BLOCK_WORDS;
    for (int i = tid; i < BLOCK_WORDS; i += 256) {
      u32x4 q;
      const u32x4* p = src + i;
      if (LD == 1) asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(q) : "v"(p) : "memory");
      else if (LD == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(q) : "v"(p) : "memory");
      else asm volatile("global_load_dwordx4 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(q) : "v"(p) : "memory");
      const unsigned v = (unsigned)(epoch * 2654435761u) ^ (unsigned)(partner * 7919 + jb * 104729 + i);
      acc += (q.x != v) + (q.y != v + 1) + (q.z != v + 2) + (q.w != v + 3);     // count of stale / wrong words
    }
  }
  atomicAdd(sums + wg, acc);
}

template <int ST, int LD>
void run(const char* name, int same_xcd, u32x4* slabs, unsigned long long* flags, unsigned* sums, long long* xcc, int reps) {
  static unsigned long long epoch = 1;
  CHECK(hipMemset(sums, 0, WG * sizeof(unsigned)));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((exchange_k<ST, LD>), dim3(WG), dim3(256), 0, 0, slabs, flags, epoch++, same_xcd, sums, xcc);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((exchange_k<ST, LD>), dim3(WG), dim3(256), 0, 0, slabs, flags, epoch++, same_xcd, sums, xcc);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned> h(WG);
  CHECK(hipMemcpy(h.data(), sums, WG * sizeof(unsigned), hipMemcpyDeviceToHost));
  unsigned long long bad = 0; for (unsigned v : h) bad += v;
  printf("%-44s team on %s  %7.2f us per launch   wrong words: %llu\n", name, same_xcd ? "ONE XCD  " : "four XCDs", ms * 1e3 / reps, bad);
}

int main() {
  u32x4* slabs; unsigned long long* flags; unsigned* sums; long long* xcc;
  CHECK(hipMalloc(&slabs, (size_t)WG * 3 * BLOCK_WORDS * 16));
  CHECK(hipMalloc(&flags, WG * 8)); CHECK(hipMemset(flags, 0, WG * 8));
  CHECK(hipMalloc(&sums, WG * 4)); CHECK(hipMalloc(&xcc, WG * 8)); CHECK(hipMemset(xcc, 0xff, WG * 8));
  const int reps = 200;
  run<1, 1>("stores sc1, loads sc1 (the product's recipe)", 0, slabs, flags, sums, xcc, reps);
  std::vector<long long> hx(WG);
  CHECK(hipMemcpy(hx.data(), xcc, WG * 8, hipMemcpyDeviceToHost));
  int rr = 1; for (int i = 0; i < WG; ++i) if (hx[i] != i % 8) rr = 0;
  printf("workgroup i runs on XCD i %% 8: %s (first 16: ", rr ? "yes" : "NO");
  for (int i = 0; i < 16; ++i) printf("%lld ", hx[i]);
  printf(")\n");
  run<1, 1>("stores sc1, loads sc1", 1, slabs, flags, sums, xcc, reps);
  run<1, 2>("stores sc1, loads sc0", 1, slabs, flags, sums, xcc, reps);
  run<0, 2>("stores plain, loads sc0", 1, slabs, flags, sums, xcc, reps);
  run<0, 0>("stores plain, loads plain", 1, slabs, flags, sums, xcc, reps);
  run<0, 2>("stores plain, loads sc0", 0, slabs, flags, sums, xcc, reps);
  run<0, 0>("stores plain, loads plain", 0, slabs, flags, sums, xcc, reps);
  return 0;
}
