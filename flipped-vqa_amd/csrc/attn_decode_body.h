// Body of the one-query-row attention of the generation path, shared by the stand-alone kernel (attn_decode.hip) and the
// persistent per-token kernel (decode.hip): one 256-thread workgroup per (head h, sequence n). See attn_decode.hip for the
// contract. LDS is the caller's: sc[SMAX] (text scores), sa[16], red[8], part[4][DH].
#pragma once
#include "common.h"

namespace fvqa_decode {



constexpr int DH = 128;
constexpr int HP = DH / 2;
constexpr int SMAX = 4096;               // scores of one row live in LDS

template <typename T> struct Chunk;      // 16 bytes of a row = CH head dims
template <> struct Chunk<bf16_t> {
  static constexpr int CH = 8;
  static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
    const uint4 q = *reinterpret_cast<const uint4*>(p);
    v[0] = h16_lo(q.x); v[1] = h16_hi(q.x);
    v[2] = h16_lo(q.y); v[3] = h16_hi(q.y);
    v[4] = h16_lo(q.z); v[5] = h16_hi(q.z);
    v[6] = h16_lo(q.w); v[7] = h16_hi(q.w);
  }
};
template <> struct Chunk<float> {
  static constexpr int CH = 4;
  static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
    const float4 q = *reinterpret_cast<const float4*>(p);
    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
  }
};

// rotate CH consecutive head dims starting at dim d0 (even) of a row at position `p` (tables (S, 64) fp32)
// `on` == false leaves the chunk as stored (cos 1, sin 0 blended in: no branch around the table loads)
template <int CH>
__device__ __forceinline__ void rope_chunk(float (&v)[CH], const float* cs, const float* sn, int p, int d0,
                                           bool on = true) {
  float c[CH / 2], s[CH / 2];
  const float* cp = cs + (size_t)p * HP + d0 / 2;
  const float* sp = sn + (size_t)p * HP + d0 / 2;
  if constexpr (CH == 8) {                         // 4 pairs: one 16-byte load per table
    const float4 cq = *reinterpret_cast<const float4*>(cp), sq = *reinterpret_cast<const float4*>(sp);
    c[0] = cq.x; c[1] = cq.y; c[2] = cq.z; c[3] = cq.w;
    s[0] = sq.x; s[1] = sq.y; s[2] = sq.z; s[3] = sq.w;
  } else {
    const float2 cq = *reinterpret_cast<const float2*>(cp), sq = *reinterpret_cast<const float2*>(sp);
    c[0] = cq.x; c[1] = cq.y; s[0] = sq.x; s[1] = sq.y;
  }
#pragma unroll
  for (int e = 0; e < CH; e += 2) {
    const float ce = on ? c[e / 2] : 1.f, se = on ? s[e / 2] : 0.f;
    const float a = v[e], b = v[e + 1];
    v[e] = a * ce - b * se;
    v[e + 1] = a * se + b * ce;
  }
}


// WT: the output row is stored write-through at agent scope (the persistent kernel's readers sit on other XCDs).
template <typename T, bool WT = false>
__device__ __forceinline__ void attn_decode_body(const T* __restrict__ qkv_row, T* __restrict__ cache, T* __restrict__ o_row,
                                                 const float* __restrict__ gate1, const float* __restrict__ gate2,
                                                 const int32_t* __restrict__ vstart, const int64_t* __restrict__ pos,
                                                 const float* __restrict__ cs, const float* __restrict__ sn, int n_seq, int S,
                                                 int H, int A, int F, int cache_rot, int h, int n, float* sc, float* sa,
                                                 float* red, float (*part)[DH]) {
  constexpr int CH = Chunk<T>::CH;                 // head dims per 16-byte chunk
  constexpr int NCH = DH / CH;                     // chunks per row
  constexpr int CPL = NCH / 4;                     // chunks per lane of a key's quad
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int D = H * DH;
  const size_t ld = (size_t)3 * D;
  int p = (int)pos[n];
  p = p < 0 ? 0 : (p >= S ? S - 1 : p);
  const T* rowq = qkv_row + (size_t)n * ld + h * DH;            // the new token: q | k | v at +0, +D, +2D
  const T* seq = cache + (size_t)n * S * ld + h * DH;
  const T* arow = cache + (size_t)n_seq * S * ld + h * DH;      // adapter rows (k, v column blocks)
  const float scale = rsqrtf((float)DH);
  const int vs = vstart[n];
  const bool biased_row = vs >= 0 && p >= vs + F;
  const float g2 = gate2[h];

  // ---- this lane's share of q: chunks part + 4u of the row, rotated at position p
  const int part_id = lane & 3, slot = lane >> 2;               // 16 keys per wave pass, 64 per workgroup pass
  float q[CPL][CH];
#pragma unroll
  for (int u = 0; u < CPL; ++u) {
    const int c = part_id + 4 * u;
    Chunk<T>::load(rowq + c * CH, q[u]);
    rope_chunk<CH>(q[u], cs, sn, p, c * CH);
#pragma unroll
    for (int e = 0; e < CH; ++e) q[u][e] = round_to<T>(q[u][e]);     // as the prefill holds it: rotated, in the storage type
  }
  auto dot_row = [&](const T* krow, int rot_pos, bool rot) {      // rot == false: the row is used as stored
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < CPL; ++u) {
      const int c = part_id + 4 * u;
      float k[CH];
      Chunk<T>::load(krow + c * CH, k);
      rope_chunk<CH>(k, cs, sn, rot_pos, c * CH, rot);
#pragma unroll
      for (int e = 0; e < CH; ++e) acc += q[u][e] * (rot ? round_to<T>(k[e]) : k[e]);   // a key rotated here is rounded as a stored one
    }
    return quad_sum(acc);
  };
  // ---- values of the first 128 keys (wave w: keys w, w+4, ...) go in flight first: they return under the
  // score pass and the softmax (one memory round trip for keys and values together)
  const int d = 2 * lane;
  auto load2 = [&](const T* vrow, float& a, float& b) {
    if constexpr (sizeof(T) == 2) {
      const unsigned u = *reinterpret_cast<const unsigned*>(vrow + d);
      a = h16_lo(u); b = h16_hi(u);
    } else {
      const float2 u = *reinterpret_cast<const float2*>(vrow + d);
      a = u.x; b = u.y;
    }
  };
  // (every load is issued unconditionally from a clamped row — a per-element "load or zero" on a runtime condition makes
  // hipcc branch around each load and wait for it alone; keys beyond p get weight 0 below)
  float va[32], vb[32];
#pragma unroll
  for (int u = 0; u < 32; ++u) {
    const int j = w + 4 * u, jj = j < p ? j : p;
    load2((jj == p) ? rowq + 2 * D : seq + (size_t)jj * ld + 2 * D, va[u], vb[u]);
  }
  // ---- scores of the text keys 0..p (key p = the new token itself)
  // (rows clamped to p and loaded unconditionally, two passes unrolled: the first 128 keys' loads are all in flight at once)
  auto score = [&](int j) {
    const int jj = j < p ? j : p;
    const bool own = jj == p;
    const T* krow = own ? rowq + D : seq + (size_t)jj * ld + D;
    float x = dot_row(krow, jj, own || !cache_rot) * scale;
    if (biased_row && jj >= vs && jj < vs + F) x += g2;
    if (j <= p && part_id == 0) sc[j] = x;
  };
  score(w * 16 + slot);
  score(64 + w * 16 + slot);
  for (int j0 = 128; j0 <= p; j0 += 64) score(j0 + w * 16 + slot);
  // ---- scores of the adapter keys (wave 0; no RoPE, no mask)
  if (w == 0 && slot < A) {
    const float x = dot_row(arow + (size_t)slot * ld + D, 0, false) * scale;
    if (part_id == 0) sa[slot] = x;
  }
  __syncthreads();
  // ---- causal softmax over sc[0..p]
  float mx = -1e30f;
  for (int j = tid; j <= p; j += 256) mx = fmaxf(mx, sc[j]);
  mx = wave_max(mx);
  if (lane == 0) red[w] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int j = tid; j <= p; j += 256) {
    const float e = __expf(sc[j] - mx);
    sc[j] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  if (lane == 0) red[4 + w] = sum;
  // adapter softmax x tanh(gate1) (wave 0, A <= 16 values)
  if (w == 0) {
    const float x = lane < A ? sa[lane] : -1e30f;
    const float m = wave_max(x);
    const float e = lane < A ? __expf(x - m) : 0.f;
    const float s = wave_sum(e);
    if (lane < A) sa[lane] = e / s * tanhf(gate1[h]);
  }
  __syncthreads();
  const float inv = 1.f / ((red[4] + red[5]) + (red[6] + red[7]));
  // ---- values: lane = 2 head dims, wave w takes keys j = w, w+4, ...
  float o0 = 0.f, o1 = 0.f;
#pragma unroll
  for (int u = 0; u < 32; ++u) {
    const int j = w + 4 * u;
    const float pj = j <= p ? sc[j] * inv : 0.f;
    o0 += pj * va[u]; o1 += pj * vb[u];
  }
  for (int j0 = 128 + w; j0 <= p; j0 += 32) {       // longer contexts: eight keys' loads in flight per trip
    float xa[8], xb[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = j0 + 4 * u, jj = j < p ? j : p;
      load2((jj == p) ? rowq + 2 * D : seq + (size_t)jj * ld + 2 * D, xa[u], xb[u]);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = j0 + 4 * u;
      const float pj = j <= p ? sc[j] * inv : 0.f;
      o0 += pj * xa[u]; o1 += pj * xb[u];
    }
  }
  for (int a_ = w; a_ < A; a_ += 4) {
    float a, b;
    load2(arow + (size_t)a_ * ld + 2 * D, a, b);
    o0 += sa[a_] * a; o1 += sa[a_] * b;
  }
  part[w][d] = o0; part[w][d + 1] = o1;
  __syncthreads();
  if (w == 0) {
    const float r0 = (part[0][d] + part[1][d]) + (part[2][d] + part[3][d]);
    const float r1 = (part[0][d + 1] + part[1][d + 1]) + (part[2][d + 1] + part[3][d + 1]);
    T* op = o_row + (size_t)n * D + h * DH + d;
    if constexpr (WT && sizeof(T) == 2) {
      const unsigned bits = (unsigned)f32_to_bf16_bits(r0) | ((unsigned)f32_to_bf16_bits(r1) << 16);
      asm volatile("global_store_dword %0, %1, off sc1" ::"v"(op), "v"(bits) : "memory");
    } else {
      op[0] = from_f32<T>(r0); op[1] = from_f32<T>(r1);
    }
  } else if (w == 1) {
    // the new token's k (raw, or rotated where the cache keeps rotated keys) and v into the cache row of position p
    T* crow = cache + ((size_t)n * S + p) * ld + h * DH;
    float a = to_f32<T>(rowq[D + d]), b = to_f32<T>(rowq[D + d + 1]);
    if (cache_rot) {
      const float c = cs[(size_t)p * HP + lane], s = sn[(size_t)p * HP + lane];
      const float ra = a * c - b * s, rb = a * s + b * c;
      a = ra; b = rb;
    }
    crow[D + d] = from_f32<T>(a); crow[D + d + 1] = from_f32<T>(b);
    crow[2 * D + d] = rowq[2 * D + d]; crow[2 * D + d + 1] = rowq[2 * D + d + 1];
  }
}

}  // namespace fvqa_decode
