// One-query-row attention for the generation path (reference llama/model.py:428-470 re-runs the whole sequence for
// every new token; Attention.forward :87-128 is what each of those runs evaluates at the new row).
//
// Decode shape: one new token per sequence. For sequence n at position p = pos[n] and head h the kernel takes the RAW
// q / k / v projections of the new token (qkv_row), rotates q and k (RoPE at position p), attends over
//   * the A adapter keys / values (no RoPE, own softmax scaled by tanh(gate1[h]), llama/model.py:98-100,115) and
//   * the cached keys / values 0..p-1 of the sequence plus the new token's own (causal softmax, gate2[h] added on the
//     frame columns [vs, vs+F) when p >= vs+F, llama/model.py:116-120),
// writes the output row, and stores the new token's k and v into the cache row n*S + p of the layer's qkv arena —
// what the full-sequence kernel + two torch index ops did per token and layer before (13.5 us + glue for 256 (n, h)
// pairs against one streaming pass over 64 KiB of cached K/V per pair here).
// HBM-bound (K/V read once: 2 * (p+1+A) * 256 B per pair); no matrix cores: a 1-row product has nothing to feed them.
// cache_rot: 1 = the cache holds ROTATED k — the fp32 vector build, and the bf16 build when the QKV projection rotates q, k in
// its epilogue (fvqa_rope_in_gemm, the default); 0 = RAW k (bf16 build with FVQA_ROPE_IN_GEMM=0: the attention kernels
// rotate on the fly). The caller's rule: cache_rotated = !fvqa_attn_rope_fused(dtype) || fvqa_rope_in_gemm(dtype). The new
// token's rotated q and k are rounded to the storage type before use, as the prefill's are. One workgroup (4 waves) per (head, sequence): a quad of lanes per key in the score pass (64-byte
// contiguous reads per quad), one lane per pair of head dims in the value pass (256-byte rows per wave instruction).
#include "attn_decode_body.h"

namespace {
using namespace fvqa_decode;

template <typename T>
__global__ __launch_bounds__(256) void attn_decode_k(const T* __restrict__ qkv_row, T* __restrict__ cache,
                                                     T* __restrict__ o_row, const float* __restrict__ gate1,
                                                     const float* __restrict__ gate2,
                                                     const int32_t* __restrict__ vstart, const int64_t* __restrict__ pos,
                                                     const float* __restrict__ cs, const float* __restrict__ sn,
                                                     int n_seq, int S, int H, int A, int F, int cache_rot) {
  __shared__ float sc[SMAX];                       // text scores, then probabilities
  __shared__ float sa[16];                         // adapter scores, then probabilities (x tanh gate1)
  __shared__ float red[8];
  __shared__ float part[4][DH];
  attn_decode_body<T>(qkv_row, cache, o_row, gate1, gate2, vstart, pos, cs, sn, n_seq, S, H, A, F, cache_rot,
                      (int)blockIdx.x, (int)blockIdx.y, sc, sa, red, part);
}

}  // namespace

extern "C" int fvqa_attn_decode(const void* qkv_row, void* qkv_cache, void* o_row, const float* gate1,
                                const float* gate2, const int32_t* vstart, const int64_t* pos, const float* cos_t,
                                const float* sin_t, int n_seq, int seq_len, int n_heads, int head_dim, int adapter_len,
                                int max_feats, int cache_rotated, int dtype, void* stream) {
  if (!qkv_row || !qkv_cache || !o_row || !gate1 || !gate2 || !vstart || !pos || !cos_t || !sin_t) return FVQA_EINVAL;
  if (!fvqa_dtype_ok(dtype)) return FVQA_EINVAL;
  if (n_seq <= 0 || seq_len <= 0 || n_heads <= 0 || head_dim != DH || adapter_len < 0 || adapter_len > 16 || max_feats < 0)
    return FVQA_ESHAPE;
  if (seq_len > SMAX || n_heads > 65535 || n_seq > 65535) return FVQA_ESHAPE;
  if (((uintptr_t)qkv_row | (uintptr_t)qkv_cache | (uintptr_t)o_row) & 15) return FVQA_EALIGN;
  dim3 grid(n_heads, n_seq), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == FVQA_H16)
    hipLaunchKernelGGL(attn_decode_k<bf16_t>, grid, block, 0, st, (const bf16_t*)qkv_row, (bf16_t*)qkv_cache,
                       (bf16_t*)o_row, gate1, gate2, vstart, pos, cos_t, sin_t, n_seq, seq_len, n_heads, adapter_len,
                       max_feats, cache_rotated);
  else
    hipLaunchKernelGGL(attn_decode_k<float>, grid, block, 0, st, (const float*)qkv_row, (float*)qkv_cache,
                       (float*)o_row, gate1, gate2, vstart, pos, cos_t, sin_t, n_seq, seq_len, n_heads, adapter_len,
                       max_feats, cache_rotated);
  FVQA_CHECK_LAUNCH();
  return FVQA_OK;
}
