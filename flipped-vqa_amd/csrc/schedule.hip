// Native layer schedule: the per-layer kernel sequence of the Flipped-VQA step walked in C++.
//
// The Python host (fvqa/step.py) can issue every kernel itself (~700 ctypes calls per step, used
// for per-launch timing and debugging) or hand this file one plan per (n_seq, S) geometry and make
// two calls per step: fvqa_layers_fwd / fvqa_layers_bwd. Same kernels, same order, same results
// (tests/test_step_gpu.py checks bitwise equality of the two schedules); what changes is that the
// launch stream no longer depends on the speed of the Python interpreter.
//
// Sequence per layer (reference llama/model.py:184-187 with Attention :87-128, FeedForward :141-142):
//   fwd: [xn = RMSNorm(x)] -> QKV GEMM on the sequence rows + the A adapter rows through the decode-shape kernel
//        (their K/V projections, model.py:98-100) -> attention (RoPE inside) -> WO GEMM + residual (split-K reduced
//        inside the launch) -> RMSNorm -> W1|W3 GEMM with SwiGLU in its epilogue -> W2 GEMM + residual -> RMSNorm of the NEXT layer
//        (or the final norm)
//   bwd: W2^T GEMM with SwiGLU' epilogue -> W1|W3^T GEMM -> RMSNorm' (+residual grad) -> WO^T GEMM -> attention'
//        -> QKV^T GEMM on the sequence rows -> RMSNorm' (+residual grad); the adapter rows of dqkv go through the
//        decode-shape kernel straight into the fp32 adapter-query gradient (+=)
#include "common.h"
#include <cstdlib>

extern "C" size_t fvqa_layers_gemm_workspace(const fvqa_layer_plan* p);
extern "C" size_t fvqa_gemm_sk_workspace(void);

namespace {

inline char* at(void* base, size_t elems, size_t esize) { return (char*)base + elems * esize; }
inline const char* at(const void* base, size_t elems, size_t esize) { return (const char*)base + elems * esize; }

#define RUN(call)            \
  do {                       \
    int rc__ = (call);       \
    if (rc__) return rc__;   \
  } while (0)

// tuning switch (same-box A/B): FVQA_SWIGLU_AB=1 keeps a and b in the `ab` buffer and the full SwiGLU' arithmetic in the
// W2^T epilogue (the round-2 form); default: the (s, t) pair of FVQA_EPI_SWIGLU_FWD_ST / _BWD_ST
bool swiglu_st() {
  static const bool v = [] { const char* e = getenv("FVQA_SWIGLU_AB"); return !(e && e[0] == '1'); }();
  return v;
}

}  // namespace

// 1: the QKV projection rotates q and k in its epilogue (fvqa_gemm_nt_rope) and the attention kernels read finished operands
// (bf16 MFMA build; FVQA_ROPE_IN_GEMM=0 keeps the round-2 form: raw q, k in the arena, rotated inside every attention kernel)
extern "C" int fvqa_rope_in_gemm(int dtype) {
  static const bool off = [] { const char* e = getenv("FVQA_ROPE_IN_GEMM"); return e && e[0] == '0'; }();
  return fvqa_attn_rope_fused(dtype) != 0 && !off ? 1 : 0;
}

extern "C" int fvqa_swiglu_st(void) { return swiglu_st() ? 1 : 0; }

// 1: the adapter K/V rows of layer i+1 are the rider of layer i's W1|W3 launch (layer 0's: a launch of their own before the
// walk). They depend on parameters only, and that launch — 1.8 rounds of whole tiles in the 4-wave kernel — has workgroups
// without a tile in its last round, while the QKV launch fills the chip with whole tiles (gemm4w.hip).
extern "C" int fvqa_kv_rider_ahead(int dtype) {
  static const bool off = [] { const char* e = getenv("FVQA_KV_AHEAD"); return e && e[0] == '0'; }();
  return fvqa_rope_in_gemm(dtype) != 0 && swiglu_st() && !off ? 1 : 0;
}

namespace {

int check_plan_dims_only(const fvqa_layer_plan* p) {
  if (!p) return FVQA_EINVAL;
  if (!fvqa_dtype_ok(p->dtype)) return FVQA_EINVAL;
  if (p->n_layers <= 0 || p->n_seq <= 0 || p->seq_len <= 0 || p->dim <= 0 || p->hidden <= 0) return FVQA_ESHAPE;
  if (p->n_heads * p->head_dim != p->dim) return FVQA_ESHAPE;
  return FVQA_OK;
}

int check_plan(const fvqa_layer_plan* p) {
  int rc = check_plan_dims_only(p);
  if (rc) return rc;
  if (p->gemm_ws_bytes < fvqa_layers_gemm_workspace(p)) return FVQA_EALIGN;
  if (!p->wqkv || !p->wo || !p->w13 || !p->w2 || !p->wqkv_t || !p->wo_t || !p->w13_t || !p->w2_t || !p->an ||
      !p->fn || !p->gate1 || !p->gate2 || !p->adapter || !p->norm_w || !p->xs || !p->rstd1 || !p->rstd2 || !p->qkv ||
      !p->o || !p->lse_a || !p->lse_t || !p->h || !p->ab || !p->xn || !p->hn || !p->z || !p->xnf || !p->rstdN ||
      !p->cos_t || !p->sin_t || !p->vstart || !p->gemm_ws)
    return FVQA_EINVAL;
  if ((uintptr_t)p->gemm_ws & 255) return FVQA_EALIGN;
  const auto& t = p->tail;
  if (t.rows < 0) return FVQA_ESHAPE;
  if (t.rows > 0) {                                   // the last layer's post-attention half on the gathered rows
    if (t.gather.n < 1 || t.gather.n > 3 || t.gather.n != t.scatter.n || t.gather.stream_rows != t.scatter.stream_rows ||
        t.gather.n * t.gather.stream_rows != p->n_seq * p->seq_len || t.gather.off[t.gather.n] != t.rows ||
        t.scatter.off[t.scatter.n] != t.rows)
      return FVQA_ESHAPE;
    if (!t.og || !t.xg || !t.h || !t.hn || !t.ab || !t.z || !t.xl || !t.xnf || !t.rstd2 || !t.rstdN) return FVQA_EINVAL;
  }
  return FVQA_OK;
}

// adapter_query.grad rows of walked layer i (fp32, +=): the A rows under the sequence rows of dqkv hold
// [0, dK_a, dV_a] (summed over the batch by the attention backward), so only the K/V column blocks take part:
// d_adapter[i] += dqkv[R:, D:3D] · Wqkv^T[:, D:3D]^T
fvqa_sk_rider adapter_grad_rider(const fvqa_layer_plan* p, int i) {
  const int D = p->dim, A = p->adapter_len, R = p->n_seq * p->seq_len;
  const size_t es = fvqa_dtype_size(p->dtype);
  return fvqa_sk_rider{at(p->dqkv, (size_t)R * 3 * D + D, es), at(p->wqkv_t[i], (size_t)D, es),
                       p->d_adapter + (size_t)i * A * D, A, D, 2 * D, 3 * D, 3 * D, D, 1};
}

}  // namespace

extern "C" size_t fvqa_layers_gemm_workspace(const fvqa_layer_plan* p) {
  if (check_plan_dims_only(p)) return 0;
  const int dt = p->dtype, D = p->dim, Hf = p->hidden, R = p->n_seq * p->seq_len;
  size_t need = 0;
  auto upd = [&](size_t v) { if (v > need) need = v; };
  upd(fvqa_gemm_workspace(R, 3 * D, D, dt));        // QKV
  upd(fvqa_gemm_workspace(R, D, D, dt));            // WO, WO^T
  upd(fvqa_gemm_sk_workspace());                    // W1|W3 (+SwiGLU) and W2^T (SwiGLU'): persistent kernel only
  upd(fvqa_gemm_workspace(R, D, Hf, dt));           // W2
  upd(fvqa_gemm_workspace(R, D, 2 * Hf, dt));       // W1|W3^T
  upd(fvqa_gemm_workspace(R, D, 3 * D, dt));        // QKV^T
  return need;
}

extern "C" int fvqa_layers_fwd(const fvqa_layer_plan* p, void* stream) {
  RUN(check_plan(p));
  const int dt = p->dtype, L = p->n_layers, D = p->dim, Hf = p->hidden, H = p->n_heads, Dh = p->head_dim;
  const int A = p->adapter_len, S = p->seq_len, n_seq = p->n_seq;
  const int R = n_seq * S, Ra = R + A;
  const size_t es = fvqa_dtype_size(dt);
  const bool fused_rope = fvqa_attn_rope_fused(dt) != 0;
  const bool rope_gemm = fvqa_rope_in_gemm(dt) != 0;
  const bool kv_ahead = fvqa_kv_rider_ahead(dt) != 0;
  if (!p->adapter_c) return FVQA_EINVAL;
  // the adapter prompts of all walked layers in storage dtype (model.py:339 `.half()`), one launch
  RUN(fvqa_cast_rows(p->adapter, p->adapter_c, L * A, D, dt, stream));
  RUN(fvqa_rmsnorm_fwd(p->xs, p->an[0], p->xn, p->rstd1, R, D, p->eps, dt, stream));
  // the A adapter rows under the sequence rows of layer i's qkv get their K and V projections (model.py:98-100; their q block
  // is never read) as a rider: beside layer i's QKV GEMM, or (kv_ahead) beside layer i-1's W1|W3 GEMM
  auto kv_rider = [&](int i) {
    void* qkv_i = at(p->qkv, (size_t)i * Ra * 3 * D, es);
    return fvqa_sk_rider{at(p->adapter_c, (size_t)i * A * D, es), at(p->wqkv[i], (size_t)D * D, es),
                         at(qkv_i, (size_t)R * 3 * D + D, es), A, 2 * D, D, D, D, 3 * D, 0};
  };
  if (kv_ahead) {
    const fvqa_sk_rider kv0 = kv_rider(0);
    RUN(fvqa_gemm_nt(kv0.A, kv0.B, kv0.C, nullptr, nullptr, kv0.M, kv0.N, kv0.K, kv0.lda, kv0.ldb, kv0.ldc, kv0.M, dt, dt,
                     FVQA_EPI_NONE, 0, nullptr, 0, stream));
  }
  for (int i = 0; i < L; ++i) {
    const void* x = at(p->xs, (size_t)i * R * D, es);
    void* x_next = at(p->xs, (size_t)(i + 1) * R * D, es);
    void* qkv = at(p->qkv, (size_t)i * Ra * 3 * D, es);
    void* o = at(p->o, (size_t)i * R * D, es);
    void* h = at(p->h, (size_t)i * R * D, es);
    void* ab = at(p->ab, (size_t)i * R * 2 * Hf, es);
    float* lse_a = p->lse_a + (size_t)i * n_seq * H * S;
    float* lse_t = p->lse_t + (size_t)i * n_seq * H * S;
    const fvqa_sk_rider kv = kv_rider(i);
    if (rope_gemm) {                                   // bf16 MFMA build: q, k are rotated where the projection produces them
      const fvqa_sk_rope rp = {p->cos_t, p->sin_t, S, Dh, 2 * D};
      RUN(fvqa_gemm_nt_rope(p->xn, p->wqkv[i], qkv, R, 3 * D, D, D, D, 3 * D, &rp, kv_ahead ? nullptr : &kv, p->gemm_ws,
                            p->gemm_ws_bytes, stream));
      RUN(fvqa_attn_fwd(qkv, o, lse_a, lse_t, p->gate1[i], p->gate2[i], p->vstart, nullptr, nullptr, n_seq, S, H, Dh, A,
                        p->max_feats, dt, stream));
    } else {
    RUN(fvqa_gemm_nt_rider(p->xn, p->wqkv[i], qkv, nullptr, R, 3 * D, D, D, D, 3 * D, dt, dt, FVQA_EPI_NONE, &kv,
                           p->gemm_ws, p->gemm_ws_bytes, stream));
    if (fused_rope) {                                  // (FVQA_ROPE_IN_GEMM=0) q,k stay raw and are rotated inside
      RUN(fvqa_attn_fwd(qkv, o, lse_a, lse_t, p->gate1[i], p->gate2[i], p->vstart, p->cos_t, p->sin_t, n_seq, S, H, Dh,
                        A, p->max_feats, dt, stream));
    } else {
      RUN(fvqa_rope_qk(qkv, p->cos_t, p->sin_t, n_seq, S, H, Dh, 0, dt, stream));
      RUN(fvqa_attn_fwd(qkv, o, lse_a, lse_t, p->gate1[i], p->gate2[i], p->vstart, nullptr, nullptr, n_seq, S, H, Dh, A,
                        p->max_feats, dt, stream));
    }
    }
    if (i == L - 1 && p->tail.rows > 0) {
      // The LAST layer's post-attention half and the final norm on the rows a head reads (include/fvqa.h "tail rows"): nothing
      // else consumes this layer's output, so WO + residual, the FFN and the norm run on the gathered rows of o and x only.
      const auto& t = p->tail;
      const int M = t.rows;
      RUN(fvqa_gather_rows(o, t.og, &t.gather, D, dt, stream));
      RUN(fvqa_gather_rows(x, t.xg, &t.gather, D, dt, stream));
      RUN(fvqa_gemm_nt(t.og, p->wo[i], t.h, t.xg, nullptr, M, D, D, D, D, D, M, dt, dt, FVQA_EPI_RESIDUAL, 0, p->gemm_ws,
                       p->gemm_ws_bytes, stream));
      RUN(fvqa_rmsnorm_fwd(t.h, p->fn[i], t.hn, t.rstd2, M, D, p->eps, dt, stream));
      if (swiglu_st())
        RUN(fvqa_gemm_nt_swiglu_fwd_st(t.hn, p->w13[i], t.ab, t.z, M, Hf, D, D, D, dt, p->gemm_ws, p->gemm_ws_bytes, stream));
      else
        RUN(fvqa_gemm_nt_swiglu_fwd(t.hn, p->w13[i], t.ab, t.z, M, Hf, D, D, D, dt, p->gemm_ws, p->gemm_ws_bytes, stream));
      RUN(fvqa_gemm_nt(t.z, p->w2[i], t.xl, t.h, nullptr, M, D, Hf, Hf, Hf, D, M, dt, dt, FVQA_EPI_RESIDUAL, 0, p->gemm_ws,
                       p->gemm_ws_bytes, stream));
      RUN(fvqa_rmsnorm_fwd(t.xl, p->norm_w, t.xnf, t.rstdN, M, D, p->eps, dt, stream));
      break;
    }
    // h = x + o·Wo^T (model.py:185), hn = RMSNorm(h)·w
    RUN(fvqa_gemm_nt(o, p->wo[i], h, x, nullptr, R, D, D, D, D, D, R, dt, dt, FVQA_EPI_RESIDUAL, 0, p->gemm_ws,
                     p->gemm_ws_bytes, stream));
    RUN(fvqa_rmsnorm_fwd(h, p->fn[i], p->hn, p->rstd2 + (size_t)i * R, R, D, p->eps, dt, stream));
    // z = silu(a)*b with (a | b) = hn·(W1|W3)^T (model.py:142) in one launch; `ab` keeps the backward's factors (s, t)
    if (kv_ahead && i + 1 < L) {
      const fvqa_sk_rider kvn = kv_rider(i + 1);
      RUN(fvqa_gemm_nt_swiglu_fwd_st_rider(p->hn, p->w13[i], ab, p->z, R, Hf, D, D, D, dt, &kvn, p->gemm_ws, p->gemm_ws_bytes,
                                           stream));
    } else if (swiglu_st())
      RUN(fvqa_gemm_nt_swiglu_fwd_st(p->hn, p->w13[i], ab, p->z, R, Hf, D, D, D, dt, p->gemm_ws, p->gemm_ws_bytes, stream));
    else
      RUN(fvqa_gemm_nt_swiglu_fwd(p->hn, p->w13[i], ab, p->z, R, Hf, D, D, D, dt, p->gemm_ws, p->gemm_ws_bytes, stream));
    // x_next = h + z·W2^T (model.py:186), then the next layer's attention norm (or the final norm)
    RUN(fvqa_gemm_nt(p->z, p->w2[i], x_next, h, nullptr, R, D, Hf, Hf, Hf, D, R, dt, dt, FVQA_EPI_RESIDUAL, 0,
                     p->gemm_ws, p->gemm_ws_bytes, stream));
    if (i + 1 < L)
      RUN(fvqa_rmsnorm_fwd(x_next, p->an[i + 1], p->xn, p->rstd1 + (size_t)(i + 1) * R, R, D, p->eps, dt, stream));
    else
      RUN(fvqa_rmsnorm_fwd(x_next, p->norm_w, p->xnf, p->rstdN, R, D, p->eps, dt, stream));
  }
  return FVQA_OK;
}

// dxnf: gradient w.r.t. the final-norm output (R, D). On return *d_x0 points at the gradient w.r.t.
// the layer-0 input (one of the plan's two ping-pong buffers).
extern "C" int fvqa_layers_bwd(const fvqa_layer_plan* p, const void* dxnf, void** d_x0, void* stream) {
  RUN(check_plan(p));
  if (!dxnf || !d_x0 || !p->dcur || !p->dnxt || !p->dz || !p->dab || !p->dh || !p->d_o || !p->dqkv || !p->attn_ws ||
      !p->dgate1 || !p->dgate2 || !p->d_adapter)
    return FVQA_EINVAL;
  const int dt = p->dtype, L = p->n_layers, D = p->dim, Hf = p->hidden, H = p->n_heads, Dh = p->head_dim;
  const int A = p->adapter_len, S = p->seq_len, n_seq = p->n_seq;
  const int R = n_seq * S, Ra = R + A;
  const size_t es = fvqa_dtype_size(dt);
  void* cur = p->dcur;
  void* nxt = p->dnxt;
  void* t = p->dz;                                     // (R, D) scratch for the GEMM outputs that feed the norm backward
  const bool fused_rope = fvqa_attn_rope_fused(dt) != 0;
  const bool rope_gemm = fvqa_rope_in_gemm(dt) != 0;
  const int epi_sw = swiglu_st() ? FVQA_EPI_SWIGLU_BWD_ST : FVQA_EPI_SWIGLU_BWD;
  const bool tail = p->tail.rows > 0;
  if (tail) {
    const auto& t = p->tail;
    if (!t.dcur || !t.dab || !t.dt || !t.dh || !t.d_o) return FVQA_EINVAL;
    RUN(fvqa_rmsnorm_bwd(dxnf, t.xl, p->norm_w, t.rstdN, nullptr, t.dcur, t.rows, D, dt, stream));      // dxnf: (rows, D)
  } else {
    RUN(fvqa_rmsnorm_bwd(dxnf, at(p->xs, (size_t)L * R * D, es), p->norm_w, p->rstdN, nullptr, cur, R, D, dt, stream));
  }
  for (int i = L - 1; i >= 0; --i) {
    const void* x = at(p->xs, (size_t)i * R * D, es);
    const void* qkv = at(p->qkv, (size_t)i * Ra * 3 * D, es);
    const void* o = at(p->o, (size_t)i * R * D, es);
    const void* h = at(p->h, (size_t)i * R * D, es);
    const void* ab = at(p->ab, (size_t)i * R * 2 * Hf, es);
    const float* lse_a = p->lse_a + (size_t)i * n_seq * H * S;
    const float* lse_t = p->lse_t + (size_t)i * n_seq * H * S;
    // dz = cur·W2 never reaches HBM: the SwiGLU backward is this GEMM's epilogue. On its idle CUs: the adapter-query
    // gradient rows of the layer walked just before (dqkv still holds that layer's [0, dK_a, dV_a] rows)
    if (i == L - 1 && tail) {
      // the compact half of the last layer, backwards: FFN', norm', WO^T on the tail rows; their d_o and dh rows are then
      // scattered under the zero rows of everything else and the walk continues dense (attention' needs every key's row)
      const auto& tr = p->tail;
      const int M = tr.rows;
      RUN(fvqa_gemm_nt(tr.dcur, p->w2_t[i], tr.dab, tr.ab, nullptr, M, Hf, D, D, D, 2 * Hf, M, dt, dt, epi_sw, 0,
                       p->gemm_ws, p->gemm_ws_bytes, stream));
      RUN(fvqa_gemm_nt(tr.dab, p->w13_t[i], tr.dt, nullptr, nullptr, M, D, 2 * Hf, 2 * Hf, 2 * Hf, D, M, dt, dt,
                       FVQA_EPI_NONE, 0, p->gemm_ws, p->gemm_ws_bytes, stream));
      RUN(fvqa_rmsnorm_bwd(tr.dt, tr.h, p->fn[i], tr.rstd2, tr.dcur, tr.dh, M, D, dt, stream));
      RUN(fvqa_gemm_nt(tr.dh, p->wo_t[i], tr.d_o, nullptr, nullptr, M, D, D, D, D, D, M, dt, dt, FVQA_EPI_NONE, 0,
                       p->gemm_ws, p->gemm_ws_bytes, stream));
      RUN(fvqa_scatter_rows(tr.d_o, p->d_o, &tr.scatter, D, dt, stream));
      RUN(fvqa_scatter_rows(tr.dh, p->dh, &tr.scatter, D, dt, stream));
    } else {
    if (i + 1 < L) {
      const fvqa_sk_rider ga = adapter_grad_rider(p, i + 1);
      RUN(fvqa_gemm_nt_rider(cur, p->w2_t[i], p->dab, ab, R, Hf, D, D, D, 2 * Hf, dt, dt, epi_sw, &ga,
                             p->gemm_ws, p->gemm_ws_bytes, stream));
    } else {
      RUN(fvqa_gemm_nt(cur, p->w2_t[i], p->dab, ab, nullptr, R, Hf, D, D, D, 2 * Hf, R, dt, dt, epi_sw, 0,
                       p->gemm_ws, p->gemm_ws_bytes, stream));
    }
    RUN(fvqa_gemm_nt(p->dab, p->w13_t[i], t, nullptr, nullptr, R, D, 2 * Hf, 2 * Hf, 2 * Hf, D, R, dt, dt,
                     FVQA_EPI_NONE, 0, p->gemm_ws, p->gemm_ws_bytes, stream));
    RUN(fvqa_rmsnorm_bwd(t, h, p->fn[i], p->rstd2 + (size_t)i * R, cur, p->dh, R, D, dt, stream));
    RUN(fvqa_gemm_nt(p->dh, p->wo_t[i], p->d_o, nullptr, nullptr, R, D, D, D, D, D, R, dt, dt, FVQA_EPI_NONE, 0,
                     p->gemm_ws, p->gemm_ws_bytes, stream));
    }
    if (rope_gemm) {
      RUN(fvqa_attn_bwd_rotated(p->d_o, qkv, o, lse_a, lse_t, p->gate1[i], p->gate2[i], p->vstart, p->cos_t, p->sin_t,
                                p->dqkv, p->dgate1[i], p->dgate2[i], p->attn_ws, p->attn_ws_bytes, n_seq, S, H, Dh, A,
                                p->max_feats, dt, stream));
    } else if (fused_rope) {
      RUN(fvqa_attn_bwd(p->d_o, qkv, o, lse_a, lse_t, p->gate1[i], p->gate2[i], p->vstart, p->cos_t, p->sin_t, p->dqkv,
                        p->dgate1[i], p->dgate2[i], p->attn_ws, p->attn_ws_bytes, n_seq, S, H, Dh, A, p->max_feats, dt,
                        stream));
    } else {
      RUN(fvqa_attn_bwd(p->d_o, qkv, o, lse_a, lse_t, p->gate1[i], p->gate2[i], p->vstart, nullptr, nullptr, p->dqkv,
                        p->dgate1[i], p->dgate2[i], p->attn_ws, p->attn_ws_bytes, n_seq, S, H, Dh, A, p->max_feats, dt,
                        stream));
      RUN(fvqa_rope_qk(p->dqkv, p->cos_t, p->sin_t, n_seq, S, H, Dh, 1, dt, stream));
    }
    RUN(fvqa_gemm_nt(p->dqkv, p->wqkv_t[i], t, nullptr, nullptr, R, D, 3 * D, 3 * D, 3 * D, D, R, dt, dt,
                     FVQA_EPI_NONE, 0, p->gemm_ws, p->gemm_ws_bytes, stream));
    RUN(fvqa_rmsnorm_bwd(t, x, p->an[i], p->rstd1 + (size_t)i * R, p->dh, nxt, R, D, dt, stream));
    void* sw = cur; cur = nxt; nxt = sw;
  }
  {                                                    // layer 0's adapter-query gradient rows: nothing left to ride on
    const fvqa_sk_rider ga = adapter_grad_rider(p, 0);
    RUN(fvqa_gemm_nt(ga.A, ga.B, nullptr, nullptr, (float*)ga.C, ga.M, ga.N, ga.K, ga.lda, ga.ldb, ga.ldc, 0, dt, dt,
                     FVQA_EPI_NONE, 0, nullptr, 0, stream));
  }
  *d_x0 = cur;
  return FVQA_OK;
}
