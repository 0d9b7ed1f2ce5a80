/*
 * fvqa.h — C ABI of libfvqa_hip.so: the MI355X (gfx950) kernels behind the Flipped-VQA
 * training hot path.
 *
 * The reference (inesriahi/Flipped-VQA) is pure Python/PyTorch: its hot path has no FFI of
 * its own — every entry below replaces an *implicit* torch op group of
 * llama/model.py:31-365, engine.py:10-56 and util/misc.py:253-294 (file:line cited per
 * entry, relative to the reference root). The Python host (flipped-vqa_amd/fvqa/_lib.py)
 * binds them with ctypes; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch caching allocator);
 *     the library never allocates, frees, synchronises or throws;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - `dtype` selects the storage type of activations / frozen weights:
 *       FVQA_F32  — exact-fp32 validation build of every kernel (fp32-in MFMA),
 *       FVQA_BF16 — production build (bf16 storage, fp32 accumulate);
 *       FVQA_F16  — the same with IEEE fp16 storage (libfvqa_hip_f16.so; wherever an entry below says "bf16" / FVQA_BF16 for
 *                   the 16-bit build, that library reads it as fp16 / FVQA_F16);
 *     trainable parameters, their gradients, softmax statistics and losses are always fp32;
 *   - matrices are row-major; `rows` = sequences*seq_len flattened (n*S + s);
 *   - return value: 0 (FVQA_OK) or a negative FVQA_E* code / -(1000+hipError_t).
 */
#ifndef FVQA_H
#define FVQA_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FVQA_F32 0
#define FVQA_BF16 1
#define FVQA_F16 2 /* IEEE fp16 storage — the reference's own (llama_vqa.py:63 builds the model under torch.cuda.HalfTensor). Served by
                      libfvqa_hip_f16.so: the same sources compiled with the 16-bit storage type switched (csrc/common.h), same
                      entry points; libfvqa_hip.so serves FVQA_BF16. Each library rejects the other's 16-bit code (FVQA_EINVAL),
                      both serve FVQA_F32. The host (fvqa/_lib.py) binds the library of the model's storage dtype. */

#define FVQA_OK 0
#define FVQA_EINVAL (-1) /* null pointer / bad enum            */
#define FVQA_ESHAPE (-2) /* dimension not supported by kernels */
#define FVQA_EALIGN (-3) /* pointer or leading dim misaligned  */

/* GEMM epilogue selector */
#define FVQA_EPI_NONE 0
#define FVQA_EPI_RESIDUAL 1 /* C = acc + R                                  */
#define FVQA_EPI_SWIGLU_BWD 3 /* acc = dz (M,N): R = ab (M,2N), C = dab (M,2N) <- d/d(a,b) of silu(a)*b
                                (llama/model.py:142 backward), both in the AB16 layout below; ldc must be 2N */
#define FVQA_EPI_SWIGLU_FWD 4 /* fvqa_gemm_nt_swiglu_fwd only: C = ab (M,N) AND z = silu(a)*b (M,N/2) */
#define FVQA_EPI_SWIGLU_FWD_ST 5 /* fvqa_gemm_nt_swiglu_fwd_st only: as 4, but C receives, in the a and b slots of the AB16
                                   layout, s = silu(a) = dz/db and t = b*sigma(a)*(1 + a*(1 - sigma(a))) = dz/da: what
                                   the backward of llama/model.py:142 multiplies by */
#define FVQA_EPI_SWIGLU_BWD_ST 6 /* as 3 with R = that (s, t) buffer: C = dab = (acc * t, acc * s) */
#define FVQA_EPI_ROPE 7          /* fvqa_gemm_nt_rope only: RoPE of the q | k columns of the QKV projection in the epilogue */
/* AB16: the layout of every (rows, 2*hidden) buffer that holds the W1 and W3 projections (or their gradients) side by
 * side: column 32k + c is a[16k + c], column 32k + 16 + c is b[16k + c] (c < 16), i.e. the rows of W1 and W3 are
 * interleaved in blocks of 16 in the packed W1|W3 matrix. One MFMA wave of the W1|W3 GEMM then holds a and b of the
 * same hidden unit in the same lane, and SwiGLU becomes that GEMM's epilogue. hidden % 16 == 0. */

int fvqa_version(void);      /* ABI version, bumped on any signature change */
/* sha256 (hex) of the kernel sources (csrc/ *.hip, *.h, include/fvqa.h) this library was compiled from, "unknown" for a build
 * outside fvqa/build.py. The host binding compares it with the sources next to the library and refuses a stale binary. */
const char* fvqa_source_hash(void);
const char* fvqa_arch(void); /* "gfx950"                                     */

/* ---- dense projections: F.linear with frozen weights ---------------------------------
 * C[M,N] = A[M,K] · B[N,K]^T (+ R[M,N]).  Replaces torch F.linear at llama/model.py:89
 * (wq/wk/wv), :99-100 (adapter k/v), :127-128 (wo), :142 (w1,w3,w2), :348 (output) and the
 * autograd dX = dY·W of each (W frozen ⇒ no dW GEMM; the host keeps a transposed copy of
 * every frozen weight so dX is the same NT form).  A and B have `dtype`; C has `out_dtype`
 * (FVQA_F32 for LM-head logits). Rows m >= m_split (if tail != NULL) are written as fp32 to
 * tail[(m - m_split)*N + n] (ACCUMULATED, +=) instead of C (adapter-query gradient rows).
 * Needs K % 64 == 0 (bf16) / K % 32 == 0 (fp32), 16-byte aligned rows.
 * variant 0 picks the kernel: the persistent 256x256-tile LDS-DMA ring kernel (M >= 192, N >= 256, N % 8 == 0, no
 * tail rows, workspace given; also 16 < M < 192 against >= 16 M weights: the projections of the tail rows, see
 * fvqa_row_segs), the weight-streaming kernel of the generation path for bf16 M <= 16, K % 256 == 0 (one
 * new token per sequence, llama/model.py:439-447 run row-wise; with C == NULL, m_split == 0 it accumulates every row
 * into `tail`: the adapter-query gradient rows), else the 128x128-tile kernel. Other variant codes force a kernel
 * (1 / 2 = 128x128 register- / DMA-staged, 12 = weight-streaming, 13 = persistent; tests, tuning). */
int fvqa_gemm_nt(const void* A, const void* B, void* C, const void* R, float* tail,
                 int M, int N, int K, int lda, int ldb, int ldc, int m_split,
                 int dtype, int out_dtype, int epilogue, int variant,
                 void* workspace, size_t workspace_bytes, void* stream);
/* ab[M, 2*hidden] = A[M,K] · B13[2*hidden,K]^T and z[M, hidden] = silu(a) * b in ONE launch of the persistent
 * kernel (llama/model.py:142: w1(x), w3(x) and their product; `ab` is what the backward needs). B13 is W1|W3 with rows
 * interleaved in blocks of 16 (AB16). Needs the fvqa_gemm_sk_workspace() workspace. */
int fvqa_gemm_nt_swiglu_fwd(const void* A, const void* B13, void* ab, void* z, int M, int hidden, int K, int lda,
                            int ldb, int dtype, void* workspace, size_t workspace_bytes, void* stream);
/* The training step's form: same product and the same z, but `st` (M, 2*hidden, AB16) receives the two factors of the
 * SwiGLU backward (FVQA_EPI_SWIGLU_FWD_ST) instead of a and b — nothing else on the training path reads a or b, and the
 * dH·W2^T GEMM's epilogue (FVQA_EPI_SWIGLU_BWD_ST) then is two multiplies per element instead of an exponential, a
 * reciprocal and a dozen operations (llama/model.py:142 and its autograd). */
int fvqa_gemm_nt_swiglu_fwd_st(const void* A, const void* B13, void* st, void* z, int M, int hidden, int K, int lda,
                               int ldb, int dtype, void* workspace, size_t workspace_bytes, void* stream);
/* fvqa_gemm_workspace: bytes of `workspace` the kernel variant 0 picks for the problem needs (0: none).
 * Its FIRST 4096 BYTES are the epoch flags of the persistent kernel (csrc/gemm_sk.hip): the caller zeroes them ONCE
 * after allocating the buffer (256-byte aligned); no call ever needs them reset. One workspace serves one stream at
 * a time. The first 64-bit word is an ERROR word: a workgroup whose bounded wait (~1 s) for a partner of a split tile
 * ran out sets it to non-zero and lets the grid drain — the outputs of that launch are then invalid. It is sticky; hand
 * its address to fvqa_grad_unscale_norm (the optimizer step of such a step is then skipped on the device) and read it
 * back at a convenient point (fvqa.ops.gemm_error / StepEngine.check_gemm_error; engine.train_one_epoch every iteration).
 * The bounded wait presumes that every workgroup of the grid is resident: the kernel asks for all 160 KiB of a CU's LDS and
 * launches at most one workgroup per CU of the CURRENT device, so the device must be this process's alone while a launch
 * runs (another process's persistent grid, or any kernel holding LDS on many CUs, can displace a partner: a 1 s stall and
 * the error word, not a hang). Launches into a stream under capture are refused (FVQA_EINVAL): the epoch number is a
 * host-side argument, and a replayed graph would reuse it. */
size_t fvqa_gemm_workspace(int M, int N, int K, int dtype);
/* The persistent kernel (variant 0 for M >= 192, N >= 256, N % 8 == 0, no tail rows; variant 13 forces it): a grid of
 * at most one workgroup per CU walks whole 256x256 output tiles, or — outputs with few tiles — one K range of a tile
 * each; a tile shared by several workgroups is reduced inside the launch (fixed order: bitwise repeatable) and stored
 * once with the epilogue.
 * fvqa_gemm_sk_workspace: 4096 flag bytes + one 256 KiB fp32 partial-tile slab per workgroup.
 * fvqa_gemm_sk_describe (host only, no GPU touched): the partition for a problem on n_cu compute units —
 * plan_out[12] = {tm, tn, wide stages per tile, stages per granule, granules per tile, team size, m groups, teams,
 * rounds of whole tiles, tiles of the last (split) round, pieces per tile there, team distance between the pieces of a
 * tile}; for team >= 0 also its segments, 5 ints each
 * {tile, k0, k1, pieces n, piece c} (up to max_segs written); returns the team's segment count. */
size_t fvqa_gemm_sk_workspace(void);

/* A second, independent product of at most 16 rows, C2[M2,N2] = A2[M2,K2] · B2[N2,K2]^T (operands in the main
 * problem's `dtype`), that rides on the compute units a projection leaves idle: the 10 adapter rows of
 * llama/model.py:98-100 (their K/V projections beside the QKV GEMM; their gradient rows into adapter_query.grad
 * beside the W2^T GEMM of the next layer walked). accumulate_f32 == 0: C2 has the storage dtype and is overwritten;
 * != 0: C2 is fp32 and the product is ADDED (ldc2 must then equal N2). */
typedef struct fvqa_sk_rider {
  const void* A; const void* B; void* C;
  int32_t M, N, K, lda, ldb, ldc;
  int32_t accumulate_f32;
} fvqa_sk_rider;
/* C = A·B^T (+ epilogue) exactly as fvqa_gemm_nt variant 0 (no tail rows), plus `rider`: inside the same launch when the
 * persistent kernel takes the main problem and leaves >= 16 CUs idle (bf16, M2 <= 16, K2 % 256 == 0), else as its
 * own launch right after it. The arithmetic of the rider is the same in both cases (bitwise-equal results). */
int fvqa_gemm_nt_rider(const void* A, const void* B, void* C, const void* R, int M, int N, int K, int lda, int ldb,
                       int ldc, int dtype, int out_dtype, int epilogue, const fvqa_sk_rider* rider,
                       void* workspace, size_t workspace_bytes, void* stream);
/* fvqa_gemm_nt_swiglu_fwd_st plus a rider (may be NULL), as fvqa_gemm_nt_rider: inside the launch when the kernel that takes
 * the main problem has workgroups to spare in its last round, else as its own launch right after. The step puts the NEXT
 * layer's adapter K/V rows here (they depend on parameters only): the W1|W3 launch of 1.8 rounds of 192-column tiles has 52
 * workgroups without a tile in its second round, the QKV launch of 256 such tiles none. */
int fvqa_gemm_nt_swiglu_fwd_st_rider(const void* A, const void* B13, void* st, void* z, int M, int hidden, int K, int lda,
                                     int ldb, int dtype, const fvqa_sk_rider* rider, void* workspace,
                                     size_t workspace_bytes, void* stream);
/* The QKV projection with RoPE where it is produced (bf16; reference llama/model.py:61-67 applied to wq(x), wk(x) :89-93):
 * C[M, N] = A · B^T with columns [0, rope->cols) — the q | k heads, rope->cols = 2 * n_heads * head_dim — rotated by the
 * tables (rows of head_dim / 2 floats per position, as fvqa_rope_qk takes them) of position (row % seq_len): value rounded to
 * bf16, rotated in fp32, rounded again, i.e. what fvqa_gemm_nt followed by fvqa_rope_qk leaves (which is what this entry runs
 * for shapes the persistent kernel does not take; ldc must then be 3 * n_heads * head_dim). M % seq_len == 0. The attention
 * kernels then read finished operands (cos_t == NULL) and no key tile is rotated once per query block; the backward takes
 * fvqa_attn_bwd_rotated. `rider` may be NULL. */
typedef struct fvqa_sk_rope {
  const float* cos_t; const float* sin_t;
  int32_t seq_len, head_dim, cols;
} fvqa_sk_rope;
int fvqa_gemm_nt_rope(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                      const fvqa_sk_rope* rope, const fvqa_sk_rider* rider, void* workspace, size_t workspace_bytes,
                      void* stream);
int fvqa_gemm_sk_describe(int M, int N, int K, int dtype, int n_cu, int32_t* plan_out, int team,
                          int32_t* segs_out, int max_segs);
/* The whole-tile bf16 projections (outputs wide enough to fill the chip: q|k|v, w1|w3, dH·W2^T, LM head) run on tiles of
 * 256 rows x 16*nbt columns, nbt in {16, 14, 13, 12, 11} chosen per problem by a cost model so that the tile count lands on the
 * CU count. fvqa_gemm4w_choose (host only): the nbt the model picks for a problem on n_cu compute units, 0 when the problem is
 * not that kernel's (fp32 build, K % 64, M < 192, N < 256, an epilogue it does not have). `rider` may be NULL.
 * fvqa_gemm4w_force: tests / tuning — every following projection call OF THE CALLING HOST THREAD uses tiles of 16*nbt columns
 * where the kernel is eligible at all (odd nbt never with FVQA_EPI_SWIGLU_FWD_ST: (a, b) column blocks pair up); 0 restores the
 * cost model. Returns the previous setting, FVQA_EINVAL for a width that does not exist. */
int fvqa_gemm4w_choose(int M, int N, int K, int dtype, int out_dtype, int epilogue, const fvqa_sk_rider* rider, int n_cu);
int fvqa_gemm4w_force(int nbt);
/* Measurement probe (bench.py roofline; no reference counterpart): while enabled, launches of the persistent
 * 256x256 GEMM kernel are bracketed by HIP events on THEIR launch stream. enable(n): n = 1 brackets every launch, n > 1
 * every n-th launch (the others are only counted) — an event pair idles the chip for ~5 us, and a stream with one after
 * each of its 258 launches per step runs at a lighter duty cycle (on a power-limited chip: a higher clock) than the
 * un-instrumented step; a co-prime stride samples every launch position over a few steps at 1/n of that disturbance.
 * fvqa_gemm_timing_read synchronises, returns the number of launches recorded since enable and fills up to `max`
 * entries: duration (us; -2 = counted but not bracketed, -1 = event error), algorithmic FLOPs (2*M*N*K of that launch)
 * and kind = epilogue | split_k << 4 | out_is_f32 << 5 | in_is_f32 << 6 | four_wave_kernel << 7; it then clears the record (max <= 0: size query
 * only, nothing cleared). enable(0) stops recording and frees the probe. Launches from any host thread are recorded (the
 * step's backward runs on the autograd thread); switch it while no launch is in flight. */
int fvqa_gemm_timing_enable(int on);
int fvqa_gemm_timing_read(int max, float* us, double* flops, int* kind);

/* ---- RMSNorm (llama/model.py:37-42; used :185,186,347) -------------------------------- */
int fvqa_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int rows, int dim,
                     float eps, int dtype, void* stream);
/* dx = (resid ? resid : 0) + rmsnorm_bwd(g; x, w, rstd)   (weight frozen ⇒ no dw) */
int fvqa_rmsnorm_bwd(const void* g, const void* x, const void* w, const float* rstd,
                     const void* resid, void* dx, int rows, int dim, int dtype, void* stream);

/* ---- RoPE on the q and k column blocks of a fused qkv buffer (llama/model.py:61-67,96).
 * qkv is (rows, 3*dim) with q at cols [0,dim), k at [dim,2dim). cos/sin are (S, head_dim/2)
 * fp32 tables (llama/model.py:45-50). inverse != 0 applies the conjugate rotation (backward).
 * In place. */
int fvqa_rope_qk(void* qkv, const float* cos_t, const float* sin_t, int n_seq, int seq_len,
                 int n_heads, int head_dim, int inverse, int dtype, void* stream);

/* ---- SwiGLU (llama/model.py:142). ab is (rows, 2*hidden) in the AB16 layout (top of this file). */
int fvqa_swiglu_fwd(const void* ab, void* z, int rows, int hidden, int dtype, void* stream);
/* dab (rows, 2*hidden) <- d/d(a,b) of silu(a)*b given dz (rows, hidden) */
int fvqa_swiglu_bwd(const void* dz, const void* ab, void* dab, int rows, int hidden, int dtype,
                    void* stream);

/* ---- adapter-gated prefix attention + causal attention (llama/model.py:98-126) --------
 * qkv: (n_seq*S + A, 3*dim) — sequence rows then the A adapter rows (whose k and v column
 * blocks are adapter·Wk^T, adapter·Wv^T, no RoPE). o: (n_seq*S, dim).
 * gate1, gate2: (H) fp32. vstart: (n_seq) int32, -1 ⇒ no gate2 bias for that sequence
 * (the QAV stream, llama/model.py:121-122). lse_a/lse_t: (n_seq, H, S) fp32 log-sum-exp of
 * the adapter softmax and of the causal softmax, saved for the backward. */
/* cos_t/sin_t (both NULL or both given): when given, q and k in `qkv` are the RAW projections and the
 * kernel rotates them on the fly with the tables of fvqa_rope_qk (identical values to the separate
 * pass; llama/model.py:96); only builds for which fvqa_attn_rope_fused(dtype) == 1 accept them. */
int fvqa_attn_rope_fused(int dtype);
/* 1 when the step schedule computes the adapter K/V rows of layer i+1 as the rider of layer i's W1|W3 launch (and layer 0's as
 * a launch of its own before the walk) instead of beside layer i+1's QKV projection: bf16 build with the (s, t) SwiGLU form.
 * fvqa_swiglu_st: 1 unless FVQA_SWIGLU_AB=1 (tuning switch: a, b saved and the full SwiGLU' arithmetic in the W2^T epilogue). */
int fvqa_kv_rider_ahead(int dtype);
int fvqa_swiglu_st(void);
/* 1 when the step schedule (fvqa_layers_forward / _backward) rotates q, k in the QKV projection's epilogue (fvqa_gemm_nt_rope):
 * the arena's qkv rows — the KV cache of the generation path — then hold ROTATED q, k (bf16 MFMA build; FVQA_ROPE_IN_GEMM=0
 * keeps raw q, k rotated inside the attention kernels). */
int fvqa_rope_in_gemm(int dtype);
int fvqa_attn_fwd(const void* qkv, void* o, float* lse_a, float* lse_t, const float* gate1,
                  const float* gate2, const int32_t* vstart, const float* cos_t, const float* sin_t,
                  int n_seq, int seq_len, int n_heads, int head_dim, int adapter_len, int max_feats,
                  int dtype, void* stream);
/* One-query-row attention of the generation path (llama/model.py:428-470 re-runs the whole sequence per new token; this
 * evaluates Attention.forward :87-128 at the new row only). qkv_row (n_seq, 3*dim): RAW q | k | v projections of each
 * sequence's new token, whose position is pos[n] (int64, device). qkv_cache: the layer's (n_seq*S + A, 3*dim) buffer of
 * fvqa_attn_fwd, holding the keys / values of positions < pos[n] and the adapter rows. cache_rotated says what the cache
 * holds: 1 = ROTATED k (the fp32 build; the bf16 build when the QKV projection rotates in its epilogue, fvqa_rope_in_gemm —
 * the default), 0 = RAW k (bf16 build with FVQA_ROPE_IN_GEMM=0, rotated on the fly); the rule for a caller is
 * cache_rotated = !fvqa_attn_rope_fused(dtype) || fvqa_rope_in_gemm(dtype) — passing 0 for a rotated cache would rotate the
 * cached keys twice. The kernel rotates the new q and k with the tables (rounded to the storage type, as the prefill holds
 * them), writes o_row (n_seq, dim) and stores the new token's k (in the cache's convention) and v into cache row
 * n*S + pos[n]. seq_len <= 4096. */
int fvqa_attn_decode(const void* qkv_row, void* qkv_cache, void* o_row, const float* gate1, const float* gate2,
                     const int32_t* vstart, const int64_t* pos, const float* cos_t, const float* sin_t, int n_seq,
                     int seq_len, int n_heads, int head_dim, int adapter_len, int max_feats, int cache_rotated,
                     int dtype, void* stream);
/* workspace bytes fvqa_attn_bwd needs (fp32 partials for the batch-summed adapter k/v
 * gradients and the per-head gate sums). Its FIRST 1024 BYTES are integer arrival counters of the
 * fused bf16 backward: the caller zeroes them once after allocating the workspace; every call
 * leaves them zero again. One workspace serves one stream at a time. */
size_t fvqa_attn_bwd_workspace(int n_seq, int seq_len, int n_heads, int head_dim, int adapter_len);
/* dqkv: (n_seq*S + A, 3*dim): dq,dk,dv for sequence rows; adapter rows get [0, dK_a, dV_a] summed over
 * sequences. With cos_t/sin_t == NULL, qkv holds rotated q,k and dq,dk come out NOT yet un-rotated
 * (fvqa_rope_qk inverse follows); with the tables, qkv is raw and dq,dk are gradients of the raw
 * projections. dgate1/dgate2 (H) fp32 are ACCUMULATED (+=). */
int fvqa_attn_bwd(const void* d_o, const void* qkv, const void* o, const float* lse_a,
                  const float* lse_t, const float* gate1, const float* gate2, const int32_t* vstart,
                  const float* cos_t, const float* sin_t,
                  void* dqkv, float* dgate1, float* dgate2, void* workspace, size_t workspace_bytes,
                  int n_seq, int seq_len, int n_heads, int head_dim, int adapter_len, int max_feats,
                  int dtype, void* stream);
/* As fvqa_attn_bwd with tables, for a `qkv` whose q, k fvqa_gemm_nt_rope has ALREADY rotated: nothing is rotated on load, dq and
 * dk are conjugate-rotated at the store, so dqkv holds the gradients of the RAW projections (bf16 MFMA build only). */
int fvqa_attn_bwd_rotated(const void* d_o, const void* qkv, const void* o, const float* lse_a,
                  const float* lse_t, const float* gate1, const float* gate2, const int32_t* vstart,
                  const float* cos_t, const float* sin_t,
                  void* dqkv, float* dgate1, float* dgate2, void* workspace, size_t workspace_bytes,
                  int n_seq, int seq_len, int n_heads, int head_dim, int adapter_len, int max_feats,
                  int dtype, void* stream);

/* ---- visual projection + temporal embedding (llama/model.py:322,324) ------------------
 * vf_raw[b,f,:] = video[b,f,:]·W^T (fp32, kept for the QAV head); vf_tok = storage-dtype
 * cast of vf_raw + temporal[f,:]. video (BF, in_dim) fp32, W (dim, in_dim) fp32. */
int fvqa_visual_proj_fwd(const float* video, const float* W, const float* temporal, float* vf_raw,
                         void* vf_tok, int n_frames_total, int max_feats, int in_dim, int dim,
                         int dtype, void* stream);
/* d_tok (BF,dim) fp32: gradient wrt the spliced frame tokens (fvqa_splice_bwd); d_qav (BF,dim)
 * fp32 or NULL: extra gradient wrt vf_raw from the QAV head.
 * dW (dim,in_dim) += (d_tok + d_qav)^T·video ; dtemporal (F,dim) += sum_b d_tok[b]. */
int fvqa_visual_proj_bwd(const float* d_tok, const float* d_qav, const float* video, float* dW,
                         float* dtemporal, int n_frames_total, int max_feats, int in_dim, int dim,
                         void* stream);

/* ---- token embedding gather + frame splice (llama/model.py:286-294,326-336) -----------
 * h[n,s,:] = emb[ids[n,s]] (zeroed where zero_labels[n,s] >= 0; NULL = never) then
 * mode 0 (vqa/vaq): rows [vstart, vstart+F) := vf_tok[n]           (slice assign, :327,:332)
 * mode 1 (qav):     h[n, index[n,f], :] += vf_tok[n,f]              (scatter_add_, :335-336) */
int fvqa_embed_splice(const int64_t* ids, const void* emb, const void* vf_tok,
                      const int64_t* zero_labels, const int64_t* index, void* h, int n_seq,
                      int seq_len, int dim, int max_feats, int vstart, int mode, int dtype,
                      void* stream);
/* d_tok[n,f,:] += dh[n, row(n,f), :] (fp32 accumulate); row = vstart+f (mode 0) or index[n,f] */
int fvqa_splice_bwd(const void* dh, const int64_t* index, float* d_tok, int n_seq, int seq_len,
                    int dim, int max_feats, int vstart, int mode, int dtype, void* stream);

/* ---- LM-head cross-entropy (llama/model.py:233-234,349-350,355-356) --------------------
 * logits (n_seq, S, V) fp32; labels (n_seq, S) int64 — row (n,s) for s < S-1 is scored
 * against labels[n, s+1]; ignore_index rows are skipped; mean over valid rows.
 * loss_sum[0] += sum of row losses, loss_sum[1] += number of valid rows (fp32, zero it first).
 * lse, rowloss (n_seq*S) receive each scored row's log-sum-exp and loss (fixed-order sum). */
int fvqa_ce_fwd(const float* logits, const int64_t* labels, float* lse, float* rowloss,
                float* loss_sum, int n_seq, int seq_len, int vocab, int64_t ignore_index,
                void* stream);
/* dlogits (n_seq,S,V) storage dtype = (softmax - onehot) * gscale[0] / n_valid on valid rows,
 * 0 elsewhere (including s = S-1). gscale: device fp32 scalar (upstream grad of the loss). */
int fvqa_ce_bwd(const float* logits, const int64_t* labels, const float* lse, const float* loss_sum,
                const float* gscale, void* dlogits, int n_seq, int seq_len, int vocab,
                int64_t ignore_index, int dtype, void* stream);

/* ---- QAV head (llama/model.py:359-361): logits[n,s,f] = xn[n,s,:]·vf_raw[n,f,:]/tau for
 * s < S-1, CE(ignore_index=-1) against labels[n,s+1]. xn has storage dtype.
 * probs (n_seq,S,F) fp32 scratch keeps the softmax for the backward. */
int fvqa_qav_head_fwd(const void* xn, const float* vf_raw, const int64_t* labels, float* probs,
                      float* rowloss, float* loss_sum, int n_seq, int seq_len, int dim,
                      int max_feats, float tau, int dtype, void* stream);
/* dxn (storage dtype, fully written) and d_raw (n_seq,F,dim) fp32 (+=) */
int fvqa_qav_head_bwd(const void* xn, const float* vf_raw, const int64_t* labels, const float* probs,
                      const float* loss_sum, const float* gscale, void* dxn, float* d_raw, int n_seq,
                      int seq_len, int dim, int max_feats, float tau, int dtype, void* stream);

/* ---- loss-scaler + grad-norm + AdamW (util/misc.py:259-273,282-294; train.py:120-121) --
 * One flat fp32 buffer holds every trainable gradient; seg_off (n_seg+1) int64 marks the
 * per-parameter segments. scale, found_inf, step, growth_tracker are DEVICE fp32 scalars, so
 * the optimizer step needs no device->host read.
 * fvqa_grad_unscale_norm: g *= 1/(scale[0]*grad_div) in place; seg_sq[i] = sum g^2 over segment i;
 * total_norm[0] = sqrt(sum_i seg_sq[i]) (the norm of per-parameter norms, util/misc.py:292);
 * found_inf[0] = 1 if any gradient is non-finite else 0 (GradScaler.unscale_).
 * grad_div >= 1: the number of data-parallel replicas whose gradients were SUMMED into `grad` by the all-reduce (DDP's
 * mean of train.py:115-117 without a pass of its own; 1 on a single GPU).
 * gemm_err (may be NULL): device address of a persistent-GEMM workspace's error word (fvqa_gemm_workspace); when it is
 * non-zero the step is skipped like an overflow and found_inf[0] = 2.
 * err_lane (may be NULL): one fp32 that data-parallel ranks all-reduce TOGETHER with the gradients (the element behind the
 * flat gradient buffer, set to 1 by a rank whose error word is raised): non-zero on every rank as soon as one rank's
 * exchange timed out, so all replicas skip the same step (found_inf[0] = 2) and stop together.
 * fvqa_scaler_update leaves the loss scale and its growth tracker untouched when found_inf[0] == 2. */
size_t fvqa_grad_norm_workspace(int n_seg);
int fvqa_grad_unscale_norm(float* grad, const int64_t* seg_off, int n_seg, const float* scale, float grad_div,
                           const void* gemm_err, const float* err_lane, float* seg_sq, float* found_inf,
                           float* total_norm, void* workspace, size_t workspace_bytes, void* stream);
/* AdamW (decoupled weight decay, bias correction with t = step[0]+1), skipped entirely when
 * found_inf[0] != 0 (GradScaler.step semantics). found_inf may be NULL. */
int fvqa_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                    float lr, float beta1, float beta2, float eps, float weight_decay,
                    const float* step, const float* found_inf, void* stream);
/* after the AdamW launches of one optimizer step: step[0] += 1 unless found_inf; dynamic loss
 * scale update of torch GradScaler (x backoff on overflow, x growth after growth_interval clean
 * steps). step or scale may be NULL. */
int fvqa_scaler_update(float* step, float* scale, float* growth_tracker, const float* found_inf,
                       float growth_factor, float backoff_factor, int growth_interval, void* stream);

/* ---- the rows a head reads ("tail rows") -----------------------------------------------------
 * The reference evaluates `output` at every position and lets the cross-entropy ignore the rows whose next label is 0
 * (llama/model.py:348-350); the QAV head likewise reads the frame-token rows only (:359-361). A row no head
 * reads contributes neither to a loss nor to any gradient — and in the LAST layer nothing else reads its output either. The
 * step therefore runs the last layer's post-attention half (WO + residual, FFN, final norm), the heads and all of their
 * backward on the tail rows only, gathered into a compact matrix; results are those of the dense form.
 * Row movers: a compact matrix of segs->off[n] rows against n streams of `stream_rows` dense rows each.
 *   gather : compact row off[k] + j  <- dense row k*stream_rows + map[k][j]      (an index outside the stream: a zero row)
 *   scatter: dense row k*stream_rows + r <- compact row off[k] + map[k][r], zeros where map[k][r] < 0 (every dense row of
 *            the n streams written exactly once: no pre-clear, no atomics)
 * map[k]: int32 device arrays (gather: off[k+1]-off[k] entries; scatter: stream_rows entries); rows 16-byte aligned,
 * dim a multiple of 16 bytes. */
typedef struct fvqa_row_segs {
  int32_t n, stream_rows;
  int32_t off[4];
  const int32_t* map[3];
} fvqa_row_segs;
int fvqa_gather_rows(const void* src, void* dst, const fvqa_row_segs* segs, int dim, int dtype, void* stream);
int fvqa_scatter_rows(const void* src, void* dst, const fvqa_row_segs* segs, int dim, int dtype, void* stream);

/* ---- native layer schedule (csrc/schedule.hip): the L transformer blocks of the step walked in
 * C++ — two calls per step instead of ~700 per-kernel calls from the host language. Every pointer
 * is a device pointer except the per-layer tables (host arrays of n_layers device pointers).
 * Layer-strided activation buffers are contiguous (layer-major) with R = n_seq*seq_len rows and
 * Ra = R + adapter_len rows. */
typedef struct fvqa_layer_plan {
  int32_t dtype, n_layers, n_seq, seq_len, n_heads, head_dim, adapter_len, max_feats, dim, hidden;
  float eps;
  int32_t reserved_;
  /* frozen weights, per layer: fused and transposed copies (storage dtype) */
  const void* const* wqkv;   /* (3D, D)  */
  const void* const* wo;     /* (D, D)   */
  const void* const* w13;    /* (2Hf, D): W1 | W3 rows interleaved in blocks of 16 (AB16) */
  const void* const* w2;     /* (D, Hf)  */
  const void* const* wqkv_t; /* (D, 3D)  */
  const void* const* wo_t;   /* (D, D)   */
  const void* const* w13_t;  /* (D, 2Hf): transpose of w13 (AB16 columns) */
  const void* const* w2_t;   /* (Hf, D)  */
  const void* const* an;     /* attention_norm weight (D) */
  const void* const* fn;     /* ffn_norm weight (D)       */
  const float* const* gate1; /* (H) fp32 */
  const float* const* gate2;
  float* const* dgate1;      /* (H) fp32, accumulated */
  float* const* dgate2;
  const float* adapter;      /* (L, A, D) fp32 adapter queries of the walked layers */
  void* adapter_c;           /* (L, A, D) scratch: their storage-dtype cast (written by fvqa_layers_fwd) */
  float* d_adapter;          /* (L, A, D) fp32, accumulated */
  const void* norm_w;        /* final norm weight (D) */
  /* forward arena */
  void* xs;                  /* (L+1, R, D): xs[0] in, xs[L] out      */
  float* rstd1;              /* (L, R) */
  float* rstd2;              /* (L, R) */
  void* qkv;                 /* (L, Ra, 3D) */
  void* o;                   /* (L, R, D)   */
  float* lse_a;              /* (L, n_seq*H*S) */
  float* lse_t;
  void* h;                   /* (L, R, D)   */
  void* ab;                  /* (L, R, 2Hf), AB16 */
  void* xn;                  /* (R, D) scratch */
  void* hn;                  /* (R, D)  scratch */
  void* z;                   /* (R, Hf) scratch */
  void* xnf;                 /* (R, D) final-norm output */
  float* rstdN;              /* (R) */
  const float* cos_t;        /* (>=S, Dh/2) */
  const float* sin_t;
  const int32_t* vstart;     /* (n_seq) */
  /* backward scratch */
  void* dcur;                /* (R, D) */
  void* dnxt;                /* (R, D) */
  void* dz;                  /* (R, max(Hf, D)) scratch */
  void* dab;                 /* (R, 2Hf), AB16 */
  void* dh;                  /* (R, D) */
  void* d_o;                 /* (R, D) */
  void* dqkv;                /* (Ra, 3D) */
  void* attn_ws;
  size_t attn_ws_bytes;
  void* gemm_ws;             /* >= fvqa_layers_gemm_workspace(plan) bytes, 256-byte aligned, first 4096 bytes
                                zeroed once after allocation (see fvqa_gemm_workspace) */
  size_t gemm_ws_bytes;
  /* tail rows (fvqa_row_segs above; rows == 0: dense, nothing here is read). With rows > 0 fvqa_layers_fwd runs the
   * last layer's post-attention half and the final norm on the gathered rows — xs[L], h[L-1], ab[L-1], rstd2[L-1], xnf and
   * rstdN are NOT written; tail.xnf (rows, D) is the final-norm output — and fvqa_layers_bwd takes `dxnf` as (rows, D). */
  struct fvqa_tail_rows {
    int32_t rows, reserved_;
    fvqa_row_segs gather, scatter;                  /* the same segments with the idx / inv maps */
    void *og, *xg, *h, *hn, *ab, *z, *xl, *xnf;   /* (rows, D) except ab (rows, 2Hf), z (rows, Hf); saved for the backward: h, ab, xl */
    float *rstd2, *rstdN;                           /* (rows) */
    void *dcur, *dab, *dt, *dh, *d_o;               /* backward scratch: (rows, D) except dab (rows, 2Hf) */
  } tail;
} fvqa_layer_plan;

/* bytes of GEMM workspace the projections of one layer (forward and backward) need at most */
size_t fvqa_layers_gemm_workspace(const fvqa_layer_plan* plan);
/* xs[0] -> ... -> xs[L], xnf = final RMSNorm; saves what the backward needs in the arena */
int fvqa_layers_fwd(const fvqa_layer_plan* plan, void* stream);
/* dxnf (R, D): gradient w.r.t. xnf. *d_x0 <- device pointer (plan->dcur or plan->dnxt) holding the
 * gradient w.r.t. xs[0]; gate / adapter gradients are accumulated into the plan's buffers. */
int fvqa_layers_bwd(const fvqa_layer_plan* plan, const void* dxnf, void** d_x0, void* stream);

/* ---- small utilities -------------------------------------------------------------------
 * dst rows [row0, row0+n_rows) of a (.., dim) storage-dtype matrix <- fp32 src (n_rows, dim)
 * (adapter_query rows appended under the normed activations, llama/model.py:339) */
int fvqa_cast_rows(const float* src, void* dst, int n_rows, int dim, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FVQA_H */
