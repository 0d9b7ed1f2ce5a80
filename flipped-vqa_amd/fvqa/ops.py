"""Tensor-level wrappers over the C ABI (include/fvqa.h): each function validates shapes, dtypes,
contiguity and device ON THE HOST before a kernel is launched (a mis-shaped operand must raise
here, never fault on the GPU), then passes raw device pointers + the current HIP stream.

PyTorch is only the owner of device memory and streams here; all arithmetic is in libfvqa_hip.so.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from ._lib import BF16, EPI_NONE, EPI_RESIDUAL, EPI_SWIGLU_BWD, EPI_SWIGLU_BWD_ST, F16, F32

_DT = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}
_DTN = {torch.float32: "f32", torch.bfloat16: "bf16", torch.float16: "f16"}

def dt_code(dtype: torch.dtype) -> int:
    try:
        return _DT[dtype]
    except KeyError:
        raise TypeError(f"storage dtype must be float32, bfloat16 or float16, got {dtype}") from None


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _dev(*ts, rows_strided: bool = False):
    """rows_strided: 2-D views with unit inner stride are accepted (column blocks of wider matrices)."""
    dev = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise ValueError("fvqa ops need device tensors (no CPU fallback exists)")
        if rows_strided:
            if t.dim() != 2 or t.stride(1) != 1 or t.stride(0) < t.shape[1]:
                raise ValueError("fvqa GEMM operands must be 2-D with unit inner stride")
        elif not t.is_contiguous():
            raise ValueError("fvqa ops need contiguous tensors")
        dev = dev or t.device
        if t.device != dev:
            raise ValueError("tensors on different devices")


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _need(cond: bool, msg: str):
    if not cond:
        raise ValueError(msg)


# ------------------------------------------------------------------------------------ GEMM
_GEMM_WS = {}


def gemm_workspace(device=None, nbytes: int = 0) -> torch.Tensor:
    """THE persistent-GEMM workspace of (device, current stream): epoch flags + partial-tile slabs, shared by every GEMM
    launched on that stream (launches are stream-ordered, and include/fvqa.h allows one workspace per stream at a time:
    a second stream gets a workspace of its own). Its size does not depend on the problem (fvqa_gemm_sk_workspace), so
    the buffer — and the address of its error word — never moves once made."""
    device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _GEMM_WS.get(key)
    if ws is None:
        need = max(int(nbytes), int(_lib.load().fvqa_gemm_sk_workspace()))
        # zero-filled: the first 4 KiB are the epoch flags of the persistent GEMM (include/fvqa.h), which must
        # start at zero and are never reset afterwards
        ws = torch.zeros(need, dtype=torch.uint8, device=device)
        _GEMM_WS[key] = ws
    if ws.numel() < nbytes:
        raise RuntimeError("fvqa: GEMM workspace request exceeds fvqa_gemm_sk_workspace()")
    return ws


def gemm_error_word(device=None) -> Optional[torch.Tensor]:
    """8-byte view of the error word of the current stream's GEMM workspace (device memory; handed to
    grad_unscale_norm so that a timed-out split-K exchange skips the optimizer step like an overflow); None while no
    persistent GEMM has been launched on this stream."""
    device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    idx = device.index if device.index is not None else torch.cuda.current_device()
    ws = _GEMM_WS.get((idx, torch.cuda.current_stream(device).cuda_stream))
    return None if ws is None else ws[:8]


def gemm_error(ws: Optional[torch.Tensor] = None, device=None) -> int:
    """OR of the error words of the persistent-GEMM workspaces of `device` (include/fvqa.h: non-zero after a launch
    whose split-K exchange timed out), or the word of the given workspace. Device->host reads: not for the step path."""
    if ws is not None:
        return int(ws[:8].view(torch.int64)[0].item())
    idx = None
    if device is not None:
        device = torch.device(device)
        idx = device.index if device.index is not None else torch.cuda.current_device()
    bad = 0
    for (di, _), w in list(_GEMM_WS.items()):
        if idx is None or di == idx:
            bad |= int(w[:8].view(torch.int64)[0].item())
    return bad


class gemm4w_width:
    """`with ops.gemm4w_width(nbt):` — the whole-tile bf16 projections launched by THIS host thread inside the block use tiles of
    16 * nbt columns (nbt in 11, 12, 13, 14, 16) instead of the cost model's pick (include/fvqa.h fvqa_gemm4w_force); tests and
    the width sweep of tools/gemm4w_widths.py. nbt = 0 / None: the cost model."""

    def __init__(self, nbt: Optional[int]):
        self.nbt = int(nbt or 0)

    def __enter__(self):
        _lib.load()
        self.prev = {k: int(lib.fvqa_gemm4w_force(self.nbt)) for k, lib in _lib._LIBS.items()}   # (every loaded library)
        _need(all(v >= 0 for v in self.prev.values()), f"gemm4w_width: no tile width of {self.nbt} x 16 columns")
        return self

    def __exit__(self, *exc):
        for k, v in self.prev.items():
            _lib._LIBS[k].fvqa_gemm4w_force(v)
        return False


def gemm4w_choose(M: int, N: int, K: int, *, out_f32: bool = False, epilogue: int = EPI_NONE, rider_nk=None,
                  n_cu: int = 256) -> int:
    """Tile width (in 16-column blocks) the cost model picks for a bf16 projection on n_cu CUs; 0 = not the whole-tile kernel's
    problem. rider_nk: (N2, K2) of a <= 16-row rider product carried by the launch. Host only (no GPU touched)."""
    import ctypes as C
    rd = None
    if rider_nk is not None:
        rd = _lib.SkRider(1, 1, 1, 10, int(rider_nk[0]), int(rider_nk[1]), int(rider_nk[1]), int(rider_nk[1]), int(rider_nk[0]), 0)
    return int(_lib.load().fvqa_gemm4w_choose(M, N, K, _lib.BF16, _lib.F32 if out_f32 else _lib.BF16, epilogue,
                                              C.addressof(rd) if rd is not None else None, n_cu))


def gemm_nt(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor, *, residual: Optional[torch.Tensor] = None,
            tail: Optional[torch.Tensor] = None, m_split: int = 0, variant: int = 0, nbt: Optional[int] = None) -> torch.Tensor:
    """out[M,N] = a[M,K] @ b[N,K]^T (+ residual). Rows >= m_split go to `tail` (fp32) when given. nbt: force the whole-tile
    kernel's tile width for this call (gemm4w_width)."""
    if nbt:
        with gemm4w_width(nbt):
            return gemm_nt(a, b, out, residual=residual, tail=tail, m_split=m_split, variant=variant)
    _dev(a, b, out, residual, rows_strided=True)
    _dev(tail)
    _need(a.dim() == 2 and b.dim() == 2, "gemm_nt: 2-D operands")
    M, K = a.shape
    N, K2 = b.shape
    _need(K == K2, f"gemm_nt: K mismatch {K} vs {K2}")
    _need(a.dtype == b.dtype, "gemm_nt: a/b dtype mismatch")
    if out is None:                                  # every row accumulates into the fp32 tail
        _need(tail is not None and m_split == 0 and residual is None, "gemm_nt: out=None needs tail and m_split=0")
        _need(tail.dtype == torch.float32 and tuple(tail.shape) == (M, N), "gemm_nt: tail shape")
        out_dtype = a.dtype
    else:
        _need(out.dim() == 2 and out.dtype in (a.dtype, torch.float32), "gemm_nt: out dtype")
        out_dtype = out.dtype
        if tail is None:
            _need(tuple(out.shape) == (M, N), f"gemm_nt: out shape {tuple(out.shape)} != {(M, N)}")
            m_split = M
        else:
            _need(0 <= m_split <= M, "gemm_nt: m_split")
            _need(out.shape[0] >= m_split and out.shape[1] == N, "gemm_nt: out too small")
            _need(tail.dtype == torch.float32 and tuple(tail.shape) == (M - m_split, N), "gemm_nt: tail shape")
    epi = EPI_NONE
    if residual is not None:
        _need(residual.dtype == a.dtype and out_dtype == a.dtype, "gemm_nt: residual dtype")
        _need(residual.shape[1] == N and residual.shape[0] >= min(M, m_split), "gemm_nt: residual shape")
        epi = EPI_RESIDUAL
    lib = _lib.load(a.dtype)
    code = dt_code(a.dtype)
    if variant == 13:
        need = int(lib.fvqa_gemm_sk_workspace())
    elif variant == 0:
        need = int(lib.fvqa_gemm_workspace(M, N, K, code))
    else:
        need = 0
    ws = gemm_workspace(a.device, need) if need else None
    ldc = out.stride(0) if out is not None else N
    _need(residual is None or residual.stride(0) == ldc, "gemm_nt: residual and out must share their row stride")
    rc = lib.fvqa_gemm_nt(_ptr(a), _ptr(b), _ptr(out), _ptr(residual), _ptr(tail), M, N, K, a.stride(0), b.stride(0),
                          ldc, m_split,
                          code, dt_code(out_dtype), epi, variant, _ptr(ws), ws.numel() if ws is not None else 0,
                          _stream())
    _lib.check(rc, "fvqa_gemm_nt")
    return out


def gemm_nt_rider(a, b, out, *, rider_a, rider_b, rider_out, accumulate: bool = False, residual=None, swiglu_ab=None,
                  swiglu_st: bool = False):
    """out = a @ b^T (+ residual | SwiGLU' epilogue with swiglu_ab) and, on the CUs that launch leaves idle (or right
    after it), the small product rider_out (<= 16 rows) = rider_a @ rider_b^T (accumulate: fp32 rider_out += product).
    Operands may be column-block views of wider matrices (rows 16-byte aligned)."""
    import ctypes as C
    for t in (a, b, out, residual, swiglu_ab, rider_a, rider_b, rider_out):
        if t is not None and (not t.is_cuda or t.dim() != 2 or t.stride(1) != 1):
            raise ValueError("gemm_nt_rider: 2-D device tensors with unit inner stride")
    M, K = a.shape
    N = b.shape[0]
    _need(b.shape[1] == K and a.dtype == b.dtype == rider_a.dtype == rider_b.dtype, "gemm_nt_rider: operands")
    epi, R, ldc = EPI_NONE, None, out.stride(0)
    if swiglu_ab is not None:
        _need(tuple(out.shape) == (M, 2 * N) and swiglu_ab.shape == out.shape and out.is_contiguous() and
              swiglu_ab.is_contiguous(), "gemm_nt_rider: SwiGLU' operands")
        epi, R = (EPI_SWIGLU_BWD_ST if swiglu_st else EPI_SWIGLU_BWD), swiglu_ab
    else:
        _need(tuple(out.shape) == (M, N), "gemm_nt_rider: out shape")
        if residual is not None:
            _need(residual.shape == out.shape and residual.stride(0) == ldc, "gemm_nt_rider: residual")
            epi, R = EPI_RESIDUAL, residual
    M2, K2 = rider_a.shape
    N2 = rider_b.shape[0]
    _need(rider_b.shape[1] == K2 and tuple(rider_out.shape) == (M2, N2) and M2 <= 16, "gemm_nt_rider: rider shapes")
    acc = bool(accumulate)
    if acc:
        _need(rider_out.dtype == torch.float32 and rider_out.stride(0) == N2,
              "gemm_nt_rider: an accumulated rider_out is contiguous fp32")
    else:
        _need(rider_out.dtype == a.dtype, "gemm_nt_rider: rider_out dtype")
    rd = _lib.SkRider(rider_a.data_ptr(), rider_b.data_ptr(), rider_out.data_ptr(), M2, N2, K2, rider_a.stride(0),
                      rider_b.stride(0), rider_out.stride(0), 1 if acc else 0)
    lib = _lib.load(a.dtype)
    ws = gemm_workspace(a.device, int(lib.fvqa_gemm_sk_workspace()))
    rc = lib.fvqa_gemm_nt_rider(_ptr(a), _ptr(b), _ptr(out), _ptr(R), M, N, K, a.stride(0), b.stride(0), ldc,
                                dt_code(a.dtype), dt_code(out.dtype), epi, C.addressof(rd), _ptr(ws), ws.numel(),
                                _stream())
    _lib.check(rc, "fvqa_gemm_nt_rider")
    return out


def gemm_timing_enable(on, stride: int = 1) -> None:
    """Measurement probe (include/fvqa.h fvqa_gemm_timing_enable): HIP events around every `stride`-th launch of
    the persistent GEMM kernel on its launch stream, under whichever schedule is running; the other launches are
    counted only (gemm_timing_read returns -2 us for them). Applies to every library loaded so far (the bf16 / fp32 one and,
    once an fp16 model exists, the fp16 one: each keeps its own record)."""
    _lib.load()
    for lib in _lib._LIBS.values():
        _lib.check(lib.fvqa_gemm_timing_enable(int(stride) if on else 0), "fvqa_gemm_timing_enable")


def gemm_timing_read():
    """-> list of (microseconds, flops, kind) per recorded launch (all loaded libraries, one after the other); clears the record."""
    import ctypes as C
    out = []
    for lib in list(_lib._LIBS.values()):
        n = int(lib.fvqa_gemm_timing_read(0, None, None, None))
        if n <= 0:
            continue
        us, fl, kd = (C.c_float * n)(), (C.c_double * n)(), (C.c_int * n)()
        got = int(lib.fvqa_gemm_timing_read(n, C.cast(us, C.c_void_p), C.cast(fl, C.c_void_p), C.cast(kd, C.c_void_p)))
        out += [(float(us[i]), float(fl[i]), int(kd[i])) for i in range(min(n, got))]
    return out


def gemm_nt_swiglu_fwd(x: torch.Tensor, w13: torch.Tensor, ab: torch.Tensor, z: torch.Tensor, st: bool = False, *,
                       rider_a=None, rider_b=None, rider_out=None):
    """ab (M, 2*Hf) = x @ w13^T and z (M, Hf) = silu(a) * b in one launch; w13 / ab in the AB16 layout (pack_ab16).
    st=True (the training step): `ab` receives the backward's factors s = silu(a), t = dz/da in the a and b slots
    (FVQA_EPI_SWIGLU_FWD_ST), to be consumed by gemm_nt_swiglu_bwd(..., st=True). Optional rider (st=True only; <= 16 rows,
    rider_out = rider_a @ rider_b^T in the operands' dtype) as gemm_nt_rider."""
    _dev(x, w13, ab, z)
    _dev(rider_a, rider_b, rider_out, rows_strided=True)
    M, K = x.shape
    N = w13.shape[0]
    _need(w13.shape[1] == K and N % 32 == 0 and x.dtype == w13.dtype == ab.dtype == z.dtype, "gemm_nt_swiglu_fwd: operands")
    _need(ab.shape[-1] == N and ab.numel() >= M * N and z.shape[-1] == N // 2 and z.numel() >= M * N // 2,
          "gemm_nt_swiglu_fwd: ab / z shape")
    lib = _lib.load(x.dtype)
    ws = gemm_workspace(x.device, int(lib.fvqa_gemm_sk_workspace()))
    if rider_a is not None:
        _need(st, "gemm_nt_swiglu_fwd: a rider needs st=True")
        M2, K2 = rider_a.shape
        N2 = rider_b.shape[0]
        _need(rider_b.shape[1] == K2 and tuple(rider_out.shape) == (M2, N2) and M2 <= 16 and
              rider_a.dtype == rider_b.dtype == rider_out.dtype == x.dtype and
              rider_a.stride(1) == rider_b.stride(1) == rider_out.stride(1) == 1, "gemm_nt_swiglu_fwd: rider")
        rd = _lib.SkRider(rider_a.data_ptr(), rider_b.data_ptr(), rider_out.data_ptr(), M2, N2, K2, rider_a.stride(0),
                          rider_b.stride(0), rider_out.stride(0), 0)
        import ctypes as C
        rc = lib.fvqa_gemm_nt_swiglu_fwd_st_rider(_ptr(x), _ptr(w13), _ptr(ab), _ptr(z), M, N // 2, K, K, K, dt_code(x.dtype),
                                                  C.byref(rd), _ptr(ws), ws.numel(), _stream())
        _lib.check(rc, "fvqa_gemm_nt_swiglu_fwd_st_rider")
        return z
    fn = lib.fvqa_gemm_nt_swiglu_fwd_st if st else lib.fvqa_gemm_nt_swiglu_fwd
    rc = fn(_ptr(x), _ptr(w13), _ptr(ab), _ptr(z), M, N // 2, K, K, K, dt_code(x.dtype), _ptr(ws), ws.numel(), _stream())
    _lib.check(rc, "fvqa_gemm_nt_swiglu_fwd")
    return z


def pack_ab16(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """(.., H) a and b -> (.., 2H) in the AB16 layout of include/fvqa.h: 16 columns (rows, for weights passed as
    (H, K) matrices: use dim=0) of a, the matching 16 of b, and so on."""
    H = a.shape[-1]
    _need(a.shape == b.shape and H % 16 == 0, "pack_ab16: equal shapes, H % 16 == 0")
    lead = a.shape[:-1]
    return torch.stack([a.reshape(*lead, H // 16, 16), b.reshape(*lead, H // 16, 16)], dim=-2).reshape(*lead, 2 * H)


def unpack_ab16(ab: torch.Tensor):
    """inverse of pack_ab16 -> (a, b)"""
    H2 = ab.shape[-1]
    lead = ab.shape[:-1]
    v = ab.reshape(*lead, H2 // 32, 2, 16)
    return v[..., 0, :].reshape(*lead, H2 // 2), v[..., 1, :].reshape(*lead, H2 // 2)


def gemm_nt_swiglu_bwd(g: torch.Tensor, w2_t: torch.Tensor, ab: torch.Tensor, dab: torch.Tensor, st: bool = False):
    """dab (M, 2*Hf) = d/d(a,b)[silu(a)*b] with dz = g (M,D) @ w2_t (Hf,D)^T formed in the GEMM
    accumulators (the SwiGLU backward is the GEMM's epilogue; dz never reaches HBM). ab / dab: AB16 layout.
    st=True: `ab` holds (s, t) as left by gemm_nt_swiglu_fwd(..., st=True)."""
    _dev(g, w2_t, ab, dab)
    M, K = g.shape
    N = w2_t.shape[0]
    _need(w2_t.shape[1] == K and g.dtype == w2_t.dtype == ab.dtype == dab.dtype, "gemm_nt_swiglu_bwd: operands")
    _need(ab.numel() >= M * 2 * N and dab.numel() >= M * 2 * N and ab.shape[-1] == 2 * N and dab.shape[-1] == 2 * N,
          "gemm_nt_swiglu_bwd: ab/dab shape")
    lib = _lib.load(g.dtype)
    need = int(lib.fvqa_gemm_sk_workspace())                 # this epilogue lives in the persistent kernel
    ws = gemm_workspace(g.device, need)
    rc = lib.fvqa_gemm_nt(_ptr(g), _ptr(w2_t), _ptr(dab), _ptr(ab), None, M, N, K, K, K, 2 * N, M, dt_code(g.dtype),
                          dt_code(g.dtype), EPI_SWIGLU_BWD_ST if st else EPI_SWIGLU_BWD, 0, _ptr(ws),
                          ws.numel() if ws is not None else 0,
                          _stream())
    _lib.check(rc, "fvqa_gemm_nt(swiglu_bwd)")
    return dab


# ------------------------------------------------------------------------------------ row ops
def rmsnorm_fwd(x, w, y, rstd, eps: float, rows: Optional[int] = None):
    _dev(x, w, y, rstd)
    dim = x.shape[-1]
    rows = rows if rows is not None else x.numel() // dim
    _need(x.dtype == w.dtype == y.dtype, "rmsnorm_fwd: dtype")
    _need(w.numel() == dim and y.shape[-1] == dim, "rmsnorm_fwd: dim")
    _need(x.numel() >= rows * dim and y.numel() >= rows * dim, "rmsnorm_fwd: rows")
    _need(rstd is None or (rstd.dtype == torch.float32 and rstd.numel() >= rows), "rmsnorm_fwd: rstd")
    rc = _lib.load(x.dtype).fvqa_rmsnorm_fwd(_ptr(x), _ptr(w), _ptr(y), _ptr(rstd), rows, dim, float(eps),
                                      dt_code(x.dtype), _stream())
    _lib.check(rc, "fvqa_rmsnorm_fwd")
    return y


def rmsnorm_bwd(g, x, w, rstd, dx, resid=None, rows: Optional[int] = None):
    _dev(g, x, w, rstd, dx, resid)
    dim = x.shape[-1]
    rows = rows if rows is not None else x.numel() // dim
    _need(g.dtype == x.dtype == w.dtype == dx.dtype, "rmsnorm_bwd: dtype")
    _need(resid is None or resid.dtype == x.dtype, "rmsnorm_bwd: resid dtype")
    for t in (g, x, dx, resid):
        _need(t is None or (t.shape[-1] == dim and t.numel() >= rows * dim), "rmsnorm_bwd: shape")
    _need(rstd.dtype == torch.float32 and rstd.numel() >= rows and w.numel() == dim, "rmsnorm_bwd: rstd/w")
    rc = _lib.load(g.dtype).fvqa_rmsnorm_bwd(_ptr(g), _ptr(x), _ptr(w), _ptr(rstd), _ptr(resid), _ptr(dx), rows, dim,
                                      dt_code(x.dtype), _stream())
    _lib.check(rc, "fvqa_rmsnorm_bwd")
    return dx


def rope_qk(qkv, cos_t, sin_t, n_seq: int, seq_len: int, n_heads: int, head_dim: int, inverse: bool = False):
    _dev(qkv, cos_t, sin_t)
    dim = n_heads * head_dim
    _need(qkv.shape[-1] == 3 * dim and qkv.numel() >= n_seq * seq_len * 3 * dim, "rope_qk: qkv shape")
    _need(cos_t.dtype == sin_t.dtype == torch.float32, "rope_qk: table dtype")
    _need(cos_t.shape[-1] == head_dim // 2 and cos_t.shape[0] >= seq_len and sin_t.shape == cos_t.shape,
          "rope_qk: table shape")
    rc = _lib.load(qkv.dtype).fvqa_rope_qk(_ptr(qkv), _ptr(cos_t), _ptr(sin_t), n_seq, seq_len, n_heads, head_dim,
                                  int(inverse), dt_code(qkv.dtype), _stream())
    _lib.check(rc, "fvqa_rope_qk")
    return qkv


def swiglu_fwd(ab, z, rows: int, hidden: int):
    _dev(ab, z)
    _need(ab.dtype == z.dtype, "swiglu_fwd: dtype")
    _need(ab.numel() >= rows * 2 * hidden and z.numel() >= rows * hidden, "swiglu_fwd: shape")
    rc = _lib.load(ab.dtype).fvqa_swiglu_fwd(_ptr(ab), _ptr(z), rows, hidden, dt_code(ab.dtype), _stream())
    _lib.check(rc, "fvqa_swiglu_fwd")
    return z


def swiglu_bwd(dz, ab, dab, rows: int, hidden: int):
    _dev(dz, ab, dab)
    _need(dz.dtype == ab.dtype == dab.dtype, "swiglu_bwd: dtype")
    _need(ab.numel() >= rows * 2 * hidden and dab.numel() >= rows * 2 * hidden and dz.numel() >= rows * hidden,
          "swiglu_bwd: shape")
    rc = _lib.load(ab.dtype).fvqa_swiglu_bwd(_ptr(dz), _ptr(ab), _ptr(dab), rows, hidden, dt_code(ab.dtype), _stream())
    _lib.check(rc, "fvqa_swiglu_bwd")
    return dab


def row_segs(maps, offs, stream_rows: int):
    """fvqa_row_segs (include/fvqa.h) of up to three streams: maps[k] the int32 device map of stream k (gather: compact -> dense row
    within the stream; scatter: dense row -> compact row within the segment, -1 none), offs the segment offsets (len(maps) + 1)."""
    n = len(maps)
    _need(1 <= n <= 3 and len(offs) == n + 1 and offs[0] == 0, "row_segs: 1..3 segments starting at 0")
    sg = _lib.RowSegs()
    sg.n, sg.stream_rows = n, int(stream_rows)
    for k in range(4):
        sg.off[k] = int(offs[min(k, n)])
    for k, m in enumerate(maps):
        _dev(m)
        _need(m.dtype == torch.int32 and m.dim() == 1 and m.is_contiguous(), "row_segs: int32 maps")
        sg.map[k] = m.data_ptr()
    sg._keep = list(maps)
    return sg


def gather_rows(src, dst, segs):
    """dst (M, dim) compact <- rows of src ((n * stream_rows), dim) dense, by segs (row_segs of idx maps)."""
    import ctypes as C
    _dev(src, dst)
    _need(src.dim() == 2 and dst.dim() == 2 and src.dtype == dst.dtype and src.shape[1] == dst.shape[1], "gather_rows: shape")
    _need(src.shape[0] >= segs.n * segs.stream_rows and dst.shape[0] >= segs.off[segs.n], "gather_rows: rows")
    rc = _lib.load(src.dtype).fvqa_gather_rows(_ptr(src), _ptr(dst), C.addressof(segs), src.shape[1], dt_code(src.dtype), _stream())
    _lib.check(rc, "fvqa_gather_rows")
    return dst


def scatter_rows(src, dst, segs):
    """dst ((n * stream_rows), dim) dense <- rows of src (M, dim) compact by segs (row_segs of inv maps), zeros where the map is < 0."""
    import ctypes as C
    _dev(src, dst)
    _need(src.dim() == 2 and dst.dim() == 2 and src.dtype == dst.dtype and src.shape[1] == dst.shape[1], "scatter_rows: shape")
    _need(dst.shape[0] >= segs.n * segs.stream_rows and src.shape[0] >= segs.off[segs.n], "scatter_rows: rows")
    rc = _lib.load(src.dtype).fvqa_scatter_rows(_ptr(src), _ptr(dst), C.addressof(segs), src.shape[1], dt_code(src.dtype), _stream())
    _lib.check(rc, "fvqa_scatter_rows")
    return dst


def cast_rows(src, dst_rows):
    """dst_rows (n, dim) storage dtype <- src (n, dim) fp32."""
    _dev(src, dst_rows)
    _need(src.dtype == torch.float32 and src.shape == dst_rows.shape and src.dim() == 2, "cast_rows: shape")
    rc = _lib.load(dst_rows.dtype).fvqa_cast_rows(_ptr(src), _ptr(dst_rows), src.shape[0], src.shape[1],
                                    dt_code(dst_rows.dtype), _stream())
    _lib.check(rc, "fvqa_cast_rows")
    return dst_rows


# ------------------------------------------------------------------------------------ attention
def _attn_shapes(qkv, n_seq, S, H, Dh, A):
    D = H * Dh
    _need(qkv.dim() == 2 and tuple(qkv.shape) == (n_seq * S + A, 3 * D),
          f"attention: qkv must be ({n_seq * S + A}, {3 * D}), got {tuple(qkv.shape)}")
    _need(Dh == 128 and 1 <= A <= 16, "attention: head_dim must be 128 and adapter_len <= 16")
    return D


def kv_rider_ahead(dtype: torch.dtype) -> bool:
    """True when the step computes the adapter K/V rows of layer i+1 as the rider of layer i's W1|W3 launch (layer 0's as a
    launch of its own before the walk) instead of beside layer i+1's QKV projection (csrc/schedule.hip)."""
    return bool(_lib.load(dtype).fvqa_kv_rider_ahead(dt_code(dtype)))


def swiglu_st() -> bool:
    """True unless FVQA_SWIGLU_AB=1: the W1|W3 launch saves the SwiGLU backward's factors (s, t) instead of (a, b)."""
    return bool(_lib.load().fvqa_swiglu_st())


def rope_in_gemm(dtype: torch.dtype) -> bool:
    """True when the step rotates q, k in the QKV projection's epilogue (gemm_nt_rope): the arena's qkv rows hold ROTATED
    q, k, attn_fwd runs without tables and the backward is attn_bwd(..., prerotated=True)."""
    return bool(_lib.load(dtype).fvqa_rope_in_gemm(dt_code(dtype)))


def gemm_nt_rope(a, b, out, rope, seq_len, head_dim, n_heads, *, rider_a=None, rider_b=None, rider_out=None):
    """out (M, 3*D) = a @ b.T with RoPE applied to the q | k columns [0, 2*D) in the epilogue (bf16; position = row % seq_len);
    optional rider (<= 16 rows) as gemm_nt_rider."""
    _dev(a, b, out, rider_a, rider_b, rider_out, rows_strided=True)
    _need(a.dtype == b.dtype == out.dtype and a.dtype in (torch.bfloat16, torch.float16), "gemm_nt_rope: 16-bit storage")
    for t in (a, b, out):
        _need(t.dim() == 2 and t.stride(1) == 1, "gemm_nt_rope: 2-D tensors with unit inner stride")
    M, K = a.shape
    N = b.shape[0]
    D = n_heads * head_dim
    _need(b.shape[1] == K and tuple(out.shape) == (M, N) and N >= 2 * D and M % seq_len == 0, "gemm_nt_rope: shapes")
    cos_t, sin_t = _rope_tables(rope, seq_len, head_dim, "gemm_nt_rope")
    _need(cos_t is not None, "gemm_nt_rope: rope tables")
    rp = _lib.SkRope(cos_t.data_ptr(), sin_t.data_ptr(), seq_len, head_dim, 2 * D)
    rd = None
    if rider_a is not None:
        M2, K2 = rider_a.shape
        N2 = rider_b.shape[0]
        _need(rider_b.shape[1] == K2 and tuple(rider_out.shape) == (M2, N2) and M2 <= 16 and rider_out.dtype == a.dtype and
              rider_a.stride(1) == rider_b.stride(1) == rider_out.stride(1) == 1, "gemm_nt_rope: rider")
        rd = _lib.SkRider(rider_a.data_ptr(), rider_b.data_ptr(), rider_out.data_ptr(), M2, N2, K2, rider_a.stride(0),
                          rider_b.stride(0), rider_out.stride(0), 0)
    ws = gemm_workspace(a.device)
    import ctypes
    rc = _lib.load(a.dtype).fvqa_gemm_nt_rope(_ptr(a), _ptr(b), _ptr(out), M, N, K, a.stride(0), b.stride(0), out.stride(0),
                                       ctypes.byref(rp), ctypes.byref(rd) if rd is not None else None, _ptr(ws), ws.numel(),
                                       _stream())
    _lib.check(rc, "fvqa_gemm_nt_rope")
    return out


def attn_rope_fused(dtype: torch.dtype) -> bool:
    """True when attn_fwd / attn_bwd of this dtype rotate q,k themselves (rope=(cos, sin) argument)."""
    return bool(_lib.load(dtype).fvqa_attn_rope_fused(dt_code(dtype)))


def _rope_tables(rope, S, Dh, what):
    if rope is None:
        return None, None
    cos_t, sin_t = rope
    _dev(cos_t, sin_t)
    _need(cos_t.dtype == sin_t.dtype == torch.float32 and cos_t.shape == sin_t.shape and
          cos_t.shape[-1] == Dh // 2 and cos_t.shape[0] >= S and cos_t.is_contiguous() and sin_t.is_contiguous(),
          f"{what}: rope tables")
    return cos_t, sin_t


def attn_fwd(qkv, o, lse_a, lse_t, gate1, gate2, vstart, n_seq, S, H, Dh, A, F, rope=None):
    """rope=(cos_t, sin_t): qkv holds the raw projections and RoPE is applied inside (attn_rope_fused)."""
    _dev(qkv, o, lse_a, lse_t, gate1, gate2, vstart)
    cos_t, sin_t = _rope_tables(rope, S, Dh, "attn_fwd")
    D = _attn_shapes(qkv, n_seq, S, H, Dh, A)
    _need(o.dtype == qkv.dtype and tuple(o.shape) == (n_seq * S, D), "attn_fwd: o shape")
    for t in (lse_a, lse_t):
        _need(t.dtype == torch.float32 and t.numel() == n_seq * H * S, "attn_fwd: lse shape")
    for t in (gate1, gate2):
        _need(t.dtype == torch.float32 and t.numel() == H, "attn_fwd: gate shape")
    _need(vstart.dtype == torch.int32 and vstart.numel() == n_seq, "attn_fwd: vstart")
    rc = _lib.load(qkv.dtype).fvqa_attn_fwd(_ptr(qkv), _ptr(o), _ptr(lse_a), _ptr(lse_t), _ptr(gate1), _ptr(gate2),
                                   _ptr(vstart), _ptr(cos_t), _ptr(sin_t), n_seq, S, H, Dh, A, F,
                                   dt_code(qkv.dtype), _stream())
    _lib.check(rc, "fvqa_attn_fwd")
    return o


def attn_decode(qkv_row, qkv_cache, o_row, gate1, gate2, vstart, pos, rope, n_seq, S, H, Dh, A, F, cache_rotated: bool):
    """One new token per sequence (generation path): attention of the new row over the cached keys / values + the
    adapter prefix; the new token's k, v are stored into cache row n*S + pos[n]. qkv_row holds RAW projections."""
    _dev(qkv_row, qkv_cache, o_row, gate1, gate2, vstart, pos)
    cos_t, sin_t = _rope_tables(rope, S, Dh, "attn_decode")
    D = _attn_shapes(qkv_cache, n_seq, S, H, Dh, A)
    _need(cos_t is not None, "attn_decode: rope tables")
    _need(qkv_row.dtype == qkv_cache.dtype == o_row.dtype, "attn_decode: dtype")
    _need(tuple(qkv_row.shape) == (n_seq, 3 * D) and tuple(o_row.shape) == (n_seq, D), "attn_decode: row shapes")
    _need(pos.dtype == torch.int64 and pos.numel() == n_seq, "attn_decode: pos")
    _need(vstart.dtype == torch.int32 and vstart.numel() == n_seq, "attn_decode: vstart")
    for t in (gate1, gate2):
        _need(t.dtype == torch.float32 and t.numel() == H, "attn_decode: gate shape")
    rc = _lib.load(qkv_cache.dtype).fvqa_attn_decode(_ptr(qkv_row), _ptr(qkv_cache), _ptr(o_row), _ptr(gate1), _ptr(gate2),
                                      _ptr(vstart), _ptr(pos), _ptr(cos_t), _ptr(sin_t), n_seq, S, H, Dh, A, F,
                                      1 if cache_rotated else 0, dt_code(qkv_row.dtype), _stream())
    _lib.check(rc, "fvqa_attn_decode")
    return o_row


DECODE_PTRS = 9


def attn_bwd_workspace(n_seq, S, H, Dh, A) -> int:
    return int(_lib.load().fvqa_attn_bwd_workspace(n_seq, S, H, Dh, A))


def attn_bwd(d_o, qkv, o, lse_a, lse_t, gate1, gate2, vstart, dqkv, dgate1, dgate2, workspace,
             n_seq, S, H, Dh, A, F, rope=None, prerotated=False):
    """`workspace`: attn_bwd_workspace(...) bytes, ZEROED once when allocated (its head holds the fused
    kernel's arrival counters, which every call leaves at zero again)."""
    _dev(d_o, qkv, o, lse_a, lse_t, gate1, gate2, vstart, dqkv, dgate1, dgate2, workspace)
    cos_t, sin_t = _rope_tables(rope, S, Dh, "attn_bwd")
    D = _attn_shapes(qkv, n_seq, S, H, Dh, A)
    _need(dqkv.dtype == qkv.dtype and dqkv.shape == qkv.shape, "attn_bwd: dqkv shape")
    for t in (d_o, o):
        _need(t.dtype == qkv.dtype and tuple(t.shape) == (n_seq * S, D), "attn_bwd: o/d_o shape")
    for t in (lse_a, lse_t):
        _need(t.dtype == torch.float32 and t.numel() == n_seq * H * S, "attn_bwd: lse shape")
    for t in (gate1, gate2, dgate1, dgate2):
        _need(t.dtype == torch.float32 and t.numel() == H, "attn_bwd: gate shape")
    _need(vstart.dtype == torch.int32 and vstart.numel() == n_seq, "attn_bwd: vstart")
    wbytes = workspace.numel() * workspace.element_size()
    _need(wbytes >= attn_bwd_workspace(n_seq, S, H, Dh, A), "attn_bwd: workspace too small")
    # prerotated: q, k in `qkv` were rotated by gemm_nt_rope; dqkv still receives the gradients of the raw projections
    fn = _lib.load(qkv.dtype).fvqa_attn_bwd_rotated if prerotated else _lib.load(qkv.dtype).fvqa_attn_bwd
    _need(not prerotated or cos_t is not None, "attn_bwd: prerotated needs the rope tables")
    rc = fn(_ptr(d_o), _ptr(qkv), _ptr(o), _ptr(lse_a), _ptr(lse_t), _ptr(gate1), _ptr(gate2), _ptr(vstart), _ptr(cos_t),
            _ptr(sin_t), _ptr(dqkv), _ptr(dgate1), _ptr(dgate2), _ptr(workspace), wbytes, n_seq, S, H, Dh, A, F,
            dt_code(qkv.dtype), _stream())
    _lib.check(rc, "fvqa_attn_bwd")
    return dqkv


# ------------------------------------------------------------------------------------ heads / splice
def visual_proj_fwd(video, W, temporal, vf_raw, vf_tok):
    _dev(video, W, temporal, vf_raw, vf_tok)
    R, K = video.shape
    D = W.shape[0]
    F = temporal.shape[0]
    _need(video.dtype == W.dtype == temporal.dtype == vf_raw.dtype == torch.float32, "visual_proj_fwd: fp32")
    _need(W.shape[1] == K and temporal.shape[1] == D and R % F == 0, "visual_proj_fwd: shapes")
    _need(tuple(vf_raw.shape) == (R, D) and tuple(vf_tok.shape) == (R, D), "visual_proj_fwd: out shapes")
    rc = _lib.load(vf_tok.dtype).fvqa_visual_proj_fwd(_ptr(video), _ptr(W), _ptr(temporal), _ptr(vf_raw), _ptr(vf_tok), R, F,
                                          K, D, dt_code(vf_tok.dtype), _stream())
    _lib.check(rc, "fvqa_visual_proj_fwd")


def visual_proj_bwd(d_tok, d_qav, video, dW, dtemporal):
    _dev(d_tok, d_qav, video, dW, dtemporal)
    R, K = video.shape
    D = dW.shape[0]
    F = dtemporal.shape[0]
    for t in (d_tok, d_qav):
        _need(t is None or (t.dtype == torch.float32 and tuple(t.shape) == (R, D)), "visual_proj_bwd: d_tok/d_qav")
    _need(dW.dtype == dtemporal.dtype == video.dtype == torch.float32, "visual_proj_bwd: fp32")
    _need(tuple(dW.shape) == (D, K) and tuple(dtemporal.shape) == (F, D) and R % F == 0, "visual_proj_bwd: shapes")
    rc = _lib.load().fvqa_visual_proj_bwd(_ptr(d_tok), _ptr(d_qav), _ptr(video), _ptr(dW), _ptr(dtemporal), R, F,
                                          K, D, _stream())
    _lib.check(rc, "fvqa_visual_proj_bwd")


def embed_splice(ids, emb, vf_tok, h, n_seq, S, F, *, vstart: int = 0, zero_labels=None, index=None, mode: int = 0):
    _dev(ids, emb, vf_tok, h, zero_labels, index)
    D = emb.shape[1]
    _need(ids.dtype == torch.int64 and ids.numel() == n_seq * S, "embed_splice: ids")
    _need(emb.dtype == vf_tok.dtype == h.dtype, "embed_splice: dtype")
    _need(vf_tok.numel() == n_seq * F * D and h.numel() >= n_seq * S * D and h.shape[-1] == D, "embed_splice: shapes")
    _need(zero_labels is None or (zero_labels.dtype == torch.int64 and zero_labels.numel() == n_seq * S),
          "embed_splice: zero_labels")
    _need(index is None or (index.dtype == torch.int64 and index.numel() == n_seq * F), "embed_splice: index")
    # ids must address the table: checked on the host copy by the caller (see model.py)
    rc = _lib.load(emb.dtype).fvqa_embed_splice(_ptr(ids), _ptr(emb), _ptr(vf_tok), _ptr(zero_labels), _ptr(index), _ptr(h),
                                       n_seq, S, D, F, vstart, mode, dt_code(h.dtype), _stream())
    _lib.check(rc, "fvqa_embed_splice")
    return h


def splice_bwd(dh, d_tok, n_seq, S, F, *, vstart: int = 0, index=None, mode: int = 0):
    _dev(dh, d_tok, index)
    D = dh.shape[-1]
    _need(dh.numel() >= n_seq * S * D, "splice_bwd: dh")
    _need(d_tok.dtype == torch.float32 and d_tok.numel() == n_seq * F * D, "splice_bwd: d_tok")
    _need(index is None or (index.dtype == torch.int64 and index.numel() == n_seq * F), "splice_bwd: index")
    rc = _lib.load(dh.dtype).fvqa_splice_bwd(_ptr(dh), _ptr(index), _ptr(d_tok), n_seq, S, D, F, vstart, mode,
                                     dt_code(dh.dtype), _stream())
    _lib.check(rc, "fvqa_splice_bwd")


def ce_fwd(logits, labels, lse, rowloss, loss_sum, n_seq, S, V, ignore_index: int):
    _dev(logits, labels, lse, rowloss, loss_sum)
    _need(logits.dtype == torch.float32 and logits.numel() == n_seq * S * V, "ce_fwd: logits")
    _need(labels.dtype == torch.int64 and labels.numel() == n_seq * S, "ce_fwd: labels")
    for t in (lse, rowloss):
        _need(t.dtype == torch.float32 and t.numel() >= n_seq * S, "ce_fwd: lse/rowloss")
    _need(loss_sum.dtype == torch.float32 and loss_sum.numel() >= 2, "ce_fwd: loss_sum")
    rc = _lib.load().fvqa_ce_fwd(_ptr(logits), _ptr(labels), _ptr(lse), _ptr(rowloss), _ptr(loss_sum), n_seq, S, V,
                                 ignore_index, _stream())
    _lib.check(rc, "fvqa_ce_fwd")


def ce_bwd(logits, labels, lse, loss_sum, gscale, dlogits, n_seq, S, V, ignore_index: int):
    _dev(logits, labels, lse, loss_sum, gscale, dlogits)
    _need(logits.dtype == torch.float32 and logits.numel() == n_seq * S * V, "ce_bwd: logits")
    _need(dlogits.numel() == n_seq * S * V, "ce_bwd: dlogits")
    _need(labels.dtype == torch.int64 and labels.numel() == n_seq * S, "ce_bwd: labels")
    _need(gscale.dtype == torch.float32 and gscale.numel() >= 1 and loss_sum.numel() >= 2, "ce_bwd: scalars")
    rc = _lib.load(dlogits.dtype).fvqa_ce_bwd(_ptr(logits), _ptr(labels), _ptr(lse), _ptr(loss_sum), _ptr(gscale), _ptr(dlogits),
                                 n_seq, S, V, ignore_index, dt_code(dlogits.dtype), _stream())
    _lib.check(rc, "fvqa_ce_bwd")


def qav_head_fwd(xn, vf_raw, labels, probs, rowloss, loss_sum, n_seq, S, D, F, tau: float):
    _dev(xn, vf_raw, labels, probs, rowloss, loss_sum)
    _need(xn.numel() >= n_seq * S * D and xn.shape[-1] == D, "qav_head_fwd: xn")
    _need(vf_raw.dtype == torch.float32 and vf_raw.numel() == n_seq * F * D, "qav_head_fwd: vf_raw")
    _need(labels.dtype == torch.int64 and labels.numel() == n_seq * S, "qav_head_fwd: labels")
    _need(probs.dtype == torch.float32 and probs.numel() >= n_seq * S * F, "qav_head_fwd: probs")
    _need(rowloss.dtype == torch.float32 and rowloss.numel() >= n_seq * S, "qav_head_fwd: rowloss")
    rc = _lib.load(xn.dtype).fvqa_qav_head_fwd(_ptr(xn), _ptr(vf_raw), _ptr(labels), _ptr(probs), _ptr(rowloss),
                                       _ptr(loss_sum), n_seq, S, D, F, float(tau), dt_code(xn.dtype), _stream())
    _lib.check(rc, "fvqa_qav_head_fwd")


def qav_head_bwd(xn, vf_raw, labels, probs, loss_sum, gscale, dxn, d_raw, n_seq, S, D, F, tau: float):
    _dev(xn, vf_raw, labels, probs, loss_sum, gscale, dxn, d_raw)
    _need(xn.numel() >= n_seq * S * D and dxn.numel() >= n_seq * S * D and dxn.dtype == xn.dtype, "qav_head_bwd: xn")
    _need(vf_raw.numel() == n_seq * F * D and d_raw.dtype == torch.float32 and d_raw.numel() == n_seq * F * D,
          "qav_head_bwd: vf")
    _need(labels.dtype == torch.int64 and labels.numel() == n_seq * S, "qav_head_bwd: labels")
    rc = _lib.load(xn.dtype).fvqa_qav_head_bwd(_ptr(xn), _ptr(vf_raw), _ptr(labels), _ptr(probs), _ptr(loss_sum),
                                       _ptr(gscale), _ptr(dxn), _ptr(d_raw), n_seq, S, D, F, float(tau),
                                       dt_code(xn.dtype), _stream())
    _lib.check(rc, "fvqa_qav_head_bwd")


# ------------------------------------------------------------------------------------ optimizer
def grad_norm_workspace(n_seg: int) -> int:
    return int(_lib.load().fvqa_grad_norm_workspace(n_seg))


def grad_unscale_norm(grad, seg_off, scale, seg_sq, found_inf, total_norm, workspace, grad_div: float = 1.0,
                      gemm_err: Optional[torch.Tensor] = None, err_lane: Optional[torch.Tensor] = None):
    """grad_div: replicas summed into `grad` (the data-parallel mean is applied here); gemm_err: the 8-byte error word of
    the GEMM workspace (gemm_error_word) — non-zero makes found_inf 2 and the optimizer step a no-op; err_lane: the fp32
    element data-parallel ranks all-reduce with the gradients (FlatParams.err_lane) — non-zero does the same on EVERY rank."""
    _dev(grad, seg_off, scale, seg_sq, found_inf, total_norm, workspace, gemm_err, err_lane)
    _need(err_lane is None or (err_lane.dtype == torch.float32 and err_lane.numel() >= 1), "grad_unscale_norm: err_lane")
    _need(grad_div >= 1.0, "grad_unscale_norm: grad_div")
    _need(gemm_err is None or gemm_err.numel() * gemm_err.element_size() >= 8, "grad_unscale_norm: gemm_err")
    n_seg = seg_off.numel() - 1
    _need(grad.dtype == torch.float32 and seg_off.dtype == torch.int64 and n_seg >= 1, "grad_unscale_norm: types")
    _need(seg_sq.numel() >= n_seg and seg_sq.dtype == torch.float32, "grad_unscale_norm: seg_sq")
    wbytes = workspace.numel() * workspace.element_size()
    rc = _lib.load().fvqa_grad_unscale_norm(_ptr(grad), _ptr(seg_off), n_seg, _ptr(scale), float(grad_div),
                                            _ptr(gemm_err), _ptr(err_lane), _ptr(seg_sq), _ptr(found_inf), _ptr(total_norm),
                                            _ptr(workspace), wbytes, _stream())
    _lib.check(rc, "fvqa_grad_unscale_norm")


def adamw_step(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step, found_inf=None):
    _dev(param, grad, exp_avg, exp_avg_sq, step, found_inf)
    n = param.numel()
    for t in (param, grad, exp_avg, exp_avg_sq):
        _need(t.dtype == torch.float32 and t.numel() == n, "adamw_step: fp32 tensors of equal size")
    rc = _lib.load().fvqa_adamw_step(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), n, float(lr),
                                     float(beta1), float(beta2), float(eps), float(weight_decay), _ptr(step),
                                     _ptr(found_inf), _stream())
    _lib.check(rc, "fvqa_adamw_step")


def scaler_update(step, scale, tracker, found_inf, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
    _dev(step, scale, tracker, found_inf)
    rc = _lib.load().fvqa_scaler_update(_ptr(step), _ptr(scale), _ptr(tracker), _ptr(found_inf),
                                        float(growth_factor), float(backoff_factor), int(growth_interval), _stream())
    _lib.check(rc, "fvqa_scaler_update")
