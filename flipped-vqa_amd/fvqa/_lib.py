"""ctypes binding of libfvqa_hip.so (include/fvqa.h). No torch types cross this boundary:
only raw device pointers, sizes and a hipStream_t.

The product path has NO fallback: if the library is missing or does not export a declared
symbol, importing / calling raises FvqaLibraryError.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

F32, BF16, F16 = 0, 1, 2
EPI_NONE, EPI_RESIDUAL, EPI_SWIGLU_BWD, EPI_SWIGLU_BWD_ST = 0, 1, 3, 6
ABI_VERSION = 16

_p, _i, _f, _i64, _sz = C.c_void_p, C.c_int, C.c_float, C.c_int64, C.c_size_t

# name -> (restype, argtypes); must list every entry point of include/fvqa.h
SIGNATURES = {
    "fvqa_version": (_i, []),
    "fvqa_source_hash": (C.c_char_p, []),
    "fvqa_arch": (C.c_char_p, []),
    "fvqa_gemm_nt": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _sz, _p]),
    "fvqa_gemm_workspace": (_sz, [_i, _i, _i, _i]),
    "fvqa_gemm_sk_workspace": (_sz, []),
    "fvqa_gemm_nt_swiglu_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _sz, _p]),
    "fvqa_gemm_nt_swiglu_fwd_st": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _sz, _p]),
    "fvqa_gemm_nt_swiglu_fwd_st_rider": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p, _sz, _p]),
    "fvqa_kv_rider_ahead": (_i, [_i]),
    "fvqa_swiglu_st": (_i, []),
    "fvqa_gemm_nt_rider": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _sz, _p]),
    "fvqa_gemm_sk_describe": (_i, [_i, _i, _i, _i, _i, _p, _i, _p, _i]),
    "fvqa_gemm4w_choose": (_i, [_i, _i, _i, _i, _i, _i, _p, _i]),
    "fvqa_gemm4w_force": (_i, [_i]),
    "fvqa_gemm_timing_enable": (_i, [_i]),
    "fvqa_gemm_timing_read": (_i, [_i, _p, _p, _p]),
    "fvqa_rmsnorm_fwd": (_i, [_p, _p, _p, _p, _i, _i, _f, _i, _p]),
    "fvqa_rmsnorm_bwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "fvqa_rope_qk": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "fvqa_swiglu_fwd": (_i, [_p, _p, _i, _i, _i, _p]),
    "fvqa_swiglu_bwd": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "fvqa_attn_rope_fused": (_i, [_i]),
    "fvqa_rope_in_gemm": (_i, [_i]),
    "fvqa_gemm_nt_rope": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p, _p, _sz, _p]),
    "fvqa_attn_fwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "fvqa_attn_decode": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "fvqa_attn_bwd_workspace": (_sz, [_i, _i, _i, _i, _i]),
    "fvqa_attn_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _i, _i, _i, _i, _i, _i, _i, _p]),
    "fvqa_attn_bwd_rotated": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _i, _i, _i, _i, _i, _i, _i, _p]),
    "fvqa_visual_proj_fwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "fvqa_visual_proj_bwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "fvqa_embed_splice": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "fvqa_splice_bwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "fvqa_ce_fwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i64, _p]),
    "fvqa_ce_bwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i64, _i, _p]),
    "fvqa_qav_head_fwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _i, _p]),
    "fvqa_qav_head_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _i, _p]),
    "fvqa_grad_norm_workspace": (_sz, [_i]),
    "fvqa_grad_unscale_norm": (_i, [_p, _p, _i, _p, _f, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "fvqa_adamw_step": (_i, [_p, _p, _p, _p, _i64, _f, _f, _f, _f, _f, _p, _p, _p]),
    "fvqa_scaler_update": (_i, [_p, _p, _p, _p, _f, _f, _i, _p]),
    "fvqa_cast_rows": (_i, [_p, _p, _i, _i, _i, _p]),
    "fvqa_gather_rows": (_i, [_p, _p, _p, _i, _i, _p]),
    "fvqa_scatter_rows": (_i, [_p, _p, _p, _i, _i, _p]),
    "fvqa_layers_gemm_workspace": (_sz, [_p]),
    "fvqa_layers_fwd": (_i, [_p, _p]),
    "fvqa_layers_bwd": (_i, [_p, _p, _p, _p]),
}

_pp = C.POINTER(C.c_void_p)


class SkRider(C.Structure):
    """Mirror of `fvqa_sk_rider` (include/fvqa.h)."""
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p)] + \
               [(n, C.c_int32) for n in ("M", "N", "K", "lda", "ldb", "ldc", "accumulate_f32")]


class SkRope(C.Structure):
    """Mirror of `fvqa_sk_rope` (include/fvqa.h)."""
    _fields_ = [("cos_t", C.c_void_p), ("sin_t", C.c_void_p)] + [(n, C.c_int32) for n in ("seq_len", "head_dim", "cols")]


class RowSegs(C.Structure):
    """Mirror of `fvqa_row_segs` (include/fvqa.h)."""
    _fields_ = [("n", C.c_int32), ("stream_rows", C.c_int32), ("off", C.c_int32 * 4), ("map", C.c_void_p * 3)]


class TailRows(C.Structure):
    """Mirror of `struct fvqa_tail_rows` inside `fvqa_layer_plan`."""
    _fields_ = ([("rows", C.c_int32), ("reserved_", C.c_int32), ("gather", RowSegs), ("scatter", RowSegs)]
                + [(n, C.c_void_p) for n in ("og", "xg", "h", "hn", "ab", "z", "xl", "xnf", "rstd2", "rstdN",
                                             "dcur", "dab", "dt", "dh", "d_o")])


class LayerPlan(C.Structure):
    """Mirror of `fvqa_layer_plan` (include/fvqa.h) — field order and types must match exactly."""
    _fields_ = (
        [(n, C.c_int32) for n in ("dtype", "n_layers", "n_seq", "seq_len", "n_heads", "head_dim", "adapter_len",
                                  "max_feats", "dim", "hidden")]
        + [("eps", C.c_float), ("reserved_", C.c_int32)]
        + [(n, _pp) for n in ("wqkv", "wo", "w13", "w2", "wqkv_t", "wo_t", "w13_t", "w2_t", "an", "fn", "gate1",
                              "gate2", "dgate1", "dgate2")]
        + [(n, C.c_void_p) for n in ("adapter", "adapter_c", "d_adapter", "norm_w", "xs", "rstd1", "rstd2", "qkv", "o", "lse_a",
                                     "lse_t", "h", "ab", "xn", "hn", "z", "xnf", "rstdN", "cos_t", "sin_t",
                                     "vstart", "dcur", "dnxt", "dz", "dab", "dh", "d_o", "dqkv", "attn_ws")]
        + [("attn_ws_bytes", C.c_size_t), ("gemm_ws", C.c_void_p), ("gemm_ws_bytes", C.c_size_t), ("tail", TailRows)]
    )


ERRORS = {-1: "FVQA_EINVAL (null pointer / bad enum)", -2: "FVQA_ESHAPE (unsupported dimension)",
          -3: "FVQA_EALIGN (misaligned pointer / short workspace)"}


class FvqaLibraryError(RuntimeError):
    pass


def kind_of(which=None) -> str:
    """"f16" for the fp16-storage library (libfvqa_hip_f16.so), "bf16" for libfvqa_hip.so (bf16 and exact-fp32 storage).
    `which`: None | "bf16" | "f16" | a dtype code (F32 / BF16 / F16) | a torch dtype."""
    if which is None or which == "bf16":
        return "bf16"
    if which == "f16" or which == F16 or str(which) == "torch.float16":
        return "f16"
    return "bf16"


def lib_path(which=None) -> str:
    here = os.path.dirname(os.path.abspath(__file__))
    if kind_of(which) == "f16":
        return os.environ.get("FVQA_LIB_F16", os.path.join(here, "libfvqa_hip_f16.so"))
    return os.environ.get("FVQA_LIB", os.path.join(here, "libfvqa_hip.so"))


_LIBS = {}
_LIB: Optional[C.CDLL] = None          # the bf16 / fp32 library once loaded (kept for callers that look at it)


def load(which=None) -> C.CDLL:
    """dlopen the library that serves `which` (kind_of) and bind every symbol of include/fvqa.h; raises if anything is missing.
    Both libraries are the same sources and the same ABI; they differ in the 16-bit storage type their kernels are built for and
    reject the other's dtype code, so a tensor of the wrong 16-bit type fails loudly instead of being reinterpreted."""
    global _LIB
    kind = kind_of(which)
    lib = _LIBS.get(kind)
    if lib is not None:
        return lib
    path = lib_path(kind)
    if not os.path.exists(path):
        raise FvqaLibraryError(
            f"{path} not found: build it with `python -m fvqa.build` (hipcc --offload-arch=gfx950). "
            "There is no CPU or PyTorch fallback for the Flipped-VQA hot path.")
    # The library must share ONE HIP runtime with whoever owns the device memory it is handed. Under PyTorch that is the
    # libamdhip64 bundled in torch/lib: torch goes first — a library loaded before it pulls /opt/rocm's copy in, the process then
    # holds two runtimes and every kernel launch of this library fails with hipErrorNoDevice (seen: build() then smoke() in one
    # process). A host without torch (a C caller of include/fvqa.h) links the system runtime and is not affected.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    try:
        lib = C.CDLL(path)
    except OSError as e:
        raise FvqaLibraryError(f"cannot load {path}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise FvqaLibraryError(f"{path} does not export {name}; rebuild it") from e
        fn.restype = res
        fn.argtypes = args
    v = lib.fvqa_version()
    if v != ABI_VERSION:
        raise FvqaLibraryError(f"{path}: ABI version {v}, host expects {ABI_VERSION}")
    check_source_hash(lib, path)
    _LIBS[kind] = lib
    if kind == "bf16":
        _LIB = lib
    return lib


def check_source_hash(lib, path: str) -> None:
    """The library is git-ignored and travels as a binary: refuse one that was built from other kernel sources than the ones
    next to it (a stale build with the same ABI number would otherwise be loaded — and tested — silently)."""
    from . import build
    if not os.path.isdir(build.CSRC):
        return                                   # a deployed library without its sources: nothing to compare with
    have = (lib.fvqa_source_hash() or b"").decode()
    want = build.source_hash()
    if have != want:
        raise FvqaLibraryError(
            f"{path} was built from kernel sources {have[:12] or '?'}, the sources here are {want[:12]}: stale binary — "
            "rebuild it (`python -m fvqa.build`, or __graft_entry__.build())")


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = ERRORS.get(rc, f"hipError_t {-(rc + 1000)}" if rc <= -1000 else str(rc))
        raise RuntimeError(f"{what} failed: {msg}")
