"""Builds libfvqa_hip.so (the gfx950 kernels + C ABI of include/fvqa.h) in-tree with hipcc.

`python -m fvqa.build` from flipped-vqa_amd/, or fvqa.build.build() from Python. hipcc
cross-compiles for gfx950 without a GPU; the .so lands next to this file so that it travels
with the source tree (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "csrc")
INCLUDE = os.path.join(os.path.dirname(os.path.dirname(HERE)), "include")
LIB = os.path.join(HERE, "libfvqa_hip.so")              # bf16 storage (+ the exact-fp32 build)
LIB_F16 = os.path.join(HERE, "libfvqa_hip_f16.so")      # the same sources with IEEE fp16 as the 16-bit storage type (-DFVQA_H16_F16)
OBJDIR = os.path.join(CSRC, "build")
OBJDIR_F16 = os.path.join(CSRC, "build", "f16")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-I", INCLUDE, "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC=)")


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def source_hash() -> str:
    """sha256 over the kernel sources (csrc/*.hip, csrc/*.h, include/fvqa.h): tags measurements (profiles/*pmc*.json)
    with the code they were taken on, independently of where or when the library was compiled."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    for f in files + [os.path.join(INCLUDE, "fvqa.h")]:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _content_key(src: str, headers, flags) -> str:
    """What an object was compiled from: sha256 of its source, every header it may include and the flags. Objects are chosen
    for recompilation by CONTENT, not by mtime (round-4 advisor finding: rsync -t, tar, git checkout of older files or clock
    skew between the build box and the GPU box leave a changed source older than its object — the library would then carry
    the new source hash over old kernels and the load-time stale-binary guard would accept it)."""
    import hashlib
    h = hashlib.sha256(" ".join(flags).encode())
    for f in [src] + sorted(headers):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


LAST_BUILD = {"compiled": 0, "linked": False}        # what the last build() call did (tests, __graft_entry__.build)


def build(force: bool = False, verbose: bool = True, out: str = LIB, defines=()) -> str:
    """Builds BOTH libraries — libfvqa_hip.so (bf16 + fp32) and libfvqa_hip_f16.so (fp16 + fp32: every source compiled a second
    time with -DFVQA_H16_F16, csrc/common.h) — and returns the path of the first.
    defines: extra -D macros (tuning builds write to a different `out`, always from scratch).
    Every build compiles the hash of the kernel sources into the library (csrc/version.hip, -DFVQA_SOURCE_HASH), which
    fvqa/_lib.py checks at load time: an old binary next to newer sources does not load."""
    if defines:
        return _build_variant(out, defines, verbose)
    n1, l1 = _build_one(force, verbose, LIB, OBJDIR, [])
    n2, l2 = _build_one(force, verbose, LIB_F16, OBJDIR_F16, ["-DFVQA_H16_F16"])
    LAST_BUILD.update(compiled=n1 + n2, linked=l1 or l2)
    return LIB


def _build_one(force, verbose, LIB, OBJDIR, extra):
    FLAGS = globals()["FLAGS"] + list(extra)
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()
    shash = source_hash()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(INCLUDE, "fvqa.h"))
    jobs, objs, keys = [], [], {}
    for src in sources():
        obj = os.path.join(OBJDIR, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if os.path.basename(src) == "version.hip":           # carries the source hash: rebuilt whenever the hash moves
            stamp = obj + ".hash"
            if force or not os.path.exists(obj) or not os.path.exists(stamp) or open(stamp).read() != shash:
                jobs.append([hipcc, *FLAGS, f'-DFVQA_SOURCE_HASH="{shash}"', "-c", src, "-o", obj])
            continue
        key = _content_key(src, headers, FLAGS)
        keyf = obj + ".key"
        if force or not os.path.exists(obj) or not os.path.exists(keyf) or open(keyf).read() != key:
            if os.path.exists(keyf):
                os.remove(keyf)                                   # (a failed compile must not leave a matching key behind)
            jobs.append([hipcc, *FLAGS, "-c", src, "-o", obj])
            keys[obj] = key

    def run(cmd):
        if verbose:
            print("[fvqa.build]", " ".join(cmd[-3:]), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    open(os.path.join(OBJDIR, "version.o.hash"), "w").write(shash)
    for obj, key in keys.items():
        open(obj + ".key", "w").write(key)
    link = bool(jobs) or force or _stale(LIB, objs)
    if link:
        run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs])
    if verbose:
        print(f"[fvqa.build] {os.path.basename(LIB)}: {len(jobs)} of {len(objs)} objects compiled, library "
              f"{'linked' if link else 'up to date'}, sources {shash[:12]}", flush=True)
    return len(jobs), link


def _build_variant(out, defines, verbose):
    hipcc = _hipcc()
    cmd = [hipcc, *FLAGS, f'-DFVQA_SOURCE_HASH="{source_hash()}"', *[f"-D{d}" for d in defines], "-shared", "-o", out,
           *sources()]
    if verbose:
        print("[fvqa.build]", " ".join(cmd[-8:]), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed\n{r.stdout}\n{r.stderr}")
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
