#!/usr/bin/env python3
"""In-loop shader clock of the persistent projection GEMM (MI355X_MICROARCH.md "DVFS give-back" item 6).

Needs the diagnostic build of the library (`-DFVQA_SK_CLOCK`: s_memtime / s_memrealtime stamped around the ring loop of
every workgroup's first segment, one slice of a 512-launch ring per launch):

    python tools/sk_clock.py --build            # here (hipcc cross-compiles): writes tools/bin/libfvqa_clock.so
    python tools/sk_clock.py [--mode step|b2b]  # on the GPU box (it loads that build through FVQA_LIB)

mode step: the benchmarked C2 training step (bench.py's model and batches) runs back to back for >= --seconds, then the
           stamps of the LAST step's launches are read: per shape x instantiation, median over workgroups and launches of
           clock = d(memtime) / d(memrealtime) x 100 MHz, loop time and the FLOP rate inside the loop.
mode b2b : each C2 shape alone, launched back to back for >= --seconds (the guide's recipe verbatim).
Stamp values leave the kernel only through the stamp ring; the numbers of this build are never quoted as run times.
"""
import argparse
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "flipped-vqa_amd")
sys.path.insert(0, PKG)
sys.path.insert(0, ROOT)
CLOCK_LIB = os.path.join(ROOT, "tools", "bin", "libfvqa_clock.so")      # (git-ignored; NOT beside the product library)
RING = 512


def build():
    from fvqa import build as fb
    os.makedirs(os.path.dirname(CLOCK_LIB), exist_ok=True)
    print(fb.build(out=CLOCK_LIB, defines=["FVQA_SK_CLOCK"]))


def read_ring(ws, need):
    import torch
    n = RING * 256 * 16
    st = ws[need - n * 8:need].view(torch.int64).view(RING, 256, 16).cpu()
    return st


def summarise(st, min_epoch, label=""):
    """st: (RING, 256, 16) int64. One line per (M, N, K, epi, split, out-bytes): launches, clock, loop us, in-loop TF/s."""
    import torch
    rows = {}
    for r in range(RING):
        blk = st[r]
        used = blk[:, 1] > 0
        if not bool(used.any()):
            continue
        ep = int(blk[used][:, 5].max())                  # a slice is reused every RING launches: this launch's rows only
        if ep < min_epoch:
            continue
        used = used & (blk[:, 5] == ep)
        b = blk[used]
        N, K = int(b[0, 4]) >> 32, int(b[0, 4]) & 0xFFFFFFFF
        meta = int(b[0, 6])
        epi, split, nw, nbt, ob, nwg = (meta & 0xFF, (meta >> 8) & 0xFF, (meta >> 16) & 0xFFFF, (meta >> 32) & 0xFF,
                                        (meta >> 40) & 0xFF, (meta >> 48))
        nbt = nbt or 16                                   # gemm_sk_256: 256-column tiles; gemm4w_k: 16 * NBT columns
        M = int(b[0, 7])
        dt = (b[:, 2] - b[:, 0]).double()
        dr = (b[:, 3] - b[:, 1]).double()
        ok = dr > 0
        clk = (dt[ok] / dr[ok] * 0.1)                       # GHz
        us = dr[ok] / 100.0
        key = (M, N, K, epi, split, ob, nbt)
        e = rows.setdefault(key, {"n": 0, "clk": [], "us": [], "nw": nw, "wgs": int(used.sum()), "nwg": nwg})
        e["n"] += 1
        e["clk"].append(float(clk.median()))
        e["us"].append(float(us.median()))
        e.setdefault("clk_min", []).append(float(clk.min()))
        e.setdefault("clk_max", []).append(float(clk.max()))
        if split > 1 and int(b[0, 14]) > 0:
            # hand-off phases of gemm4w_sk_k (100 MHz stamps of wave 0, words 8-14): loop end -> publish issued -> stores drained ->
            # barrier -> partners' flags seen -> partials arrived -> added -> tile stored
            t = torch.stack([b[:, 3], b[:, 8], b[:, 9], b[:, 10], b[:, 11], b[:, 12], b[:, 13], b[:, 14]], 1).double()
            good = (t[:, 1:] >= t[:, :-1]).all(1)
            if bool(good.any()):
                d = (t[good][:, 1:] - t[good][:, :-1]) / 100.0
                e.setdefault("xchg", []).append([float(x) for x in d.median(0).values])
    epi_name = {0: "none", 1: "residual", 3: "swiglu_bwd", 4: "swiglu_fwd", 5: "swiglu_fwd_st", 6: "swiglu_bwd_st", 7: "rope"}
    print(f"# {label}: per shape — launches read, workgroups stamped, wide stages in the first segment, in-loop clock GHz "
          f"(median of per-launch medians; min / max over workgroups), loop us, TF/s inside the loop (all workgroups)")
    tot_flop = tot_cyc = 0.0
    for key in sorted(rows):
        M, N, K, epi, split, ob, nbt = key
        e = rows[key]
        import statistics as S
        clk, us = S.median(e["clk"]), S.median(e["us"])
        # FLOPs inside the stamped loops: every stamped workgroup runs a 256 x (16 nbt) x (nw * 64) product
        fl = e["wgs"] * 2.0 * 256 * 16 * nbt * e["nw"] * 64
        print(f"{M:5d} x {N:6d} x {K:6d} {epi_name.get(epi, epi):14s} split {split} tile 256x{16 * nbt:3d} out{ob}B  n={e['n']:3d} wgs={e['wgs']:3d} "
              f"nw={e['nw']:4d}  clock {clk:5.3f} GHz ({min(e['clk_min']):.3f} / {max(e['clk_max']):.3f})  loop {us:7.1f} us  "
              f"{fl / us / 1e6:7.0f} TF/s  -> MFMA pipe busy in-loop {fl / us / 1e6 / (256 * 4096 * clk * 1e-3) * (256 / e['wgs']):.3f} of the busy CUs' cycles")
        if e.get("xchg"):
            import numpy as np
            m = np.median(np.array(e["xchg"]), 0)
            print("        hand-off after the loop, us (median over workgroups and launches): publish issued %.2f | stores drained %.2f | barrier %.2f | "
                  "flag + poll + barrier %.2f | partners' partials arrived %.2f | added %.2f | rows stored %.2f  (sum %.2f)" % (*m, m.sum()))
        tot_flop += fl * e["n"]
        tot_cyc += us * e["n"]
    if tot_cyc:
        print(f"# all stamped loops: {tot_flop / tot_cyc / 1e6:.0f} TF/s inside the loops")


def per_xcd(st, min_epoch):
    """Is the spread of the loop times over workgroups systematic? Per stamped launch of the long split-K shapes: median loop time and
    clock of the 32 workgroups of each XCD chunk (work ids 32 x .. 32 x + 31 run on one XCD)."""
    import torch
    print("# per-XCD medians (loop us @ clock GHz) of the last launches of the K >= 11008 split shapes; rows = launches")
    for r in range(RING):
        blk = st[r]
        used = blk[:, 1] > 0
        if not bool(used.any()):
            continue
        ep = int(blk[used][:, 5].max())
        if ep < min_epoch:
            continue
        meta = int(blk[used][0, 6])
        split = (meta >> 8) & 0xFF
        K = int(blk[used][0, 4]) & 0xFFFFFFFF
        N = int(blk[used][0, 4]) >> 32
        if split < 2 and os.environ.get("SK_CLOCK_XCD") != "all":
            continue
        if split >= 2 and K < 11008:
            continue
        ok = used & (blk[:, 5] == ep)
        if int(ok.sum()) < 250:
            continue
        dr = (blk[:, 3] - blk[:, 1]).double() / 100.0
        clk = (blk[:, 2] - blk[:, 0]).double() / (blk[:, 3] - blk[:, 1]).double().clamp(min=1) * 0.1
        cells = [f"{float(dr[32 * x:32 * x + 32].median()):6.1f}@{float(clk[32 * x:32 * x + 32].median()):.2f}" for x in range(8)]
        print(f"  N={N:6d} K={K:6d} split {split} epoch {ep:8d}: " + " ".join(cells) + f"   spread {float(dr.max() - dr.min()):5.1f} us")


def mode_step(seconds):
    import contextlib
    import torch
    import util.misc as misc
    from fvqa import ops, synth, _lib
    from fvqa.optim import FusedAdamW, param_groups_weight_decay
    from llama_vqa import LLaMA_VQA
    os.environ.setdefault("FVQA_SYNTHETIC_TOKENIZER", "1")
    dev = torch.device("cuda", 0)
    args = types.SimpleNamespace(
        llama_model_path="/nonexistent/", model="7B", max_seq_len=128, adapter_len=10, adapter_layer=32, max_feats=10,
        bias=3.5, tau=100.0, vaq=False, qav=False, audio=False, audio_only=False, audio_merge="none", debug=False,
        synthetic=True, random_init=True, dtype="bf16", accum_iter=1, weight_decay=0.14)
    with contextlib.redirect_stdout(sys.stderr):
        model = LLaMA_VQA(args)
    model.to(dev)
    p = model.params
    opt = FusedAdamW(param_groups_weight_decay(model, 0.14), lr=9e-2 * 8 / 256, betas=(0.9, 0.95), flat=model.flat_params())
    scaler = misc.NativeScalerWithGradNormCount()
    cfg = synth.SynthConfig(dim=p.dim, n_heads=p.n_heads, n_layers=p.n_layers, vocab_size=model.vocab_size,
                            max_seq_len=128, batch_size=8, vaq=False, qav=False)
    batches = []
    for i in range(4):
        b = synth.make_batch(cfg, seed=1234 + i)
        b["video"] = b["video"].to(dev)
        for k in ("text_id", "label", "video_index"):
            b[k] = {t: v.to(dev) for t, v in b[k].items()}
        batches.append(b)

    def one_step(i):
        opt.zero_grad()
        vqa, vaq, qav = model(batches[i % 4])
        scaler(vqa + vaq + qav, opt, parameters=None, update_grad=True)

    for i in range(3):
        one_step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(10):
            one_step(n)
            n += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"# {n} steps in {dt:.2f} s = {dt / n * 1e3:.3f} ms/step (diagnostic build: its stamps fence the loop; not a timing)")
    need = int(_lib.load().fvqa_gemm_sk_workspace())
    ws = ops.gemm_workspace(dev, need)
    st = read_ring(ws, need)
    last = int(st[:, :, 5].max())
    summarise(st, last - 257, "C2 step, last 258 launches (gemm4w_sk_k or gemm_sk_256: split 2/4; gemm4w_k: split 1, first tile of a workgroup)")
    if os.environ.get("SK_CLOCK_XCD"):
        per_xcd(st, last - 257)


def mode_b2b(seconds):
    import torch
    from fvqa import ops, _lib
    dev = "cuda"
    shapes = [("qkv_fwd", 1024, 12288, 4096), ("wo_fwd", 1024, 4096, 4096), ("w13_fwd", 1024, 22016, 4096),
              ("w2_fwd", 1024, 4096, 11008), ("w2t_bwd", 1024, 11008, 4096), ("w13t_bwd", 1024, 4096, 22016),
              ("qkvt_bwd", 1024, 4096, 12288)]
    need = int(_lib.load().fvqa_gemm_sk_workspace())
    for name, M, N, K in shapes:
        for fill in ("random", "zeros"):
            if fill == "random":
                a = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16()
                b = ((torch.rand(N, K, device=dev) * 2 - 1) / K ** 0.5).bfloat16()
            else:
                a = torch.zeros(M, K, device=dev, dtype=torch.bfloat16)
                b = torch.zeros(N, K, device=dev, dtype=torch.bfloat16)
            o = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            ops.gemm_nt(a, b, o, variant=13)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 0
            lim = seconds if fill == "random" else min(seconds, 1.0)
            while time.perf_counter() - t0 < lim:
                for _ in range(200):
                    ops.gemm_nt(a, b, o, variant=13)
                n += 200
                if n % 2000 == 0:
                    torch.cuda.synchronize()          # keep the launch queue bounded
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            ws = ops.gemm_workspace(torch.device(dev, 0), need)
            st = read_ring(ws, need)
            last = int(st[:, :, 5].max())
            summarise(st, last - 63, f"{name} {fill}, {n} launches back to back in {dt:.2f} s ({dt / n * 1e6:.1f} us per launch incl. host)")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--mode", default="step", choices=["step", "b2b", "both"])
    ap.add_argument("--seconds", type=float, default=3.0)
    a = ap.parse_args()
    if a.build:
        build()
        sys.exit(0)
    if not os.path.exists(CLOCK_LIB):
        raise SystemExit(f"{CLOCK_LIB} missing: run `python tools/sk_clock.py --build` first")
    os.environ["FVQA_LIB"] = CLOCK_LIB
    if a.mode in ("step", "both"):
        mode_step(a.seconds)
    if a.mode in ("b2b", "both"):
        mode_b2b(a.seconds)
