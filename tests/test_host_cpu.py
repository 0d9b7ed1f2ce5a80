"""Host-side logic and the C-ABI surface, no GPU needed."""
import ctypes as C
import math
import os
import re
import types

import pytest
import torch

from fvqa import _lib, synth
from fvqa.optim import param_groups_weight_decay
from fvqa.parallel import shard_indices
from util import lr_sched
import llama
import llama_vqa
import util.misc as misc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_exports_every_declared_symbol():
    """include/fvqa.h <-> ctypes table <-> the built .so agree symbol for symbol."""
    text = open(os.path.join(ROOT, "include", "fvqa.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = set(re.findall(r"\b(fvqa_[a-z0-9_]+)\s*\(", text))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()                                   # binds all of them or raises
    assert lib.fvqa_version() == _lib.ABI_VERSION
    assert lib.fvqa_arch() == b"gfx950"
    # argument validation happens before any launch: callable without a GPU
    assert lib.fvqa_gemm_nt(None, None, None, None, None, 1, 1, 64, 64, 64, 1, 1, 1, 1, 0, 0, None, 0, None) == -1
    # the fp16-storage build of the same sources (libfvqa_hip_f16.so): same ABI, same symbols, same source hash; each library
    # serves its own 16-bit dtype code and refuses the other's before anything is launched
    lib16 = _lib.load("f16")
    assert lib16 is not lib and _lib.load(_lib.F16) is lib16 and _lib.load(_lib.BF16) is lib and _lib.load(_lib.F32) is lib
    assert lib16.fvqa_version() == _lib.ABI_VERSION and lib16.fvqa_source_hash() == lib.fvqa_source_hash()
    assert lib.fvqa_attn_rope_fused(_lib.BF16) == 1 and lib.fvqa_attn_rope_fused(_lib.F16) == 0
    assert lib16.fvqa_attn_rope_fused(_lib.F16) == 1 and lib16.fvqa_attn_rope_fused(_lib.BF16) == 0
    one = C.c_void_p(256)                               # (a non-null pointer: the dtype check comes first, nothing is read)
    for L_, ok, bad in ((lib, _lib.BF16, _lib.F16), (lib16, _lib.F16, _lib.BF16)):
        assert L_.fvqa_rmsnorm_fwd(one, one, one, None, 1, 7, 1e-6, bad, None) == -1        # FVQA_EINVAL: not this build's type
        assert L_.fvqa_rmsnorm_fwd(one, one, one, None, 1, 7, 1e-6, ok, None) == -2         # FVQA_ESHAPE: type accepted, dim % 8
    assert lib.fvqa_attn_bwd_workspace(2, 128, 32, 128, 10) > 0
    assert lib.fvqa_attn_bwd_workspace(2, 128, 32, 64, 10) == 0      # head_dim != 128 unsupported


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setenv("FVQA_LIB", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_lib, "_LIBS", {})
    with pytest.raises(_lib.FvqaLibraryError):
        _lib.load()


def _tiny_model():
    cfg = synth.preset("tiny", vaq=True, qav=True)
    args = types.SimpleNamespace(max_feats=10, bias=3.5, tau=100.0, llama_model_path="/nonexistent/", vaq=True,
                                 qav=True, synthetic=True, vocab_size=cfg.vocab_size, audio=False)
    ma = llama.ModelArgs(max_seq_len=cfg.max_seq_len, adapter_len=10, adapter_layer=2, **cfg.params_json())
    ma.vocab_size = cfg.vocab_size
    return cfg, llama.Transformer(ma, args)


def test_model_has_reference_state_dict_and_no_cpu_path():
    cfg, model = _tiny_model()
    want = {n: s for n, s, _ in synth.state_spec(cfg)}
    have = {n: tuple(p.shape) for n, p in model.named_parameters()}
    assert have == want
    assert llama.model.swiglu_hidden(4096, 256) == 11008 and llama.model.swiglu_hidden(5120, 256) == 13824
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model(synth.make_batch(cfg))
    with pytest.raises(AssertionError):                     # like the reference: tokenizer file required
        llama.Tokenizer("/nonexistent/tokenizer.model", args=types.SimpleNamespace())


def test_freeze_policy_and_param_groups():
    cfg, model = _tiny_model()
    for n, p in model.named_parameters():
        p.requires_grad = synth.is_trainable(n)
    no_decay, decay = param_groups_weight_decay(model, 0.14)
    names = {id(p): n for n, p in model.named_parameters()}
    got = sorted(names[id(p)] for p in decay["params"])
    assert no_decay["params"] == [] and no_decay["weight_decay"] == 0.0 and decay["weight_decay"] == 0.14
    assert got == sorted(n for n in names.values() if synth.is_trainable(n))
    n7b = sum(math.prod(s) for n, s, _ in synth.state_spec(synth.preset("7b")) if synth.is_trainable(n))
    assert n7b == 4_499_456                                # 18.0 MB fp32


def test_lr_schedule():
    a = types.SimpleNamespace(lr=0.1, min_lr=0.01, warmup_epochs=2, epochs=10)
    opt = types.SimpleNamespace(param_groups=[{"lr": 0.0}, {"lr": 0.0, "lr_scale": 0.5}])
    assert lr_sched.adjust_learning_rate(opt, 1.0, a) == pytest.approx(0.05)
    assert opt.param_groups[1]["lr"] == pytest.approx(0.025)
    assert lr_sched.adjust_learning_rate(opt, 2.0, a) == pytest.approx(0.1)
    assert lr_sched.adjust_learning_rate(opt, 6.0, a) == pytest.approx(0.01 + 0.09 * 0.5)
    assert lr_sched.adjust_learning_rate(opt, 10.0, a) == pytest.approx(0.01)


def test_synthetic_batch_schema():
    cfg = synth.preset("7b", vaq=True, qav=True, batch_size=4)
    b = synth.make_batch(cfg, seed=3)
    B, S, F = 4, 128, 10
    assert b["video"].shape == (B, F, 768) and b["video"].dtype == torch.float32
    for t in ("vqa", "vaq", "qav"):
        assert b["text_id"][t].shape == (B, 1, S) and b["text_id"][t].dtype == torch.int64
        assert b["label"][t].shape == (B, 1, S)
        assert 0 <= int(b["text_id"][t].min()) and int(b["text_id"][t].max()) < 32000
    vs = b["video_start"]["vqa"][0]
    assert (b["text_id"]["vqa"][:, 0, vs:vs + F] == 0).all()
    assert (b["label"]["vqa"] != 0).sum(-1).min() >= 1 and (b["label"]["vqa"][..., : vs + F] == 0).all()
    q = b["label"]["qav"][:, 0]
    idx = b["video_index"]["qav"]
    assert ((q >= 0).sum(-1) == F).all()
    assert torch.equal(q.gather(1, idx), torch.arange(F).repeat(B, 1))
    b2 = synth.make_batch(cfg, seed=3)
    assert torch.equal(b["video"], b2["video"]) and torch.equal(b["text_id"]["vaq"], b2["text_id"]["vaq"])


def test_closed_form_generator_is_stable():
    t = synth.hashed_uniform("layers.0.attention.wq.weight", (4, 8), 0.5)
    assert t.shape == (4, 8) and float(t.abs().max()) <= 0.5
    assert torch.equal(t, synth.hashed_uniform("layers.0.attention.wq.weight", (4, 8), 0.5))
    big = synth.hashed_uniform("x", (1 << 16,), 1.0)
    assert abs(float(big.mean())) < 0.02 and abs(float(big.var()) - 1 / 3) < 0.02


def test_merge_shards_follows_meta_split_dims():
    D, Hf, V = 8, 12, 10
    full = {"tok_embeddings.weight": torch.randn(V, D), "norm.weight": torch.randn(D), "output.weight": torch.randn(V, D),
            "layers.0.attention.wq.weight": torch.randn(D, D), "layers.0.attention.wo.weight": torch.randn(D, D),
            "layers.0.feed_forward.w1.weight": torch.randn(Hf, D), "layers.0.feed_forward.w2.weight": torch.randn(D, Hf),
            "layers.0.attention_norm.weight": torch.randn(D), "rope.freqs": torch.randn(4)}
    col = ("wq.weight", "w1.weight", "output.weight")
    row = ("wo.weight", "w2.weight", "tok_embeddings.weight")
    shards = [{}, {}]
    for n, t in full.items():
        for r in range(2):
            if n.endswith(col):
                shards[r][n] = t.chunk(2, 0)[r]
            elif n.endswith(row):
                shards[r][n] = t.chunk(2, 1)[r]
            else:
                shards[r][n] = t
    merged = llama_vqa.merge_shards(shards, 1)
    for n, t in full.items():
        if n == "rope.freqs":
            assert n not in merged
        else:
            assert torch.equal(merged[n], t), n
    assert llama_vqa.merge_shards([full], 1) is full


def test_merge_shards_equals_the_reference_merge(golden_dir):
    """tests/golden/ckpt_merge.npz: a synthetic 2-shard Meta checkpoint (fp16, with rope.freqs) and what the merge loop
    INSIDE the reference's llama_vqa.LLaMA_VQA (llama_vqa.py:24-58) made of it (oracle/gen_golden_ckpt.py ran that
    function itself). The product's merge must give the same keys, dtypes and bits."""
    import numpy as np
    g = dict(np.load(os.path.join(golden_dir, "ckpt_merge.npz")))
    keys = [str(k) for k in g["shard_keys"]]
    shards = [{k: torch.from_numpy(g[f"shard{r}__{k}"]) for k in keys} for r in range(2)]
    merged = llama_vqa.merge_shards(shards, int(g["n_layers"]))
    want = [str(k) for k in g["merged_keys"]]
    assert set(merged) == set(want) and "rope.freqs" not in merged
    for k in want:
        ref = torch.from_numpy(g[f"merged__{k}"])
        assert merged[k].dtype == ref.dtype == torch.float16 and merged[k].shape == ref.shape, k
        assert torch.equal(merged[k], ref), k


def test_shard_indices_match_distributed_sampler():
    from torch.utils.data import DistributedSampler
    ds = list(range(23))
    for epoch in (0, 3):
        for rank in range(4):
            s = DistributedSampler(ds, num_replicas=4, rank=rank, shuffle=True, seed=0)
            s.set_epoch(epoch)
            assert list(s) == shard_indices(23, rank, 4, epoch=epoch, shuffle=True, seed=0)


def test_metric_logger_and_scaler_state():
    log = misc.MetricLogger(delimiter="  ")
    log.add_meter("lr", misc.SmoothedValue(window_size=1, fmt="{value:.6f}"))
    seen = [x for x in log.log_every([1, 2, 3], 0, "Epoch: [0]")]      # print_freq 0 must not crash
    assert seen == [1, 2, 3]
    log.update(loss=2.0, lr=0.1)
    log.update(loss=4.0)
    assert log.meters["loss"].global_avg == 3.0 and "lr: 0.100000" in str(log)
    sc = misc.NativeScalerWithGradNormCount()
    sd = sc.state_dict()
    assert sd["scale"] == 65536.0 and set(sd) == {"scale", "growth_factor", "backoff_factor", "growth_interval", "_growth_tracker"}
    sc.load_state_dict({"scale": 128.0, "_growth_tracker": 5})
    assert sc.state_dict()["scale"] == 128.0
    assert misc.NativeScalerWithGradNormCount.state_dict_key == "amp_scaler"


def test_train_cli_has_reference_flags():
    import train
    p = train.get_args_parser()
    a = p.parse_args(["--model", "7B", "--max_seq_len", "128", "--batch_size", "8", "--vaq", "--qav", "--bias", "3.5",
                      "--tau", "100", "--blr", "9e-2", "--weight_decay", "0.14", "--accum_iter", "2", "--dataset", "nextqa"])
    for k in ("batch_size epochs accum_iter llama_model_path model adapter_layer adapter_len max_seq_len max_feats "
              "weight_decay lr blr min_lr warmup_epochs dataset output_dir device seed resume start_epoch num_workers "
              "pin_mem world_size local_rank dist_on_itp dist_url vaq qav bias tau sub is_generation_task debug jobid "
              "audio audio_only audio_merge").split():
        assert hasattr(a, k), k
    assert a.adapter_layer == 32 and a.adapter_len == 10 and a.bias == 3.5 and a.pin_mem is True
    train.validate_args(a)
    a.audio_only = True
    with pytest.raises(AssertionError):
        train.validate_args(a)


def test_bench_launcher_command_and_env():
    """bench.py --gpus N from a plain shell: the child command is the driver's own N>1 form, and a box with fewer
    devices than ranks turns the run into a gloo rehearsal (reference launch: run.sh:20-22 torchrun)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cmd = bench.launcher_command(4, ["--gpus", "4", "--steps", "5", "--warmup", "2"], port=29511, python="python3")
    assert cmd[:3] == ["python3", "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "5", "--warmup", "2"]
    c2 = bench.launcher_command(2, [])
    port = int(c2[c2.index("--master-port") + 1])
    assert 1024 < port < 65536
    env = bench.launcher_env(8, 8, env={})
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and "FVQA_DIST_BACKEND" not in env
    env = bench.launcher_env(2, 1, env={})
    assert env["FVQA_DIST_BACKEND"] == "gloo" and env["FVQA_BENCH_REHEARSAL"] == "1"
    env = bench.launcher_env(2, 0, env={})                     # no device visible: nothing to rehearse on
    assert "FVQA_DIST_BACKEND" not in env


def _sk_plan(M, N, K, dtype, n_cu=256):
    import ctypes as C
    lib = _lib.load()
    plan = (C.c_int32 * 12)()
    lib.fvqa_gemm_sk_describe(M, N, K, dtype, n_cu, C.cast(plan, C.c_void_p), -1, None, 0)
    keys = ("tm", "tn", "nw_tile", "gran", "gpt", "ts", "mgroups", "n_teams", "full", "rem", "s", "pstride")
    p = dict(zip(keys, list(plan)))
    segs = []
    for g in range(p["n_teams"]):
        buf = (C.c_int32 * (5 * 64))()
        n = lib.fvqa_gemm_sk_describe(M, N, K, dtype, n_cu, None, g, C.cast(buf, C.c_void_p), 64)
        assert 0 <= n <= 64
        for i in range(n):
            tile, k0, k1, pieces, c = list(buf[5 * i:5 * i + 5])
            segs.append(dict(team=g, order=i, tile=tile, k0=k0, k1=k1, n=pieces, c=c))
    return p, segs


@pytest.mark.parametrize("M,N,K,dtype", [
    (1024, 12288, 4096, 1), (1024, 4096, 4096, 1), (1024, 22016, 4096, 1), (1024, 4096, 11008, 1),
    (1024, 11008, 4096, 1), (1024, 4096, 22016, 1), (1024, 4096, 12288, 1), (1024, 32000, 4096, 1),
    (1024, 4096, 32000, 1), (3072, 12288, 4096, 1), (3072, 4096, 11008, 1), (3072, 22016, 4096, 1),
    (1950, 4096, 4096, 1), (1950, 11008, 4096, 1), (1536, 15360, 5120, 1), (1536, 5120, 13824, 1),
    (1536, 27648, 5120, 1), (256, 256, 64, 1), (300, 768, 2112, 0), (1034, 512, 1024, 0), (200, 256, 128, 1),
    (2048, 2048, 2048, 1), (1024, 4096, 4096, 0), (4096, 4096, 4096, 1), (8192, 8192, 8192, 1), (256, 4096, 4096, 1)])
def test_persistent_gemm_partition_covers_every_tile_once(M, N, K, dtype):
    """The (tile, K range) partition of csrc/gemm_sk_plan.h, walked through the host-only C entry: every wide stage of
    every tile is computed exactly once; a split tile has 2, 4 or 8 pieces in K order held by different teams as
    their LAST segment after the same number of whole tiles (partners finish together; nobody waits on a workgroup
    that still has other work to do first); with pstride > 1 all teams of an XCD chunk hold the same piece index."""
    for n_cu in (256, 304, 64):
        p, segs = _sk_plan(M, N, K, dtype, n_cu)
        assert p["n_teams"] * p["ts"] <= n_cu and p["ts"] * p["mgroups"] >= p["tm"]
        wide = 64 if dtype == 1 else 32
        assert p["nw_tile"] == K // wide and p["gpt"] == -(-p["nw_tile"] // p["gran"])
        tiles = p["mgroups"] * p["tn"]
        assert p["full"] * p["n_teams"] + p["rem"] == tiles and p["rem"] * p["s"] <= p["n_teams"]
        by_tile, n_seg = {}, {}
        for s in segs:
            assert 0 <= s["tile"] < tiles and 0 <= s["k0"] < s["k1"] <= p["nw_tile"]
            by_tile.setdefault(s["tile"], []).append(s)
            n_seg[s["team"]] = n_seg.get(s["team"], 0) + 1
        assert sorted(by_tile) == list(range(tiles))
        for t, ss in by_tile.items():
            ss.sort(key=lambda s: s["k0"])
            assert ss[0]["k0"] == 0 and ss[-1]["k1"] == p["nw_tile"]
            assert all(a["k1"] == b["k0"] for a, b in zip(ss, ss[1:]))          # no gap, no overlap
            n = len(ss)
            assert n in (1, 2, 4, 8) and n == (p["s"] if t >= p["full"] * p["n_teams"] else 1)
            for c, s in enumerate(ss):
                assert s["n"] == n and s["c"] == c
                if n > 1:
                    assert s["order"] == p["full"] == n_seg[s["team"]] - 1
                    assert s["k1"] - s["k0"] > p["gran"]
                    if p["pstride"] == 1:
                        assert s["team"] == ss[0]["team"] + c
                    else:
                        # one piece index per XCD chunk of `pstride` teams; a 4-way split holds pieces 0, 2, 1, 3 on the chunks
                        # 0, 1, 2, 3 of a group (the slow first-half-of-K pieces on the fast even XCDs, gemm_sk_plan.h)
                        chunk, y = divmod(s["team"], p["pstride"])
                        want = {0: 0, 1: 2, 2: 1, 3: 3}[chunk % 4] if n == 4 else chunk % n
                        assert s["c"] == want
                        c0, y0 = divmod(ss[0]["team"], p["pstride"])
                        assert y == y0 and chunk // n == c0 // n                 # same slot of the same group of chunks
            if n > 1 and p["pstride"] > 1:
                assert len({s["team"] // p["pstride"] for s in ss}) == n            # the pieces sit on n different chunks


def test_persistent_gemm_plans_of_the_c2_step():
    """Which partition each projection of the C2 step (7B, M = 1024) gets on 256 CUs: (teams, whole rounds, tiles of the
    split round, pieces per tile)."""
    want = {(12288, 4096): (48, 0, 48, 1), (4096, 4096): (64, 0, 16, 4), (22016, 4096): (64, 1, 22, 2),
            (4096, 11008): (64, 0, 16, 4), (11008, 4096): (43, 0, 43, 1), (4096, 22016): (64, 0, 16, 4),
            (4096, 12288): (64, 0, 16, 4), (32000, 4096): (64, 1, 61, 1), (4096, 32000): (64, 0, 16, 4)}
    for (N, K), w in want.items():
        p, _ = _sk_plan(1024, N, K, 1)
        assert (p["n_teams"], p["full"], p["rem"], p["s"]) == w and p["ts"] == 4, (N, K, p)
        assert p["pstride"] == (8 if w[3] == 4 else 1)


def test_stale_library_is_refused(monkeypatch):
    """The library carries the hash of the kernel sources it was built from (csrc/version.hip, -DFVQA_SOURCE_HASH); a binary
    older than the sources next to it — same ABI number, different kernels — must not load."""
    from fvqa import _lib, build
    lib = _lib.load()                                     # the in-tree build matches its sources
    assert lib.fvqa_source_hash().decode() == build.source_hash()
    monkeypatch.setattr(_lib, "_LIBS", {})
    monkeypatch.setattr(build, "source_hash", lambda: "0" * 64)      # = a kernel source touched after the build
    with pytest.raises(_lib.FvqaLibraryError, match="stale binary"):
        _lib.load()
    monkeypatch.undo()
    assert _lib.load() is not None


def test_rank_startup_checks_name_what_is_wrong():
    """fvqa/rankcheck.py: the reports every rank gathers before the first data-parallel step (reference train.py:104-117,
    util/misc.py:220-250): a sound 8-rank node passes; a wrong world size, two ranks on one device, a compute-partitioned
    device (fewer than 256 CUs), too few devices and unpinned host threads are each named."""
    from fvqa import rankcheck

    def rep(rank, dev, cus=256, host="n0", ndev=8, omp="8"):
        return {"rank": rank, "local_rank": rank, "host": host, "device_index": dev, "device_count": ndev,
                "device_id": f"uuid:gpu-{host}-{dev}", "cu_count": cus, "gcn_arch": "gfx950:sramecc+:xnack-",
                "omp_num_threads": omp, "cpu_count": 128}

    good = [rep(r, r) for r in range(8)]
    assert rankcheck.verify(good, 8, 8) == []
    assert any("were asked for" in p for p in rankcheck.verify(good[:4], 8, 4))
    shared = [rep(0, 0), rep(1, 0)]
    assert any("both use device" in p for p in rankcheck.verify(shared, 2, 2))
    assert rankcheck.verify(shared, 2, 2, rehearsal=True) == []                   # a one-GPU rehearsal may share the device
    part = [rep(0, 0, cus=32), rep(1, 1)]
    assert any("32 CUs" in p and "partition" in p for p in rankcheck.verify(part, 2, 2))
    few = [rep(0, 0, ndev=1), rep(1, 1, ndev=1)]
    assert any("visible devices" in p for p in rankcheck.verify(few, 2, 2))
    assert any("OMP_NUM_THREADS" in p for p in rankcheck.verify([rep(0, 0, omp=None), rep(1, 1)], 2, 2))
    wrong_arch = [dict(rep(0, 0), gcn_arch="gfx942")]
    assert any("gfx950" in p for p in rankcheck.verify(wrong_arch, 1, 1))
    assert rankcheck.verify([rep(0, 0, omp=None)], 1, 1) == []                    # a single rank need not pin its threads
    # the effective thread count is what is judged once a rank reports it: a rank that pinned itself without the variable
    # (mpirun --dist_on_itp, srun) is sound, one still at a thread per core is warned about — and a warning never raises
    pinned = [dict(rep(r, r, omp=None), torch_threads=16) for r in range(8)]
    assert rankcheck.verify(pinned, 8, 8) == []
    greedy = [dict(rep(r, r, omp=None), torch_threads=128) for r in range(8)]
    assert all(p.startswith("warning:") and "oversubscribe" in p for p in rankcheck.verify(greedy, 8, 8))


def test_init_distributed_mode_without_omp_num_threads(monkeypatch):
    """Round-4 advisor finding: launchers the reference's init_distributed_mode accepts (util/misc.py:221-236: mpirun with
    --dist_on_itp, SLURM srun) set no OMP_NUM_THREADS; pin_host_threads then chooses the rank's share, EXPORTS it, and the
    start-up check (which used to read only the environment and abort every rank) passes."""
    import torch
    from fvqa import rankcheck
    monkeypatch.delenv("OMP_NUM_THREADS", raising=False)
    before = torch.get_num_threads()
    try:
        n = rankcheck.pin_host_threads(8, workers_per_rank=2)
        assert n == max(1, (os.cpu_count() or 1) // 8 - 2)
        assert os.environ["OMP_NUM_THREADS"] == str(n) and torch.get_num_threads() == n
        mine = rankcheck.local_report(0, 0, 0)
        assert mine["omp_num_threads"] == str(n) and mine["torch_threads"] == n
        peers = [dict(mine, rank=r, local_rank=r, device_index=r, device_id=f"uuid:{r}", cu_count=256, device_count=8,
                      gcn_arch="gfx950") for r in range(8)]
        assert rankcheck.verify(peers, 8, 8) == []
    finally:
        torch.set_num_threads(before)


def test_train_one_epoch_stops_every_rank_on_an_error_lane_set_in_the_last_iteration():
    """Round-4 advisor finding: found_inf = 2 (the all-reduced error lane: SOME rank's split-K exchange timed out) left by the
    LAST iteration of an epoch used to be read by nobody — the faulty rank raised on its own error word, the healthy ones walked
    into the meter all-reduce and waited for it forever. engine.train_one_epoch (reference engine.py:10-56) now reads the loss
    scaler's found-inf word after the loop on EVERY rank. Host logic only: stub model and scaler, no device."""
    import types
    import engine

    class Scaler:
        def __init__(self, fault_at):
            self._found, self.calls, self.fault_at = torch.zeros(1), 0, fault_at

        def __call__(self, loss, optimizer, parameters=None, update_grad=True):
            self.calls += 1
            self._found = torch.tensor([2.0 if self.calls == self.fault_at else 0.0])

    class Model(torch.nn.Module):               # a HEALTHY rank: no engine error word of its own
        def forward(self, data):
            return torch.tensor(1.0), torch.tensor(0.5), torch.tensor(0.25)

    opt = types.SimpleNamespace(param_groups=[{"lr": 0.1}], zero_grad=lambda: None)
    args = types.SimpleNamespace(accum_iter=1, lr=0.1, min_lr=0.0, warmup_epochs=1, epochs=2, debug=False)
    stats = engine.train_one_epoch(Model(), [0, 1, 2, 3], opt, 0, Scaler(fault_at=0), args=args)
    assert abs(stats["loss"] - 1.75) < 1e-6
    with pytest.raises(RuntimeError, match="found_inf = 2"):
        engine.train_one_epoch(Model(), [0, 1, 2, 3], opt, 0, Scaler(fault_at=4), args=args)


def test_scored_rows_lists_match_the_cross_entropy_rule():
    """fvqa/scored.py against a plain loop over the reference's rule (llama/model.py:348-350: logits[:, :-1] against label[:, 1:],
    ignore_index 0): which rows, in which order, their shifted labels, pads and the inverse map."""
    import torch
    from fvqa import scored, synth
    cfg = synth.preset("small", vaq=True, qav=True)
    batch = synth.make_batch(cfg, seed=2)
    B, S = batch["video"].shape[0], cfg.max_seq_len
    batch["label"] = {t: v.clone() for t, v in batch["label"].items()}
    batch["label"]["vqa"][1] = 0                              # a sample with no scored row
    batch["label"]["vaq"][0, 0, 7] = -3                       # a negative label is not a class id
    out = scored.annotate(batch)
    assert out is batch and set(batch[scored.COUNT]) == {"vqa", "vaq", "qav"}
    for t in ("vqa", "vaq", "qav"):
        L = batch["label"][t].reshape(B, S)
        want_rows, want_lab = [], []
        for n in range(B):
            for p in range(S - 1):
                if (int(L[n, p + 1]) >= 0) if t == "qav" else (int(L[n, p + 1]) > 0):
                    want_rows.append(n * S + p)
                    want_lab.append(int(L[n, p + 1]))
        m = batch[scored.COUNT][t]
        idx, inv, lab = (batch[f][t] for f in scored.FIELDS)
        assert m == len(want_rows) and idx.shape == inv.shape == lab.shape == (B, S)
        assert idx.dtype == inv.dtype == torch.int32 and lab.dtype == torch.int64
        rows = scored.rows_of(m)
        assert rows == m + 1
        assert idx.view(-1)[:m].tolist() == want_rows and int(idx.view(-1)[m]) == 0
        assert lab.view(-1)[0] == 0 and lab.view(-1)[1:m + 1].tolist() == want_lab and int(lab.view(-1)[m + 1:].abs().sum()) == 0
        flat = inv.view(-1)
        assert [int(flat[r]) for r in want_rows] == list(range(m)) and int((flat >= 0).sum()) == m
    again = {f: {t: v.clone() for t, v in batch[f].items()} for f in scored.FIELDS}
    scored.annotate(batch)                                     # idempotent
    assert all(torch.equal(again[f][t], batch[f][t]) for f in scored.FIELDS for t in again[f])
    # a stream with no scored row at all keeps a two-row segment (the CE kernels need a "sequence" of >= 2 rows; the loss is 0 / 0
    # = NaN there, as torch's cross_entropy gives and engine.py:33-35 exits on)
    idx, inv, lab, m = scored.lists_of(torch.zeros(B, 1, S, dtype=torch.int64))
    assert m == 0 and scored.rows_of(0) == 2 and int(lab.abs().sum()) == 0 and int((inv >= 0).sum()) == 0


def test_ctypes_mirrors_match_the_c_header(tmp_path):
    """fvqa/_lib.py mirrors fvqa_layer_plan, fvqa_row_segs, fvqa_sk_rider and fvqa_sk_rope by hand: compile include/fvqa.h with gcc and
    compare sizes and the offsets of the fields a mismatch would silently corrupt."""
    import shutil
    import subprocess
    from fvqa import _lib
    gcc = shutil.which("gcc")
    assert gcc, "gcc is part of the image"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "sz.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "fvqa.h"\n'
        'int main(void) {\n'
        '  printf("%zu %zu %zu %zu\\n", sizeof(fvqa_layer_plan), sizeof(fvqa_row_segs), sizeof(fvqa_sk_rider), sizeof(fvqa_sk_rope));\n'
        '  printf("%zu %zu %zu %zu %zu %zu\\n", offsetof(fvqa_layer_plan, eps), offsetof(fvqa_layer_plan, wqkv), offsetof(fvqa_layer_plan, xs),\n'
        '         offsetof(fvqa_layer_plan, gemm_ws_bytes), offsetof(fvqa_layer_plan, tail), offsetof(fvqa_row_segs, map));\n'
        '  fvqa_layer_plan p; printf("%zu %zu %zu\\n", (size_t)((char*)&p.tail.gather - (char*)&p.tail), (size_t)((char*)&p.tail.og - (char*)&p.tail),\n'
        '         (size_t)((char*)&p.tail.d_o - (char*)&p.tail));\n'
        '  return 0;\n}\n')
    exe = tmp_path / "sz"
    subprocess.run([gcc, "-I", os.path.join(root, "include"), "-o", str(exe), str(src)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    got = [int(x) for x in out]
    LP, RS, TR = _lib.LayerPlan, _lib.RowSegs, _lib.TailRows
    want = [C.sizeof(LP), C.sizeof(RS), C.sizeof(_lib.SkRider), C.sizeof(_lib.SkRope),
            LP.eps.offset, LP.wqkv.offset, LP.xs.offset, LP.gemm_ws_bytes.offset, LP.tail.offset, RS.map.offset,
            TR.gather.offset, TR.og.offset, TR.d_o.offset]
    assert got == want, (got, want)


def test_bench_accounting_of_tail_rows():
    """bench.py's two bookkeeping helpers for the tail rows: executed FLOPs (what the kernels multiplied) beside the contract's, and the
    algorithmic bytes of a projection launch whose row count is its own."""
    import importlib.util
    import types
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    D, Hf, V, B, S = 4096, 11008, 32000, 8, 128
    fl = bench.step_flops(D, 32, 32, Hf, V, B, S, 10, 10, ["vqa"])
    assert abs(fl - 27.366e12) < 0.01e12                                   # 27.37 TF per C2 step
    eng = types.SimpleNamespace(lm_head_rows="scored")
    model = types.SimpleNamespace(ensure_engine=lambda: eng, params=types.SimpleNamespace(dim=D), vocab_size=V,
                                  layers=[types.SimpleNamespace(feed_forward=types.SimpleNamespace(
                                      w1=types.SimpleNamespace(weight=types.SimpleNamespace(shape=(Hf, D)))))])
    batches = [{"scored_count": {"vqa": 32}} for _ in range(4)]
    sr = bench.step_roofline_of(fl, 25.0, 2.5e15, model, batches, B, S, ["vqa"])
    skipped = (1024 - 33) * (2.0 * D * V * 2 + 2.0 * (2.0 * D * D + 6.0 * D * Hf) * 2)
    assert sr["lm_head_rows"] == "scored" and sr["tail_rows_per_step"] == 33 and sr["tail_rows_dense"] == 1024
    assert abs(sr["executed_flops_per_step"] - (fl - skipped)) < 1e6 and sr["frac_executed"] < sr["frac"]
    assert abs(sr["frac"] - fl / 25.0e-3 / 2.5e15) < 1e-12
    eng.lm_head_rows = "all"
    assert "frac_executed" not in bench.step_roofline_of(fl, 25.0, 2.5e15, model, batches, B, S, ["vqa"])
    # a launch of 33 rows against W_out (fp32 logits) and against W1|W3^T: the row count comes from the launch's own FLOPs
    b33 = bench.launch_alg_bytes(32, 2.0 * 33 * V * D, 1024, D, Hf, V)
    assert b33 == 2.0 * 33 * D + 2.0 * V * D + 4.0 * 33 * V
    b1024 = bench.launch_alg_bytes(32, 2.0 * 1024 * V * D, 1024, D, Hf, V)
    assert b1024 == 2.0 * 1024 * D + 2.0 * V * D + 4.0 * 1024 * V
    assert bench.launch_alg_bytes(0, 2.0 * 33 * D * 2 * Hf, 1024, D, Hf, V) == 2.0 * 33 * 2 * Hf + 2.0 * D * 2 * Hf + 2.0 * 33 * D
