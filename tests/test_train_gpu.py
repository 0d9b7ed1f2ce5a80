"""engine.train_one_epoch / loss scaler / FusedAdamW end to end on the GPU (drop-in entry points of
reference engine.py:10-56, util/misc.py:253-279, train.py:120-121)."""
import json
import math
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

import engine  # noqa: E402
import util.misc as misc  # noqa: E402
from fvqa import synth  # noqa: E402
from fvqa.optim import FusedAdamW, param_groups_weight_decay  # noqa: E402
from oracle import ref_cpu  # noqa: E402
from tests.gpu_util import build_model  # noqa: E402


def _setup(dtype=torch.float32, **over):
    cfg = synth.preset("tiny", vaq=True, qav=True, **over)
    model, args = build_model(cfg, dtype)
    groups = param_groups_weight_decay(model, args.weight_decay)
    opt = FusedAdamW(groups, lr=args.lr, betas=(0.9, 0.95), flat=model.flat_params())
    return cfg, model, args, opt


def test_param_groups_follow_timm_rule():
    cfg, model, args, opt = _setup()
    no_decay, decay = opt.param_groups
    assert no_decay["weight_decay"] == 0.0 and decay["weight_decay"] == args.weight_decay
    assert len(no_decay["params"]) == 0                    # every trainable is >= 2-D (gates are 4-D)
    n_train = sum(1 for n, p in model.named_parameters() if p.requires_grad)
    assert len(decay["params"]) == n_train == 3 + 2 * cfg.n_layers
    assert sum(p.numel() for p in decay["params"]) == model.flat_params().flat.numel()


def test_one_optimizer_step_matches_oracle_plus_torch_adamw():
    cfg, model, args, opt = _setup()
    batch = synth.make_batch(cfg, seed=7)
    scaler = misc.NativeScalerWithGradNormCount()
    opt.zero_grad()
    for g in opt.param_groups:
        g["lr"] = 0.01
    a, b, c = model(batch)
    norm = scaler(a + b + c, opt, parameters=model.parameters(), update_grad=True)
    torch.cuda.synchronize()
    # oracle gradient + torch AdamW on CPU
    sd = synth.state_dict(cfg)
    ref = ref_cpu.RefModel(cfg, sd, dtype=torch.float64).step(batch)
    names = [n for n in ref["grads"]]
    ps = [torch.nn.Parameter(sd[n].double().clone()) for n in names]
    for p, n in zip(ps, names):
        p.grad = ref["grads"][n].clone()
    ref_norm = torch.norm(torch.stack([p.grad.norm() for p in ps]))
    torch.optim.AdamW(ps, lr=0.01, betas=(0.9, 0.95), weight_decay=args.weight_decay).step()
    assert abs(float(norm) - float(ref_norm)) / float(ref_norm) < 1e-3
    own = dict(model.named_parameters())
    for p, n in zip(ps, names):
        got = own[n].detach().double().cpu()
        # Adam's first step moves every element by ~lr regardless of gradient scale; compare updates
        upd_ref = p.detach() - sd[n].double()
        upd_got = got - sd[n].double()
        big = ref["grads"][n].abs() > 1e-3 * ref["grads"][n].abs().max()    # sign-stable elements
        assert torch.allclose(upd_got[big], upd_ref[big], rtol=2e-2, atol=1e-6), n
    assert opt.step_dev.item() == 1.0
    assert scaler.state_dict()["scale"] == 65536.0


def test_gradient_accumulation_matches_the_oracle():
    """accum_iter = 2 as engine.train_one_epoch / the reference drive it (engine.py:37-41: loss / accum_iter, backward every
    micro-batch, unscale + step only at the boundary): the gradient the optimizer sees is the MEAN of the two micro-batch
    gradients — elementwise against the fp64 oracle — and the returned norm is its norm (util/misc.py:266-276)."""
    cfg, model, args, opt = _setup()
    b0, b1 = synth.make_batch(cfg, seed=11), synth.make_batch(cfg, seed=12)
    scaler = misc.NativeScalerWithGradNormCount()
    opt.zero_grad()
    norm = None
    for i, batch in enumerate((b0, b1)):
        a, b, c = model(batch)
        norm = scaler((a + b + c) / 2, opt, parameters=model.parameters(), update_grad=(i == 1))
    torch.cuda.synchronize()
    sd = synth.state_dict(cfg)
    r0 = ref_cpu.RefModel(cfg, sd, dtype=torch.float64).step(b0)
    r1 = ref_cpu.RefModel(cfg, sd, dtype=torch.float64).step(b1)
    want = {n: (r0["grads"][n] + r1["grads"][n]) / 2 for n in r0["grads"]}
    own = {n: p for n, p in model.named_parameters() if p.requires_grad}
    assert set(own) == set(want)
    for n, p in own.items():
        got = p.grad.detach().double().cpu()
        ref = want[n].reshape(got.shape)
        assert float((got - ref).abs().max() / ref.abs().max()) < 2e-4, n
        # and it is NOT one micro-batch's gradient (the fixture can tell them apart)
        assert float((got - r1["grads"][n].reshape(got.shape)).abs().max() / ref.abs().max()) > 1e-2, n
    ref_norm = torch.sqrt(sum(g.pow(2).sum() for g in want.values()))
    assert abs(float(norm) - float(ref_norm)) / float(ref_norm) < 1e-4
    assert opt.step_dev.item() == 1.0


def test_train_one_epoch_contract_and_learning():
    cfg, model, args, opt = _setup(torch.bfloat16)
    args.accum_iter, args.lr, args.warmup_epochs, args.epochs = 2, 0.02, 0, 2
    loader = synth.SyntheticLoader(cfg, 4)
    loader.batches = [loader.batches[0]] * 4                # same batch: the loss must go down
    scaler = misc.NativeScalerWithGradNormCount()
    before = model.flat_params().flat.clone()
    s0 = engine.train_one_epoch(model, loader, opt, 0, scaler, args=args)
    s1 = engine.train_one_epoch(model, loader, opt, 1, scaler, args=args)
    assert set(s0) == {"lr", "loss", "vqa_loss", "vaq_loss", "qav_loss"}
    assert all(math.isfinite(v) for v in s0.values())
    assert abs(s0["loss"] - (s0["vqa_loss"] + s0["vaq_loss"] + s0["qav_loss"])) < 1e-3
    assert not torch.equal(before, model.flat_params().flat)
    assert s1["loss"] < s0["loss"]
    assert opt.step_dev.item() == 4.0                        # 8 micro-steps / accum_iter 2


def test_train_one_epoch_stops_on_a_non_finite_loss(capsys):
    """All labels ignored -> the CE mean is 0/0 = NaN -> the reference prints and exits with status 1
    (engine.py:33-35). The loss is read after the backward has been launched; the exit is the same."""
    cfg, model, args, opt = _setup(torch.bfloat16)
    args.accum_iter, args.lr, args.warmup_epochs, args.epochs = 1, 0.02, 0, 1
    loader = synth.SyntheticLoader(cfg, 4)
    for b in loader.batches:
        b["label"]["vqa"] = torch.zeros_like(b["label"]["vqa"])          # ignore_index everywhere
    with pytest.raises(SystemExit) as e:
        engine.train_one_epoch(model, loader, opt, 0, misc.NativeScalerWithGradNormCount(), args=args)
    assert e.value.code == 1
    assert "Loss is nan, stopping training" in capsys.readouterr().out
    torch.cuda.synchronize()
    assert opt.step_dev.item() <= 1.0                        # it stopped at the FIRST iteration


def test_train_one_epoch_raises_on_a_set_gemm_error_word():
    """A timed-out split-K exchange (sticky error word of the stream's GEMM workspace) must end the epoch at the next
    iteration, not after a whole epoch of skipped steps: the step it hits is a no-op on the device (found_inf = 2), and the
    word reaches the host in the NEXT iteration's loss transfer (no device-to-host read of its own)."""
    from fvqa import ops
    cfg = synth.preset("7b_l2", batch_size=2, vaq=True, qav=True)
    model, args = build_model(cfg, torch.bfloat16)
    opt = FusedAdamW(param_groups_weight_decay(model, args.weight_decay), lr=0.01, betas=(0.9, 0.95),
                     flat=model.flat_params())
    args.accum_iter, args.lr, args.warmup_epochs, args.epochs = 1, 0.02, 0, 1
    scaler = misc.NativeScalerWithGradNormCount()
    loader = synth.SyntheticLoader(cfg, 4)
    engine.train_one_epoch(model, synth.SyntheticLoader(cfg, 2), opt, 0, scaler, args=args)    # a clean epoch first
    torch.cuda.synchronize()
    steps0, p0 = opt.step_dev.item(), model.flat_params().flat.clone()
    word = ops.gemm_error_word(model.flat_params().flat.device)
    assert word is not None
    word.view(torch.int64)[0] = 1
    try:
        with pytest.raises(RuntimeError, match="split-K exchange"):
            engine.train_one_epoch(model, loader, opt, 1, scaler, args=args)
        torch.cuda.synchronize()
        assert opt.step_dev.item() == steps0                 # no step was applied while the word was set
        assert torch.equal(model.flat_params().flat, p0)
    finally:
        word.view(torch.int64)[0] = 0


def test_switched_off_losses_are_the_reference_placeholders():
    """vaq / qav off: `tensor([0])` int64 on the model's device (llama/model.py:302), so that the summed loss has shape [1]."""
    cfg = synth.preset("tiny", vaq=False, qav=False)
    model, args = build_model(cfg, torch.float32)
    vqa, vaq, qav = model(synth.make_batch(cfg, seed=3))
    for z in (vaq, qav):
        assert z.dtype == torch.int64 and tuple(z.shape) == (1,) and z.is_cuda and int(z) == 0
    assert tuple((vqa + vaq + qav).shape) == (1,)


def test_checkpoint_roundtrip(tmp_path):
    cfg, model, args, opt = _setup()
    scaler = misc.NativeScalerWithGradNormCount()
    batch = synth.make_batch(cfg, seed=9)
    a, b, c = model(batch)
    scaler(a + b + c, opt, parameters=model.parameters())
    args.output_dir = str(tmp_path)
    misc.save_model(args, 3, model, model, opt, scaler, "checkpoint_best")
    ck = torch.load(tmp_path / "checkpoint_best.pth", map_location="cpu", weights_only=False)
    assert set(ck) == {"model", "optimizer", "epoch", "scaler", "args"}
    want = {n for n, _ in model.named_parameters() if synth.is_trainable(n)}
    assert set(ck["model"]) == want                         # reference key names, trainables only
    cfg2, model2, args2, opt2 = _setup()
    args2.resume = str(tmp_path / "checkpoint_best.pth")
    scaler2 = misc.NativeScalerWithGradNormCount()
    misc.load_model(args2, model2, opt2, scaler2)
    assert args2.start_epoch == 4
    assert torch.equal(model2.flat_params().flat, model.flat_params().flat)
    assert torch.equal(opt2.exp_avg, opt.exp_avg) and opt2.step_dev.item() == opt.step_dev.item()


def _write_meta_checkpoint(model_dir, cfg, n_shards=2):
    """What the reference reads (llama_vqa.py:8-22): params.json + `n_shards` model-parallel fp16 shards in Meta's
    layout (column-parallel tensors cut on dim 0, wo / w2 / tok_embeddings on dim 1, norms replicated, plus the
    `rope.freqs` buffer the reference's strict=False load ignores). Values: the closed-form frozen weights."""
    from llama_vqa import _SPLIT_DIM
    os.makedirs(model_dir, exist_ok=True)
    sd = synth.state_dict(cfg)
    shards = [dict() for _ in range(n_shards)]
    for name, t in sd.items():
        if synth.is_trainable(name):
            continue                                       # LLaMA checkpoints hold no adapter / gate / projection
        short = name.split(".", 2)[2] if name.startswith("layers.") else name
        dim = _SPLIT_DIM[short]
        pieces = [t.half().clone() for _ in range(n_shards)] if dim < 0 else [c.half().contiguous() for c in t.chunk(n_shards, dim)]
        for sh, pc in zip(shards, pieces):
            sh[name] = pc
    for i, sh in enumerate(shards):
        sh["rope.freqs"] = torch.arange(cfg.head_dim // 2, dtype=torch.float16)
        torch.save(sh, os.path.join(model_dir, f"consolidated.{i:02d}.pth"))
    with open(os.path.join(model_dir, "params.json"), "w") as f:
        json.dump(cfg.params_json(), f)
    return sd


@pytest.mark.parametrize("dtype", ["bf16", "fp16", "fp32"])
def test_llama_vqa_from_sharded_fp16_checkpoint_end_to_end(tmp_path, dtype):
    """The REAL loading path (reference llama_vqa.py:15-76), disk to step: params.json + a 2-shard fp16 Meta checkpoint
    -> LLaMA_VQA(args) (no random_init: glob, merge, strict=False load, freeze policy) -> one training step on the
    HIP path, against the oracle fed the SAME weights (fp16-rounded as stored; bf16-rounded on top for the bf16
    storage build). dtype 'fp16' holds the shards' values exactly in the module, as the reference does, and computes on
    them with the fp16 build of the kernels (libfvqa_hip_f16.so)."""
    import types
    from llama_vqa import LLaMA_VQA
    cfg = synth.preset("tiny", vaq=True, qav=True)
    sd = _write_meta_checkpoint(tmp_path / "7B", cfg)
    args = types.SimpleNamespace(
        llama_model_path=str(tmp_path) + "/", model="7B", max_seq_len=cfg.max_seq_len, adapter_len=cfg.adapter_len,
        adapter_layer=cfg.adapter_layer, max_feats=cfg.max_feats, bias=cfg.bias, tau=cfg.tau, vaq=True, qav=True,
        audio=False, audio_only=False, audio_merge="none", debug=False, synthetic=True, vocab_size=cfg.vocab_size,
        dtype=dtype, accum_iter=1, weight_decay=0.14)
    model = LLaMA_VQA(args)                                  # reads the shards: no random_init
    model.to("cuda")
    store = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[dtype]
    assert not any(n.startswith("rope") for n, _ in model.named_parameters())
    for n, p in model.named_parameters():
        if synth.is_trainable(n):
            assert p.requires_grad and p.dtype == torch.float32, n
        else:
            assert not p.requires_grad and p.dtype == store, (n, p.dtype)
            want = sd[n].half().to(store)                    # the shard values, merged
            assert torch.equal(p.detach().cpu(), want), f"{n}: merged shard values differ"
    with torch.no_grad():                                    # trainables are not in a LLaMA checkpoint: known values
        own = dict(model.named_parameters())
        for n, t in sd.items():
            if synth.is_trainable(n):
                own[n].copy_(t)
    batch = synth.make_batch(cfg, seed=5)
    from tests.gpu_util import run_step
    losses, grads, _, _ = run_step(model, batch)
    # what the kernels compute on: the module's own storage type — 'fp16' runs the fp16 build of the kernels on the shards'
    # values exactly as the reference holds them (round 5; rounds 2-4 re-rounded them to bf16 when the engine packed them)
    assert model.tok_embeddings.weight.dtype == store
    assert all(p.dtype == store for n, p in model.named_parameters() if not synth.is_trainable(n))
    from fvqa import _lib
    assert ("f16" in _lib._LIBS) == (dtype == "fp16") or dtype != "fp16"      # the fp16 library is what served this model
    ref_sd = {}
    for n, t in sd.items():
        if synth.is_trainable(n):
            ref_sd[n] = t
        else:
            r = t.half()
            ref_sd[n] = (r.to(torch.bfloat16) if dtype == "bf16" else r).float()
    ref = ref_cpu.RefModel(cfg, ref_sd, dtype=torch.float64).step(batch)
    ltol, gtol = (1e-3, 1e-3) if dtype == "fp32" else (2e-2, 8e-2)
    for t in ref["tasks"]:
        r = float(ref["losses"][t])
        assert abs(losses[t] - r) / abs(r) < ltol, (dtype, t, losses[t], r)
    for n, g in ref["grads"].items():
        gn = float(g.norm())
        if gn > 0:
            assert float((grads[n].double() - g).norm()) / gn < gtol, (dtype, n)
    model._engine.check_gemm_error()


def test_gates_of_skipped_layers_are_left_alone():
    """adapter_layer < n_layers: the engine skips the first layers (reference llama/model.py:338), their gates get no
    gradient in the reference (.grad is None -> torch AdamW neither decays nor moves them)."""
    cfg, model, args, opt = _setup(adapter_layer=1)          # tiny has 2 layers: layer 0 is skipped
    assert model.engine_layer_ids() == [1]
    flat = model.flat_params()
    named = dict(model.named_parameters())
    g_idle = [named["layers.0.attention.gate1"], named["layers.0.attention.gate2"]]
    g_live = [named["layers.1.attention.gate1"], named["layers.1.attention.gate2"]]
    before_idle = [g.detach().clone() for g in g_idle]
    before_live = [g.detach().clone() for g in g_live]
    assert float(before_idle[1].abs().max()) > 0             # gate2 = -bias: weight decay WOULD move it
    scaler = misc.NativeScalerWithGradNormCount()
    for g in opt.param_groups:
        g["lr"] = 0.01
    for s in range(2):
        opt.zero_grad()
        a, b, c = model(synth.make_batch(cfg, seed=3 + s))
        scaler(a + b + c, opt, parameters=None, update_grad=True)
    torch.cuda.synchronize()
    for g, b in zip(g_idle, before_idle):
        assert torch.equal(g.detach(), b)
    assert all(not torch.equal(g.detach(), b) for g, b in zip(g_live, before_live))
    for off in flat.idle_offsets():
        assert float(opt.exp_avg[off:off + cfg.n_heads].abs().max()) == 0.0


# ------------------------------------------------------------------------------ checkpoint interchange
def _ckpt_setup():
    from oracle.gen_golden_ckpt import CKPT_CFG          # the configuration the reference-written fixture was made at
    cfg = synth.SynthConfig(**CKPT_CFG)
    model, args = build_model(cfg, torch.float32)
    opt = FusedAdamW(param_groups_weight_decay(model, 0.14), lr=0.01, betas=(0.9, 0.95), flat=model.flat_params())
    return cfg, model, args, opt


def test_resume_from_a_checkpoint_written_by_the_reference(golden_dir):
    """tests/golden/ckpt_reference.pth was written by the reference's own util.misc.save_model after ONE optimizer step
    of the reference model (torch AdamW + its GradScaler wrapper; oracle/gen_golden_ckpt.py). The product resumes from
    that file — trainables, AdamW moments and step, loss-scale state, start epoch — and its NEXT step must land where
    the reference's own next step landed (reference util/misc.py:297-336)."""
    import json
    import os
    import numpy as np
    g = dict(np.load(os.path.join(golden_dir, "ckpt_reference.npz")))
    cfg, model, args, opt = _ckpt_setup()
    scaler = misc.NativeScalerWithGradNormCount()
    args.resume = os.path.join(golden_dir, "ckpt_reference.pth")
    misc.load_model(args, model, opt, scaler)
    assert args.start_epoch == int(g["epoch"]) + 1 == 4
    own = dict(model.named_parameters())
    names = [str(n) for n in g["model_keys"]]
    assert names == [n for n, p in model.named_parameters() if p.requires_grad]       # the reference's key set and order
    for n in names:
        assert torch.equal(own[n].detach().cpu(), torch.from_numpy(g[f"model__{n}"]).view_as(own[n])), n
    # optimizer: index i of the reference's state dict is the i-th trainable in timm group order
    assert [str(n) for n in g["opt_param_names"]] == names
    for i, n in enumerate(names):
        st = opt.state[own[n]]
        assert torch.equal(st["exp_avg"].cpu(), torch.from_numpy(g[f"state__{i}__exp_avg"])), n
        assert torch.equal(st["exp_avg_sq"].cpu(), torch.from_numpy(g[f"state__{i}__exp_avg_sq"])), n
        assert float(g[f"state__{i}__step"]) == 1.0
    assert opt.step_dev.item() == 1.0
    sc = json.loads(str(g["scaler_json"]))
    assert scaler.state_dict()["scale"] == sc["scale"] == 65536.0
    # ... and training goes on as the reference's did: one more step on the batch the reference used next
    for grp in opt.param_groups:
        assert grp["lr"] == 0.01 and tuple(grp["betas"]) == (0.9, 0.95)
    before = {n: own[n].detach().clone() for n in names}
    opt.zero_grad()
    a, b, c = model(synth.make_batch(cfg, seed=1))
    scaler(a + b + c, opt, parameters=None, update_grad=True)
    torch.cuda.synchronize()
    for got, ref in zip((a, b, c), g["losses_step2"]):
        assert abs(float(got.detach()) - float(ref)) / float(ref) < 1e-4
    for n in names:
        upd_ref = torch.from_numpy(g[f"after2__{n}"]).view_as(own[n]).double() - before[n].double().cpu()
        upd_got = own[n].detach().double().cpu() - before[n].double().cpu()
        # AdamW moves every element by ~lr whatever the gradient scale: compare the UPDATES, relative to their size
        assert float((upd_got - upd_ref).abs().max()) <= 2e-2 * float(upd_ref.abs().max()), n
        assert float((upd_got - upd_ref).norm()) <= 5e-3 * float(upd_ref.norm()), n
    assert scaler.state_dict()["_growth_tracker"] == json.loads(str(g["scaler_after2_json"]))["_growth_tracker"] == 2


def test_written_checkpoint_has_the_reference_layout(golden_dir, tmp_path):
    """A checkpoint written by the product after one step: the same top-level keys, trainable names / dtypes / shapes,
    optimizer state entries and param-group keys, scaler keys as the file the reference wrote (so that either side can
    load the other's)."""
    import json
    import os
    import numpy as np
    g = dict(np.load(os.path.join(golden_dir, "ckpt_reference.npz")))
    cfg, model, args, opt = _ckpt_setup()
    scaler = misc.NativeScalerWithGradNormCount()
    opt.zero_grad()
    a, b, c = model(synth.make_batch(cfg, seed=0))
    scaler(a + b + c, opt, parameters=None, update_grad=True)
    for got, ref in zip((a, b, c), g["losses"]):
        assert abs(float(got.detach()) - float(ref)) / float(ref) < 1e-4       # same model, same batch as the reference's step 1
    args.output_dir = str(tmp_path)
    misc.save_model(args, 3, model, model, opt, scaler, "checkpoint_best")
    ck = torch.load(tmp_path / "checkpoint_best.pth", map_location="cpu", weights_only=False)
    assert list(ck.keys()) == [str(k) for k in g["top_keys"]]
    assert list(ck["model"].keys()) == [str(k) for k in g["model_keys"]]
    for n, t in ck["model"].items():
        ref = g[f"model__{n}"]
        assert t.dtype == torch.float32 and tuple(t.shape) == ref.shape, n
        # one AdamW step from the same start: the product's trainables after step 1 track the reference's
        assert float((t.double() - torch.from_numpy(ref).double()).abs().max()) <= 2e-2 * 0.01 + 1e-6, n
    osd = ck["optimizer"]
    ref_groups = json.loads(str(g["param_groups_json"]))
    assert len(osd["param_groups"]) == len(ref_groups)
    for gp, gr in zip(osd["param_groups"], ref_groups):
        assert set(gp.keys()) == set(gr.keys())
        assert list(gp["params"]) == gr["params"] and gp["weight_decay"] == gr["weight_decay"]
        assert gp["lr"] == gr["lr"] and list(gp["betas"]) == gr["betas"] and gp["eps"] == gr["eps"]
    assert sorted(osd["state"].keys()) == [int(i) for i in g["state_ids"]]
    for i in osd["state"]:
        assert list(osd["state"][i].keys()) == [str(k) for k in g[f"state_keys__{i}"]]
        for k, v in osd["state"][i].items():
            ref = g[f"state__{i}__{k}"]
            assert torch.is_tensor(v) and tuple(v.shape) == ref.shape and str(v.dtype).endswith(str(ref.dtype)), (i, k)
    assert set(ck["scaler"].keys()) == set(json.loads(str(g["scaler_json"])).keys())
    assert ck["epoch"] == 3


def test_train_py_end_to_end_and_resume(tmp_path):
    """`python train.py` with the reference's CLI (train.py:24-176) on synthetic batches: one epoch of the full-size 7B
    (closed-form weights), log.txt + checkpoint_best.pth written as the reference writes them; a second run resumed from
    that checkpoint continues at the next epoch with a lower loss."""
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "flipped-vqa_amd")
    env = dict(os.environ, FVQA_SYNTHETIC_TOKENIZER="1")
    out = str(tmp_path / "run")
    base = [sys.executable, "train.py", "--model", "7B", "--random_init", "--synthetic", "--synthetic_batches", "6",
            "--batch_size", "4", "--max_seq_len", "128", "--warmup_epochs", "0", "--blr", "0.64", "--output_dir", out,
            "--llama_model_path", str(tmp_path / "no_assets") + "/"]
    r = subprocess.run(base + ["--epochs", "1"], cwd=pkg, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    log = [json.loads(x) for x in open(os.path.join(out, "log.txt"))]
    assert len(log) == 1 and log[0]["epoch"] == 0 and math.isfinite(log[0]["train_loss"])
    ck = torch.load(os.path.join(out, "checkpoint_best.pth"), map_location="cpu", weights_only=False)
    assert set(ck) == {"model", "optimizer", "epoch", "scaler", "args"} and ck["epoch"] == 0
    assert len(ck["model"]) == 3 + 2 * 32                    # trainables only, reference key names
    r2 = subprocess.run(base + ["--epochs", "2", "--resume", os.path.join(out, "checkpoint_best.pth")], cwd=pkg, env=env,
                        capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0, (r2.stdout[-1500:], r2.stderr[-1500:])
    log = [json.loads(x) for x in open(os.path.join(out, "log.txt"))]
    assert [x["epoch"] for x in log] == [0, 1]               # resumed at start_epoch = 1
    assert log[1]["train_loss"] < log[0]["train_loss"]
