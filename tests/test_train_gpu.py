"""engine.train_one_epoch / loss scaler / FusedAdamW end to end on the GPU (drop-in entry points of
reference engine.py:10-56, util/misc.py:253-279, train.py:120-121)."""
import math

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
