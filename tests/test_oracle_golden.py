"""The oracle (oracle/ref_cpu.py, our CPU restatement) against fixtures generated from the
reference itself (oracle/gen_golden.py, fp32-shim mode). CPU only."""
import pytest
import torch

from fvqa import synth
from oracle import ref_cpu
from tests.parity import CASES, compare_with_golden, load_golden


def run_oracle(case, dtype):
    pname, over = CASES[case]
    cfg = synth.preset(pname, **over)
    sd = synth.state_dict(cfg)
    model = ref_cpu.RefModel(cfg, sd, dtype=dtype)
    batch = synth.make_batch(cfg, seed=0)
    res = model.step(batch, keep=True)
    # reference hook order: for each layer, for each stream
    lo = []
    L = len(model.layer_ids())
    for i in range(L):
        for t in res["tasks"]:
            lo.append(res["extras"]["layer_out"][t][i])
    return res, lo


@pytest.mark.parametrize("case", ["tiny_vqa", "tiny_all", "tiny_cold", "small_all", "tiny_all_peaked"])
def test_oracle_matches_reference_fp64(case):
    g = load_golden(case)
    res, lo = run_oracle(case, torch.float64)
    rep = compare_with_golden(g, res["losses"], res["grads"], res["extras"]["logits"], lo, rtol=2e-5)
    assert rep["loss_vqa"] < 2e-6


@pytest.mark.parametrize("case", ["tiny_all"])
def test_oracle_matches_reference_fp32(case):
    g = load_golden(case)
    res, lo = run_oracle(case, torch.float32)
    compare_with_golden(g, res["losses"], res["grads"], res["extras"]["logits"], lo, rtol=1e-4)


def test_oracle_matches_reference_7b_width():
    """7B-width, 2 layers, B=2, S=128, triple loss (fp32 to bound memory/time)."""
    g = load_golden("7b_l2_all")
    res, lo = run_oracle("7b_l2_all", torch.float32)
    compare_with_golden(g, res["losses"], res["grads"], res["extras"]["logits"], lo, rtol=2e-4)


@pytest.mark.parametrize("case", ["7b_l2_b8_vqa_peaked", "7b_l2_s256_b4_all_peaked"])
def test_oracle_matches_reference_at_benchmark_shapes(case):
    """C2's shape (B = 8, S = 128, VQA loss) and a longer three-loss shape (S = 256, B = 4), 7B width, two layers, peaked
    logits: every token argmax of the oracle equals the reference's (fp32 to bound memory / time)."""
    g = load_golden(case)
    res, lo = run_oracle(case, torch.float32)
    rep = compare_with_golden(g, res["losses"], res["grads"], res["extras"]["logits"], lo, rtol=2e-4, min_decided=0.99)
    assert rep["argmax_vqa_decided"].split("/")[0] == rep["argmax_vqa_decided"].split("/")[1]


def test_oracle_backward_matches_autograd():
    """The hand-derived attention backward against autograd of the same forward."""
    torch.manual_seed(0)
    N, S, H, Dh, A, F = 2, 24, 2, 8, 3, 3
    dt = torch.float64
    q, k, v = (torch.randn(N, S, H, Dh, dtype=dt, requires_grad=True) for _ in range(3))
    ak, av = (torch.randn(A, H, Dh, dtype=dt, requires_grad=True) for _ in range(2))
    g1 = torch.randn(H, dtype=dt, requires_grad=True)
    g2 = torch.randn(H, dtype=dt, requires_grad=True)
    vstart = [4, -1]
    o, cache = ref_cpu.attn_fwd(q, k, v, ak, av, g1, g2, vstart, F)
    do = torch.randn_like(o)
    auto = torch.autograd.grad((o * do).sum(), [q, k, v, ak, av, g1, g2])
    with torch.no_grad():
        mine = ref_cpu.attn_bwd(do, q, k, v, ak, av, g1, g2, vstart, F, cache)
    for a, m in zip(auto, mine):
        assert (a - m).abs().max() < 1e-12
