"""bench.py contract (driver-facing): one JSON line with the agreed keys, roofline and timing fields sane.
Runs the real script in a child process at reduced depth (the line is then marked invalid by the script itself)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_emits_the_contract_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--n_layers", "2", "--steps", "3", "--warmup",
                        "1", "--no_cpu_baseline"], capture_output=True, text=True, timeout=500, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    assert [ln for ln in r.stdout.splitlines() if ln.strip()] == lines, r.stdout[-2000:]   # nothing else on stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["unit"] == "samples/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "bf16"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 8 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and 0 < rf["frac"] < 1
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9


@pytest.mark.gpu
def test_bench_fails_on_a_set_gemm_error_word():
    """A timed-out split-K exchange during the run (the workspace's sticky error word, injected after the warm-up) must
    not produce a valid-looking line: every optimizer step of the timed region is skipped on the device, the JSON says
    `invalid`, the exit status is non-zero."""
    env = dict(os.environ, FVQA_BENCH_INJECT_GEMM_ERROR="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--n_layers", "2", "--steps", "3", "--warmup",
                        "1", "--no_cpu_baseline"], capture_output=True, text=True, timeout=500, cwd=ROOT, env=env)
    assert r.returncode != 0, r.stdout[-1000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert "split-K exchange" in d["invalid"]
