"""bench.py prints ONE compact line the driver can parse (BENCH_r05.json: parsed = null because the line had grown to 21.9 KB); the full
object goes to bench_detail.json.  CPU-only: the line builder is fed the full objects of earlier GPU runs (profiles/bench_r0*.json)."""
import glob
import importlib.util
import io
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


FULL = sorted(glob.glob(os.path.join(ROOT, "profiles", "bench_r0[3-9]*.json")) + glob.glob(os.path.join(ROOT, "profiles", "r0*_multigpu_bench_*.json")))


@pytest.mark.parametrize("path", FULL, ids=[os.path.basename(p) for p in FULL])
def test_compact_line_is_small_and_carries_the_contract(bench, path, tmp_path, monkeypatch):
    with open(path) as f:
        full = json.load(f)
    if "metric" not in full or "detail" in full:
        pytest.skip("not a full bench object (a compact line or another record)")
    detail = tmp_path / "bench_detail.json"
    buf = io.StringIO()
    monkeypatch.setattr(sys, "stdout", buf)
    bench.emit(full, str(detail))
    monkeypatch.undo()
    lines = buf.getvalue().splitlines()
    assert len(lines) == 1
    assert len(lines[0].encode()) < 8192
    line = json.loads(lines[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in line, key
    assert "workload" in line["config"] and "model" not in line["config"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in line["roofline"], key
    if "cpu_baseline" in full and "value" in full["cpu_baseline"]:
        for key in ("value", "unit", "cores", "kind", "sample"):
            assert key in line["cpu_baseline"], key
    assert abs(line["value"] - full["value"]) <= 1e-5 * full["value"]
    assert abs(line["ms_per_step"] - full["ms_per_step"]) <= 1e-5 * full["ms_per_step"]
    with open(detail) as f:
        assert json.load(f) == full


def test_compact_line_sheds_legs_before_it_breaks_the_limit(bench):
    with open(os.path.join(ROOT, "profiles", "bench_r05_final.json")) as f:
        full = json.load(f)
    full["sizes"] = {"2^%d" % k: dict(step_resident={"ms_per_step": 1.0}, product_entry={"ms_per_proof": 1.0, "value": 1.0}, roofline={"frac": 0.1}) for k in range(400)}
    text = bench.compact_line(full, "bench_detail.json")
    assert len(text) <= bench.COMPACT_LIMIT
    line = json.loads(text)
    assert "sizes" not in line and "roofline" in line and "cpu_baseline" in line


def test_cpu_twins_of_the_headline_legs_run_on_the_host():
    """cpu_baseline's twins: one oracle party on the reference's bench circuit, and the host mask draws of one proof"""
    import oracle_lib as orc
    fx = os.path.join(ROOT, "tests", "golden", "groth16", "bn254", "poseidon")
    t, st = orc.bench_rep3_party_file(orc.BN254, os.path.join(fx, "circuit.zkey"), os.path.join(fx, "witness.wtns"), 2, 2)
    assert 0 < t < 5 and set(st) == {"rows_products_s", "ntt_s", "msm_g1_s", "msm_g2_tail_s", "mask_draws_s"}
    assert abs(sum(st.values()) - t) < 0.2 * t + 1e-3
    assert 0 < orc.bench_mask_draws(orc.BN254, 1 << 10) < 1
