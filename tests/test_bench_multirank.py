"""bench.py's multi-rank path on real hardware: several ranks share GPU 0 (gloo for the exchanges, staged through the host) and
must fold to exactly the points the single-rank run produces.  Covers plan_units, table slices, the h-owner-only witness map
(N = 2), the distributed witness map + all_to_all (N = 4, 8) and the all_gather + host EC-add fold."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def run_bench(tmp_path, world, extra=(), log_m=14, port=29611):
    out = str(tmp_path / f"res_{world}_{len(extra)}.npz")
    common = ["--gpus", str(world), "--steps", "1", "--warmup", "1", "--log-m", str(log_m), "--no-cpu-baseline", "--no-session", "--dump-result", out, *extra]
    if world == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), *common]
    else:
        import torch
        # a node with enough GPUs runs one rank per GPU over RCCL (what the driver's scaling run does); a one-GPU box puts every rank on
        # GPU 0 and stages the exchanges through the host (gloo)
        one_box = [] if torch.cuda.device_count() >= world else ["--backend", "gloo", "--shared-device"]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(port + world), os.path.join(ROOT, "bench.py"), *common, *one_box]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line), dict(np.load(out))


def test_bench_step_matches_oracle(tmp_path):
    """the step bench.py times computes the right thing: all ten MSM results of a 2^12 step equal the oracle's evaluation of the
    same inputs (SpMV, REP3 products, coset NTTs, h = ab - c, MSMs)"""
    sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
    import bench
    import bench_check
    assert bench_check.TABLE_FIRST == bench.TABLE_FIRST
    for extra, log_m in (((), 12), (("--precompute", "0"), 12), ((), 18)):       # 2^18: partition-sort path, window chosen by table size
        _, d = run_bench(tmp_path, 1, extra=("--dump-inputs",) + extra, log_m=log_m)
        exp = bench_check.expected_results(d)
        for t in bench.TABLES:
            np.testing.assert_array_equal(d[t], exp[t], err_msg=f"table {t} {extra} 2^{log_m}")


def test_bench_step_matches_oracle_at_full_size(tmp_path):
    """BASELINE size (2^22 constraints-domain): the exact workload bench.py times, checked bit for bit against the oracle's evaluation
    of the same inputs (the synthetic tables [(first + i) G] collapse every MSM to one generator multiplication, so the CPU side
    stays at a dozen 2^22 NTTs and vectorised field operations)"""
    sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
    import bench
    import bench_check
    _, d = run_bench(tmp_path, 1, extra=("--dump-inputs",), log_m=22)
    exp = bench_check.expected_results(d)
    for t in bench.TABLES:
        np.testing.assert_array_equal(d[t], exp[t], err_msg=f"table {t} at 2^22")


def test_bench_step_matches_oracle_at_2_24(tmp_path):
    """BASELINE configs[3] size on one GPU: the 2^24 step (2^24-point tables with their precomputed window copies, 2^24 transforms),
    all ten MSM results bit for bit against the oracle's evaluation of the same inputs.  What stays unexercised of configs[3] is RCCL
    between different devices."""
    if os.environ.get("CG_SKIP_2_24"):
        pytest.skip("CG_SKIP_2_24 set")
    sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
    import bench
    import bench_check
    _, d = run_bench(tmp_path, 1, extra=("--dump-inputs",), log_m=24)
    exp = bench_check.expected_results(d)
    for t in bench.TABLES:
        np.testing.assert_array_equal(d[t], exp[t], err_msg=f"table {t} at 2^24")


def test_eight_ranks_fold_at_2_20(tmp_path):
    """the 8-rank plan (table slices, distributed witness map, all_to_all, all_gather + fold) at 2^20, eight ranks sharing GPU 0"""
    _, r1 = run_bench(tmp_path, 1, log_m=20)
    _, r8 = run_bench(tmp_path, 8, log_m=20)
    for t in r1:
        assert r1[t].any()
        np.testing.assert_array_equal(r1[t], r8[t], err_msg=f"world 8 at 2^20, table {t}")


def test_single_rank_through_rccl_collectives(tmp_path):
    """one rank, but with torch.distributed (nccl = RCCL) initialised and the witness map in its distributed form: the all_to_all,
    all_gather, all_reduce and barrier calls of the N >= 4 path run on real device tensors and must not change the result"""
    _, ref = run_bench(tmp_path, 1)
    _, got = run_bench(tmp_path, 1, extra=("--force-dist",))
    for t in ref:
        np.testing.assert_array_equal(ref[t], got[t], err_msg=f"table {t}")


def test_ranks_fold_to_the_single_rank_result(tmp_path):
    j1, r1 = run_bench(tmp_path, 1)
    assert j1["n_gpus"] == 1 and set(r1) == {"h", "l", "a", "b1", "b2"}
    _, r1b = run_bench(tmp_path, 1, extra=("--one-context",))
    for t in r1:
        np.testing.assert_array_equal(r1[t], r1b[t], err_msg=f"one-context vs two-context, table {t}")
        assert r1[t].any()
    for world in (2, 4, 8):
        jw, rw = run_bench(tmp_path, world)
        assert jw["n_gpus"] == world
        for t in r1:
            np.testing.assert_array_equal(r1[t], rw[t], err_msg=f"world {world}, table {t}")


def test_gpus_flag_starts_its_own_ranks_and_keeps_the_entry_basis(tmp_path):
    """`python bench.py --gpus 2` with NO launcher around it (the shape of the driver's command) starts two ranks itself and prints a
    two-rank line whose `value` is on the N = 1 line's basis: one REP3 party through the product entry on a cgh_session_open_multi
    session over the job's devices.  On a one-GPU box the ranks share GPU 0 (gloo); a box with two GPUs runs RCCL between them."""
    import torch
    one_box = [] if torch.cuda.device_count() >= 2 else ["--backend", "gloo", "--shared-device"]
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--log-m", "14", "--no-cpu-baseline", *one_box]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    j = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 2 and j["rccl_ranks_seen"] == 2
    ent = j["product_entry"]
    assert "error" not in ent, ent
    assert len(ent["devices"]) == 2 and ent["three_parties_agree"]
    assert j["value_basis"].startswith("product entry over 2 GPUs")
    assert j["value"] == ent["value"] and j["step_resident"]["value"] > 0


def test_gpus_flag_refuses_a_box_with_too_few_gpus():
    """without --shared-device a rank count above the visible GPUs is an error, not a silent one-GPU line"""
    import torch
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "GPU(s) visible" in (p.stderr + p.stdout)
    # a launcher that started the wrong number of ranks is refused as well
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu-baseline"], cwd=ROOT, env=env2, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=1" in (p.stderr + p.stdout)
