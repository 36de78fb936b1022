"""N > 1 path of bench.py on CPU: world_size-2 gloo. Each rank owns a contiguous point range (shard_range), produces the
partial MSM results of its shard (here with the oracle standing in for the GPU kernels), and bench.exchange() all_gathers the
partial Jacobian points and folds them with the product's host EC addition — the result must equal the unsharded MSM."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n, q):
    sys.path.insert(0, HERE); sys.path.insert(0, ROOT)
    import oracle_lib as orc
    import bench
    cg = bench.cg
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    curve = orc.BN254
    rng = np.random.default_rng(1234)                     # same inputs on every rank
    ks = orc.random_field(curve, orc.FR, n, rng)
    results, full = [], []
    for group, k in ((orc.G1, 2), (orc.G2, 2)):
        pts = np.stack([orc.generator_mul(curve, group, s) for s in ks])
        scal = [orc.random_field(curve, orc.FR, n, rng) for _ in range(k)]
        lo, hi = bench.shard_range(n, rank, world)
        part = np.stack([cg.point_from_affine(curve, group, orc.msm(curve, group, pts[lo:hi], s[lo:hi])) for s in scal])
        results.append(part)
        full.append(np.stack([orc.msm(curve, group, pts, s) for s in scal]))
    combined = bench.exchange(results, dist, world, torch.device("cpu"))
    ok = True
    for r, f, group in zip(combined, full, (orc.G1, orc.G2)):
        for j in range(r.shape[0]):
            ok &= bool(np.array_equal(cg.point_to_affine(curve, group, r[j]), f[j]))
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions():
    sys.path.insert(0, ROOT)
    import bench
    for n in (0, 1, 7, 4194302, 4194304):
        for world in (1, 2, 3, 4, 8):
            rs = [bench.shard_range(n, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in rs]
            assert max(sizes) - min(sizes) <= 1


def test_msm_shard_allgather_combine_world2():
    from product import ensure_built
    ensure_built()
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 37, q)) for r in range(world)]
    for p in procs: p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs: p.join(timeout=60)
    assert sorted(got) == [(0, True), (1, True)]
