"""N > 1 path of bench.py on CPU: world_size-2 gloo. bench.plan_units() assigns MSM work units (whole tables or contiguous
table slices) to ranks; each rank produces the results of its units (here the oracle stands in for the GPU kernels) and
bench.exchange() all_gathers them and folds the slices of every table with the product's host EC addition — the five results
must equal the unsharded MSMs on every rank."""
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
    plan = bench.plan_units(world)
    tables, scal, full = {}, {}, {}
    for t in bench.TABLES:
        group = orc.G1 if bench.TABLE_GROUP[t] == 0 else orc.G2
        tables[t] = np.stack([orc.generator_mul(curve, group, s) for s in orc.random_field(curve, orc.FR, n, rng)])
    scal["h"] = [orc.random_field(curve, orc.FR, n, rng) for _ in range(2)]
    scal["aux"] = [orc.random_field(curve, orc.FR, n, rng) for _ in range(2)]
    results = {}
    for (t, i, parts, owner) in plan:
        group = orc.G1 if bench.TABLE_GROUP[t] == 0 else orc.G2
        sc = scal["h" if t == "h" else "aux"]
        if t not in full:
            full[t] = np.stack([orc.msm(curve, group, tables[t], s) for s in sc])
        if owner != rank:
            continue
        lo, hi = bench.shard_range(n, i, parts)           # the oracle stands in for the GPU kernels on this unit
        results[(t, i, parts)] = np.stack([cg.point_from_affine(curve, group, orc.msm(curve, group, tables[t][lo:hi], s[lo:hi])) for s in sc])
    final = bench.exchange(results, plan, bench.Comm(dist, world, torch.device("cpu")))
    ok = set(final.keys()) == set(bench.TABLES)
    for t in bench.TABLES:
        group = orc.G1 if bench.TABLE_GROUP[t] == 0 else orc.G2
        for j in range(2):
            ok &= bool(np.array_equal(cg.point_to_affine(curve, group, final[t][j]), full[t][j]))
    q.put((rank, ok, len(results)))
    dist.barrier()
    dist.destroy_process_group()


def test_plan_units_covers_every_table_once():
    sys.path.insert(0, ROOT)
    import bench
    for world in (1, 2, 3, 4, 8):
        plan = bench.plan_units(world)
        for t in bench.TABLES:
            mine = sorted((i, parts) for (tt, i, parts, o) in plan if tt == t)
            parts = mine[0][1]
            assert mine == [(i, parts) for i in range(parts)]            # each table: slices 0..parts-1 exactly once
        owners = [o for (_, _, _, o) in plan]
        assert set(owners) == set(range(world))                           # every rank gets work
        plan2, load = bench.plan_units(world, with_load=True)
        assert plan2 == plan                                              # deterministic: every rank computes the same plan
        assert max(load) <= 1.35 * (sum(load) / world) + 1e-9             # balanced within 35 % under the planner's cost model
    # more ranks never make the modelled critical path longer
    crit = [max(bench.plan_units(w, with_load=True)[1]) for w in (1, 2, 4, 8)]
    assert all(crit[i + 1] < crit[i] for i in range(3))


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
    procs = [ctx.Process(target=_worker, args=(r, world, port, 23, q)) for r in range(world)]
    for p in procs: p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs: p.join(timeout=60)
    assert sorted((r, ok) for r, ok, _ in got) == [(0, True), (1, True)]
    assert all(nunits > 0 for _, _, nunits in got)


def _wm_worker(rank, world, port, m, q):
    sys.path.insert(0, HERE); sys.path.insert(0, ROOT)
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = bench.Comm(dist, world, torch.device("cpu"))
    plan = bench.plan_units(world)
    h_units = [(i, parts, owner) + bench.shard_range(m, i, parts) for (t, i, parts, owner) in plan if t == "h"]
    my_vecs = [v for v in range(bench.WM_VECTORS) if bench.wm_vector_owner(v, world) == rank]
    # vector v, row i, limb j = 1000 * i + 10 * v + j on its owner; poison everywhere else (must never be sent)
    rows = torch.arange(m, dtype=torch.int64).reshape(m, 1) * 1000 + torch.arange(4, dtype=torch.int64).reshape(1, 4)
    vecs = [(rows + 10 * v) if v in my_vecs else torch.full((m, 4), -1, dtype=torch.int64) for v in range(bench.WM_VECTORS)]
    sl = bench.wm_exchange(comm, vecs, my_vecs, h_units, world, rank, m)
    (lo, hi), = [(lo, hi) for (_, _, owner, lo, hi) in h_units if owner == rank]
    ok = sorted(sl) == list(range(bench.WM_VECTORS))
    for v in range(bench.WM_VECTORS):
        ok &= bool(torch.equal(sl[v], rows[lo:hi] + 10 * v))
    q.put((rank, ok, hi - lo))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,m", [(4, 64), (4, 61), (5, 128)])
def test_distributed_witness_map_exchange(world, m):
    """world >= 4: vector pipelines live on different ranks; one all_to_all must hand every rank rows [lo, hi) of all six
    vectors for its own h slice (bench.wm_exchange), for even and ragged splits"""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_wm_worker, args=(r, world, port, m, q)) for r in range(world)]
    for p in procs: p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs: p.join(timeout=60)
    assert sorted(r for r, _, _ in got) == list(range(world))
    assert all(ok for _, ok, _ in got)
    assert sum(n for _, _, n in got) == m


def test_plan_h_split_matches_all_to_all():
    sys.path.insert(0, ROOT)
    import bench
    for world in (4, 5, 8):
        plan = bench.plan_units(world)
        hs = [(i, parts, o) for (t, i, parts, o) in plan if t == "h"]
        assert hs == [(i, world, i) for i in range(world)]                 # slice i of h lives on rank i
        m = 1 << 12
        hu = [(i, parts, o) + bench.shard_range(m, i, parts) for (i, parts, o) in hs]
        sp = [bench.a2a_splits(world, r, hu, m) for r in range(world)]
        for s in range(world):
            for r in range(world):
                assert sp[s][0][r] == sp[r][1][s]                          # what s sends to r is what r expects from s
    for world in (1, 2, 3):
        assert sum(1 for (t, _, _, _) in bench.plan_units(world) if t == "h") == 1   # one rank owns h and runs the witness map
