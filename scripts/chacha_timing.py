"""Device draws of Rep3Rand's masking elements (cg_chacha12_fr_rand_dev) against the host library's single-thread draws, both curves.
Run under `rocprofv3 --kernel-trace --stats` for the per-kernel times (profiles/r03_chacha_kernel_stats.csv)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from product import cg, ensure_built  # noqa: E402

ensure_built()
ctx = cg.Context()
seed = bytes(range(11, 43))
for curve, name in ((cg.BN254, "BN254"), (cg.BLS12_381, "BLS12-381")):
    for log_n in (16, 20, 22, 24):
        n = 1 << log_n
        buf, after = ctx.chacha12_fr_rand(curve, seed, 0, n); buf.free()              # warm-up
        ts = []
        for _ in range(10):
            t0 = time.perf_counter(); buf, after = ctx.chacha12_fr_rand(curve, seed, 0, n); ts.append(time.perf_counter() - t0); buf.free()
        line = f"{name} Fr, 2^{log_n} draws: device {min(ts) * 1e3:.3f} ms (mean {sum(ts) / len(ts) * 1e3:.3f}), {8 * n / after:.4f} accepted per candidate"
        if log_n <= 22:
            t0 = time.perf_counter(); want, wa = cg.chacha12_fr_rand_host(curve, seed, 0, n); th = time.perf_counter() - t0
            buf, after = ctx.chacha12_fr_rand(curve, seed, 0, n); got = buf.download((n, 4)); buf.free()
            assert wa == after and (got == want).all()
            line += f"; host, one thread {th * 1e3:.1f} ms ({th / min(ts):.0f}x); equal"
        print(line, flush=True)
ctx.close()
