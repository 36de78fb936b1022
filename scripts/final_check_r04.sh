#!/bin/bash
# Last step of round 4 on the GPU box: the GPU test-suite, smoke() and the default bench line of the final tree (what the driver runs at round end).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04_final; rm -rf $O; mkdir -p $O; cd $R
timeout 2700 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
tail -2 $O/pytest_gpu.txt; tail -1 $O/smoke.txt; head -c 330 $O/bench_default.json; echo; head -c 330 $O/bench.json; echo
