#!/bin/bash
# small-circuit A/B runs with the existing knobs: what the chain's wait for workgroup slots, the context pair and the window cost at 2^16 / m = 256
set -u
export TMPDIR=/tmp NO_EXTRAS=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r05d}
mkdir -p $O
cd $R
run() { local tag=$1; shift; echo "== $tag" >> $O/ab.txt; env "$@" timeout 300 python scripts/session_leg.py $WHAT 20 2>&1 | grep -v "^/opt" | tail -n 1 | cut -c1-220 >> $O/ab.txt; }
WHAT=16
run "2^16 default" X=1
run "2^16 CG_MSM_CHUNK=32" CG_MSM_CHUNK=32
run "2^16 CG_MSM_CHUNK=16" CG_MSM_CHUNK=16
run "2^16 one context" CGH_SECOND_CONTEXT_MIN=20
run "2^16 one context, chunk 32" CGH_SECOND_CONTEXT_MIN=20 CG_MSM_CHUNK=32
run "2^16 no wide" CG_MSM_WIDE_SMALL=0
run "2^16 no wide chunk 32" CG_MSM_WIDE_SMALL=0 CG_MSM_CHUNK=32
run "2^16 window 14" BENCH_PRECOMPUTE=14
run "2^16 window 13 chunk 32" BENCH_PRECOMPUTE=13 CG_MSM_CHUNK=32
WHAT=poseidon
run "poseidon default" X=1
run "poseidon two contexts" CGH_SECOND_CONTEXT_MIN=5
run "poseidon window 13" BENCH_PRECOMPUTE=13
run "poseidon no precompute" BENCH_PRECOMPUTE=0
run "poseidon window 13 two contexts" BENCH_PRECOMPUTE=13 CGH_SECOND_CONTEXT_MIN=5
WHAT=12
run "2^12 default" X=1
run "2^12 window 13" BENCH_PRECOMPUTE=13
run "2^12 no precompute" BENCH_PRECOMPUTE=0
cat $O/ab.txt
