#!/bin/bash
# small-circuit A/B, late aux start, precompute window by size (bitsum from 2^7 buckets), batched frees
set -u
export TMPDIR=/tmp NO_EXTRAS=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r05h}
mkdir -p $O
cd $R
run() { local tag=$1; shift; echo "== $tag" >> $O/ab.txt; env "$@" timeout 300 python scripts/session_leg.py $WHAT 20 2>&1 | grep -v "^/opt" | tail -n 1 | cut -c1-200 >> $O/ab.txt; }
WHAT=16
run "2^16 default (late aux)" X=1
run "2^16 CGH_NO_LATE_AUX" CGH_NO_LATE_AUX=1
run "2^16 late aux, one context" CGH_SECOND_CONTEXT_MIN=20
WHAT=14
run "2^14 default" X=1
run "2^14 CGH_NO_LATE_AUX" CGH_NO_LATE_AUX=1
run "2^14 window 14" BENCH_PRECOMPUTE=14
WHAT=poseidon
run "poseidon default" X=1
for c in 8 9 10 11 12 13; do run "poseidon window $c" BENCH_PRECOMPUTE=$c; done
run "poseidon window 10 two contexts" BENCH_PRECOMPUTE=10 CGH_SECOND_CONTEXT_MIN=5
WHAT=12
run "2^12 default" X=1
for c in 10 11 12 13 14; do run "2^12 window $c" BENCH_PRECOMPUTE=$c; done
cat $O/ab.txt
