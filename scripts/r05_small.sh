#!/bin/bash
# small-circuit timelines: the product entry on the Poseidon fixture (m = 256) and at 2^16, with the host-side marks (CGH_TIMING)
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r05b}
mkdir -p $O
cd $R
for what in poseidon 12 16; do
  NO_EXTRAS=1 timeout 300 python scripts/session_leg.py $what 20 > $O/entry_$what.txt 2>&1
  NO_EXTRAS=1 CGH_TIMING=1 timeout 300 python scripts/session_leg.py $what 4 > $O/entry_marks_$what.txt 2>&1
done
tail -n 2 $O/entry_*.txt | cut -c1-1500
