#!/bin/bash
# Round-5 baseline on this round's boxes, before any change: default bench line, in-library multi-device emulation, planner emulation.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r05a}
mkdir -p $O
cd $R
timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
CGH_EMULATE_PRIMARY_ONLY=1 timeout 900 python scripts/multi_device_emulation.py 22 1,2,4,8 > $O/multi_device_emulation.txt 2>&1
for r in 0 7; do timeout 300 python bench.py --emulate 8:$r --steps 10 --warmup 3 --no-session --no-cpu-baseline >> $O/planner_emulate8.txt 2>&1; done
head -c 600 $O/bench.json; echo; cat $O/multi_device_emulation.txt | tail -5; tail -2 $O/planner_emulate8.txt | cut -c1-400
