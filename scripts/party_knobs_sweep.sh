#!/bin/bash
# One REP3 party through the entry at 2^22 under one tuning knob at a time (same box, alternating with the default), now that a party
# proving alone always runs on the same context pair (proof-to-proof spread +-0.3 ms).  usage (gpurun): bash scripts/party_knobs_sweep.sh [proofs=8]
cd ${GRAFT_REPO_ROOT:-.}; O=gpurun_out/knobs; mkdir -p $O; : > $O/sweep.txt
P=${1:-8}
run() { echo "[$1] $(env NO_EXTRAS=1 $1 python scripts/session_leg.py 22 $P 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['each'][1:]; print('mean(after first) %.2f  min %.2f  each %s' % (sum(e)/len(e), min(e), d['each']))")" | tee -a $O/sweep.txt; }
for rep in 1 2; do
run "A=0"
run "CGH_G2_AFTER=0"; run "CGH_G2_AFTER=1"; run "CGH_G2_AFTER=3"; run "CGH_G2_AFTER=4"; run "CGH_G2_AFTER=6"
run "A=0"
run "CG_MSM_REDUCE_BATCH=1"; run "CG_MSM_ACC_SLOTS=6"; run "CG_MSM_ACC_SLOTS=8"
run "CGH_BULK_CHUNK=0"; run "CGH_BULK_CHUNK=96"; run "CG_G2_CHUNK=96"; run "CG_G2_CHUNK=48"
run "A=0"
run "CG_MSM_CHUNK=160"; run "CG_MSM_CHUNK=96"; run "CG_BULK_CLASS=0"
done
