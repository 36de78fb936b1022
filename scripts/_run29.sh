set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04ac; rm -rf $O; mkdir -p $O
cd $R
for i in 1 2; do for pad in 0 61440; do
  echo "[2^22 pad=$pad] $(env NO_EXTRAS=1 CG_ACC_G1_PAD_LDS=$pad python scripts/session_leg.py 22 8 2>&1 | tail -1 | cut -c40-200)" >> $O/legs.txt
done; done
for pad in 0 61440; do
  CG_ACC_G1_PAD_LDS=$pad python bench.py --no-session --no-sizes --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pad=$pad step', d['ms_per_step'], 'acc_g1', d['isolated_ms']['acc_g1_ms'] if 'isolated_ms' in d else d.get('roofline',{}).get('launch_ms'))" >> $O/legs.txt
done
cat $O/legs.txt
