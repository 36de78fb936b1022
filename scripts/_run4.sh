set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04d; rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "msm" > $O/pytest_msm.txt 2>&1
timeout 900 python -m pytest tests/test_host_mirror.py tests/test_rep3_party_abi.py tests/test_chacha_rand.py tests/test_synthetic_scale.py -m gpu -x -q > $O/pytest_host.txt 2>&1
for i in 1 2 3; do
 for cfg in "A=0" "CG_MSM_TABLE_ORDER=1 G2LAST=1" "CG_DEBUG_NO_REDUCE=1" "CG_MSM_REDUCE_BATCH=0"; do
  extra=""; case "$cfg" in *G2LAST*) extra="--g2-last";; esac
  echo "[$cfg] $(env $cfg python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-session $extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],2))")" >> $O/step_ab.txt
 done
done
for i in 1 2 3; do
 for cfg in "A=0" "CGH_G2_ORDER=first" "CG_MSM_REDUCE_BATCH=0" "CGH_G2_ORDER=first CG_MSM_REDUCE_BATCH=0"; do
  echo "[$cfg] $(env NO_EXTRAS=1 $cfg python scripts/session_leg.py 22 10 2>/dev/null)" >> $O/entry_ab.txt
 done
done
tail -3 $O/pytest_msm.txt; tail -3 $O/pytest_host.txt; cat $O/step_ab.txt $O/entry_ab.txt
