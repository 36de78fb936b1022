set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04r; rm -rf $O; mkdir -p $O
cd $R
for i in 1 2 3; do
 for cfg in "CG_MSM_REDUCE_BATCH=2" "CG_MSM_REDUCE_BATCH=3" "CG_DEBUG_NO_REDUCE=1" "CG_DEBUG_NO_REDUCE=2"; do
  echo "[$cfg] $(env $cfg python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-session 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],2))")" >> $O/step_ab.txt
 done
done
cat $O/step_ab.txt
