set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04ad; rm -rf $O; mkdir -p $O
cd $R
for lm in 16 18 20; do for c in 13 14 15 16 17 18 19; do
  r=$(python bench.py --log-m $lm --precompute $c --no-session --no-sizes --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
  echo "2^$lm c=$c step_ms=$r" >> $O/sweep.txt
done; done
cat $O/sweep.txt
