set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04ad; rm -rf $O; mkdir -p $O
cd $R
for i in 1 2; do for lm in 16 18 20 22; do for f in "A=0" "BENCH_CTX_FLAGS=1"; do
  r=$(env $f python bench.py --log-m $lm --no-session --no-sizes --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
  echo "2^$lm $f step_ms=$r" >> $O/sweep.txt
done; done; done
cat $O/sweep.txt
