set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04o; rm -rf $O; mkdir -p $O
cd $R
CGH_EMULATE_PRIMARY_ONLY=1 python scripts/multi_device_emulation.py 22 1,2,4,8 > $O/multi_device_emulation.txt 2>&1
for n in 2 4 8; do
  for r in $(seq 0 $((n-1))); do
    echo "N=$n rank $r: $(python bench.py --emulate $n:$r --steps 8 --warmup 2 --no-cpu-baseline --no-session 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],2), d['units'], d['vectors'])")" >> $O/plan_emulation.txt
  done
done
cat $O/multi_device_emulation.txt $O/plan_emulation.txt
