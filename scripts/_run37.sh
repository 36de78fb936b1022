cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_final; mkdir -p $O
timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
head -c 300 $O/bench_default.json; echo; head -c 300 $O/bench.json; echo
