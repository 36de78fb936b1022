#!/bin/bash
# Runs on the GPU box (via gpurun): GPU test log, default bench line, rocprofv3 kernel stats and the two PMC passes.
# Outputs land in gpurun_out/final/; scripts/pmc_summary.py condenses them into profiles/.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_write.log 2>&1
cd $R
python scripts/pmc_summary.py $O > $O/summary.log 2>&1
# the raw counter csvs are large; keep only the condensed files
find $O -name '*counter_collection.csv' -size +1M -delete
find $O -name '*kernel_trace.csv' -size +1M -delete
tail -3 $O/pytest_gpu.txt; cat $O/smoke.txt | tail -1; cat $O/bench.json; cat $O/summary.log | tail -20
