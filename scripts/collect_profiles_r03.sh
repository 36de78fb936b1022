#!/bin/bash
# Round-3 evidence, run on the GPU box (gpurun): GPU test log, default bench line, rocprofv3 kernel stats of the overlapped step AND of
# every kernel alone (scripts/serial_kernels.py), PMC traffic (FETCH_SIZE / WRITE_SIZE, separate passes) with a calibration on known
# byte counts, SQ issue counters, sustained clock per kernel.  scripts/profile_summary_r03.py condenses gpurun_out/r03/ into profiles/.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03
rm -rf $O; mkdir -p $O
cd $R
if [ "${SKIP_TESTS:-0}" != "1" ]; then timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; fi
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
scripts/_build/microbench_clock > $O/microbench_clock.txt 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-session > $O/stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -- python $R/scripts/serial_kernels.py 22 5 > $O/serial.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/scripts/serial_kernels.py 22 2 > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/scripts/serial_kernels.py 22 2 > $O/pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES \
    --output-format csv -d $O/pmc_sq -- python $R/scripts/serial_kernels.py 22 2 > $O/pmc_sq.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_clk -- python $R/scripts/serial_kernels.py 22 3 > $O/pmc_clk.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/cal_fetch -- $R/scripts/_build/pmc_calibrate > $O/cal_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/cal_write -- $R/scripts/_build/pmc_calibrate > $O/cal_write.log 2>&1
cd $R
python scripts/profile_summary_r03.py $O > $O/summary.log 2>&1
find $O -name '*counter_collection.csv' -size +256k -delete
find $O -name '*kernel_trace.csv' -size +256k -delete
find $O -name '*agent_info.csv' -delete
tail -3 $O/pytest_gpu.txt 2>/dev/null; tail -1 $O/smoke.txt; head -c 600 $O/bench.json; echo; tail -40 $O/summary.log
# Rep3Rand's draws on the device (csrc/chacha_rand.hip): per-kernel times and HBM counters (separate passes)
C=$R/gpurun_out/r03/chacha; mkdir -p $C
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $C/stats -- python $R/scripts/chacha_timing.py > $C/timing.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $C/pmc_fetch -- python $R/scripts/chacha_timing.py > $C/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $C/pmc_write -- python $R/scripts/chacha_timing.py > $C/pmc_write.log 2>&1
cd $R
python scripts/chacha_pmc_summary.py $C > $C/summary.txt 2>&1
find $C -name '*counter_collection.csv' -size +256k -delete
find $C -name '*kernel_trace.csv' -size +256k -delete
find $C -name '*agent_info.csv' -delete
grep draws $C/timing.txt; cat $C/summary.txt
