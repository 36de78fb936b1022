#!/bin/bash
# Round-4 evidence, run on the GPU box (gpurun): GPU test log, default bench line, the second-curve line, rocprofv3 kernel stats of the
# overlapped resident step, of every kernel alone on both curves (scripts/serial_kernels.py) and of the party entry, PMC traffic
# (FETCH_SIZE / WRITE_SIZE, separate passes) with a calibration on known byte counts, SQ issue counters, sustained clock per kernel,
# the MSM fuzz.  scripts/profile_summary_r04.py condenses gpurun_out/r04/ into the files kept under profiles/.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04
rm -rf $O; mkdir -p $O
cd $R
if [ "${SKIP_TESTS:-0}" != "1" ]; then timeout 2700 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; fi
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
timeout 900 python bench.py --curve bls12_381 --steps 10 --warmup 3 --no-sizes --no-cpu-baseline > $O/bench_bls.json 2> $O/bench_bls.err
timeout 400 python scripts/fuzz_msm.py 300 11 > $O/fuzz_msm.txt 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-session > $O/stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -- python $R/scripts/serial_kernels.py 22 5 > $O/serial.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial_bls -- python $R/scripts/serial_kernels.py 22 3 bls12_381 > $O/serial_bls.log 2>&1
NO_EXTRAS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/session -- python $R/scripts/session_leg.py 22 5 > $O/session.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/scripts/serial_kernels.py 22 2 > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/scripts/serial_kernels.py 22 2 > $O/pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES \
    --output-format csv -d $O/pmc_sq -- python $R/scripts/serial_kernels.py 22 2 > $O/pmc_sq.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_clk -- python $R/scripts/serial_kernels.py 22 3 > $O/pmc_clk.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/cal_fetch -- $R/scripts/_build/pmc_calibrate > $O/cal_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/cal_write -- $R/scripts/_build/pmc_calibrate > $O/cal_write.log 2>&1
cd $R
python scripts/profile_summary_r04.py $O > $O/summary.log 2>&1
find $O -name '*counter_collection.csv' -size +256k -delete
find $O -name '*kernel_trace.csv' -size +256k -delete
find $O -name '*agent_info.csv' -delete
tail -3 $O/pytest_gpu.txt 2>/dev/null; tail -1 $O/smoke.txt; head -c 400 $O/bench.json; echo; head -c 300 $O/bench_bls.json; echo; cat $O/fuzz_msm.txt; tail -30 $O/summary.log
