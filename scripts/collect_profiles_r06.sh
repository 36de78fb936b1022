#!/bin/bash
# Round-6 evidence, run on the GPU box (gpurun): GPU test log, default bench line (compact line + bench_detail.json, with its in-run counter
# passes), the two-rank line on one GPU (with the preflight report), rocprofv3 kernel stats of the overlapped resident step, of every kernel
# alone on both curves and of the party entry (2^22 and the Poseidon fixture), PMC traffic with its calibration, SQ issue counters, clock per
# kernel, the multi-GPU emulations (every device, max over devices: needs the KNOBS=1 host library), the transforms alone, the MSM fuzz.
# scripts/profile_summary_r06.py condenses gpurun_out/r06c/ into profiles/.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06c
rm -rf $O; mkdir -p $O
cd $R
if [ "${SKIP_TESTS:-0}" != "1" ]; then timeout 2700 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; timeout 900 python -m pytest tests -q -m variant > $O/pytest_variant.txt 2>&1; fi
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err; cp bench_detail.json $O/bench_detail.json
timeout 900 python bench.py --gpus 2 --backend gloo --shared-device --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_n2_shared_line.json 2> $O/bench_n2_shared.err; cp bench_detail.json $O/bench_n2_shared_detail.json
timeout 400 python scripts/fuzz_msm.py 300 11 > $O/fuzz_msm.txt 2>&1
timeout 300 python scripts/ntt_timing.py 16,20,22,24 > $O/ntt_timing.txt 2>&1
# multi-GPU emulations (one GPU: no node)
COGROTH16_HOST_LIB=collaborative-circom_amd/libcogroth16_host_knobs.so timeout 900 python scripts/multi_device_emulation.py 22 1,2,4,8 > $O/multi_device_emulation.txt 2>&1
for w in 2 4 8; do for r in 0 $((w-1)); do timeout 300 python bench.py --emulate $w:$r --steps 10 --warmup 3 --no-session --no-cpu-baseline 2>/dev/null | cut -c1-700 >> $O/planner_emulation.txt; done; done
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-session > $O/stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -- python $R/scripts/serial_kernels.py 22 5 > $O/serial.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial_bls -- python $R/scripts/serial_kernels.py 22 3 bls12_381 > $O/serial_bls.log 2>&1
NO_EXTRAS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/session -- python $R/scripts/session_leg.py 22 5 > $O/session.log 2>&1
NO_EXTRAS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/session_poseidon -- python $R/scripts/session_leg.py poseidon 20 > $O/session_poseidon.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/scripts/serial_kernels.py 22 2 > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/scripts/serial_kernels.py 22 2 > $O/pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES \
    --output-format csv -d $O/pmc_sq -- python $R/scripts/serial_kernels.py 22 2 > $O/pmc_sq.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_clk -- python $R/scripts/serial_kernels.py 22 3 > $O/pmc_clk.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/cal_fetch -- $R/scripts/_build/pmc_calibrate > $O/cal_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/cal_write -- $R/scripts/_build/pmc_calibrate > $O/cal_write.log 2>&1
cd $R
python scripts/profile_summary_r06.py $O > $O/summary.log 2>&1
python scripts/proof_timeline.py $O/session_poseidon 2.6 > $O/timeline_poseidon.txt 2>/dev/null
find $O -name '*counter_collection.csv' -size +256k -delete
find $O -name '*kernel_trace.csv' -size +256k -delete
find $O -name '*agent_info.csv' -delete
tail -3 $O/pytest_gpu.txt 2>/dev/null; tail -2 $O/pytest_variant.txt 2>/dev/null; tail -1 $O/smoke.txt; head -c 600 $O/bench_line.json; echo; grep -o '"n_gpus":[0-9]*' $O/bench_n2_shared_line.json | tail -1; cat $O/fuzz_msm.txt | tail -2; cat $O/ntt_timing.txt | tail -4; tail -4 $O/multi_device_emulation.txt; tail -30 $O/summary.log
