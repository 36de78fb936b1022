set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04s; rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "msm" > $O/pytest_msm.txt 2>&1
timeout 900 python -m pytest tests/test_host_mirror.py tests/test_rep3_party_abi.py tests/test_synthetic_scale.py tests/test_shamir.py tests/test_plonk_rounds.py -m gpu -x -q > $O/pytest_host.txt 2>&1
for lm in 14 16 17 18; do
 for cfg in "CG_MSM_WIDE_SMALL=0" "CG_MSM_WIDE_SMALL=1"; do
  echo "[2^$lm $cfg] step $(env $cfg python bench.py --log-m $lm --steps 20 --warmup 5 --no-cpu-baseline --no-session 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],3))") entry $(env NO_EXTRAS=1 $cfg python scripts/session_leg.py $lm 10 2>/dev/null | cut -c1-140)" >> $O/wide.txt
 done
done
echo "[2^22] step $(python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-session 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],3), d['isolated_ms']['acc_g1_ms'], d['isolated_ms']['acc_g2_ms'])")" >> $O/wide.txt
tail -3 $O/pytest_msm.txt; tail -3 $O/pytest_host.txt; cat $O/wide.txt
