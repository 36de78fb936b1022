set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04ac; rm -rf $O; mkdir -p $O
cd $R
for i in 1 2 3; do for e in "A=0" "CG_OLD_STREAM_CLASSES=1"; do
  echo "[2^22 $e] $(env NO_EXTRAS=1 $e python scripts/session_leg.py 22 8 2>&1 | tail -1 | cut -c40-200)" >> $O/legs.txt
done; done
for lm in 16 18 20; do for e in "A=0" "CG_OLD_STREAM_CLASSES=1"; do
  echo "[2^$lm $e] $(env NO_EXTRAS=1 $e python scripts/session_leg.py $lm 10 2>&1 | tail -1 | cut -c40-200)" >> $O/legs.txt
done; done
cat $O/legs.txt
timeout 1200 python -m pytest tests/test_rep3_party_abi.py tests/test_synthetic_scale.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
