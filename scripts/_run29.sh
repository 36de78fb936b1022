set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04ac; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_chacha_rand.py tests/test_rep3_party_abi.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for i in 1 2; do for lm in 16 18 22; do
  echo "[2^$lm] $(NO_EXTRAS=1 python scripts/session_leg.py $lm 10 2>/dev/null | cut -c1-200)" >> $O/legs.txt
done; done
cat $O/legs.txt
