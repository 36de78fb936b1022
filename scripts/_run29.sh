set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04ac; rm -rf $O; mkdir -p $O
cd $R
for i in 1 2 3; do for at in 0 1; do
  echo "[2^22 at=$at] $(NO_EXTRAS=1 CGH_AUX_AT=$at python scripts/session_leg.py 22 10 2>/dev/null | cut -c40-230)" >> $O/legs.txt
done; done
for lm in 16 18 20; do
  echo "[2^$lm] $(NO_EXTRAS=1 python scripts/session_leg.py $lm 12 2>/dev/null | cut -c40-230)" >> $O/legs.txt
done
cat $O/legs.txt
