set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04ac; rm -rf $O; mkdir -p $O
cd $R
for i in 1 2; do for lm in 16 18 20 22; do for rs in 0 16 32 64; do
  if [ $rs = 0 ]; then e="A=0"; else e="CG_BULK_CU_RESERVE=$rs"; fi
  echo "[2^$lm reserve=$rs] $(env NO_EXTRAS=1 $e python scripts/session_leg.py $lm 10 2>&1 | tail -1 | cut -c40-200)" >> $O/legs.txt
done; done; done
cat $O/legs.txt
