set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04p; rm -rf $O; mkdir -p $O
cd $R
python scripts/sweep_precompute.py 16,17,18,19,20,21 13,15,16,17,18,19,20 > $O/sweep_precompute.txt 2>&1
python scripts/unit_cost_table.py > $O/unit_cost.txt 2>&1
cat $O/sweep_precompute.txt $O/unit_cost.txt
