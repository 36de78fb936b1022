set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04y; rm -rf $O; mkdir -p $O
cd $R
timeout 2700 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1
tail -15 $O/pytest_gpu.txt
