set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04z; rm -rf $O; mkdir -p $O
cd $R
CGH_TIMING=1 CGH_EMULATE_PRIMARY_ONLY=1 python scripts/multi_device_emulation.py 22 8 > $O/md8.txt 2>&1
tail -12 $O/md8.txt
