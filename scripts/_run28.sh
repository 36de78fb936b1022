set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04ab; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
NO_EXTRAS=1 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tr22 -o t -- python $R/scripts/session_leg.py 22 4 > $O/leg22.txt 2>&1
python $R/scripts/solo_timeline_detail.py $O/tr22 0 80 80 > $O/detail22.txt 2>&1
rm -rf $O/tr22
