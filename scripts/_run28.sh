set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04ab; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
NO_EXTRAS=1 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tr16 -o t -- python $R/scripts/session_leg.py 16 10 > $O/leg16.txt 2>&1
python $R/scripts/solo_timeline_detail.py $O/tr16 0 6 6 > $O/timeline16.txt 2>&1
NO_EXTRAS=1 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tr18 -o t -- python $R/scripts/session_leg.py 18 10 > $O/leg18.txt 2>&1
python $R/scripts/solo_timeline_detail.py $O/tr18 0 12 12 > $O/timeline18.txt 2>&1
rm -rf $O/tr16 $O/tr18
tail -3 $O/leg16.txt | cut -c1-300
