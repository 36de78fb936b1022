set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04ab; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
NO_EXTRAS=1 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tr16 -o t -- python $R/scripts/session_leg.py 16 9 > $O/leg16.txt 2>&1
python $R/scripts/proof_phases.py $O/tr16 9 > $O/phases16.txt 2>&1
python $R/scripts/solo_timeline_detail.py $O/tr16 0 40 40 > $O/timeline16.txt 2>&1
rm -rf $O/tr16
