cd $GRAFT_REPO_ROOT; O=gpurun_out/r04ag; mkdir -p $O
for i in 1 2; do for e in "A=0" "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=2"; do
echo "$e leg $(env NO_EXTRAS=1 $e python scripts/session_leg.py 22 10 2>/dev/null | cut -c40-230)"
done; done
for e in "A=0" "GPU_MAX_HW_QUEUES=8"; do
env $e python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sizes 2>$O/err.txt | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); pe=d['product_entry']; print('$e bench  ', round(pe['ms_per_proof'],2), round(pe['ms_per_proof_mean_inner'],2), 'step', round(d['step_resident']['ms_per_step'],2))"
done
