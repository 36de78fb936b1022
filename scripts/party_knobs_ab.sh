# A/B of the scheduling knobs of a REP3 party's two contexts (chain / bulk priorities, bulk chunk length, sliced G2 launches): bench.py's
# session leg (one party through cgh_session_prove_rep3_party, mean and min of 5 proofs) under each setting, on ONE box.
for cfg in "" "CGH_BULK_CHUNK=128" "CGH_BULK_CHUNK=128" "CGH_BULK_CHUNK=96" "CGH_BULK_CHUNK=128 CG_G2_NO_SLICE=1" "CGH_BULK_CHUNK=128 CGH_BULK_FLAG=0" "CGH_BULK_CHUNK=128 CGH_CHAIN_FLAG=0" "CGH_BULK_FLAG=0" "${EXTRA_CFG:-}"; do
  env $cfg python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); s=d['session']; print('$cfg', '| step', round(d['ms_per_step'],1), '| party mean', round(s['rep3_party_ms'],1), 'min', round(s['rep3_party_ms_min'],1), '| plain', round(s['plain_ms'],1), '| three', round(s['rep3_three_parties_one_gpu_ms'],1))"
done
