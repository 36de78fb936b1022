# A/B of a scheduling choice of a REP3 party's two contexts: bench.py's session leg (one party through cgh_session_prove_rep3_party, mean
# and min of 5 proofs) under each setting, alternating, on ONE box.  usage: bash scripts/party_knobs_ab.sh "CGH_G2_LAST=1" [repeats]
ALT="${1:-CGH_G2_LAST=1}"; N=${2:-2}
for i in $(seq 1 $N); do for cfg in "" "$ALT"; do
  env $cfg python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); s=d['session']; print('[$cfg]', '| step', round(d['ms_per_step'],1), '| party mean', round(s['rep3_party_ms'],1), 'min', round(s['rep3_party_ms_min'],1), '| plain', round(s['plain_ms'],1), '| three', round(s['rep3_three_parties_one_gpu_ms'],1))"
done; done
