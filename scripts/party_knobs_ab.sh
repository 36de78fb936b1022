# A/B of a scheduling choice of a REP3 party's two contexts: bench.py's entry leg (one party through cgh_session_prove_rep3_party_ex, its
# draws inside the call; mean of 6 proofs between barriers and best single proof) under each setting, alternating, on ONE box.
# usage: bash scripts/party_knobs_ab.sh "CGH_G2_LAST=1" [repeats] [log_m]
ALT="${1:-CGH_G2_LAST=1}"; N=${2:-2}; LM=${3:-22}
for i in $(seq 1 $N); do for cfg in "" "$ALT"; do
  env NO_EXTRAS=1 $cfg python scripts/session_leg.py $LM 6 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('[$cfg]', '| party mean', d['ms_per_proof'], 'best', d['ms_per_proof_min_inner'], '| three', d.get('rep3_three_parties_one_gpu_ms'))"
done; done
