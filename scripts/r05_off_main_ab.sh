#!/bin/bash
# A/B of CG_MSM_OFF_MAIN_LOG (tiny wide MSM calls accumulate off the main stream): the REP3 party entry at small sizes, own process per run, and as
# later legs of one process.  usage: bash scripts/r05_off_main_ab.sh > profiles/r05_small_circuit_off_main_ab.txt
for rep in 1 2; do
for what in poseidon 12 14 16; do
  for knob in 19 0; do
    echo "== $what CG_MSM_OFF_MAIN_LOG=$knob (run $rep)"
    NO_EXTRAS=1 CG_MSM_OFF_MAIN_LOG=$knob timeout 300 python scripts/session_leg.py $what 20 2>/dev/null | cut -c1-400
  done
done
done
for knob in 19 0 19 0; do
  echo "== later legs, CG_MSM_OFF_MAIN_LOG=$knob"
  CG_MSM_OFF_MAIN_LOG=$knob timeout 600 python scripts/later_leg.py 2>/dev/null
done
