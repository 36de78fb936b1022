#!/bin/bash
# Times the REAL reference (co-circom generate-proof groth16, three parties on loopback) on this host, when that is possible: it needs
# `cargo` on PATH and COCIRCOM_REF pointing at a checkout of the reference (neither exists on the GPU boxes this repository is
# measured on: the image has no Rust toolchain and no network, so this script then says so and exits 0 without a number).
# Its result goes into its own column and is never substituted for the C++ restatement bench.py times (cpu_baseline).
#   usage: COCIRCOM_REF=/path/to/collaborative-circom scripts/run_reference.sh <circuit.zkey> <witness.wtns> [work dir]
set -u
ZKEY=${1:-}; WTNS=${2:-}; WORK=${3:-/tmp/cocircom_ref_run}
if ! command -v cargo >/dev/null 2>&1; then echo "run_reference: no cargo on PATH - the reference cannot be built here (skipped)"; exit 0; fi
if [ -z "${COCIRCOM_REF:-}" ] || [ ! -d "$COCIRCOM_REF/co-circom" ]; then echo "run_reference: COCIRCOM_REF does not point at a reference checkout (skipped)"; exit 0; fi
if [ ! -f "$ZKEY" ] || [ ! -f "$WTNS" ]; then echo "usage: COCIRCOM_REF=... $0 <circuit.zkey> <witness.wtns> [work dir]"; exit 2; fi
mkdir -p "$WORK"; cd "$COCIRCOM_REF/co-circom/co-circom" || exit 1
cargo build --release --bin co-circom || { echo "run_reference: build failed"; exit 1; }
BIN="$COCIRCOM_REF/target/release/co-circom"
EX="$COCIRCOM_REF/co-circom/co-circom/examples"            # shipped three-party loopback configs and keys (examples/configs, data/)
"$BIN" split-witness --witness "$WTNS" --r1cs "${R1CS:-/dev/null}" --protocol REP3 --curve BN254 --out-dir "$WORK" || { echo "run_reference: split-witness failed (set R1CS=...)"; exit 1; }
for rep in 1 2 3; do
  T0=$(date +%s.%N)
  for p in 0 1 2; do
    "$BIN" generate-proof groth16 --witness "$WORK/$(basename "$WTNS").$p.shared" --zkey "$ZKEY" --protocol REP3 --curve BN254 \
        --config "$EX/configs/party$((p + 1)).toml" --out "$WORK/proof.$p.json" --public-input "$WORK/public.$p.json" > "$WORK/party$p.log" 2>&1 &
  done
  wait
  T1=$(date +%s.%N)
  echo "run_reference: rep $rep: three parties on loopback, wall $(echo "$T1 - $T0" | bc) s (includes zkey parsing; the prove time the CLI logs is in $WORK/party*.log)"
done
