//! Values the MI355X backend restates from third-party crates and that no test of the reference holds:
//!  * `F::rand(&mut ChaCha12Rng)` (ark-ff 0.4.2 `Fp::rand` over rand_chacha 0.3; mpc-core/src/lib.rs:10 `RngType = ChaCha12Rng`,
//!    mpc-core/src/protocols/rep3/rngs.rs:37-46) — word stream -> limbs -> rejection order, at stream positions 0, unaligned and
//!    beyond 2^32 blocks (the 64-bit block counter's carry);
//!  * `Rep3Rand::masking_field_element` = `F::rand(rng1) - F::rand(rng2)` (rngs.rs:37-40; Rep3Rand is pub(crate), so its two lines are
//!    repeated here over the same generator type);
//!  * the `.shared` witness container (co-circom/co-circom-snarks/src/lib.rs:24-41 through bincode, co-circom.rs:215,244).
//! usage: cargo run --release -- <repo>/tests/golden     (writes rust_pins.json and shared/rust_*.shared there)
use ark_ec::pairing::Pairing;
use ark_ff::{PrimeField, UniformRand};
use circom_types::Witness;
use co_circom_snarks::SharedWitness;
use mpc_core::protocols::rep3::{network::Rep3MpcNet, Rep3Protocol};
use mpc_core::protocols::shamir::{network::ShamirMpcNet, ShamirProtocol};
use rand::SeedableRng;
use rand_chacha::ChaCha12Rng;
use serde_json::{json, Value};
use std::{env, fs, path::PathBuf};

/// Montgomery limbs as the backend's ABI carries them (`Fp.0 .0`: 4 x u64 little-endian, fully reduced) and the canonical integer
fn fe<F: PrimeField>(x: &F) -> Value {
    // x * R mod p with R = 2^(64 N) is, as an integer, what `Fp.0` holds in ark-ff 0.4 (MontBackend) — computed through the public API so that
    // nothing here depends on the struct's private layout
    let n = (F::MODULUS_BIT_SIZE as usize + 63) / 64;
    let r = F::from(2u64).pow([(64 * n) as u64]);
    let m = (*x * r).into_bigint();
    json!({
        "montgomery_limbs_le": m.as_ref().iter().map(|l| format!("{l:#018x}")).collect::<Vec<_>>(),
        "canonical_decimal": x.into_bigint().to_string(),
    })
}

fn seed(tag: u8) -> [u8; 32] {
    core::array::from_fn(|i| tag.wrapping_mul(37).wrapping_add((i as u8).wrapping_mul(11)).wrapping_add(5))
}

fn draws<F: PrimeField>(s: [u8; 32], word_pos: u128, n: usize) -> Value {
    let mut rng = ChaCha12Rng::from_seed(s);
    rng.set_word_pos(word_pos);
    let vals: Vec<Value> = (0..n).map(|_| fe(&F::rand(&mut rng))).collect();
    json!({"seed_hex": s.iter().map(|b| format!("{b:02x}")).collect::<String>(), "word_pos": word_pos.to_string(),
           "word_pos_after": rng.get_word_pos().to_string(), "draws": vals})
}

fn masks<F: PrimeField>(s1: [u8; 32], s2: [u8; 32], n: usize) -> Value {
    let (mut rng1, mut rng2) = (ChaCha12Rng::from_seed(s1), ChaCha12Rng::from_seed(s2));
    let vals: Vec<Value> = (0..n).map(|_| { let a = F::rand(&mut rng1); let b = F::rand(&mut rng2); fe(&(a - b)) }).collect();   // rngs.rs:37-46
    json!({"seed1_hex": s1.iter().map(|b| format!("{b:02x}")).collect::<String>(), "seed2_hex": s2.iter().map(|b| format!("{b:02x}")).collect::<String>(),
           "masking_field_elements": vals, "word_pos1_after": rng1.get_word_pos().to_string(), "word_pos2_after": rng2.get_word_pos().to_string()})
}

fn curve_pins<F: PrimeField>() -> Value {
    json!({
        "fr_rand": [
            draws::<F>(seed(1), 0, 8),                              // from the start of the stream
            draws::<F>(seed(2), 5, 8),                              // a position that is not a multiple of the 16-word block
            draws::<F>(seed(3), (1u128 << 36) + 3, 8),              // block index above 2^32: the counter's upper word
            draws::<F>(seed(4), 0, 4096),                           // long enough to contain rejections in sequence (BN254 rejects 24.4 % of the candidates)
        ],
        "rep3_masks": masks::<F>(seed(5), seed(6), 8),
    })
}

/// the multiplier2 witness of the reference's test vectors ([1, 33, 3, 11], circom-types/src/witness.rs:101-134), shared with seeded generators
fn shared_files<P: Pairing>(out: &PathBuf, curve: &str, wtns: &PathBuf) -> Value
where
    P::ScalarField: circom_types::traits::CircomArkworksPrimeFieldBridge,
{
    let witness = Witness::<P::ScalarField>::from_reader(fs::File::open(wtns).expect("witness.wtns of the multiplier2 fixture")).unwrap();
    let mut rng = ChaCha12Rng::from_seed(seed(7));
    let shares = SharedWitness::<Rep3Protocol<P::ScalarField, Rep3MpcNet>, P>::share_rep3(witness.clone(), 2, &mut rng);
    let mut names = vec![];
    for (i, s) in shares.iter().enumerate() {
        let p = out.join("shared").join(format!("rust_{curve}_multiplier2_rep3_{i}.shared"));
        bincode::serialize_into(fs::File::create(&p).unwrap(), s).unwrap();                     // co-circom.rs:215
        names.push(p.file_name().unwrap().to_string_lossy().into_owned());
    }
    let mut rng = ChaCha12Rng::from_seed(seed(8));
    let shares = SharedWitness::<ShamirProtocol<P::ScalarField, ShamirMpcNet>, P>::share_shamir(witness, 2, 1, 3, &mut rng);
    for (i, s) in shares.iter().enumerate() {
        let p = out.join("shared").join(format!("rust_{curve}_multiplier2_shamir_{i}.shared"));
        bincode::serialize_into(fs::File::create(&p).unwrap(), s).unwrap();                     // co-circom.rs:244
        names.push(p.file_name().unwrap().to_string_lossy().into_owned());
    }
    json!({"files": names, "witness_canonical_decimal": ["1", "33", "3", "11"], "num_pub_inputs": 2,
           "note": "REP3: the three parties' (a, b) vectors must satisfy a_i = b_(i+1) and sum to the witness; Shamir (t = 1, n = 3): shares at x = 1, 2, 3 interpolate to it"})
}

fn main() {
    let out = PathBuf::from(env::args().nth(1).expect("usage: pin-vectors <repo>/tests/golden"));
    fs::create_dir_all(out.join("shared")).unwrap();
    let fx = |curve: &str| out.join("groth16").join(curve).join("multiplier2").join("witness.wtns");
    let pins = json!({
        "generator": "rust/pin-vectors (ark-ff 0.4.2, rand_chacha 0.3, reference tag 2024_10_08)",
        "bn254": curve_pins::<ark_bn254::Fr>(),
        "bls12_381": curve_pins::<ark_bls12_381::Fr>(),
        "shared": {
            "bn254": shared_files::<ark_bn254::Bn254>(&out, "bn254", &fx("bn254")),
            "bls12_381": shared_files::<ark_bls12_381::Bls12_381>(&out, "bls12_381", &fx("bls12_381")),
        },
    });
    fs::write(out.join("rust_pins.json"), serde_json::to_string_pretty(&pins).unwrap()).unwrap();
    println!("wrote {}", out.join("rust_pins.json").display());
}
