//! The fast path behind `co-circom generate-proof --protocol REP3`: ONE party of a co-groth16 proof on a proving session
//! (`cgh_session_open` keeps the zkey's tables resident, validated and precomputed on the GPU), with the party's own network and its own
//! correlated randomness handed to the library as C callbacks (`cgh_session_prove_rep3_party`, include/cogroth16_host.h).
//!
//! What stays in Rust, unchanged: `Rep3MpcNet` (every message of the proof goes through `send_bytes` / `recv_bytes`,
//! mpc-core/src/protocols/rep3/network.rs:137-176), `Rep3Protocol::new` (the PRF set-up, rep3.rs:385-398) and every random draw
//! (`Rep3Rand`, rep3/rngs.rs:25-62).  What moves: the whole of `CoGroth16::prove` (co-groth16/src/groth16.rs:113-326) between the
//! draws and the messages — witness map, the two `mul_vec` local products, the five MSMs and the O(1) point algebra — runs inside the
//! library on device-resident vectors, so the 2 x 128 MiB share vectors cross PCIe once instead of once per trait call (the per-call
//! drivers of rep3.rs / plain.rs / shamir.rs in this crate remain the drop-in for co-plonk and for mixed deployments).
//!
//! Wire format: the messages are byte strings between parties that all run this backend (field vectors as 32-byte Montgomery limbs in
//! chunks of at most 4 MiB, points as packed affine coordinates); a party running the stock CPU prover cannot be mixed into such a run.
//!
//! SOURCE ONLY (no Rust toolchain in the image this repository is built in).  The same entry point is exercised by
//! tests/test_rep3_party_abi.py: three threads, each through the callback ABI over sockets, bit-identical to the oracle's proofs
//! (`plonk_prove_rep3`, the co-plonk twin over the same tables: tests/test_plonk_rounds.py).
use crate::gpu::curve_id;
use ark_ec::{pairing::Pairing, short_weierstrass::{Affine, Projective, SWCurveConfig}, AffineRepr, CurveGroup};
use ark_ff::{PrimeField, Zero};
use bytes::Bytes;
use mpc_core::protocols::rep3::{id::PartyID, network::{Rep3MpcNet, Rep3Network}, Rep3Protocol};
use std::{ffi::{c_void, CStr, CString}, io, mem::size_of, os::raw::c_char, path::Path, ptr, slice};

// ---- include/cogroth16_host.h ------------------------------------------------------------------------------------------------------------
#[repr(C)]
pub struct cgh_rep3_net {
    pub user: *mut c_void,
    pub party_id: i32,
    pub send_next: Option<unsafe extern "C" fn(*mut c_void, *const c_void, usize) -> i32>,
    pub recv_prev: Option<unsafe extern "C" fn(*mut c_void, *mut c_void, usize) -> i32>,
    pub send_prev: Option<unsafe extern "C" fn(*mut c_void, *const c_void, usize) -> i32>,
    pub recv_next: Option<unsafe extern "C" fn(*mut c_void, *mut c_void, usize) -> i32>,
    pub recv_prev_pinned: Option<unsafe extern "C" fn(*mut c_void, usize) -> *const c_void>,
}
#[repr(C)]
pub struct cgh_rep3_rand {
    pub user: *mut c_void,
    pub masking_field_elements: Option<unsafe extern "C" fn(*mut c_void, usize, *mut u64, *mut *const u64) -> i32>,
    pub random_fes: Option<unsafe extern "C" fn(*mut c_void, *mut u64, *mut u64) -> i32>,
    pub masking_ec_element: Option<unsafe extern "C" fn(*mut c_void, i32, *mut u64) -> i32>,
}
/// Rep3Rand's two ChaCha12 generators described by seed and word position: the library then draws the masking vectors of `mul_vec` on the
/// GPU (`cg_chacha12_fr_rand_dev`) instead of asking for them element by element.
#[repr(C)]
pub struct cgh_rep3_chacha {
    pub user: *mut c_void,
    pub get_state: Option<unsafe extern "C" fn(*mut c_void, *mut u8, *mut u64, *mut u8, *mut u64) -> i32>,
    pub set_word_pos: Option<unsafe extern "C" fn(*mut c_void, u64, u64) -> i32>,
}
#[link(name = "cogroth16_host")]
extern "C" {
    fn cgh_last_error() -> *const c_char;
    fn cgh_session_prove_rep3_party_ex(session: *mut c_void, pub_in: *const u64, wit_a: *const u64, wit_b: *const u64, net: *const cgh_rep3_net,
                                       rnd: *const cgh_rep3_rand, streams: *const cgh_rep3_chacha, out_proof: *mut u64, seconds: *mut f64) -> i32;
    fn cgh_session_open_multi(devices: *const i32, n_devices: i32, curve: i32, zkey_path: *const c_char, precompute: i32, flags: u32, out: *mut *mut c_void) -> i32;
    fn cgh_session_close(session: *mut c_void) -> i32;
    fn cgh_plonk_prove_rep3_party(device: i32, curve: i32, zkey_path: *const c_char, pub_in: *const u64, wit_a: *const u64, wit_b: *const u64,
                                  blind_a: *const u64, blind_b: *const u64, net: *const cgh_rep3_net, rnd: *const cgh_rep3_rand, upto: i32,
                                  out_commits: *mut u64, out_evals: *mut u64, out_challenges: *mut u64) -> i32;
    fn cgh_plonk_prove_rep3_party_ex(device: i32, curve: i32, zkey_path: *const c_char, pub_in: *const u64, wit_a: *const u64, wit_b: *const u64,
                                     blind_a: *const u64, blind_b: *const u64, net: *const cgh_rep3_net, rnd: *const cgh_rep3_rand,
                                     streams: *const cgh_rep3_chacha, upto: i32, out_commits: *mut u64, out_evals: *mut u64, out_challenges: *mut u64) -> i32;
    fn cgh_session_prove_rep3_party(session: *mut c_void, pub_in: *const u64, wit_a: *const u64, wit_b: *const u64, net: *const cgh_rep3_net,
                                    rnd: *const cgh_rep3_rand, out_proof: *mut u64, seconds: *mut f64) -> i32;
}
fn host_error() -> String {
    unsafe {
        let p = cgh_last_error();
        if p.is_null() { String::new() } else { CStr::from_ptr(p).to_string_lossy().into_owned() }
    }
}

/// A zkey resident on one or several GPUs of this machine (one session per prover process; `ZKey` is fixed for its life, zkey.rs:48-71).
/// Opening reads the file (mapped, the point sections go to the device where they lie), runs the parser's per-point checks
/// (circom-types/src/traits.rs:107-155) on the GPU and builds the per-window tables — the work `co-circom.rs:482` does before its timer
/// starts at `:503`.
pub struct Groth16Session {
    handle: *mut c_void,
}
unsafe impl Send for Groth16Session {}

impl Groth16Session {
    /// `devices`: the party's GPUs (devices[0] runs the witness map; every device holds a slice of the five queries).
    /// `validate = false` skips the point checks for a file that was validated before.
    pub fn open<F: PrimeField>(devices: &[i32], zkey: &Path, validate: bool) -> eyre::Result<Self> {
        Self::open_with::<F>(devices, zkey, validate, false)
    }

    /// `additive_h = true` (`CGH_SESSION_ADDITIVE_H`): the opt-in REP3 / Shamir variant in which the witness map's two products are not
    /// re-shared (no 2 x 32 B x m exchange), every MSM multiplies the party's own share component and the five results are re-shared as
    /// points in one round.  NOT the message sequence of `CoGroth16::prove`; the proof is the same.  All parties must agree on it.
    pub fn open_with<F: PrimeField>(devices: &[i32], zkey: &Path, validate: bool, additive_h: bool) -> eyre::Result<Self> {
        let path = CString::new(zkey.to_string_lossy().as_bytes())?;
        let mut handle = ptr::null_mut();
        let flags = (if validate { 0 } else { 1 }) | (if additive_h { 2 } else { 0 });
        let rc = unsafe { cgh_session_open_multi(devices.as_ptr(), devices.len() as i32, curve_id::<F>(), path.as_ptr(), -1, flags, &mut handle) };
        if rc != 0 {
            eyre::bail!("cgh_session_open_multi: {}", host_error());
        }
        Ok(Self { handle })
    }

    /// `CoGroth16::<Rep3Protocol<_, Rep3MpcNet>, P>::prove` (groth16.rs:113-139) for this party.  `public_inputs` includes the leading one
    /// (`SharedWitness::public_inputs`, co-circom-snarks/src/lib.rs:32-34); `(wit_a, wit_b)` = `SharedWitness::witness.get_ab()`
    /// (rep3/fieldshare.rs:245-247).  Returns (pi_a, pi_b, pi_c) — the fields of `Groth16Proof` (circom-types/src/groth16/proof.rs:8-29).
    pub fn prove_rep3<P: Pairing>(
        &self,
        protocol: &mut Rep3Protocol<P::ScalarField, Rep3MpcNet>,
        public_inputs: &[P::ScalarField],
        wit_a: &[P::ScalarField],
        wit_b: &[P::ScalarField],
    ) -> io::Result<(P::G1Affine, P::G2Affine, P::G1Affine)>
    where
        P::G1: PackedAffine,
        P::G2: PackedAffine,
    {
        assert_eq!(wit_a.len(), wit_b.len());
        let span = tracing::trace_span!("cogroth16_hip::prove_rep3", n = wit_a.len());
        let _enter = span.enter();
        let mut state = Callbacks::<P> { protocol, error: None };
        let id: usize = state.protocol.network_mut().get_id().into();
        let net = cgh_rep3_net {
            user: &mut state as *mut _ as *mut c_void,
            party_id: id as i32,
            send_next: Some(send_next::<P>),
            recv_prev: Some(recv_prev::<P>),
            send_prev: Some(send_prev::<P>),
            recv_next: Some(recv_next::<P>),
            recv_prev_pinned: None, // BytesMut frames live in pageable memory
        };
        let rnd = cgh_rep3_rand {
            user: &mut state as *mut _ as *mut c_void,
            masking_field_elements: Some(masking_field_elements::<P>),
            random_fes: Some(random_fes::<P>),
            masking_ec_element: Some(masking_ec_element::<P>),
        };
        // the O(n) draws (2 x m per mul_vec, rngs.rs:37-46) happen on the GPU from the generators' seeds and positions; the O(1) draws
        // (random_fes, masking_ec_element) stay with `rnd` above and continue behind them
        let streams = cgh_rep3_chacha {
            user: &mut state as *mut _ as *mut c_void,
            get_state: Some(chacha_get_state::<P>),
            set_word_pos: Some(chacha_set_word_pos::<P>),
        };
        let fq = size_of::<<P::G1 as CurveGroup>::BaseField>() / 8;
        let mut proof = vec![0u64; 8 * fq];
        let rc = unsafe {
            cgh_session_prove_rep3_party_ex(self.handle, public_inputs.as_ptr() as *const u64, wit_a.as_ptr() as *const u64, wit_b.as_ptr() as *const u64,
                                            &net, &rnd, &streams, proof.as_mut_ptr(), ptr::null_mut())
        };
        if rc != 0 {
            // the io::Error a callback met (rep3.rs:661-669 would have returned it with `?`), else the library's message
            return Err(state.error.take().unwrap_or_else(|| io::Error::new(io::ErrorKind::Other, host_error())));
        }
        let (a, rest) = proof.split_at(2 * fq);
        let (b, c) = rest.split_at(4 * fq);
        Ok((<P::G1 as PackedAffine>::from_packed(a), <P::G2 as PackedAffine>::from_packed(b), <P::G1 as PackedAffine>::from_packed(c)))
    }
}
impl Drop for Groth16Session {
    fn drop(&mut self) {
        unsafe { cgh_session_close(self.handle) };
    }
}

/// `CoPlonk::<Rep3Protocol<_, Rep3MpcNet>, P>::prove` (co-plonk/src/plonk.rs:133-271) for this party on GPU `device`, through the same callback
/// tables as `Groth16Session::prove_rep3` (`cgh_plonk_prove_rep3_party`; the CLI's plonk branch is co-circom.rs:560-600).  The eleven blinding
/// shares are drawn with `protocol.rand()` inside the library, first, as `Round1Challenges::random` does (round1.rs:93-99).
/// Returns the nine commitments (a, b, c, z, t1, t2, t3, wxi, wxiw) and six evaluations (a, b, c, s1, s2, zw) of `PlonkProof`
/// (circom-types/src/plonk/proof.rs).
pub fn plonk_prove_rep3<P: Pairing>(
    device: i32,
    zkey: &Path,
    protocol: &mut Rep3Protocol<P::ScalarField, Rep3MpcNet>,
    public_inputs: &[P::ScalarField],
    wit_a: &[P::ScalarField],
    wit_b: &[P::ScalarField],
) -> io::Result<(Vec<P::G1Affine>, Vec<P::ScalarField>)>
where
    P::G1: PackedAffine,
{
    assert_eq!(wit_a.len(), wit_b.len());
    let span = tracing::trace_span!("cogroth16_hip::plonk_prove_rep3", n = wit_a.len());
    let _enter = span.enter();
    let path = CString::new(zkey.to_string_lossy().as_bytes()).map_err(|e| io::Error::new(io::ErrorKind::InvalidInput, e))?;
    let mut state = Callbacks::<P> { protocol, error: None };
    let id: usize = state.protocol.network_mut().get_id().into();
    let net = cgh_rep3_net {
        user: &mut state as *mut _ as *mut c_void,
        party_id: id as i32,
        send_next: Some(send_next::<P>),
        recv_prev: Some(recv_prev::<P>),
        send_prev: Some(send_prev::<P>),
        recv_next: Some(recv_next::<P>),
        recv_prev_pinned: None,
    };
    let rnd = cgh_rep3_rand {
        user: &mut state as *mut _ as *mut c_void,
        masking_field_elements: Some(masking_field_elements::<P>),
        random_fes: Some(random_fes::<P>),
        masking_ec_element: Some(masking_ec_element::<P>),
    };
    let fq = size_of::<<P::G1 as CurveGroup>::BaseField>() / 8;
    let mut commits = vec![0u64; 9 * 2 * fq];
    let mut evals = vec![P::ScalarField::zero(); 6];
    // the masking vectors of the rounds' mul_vec calls are drawn on the GPU from the generators' seeds and positions (see prove_rep3)
    let streams = cgh_rep3_chacha {
        user: &mut state as *mut _ as *mut c_void,
        get_state: Some(chacha_get_state::<P>),
        set_word_pos: Some(chacha_set_word_pos::<P>),
    };
    let rc = unsafe {
        cgh_plonk_prove_rep3_party_ex(device, curve_id::<P::ScalarField>(), path.as_ptr(), public_inputs.as_ptr() as *const u64, wit_a.as_ptr() as *const u64,
                                      wit_b.as_ptr() as *const u64, ptr::null(), ptr::null(), &net, &rnd, &streams, 5, commits.as_mut_ptr(),
                                      evals.as_mut_ptr() as *mut u64, ptr::null_mut())
    };
    if rc != 0 {
        return Err(state.error.take().unwrap_or_else(|| io::Error::new(io::ErrorKind::Other, host_error())));
    }
    Ok((commits.chunks(2 * fq).map(<P::G1 as PackedAffine>::from_packed).collect(), evals))
}

/// packed affine `x || y` in Montgomery limbs, all zero = the point at infinity (the zkey's own encoding, circom-types/src/traits.rs:107-155)
pub trait PackedAffine: CurveGroup {
    fn from_packed(limbs: &[u64]) -> Self::Affine;
}
impl<Q: SWCurveConfig> PackedAffine for Projective<Q> {
    fn from_packed(limbs: &[u64]) -> Affine<Q> {
        if limbs.iter().all(|&w| w == 0) {
            return Affine::<Q>::zero();
        }
        let n = size_of::<Q::BaseField>();
        // Fp<MontBackend, N> / QuadExtField are plain limb arrays in memory (layout asserted in gpu.rs::Layout::CHECK)
        unsafe {
            let x = ptr::read_unaligned(limbs.as_ptr() as *const Q::BaseField);
            let y = ptr::read_unaligned((limbs.as_ptr() as *const u8).add(n) as *const Q::BaseField);
            Affine::<Q>::new_unchecked(x, y)
        }
    }
}

// ---- the callbacks: closures over the stock protocol object ------------------------------------------------------------------------------------
struct Callbacks<'a, P: Pairing> {
    protocol: &'a mut Rep3Protocol<P::ScalarField, Rep3MpcNet>,
    error: Option<io::Error>,
}
fn fail<P: Pairing>(s: &mut Callbacks<P>, e: io::Error) -> i32 {
    let code = e.raw_os_error().unwrap_or(5);
    s.error.get_or_insert(e);
    if code == 0 { 5 } else { code }
}
unsafe fn send<P: Pairing>(user: *mut c_void, to_next: bool, data: *const c_void, bytes: usize) -> i32 {
    let s = &mut *(user as *mut Callbacks<P>);
    let net = s.protocol.network_mut();
    let id = net.get_id();
    let target: PartyID = if to_next { id.next_id() } else { id.prev_id() };
    tracing::trace!("> send {bytes} bytes to {target:?}");
    match net.send_bytes(target, Bytes::copy_from_slice(slice::from_raw_parts(data as *const u8, bytes))) {
        Ok(()) => 0,
        Err(e) => fail(s, e),
    }
}
unsafe fn recv<P: Pairing>(user: *mut c_void, from_prev: bool, data: *mut c_void, bytes: usize) -> i32 {
    let s = &mut *(user as *mut Callbacks<P>);
    let net = s.protocol.network_mut();
    let id = net.get_id();
    let from: PartyID = if from_prev { id.prev_id() } else { id.next_id() };
    match net.recv_bytes(from) {
        Ok(frame) if frame.len() == bytes => {
            ptr::copy_nonoverlapping(frame.as_ptr(), data as *mut u8, bytes);
            tracing::trace!("< received {bytes} bytes from {from:?}");
            0
        }
        Ok(_) => fail(s, io::Error::new(io::ErrorKind::InvalidData, "During execution of mul_vec in MPC: Invalid number of elements received")), // rep3.rs:663-668
        Err(e) => fail(s, e),
    }
}
unsafe extern "C" fn send_next<P: Pairing>(u: *mut c_void, d: *const c_void, n: usize) -> i32 { send::<P>(u, true, d, n) }
unsafe extern "C" fn send_prev<P: Pairing>(u: *mut c_void, d: *const c_void, n: usize) -> i32 { send::<P>(u, false, d, n) }
unsafe extern "C" fn recv_prev<P: Pairing>(u: *mut c_void, d: *mut c_void, n: usize) -> i32 { recv::<P>(u, true, d, n) }
unsafe extern "C" fn recv_next<P: Pairing>(u: *mut c_void, d: *mut c_void, n: usize) -> i32 { recv::<P>(u, false, d, n) }

/// `n` draws of `Rep3Rand::masking_field_element` (rngs.rs:37-40) written into the library's page-locked buffer: the upload to the GPU
/// starts from there without another copy.  The draws are the reference's, in the reference's order (both `mul_vec` calls of
/// groth16.rs:174,190 precede every other draw of the proof), so the peers' `rng2` streams stay in step.
unsafe extern "C" fn masking_field_elements<P: Pairing>(u: *mut c_void, n: usize, buf: *mut u64, out: *mut *const u64) -> i32 {
    let s = &mut *(u as *mut Callbacks<P>);
    let dst = slice::from_raw_parts_mut(buf as *mut P::ScalarField, n);
    for d in dst.iter_mut() {
        *d = s.protocol.masking_field_element();
    }
    *out = buf;
    0
}
/// `ChaCha12Rng::get_seed` / `get_word_pos` of Rep3Rand's rng1 and rng2 (accessor added by mpc-core-accessors.patch).  A position beyond
/// 2^64 words (2^66 bytes drawn from one generator) is refused rather than truncated.
unsafe extern "C" fn chacha_get_state<P: Pairing>(u: *mut c_void, seed1: *mut u8, pos1: *mut u64, seed2: *mut u8, pos2: *mut u64) -> i32 {
    let s = &mut *(u as *mut Callbacks<P>);
    let (s1, p1, s2, p2) = s.protocol.rand_stream_state();
    if p1 > u64::MAX as u128 >> 1 || p2 > u64::MAX as u128 >> 1 {
        return fail(s, io::Error::new(io::ErrorKind::Other, "ChaCha12 word position out of the backend's range"));
    }
    ptr::copy_nonoverlapping(s1.as_ptr(), seed1, 32);
    ptr::copy_nonoverlapping(s2.as_ptr(), seed2, 32);
    *pos1 = p1 as u64;
    *pos2 = p2 as u64;
    0
}
/// `ChaCha12Rng::set_word_pos` on both generators: the next draw made in Rust is the one behind the GPU's last.
unsafe extern "C" fn chacha_set_word_pos<P: Pairing>(u: *mut c_void, pos1: u64, pos2: u64) -> i32 {
    let s = &mut *(u as *mut Callbacks<P>);
    s.protocol.set_rand_stream_positions(pos1 as u128, pos2 as u128);
    tracing::trace!("rng1 / rng2 moved to words {pos1} / {pos2} behind the device's draws");
    0
}
unsafe extern "C" fn random_fes<P: Pairing>(u: *mut c_void, a: *mut u64, b: *mut u64) -> i32 {
    let s = &mut *(u as *mut Callbacks<P>);
    let (x, y) = s.protocol.random_fes();
    ptr::write_unaligned(a as *mut P::ScalarField, x);
    ptr::write_unaligned(b as *mut P::ScalarField, y);
    0
}
unsafe extern "C" fn masking_ec_element<P: Pairing>(u: *mut c_void, group: i32, out: *mut u64) -> i32 {
    let s = &mut *(u as *mut Callbacks<P>);
    // Projective{x, y, z} is the ABI's Jacobian layout (gpu.rs::Layout::CHECK)
    if group == crate::ffi::CG_G1 {
        ptr::write_unaligned(out as *mut P::G1, s.protocol.masking_ec_element::<P::G1>());
    } else {
        ptr::write_unaligned(out as *mut P::G2, s.protocol.masking_ec_element::<P::G2>());
    }
    0
}

// ---- the Shamir twin (co-circom.rs:507-527) ----------------------------------------------------------------------------------------------------
// ONE of n Shamir parties: the any-to-any network stays the caller's `ShamirMpcNet` (send_bytes / recv_bytes, shamir/network.rs:134-160) and
// the party's PRIVATE randomness its own `rand::Rng` — `ShamirProtocol` keeps both in private fields (`network`, `rng_buffer.rng`,
// shamir.rs:197-205, 873-880), so this entry takes them directly instead of through the protocol object: construct the net as the CLI
// does (`ShamirMpcNet::new(config.network)`), seed an `RngType` as `ShamirProtocol::new` does (`RngType::from_entropy()`), and call this
// instead of `ShamirProtocol::new(t, net)` + `CoGroth16::prove`.
#[repr(C)]
pub struct cgh_shamir_net {
    pub user: *mut c_void,
    pub party_id: i32,
    pub num_parties: i32,
    pub send: Option<unsafe extern "C" fn(*mut c_void, i32, *const c_void, usize) -> i32>,
    pub recv: Option<unsafe extern "C" fn(*mut c_void, i32, *mut c_void, usize) -> i32>,
}
#[repr(C)]
pub struct cgh_shamir_rand {
    pub user: *mut c_void,
    pub random_field_elements: Option<unsafe extern "C" fn(*mut c_void, usize, *mut u64) -> i32>,
}
extern "C" {
    fn cgh_session_prove_shamir_party(session: *mut c_void, threshold: i32, pub_in: *const u64, wit: *const u64, net: *const cgh_shamir_net,
                                      rnd: *const cgh_shamir_rand, preprocess: usize, out_proof: *mut u64, seconds: *mut f64) -> i32;
    fn cgh_session_prove_shamir_party_seeded(session: *mut c_void, threshold: i32, pub_in: *const u64, wit: *const u64, net: *const cgh_shamir_net,
                                             seed32: *const u8, preprocess: usize, out_proof: *mut u64, seconds: *mut f64) -> i32;
}
struct ShamirCallbacks<'a, F: PrimeField, R: rand::Rng> {
    net: &'a mut mpc_core::protocols::shamir::network::ShamirMpcNet,
    rng: &'a mut R,
    error: Option<io::Error>,
    _f: std::marker::PhantomData<F>,
}
unsafe extern "C" fn sh_send<F: PrimeField, R: rand::Rng>(u: *mut c_void, to: i32, d: *const c_void, n: usize) -> i32 {
    let s = &mut *(u as *mut ShamirCallbacks<F, R>);
    match s.net.send_bytes(to as usize, Bytes::copy_from_slice(slice::from_raw_parts(d as *const u8, n))) {
        Ok(()) => 0,
        Err(e) => { let c = e.raw_os_error().unwrap_or(5); s.error.get_or_insert(e); if c == 0 { 5 } else { c } }
    }
}
unsafe extern "C" fn sh_recv<F: PrimeField, R: rand::Rng>(u: *mut c_void, from: i32, d: *mut c_void, n: usize) -> i32 {
    let s = &mut *(u as *mut ShamirCallbacks<F, R>);
    match s.net.recv_bytes(from as usize) {
        Ok(frame) if frame.len() == n => { ptr::copy_nonoverlapping(frame.as_ptr(), d as *mut u8, n); 0 }
        Ok(_) => { s.error.get_or_insert(io::Error::new(io::ErrorKind::InvalidData, "During execution of degree_reduce_vec in MPC: Invalid number of elements received")); 74 } // shamir.rs:324-329
        Err(e) => { let c = e.raw_os_error().unwrap_or(5); s.error.get_or_insert(e); if c == 0 { 5 } else { c } }
    }
}
unsafe extern "C" fn sh_rand<F: PrimeField, R: rand::Rng>(u: *mut c_void, n: usize, out: *mut u64) -> i32 {
    let s = &mut *(u as *mut ShamirCallbacks<F, R>);
    for d in slice::from_raw_parts_mut(out as *mut F, n).iter_mut() {
        *d = F::rand(s.rng); // the draws of ShamirCore::share / buffer_triples, in the reference's order
    }
    0
}
impl Groth16Session {
    /// `CoGroth16::<ShamirProtocol<_, ShamirMpcNet>, P>::prove` for this party.  `witness` = `SharedWitness::witness` as plain field elements
    /// (`ShamirPrimeFieldShareVec`, shamir/fieldshare.rs:152-155); `preprocess` > 0 generates that many double sharings on the GPU up front.
    pub fn prove_shamir<P: Pairing, R: rand::Rng>(
        &self,
        net: &mut mpc_core::protocols::shamir::network::ShamirMpcNet,
        rng: &mut R,
        threshold: usize,
        public_inputs: &[P::ScalarField],
        witness: &[P::ScalarField],
        preprocess: usize,
    ) -> io::Result<(P::G1Affine, P::G2Affine, P::G1Affine)>
    where
        P::G1: PackedAffine,
        P::G2: PackedAffine,
    {
        use mpc_core::protocols::shamir::network::ShamirNetwork;
        let (id, n) = (net.get_id(), net.get_num_parties());
        let mut state = ShamirCallbacks::<P::ScalarField, R> { net, rng, error: None, _f: std::marker::PhantomData };
        let user = &mut state as *mut _ as *mut c_void;
        let net_cb = cgh_shamir_net { user, party_id: id as i32, num_parties: n as i32, send: Some(sh_send::<P::ScalarField, R>), recv: Some(sh_recv::<P::ScalarField, R>) };
        let rnd_cb = cgh_shamir_rand { user, random_field_elements: Some(sh_rand::<P::ScalarField, R>) };
        let fq = size_of::<<P::G1 as CurveGroup>::BaseField>() / 8;
        let mut proof = vec![0u64; 8 * fq];
        // ShamirProtocol's generator is private (`RngType::from_entropy()`, shamir.rs:211-246): with `preprocess` > 0 the party hands the library ONE
        // 32-byte seed drawn from its rng and the amount * (1 + 3t) draws of the batch are made on the GPU from a ChaCha12 stream of that seed
        // (the same generator type the reference seeds), instead of `sh_rand` filling them one F::rand at a time on this thread.
        let rc = if preprocess > 0 {
            let seed: [u8; 32] = state.rng.gen();
            unsafe {
                cgh_session_prove_shamir_party_seeded(self.handle, threshold as i32, public_inputs.as_ptr() as *const u64, witness.as_ptr() as *const u64, &net_cb,
                                                      seed.as_ptr(), preprocess, proof.as_mut_ptr(), ptr::null_mut())
            }
        } else {
            unsafe {
                cgh_session_prove_shamir_party(self.handle, threshold as i32, public_inputs.as_ptr() as *const u64, witness.as_ptr() as *const u64, &net_cb, &rnd_cb,
                                               preprocess, proof.as_mut_ptr(), ptr::null_mut())
            }
        };
        if rc != 0 {
            return Err(state.error.take().unwrap_or_else(|| io::Error::new(io::ErrorKind::Other, host_error())));
        }
        let (a, rest) = proof.split_at(2 * fq);
        let (b, c) = rest.split_at(4 * fq);
        Ok((<P::G1 as PackedAffine>::from_packed(a), <P::G2 as PackedAffine>::from_packed(b), <P::G1 as PackedAffine>::from_packed(c)))
    }
}
