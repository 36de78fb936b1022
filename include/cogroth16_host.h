/* cogroth16_host.h — C ABI of the host mirror (libcogroth16_host.so): the co-circom prover entry points on top of cogroth16_hip.h.
 *
 * The reference is Rust and cannot be built in this image (no cargo), so the host side above the kernel ABI is C++
 * (collaborative-circom_amd/host/ (headers per layer: formats, network, driver, groth16, plonk, codecs, synth; entry points in capi_*.cpp)): drivers PlainHipDriver / Rep3HipProtocol / ShamirHipProtocol with the method names of
 * the reference's traits (mpc-core/src/traits.rs:43-223, 535-568), CoGroth16::prove (co-groth16/src/groth16.rs:113-326) and CoPlonk
 * rounds 1-5 (co-plonk/src/round{1..5}.rs) with the reference's call sequence, in-process REP3 / Shamir networks in the role of
 * tests/src/rep3_network.rs, and the readers / writers of the file formats (circom-types).  These entry points are what the CLI's
 * proving commands do once their arguments are parsed — co-circom/co-circom/src/bin/co-circom.rs:455-543 (generate-proof: read
 * zkey :482, read shares :487-495, construct the driver :497-499, prove :503-506, write proof :512-532) — with buffers instead of
 * files where the CLI reads them into memory first.  INTEGRATION.md shows the Rust-side binding.
 *
 * Environment variables (complete list for libcogroth16_host.so; process-wide; none is needed, none changes a proof):
 *   CGH_SKIP_ZKEY_VALIDATION      sessions / one-shot proves skip the on-curve + subgroup validation of the zkey points (= cgh_set_zkey_validation(0))
 *   CGH_TIMING                    wall-clock marks of the host-side protocol steps on stderr (microseconds since the previous mark)
 * Everything else that used to be an environment variable is an OPTION (cgh_set_option below: thresholds and layout switches) or exists only in
 * the planning build (`make -C collaborative-circom_amd/host KNOBS=1` -> libcogroth16_host_knobs.so, -DCG_DEBUG_KNOBS): the A/B knobs of the
 * measurement scripts (CGH_NO_CHAIN_PRIORITY, CGH_CHAIN_FLAG, CGH_BULK_FLAG, CGH_BULK_CHUNK, CGH_PLAIN_CHUNK, CGH_G2_ORDER, CGH_G2_AFTER,
 * CGH_LATE_AUX) and CGH_EMULATE_DEVICE, which makes proofs WRONG on purpose (one device's share of a multi-device proof timed on one GPU).
 * The release library does not contain those names (tests/test_abi_surface.py).
 *
 * Conventions: every function returns 0 on success; on failure a non-zero value, message from cgh_last_error() (thread local).
 * Field elements are 4 x u64 (6 for the BLS12-381 base field) little-endian Montgomery limbs, exactly arkworks' in-memory form
 * (circom-types/src/traits.rs:57-67).  Proofs are packed affine points A (G1) || B (G2) || C (G1), (0, 0) = infinity
 * (groth16/proof.rs:8-29).  curve: CG_BN254 / CG_BLS12_381 of cogroth16_hip.h.  No function here has a CPU fallback.
 */
#ifndef COGROTH16_HOST_H
#define COGROTH16_HOST_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* cgh_last_error(void);

/* ---- zkey / witness files (circom-types/src/groth16/zkey.rs:139-316, witness.rs:51-91, plonk/zkey.rs:83-255) -------------------- */
/* info[7]: n_vars, n_public, domain_size, log2(domain_size), num_constraints, nnz(A), nnz(B) */
int32_t cgh_zkey_info(int32_t curve, const char* path, size_t* info);
/* the parser's per-point checks (traits.rs:107-155: on the curve, in the prime-order subgroup) for every point of the file, on the GPU;
 * seconds[2] (optional): host read + decode, upload + validation.  0 = every point valid. */
int32_t cgh_zkey_validate(int32_t device, int32_t curve, const char* path, double* seconds);
/* The prove entry points and cgh_session_open run that validation themselves, as the reference's parser does.  Callers that validated
 * the file before can switch it off process-wide (or set the environment variable CGH_SKIP_ZKEY_VALIDATION). */
int32_t cgh_set_zkey_validation(int32_t on);
/* Process-wide options (thresholds and layout switches; none changes a proof; read when a driver / session is made unless noted):
 *   option                           value                                                                                   default
 *   CGH_OPT_XCHG_ASYNC_MIN           elements from which a mul_vec exchange streams in chunks over the copy streams         2^17
 *   CGH_OPT_DEVICE_MASKS_MIN         elements from which described ChaCha12 generators are drawn on the device              2^11
 *   CGH_OPT_XCHG_COPY_STREAM_MIN     elements from which a single-message exchange of a two-context party crosses PCIe on   2^14
 *                                    the copy streams (behind the product, beside the transforms)
 *   CGH_OPT_SECOND_CONTEXT_MIN_LOG   log2 of the variables from which a session's proofs use a chain + a bulk context       15
 *   CGH_OPT_DISTRIBUTED_MAP          multi-device sessions spread the witness map over the devices (read per proof);        1
 *                                    0 = the whole map on the primary device (round-2 layout)
 *   CGH_OPT_ONE_CONTEXT              1 = one context per proof (no second context for the witness-independent MSMs)         0
 *   CGH_OPT_SPLIT_FIRST_MSM_MIN      private-witness elements from which a page-locked witness goes up with component a in  0
 *                                    two pieces and the first aux MSM starts on the first piece (HipDriver::msm_begin_aux_split); 0 = never
 *                                    (the default: measured slower on MI355X at 2^20 .. 2^24 — two more schedules and bucket sets cost
 *                                    more than the ~1.6 ms of idle chip they fill)
 *   CGH_OPT_CTX_WIDE_LOG             CG_OPT_MSM_WIDE_SMALL of the contexts a session makes from now on (0 = the library's default)   0
 *   CGH_OPT_CTX_OFF_MAIN_LOG         CG_OPT_MSM_OFF_MAIN_LOG of those contexts (0 = the library's default)                           0
 *   CGH_OPT_CTX_SOLO_LOG             CG_OPT_MSM_SOLO_LOG of those contexts (0 = the library's default)                               0 */
enum { CGH_OPT_XCHG_ASYNC_MIN = 1, CGH_OPT_DEVICE_MASKS_MIN = 2, CGH_OPT_XCHG_COPY_STREAM_MIN = 3, CGH_OPT_SECOND_CONTEXT_MIN_LOG = 4,
       CGH_OPT_DISTRIBUTED_MAP = 5, CGH_OPT_ONE_CONTEXT = 6, CGH_OPT_SPLIT_FIRST_MSM_MIN = 7, CGH_OPT_CTX_WIDE_LOG = 8, CGH_OPT_CTX_OFF_MAIN_LOG = 9,
       CGH_OPT_CTX_SOLO_LOG = 10, CGH_OPT_COUNT = 11 };
int32_t cgh_set_option(int32_t option, int64_t value);
int32_t cgh_get_option(int32_t option, int64_t* value);
/* witness.rs:51-91: n values in Montgomery form; out == NULL: only *n */
int32_t cgh_read_wtns(int32_t curve, const char* path, uint64_t* out, size_t cap, size_t* n);
/* info[6]: n_vars, n_public, domain_size, power, n_additions, n_constraints */
int32_t cgh_plonk_zkey_info(int32_t curve, const char* path, size_t* info);

/* ---- proof / public-input JSON (groth16/proof.rs:8-29, traits.rs:186-233; plonk/proof.rs) ------------------------------------------ */
int32_t cgh_proof_to_json(int32_t curve, const uint64_t* proof, char* out, size_t cap);
int32_t cgh_proof_from_json(int32_t curve, const char* json, uint64_t* out_proof);
int32_t cgh_public_to_json(int32_t curve, const uint64_t* pub, size_t n, char* out, size_t cap);
int32_t cgh_plonk_proof_to_json(int32_t curve, const uint64_t* commits, const uint64_t* evals, char* out, size_t cap);
int32_t cgh_plonk_proof_from_json(int32_t curve, const char* json, uint64_t* out_commits, uint64_t* out_evals);

/* ---- secret-shared witness container (co-circom-snarks/src/lib.rs:24-41; rep3/fieldshare.rs:233-236).  PARITY UNPINNED: the layout is
 * restated from the reference's types (bincode u64 length || ark-serialize compressed vectors); the snapshot ships no `.shared`
 * file to check it against.  protocol: 0 = REP3 (two vectors a, b), 1 = Shamir (one vector). ------------------------------------- */
int32_t cgh_shared_witness_write(int32_t curve, const char* path, int32_t protocol, const uint64_t* pub, size_t n_pub, const uint64_t* a, const uint64_t* b, size_t n);
/* sizes[0] = n_pub, sizes[1] = n; with the output buffers NULL only the sizes are returned */
int32_t cgh_shared_witness_read(int32_t curve, const char* path, int32_t protocol, size_t* sizes, uint64_t* pub, uint64_t* a, uint64_t* b);

/* ---- one-shot Groth16 proofs: zkey file -> proof (co-circom.rs:482-506; groth16.rs:113-139) ---------------------------------------- */
/* PlainHipDriver (plain.rs).  full_witness = n_vars elements (leading one, public inputs, private part); r, s: the two blinding
 * scalars the reference draws at groth16.rs:134-135.  out_h (optional): the m quotient evaluations of witness_map_from_matrices. */
int32_t cgh_prove_plain(int32_t device, int32_t curve, const char* zkey_path, const uint64_t* full_witness, const uint64_t* r, const uint64_t* s, uint64_t* out_proof, uint64_t* out_h);
/* Rep3HipProtocol x 3 (three threads, in-process network like tests/src/rep3_network.rs).  pub_in = n_public + 1 values;
 * wit_a[i] / wit_b[i] = party i's replicated shares of the private witness (rep3/fieldshare.rs:233-236); streams[i] = the field
 * elements party i's rng1 produces (party i: rng1 = S_i, rng2 = S_(i-1), rngs.rs:25-46), stream_len each.  out_proofs = 3 proofs
 * (all equal when the run is correct); out_h (optional) = party 0's share of h (a then b). */
int32_t cgh_prove_rep3(int32_t device, int32_t curve, const char* zkey_path, const uint64_t* pub_in, const uint64_t* const* wit_a, const uint64_t* const* wit_b,
                       const uint64_t* const* streams, size_t stream_len, uint64_t* out_proofs, uint64_t* out_h);
/* ShamirHipProtocol x n, threshold t (shamir.rs:196-246).  wit[i] = party i's Shamir shares; streams[i] = its private randomness in
 * the order the reference draws values; preprocess > 0: that many double sharings are generated on the GPU first (shamir.rs:248-300). */
int32_t cgh_prove_shamir(int32_t device, int32_t curve, const char* zkey_path, int32_t n, int32_t t, const uint64_t* pub_in, const uint64_t* const* wit,
                         const uint64_t* const* streams, size_t stream_len, size_t preprocess, uint64_t* out_proofs, uint64_t* out_h);

/* ---- proving sessions: the zkey is read, uploaded, validated (and given per-window precomputed tables) once; a proof then costs what
 * co-circom.rs:503-506 times.  A zkey is fixed for the life of a prover process (zkey.rs:48-71). ------------------------------------- */
/* precompute: 0 = none, -1 = window chosen per table size, > 0 = that window */
int32_t cgh_session_open(int32_t device, int32_t curve, const char* zkey_path, int32_t precompute, void** out_session);
/* flags: bit 0 = skip the point validation.
 * bit 1 (CGH_SESSION_ADDITIVE_H) = REP3 proofs of the session run the ADDITIVE-QUOTIENT variant — an opt-in protocol variant, not the
 * reference's message sequence: the two mul_vec calls of the witness map (groth16.rs:174,190) keep their masked local products and are not
 * re-shared (no 2 x 32 B x m exchange), every MSM multiplies the party's own share component, and the five MSM results become replicated
 * shares again in one round of five points; from there on every value and message is the reference's and the proof is bit-identical.
 * Randomness is drawn exactly as in the reference.  All three parties must open their sessions with the same flag. */
#define CGH_SESSION_SKIP_VALIDATION 1u
#define CGH_SESSION_ADDITIVE_H 2u
/* cgh_session_open_multi with more than one device first runs cg_device_preflight (distinct GPUs, one checked 1 MiB copy per ordered pair;
 * pairs WITHOUT peer access are accepted — their copies go through the host, slower but checked) and refuses the list if it fails.  CGH_SESSION_SHARED_DEVICES: the list may name a GPU more than once (one-GPU tests and planning
 * runs: every "device" is then a context pair on that GPU) — pairs on one GPU are checked as local copies. */
#define CGH_SESSION_SHARED_DEVICES 4u
int32_t cgh_session_open_ex(int32_t device, int32_t curve, const char* zkey_path, int32_t precompute, uint32_t flags, void** out_session);
/* Several GPUs of one node for one party: devices[0] runs the witness map and slice 0 of every MSM
 * (mpc-core/src/protocols/rep3.rs:934-947 is linear in the (scalar, point) pairs), devices[i] slice i; scalar slices move device to
 * device (cg_dev_copy_peer), partial sums are folded on the host.  The prove calls below accept either kind of session. */
int32_t cgh_session_open_multi(const int32_t* devices, int32_t n_devices, int32_t curve, const char* zkey_path, int32_t precompute, uint32_t flags, void** out_session);
int32_t cgh_session_close(void* session);
/* seconds[1] (optional): wall time of the prove */
int32_t cgh_session_prove_plain(void* session, const uint64_t* full_witness, const uint64_t* r, const uint64_t* s, uint64_t* out_proof, double* seconds);
/* three REP3 parties sharing the session's GPU(s).  seconds[2] (optional): [0] the three parties together; [1] party 0 ALONE on the GPU,
 * replaying the messages it received in the first run (its proof must repeat bit for bit): one party's cost with its peers elsewhere. */
int32_t cgh_session_prove_rep3(void* session, const uint64_t* pub_in, const uint64_t* const* wit_a, const uint64_t* const* wit_b,
                               const uint64_t* const* streams, size_t stream_len, uint64_t* out_proofs, double* seconds);

/* ---- ONE party of a REP3 proof, with the caller's network and the caller's correlated randomness ------------------------------------
 * This is what `co-circom generate-proof --protocol REP3` runs per process (co-circom.rs:484-506): the party's own shares, a network
 * to its two peers (Rep3MpcNet, mpc-core/src/protocols/rep3/network.rs:13-64) and the randomness it agreed on with them
 * (Rep3CorrelatedRng, rep3/rngs.rs:25-62, set up by Rep3Protocol::new, rep3.rs:385-398).  The network rounds and every random draw stay
 * with the caller — the callbacks below are closures over `Rep3MpcNet::{send_bytes, recv_bytes}` and `Rep3Rand` in the Rust binding
 * (rust/mpc-core-hip/src/session.rs) — and the prove itself runs on the session's resident tables exactly as cgh_session_prove_rep3
 * does for three co-located parties.
 *
 * Messages are opaque byte strings between parties that run THIS backend: field vectors travel as 32-byte Montgomery limbs, points
 * as packed affine coordinates; one send_* call is one message and must be delivered whole, in order, to the matching recv_* call of
 * the peer (which asks for the same number of bytes).  The two `mul_vec` exchanges (rep3.rs:650-670) are sent in chunks of at most
 * 4 MiB so that the transfer overlaps the GPU work; the number and order of messages is a function of the circuit size only.
 * All callbacks are called from the thread that called the prove, return 0 on success and anything else on an I/O failure (the
 * prove then fails with that code in its message). */
typedef struct cgh_rep3_net {
    void* user;
    int32_t party_id;                                                      /* get_id(): 0, 1 or 2 (network.rs:15) */
    int32_t (*send_next)(void* user, const void* data, size_t bytes);      /* to party id + 1 (network.rs:30-37) */
    int32_t (*recv_prev)(void* user, void* data, size_t bytes);            /* from party id - 1 (network.rs:57-64) */
    int32_t (*send_prev)(void* user, const void* data, size_t bytes);      /* send(id.prev_id(), ..) (rep3.rs:746-753; co-plonk openings) */
    int32_t (*recv_next)(void* user, void* data, size_t bytes);
    /* optional (may be NULL): the next message from the previous party where the transport already holds it, in page-locked memory that
     * stays valid until the prove returns; NULL result = not available, recv_prev is used */
    const void* (*recv_prev_pinned)(void* user, size_t bytes);
} cgh_rep3_net;
typedef struct cgh_rep3_rand {
    void* user;
    /* n x Rep3Rand::masking_field_element (rngs.rs:37-40: rand(rng1) - rand(rng2)), Montgomery.  `buf` is page-locked scratch of n
     * elements owned by the library; the callee either fills it and stores buf in *out, or stores a pointer to n elements of its own
     * (valid until the prove returns; page-locked memory from cg_host_alloc is uploaded without a staging copy). */
    int32_t (*masking_field_elements)(void* user, size_t n, uint64_t* buf, const uint64_t** out);
    /* Rep3Rand::random_fes (rngs.rs:42-46): a = rand(rng1), b = rand(rng2) — one replicated random share (Rep3Protocol::rand, rep3.rs:595-598) */
    int32_t (*random_fes)(void* user, uint64_t* a, uint64_t* b);
    /* Rep3Rand::masking_ec_element::<C> (rngs.rs:48-51), Jacobian Montgomery (X, Y, Z); group: CG_G1 / CG_G2 */
    int32_t (*masking_ec_element)(void* user, int32_t group, uint64_t* out_jacobian);
} cgh_rep3_rand;
/* Groth16 proof of one REP3 party on an open session.  pub_in = n_public + 1 values, wit_a / wit_b = this party's replicated shares of
 * the private witness (page-locked buffers are read by DMA where they lie).  seconds[1] (optional) = wall time of the prove. */
int32_t cgh_session_prove_rep3_party(void* session, const uint64_t* pub_in, const uint64_t* wit_a, const uint64_t* wit_b,
                                     const cgh_rep3_net* net, const cgh_rep3_rand* rnd, uint64_t* out_proof, double* seconds);

/* Optional description of Rep3Rand's two generators, for the O(n) draws.  Rep3Rand::masking_field_element (rngs.rs:37-46) is
 * `F::rand(&mut rng1) - F::rand(&mut rng2)` with `RngType = rand_chacha::ChaCha12Rng` (mpc-core/src/lib.rs:10): 2 n rejection-sampled draws
 * per mul_vec on one host thread in the reference (4 x 2^22 per 2^22-constraint proof — several times the GPU's whole prove).  A ChaCha
 * stream is addressable by position, so with this table the library makes those draws on the GPU (cg_chacha12_fr_rand_dev: every candidate
 * in parallel, accepted ones compacted in order) and never moves a mask over PCIe: get_state reports seed and word position of rng1 / rng2
 * (ChaCha12Rng::get_seed, get_word_pos) right before a vector of masks is due, set_word_pos (ChaCha12Rng::set_word_pos) puts both
 * generators behind the draws taken, so the caller's next draw — random_fes, masking_ec_element, the next proof — is the one the reference
 * would make.  The O(1) draws stay with cgh_rep3_rand.  Vectors shorter than 2^11 elements (CGH_DEVICE_MASKS_MIN overrides) still come
 * through masking_field_elements. */
typedef struct cgh_rep3_chacha {
    void* user;
    int32_t (*get_state)(void* user, uint8_t* seed1_32, uint64_t* word_pos1, uint8_t* seed2_32, uint64_t* word_pos2);
    int32_t (*set_word_pos)(void* user, uint64_t word_pos1, uint64_t word_pos2);
} cgh_rep3_chacha;
/* cgh_session_prove_rep3_party with the generators described (streams may be NULL: the entry above) */
int32_t cgh_session_prove_rep3_party_ex(void* session, const uint64_t* pub_in, const uint64_t* wit_a, const uint64_t* wit_b, const cgh_rep3_net* net,
                                        const cgh_rep3_rand* rnd, const cgh_rep3_chacha* streams, uint64_t* out_proof, double* seconds);

/* The Shamir twin (co-circom.rs:507-527: `ShamirMpcNet::new`, `ShamirProtocol::new(t, net)`, `prover.prove`): ONE of n parties, threshold t,
 * with the caller's any-to-any network (shamir/network.rs:17-59: one message = one send / recv pair between two parties) and the caller's
 * PRIVATE randomness (ShamirProtocol's own RNG: the coefficients of ShamirCore::share, shamir/shamir_core.rs:8-31, the secrets of
 * buffer_triples, shamir.rs:923-1010 — F::rand draws in the order the reference makes them).  preprocess > 0: that many double sharings are
 * generated up front on the GPU (ShamirProtocol::preprocess, shamir.rs:248-250); 0 = the reference's lazy batches of 1024. */
typedef struct cgh_shamir_net {
    void* user;
    int32_t party_id, num_parties;                                              /* get_id(), get_num_parties() */
    int32_t (*send)(void* user, int32_t to, const void* data, size_t bytes);
    int32_t (*recv)(void* user, int32_t from, void* data, size_t bytes);
} cgh_shamir_net;
typedef struct cgh_shamir_rand {
    void* user;
    int32_t (*random_field_elements)(void* user, size_t n, uint64_t* out);      /* n x F::rand(&mut self.rng), Montgomery */
} cgh_shamir_rand;
int32_t cgh_session_prove_shamir_party(void* session, int32_t threshold, const uint64_t* pub_in, const uint64_t* wit, const cgh_shamir_net* net,
                                       const cgh_shamir_rand* rnd, size_t preprocess, uint64_t* out_proof, double* seconds);
/* The same party with its private randomness as ONE 32-byte seed (what the caller's rng yields with `rng.gen::<[u8; 32]>()`): ShamirProtocol's
 * generator is `RngType::from_entropy()` (shamir.rs:211-246) and no peer ever reproduces its draws, so the library runs a ChaCha12 generator of
 * its own from that seed — `F::rand` per draw as in the reference, the preprocess(amount) batch of amount * (1 + 3t) draws on the GPU
 * (cg_chacha12_fr_rand_dev), the lazy batches and O(1) draws on the host, all from the one stream in the reference's draw order. */
int32_t cgh_session_prove_shamir_party_seeded(void* session, int32_t threshold, const uint64_t* pub_in, const uint64_t* wit, const cgh_shamir_net* net,
                                              const uint8_t* seed32, size_t preprocess, uint64_t* out_proof, double* seconds);

/* Transports and randomness sources for tests, benches and single-box deployments; each fills a callback table for the entry above.
 * cgh_loopback_*: three parties of one process joined by in-memory queues (the role of tests/src/rep3_network.rs).  record != 0 keeps
 * a copy of everything that party receives (large messages in page-locked memory); cgh_loopback_replay_net then serves that traffic
 * back to the same party running ALONE — one party's cost with its peers on other machines, network time excluded. */
int32_t cgh_loopback_create(void** out_hub);
int32_t cgh_loopback_net(void* hub, int32_t party, int32_t record, cgh_rep3_net* out);
int32_t cgh_loopback_replay_net(void* hub, int32_t party, cgh_rep3_net* out);
/* a party failed: the others' pending and later receives fail instead of waiting for messages that will never come */
int32_t cgh_loopback_abort(void* hub);
int32_t cgh_loopback_destroy(void* hub);
/* The Shamir twin of the loopback: num_parties (3..64) parties of one process joined by in-memory queues, any to any
 * (the role of tests/src/shamir_network.rs); net(party) fills that party's callback table. */
int32_t cgh_shamir_loopback_create(int32_t num_parties, void** out_hub);
int32_t cgh_shamir_loopback_net(void* hub, int32_t party, int32_t record, cgh_shamir_net* out);
/* record != 0 keeps what that party receives (large messages in page-locked memory); the replay table then serves it again to the same
 * party running ALONE, sends dropped — one party's cost with its peers elsewhere, network time excluded (as cgh_loopback_replay_net). */
int32_t cgh_shamir_loopback_replay_net(void* hub, int32_t party, cgh_shamir_net* out);
int32_t cgh_shamir_loopback_abort(void* hub);
int32_t cgh_shamir_loopback_destroy(void* hub);
/* Rep3Rand over two pre-generated streams of field elements (rng1[k], rng2[k] = the k-th F::rand of each ChaCha stream): masks are
 * rng1[k] - rng2[k] (computed here, once, into page-locked memory), random_fes returns (rng1[k], rng2[k]), masking_ec_element is
 * G * rng1[k] - G * rng2[k] (a stand-in for C::rand, which no caller can reproduce without arkworks; the proof does not depend on it).
 * Every draw advances k by one, a masking vector of n elements by n. */
int32_t cgh_stream_rand_create(int32_t curve, const uint64_t* rng1, const uint64_t* rng2, size_t len, void** out_handle, cgh_rep3_rand* out);
int32_t cgh_stream_rand_destroy(void* handle);
/* Rep3Rand over two ChaCha12 generators, as the reference holds it (rngs.rs:25-35: `RngType::from_seed(seed1)`, `from_seed(seed2)`): the
 * host side of the draws (rand_chacha's block function and ark-ff's F::rand restated in this library), for tests, benches and single-box
 * deployments — a Rust caller passes its own Rep3Rand instead (rust/mpc-core-hip/src/session.rs).  out = the three O(1) / host callbacks,
 * out_streams (optional) = the table above.  masking_ec_element is the stand-in of cgh_stream_rand_create (G * rand(rng1) - G * rand(rng2)).
 * cgh_chacha12_fr_rand_host: n x F::rand on the host from a seed and word position, single thread — what the reference's party does. */
int32_t cgh_chacha_rand_create(int32_t curve, const uint8_t* seed1_32, const uint8_t* seed2_32, void** out_handle, cgh_rep3_rand* out, cgh_rep3_chacha* out_streams);
int32_t cgh_chacha_rand_positions(void* handle, uint64_t* word_pos1, uint64_t* word_pos2);
int32_t cgh_chacha_rand_destroy(void* handle);
int32_t cgh_chacha12_fr_rand_host(int32_t curve, const uint8_t* seed32, uint64_t word_pos, size_t n, uint64_t* out, uint64_t* word_pos_after);

/* ---- co-plonk (co-plonk/src/plonk.rs:133-271 drives round1..round5) ------------------------------------------------------------------ */
/* PlainHipDriver through rounds 1..upto (<= 5).  full_witness = n_vars - n_additions elements; blind = the 11 blinding scalars b_1..b_11
 * (round1.rs:346-383 fixes them to 1..11 in its test); commits = 9 packed G1 (a, b, c, z, t1, t2, t3, wxi, wxiw; zero = not reached),
 * challenges = beta, gamma, alpha, xi, v; evals = a, b, c, s1, s2, zw; optional: t_polys = t1 (n+1) | t2 (n+1) | t3 (n+6), poly_z (n+3). */
int32_t cgh_plonk_prove_plain(int32_t device, int32_t curve, const char* zkey_path, const uint64_t* full_witness, const uint64_t* blind, int32_t upto,
                              uint64_t* commits, uint64_t* challenges, uint64_t* evals, uint64_t* t_polys, uint64_t* poly_z);
/* Rep3HipProtocol x 3 through rounds 1..upto.  blind_a[i] / blind_b[i] = party i's shares of b_1..b_11.  out_commits = 3 x 9 packed G1,
 * out_evals = 3 x 6, out_challenges = 3 x 5 (every party must report the same values). */
int32_t cgh_plonk_prove_rep3(int32_t device, int32_t curve, const char* zkey_path, const uint64_t* pub_in, const uint64_t* const* wit_a, const uint64_t* const* wit_b,
                             const uint64_t* const* blind_a, const uint64_t* const* blind_b, const uint64_t* const* streams, size_t stream_len, int32_t upto,
                             uint64_t* out_commits, uint64_t* out_evals, uint64_t* out_challenges);
/* ONE REP3 party of co-plonk over the caller's network and randomness tables (co-circom.rs:560-600: the CLI's plonk branch hands its
 * Rep3Protocol to CoPlonk::prove).  blind_a / blind_b: this party's shares of b_1..b_11, or both NULL to draw them with rand() first
 * (round1.rs:93-99).  out_commits = 9 packed G1, out_evals = 6, out_challenges = 5 (both optional). */
int32_t cgh_plonk_prove_rep3_party(int32_t device, int32_t curve, const char* zkey_path, const uint64_t* pub_in, const uint64_t* wit_a, const uint64_t* wit_b,
                                   const uint64_t* blind_a, const uint64_t* blind_b, const cgh_rep3_net* net, const cgh_rep3_rand* rand, int32_t upto,
                                   uint64_t* out_commits, uint64_t* out_evals, uint64_t* out_challenges);
/* The same party with Rep3Rand's generators described (cgh_rep3_chacha, above): the masking vectors of the rounds' mul_vec calls are drawn on the GPU */
int32_t cgh_plonk_prove_rep3_party_ex(int32_t device, int32_t curve, const char* zkey_path, const uint64_t* pub_in, const uint64_t* wit_a, const uint64_t* wit_b,
                                      const uint64_t* blind_a, const uint64_t* blind_b, const cgh_rep3_net* net, const cgh_rep3_rand* rand,
                                      const cgh_rep3_chacha* streams, int32_t upto, uint64_t* out_commits, uint64_t* out_evals, uint64_t* out_challenges);
/* ShamirHipProtocol x n (threshold t) through rounds 1..upto; outputs as for cgh_plonk_prove_rep3, n parties. */
int32_t cgh_plonk_prove_shamir(int32_t device, int32_t curve, const char* zkey_path, int32_t n, int32_t t, const uint64_t* pub_in, const uint64_t* const* wit,
                               const uint64_t* const* blind, const uint64_t* const* streams, size_t stream_len, int32_t upto,
                               uint64_t* out_commits, uint64_t* out_evals, uint64_t* out_challenges);
/* Keccak256 transcript (co-plonk/src/types.rs:102-227): kinds[i] 0 = scalar, 1 = packed G1 point; out = the challenge */
int32_t cgh_plonk_transcript(int32_t curve, const int32_t* kinds, const uint64_t* const* payloads, int32_t n_items, uint64_t* out_challenge);

/* ---- tooling (bench / tests; not on the prover path) ---------------------------------------------------------------------------------- */
/* synthetic satisfiable circuit of 2^log_m - 2 constraints (n_public = 1, n_vars = domain size = 2^log_m) with a VALID Groth16 CRS from
 * seeded toxic waste, written as snarkjs-format .zkey + .wtns; the point tables are built on the GPU */
int32_t cgh_synth_circuit(int32_t device, int32_t curve, int32_t log_m, uint64_t seed, const char* zkey_path, const char* wtns_path);

#ifdef __cplusplus
}
#endif
#endif
