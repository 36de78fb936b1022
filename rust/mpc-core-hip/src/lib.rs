//! MI355X drivers for co-circom: `PlainHipDriver`, `Rep3HipProtocol`, `ShamirHipProtocol` implement the reference's driver traits
//! (`mpc-core/src/traits.rs`: `PrimeFieldMpcProtocol` :43-223, `EcMpcProtocol` :472-521, `PairingEcMpcProtocol` :523-533, `FFTProvider`
//! :535-558, `MSMProvider` :561-568), so `CoGroth16<T, P>` (`co-groth16/src/groth16.rs:80-87`) and `CoPlonk<T, P>` accept them as `T`
//! unchanged; the only edit in the CLI is which driver is constructed (`co-circom/co-circom/src/bin/co-circom.rs:497-499`):
//!
//! ```ignore
//! // before:  let protocol = Rep3Protocol::<P::ScalarField, _>::new(net)?;
//! let mut protocol = Rep3HipProtocol::<P::ScalarField, _>::new(net, /* device */ 0)?;
//! protocol.gpu().precompute = -1;                                   // per-window tables, window by table size
//! for q in [&zkey.a_query, &zkey.b_g1_query, &zkey.l_query, &zkey.h_query] { protocol.gpu().register(q); }
//! protocol.gpu().register(&zkey.b_g2_query);
//! let mut prover = CoGroth16::<_, P>::new(protocol);
//! ```
//!
//! The per-call drivers above move the caller's `Vec`s over PCIe on every trait call.  The fast path for Groth16 is `Groth16Session`
//! (session.rs): the zkey stays resident on the GPU(s) and one call proves for this party, with the network and the randomness of the
//! stock `Rep3Protocol` behind C callbacks (`rust/co-circom-hip-backend.patch` wires it to `co-circom generate-proof --backend hip`).
//!
//! SOURCE ONLY here: the image this repository is built in has no Rust toolchain.  The C++ host mirror
//! (`collaborative-circom_amd/host/ (headers per layer: formats, network, driver, groth16, plonk, codecs, synth; entry points in capi_*.cpp)`) runs the same call sequence against the same library and is what the tests
//! exercise; `include/cogroth16_host.h` is its ABI.
pub mod ffi;
pub mod gpu;
pub mod plain;
pub mod rep3;
pub mod session;
pub mod shamir;

pub use gpu::{curve_id, group_id, Gpu, Layout};
pub use plain::PlainHipDriver;
pub use rep3::Rep3HipProtocol;
pub use session::{plonk_prove_rep3, Groth16Session};
pub use shamir::ShamirHipProtocol;
