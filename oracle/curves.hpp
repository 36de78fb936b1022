// ORACLE — TEST INFRASTRUCTURE ONLY (see ff.hpp header).
// Curve constants the reference obtains from ark-bn254 0.4.0 / ark-bls12-381 0.4.0 (`/root/reference/Cargo.toml:33-34`).
// Moduli cross-checked against SURVEY.md §9 and against the zkey headers of the shipped fixtures
// (`co-circom/circom-types/src/groth16/zkey.rs:262-284` rejects a zkey whose header primes differ).
#pragma once
#include "ff.hpp"
#include "ec.hpp"

namespace orc {

struct Bn254 {
    static constexpr int ID = 0;
    typedef Fp<4, 0> Fr;
    typedef Fp<4, 1> Fq;
    typedef Fp2T<Fq> Fq2;
    typedef JacT<Fq, 0> G1;
    typedef JacT<Fq2, 1> G2;
    static const char* circom_name() { return "bn128"; }   // traits.rs:18
    static void init() {
        if (Fr::K.ready) return;
        Fr::init("30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001");
        Fq::init("30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47");
        G1::B = Fq::from_u64(3);
        Fq2 xi = {Fq::from_u64(9), Fq::one()};
        G2::B = xi.inverse().mul_base(Fq::from_u64(3));   // D-type twist: b' = 3/(9+u)
    }
    static Fq2 xi() { return {Fq::from_u64(9), Fq::one()}; }
    static constexpr bool TWIST_D = true;
    static G1::Affine g1_generator() { return {Fq::from_u64(1), Fq::from_u64(2), false}; }
    static G2::Affine g2_generator() {
        return {{Fq::from_dec("10857046999023057135944570762232829481370756359578518086990519993285655852781"),
                 Fq::from_dec("11559732032986387107991004021392285783925812861821192530917403151452391805634")},
                {Fq::from_dec("8495653923123431417604973247489272438418190587263600148770280649306958101930"),
                 Fq::from_dec("4082367875863433681332203403145435568316851327593401208105741076214120093531")},
                false};
    }
};

struct Bls12_381 {
    static constexpr int ID = 1;
    typedef Fp<4, 2> Fr;
    typedef Fp<6, 3> Fq;
    typedef Fp2T<Fq> Fq2;
    typedef JacT<Fq, 2> G1;
    typedef JacT<Fq2, 3> G2;
    static const char* circom_name() { return "bls12381"; }   // traits.rs:31
    static void init() {
        if (Fr::K.ready) return;
        Fr::init("73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001");
        Fq::init("1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab");
        G1::B = Fq::from_u64(4);
        G2::B = {Fq::from_u64(4), Fq::from_u64(4)};       // M-type twist: b' = 4(1+u)
    }
    static Fq2 xi() { return {Fq::one(), Fq::one()}; }
    static constexpr bool TWIST_D = false;
    static G1::Affine g1_generator() {
        return {Fq::from_dec("3685416753713387016781088315183077757961620795782546409894578378688607592378376318836054947676345821548104185464507"),
                Fq::from_dec("1339506544944476473020471379941921221584933875938349620426543736416511423956333506472724655353366534992391756441569"),
                false};
    }
    static G2::Affine g2_generator() {
        return {{Fq::from_dec("352701069587466618187139116011060144890029952792775240219908644239793785735715026873347600343865175952761926303160"),
                 Fq::from_dec("3059144344244213709971259814753781636986470325476647558659373206291635324768958432433509563104347017837885763365758")},
                {Fq::from_dec("1985150602287291935568054521177171638300868978215655730859378665066344726373823718423869104263333984641494340347905"),
                 Fq::from_dec("927553665492332455747201965776037880757740193453592970025027978793976877002675564980949289727957565575433344219582")},
                false};
    }
};

}  // namespace orc
