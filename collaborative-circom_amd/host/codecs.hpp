// JSON encodings of proofs / public inputs and the .shared witness container
#pragma once
#include "formats.hpp"

namespace cgh {

// ---- JSON encodings of proofs and public inputs (circom-types/src/groth16/proof.rs:8-29, traits.rs:186-233, co-circom.rs:540,628) ----
static std::string limbs_to_dec(const uint64_t* limbs, int n) {          // canonical little-endian -> decimal
    std::vector<uint32_t> w(2 * n);
    for (int i = 0; i < n; i++) { w[2 * i] = (uint32_t)limbs[i]; w[2 * i + 1] = (uint32_t)(limbs[i] >> 32); }
    std::string out;
    while (true) {
        uint64_t rem = 0; bool nz = false;
        for (int i = (int)w.size() - 1; i >= 0; i--) { uint64_t cur = (rem << 32) | w[i]; w[i] = (uint32_t)(cur / 1000000000u); rem = cur % 1000000000u; nz = nz || w[i]; }
        char buf[16];
        if (nz) { snprintf(buf, sizeof buf, "%09u", (unsigned)rem); out.insert(0, buf); }
        else { snprintf(buf, sizeof buf, "%u", (unsigned)rem); out.insert(0, buf); break; }
    }
    return out;
}
static void dec_to_limbs(const std::string& sdec, uint64_t* limbs, int n) {   // decimal -> canonical little-endian (must fit)
    std::vector<uint32_t> w(2 * n, 0);
    if (sdec.empty()) throw std::runtime_error("empty number");
    for (char ch : sdec) {
        if (ch < '0' || ch > '9') throw std::runtime_error("invalid decimal digit");
        uint64_t carry = (uint64_t)(ch - '0');
        for (size_t i = 0; i < w.size(); i++) { uint64_t cur = (uint64_t)w[i] * 10u + carry; w[i] = (uint32_t)cur; carry = cur >> 32; }
        if (carry) throw std::runtime_error("number too large for the field");
    }
    for (int i = 0; i < n; i++) limbs[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
}
static std::string fq_dec(const Curve& c, const uint8_t* mont) {
    uint64_t can[6]; CG(cg_fq_to_canonical(c.id, mont, can, 1));
    return limbs_to_dec(can, (int)c.fq() / 8);
}
static bool all_zero(const uint8_t* p, size_t n) { for (size_t i = 0; i < n; i++) if (p[i]) return false; return true; }
static const char* curve_name(const Curve& c) { return c.id == CG_BN254 ? "bn128" : "bls12381"; }    // traits.rs:18,31
static std::string g1_json(const Curve& c, const uint8_t* aff) {
    if (all_zero(aff, c.aff(CG_G1))) return "[\"0\",\"1\",\"0\"]";                                       // traits.rs:190-192
    return "[\"" + fq_dec(c, aff) + "\",\"" + fq_dec(c, aff + c.fq()) + "\",\"1\"]";
}
static std::string g2_json(const Curve& c, const uint8_t* aff) {
    if (all_zero(aff, c.aff(CG_G2))) throw std::runtime_error("the point at infinity has no G2 JSON encoding (the reference unwraps xy(), traits.rs:227)");
    const size_t q = c.fq();
    return "[[\"" + fq_dec(c, aff) + "\",\"" + fq_dec(c, aff + q) + "\"],[\"" + fq_dec(c, aff + 2 * q) + "\",\"" + fq_dec(c, aff + 3 * q) + "\"],[\"1\",\"0\"]]";
}
static std::string proof_to_json(const Curve& c, const uint8_t* packed) {   // packed = A (G1) || B (G2) || C (G1)
    const uint8_t *a = packed, *b = packed + c.aff(CG_G1), *cc = b + c.aff(CG_G2);
    return "{\"pi_a\":" + g1_json(c, a) + ",\"pi_b\":" + g2_json(c, b) + ",\"pi_c\":" + g1_json(c, cc) + ",\"protocol\":\"groth16\",\"curve\":\"" + curve_name(c) + "\"}";
}
// the decimal strings of a JSON document, in order (the proof schema is fixed: keys pi_a, pi_b, pi_c carry 3 + 6 + 3 numbers)
static std::vector<std::string> json_numbers_after(const std::string& js, const char* key, size_t count) {
    size_t pos = js.find(std::string("\"") + key + "\"");
    if (pos == std::string::npos) throw std::runtime_error(std::string("missing key ") + key);
    pos = js.find(':', pos);
    std::vector<std::string> out;
    while (out.size() < count) {
        size_t q0 = js.find('"', pos + 1);
        if (q0 == std::string::npos) throw std::runtime_error("truncated proof JSON");
        size_t q1 = js.find('"', q0 + 1);
        if (q1 == std::string::npos) throw std::runtime_error("truncated proof JSON");
        out.push_back(js.substr(q0 + 1, q1 - q0 - 1));
        pos = q1;
    }
    return out;
}
static void proof_from_json(const Curve& c, const std::string& js, uint8_t* packed) {
    if (js.find(std::string("\"") + curve_name(c) + "\"") == std::string::npos) throw std::runtime_error("proof is for another curve");
    const int nl = (int)c.fq() / 8;
    auto put = [&](const std::string& d, uint8_t* dst) { uint64_t can[6] = {0}; dec_to_limbs(d, can, nl); CG(cg_fq_from_canonical(c.id, can, dst, 1)); };
    auto g1 = [&](const char* key, uint8_t* dst) {
        auto v = json_numbers_after(js, key, 3);
        if (v[2] == "0") { memset(dst, 0, c.aff(CG_G1)); return; }          // projective z = 0: infinity
        if (v[2] != "1") throw std::runtime_error("only z = 1 / z = 0 G1 encodings are produced by circom tools");
        put(v[0], dst); put(v[1], dst + c.fq());
    };
    g1("pi_a", packed);
    auto v = json_numbers_after(js, "pi_b", 6);
    if (v[4] != "1" || v[5] != "0") throw std::runtime_error("only z = (1, 0) G2 encodings are produced by circom tools");
    uint8_t* b = packed + c.aff(CG_G1);
    for (int i = 0; i < 4; i++) put(v[i], b + i * c.fq());
    g1("pi_c", b + c.aff(CG_G2));
}

// `.shared` witness files (co-circom.rs:330,400,449: `bincode::serialize_into(file, &SharedWitness)`).  Layout restated from the
// types, NOT pinned by a reference fixture (the snapshot ships no .shared file):
//   SharedWitness { public_inputs, witness } with both fields going through serde_compat::ark_se (co-circom-snarks/src/lib.rs:24-41,
//   serde_compat.rs:5-13) = serialize_bytes(ark-compressed value) = u64 LE byte length, then the bytes;
//   ark-compressed Vec<F> = u64 LE element count, then 32-byte canonical little-endian field elements;
//   Rep3PrimeFieldShareVec { a, b } (rep3/fieldshare.rs:232-236) = Vec a then Vec b; ShamirPrimeFieldShareVec { a } (shamir/fieldshare.rs:152-155) = Vec a.
static void put_u64(Bytes& o, uint64_t v) { for (int i = 0; i < 8; i++) o.push_back((uint8_t)(v >> (8 * i))); }
static void put_vec(const Curve& c, Bytes& o, const Fr* v, size_t n) {
    put_u64(o, n);
    std::vector<Fr> can(n); if (n) CG(cg_fr_to_canonical(c.id, v, can.data(), n));
    const uint8_t* p = (const uint8_t*)can.data(); o.insert(o.end(), p, p + n * 32);
}
static std::vector<Fr> get_vec(const Curve& c, Cursor& cur) {
    const uint64_t n = cur.u64();
    if (cur.off > cur.n || n > (cur.n - cur.off) / 32) throw std::runtime_error("invalid data: vector length exceeds the file");   // (n * 32 would wrap for n >= 2^59)
    cur.need(n * 32);
    std::vector<Fr> raw(n), out(n); cur.bytes(raw.data(), n * 32);
    for (const Fr& e : raw) { for (int l = 3; l >= 0; l--) { if (e.v[l] < MOD_R[c.id][l]) break; if (e.v[l] > MOD_R[c.id][l] || l == 0) throw std::runtime_error("invalid data: field element not reduced"); } }
    if (n) CG(cg_fr_from_canonical(c.id, raw.data(), out.data(), n));
    return out;
}
static void write_shared_witness(const Curve& c, const std::string& path, const std::vector<Fr>& pub, const std::vector<Fr>& a, const std::vector<Fr>* b) {
    Bytes f1, f2, out;
    put_vec(c, f1, pub.data(), pub.size());
    put_vec(c, f2, a.data(), a.size()); if (b) put_vec(c, f2, b->data(), b->size());
    put_u64(out, f1.size()); out.insert(out.end(), f1.begin(), f1.end());
    put_u64(out, f2.size()); out.insert(out.end(), f2.begin(), f2.end());
    FILE* f = fopen(path.c_str(), "wb"); if (!f) throw std::runtime_error("cannot open " + path);
    const bool ok = fwrite(out.data(), 1, out.size(), f) == out.size(); fclose(f);
    if (!ok) throw std::runtime_error("short write " + path);
}
static void read_shared_witness(const Curve& c, const std::string& path, bool rep3, std::vector<Fr>& pub, std::vector<Fr>& a, std::vector<Fr>& b) {
    Bytes buf = slurp(path);
    Cursor cur{buf.data(), buf.size()};
    const uint64_t l1 = cur.u64(); cur.need(l1);
    { Cursor f{buf.data() + cur.off, (size_t)l1}; pub = get_vec(c, f); if (f.off != l1) throw std::runtime_error("trailing bytes in public_inputs"); }
    cur.off += l1;
    const uint64_t l2 = cur.u64(); cur.need(l2);
    { Cursor f{buf.data() + cur.off, (size_t)l2}; a = get_vec(c, f); if (rep3) b = get_vec(c, f); if (f.off != l2) throw std::runtime_error("witness share does not match the protocol (REP3 has two vectors, Shamir one)"); }
    cur.off += l2;
    if (cur.off != buf.size()) throw std::runtime_error("trailing bytes after the shared witness");
    if (rep3 && a.size() != b.size()) throw std::runtime_error("REP3 share components differ in length");
}

// PlonkProof <-> JSON (circom-types/src/plonk/proof.rs:8-74): nine G1 points A, B, C, Z, T1, T2, T3, Wxi, Wxiw, six evaluations, tags
static const char* const PLONK_PT_KEYS[9] = {"A", "B", "C", "Z", "T1", "T2", "T3", "Wxi", "Wxiw"};
static const char* const PLONK_EV_KEYS[6] = {"eval_a", "eval_b", "eval_c", "eval_s1", "eval_s2", "eval_zw"};
static std::string plonk_proof_to_json(const Curve& c, const uint8_t* commits /* 9 packed G1 */, const Fr* evals /* 6 */) {
    std::string js = "{";
    for (int i = 0; i < 9; i++) js += std::string("\"") + PLONK_PT_KEYS[i] + "\":" + g1_json(c, commits + i * c.aff(CG_G1)) + ",";
    for (int i = 0; i < 6; i++) { uint64_t can[4]; CG(cg_fr_to_canonical(c.id, evals[i].v, can, 1)); js += std::string("\"") + PLONK_EV_KEYS[i] + "\":\"" + limbs_to_dec(can, 4) + "\","; }
    return js + "\"protocol\":\"plonk\",\"curve\":\"" + curve_name(c) + "\"}";
}
static void plonk_proof_from_json(const Curve& c, const std::string& js, uint8_t* commits, Fr* evals) {
    if (js.find(std::string("\"") + curve_name(c) + "\"") == std::string::npos) throw std::runtime_error("proof is for another curve");
    if (js.find("\"plonk\"") == std::string::npos) throw std::runtime_error("not a plonk proof");
    const int nl = (int)c.fq() / 8;
    for (int i = 0; i < 9; i++) {
        auto v = json_numbers_after(js, PLONK_PT_KEYS[i], 3);
        uint8_t* dst = commits + i * c.aff(CG_G1);
        if (v[2] == "0") { memset(dst, 0, c.aff(CG_G1)); continue; }
        if (v[2] != "1") throw std::runtime_error("only z = 1 / z = 0 G1 encodings are produced by circom tools");
        for (int k = 0; k < 2; k++) { uint64_t can[6] = {0}; dec_to_limbs(v[k], can, nl); CG(cg_fq_from_canonical(c.id, can, dst + k * c.fq(), 1)); }
    }
    for (int i = 0; i < 6; i++) { auto v = json_numbers_after(js, PLONK_EV_KEYS[i], 1); uint64_t can[4] = {0}; dec_to_limbs(v[0], can, 4); CG(cg_fr_from_canonical(c.id, can, evals[i].v, 1)); }
}


}  // namespace cgh
