// C entry points: files, codecs, validation switch, synthetic circuits (include/cogroth16_host.h)
#include "codecs.hpp"
#include "synth.hpp"
#include "capi_common.hpp"

thread_local std::string g_host_err;

extern "C" {


const char* cgh_last_error(void) { return g_host_err.c_str(); }

// info: n_vars, n_public, domain_size, pow, num_constraints, nnzA, nnzB
int32_t cgh_zkey_info(int32_t curve, const char* path, size_t* info) {
    try {
        cgh::ZKey z = cgh::read_zkey(curve, path, true);
        info[0] = z.n_vars; info[1] = z.n_public; info[2] = z.domain_size; info[3] = z.pow; info[4] = z.num_constraints; info[5] = z.col[0].size(); info[6] = z.col[1].size();
        return 0;
    } catch (const std::exception& e) { g_host_err = e.what(); return 1; }
}
// zkey -> device with the parser's point validation done on the GPU (traits.rs:107-155); 0 = every point valid.  seconds[0] = file
// read + section decode (host), seconds[1] = upload + validation (device)
int32_t cgh_zkey_validate(int32_t device, int32_t curve, const char* path, double* seconds) {
    cg_ctx* ctx = nullptr;
    try {
        using namespace cgh;
        auto t0 = std::chrono::steady_clock::now();
        ZKey z = read_zkey(curve, path);
        auto t1 = std::chrono::steady_clock::now();
        if (cg_ctx_create(device, &ctx)) die("cg_ctx_create");
        std::vector<Fr> pub(z.n_public + 1);
        DeviceZKey dz = upload_zkey(ctx, z, pub, 1);
        release_zkey(ctx, dz);
        cg_ctx_destroy(ctx);
        auto t2 = std::chrono::steady_clock::now();
        if (seconds) { seconds[0] = std::chrono::duration<double>(t1 - t0).count(); seconds[1] = std::chrono::duration<double>(t2 - t1).count(); }
        return 0;
    } catch (const std::exception& e) { g_host_err = e.what(); if (ctx) cg_ctx_destroy(ctx); return 1; }
}
static int32_t copy_out(const std::string& js, char* out, size_t cap) {
    if (js.size() + 1 > cap) { g_host_err = "buffer too small"; return 1; }
    memcpy(out, js.c_str(), js.size() + 1);
    return 0;
}
// Groth16Proof <-> JSON (proof.rs:8-29); proof = A || B || C packed affine Montgomery as returned by cgh_prove_*
int32_t cgh_proof_to_json(int32_t curve, const uint64_t* proof, char* out, size_t cap) {
    try { return copy_out(cgh::proof_to_json(cgh::Curve{curve}, (const uint8_t*)proof), out, cap); }
    catch (const std::exception& e) { g_host_err = e.what(); return 1; }
}
int32_t cgh_proof_from_json(int32_t curve, const char* json, uint64_t* out_proof) {
    try { cgh::proof_from_json(cgh::Curve{curve}, json, (uint8_t*)out_proof); return 0; }
    catch (const std::exception& e) { g_host_err = e.what(); return 1; }
}
int32_t cgh_plonk_proof_to_json(int32_t curve, const uint64_t* commits, const uint64_t* evals, char* out, size_t cap) {
    try { return copy_out(cgh::plonk_proof_to_json(cgh::Curve{curve}, (const uint8_t*)commits, (const cgh::Fr*)evals), out, cap); }
    catch (const std::exception& e) { g_host_err = e.what(); return 1; }
}
int32_t cgh_plonk_proof_from_json(int32_t curve, const char* json, uint64_t* out_commits, uint64_t* out_evals) {
    try { cgh::plonk_proof_from_json(cgh::Curve{curve}, json, (uint8_t*)out_commits, (cgh::Fr*)out_evals); return 0; }
    catch (const std::exception& e) { g_host_err = e.what(); return 1; }
}
// .shared witness files; protocol: 0 = REP3 (components a, b), 1 = Shamir (a only).  All values Montgomery on this side of the call.
int32_t cgh_shared_witness_write(int32_t curve, const char* path, int32_t protocol, const uint64_t* pub, size_t n_pub, const uint64_t* a, const uint64_t* b, size_t n) {
    try {
        using namespace cgh;
        std::vector<Fr> p((const Fr*)pub, (const Fr*)pub + n_pub), va((const Fr*)a, (const Fr*)a + n), vb;
        if (protocol == 0) vb.assign((const Fr*)b, (const Fr*)b + n);
        write_shared_witness(Curve{curve}, path, p, va, protocol == 0 ? &vb : nullptr);
        return 0;
    } catch (const std::exception& e) { g_host_err = e.what(); return 1; }
}
// sizes[0] = n_pub, sizes[1] = n; with out buffers NULL only the sizes are returned
int32_t cgh_shared_witness_read(int32_t curve, const char* path, int32_t protocol, size_t* sizes, uint64_t* pub, uint64_t* a, uint64_t* b) {
    try {
        using namespace cgh;
        std::vector<Fr> p, va, vb;
        read_shared_witness(Curve{curve}, path, protocol == 0, p, va, vb);
        sizes[0] = p.size(); sizes[1] = va.size();
        if (pub) memcpy(pub, p.data(), p.size() * 32);
        if (a) memcpy(a, va.data(), va.size() * 32);
        if (b && protocol == 0) memcpy(b, vb.data(), vb.size() * 32);
        return 0;
    } catch (const std::exception& e) { g_host_err = e.what(); return 1; }
}
// public.json (co-circom.rs:620-628): the public signals without the leading constant 1, as decimal strings; pub = n Montgomery elements
int32_t cgh_public_to_json(int32_t curve, const uint64_t* pub, size_t n, char* out, size_t cap) {
    try {
        std::string js = "[";
        for (size_t i = 0; i < n; i++) {
            uint64_t can[4]; if (cg_fr_to_canonical(curve, pub + 4 * i, can, 1)) cgh::die("cg_fr_to_canonical");
            js += (i ? ",\"" : "\"") + cgh::limbs_to_dec(can, 4) + "\"";
        }
        return copy_out(js + "]", out, cap);
    } catch (const std::exception& e) { g_host_err = e.what(); return 1; }
}
int32_t cgh_read_wtns(int32_t curve, const char* path, uint64_t* out, size_t cap, size_t* n) {
    try {
        auto w = cgh::read_wtns(curve, path);
        *n = w.size();
        if (out) { if (w.size() > cap) { g_host_err = "buffer too small"; return 1; } memcpy(out, w.data(), w.size() * 32); }
        return 0;
    } catch (const std::exception& e) { g_host_err = e.what(); return 1; }
}
int32_t cgh_set_zkey_validation(int32_t on) { cgh::g_validate_zkey.store(on ? 1 : 0); return 0; }
int32_t cgh_set_option(int32_t option, int64_t value) {
    if (option < 1 || option >= CGH_OPT_COUNT || value < 0) { g_host_err = "cgh_set_option: unknown option or negative value"; return 1; }
    cgh::g_host_options.v[option].store(value); return 0;
}
int32_t cgh_get_option(int32_t option, int64_t* value) {
    if (option < 1 || option >= CGH_OPT_COUNT || !value) { g_host_err = "cgh_get_option: unknown option"; return 1; }
    *value = cgh::g_host_options.v[option].load(); return 0;
}
// synthetic satisfiable circuit of 2^log_m - 2 constraints with a valid CRS, written as .zkey + .wtns (bench / test tooling)
int32_t cgh_synth_circuit(int32_t device, int32_t curve, int32_t log_m, uint64_t seed, const char* zkey_path, const char* wtns_path) {
    try { cgh::synth_circuit(device, curve, log_m, seed, zkey_path, wtns_path); return 0; }
    catch (const std::exception& e) { g_host_err = e.what(); return 1; }
}

}  // extern "C"
