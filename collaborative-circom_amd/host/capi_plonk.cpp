// C entry points: co-plonk (co-plonk/src/plonk.rs:133-271 drives round1..round5)
#include "plonk.hpp"
#include "capi_common.hpp"

extern "C" {

// info: n_vars, n_public, domain_size, power, n_additions, n_constraints
int32_t cgh_plonk_zkey_info(int32_t curve, const char* path, size_t* info) {
    try {
        cgh::PlonkZKey z = cgh::read_plonk_zkey(curve, path);
        info[0] = z.n_vars; info[1] = z.n_public; info[2] = z.domain_size; info[3] = z.power; info[4] = z.n_additions; info[5] = z.n_constraints;
        return 0;
    } catch (const std::exception& e) { g_host_err = e.what(); return 1; }
}
// ---- co-plonk entry points -------------------------------------------------------------------------------------------------------
namespace {
struct PlonkOut { uint64_t* commits; uint64_t* challenges; uint64_t* evals; uint64_t* t_polys; uint64_t* poly_z; };
// runs rounds 1..upto on `driver` and stores what has been computed (slot layout of cgh_plonk_prove_plain)
void plonk_run(cgh::HipDriver& driver, const cgh::PlonkZKey& z, const cg_bases* tau, const std::vector<cgh::Fr>& pub, const cgh::ShareVec& wit, const cgh::FieldShare* b, int upto, const PlonkOut& o) {
    using namespace cgh;
    const Curve& c = z.curve; const size_t psz = c.aff(CG_G1);
    auto put = [&](int slot, const Point& p) { if (!o.commits) return; Bytes a = pt_to_affine(c, p); memcpy((uint8_t*)o.commits + slot * psz, a.data(), psz); };
    auto putf = [&](uint64_t* dst, int slot, const Fr& f) { if (dst) memcpy(dst + 4 * slot, f.v, 32); };
    CoPlonk pk(driver, z, tau, pub, b);
    pk.round1(wit);
    for (int k = 0; k < 3; k++) put(k, pk.commit[k]);
    if (upto >= 2) {
        pk.round2(); put(3, pk.commit_z); putf(o.challenges, 0, pk.beta); putf(o.challenges, 1, pk.gamma);
        if (o.poly_z) CG(cg_dev_download(driver.ctx, o.poly_z, pk.poly_z.c[0], pk.poly_z.n * 32));
    }
    if (upto >= 3) {
        pk.round3(); for (int k = 0; k < 3; k++) put(4 + k, pk.commit_t[k]); putf(o.challenges, 2, pk.alpha);
        if (o.t_polys) { size_t off = 0; for (int k = 0; k < 3; k++) { CG(cg_dev_download(driver.ctx, o.t_polys + off * 4, pk.tpart[k].c[0], pk.tpart[k].n * 32)); off += pk.tpart[k].n; } }
    }
    if (upto >= 4) {
        pk.round4(); putf(o.challenges, 3, pk.xi);
        const Fr ev[6] = {pk.ev_a, pk.ev_b, pk.ev_c, pk.ev_s1, pk.ev_s2, pk.ev_zw};
        for (int i = 0; i < 6; i++) putf(o.evals, i, ev[i]);
    }
    if (upto >= 5) { pk.round5(); putf(o.challenges, 4, pk.v[0]); put(7, pk.commit_wxi); put(8, pk.commit_wxiw); }
    driver.verify_received_vectors();                                              // range check of the vectors received from peers (counted on the device)
}
}  // namespace
// PlainHipDriver through rounds 1..upto (<= 5).  full_witness = n_vars - n_additions Montgomery elements (Groth16-style, leading one);
// blind = 11 Fr; commits = 9 packed G1 (a, b, c, z, t1, t2, t3, wxi, wxiw; zero = not reached), challenges = beta, gamma, alpha, xi, v;
// evals = a, b, c, s1, s2, zw; optional: t_polys = t1 (n+1) | t2 (n+1) | t3 (n+6), poly_z (n+3)
int32_t cgh_plonk_prove_plain(int32_t device, int32_t curve, const char* zkey_path, const uint64_t* full_witness, const uint64_t* blind, int32_t upto,
                              uint64_t* commits, uint64_t* challenges, uint64_t* evals, uint64_t* t_polys, uint64_t* poly_z) {
    cg_ctx* ctx = nullptr;
    try {
        using namespace cgh;
        PlonkZKey z = read_plonk_zkey(curve, zkey_path);
        if (cg_ctx_create(device, &ctx)) die("cg_ctx_create");
        const Curve& c = z.curve;
        if (commits) memset(commits, 0, 9 * c.aff(CG_G1)); if (challenges) memset(challenges, 0, 5 * 32); if (evals) memset(evals, 0, 6 * 32);
        cg_bases* tau = nullptr; CG(cg_bases_register(ctx, c.id, CG_G1, z.p_tau.data(), z.domain_size + 6, c.aff(CG_G1), -1, &tau));
        if (validate_by_default()) { try { validate_bases(ctx, tau, "p_tau"); } catch (...) { cg_bases_release(tau); throw; } }   // the zkey parser's per-point checks
        const Fr* w = (const Fr*)full_witness;
        std::vector<Fr> pub(w, w + z.n_public + 1);
        {
            HipDriver driver(ctx, c, Mode::Plain, nullptr);
            ShareVec wit = driver.upload_vec(w + z.n_public + 1, nullptr, z.n_vars - z.n_additions - z.n_public - 1);
            FieldShare b[11]; for (int i = 0; i < 11; i++) { memcpy(b[i].c[0].v, blind + 4 * i, 32); b[i].c[1] = b[i].c[0]; }
            plonk_run(driver, z, tau, pub, wit, b, upto, PlonkOut{commits, challenges, evals, t_polys, poly_z});
            driver.free_vec(wit);
        }
        cg_bases_release(tau); cg_ctx_destroy(ctx);
        return 0;
    } catch (const std::exception& e) { g_host_err = e.what(); if (ctx) cg_ctx_destroy(ctx); return 1; }
}
// Keccak256 transcript hook (tests): kinds[i] 0 = scalar (Fr), 1 = packed G1 point
int32_t cgh_plonk_transcript(int32_t curve, const int32_t* kinds, const uint64_t* const* payloads, int32_t n_items, uint64_t* out_challenge) {
    try {
        using namespace cgh;
        Curve c{curve};
        PlonkTranscript t(c);
        for (int i = 0; i < n_items; i++) { if (kinds[i] == 0) { Fr s; memcpy(s.v, payloads[i], 32); t.add_scalar(s); } else t.add_point((const uint8_t*)payloads[i]); }
        Fr r = t.get_challenge(); memcpy(out_challenge, r.v, 32);
        return 0;
    } catch (const std::exception& e) { g_host_err = e.what(); return 1; }
}
// ShamirHipProtocol x n (threshold t) through rounds 1..upto.  wit[i] / blind[i] = party i's Shamir shares of the private witness and of
// b_1..b_11; streams[i] = party i's private randomness.  Outputs as for cgh_plonk_prove_rep3, n parties.
int32_t cgh_plonk_prove_shamir(int32_t device, int32_t curve, const char* zkey_path, int32_t n, int32_t t, const uint64_t* pub_in, const uint64_t* const* wit,
                               const uint64_t* const* blind, const uint64_t* const* streams, size_t stream_len, int32_t upto,
                               uint64_t* out_commits, uint64_t* out_evals, uint64_t* out_challenges) {
    try {
        using namespace cgh;
        if (n < 3) throw std::runtime_error("Shamir protocol requires at least 3 parties");
        PlonkZKey z = read_plonk_zkey(curve, zkey_path);
        const Curve c = z.curve;
        const size_t n_priv = z.n_vars - z.n_additions - z.n_public - 1, psz = c.aff(CG_G1);
        std::vector<Fr> pub((const Fr*)pub_in, (const Fr*)pub_in + z.n_public + 1);
        memset(out_commits, 0, (size_t)n * 9 * psz); if (out_evals) memset(out_evals, 0, (size_t)n * 6 * 32); if (out_challenges) memset(out_challenges, 0, (size_t)n * 5 * 32);
        cg_ctx* ctx0 = nullptr;
        if (cg_ctx_create(device, &ctx0)) die("cg_ctx_create");
        cg_bases* tau = nullptr; CG(cg_bases_register(ctx0, c.id, CG_G1, z.p_tau.data(), z.domain_size + 6, psz, -1, &tau));
        if (validate_by_default()) { try { validate_bases(ctx0, tau, "p_tau"); } catch (...) { cg_bases_release(tau); throw; } }   // the zkey parser's per-point checks
        InProcShamirHub hub(n);
        std::vector<std::string> errs(n);
        std::vector<std::thread> th;
        for (int i = 0; i < n; i++) th.emplace_back([&, i] {
            cg_ctx* ctx = nullptr;
            try {
                if (cg_ctx_create(device, &ctx)) die("cg_ctx_create");
                InProcShamirNet net(&hub, i);
                {
                    HipDriver driver(ctx, c, Mode::Shamir, nullptr);
                    driver.rng1 = (const Fr*)streams[i]; driver.rng_len = stream_len;
                    driver.shamir_init(&net, t);
                    ShareVec w = driver.upload_vec((const Fr*)wit[i], nullptr, n_priv);
                    FieldShare b[11]; for (int q = 0; q < 11; q++) { memcpy(b[q].c[0].v, blind[i] + 4 * q, 32); b[q].c[1] = b[q].c[0]; }
                    plonk_run(driver, z, tau, pub, w, b, upto, PlonkOut{(uint64_t*)((uint8_t*)out_commits + (size_t)i * 9 * psz), out_challenges ? out_challenges + i * 20 : nullptr,
                                                                        out_evals ? out_evals + i * 24 : nullptr, nullptr, nullptr});
                    driver.free_vec(w);
                }
                cg_ctx_destroy(ctx);
            } catch (const std::exception& e) { errs[i] = e.what(); hub.abort(); if (ctx) cg_ctx_destroy(ctx); }
        });
        for (auto& x : th) x.join();
        cg_bases_release(tau);
        cg_ctx_destroy(ctx0);
        if (report_party_errors(errs, n)) return 1;
        return 0;
    } catch (const std::exception& e) { g_host_err = e.what(); return 1; }
}
// Rep3HipProtocol x 3 (three threads, in-process network) through rounds 1..upto.  blind_a[i] / blind_b[i] = party i's (a, b) shares of
// b_1..b_11; streams[i] = S_i (party i: rng1 = S_i, rng2 = S_{i-1}; rounds 2 and 3 consume masks and random shares).
// out_commits = 3 parties x 9 packed G1, out_evals = 3 x 6 Fr, out_challenges = 3 x 5 Fr (every party must report the same values)
int32_t cgh_plonk_prove_rep3(int32_t device, int32_t curve, const char* zkey_path, const uint64_t* pub_in, const uint64_t* const* wit_a, const uint64_t* const* wit_b,
                             const uint64_t* const* blind_a, const uint64_t* const* blind_b, const uint64_t* const* streams, size_t stream_len, int32_t upto,
                             uint64_t* out_commits, uint64_t* out_evals, uint64_t* out_challenges) {
    try {
        using namespace cgh;
        PlonkZKey z = read_plonk_zkey(curve, zkey_path);
        const Curve c = z.curve;
        const size_t n_priv = z.n_vars - z.n_additions - z.n_public - 1, psz = c.aff(CG_G1);
        std::vector<Fr> pub((const Fr*)pub_in, (const Fr*)pub_in + z.n_public + 1);
        memset(out_commits, 0, 3 * 9 * psz); if (out_evals) memset(out_evals, 0, 3 * 6 * 32); if (out_challenges) memset(out_challenges, 0, 3 * 5 * 32);
        cg_ctx* ctx0 = nullptr;
        if (cg_ctx_create(device, &ctx0)) die("cg_ctx_create");
        cg_bases* tau = nullptr; CG(cg_bases_register(ctx0, c.id, CG_G1, z.p_tau.data(), z.domain_size + 6, psz, -1, &tau));
        if (validate_by_default()) { try { validate_bases(ctx0, tau, "p_tau"); } catch (...) { cg_bases_release(tau); throw; } }   // the zkey parser's per-point checks
        InProcHub hub;
        std::string errs[3];
        std::vector<std::thread> th;
        for (int i = 0; i < 3; i++) th.emplace_back([&, i] {
            cg_ctx* ctx = nullptr;
            try {
                if (cg_ctx_create(device, &ctx)) die("cg_ctx_create");
                InProcNetwork net(&hub, i);
                {
                    HipDriver driver(ctx, c, Mode::Rep3, &net);
                    if (streams) { driver.rng1 = (const Fr*)streams[i]; driver.rng2 = (const Fr*)streams[(i + 2) % 3]; driver.rng_len = stream_len; }
                    ShareVec wit = driver.upload_vec((const Fr*)wit_a[i], (const Fr*)wit_b[i], n_priv);
                    FieldShare b[11]; for (int t = 0; t < 11; t++) { memcpy(b[t].c[0].v, blind_a[i] + 4 * t, 32); memcpy(b[t].c[1].v, blind_b[i] + 4 * t, 32); }
                    plonk_run(driver, z, tau, pub, wit, b, upto, PlonkOut{(uint64_t*)((uint8_t*)out_commits + (size_t)i * 9 * psz), out_challenges ? out_challenges + i * 20 : nullptr,
                                                                          out_evals ? out_evals + i * 24 : nullptr, nullptr, nullptr});
                    driver.free_vec(wit);
                }
                cg_ctx_destroy(ctx);
            } catch (const std::exception& e) { errs[i] = e.what(); hub.abort(); if (ctx) cg_ctx_destroy(ctx); }
        });
        for (auto& t : th) t.join();
        cg_bases_release(tau);
        cg_ctx_destroy(ctx0);
        if (report_party_errors(errs, 3)) return 1;
        return 0;
    } catch (const std::exception& e) { g_host_err = e.what(); return 1; }
}


// ONE REP3 party of co-plonk with the caller's network and correlated randomness (co-circom.rs:560-600 builds a Rep3Protocol and hands it to
// CoPlonk::prove; the tables are those of cgh_session_prove_rep3_party).  blind_a / blind_b = this party's shares of b_1..b_11, or both NULL to
// draw them with rand() in the reference's order (round1.rs:93-99) before anything else.  Outputs: 9 packed G1, 6 evaluations, 5 challenges.
int32_t cgh_plonk_prove_rep3_party(int32_t device, int32_t curve, const char* zkey_path, const uint64_t* pub_in, const uint64_t* wit_a, const uint64_t* wit_b,
                                   const uint64_t* blind_a, const uint64_t* blind_b, const cgh_rep3_net* net_cb, const cgh_rep3_rand* rnd_cb, int32_t upto,
                                   uint64_t* out_commits, uint64_t* out_evals, uint64_t* out_challenges) {
    return cgh_plonk_prove_rep3_party_ex(device, curve, zkey_path, pub_in, wit_a, wit_b, blind_a, blind_b, net_cb, rnd_cb, nullptr, upto, out_commits, out_evals, out_challenges);
}
// streams_cb != NULL: the masking vectors of the rounds' mul_vec calls (round2.rs / round3.rs) are drawn on the GPU from the described generators
int32_t cgh_plonk_prove_rep3_party_ex(int32_t device, int32_t curve, const char* zkey_path, const uint64_t* pub_in, const uint64_t* wit_a, const uint64_t* wit_b,
                                      const uint64_t* blind_a, const uint64_t* blind_b, const cgh_rep3_net* net_cb, const cgh_rep3_rand* rnd_cb,
                                      const cgh_rep3_chacha* streams_cb, int32_t upto, uint64_t* out_commits, uint64_t* out_evals, uint64_t* out_challenges) {
    cg_ctx* ctx = nullptr; cg_bases* tau = nullptr;
    try {
        using namespace cgh;
        if (!zkey_path || !pub_in || !wit_a || !wit_b || !net_cb || !rnd_cb || !out_commits) throw std::runtime_error("cgh_plonk_prove_rep3_party: null argument");
        if ((blind_a == nullptr) != (blind_b == nullptr)) throw std::runtime_error("cgh_plonk_prove_rep3_party: blind_a and blind_b go together");
        if (upto < 1 || upto > 5) throw std::runtime_error("cgh_plonk_prove_rep3_party: upto must be 1..5");
        PlonkZKey z = read_plonk_zkey(curve, zkey_path);
        const Curve c = z.curve;
        const size_t n_priv = z.n_vars - z.n_additions - z.n_public - 1, psz = c.aff(CG_G1);
        std::vector<Fr> pub((const Fr*)pub_in, (const Fr*)pub_in + z.n_public + 1);
        memset(out_commits, 0, 9 * psz); if (out_evals) memset(out_evals, 0, 6 * 32); if (out_challenges) memset(out_challenges, 0, 5 * 32);
        if (cg_ctx_create(device, &ctx)) die("cg_ctx_create");
        CG(cg_bases_register(ctx, c.id, CG_G1, z.p_tau.data(), z.domain_size + 6, psz, -1, &tau));
        if (validate_by_default()) validate_bases(ctx, tau, "p_tau");
        CallbackNetwork net(*net_cb);
        CallbackRand rnd(*rnd_cb);
        rnd.describe_streams(streams_cb);
        {
            HipDriver driver(ctx, c, Mode::Rep3, &net);
            driver.rsrc = &rnd;
            FieldShare b[11];
            for (int t = 0; t < 11; t++) {
                if (blind_a) { memcpy(b[t].c[0].v, blind_a + 4 * t, 32); memcpy(b[t].c[1].v, blind_b + 4 * t, 32); }
                else b[t] = driver.rand();
            }
            ShareVec wit = driver.upload_vec((const Fr*)wit_a, (const Fr*)wit_b, n_priv);
            try { plonk_run(driver, z, tau, pub, wit, b, upto, PlonkOut{out_commits, out_challenges, out_evals, nullptr, nullptr}); }
            catch (...) { driver.free_vec(wit); throw; }
            driver.free_vec(wit);
            rnd.settle();
        }
        cg_bases_release(tau);
        cg_ctx_destroy(ctx);
        return 0;
    } catch (const std::exception& e) { g_host_err = e.what(); if (tau) cg_bases_release(tau); if (ctx) cg_ctx_destroy(ctx); return 1; }
}

}  // extern "C"
