// C entry points: one-shot Groth16 proofs, zkey file -> proof (co-circom.rs:482-506)
#include "groth16.hpp"
#include "capi_common.hpp"

extern "C" {

// ShamirHipProtocol x n (n threads, in-process any-to-any network), threshold t.  wit[i] = party i's Shamir shares of the private
// witness; streams[i] = party i's private randomness (consumed in the order the reference draws values).  out_proofs = n proofs.
int32_t cgh_prove_shamir(int32_t device, int32_t curve, const char* zkey_path, int32_t n, int32_t t, const uint64_t* pub_in, const uint64_t* const* wit,
                         const uint64_t* const* streams, size_t stream_len, size_t preprocess, uint64_t* out_proofs, uint64_t* out_h) {
    try {
        using namespace cgh;
        if (n < 3) throw std::runtime_error("Shamir protocol requires at least 3 parties");        // shamir/network.rs:75-77
        SecondContexts second(device, zkey_path, n);
        ZKey z = read_zkey(curve, zkey_path);
        const size_t n_aux = z.n_vars - z.n_public - 1;
        std::vector<Fr> pub((const Fr*)pub_in, (const Fr*)pub_in + z.n_public + 1);
        cg_ctx* ctx0 = nullptr;
        if (cg_ctx_create(device, &ctx0)) die("cg_ctx_create");
        DeviceZKey dz = upload_zkey(ctx0, z, pub);
        second.ready();
        InProcShamirHub hub(n);
        const size_t psz = 8 * z.curve.fq();
        std::vector<std::string> errs(n);
        std::vector<std::thread> th;
        for (int i = 0; i < n; i++) th.emplace_back([&, i] {
            cg_ctx* ctx = nullptr;
            try {
                if (cg_ctx_create(device, &ctx)) die("cg_ctx_create");
                InProcShamirNet net(&hub, i);
                HipDriver driver(ctx, z.curve, Mode::Shamir, nullptr);
                driver.use_second_context(second.take(i));
                driver.rng1 = (const Fr*)streams[i]; driver.rng_len = stream_len;
                driver.shamir_init(&net, t);
                const auto ta = std::chrono::steady_clock::now();
                driver.preprocess(preprocess);                                            // 0 = the reference's lazy batches of 1024
                const auto tb = std::chrono::steady_clock::now();
                ShareVec w = driver.upload_vec((const Fr*)wit[i], nullptr, n_aux);
                CoGroth16 prover(driver);
                ShareVec h;
                Proof p = prover.prove(dz, pub, w, nullptr, &h);
                if (i == 0 && getenv("CGH_TIMING"))
                    fprintf(stderr, "cgh_prove_shamir party 0: preprocess %.1f ms, prove %.1f ms\n", std::chrono::duration<double, std::milli>(tb - ta).count(),
                            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tb).count());
                store_proof(p, (uint8_t*)out_proofs + i * psz);
                if (out_h && i == 0) CG(cg_dev_download(ctx, out_h, h.c[0], h.n * 32));
                driver.free_vec(h); driver.free_vec(w); driver.shutdown();
                cg_ctx_destroy(ctx);
            } catch (const std::exception& e) { errs[i] = e.what(); hub.abort(); if (ctx) cg_ctx_destroy(ctx); }
        });
        for (auto& x : th) x.join();
        release_zkey(ctx0, dz);
        cg_ctx_destroy(ctx0);
        if (report_party_errors(errs, n)) return 1;
        return 0;
    } catch (const std::exception& e) { g_host_err = e.what(); return 1; }
}
// PlainHipDriver: full_witness = n_vars Montgomery elements; proof = A || B || C packed affine
int32_t cgh_prove_plain(int32_t device, int32_t curve, const char* zkey_path, const uint64_t* full_witness, const uint64_t* r, const uint64_t* s, uint64_t* out_proof, uint64_t* out_h) {
    try {
        using namespace cgh;
        const bool timing = getenv("CGH_TIMING") != nullptr;
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        const auto t0 = now();
        SecondContexts second(device, zkey_path, 1);
        ZKey z = read_zkey(curve, zkey_path);
        const auto t1 = now();
        CtxGuard cg;                                           // destroyed last: everything below lives on it
        if (cg_ctx_create(device, &cg.ctx)) die("cg_ctx_create");
        cg_ctx* ctx = cg.ctx;
        const Fr* w = (const Fr*)full_witness;
        std::vector<Fr> pub(w, w + z.n_public + 1);
        DeviceZKeyGuard dzg(ctx, upload_zkey(ctx, z, pub));
        const auto t2 = now();
        std::chrono::steady_clock::time_point t3;
        {
            HipDriver driver(ctx, z.curve, Mode::Plain, nullptr);
            driver.use_second_context(second.take(0));
            VecGuard wit(driver, driver.upload_vec(w + z.n_public + 1, nullptr, z.n_vars - z.n_public - 1)), h(driver);
            FieldShare rs[2]; memcpy(rs[0].c[0].v, r, 32); rs[0].c[1] = rs[0].c[0]; memcpy(rs[1].c[0].v, s, 32); rs[1].c[1] = rs[1].c[0];
            CoGroth16 prover(driver);
            Proof p = prover.prove(dzg.dz, pub, wit.v, rs, &h.v);
            t3 = now();
            store_proof(p, (uint8_t*)out_proof);
            if (out_h) CG(cg_dev_download(ctx, out_h, h.v.c[0], h.v.n * 32));
        }
        if (timing) fprintf(stderr, "cgh_prove_plain: read+decode zkey %.1f ms, context + upload%s %.1f ms, prove %.1f ms, teardown %.1f ms\n", ms(t0, t1),
                            validate_by_default() ? " + point validation" : "", ms(t1, t2), ms(t2, t3), ms(t3, now()));
        return 0;
    } catch (const std::exception& e) { g_host_err = e.what(); return 1; }
}
// Rep3HipProtocol x 3 on three threads over the in-process network; streams[i] = S_i (party i: rng1 = S_i, rng2 = S_{i-1})
int32_t cgh_prove_rep3(int32_t device, int32_t curve, const char* zkey_path, const uint64_t* pub_in, const uint64_t* const* wit_a, const uint64_t* const* wit_b,
                       const uint64_t* const* streams, size_t stream_len, uint64_t* out_proofs, uint64_t* out_h) {
    try {
        using namespace cgh;
        SecondContexts second(device, zkey_path, 3);
        ZKey z = read_zkey(curve, zkey_path);
        const size_t n_aux = z.n_vars - z.n_public - 1;
        std::vector<Fr> pub((const Fr*)pub_in, (const Fr*)pub_in + z.n_public + 1);
        CtxGuard cg0;
        if (cg_ctx_create(device, &cg0.ctx)) die("cg_ctx_create");
        DeviceZKeyGuard dzg(cg0.ctx, upload_zkey(cg0.ctx, z, pub));   // one device-resident zkey shared by the three co-located parties
        const DeviceZKey& dz = dzg.dz;
        second.ready();
        InProcHub hub;
        const size_t psz = 8 * z.curve.fq();
        std::string errs[3];
        std::vector<std::thread> th;
        for (int i = 0; i < 3; i++) th.emplace_back([&, i] {
            try {
                CtxGuard cg;
                if (cg_ctx_create(device, &cg.ctx)) die("cg_ctx_create");
                InProcNetwork net(&hub, i);
                HipDriver driver(cg.ctx, z.curve, Mode::Rep3, &net);
                driver.use_second_context(second.take(i));
                driver.rng1 = (const Fr*)streams[i]; driver.rng2 = (const Fr*)streams[(i + 2) % 3]; driver.rng_len = stream_len;
                VecGuard wit(driver, driver.upload_vec((const Fr*)wit_a[i], (const Fr*)wit_b[i], n_aux)), h(driver);
                CoGroth16 prover(driver);
                Proof p = prover.prove(dz, pub, wit.v, nullptr, &h.v);
                store_proof(p, (uint8_t*)out_proofs + i * psz);
                if (out_h && i == 0) { CG(cg_dev_download(cg.ctx, out_h, h.v.c[0], h.v.n * 32)); CG(cg_dev_download(cg.ctx, out_h + h.v.n * 4, h.v.c[1], h.v.n * 32)); }
            } catch (const std::exception& e) { errs[i] = e.what(); hub.abort(); }
        });
        for (auto& t : th) t.join();
        if (report_party_errors(errs, 3)) return 1;
        return 0;
    } catch (const std::exception& e) { g_host_err = e.what(); return 1; }
}

}  // extern "C"
