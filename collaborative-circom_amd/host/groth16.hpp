// CoGroth16::prove (co-circom/co-groth16/src/groth16.rs:113-326) over HipDriver, zkey -> device tables, scope guards of the entry points
#pragma once
#include "driver.hpp"
#include "multidev.hpp"

namespace cgh {

// ---- prover --------------------------------------------------------------------------------------------------------------
struct VecGuard {   // device share vector released when the entry point leaves, however it leaves
    HipDriver& d; ShareVec v;
    explicit VecGuard(HipDriver& drv) : d(drv) {}
    VecGuard(HipDriver& drv, ShareVec x) : d(drv), v(x) {}
    // Leaving by an exception, kernels of the party's OTHER contexts (the witness-independent MSMs on the second context, the slices on
    // further GPUs) may still read the vector: cg_dev_free parks a block behind the RELEASING context's streams only, so those contexts
    // are drained first — on the regular path every MSM has been collected before the guard runs.
    // (the block itself joins the driver's list of deferred releases: one release mark for everything a proof gives back, see HipDriver::shutdown)
    ~VecGuard() { try { if (std::uncaught_exceptions() > 0) d.sync_other_contexts(); d.defer_vec(v); } catch (...) {} }
    VecGuard(const VecGuard&) = delete; VecGuard& operator=(const VecGuard&) = delete;
};
struct Proof { Bytes a, b, c; };   // packed affine, (0,0) = infinity  (Groth16Proof, groth16/proof.rs:8-29)

class CoGroth16 {
public:
    HipDriver& driver;
    explicit CoGroth16(HipDriver& d) : driver(d) {}
    bool additive() const { return driver.additive_h && driver.mode != Mode::Plain; }

    // groth16.rs:141-204
    // after_first_launches (optional): called once the constraint evaluations, the first local product and the transforms of a and b are
    // enqueued, before the host waits for the first exchange (prove() starts the witness-independent MSMs there on small circuits)
    ShareVec witness_map_from_matrices(const DeviceZKey& dz, const std::vector<Fr>& public_inputs, const ShareVec& private_witness, const std::function<void()>& after_first_launches = nullptr) {
        const ZKey& z = *dz.z;
        const size_t num_inputs = z.n_public + 1, num_constraints = z.num_constraints;
        const Domain dom = groth16_domain(driver.curve, z.pow, num_constraints, num_inputs);          // :150-153
        HipDriver::Marks mk("witness_map party 0", driver.party() <= 0);
        ShareVec a = driver.evaluate_constraints(dz.mat[0], dz.pub_dev, (uint32_t)num_inputs, private_witness, dom.m);   // :156-166
        ShareVec b = driver.evaluate_constraints(dz.mat[1], dz.pub_dev, (uint32_t)num_inputs, private_witness, dom.m);
        driver.clone_public_into(a, num_constraints, public_inputs, dz.pub_dev);                       // :168-171
        // The two mul_vec exchanges (:174, :190) run under the transforms that do not depend on them: the local product is started,
        // the independent NTTs are enqueued, then the party-to-party exchange proceeds while the GPU works (values as in the reference).
        if (driver.prefetched.empty()) driver.prefetch_masks(2, dom.m);                                // :174 and :190 draw next to each other
        mk.mark("spmv enqueue");
        if (additive()) {
            // Shamir: the same with degree-2t sharings — products of degree-t shares, linear transforms, a difference — reduced to degree t
            // as ONE point after the MSM (degree_reduce_point, shamir.rs:386-436) instead of 2 x m field elements before it.
            // Additive-quotient variant: neither product is re-shared.  The masked local products (rep3.rs:655-660, same masks in the same
            // order) are additive shares of a*b; the transforms are linear; so h_i = ab_i - c_i sums to the reference's h, and the only reader
            // of h is the MSM against h_query.  The 2 x 32 B x m exchange of :174 and :190 and the second component of c and h disappear.
            auto c_local = driver.mul_vec_begin(a, b, false);
            driver.ifft_coset_fft_in_place2(a, b, dom.omega, dom.coset_g);
            ShareVec c = c_local.out;
            { HipDriver::Components own(driver, 1); driver.ifft_coset_fft_in_place(c, dom.omega, dom.coset_g); }
            ShareVec ab = driver.mul_vec_begin(a, b, false).out;
            { HipDriver::Components own(driver, 1); driver.sub_assign_vec(ab, c); }
            mk.mark("products + ntt enqueue (additive)");
            driver.defer_vec(a); driver.defer_vec(b); driver.defer_vec(c);      // released by prove() once the quotient's MSM is enqueued
            return ab;
        }
        auto c_pending = driver.mul_vec_begin(a, b);                                                   // :174
        mk.mark("mul_vec_begin");
        driver.ifft_coset_fft_in_place2(a, b, dom.omega, dom.coset_g);                                 // :175-188: both vectors, both components, one sequence of launches
        mk.mark("ntt enqueue");
        if (after_first_launches) { after_first_launches(); mk.mark("aux msm enqueued (late)"); }
        ShareVec c = driver.mul_vec_finish(c_pending);
        mk.mark("mul_vec_finish");
        auto ab_pending = driver.mul_vec_begin(a, b);                                                  // :190
        mk.mark("mul_vec_begin");
        driver.ifft_coset_fft_in_place(c, dom.omega, dom.coset_g);                                     // :194-200
        mk.mark("ntt enqueue");
        ShareVec ab = driver.mul_vec_finish(ab_pending);
        mk.mark("mul_vec_finish");
        driver.sub_assign_vec(ab, c);                                                                  // :202
        // a, b, c and the mask buffers are released by prove() once the quotient's MSM is enqueued: ~10 cg_dev_free calls (an event on every
        // stream of the context each) used to sit between the last subtraction and that MSM's first kernel — 0.45 ms of a 2^16 party
        driver.defer_vec(a); driver.defer_vec(b); driver.defer_vec(c);
        return ab;
    }

    // groth16.rs:206-235
    // priv_acc = msm_public_points(&query[1 + pub_len..], aux_assignment) (:221), started before the witness map (see prove)
    PointShare calculate_coeff(PointShare initial, const View& query_host, int group, const Bytes& vk_param,
                               const std::vector<Fr>& input_assignment, const PointShare& priv_acc, const std::vector<FixedTable>* pub_tabs = nullptr) {
        const Curve& c = driver.curve;
        const size_t pub_len = input_assignment.size(), rec = c.aff(group);
        Point pub_acc = pt_inf(c, group);                                                              // :220 (tiny, plain scalars)
        for (size_t i = 0; i < pub_len; i++)
            pub_acc = pt_add(c, pub_acc, pt_mul_fixed(c, pub_tabs && i < pub_tabs->size() ? (*pub_tabs)[i].t : nullptr, pt_from_affine(c, group, query_host.data() + (1 + i) * rec), input_assignment[i]));
        PointShare res = initial;
        driver.add_assign_points_public(res, pt_from_affine(c, group, query_host.data()));             // :227
        driver.add_assign_points_public(res, pt_from_affine(c, group, vk_param.data()));               // :228
        driver.add_assign_points_public(res, pub_acc);                                                 // :229
        driver.add_assign_points(res, priv_acc);                                                       // :230
        return res;
    }

    // groth16.rs:113-139 + :237-326
    Proof prove(const DeviceZKey& dz, const std::vector<Fr>& public_inputs, const ShareVec& private_witness, const FieldShare* rs_plain, ShareVec* h_out = nullptr) {
        const ZKey& z = *dz.z; const Curve& c = driver.curve;
        HipDriver::Marks mk("prove party 0", driver.party() <= 0);
        std::vector<Fr> input_assignment(public_inputs.begin() + 1, public_inputs.end());
        const size_t first_aux = 1 + input_assignment.size();
        // a (:267 -> :221), b1 (:284), b2 (:298), l (:251): one call, one scalar schedule, on the second context.  Launch order within a
        // share component: b2 (G2 first, HipDriver::begin_multi_ordered), a, b1, l — the order in which the results are consumed below, so
        // that the host's work on a result (public-input terms, openings, scalar multiplications) runs under the accumulation of the next
        enum { AUX_A = 0, AUX_B1 = 1, AUX_B2 = 2, AUX_L = 3 };
        // Additive-quotient variant (HipDriver::additive_h, REP3, opt-in): every MSM multiplies the party's OWN component only — the second
        // component of an MSM result is by definition the previous party's first (rep3.rs:934-947 on the pair (x_i, x_{i-1})) — and the five
        // results become replicated shares again in ONE round of five points (reshare_points).  From there on every value and every message
        // is the reference's, and so is the proof.  What changes on the wire: the two vector exchanges of the witness map are gone, one
        // 384-byte (BN254) message is added; all three parties must run the variant.
        const bool add_h = additive();
        std::unique_ptr<HipDriver::Components> own(add_h ? new HipDriver::Components(driver, 1) : nullptr);   // (Shamir has one component anyway)
        // Several GPUs: the witness goes up by rows, the witness map itself is spread over the devices (multidev.hpp) and every device
        // multiplies its own rows of the witness and of h
        const bool distributed = DistributedWitnessMap::usable(driver, dz);
        std::unique_ptr<DistributedWitnessMap> dmap;
        HipDriver::PendingMsm aux_msm;
        if (distributed) {
            dmap.reset(new DistributedWitnessMap(driver, dz, *driver.md));
            const bool from_host = private_witness.c[0] == nullptr;                                   // the entry left the vectors on the host (driver.host_wit)
            if (from_host && !driver.host_wit[0]) throw std::runtime_error("multi-device proof: no witness given");
            dmap->place_witness(driver.host_wit[0], driver.host_wit[1], from_host ? nullptr : &private_witness, private_witness.n, public_inputs);
            aux_msm = dmap->begin_aux_msms();
            dmap->gather_witness(from_host);
        }
        // Small REP3 circuits (the two mul_vec exchanges are single synchronous messages): the witness map is a chain of short kernels and
        // two host round trips, and kernels of another stream only get onto the chip as the accumulations' workgroups retire (a 2^16 party:
        // the chain's first kernel waited 1.3 ms for a slot).  The chain's first leg — constraint rows, first product, transforms of a and
        // b — is therefore enqueued FIRST, on an idle chip, and the witness-independent MSMs right behind it, while the host would wait
        // for the first exchange anyway.  MSMs involve no network: the message order is untouched.
        static const bool late_knob = tune_env("CGH_LATE_AUX") != nullptr;                              // A/B knob; measured SLOWER on MI355X (2^16: 3.68 -> 3.86 ms, 2^14: 2.73 -> 2.94, profiles/r05_small_circuit_ab2.txt): off
        const bool late_aux = late_knob && !distributed && !dz.sliced && !add_h && driver.mode == Mode::Rep3 && private_witness.n < driver.XCHG_ASYNC_MIN;
        auto begin_aux = [&] {
            if (!dz.sliced && !add_h && driver.aux && private_witness.up_ctx && private_witness.up_first >= 0)      // a large witness still on its way up: first table in pieces
                aux_msm = driver.msm_begin_aux_split({dz.a, dz.b1, dz.b2, dz.l}, {first_aux, first_aux, first_aux, 0}, {CG_G1, CG_G1, CG_G2, CG_G1}, private_witness.n, private_witness, AUX_A);
            else
            aux_msm = dz.sliced ? driver.msm_begin_sharded(dz, true, private_witness)
                                : driver.msm_begin_multi({dz.a, dz.b1, dz.b2, dz.l}, {first_aux, first_aux, first_aux, 0}, {CG_G1, CG_G1, CG_G2, CG_G1}, private_witness.n, private_witness, true);
        };
        if (!distributed && !late_aux) begin_aux();
        own.reset();
        mk.mark("aux msm enqueued");
        struct HParts : DistributedH {     // the rows of h on their devices: released when prove leaves, however it leaves (after the map's own buffers)
            ~HParts() { for (auto& part : parts) cg_dev_free_many(part.ctx, part.h.c, 2); }
        } dh;
        VecGuard hg(driver);               // the quotient vector is released when prove leaves, however it leaves (a failing network round, invalid data)
        ShareVec& h = hg.v;
        HipDriver::PendingMsm h_msm;
        if (distributed) {
            if (h_out) throw std::runtime_error("the quotient vector of a multi-device proof stays distributed (h_out is a single-device option)");
            static_cast<DistributedH&>(dh) = dmap->run(dz, public_inputs);
            mk.mark("witness map (distributed)");
            const int hk = add_h ? 1 : driver.k();                                                     // the variant's h has one component
            h_msm.on = driver.ctx; h_msm.groups = {CG_G1}; h_msm.tickets.resize(1); h_msm.k = hk;
            for (size_t d = 0; d < dh.parts.size(); d++) {                                             // :248, rows of device d against its slice of h_query
                const DistributedH::Part& part = dh.parts[d];
                if (d && dmap->skip(d)) continue;
                const cg_bases* tab = dmap->devs[d].dz->h; const size_t off0 = 0;
                const void* sc[2] = {part.h.c[0], part.h.c[1]};
                if (d == 0) CG(cg_msm_dev_begin_multi(part.ctx, 1, &tab, &off0, dmap->skip(0) ? 0 : part.h.n, sc, hk, h_msm.tickets.data()));
                else {
                    HipDriver::PendingMsm::Part p{part.ctx, std::vector<int32_t>(1), {nullptr, nullptr}};
                    CG(cg_msm_dev_begin_multi(part.ctx, 1, &tab, &off0, part.h.n, sc, hk, p.tickets.data()));
                    h_msm.parts.push_back(p);
                }
            }
        } else {
        // the masks of the witness map's two mul_vec calls (:174, :190) start their way to the device now: behind the witness shares and the
        // few small synchronous uploads of the MSM set-up (the copy engine serves its requests in order), ahead of everything else
        if (driver.prefetched.empty()) driver.prefetch_masks(2, groth16_domain(c, z.pow, z.num_constraints, public_inputs.size()).m);   // (a party entry draws them before its shares have arrived)
        mk.mark("mask uploads enqueued");
        h = witness_map_from_matrices(dz, public_inputs, private_witness, late_aux ? std::function<void()>(begin_aux) : nullptr);
        mk.mark("witness map");
        if (add_h) own.reset(new HipDriver::Components(driver, 1));
        h_msm = dz.sliced ? driver.msm_begin_sharded(dz, false, h) : driver.msm_begin_multi({dz.h}, {0}, {CG_G1}, h.n, h, false);   // :248
        own.reset();
        if (h.n >= driver.XCHG_ASYNC_MIN) driver.free_deferred();     // large circuits give the witness map's vectors back now (GBs); small ones with everything else at the end: one release mark per proof
        mk.mark("h msm enqueued, vectors released");
        }
        FieldShare r = rs_plain ? rs_plain[0] : driver.rand();                                         // :134-135
        FieldShare s = rs_plain ? rs_plain[1] : driver.rand();
        // The GPU is busy with the MSMs for tens of milliseconds from here on.  Everything of :258-297 that does not read an MSM result
        // — the product r*s (one network round) and the five multiplications of public points by r, s, r*s (254-bit scalar
        // multiplications on the host) — is done while it works, and the MSM results are then collected in the order the aux context
        // completes them (a, b1, b2, l) with h, which starts last, at the end.  Messages between the parties keep the reference's order:
        // mul (:258), open_point (:276), scalar_mul (:291), open_two_points (:316); MSMs involve no network.
        const Point delta_g1 = pt_from_affine(c, CG_G1, z.delta_g1.data());
        const Point delta_g2 = pt_from_affine(c, CG_G2, z.delta_g2.data());
        FieldShare rs = driver.mul(r, s);                                                              // :258
        const SessionFixed* fx = dz.fixed;                                                             // a session's window tables (else variable-base products)
        // (the G2 product costs as much as the three G1 products together: it runs on a helper thread beside them)
        auto s_g2_pending = Helpers::get().run([&] { return driver.scalar_mul_public_point(delta_g2, s, fx ? fx->delta_g2.t : nullptr); });   // :297
        struct Joined { std::future<PointShare>& f; ~Joined() { if (f.valid()) f.wait(); } } s_g2_joined{s_g2_pending};   // (the helper reads this frame: never left behind)
        PointShare r_s_delta_g1 = driver.scalar_mul_public_point(delta_g1, rs, fx ? fx->delta_g1.t : nullptr);   // :259
        PointShare r_g1 = driver.scalar_mul_public_point(delta_g1, r, fx ? fx->delta_g1.t : nullptr);  // :265
        PointShare s_g1 = driver.scalar_mul_public_point(delta_g1, s, fx ? fx->delta_g1.t : nullptr);  // :283
        PointShare s_g2 = s_g2_pending.get();
        mk.mark("scalar steps under the msms");
        PointShare early[5]; bool have_early = false;
        if (add_h) {                                                                                   // all five results, then the one re-sharing round
            for (int i : {AUX_A, AUX_B1, AUX_B2, AUX_L}) early[i] = driver.msm_finish(aux_msm, i);
            early[4] = driver.msm_finish(h_msm, 0);
            if (driver.mode == Mode::Rep3) driver.reshare_points({&early[0], &early[1], &early[2], &early[3], &early[4]});
            else early[4].c[0] = driver.degree_reduce_point(early[4].c[0]);                            // Shamir: h was a degree-2t sharing
            have_early = true;
            mk.mark("msms + reshare (additive)");
        }
        auto aux_result = [&](int i) { return have_early ? early[i] : driver.msm_finish(aux_msm, i); };
        PointShare g_a = calculate_coeff(r_g1, z.a_query, CG_G1, z.alpha_g1, input_assignment, aux_result(AUX_A), fx ? &fx->a_pub : nullptr);   // :267
        mk.mark("msm a");
        Point g_a_opened = driver.open_point(g_a);                                                     // :276
        mk.mark("open a");
        PointShare s_g_a = driver.scalar_mul_public_point(g_a_opened, s);                              // :277
        mk.mark("s * a");
        PointShare g1_b = calculate_coeff(s_g1, z.b_g1_query, CG_G1, z.beta_g1, input_assignment, aux_result(AUX_B1), fx ? &fx->b1_pub : nullptr);   // :284
        mk.mark("msm b1");
        PointShare r_g1_b = driver.scalar_mul(g1_b, r);                                                // :291
        mk.mark("r * b1");
        PointShare g2_b = calculate_coeff(s_g2, z.b_g2_query, CG_G2, z.beta_g2, input_assignment, aux_result(AUX_B2), fx ? &fx->b2_pub : nullptr);   // :298
        mk.mark("msm b2");
        PointShare l_aux_acc = aux_result(AUX_L);                                          // :251
        PointShare h_acc = have_early ? early[4] : driver.msm_finish(h_msm, 0);                                               // :248
        mk.mark("msm l + h");
        if (dmap) dmap->verify_received();
        driver.verify_received_vectors();      // rep3.rs:663-669: what deserialising the two mul_vec messages would have refused (counted on the device, read here where the stream is idle)
        PointShare g_c = s_g_a;                                                                        // :308-312
        driver.add_assign_points(g_c, r_g1_b);
        driver.sub_assign_points(g_c, r_s_delta_g1);
        driver.add_assign_points(g_c, l_aux_acc);
        driver.add_assign_points(g_c, h_acc);
        auto opened = driver.open_two_points(g_c, g2_b);                                               // :316
        mk.mark("open");
        driver.msm_release(aux_msm); driver.msm_release(h_msm);
        dmap.reset();
        if (h_out) { *h_out = h; h = ShareVec(); }                                                      // handed to the caller
        return Proof{pt_to_affine(c, g_a_opened), pt_to_affine(c, opened.second), pt_to_affine(c, opened.first)};   // :319-325
    }
};

// The reference's parser validates every point while decoding (circom-types/src/traits.rs:107-155: is_on_curve, then
// is_in_correct_subgroup_assuming_on_curve; failure = SerializationError::InvalidData).  Here the packed sections go to the device
// as they are and the same two predicates run there, one pass per table.
static void validate_bases(cg_ctx* ctx, const cg_bases* b, const char* name) {
    uint64_t bad = 0, first = 0;
    CG(cg_bases_check_on_curve(ctx, b, &bad, &first));
    if (bad) throw std::runtime_error(std::string("invalid data: ") + name + "[" + std::to_string(first) + "] is not on the curve (" + std::to_string(bad) + " bad points)");
    CG(cg_bases_check_subgroup(ctx, b, &bad, &first));
    if (bad) throw std::runtime_error(std::string("invalid data: ") + name + "[" + std::to_string(first) + "] is not in the correct subgroup (" + std::to_string(bad) + " bad points)");
}

// The reference validates every zkey point while parsing (traits.rs:116-123, 147-153), so the prove entry points and
// cgh_session_open do too, by default.  Opt-out for callers that validated the file before (cgh_zkey_validate): the environment
// variable CGH_SKIP_ZKEY_VALIDATION or cgh_set_zkey_validation(0).
inline std::atomic<int> g_validate_zkey{-1};
static bool validate_by_default() {
    int v = g_validate_zkey.load();
    if (v < 0) { v = getenv("CGH_SKIP_ZKEY_VALIDATION") ? 0 : 1; g_validate_zkey.store(v); }
    return v != 0;
}
struct DeviceZKeyGuard;
static void release_zkey(cg_ctx* ctx, DeviceZKey& d);
// Slice `rank` of `world` of a range of n items.  Devices of a party are NOT equally loaded: the six vector pipelines of the witness map
// (iNTT -> coset shift -> NTT of a.a, a.b, b.a, b.b, c.a, c.b) run one per device on device (v + 1) mod world (DistributedWitnessMap::owner),
// each worth ~1 % of the proof's MSM work (emulated 8-device REP3 party at 2^22: owners 13.6-14.1 ms, the two devices without a pipeline
// 13.0-13.1, profiles/r06_multigpu_inlibrary_emulation.txt).  A device's share of every sliced range — its MSM table slices and its rows
// of the witness map — is therefore 1/world of (1 + PIPE * 6) minus PIPE per pipeline it owns (VERDICT r5 #4b).  Small ranges (fewer
// than 2^16 items) are cut evenly: sizes then differ by at most one and empty slices only appear when there are fewer items than devices.
static constexpr double PIPELINE_SHARE = 0.01;
static int pipelines_owned(int rank, int world) { int c = 0; for (int v = 0; v < 6; v++) c += (v + 1) % world == rank; return c; }
static std::pair<size_t, size_t> slice_of(size_t n, int rank, int world) {
    if (world > 1 && n >= ((size_t)1 << 16)) {
        auto upto = [&](int r) {                                                     // items of ranks 0 .. r-1
            double share = 0;
            for (int d = 0; d < r; d++) share += (1.0 + PIPELINE_SHARE * 6) / world - PIPELINE_SHARE * pipelines_owned(d, world);
            return r >= world ? n : std::min(n, (size_t)(share * (double)n) / 64 * 64);
        };
        return {upto(rank), upto(rank + 1)};
    }
    const size_t base = n / world, rem = n % world, lo = (size_t)rank * base + std::min<size_t>((size_t)rank, rem);
    return {lo, lo + base + ((size_t)rank < rem ? 1 : 0)};
}
// rank/world: this device's share of a party's GPUs (world == 1: the whole zkey).  Rank 0 also holds the constraint matrices (the
// witness map runs there); every rank holds slice `rank` of the five queries.
static DeviceZKey upload_zkey(cg_ctx* ctx, const ZKey& z, const std::vector<Fr>& public_inputs, int validate_flag = -1, int rank = 0, int world = 1) {
    const bool validate = validate_flag < 0 ? validate_by_default() : validate_flag != 0;
    DeviceZKey d; d.z = &z; d.owner = ctx;
    struct Undo { cg_ctx* c; DeviceZKey* d; bool armed = true; ~Undo() { if (armed) release_zkey(c, *d); } } undo{ctx, &d};   // a failing table must not leak the ones before it
    const Curve& c = z.curve;
    auto reg = [&](const auto& pts, int group, const char* name = "") {
        cg_bases* b; CG(cg_bases_register(ctx, c.id, group, pts.data(), pts.size() / c.aff(group), c.aff(group), -1, &b));
        if (validate) { try { validate_bases(ctx, b, name); } catch (...) { cg_bases_release(b); throw; } }
        return b;
    };
    if (validate && rank == 0) {   // the O(1) verifying-key points and IC go through the same kernels
        Bytes g1 = z.alpha_g1; g1.insert(g1.end(), z.beta_g1.begin(), z.beta_g1.end()); g1.insert(g1.end(), z.delta_g1.begin(), z.delta_g1.end()); g1.insert(g1.end(), z.ic.begin(), z.ic.end());
        Bytes g2 = z.beta_g2; g2.insert(g2.end(), z.gamma_g2.begin(), z.gamma_g2.end()); g2.insert(g2.end(), z.delta_g2.begin(), z.delta_g2.end());
        cg_bases_release(reg(g1, CG_G1, "vk_g1/ic")); cg_bases_release(reg(g2, CG_G2, "vk_g2"));
    }
    auto up = [&](const void* src, size_t bytes) { void* p; CG(cg_dev_alloc(ctx, bytes, &p)); if (bytes) CG(cg_dev_upload(ctx, p, src, bytes)); return p; };
    if (world > 1) {
        const size_t first_aux = z.n_public + 1, n_aux = z.n_vars - first_aux;
        const auto ar = slice_of(n_aux, rank, world), hr = slice_of(z.domain_size, rank, world);
        d.sliced = true; d.aux_lo = ar.first; d.aux_n = ar.second - ar.first; d.h_lo = hr.first; d.h_n = hr.second - hr.first;
        auto cut = [&](const View& v, int group, size_t first, size_t count) { return View{v.data() + first * c.aff(group), count * c.aff(group)}; };
        if (validate && rank == 0) {   // the public-input records of a, b1, b2 stay on the host (calculate_coeff): checked here once
            cg_bases_release(reg(cut(z.a_query, CG_G1, 0, first_aux), CG_G1, "a_query")); cg_bases_release(reg(cut(z.b_g1_query, CG_G1, 0, first_aux), CG_G1, "b_g1_query"));
            cg_bases_release(reg(cut(z.b_g2_query, CG_G2, 0, first_aux), CG_G2, "b_g2_query"));
        }
        d.a = reg(cut(z.a_query, CG_G1, first_aux + d.aux_lo, d.aux_n), CG_G1, "a_query"); d.b1 = reg(cut(z.b_g1_query, CG_G1, first_aux + d.aux_lo, d.aux_n), CG_G1, "b_g1_query");
        d.b2 = reg(cut(z.b_g2_query, CG_G2, first_aux + d.aux_lo, d.aux_n), CG_G2, "b_g2_query");
        d.l = reg(cut(z.l_query, CG_G1, d.aux_lo, d.aux_n), CG_G1, "l_query"); d.h = reg(cut(z.h_query, CG_G1, d.h_lo, d.h_n), CG_G1, "h_query");
        // this device's rows of the two matrices (constraint rows only: rows past num_constraints are the zero / public-input rows of the domain)
        const size_t r0 = std::min<size_t>(d.h_lo, z.num_constraints), r1 = std::min<size_t>(d.h_lo + d.h_n, z.num_constraints);
        for (int m = 0; m < 2; m++) {
            const uint32_t base = z.row_ptr[m][r0];
            std::vector<uint32_t> rp(r1 - r0 + 1);
            for (size_t i = 0; i <= r1 - r0; i++) rp[i] = z.row_ptr[m][r0 + i] - base;
            d.mat_rows[m].row_ptr = (uint32_t*)up(rp.data(), rp.size() * 4);
            d.mat_rows[m].col = (uint32_t*)up(z.col[m].data() + base, (size_t)rp.back() * 4);
            d.mat_rows[m].coeff = up(z.coeff[m].data() + base, (size_t)rp.back() * 32);
            d.mat_rows[m].rows = r1 - r0;
        }
        if (rank != 0) { undo.armed = false; return d; }
    } else {
        d.a = reg(z.a_query, CG_G1, "a_query"); d.b1 = reg(z.b_g1_query, CG_G1, "b_g1_query"); d.b2 = reg(z.b_g2_query, CG_G2, "b_g2_query");
        d.l = reg(z.l_query, CG_G1, "l_query"); d.h = reg(z.h_query, CG_G1, "h_query");
    }
    for (int m = 0; m < 2; m++) {
        d.mat[m].row_ptr = (uint32_t*)up(z.row_ptr[m].data(), z.row_ptr[m].size() * 4);
        d.mat[m].col = (uint32_t*)up(z.col[m].data(), z.col[m].size() * 4);
        d.mat[m].coeff = up(z.coeff[m].data(), z.coeff[m].size() * 32);
        d.mat[m].rows = z.num_constraints;
    }
    d.pub_dev = up(public_inputs.data(), public_inputs.size() * 32);
    undo.armed = false;
    return d;
}
static void release_zkey(cg_ctx* ctx, DeviceZKey& d) {
    for (cg_bases** b : {&d.a, &d.b1, &d.b2, &d.l, &d.h}) { if (*b) cg_bases_release(*b); *b = nullptr; }
    for (int m = 0; m < 2; m++) for (DeviceMatrix* mt : {&d.mat[m], &d.mat_rows[m]}) {
        if (mt->row_ptr) cg_dev_free(ctx, mt->row_ptr); if (mt->col) cg_dev_free(ctx, mt->col); if (mt->coeff) cg_dev_free(ctx, mt->coeff);
        *mt = DeviceMatrix{nullptr, nullptr, nullptr, 0};
    }
    if (d.pub_dev) cg_dev_free(ctx, d.pub_dev);
    d.pub_dev = nullptr;
}
// scope guards of the C entry points: whatever a failing proof leaves behind on the device is released (a long-lived prover that
// hits "randomness stream exhausted" a few times must not run out of HBM)
struct DeviceZKeyGuard {
    cg_ctx* ctx; DeviceZKey dz; bool live = true;
    DeviceZKeyGuard(cg_ctx* c, DeviceZKey d) : ctx(c), dz(d) {}
    ~DeviceZKeyGuard() { if (live) release_zkey(ctx, dz); }
    DeviceZKeyGuard(const DeviceZKeyGuard&) = delete; DeviceZKeyGuard& operator=(const DeviceZKeyGuard&) = delete;
};
struct CtxGuard {
    cg_ctx* ctx = nullptr;
    ~CtxGuard() { if (ctx) cg_ctx_destroy(ctx); }
    cg_ctx* release() { cg_ctx* c = ctx; ctx = nullptr; return c; }
};
struct DevBufGuard { cg_ctx* ctx; void* p; ~DevBufGuard() { if (p) cg_dev_free(ctx, p); } };

// Second contexts for the witness-independent MSMs (HipDriver::aux).  Creating a context costs 15-25 ms (its streams), so they are
// made on a helper thread while the zkey is read and uploaded, and only for zkeys large enough (>= ~2^19 constraints) to gain.
struct SecondContexts {
    std::vector<cg_ctx*> made; std::thread worker;
    SecondContexts(int device, const char* zkey_path, int count) : made(count, nullptr) {
        struct stat st{};
        if (host_option(CGH_OPT_ONE_CONTEXT) || stat(zkey_path, &st) != 0 || st.st_size < (off_t)200 << 20) return;
        worker = std::thread([this, device] { for (auto& c : made) if (cg_ctx_create(device, &c) != 0) c = nullptr; });
    }
    void ready() { if (worker.joinable()) worker.join(); }
    cg_ctx* take(int i) { ready(); cg_ctx* c = made[i]; made[i] = nullptr; return c; }
    ~SecondContexts() { ready(); for (cg_ctx* c : made) if (c) cg_ctx_destroy(c); }
};

static void store_proof(const Proof& p, uint8_t* out) { memcpy(out, p.a.data(), p.a.size()); memcpy(out + p.a.size(), p.b.data(), p.b.size()); memcpy(out + p.a.size() + p.b.size(), p.c.data(), p.c.size()); }

static bool all_zero_bytes(const uint8_t* p, size_t n) { for (size_t i = 0; i < n; i++) if (p[i]) return false; return true; }

}  // namespace cgh
