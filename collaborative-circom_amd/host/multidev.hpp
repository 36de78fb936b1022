// witness_map_from_matrices (co-circom/co-groth16/src/groth16.rs:141-204) of ONE party over SEVERAL GPUs of its node:
//   * the private witness goes up BY ROWS: device d uploads rows [aux_lo_d, aux_lo_d + aux_n_d) of the caller's host vectors over ITS PCIe link
//     (1/N of 2 x 32 B x n_aux per link instead of all of it over the primary's) — the rows its slices of the four aux queries multiply, so its
//     MSMs start as soon as they land — and the devices then complete their copies from each other (all-gather of row blocks, xGMI);
//   * evaluate_constraint (:156-166) runs by ROWS as well: device d holds rows [lo_d, lo_d + n_d) of both matrices (DeviceZKey::mat_rows, the
//     same split as its slice of h_query) and evaluates them against its full copy of the witness;
//   * the six vector pipelines iNTT -> coset shift -> NTT (a.a, a.b, b.a, b.b, then c.a, c.b; groth16.rs:175-200) run one vector per device
//     ("owner" of the vector), which gathers the vector's rows from the devices that hold them and hands rows back (cg_dev_copy_peer);
//   * the two mul_vec calls (:174, :190; rep3.rs:650-670) are cut by the same rows: device d multiplies its rows with its slice of the masks,
//     its slice of the local product goes down over its own link and the previous party's slice comes up over it; the host thread moves the
//     messages in the reference's order (one vector = chunks of at most 4 MiB in index order, whatever devices they come from: the peers need
//     not know);
//   * h = a.b - c (:202) is formed where its rows are, and the h MSM of a device runs over its own table slice and its own rows: the
//     quotient vector is never gathered.
// Nothing but the draws of the masking vectors (one stream position depends on the rejections before it) is left to the primary device.
// Values are those of the single-device path bit for bit: the same kernels on the same numbers, only placed elsewhere.
// Modes: Plain and Rep3 (Shamir's degree reduction keeps the single-device path).  Devices: [primary] + md->workers, in table-slice order.
#pragma once
#include "driver.hpp"

namespace cgh {

struct DistributedH {                          // the quotient evaluations, rows [lo, lo + n) per device, still on their devices
    struct Part { cg_ctx* ctx; ShareVec h; };
    std::vector<Part> parts;
    bool valid() const { return !parts.empty(); }
};

class DistributedWitnessMap {
public:
    HipDriver& drv; const Curve curve; const int k;
    const bool additive;                       // REP3 additive-quotient variant: products are not exchanged, c and h have one component
    const int kc;                              // components of c and h
    struct Dev {
        cg_ctx* ctx = nullptr; const DeviceZKey* dz = nullptr; size_t lo = 0, n = 0;
        cg_ctx* msm = nullptr;                                                     // the context that carries this device's MSM slices
        void* wit[2] = {nullptr, nullptr}; int32_t wit_up[2] = {-1, -1};          // full copy of the private witness (components); upload of its own rows
        void* pub = nullptr;                                                       // the public inputs on this device
        void* av[2] = {nullptr, nullptr}; void* bv[2] = {nullptr, nullptr};      // rows of a and b (components)
        void* prod = nullptr; void* recv = nullptr;                                // mul_vec result rows: local component, received component
        int32_t up_tk = -1;                                                        // last upload into this device
        void* bad = nullptr;                                                       // device counter: received elements that are not below the modulus
        std::vector<void*> owned;                                                  // device allocations to release at the end
    };
    std::vector<Dev> devs;
    const int only_dev;                        // planning builds only (emulate_only_device, base.hpp): the one device whose share runs; -1 = all (always, in the release library)
    std::vector<void*> pinned;                 // page-locked staging released at the end

    DistributedWitnessMap(HipDriver& d, const DeviceZKey& dz0, const MultiDevice& md)
        // (k comes from the MODE, not from d.k(): prove() builds the map while its one-component override for the additive variant's MSMs is live — ADVICE r5)
        : drv(d), curve(d.curve), k(d.mode == Mode::Rep3 ? 2 : 1), additive(d.additive_h && d.mode == Mode::Rep3), kc(additive ? 1 : d.k()), only_dev(emulate_only_device()) {
        Dev p; p.ctx = d.ctx; p.msm = d.aux ? d.aux : d.ctx; p.dz = &dz0; p.lo = dz0.h_lo; p.n = dz0.h_n; devs.push_back(p);
        for (const WorkerDevice& w : md.workers) { Dev x; x.ctx = w.chain ? w.chain : w.ctx; x.msm = w.ctx; x.dz = w.dz; x.lo = w.dz->h_lo; x.n = w.dz->h_n; devs.push_back(x); }
    }
    static bool usable(const HipDriver& d, const DeviceZKey& dz0) {
        const bool off = !host_option(CGH_OPT_DISTRIBUTED_MAP);                    // (read per proof) 0: keep the whole witness map on the primary device (round-2 layout)
        return !off && d.md && !d.md->workers.empty() && dz0.sliced && (d.mode == Mode::Plain || d.mode == Mode::Rep3);
    }
    ~DistributedWitnessMap() {
        for (Dev& d : devs) { cg_ctx_sync(d.ctx); if (d.msm && d.msm != d.ctx && std::uncaught_exceptions() > 0) cg_ctx_sync(d.msm); }   // blocks of one device are read by the others' peer copies: all quiet first (a proof that died may have MSMs in flight over its witness copy)
        for (Dev& d : devs) if (!d.owned.empty()) cg_dev_free_many(d.ctx, d.owned.data(), d.owned.size());   // one release mark per device
        for (void* p : pinned) cg_host_free(p);
        for (void* p : host_tmp) free(p);
    }
    bool skip(size_t d) const { return only_dev >= 0 && d != (size_t)only_dev; }
    void* dalloc(Dev& d, size_t bytes) { void* p; CG(cg_dev_alloc(d.ctx, std::max<size_t>(bytes, 32), &p)); d.owned.push_back(p); return p; }
    void release(Dev& d, void* p) { for (auto it = d.owned.begin(); it != d.owned.end(); ++it) if (*it == p) { d.owned.erase(it); break; } }
    size_t owner(int v) const { return (size_t)(v + 1) % devs.size(); }            // vector v = 0 .. 3k-1 (a components, b components, c components)

    // masks of one mul_vec: rows of device d to that device, over its own link (Rep3 only)
    void upload_masks(std::vector<void*>& mask, size_t m) {
        if (drv.mode != Mode::Rep3) return;
        const bool async = m >= drv.XCHG_ASYNC_MIN;
        const Fr* whole = nullptr; const Fr* s1 = nullptr; const Fr* s2 = nullptr;
        // generators described (cgh_rep3_chacha): the whole vector is drawn on the primary device — a row's stream position depends on the
        // rejections before it — and every device takes its rows over its link to the primary
        if (drv.rsrc && m >= drv.DEVICE_MASKS_MIN && only_dev < 0) {
            Dev& P = devs[0];
            void* all = dalloc(P, m * 32); void* tmp = dalloc(P, m * 32);
            if (drv.rsrc->masks_on_device(P.ctx, curve.id, m, all, tmp)) {
                for (size_t d = 0; d < devs.size(); d++) {
                    Dev& D = devs[d];
                    mask[d] = dalloc(D, D.n * 32);
                    if (D.n) CG(cg_dev_copy_peer(D.ctx, mask[d], P.ctx, (const uint8_t*)all + D.lo * 32, D.n * 32));
                }
                return;
            }
        }
        if (drv.rsrc) {
            Fr* buf = nullptr;
            if (async) { void* p; CG(cg_host_alloc(m * 32, &p)); pinned.push_back(p); buf = (Fr*)p; }
            else { void* p = malloc(std::max<size_t>(m, 1) * 32); if (!p) throw std::runtime_error("out of memory"); host_tmp.push_back(p); buf = (Fr*)p; }
            whole = drv.rsrc->masking_field_elements(m, buf);
            if (async && whole != buf && !cg_host_is_pinned(whole)) { memcpy(buf, whole, m * 32); whole = buf; }
        } else {
            if (drv.cursor + m > drv.rng_len) throw std::runtime_error("randomness stream exhausted");
            s1 = drv.rng1 + drv.cursor; s2 = drv.rng2 + drv.cursor; drv.cursor += m;
        }
        for (size_t d = 0; d < devs.size(); d++) {
            Dev& D = devs[d];
            mask[d] = dalloc(D, D.n * 32);
            if (skip(d) || !D.n) continue;
            if (whole) up(D, mask[d], whole + D.lo, D.n, async);
            else {
                void* m2 = dalloc(D, D.n * 32);
                up(D, mask[d], s1 + D.lo, D.n, async); up(D, m2, s2 + D.lo, D.n, async);
                if (D.up_tk >= 0) CG(cg_copy_fence(D.ctx, D.up_tk));
                CG(cg_vec_sub_dev(D.ctx, curve.id, mask[d], mask[d], m2, D.n));     // masking_field_element = rand(rng1) - rand(rng2)
            }
        }
    }
    std::vector<void*> host_tmp;
    void up(Dev& D, void* dst, const Fr* src, size_t n, bool async) {
        if (async && cg_host_is_pinned(src)) CG(cg_dev_upload_begin(D.ctx, dst, src, n * 32, 0, &D.up_tk));
        else CG(cg_dev_upload(D.ctx, dst, src, n * 32));
    }

    // local products of one mul_vec on every device (prod = rows of component a of the result), then the exchange: the whole vector as
    // the single-device path sends it — chunk by chunk in index order — with every chunk assembled from / scattered to the devices that
    // hold its rows.  recv = rows of component b.
    void mul_rows(size_t m) {
        std::vector<void*> mask(devs.size(), nullptr);
        upload_masks(mask, m);
        for (size_t d = 0; d < devs.size(); d++) {
            Dev& D = devs[d];
            D.prod = dalloc(D, D.n * 32); D.recv = kc == 2 ? dalloc(D, D.n * 32) : nullptr;
            if (skip(d) || !D.n) continue;
            if (drv.mode == Mode::Plain) { CG(cg_vec_mul_dev(D.ctx, curve.id, D.prod, D.av[0], D.bv[0], D.n)); continue; }
            if (D.up_tk >= 0) CG(cg_copy_fence(D.ctx, D.up_tk));
            CG(cg_vec_rep3_mul_local_dev(D.ctx, curve.id, D.prod, D.av[0], D.av[1], D.bv[0], D.bv[1], mask[d], D.n));
        }
        if (drv.mode != Mode::Rep3 || additive) return;                            // (variant: the masked local products are the result)
        const bool async = m >= drv.XCHG_ASYNC_MIN;
        const size_t ch = async ? HipDriver::xchg_chunk(m) : m, nch = (m + ch - 1) / ch;
        const int S = async ? HipDriver::XCHG_SLOTS : 1, P = async ? S - 1 : 1;
        uint8_t* out_buf; uint8_t* in_buf;
        if (async) { void* p; CG(cg_host_alloc((size_t)S * ch * 32, &p)); pinned.push_back(p); out_buf = (uint8_t*)p; CG(cg_host_alloc((size_t)S * ch * 32, &p)); pinned.push_back(p); in_buf = (uint8_t*)p; }
        else { void* p = malloc(2 * std::max<size_t>(ch, 1) * 32); if (!p) throw std::runtime_error("out of memory"); host_tmp.push_back(p); out_buf = (uint8_t*)p; in_buf = out_buf + ch * 32; }
        struct Tk { cg_ctx* ctx; int32_t tk; };
        std::vector<std::vector<Tk>> down(nch), upl((size_t)S);
        auto pieces = [&](size_t off, size_t len, auto&& fn) {                      // the devices whose rows meet [off, off + len)
            for (size_t d = 0; d < devs.size(); d++) {
                const Dev& D = devs[d];
                const size_t lo = std::max(off, D.lo), hi = std::min(off + len, D.lo + D.n);
                if (lo < hi) fn(d, lo - off, lo - D.lo, hi - lo);
            }
        };
        size_t issued = 0;
        auto issue = [&](size_t upto) {
            for (; issued < nch && issued < upto; issued++) {
                const size_t off = issued * ch, len = std::min(ch, m - off);
                uint8_t* slot = out_buf + (issued % S) * ch * 32;
                pieces(off, len, [&](size_t d, size_t at_chunk, size_t at_dev, size_t cnt) {
                    if (skip(d)) return;
                    Dev& D = devs[d];
                    if (async) { int32_t tk; CG(cg_dev_download_begin(D.ctx, slot + at_chunk * 32, (const uint8_t*)D.prod + at_dev * 32, cnt * 32, &tk)); down[issued].push_back({D.ctx, tk}); }
                    else CG(cg_dev_download(D.ctx, slot + at_chunk * 32, (const uint8_t*)D.prod + at_dev * 32, cnt * 32));
                });
            }
        };
        issue((size_t)P);
        for (size_t c = 0; c < nch; c++) {
            issue(c + (size_t)P);
            const size_t off = c * ch, len = std::min(ch, m - off);
            for (const Tk& t : down[c]) CG(cg_copy_wait(t.ctx, t.tk));
            drv.net->send_next(out_buf + (c % S) * ch * 32, len * 32);                // rep3.rs:661-662 (chunked send_next_many)
            uint8_t* slot = in_buf + (c % S) * ch * 32;
            for (const Tk& t : upl[c % S]) CG(cg_copy_wait(t.ctx, t.tk));            // the uploads that read this slot S chunks ago
            upl[c % S].clear();
            const void* src = slot;
            if (const void* direct = async ? drv.net->recv_prev_pinned(len * 32) : nullptr) src = direct;
            else drv.net->recv_prev(slot, len * 32);                                  // rep3.rs:663-669
            pieces(off, len, [&](size_t d, size_t at_chunk, size_t at_dev, size_t cnt) {
                if (skip(d)) return;
                Dev& D = devs[d];
                if (async) { CG(cg_dev_upload_begin(D.ctx, (uint8_t*)D.recv + at_dev * 32, (const uint8_t*)src + at_chunk * 32, cnt * 32, 0, &D.up_tk)); if (src == slot) upl[c % S].push_back({D.ctx, D.up_tk}); }
                else CG(cg_dev_upload(D.ctx, (uint8_t*)D.recv + at_dev * 32, (const uint8_t*)src + at_chunk * 32, cnt * 32));
            });
        }
        for (size_t d = 0; d < devs.size(); d++) if (!skip(d) && devs[d].up_tk >= 0) CG(cg_copy_fence(devs[d].ctx, devs[d].up_tk));   // later launches see the received rows
        for (size_t d = 0; d < devs.size(); d++) if (!skip(d) && devs[d].recv && devs[d].n) {                                         // the receiver's range check (HipDriver::check_received_dev)
            Dev& D = devs[d];
            if (!D.bad) { D.bad = dalloc(D, 32); CG(cg_dev_memset_zero(D.ctx, D.bad, 32)); }
            CG(cg_vec_check_canonical_dev(D.ctx, curve.id, D.recv, D.n, D.bad));
        }
    }

    // read where the proof's streams are idle (CoGroth16::prove, before the last opening)
    void verify_received() {
        uint64_t total = 0;
        for (Dev& D : devs) if (D.bad) { uint64_t bad = 0; CG(cg_dev_download(D.ctx, &bad, D.bad, 8)); total += bad; }
        if (total) throw std::runtime_error("invalid data: " + std::to_string(total) + " field element(s) of a vector received from a peer are not below the modulus");
    }
    // ---- step 0: the private witness on every device.  host_a / host_b (the caller's vectors): every device uploads its own rows over its own
    // link, then the row blocks travel device to device.  resident (a vector the caller already holds on the primary device): whole copies.
    size_t n_aux = 0;
    void place_witness(const Fr* host_a, const Fr* host_b, const ShareVec* resident, size_t n, const std::vector<Fr>& public_inputs) {
        n_aux = n;
        const bool async = n >= drv.XCHG_ASYNC_MIN;
        // the public inputs first (a few elements, read by the constraint rows that mention them): one page-locked copy, uploaded everywhere
        void* pp = nullptr; CG(cg_host_alloc(std::max<size_t>(public_inputs.size(), 1) * 32, &pp)); pinned.push_back(pp);
        memcpy(pp, public_inputs.data(), public_inputs.size() * 32);
        for (size_t d = 0; d < devs.size(); d++) {
            Dev& D = devs[d];
            D.pub = dalloc(D, public_inputs.size() * 32);
            for (int j = 0; j < k; j++) D.wit[j] = dalloc(D, n * 32);
            if (skip(d)) continue;
            CG(cg_dev_upload_begin(D.ctx, D.pub, pp, public_inputs.size() * 32, 0, &D.up_tk));
            if (resident) continue;
            const size_t lo = D.dz->aux_lo, cnt = D.dz->aux_n;
            for (int j = 0; j < k; j++) {
                const Fr* src = (j == 0 ? host_a : host_b) + lo;
                if (!cnt) continue;
                if (async && cg_host_is_pinned(src)) { CG(cg_dev_upload_begin(D.ctx, (uint8_t*)D.wit[j] + lo * 32, src, cnt * 32, 0, &D.wit_up[j])); D.up_tk = D.wit_up[j]; }
                else CG(cg_dev_upload(D.ctx, (uint8_t*)D.wit[j] + lo * 32, src, cnt * 32));
            }
        }
        if (resident) {
            for (size_t d = 0; d < devs.size(); d++) if (!skip(d)) for (int j = 0; j < k; j++)
                CG(cg_dev_copy_peer(devs[d].ctx, devs[d].wit[j], drv.ctx, resident->c[j], n * 32));
        }
    }
    // the rows device d multiplies in its MSM slices of the four aux queries: [aux_lo_d, aux_lo_d + aux_n_d) of its own copy (still on their
    // way up, possibly: the schedules wait for the copies on the device)
    ShareVec aux_rows(size_t d) const {
        const Dev& D = devs[d];
        ShareVec v; v.n = D.dz->aux_n;
        for (int j = 0; j < k; j++) { v.c[j] = (uint8_t*)D.wit[j] + D.dz->aux_lo * 32; v.up[j] = D.wit_up[j]; }
        if (D.wit_up[0] >= 0) v.up_ctx = D.ctx;
        return v;
    }
    // the four aux MSMs (groth16.rs:251,267,284,298), every device over its own table slices and its own rows, enqueued from one host
    // thread per device (a launch sequence costs the host ~0.3 ms; eight in a row would be the critical path of an eight-GPU proof)
    HipDriver::PendingMsm begin_aux_msms() {
        HipDriver::PendingMsm p;
        const DeviceZKey& dz0 = *devs[0].dz;
        const int kk = drv.k();                                                        // share components multiplied (the additive variant: the own one)
        std::vector<std::thread> th; std::vector<std::string> errs(devs.size());
        std::vector<HipDriver::PendingMsm::Part> parts(devs.size());
        struct Joined { std::vector<std::thread>& t; ~Joined() { for (auto& x : t) if (x.joinable()) x.join(); } } joined{th};
        for (size_t d = 1; d < devs.size(); d++) {
            if (skip(d)) continue;
            th.emplace_back([this, d, kk, &parts, &errs] {
                try {
                    const Dev& D = devs[d]; const DeviceZKey& wz = *D.dz;
                    const ShareVec sv = aux_rows(d);
                    HipDriver::PendingMsm::Part part{D.msm, std::vector<int32_t>(4), {nullptr, nullptr}};
                    if (sv.up_ctx) { for (int j = 0; j < kk; j++) if (sv.up[j] >= 0) CG(cg_msm_scalars_after(D.msm, j, sv.up_ctx, sv.up[j])); }
                    else CG(cg_ctx_sync(D.ctx));                                         // synchronous uploads / peer copies on the chain context's stream
                    const void* sc[2] = {sv.c[0], sv.c[1]};
                    drv.begin_multi_ordered(D.msm, {wz.a, wz.b1, wz.b2, wz.l}, {0, 0, 0, 0}, {CG_G1, CG_G1, CG_G2, CG_G1}, sv.n, sc, part.tickets);
                    parts[d] = part;
                } catch (const std::exception& e) { errs[d] = e.what(); }
            });
        }
        if (!drv.aux && devs[0].up_tk >= 0) CG(cg_copy_fence(drv.ctx, devs[0].up_tk));    // one context: its own stream carries the schedules
        p = drv.msm_begin_multi({dz0.a, dz0.b1, dz0.b2, dz0.l}, {0, 0, 0, 0}, {CG_G1, CG_G1, CG_G2, CG_G1}, skip(0) ? 0 : dz0.aux_n, aux_rows(0), true);   // (a skipped device: empty MSMs = infinity)
        for (auto& x : th) x.join();
        for (const std::string& e : errs) if (!e.empty()) throw std::runtime_error(e);
        for (size_t d = 1; d < devs.size(); d++) if (!skip(d)) p.parts.push_back(parts[d]);
        return p;
    }
    // all-gather of the row blocks: every device completes its copy of the witness from the devices that uploaded the other rows
    void gather_witness(bool from_rows) {
        if (!from_rows) return;
        for (size_t d = 0; d < devs.size(); d++) if (!skip(d) && devs[d].up_tk >= 0) CG(cg_copy_fence(devs[d].ctx, devs[d].up_tk));   // own rows (and the public inputs) have landed
        for (size_t d = 0; d < devs.size(); d++) {
            if (skip(d)) continue;
            Dev& D = devs[d];
            for (size_t s_ = 0; s_ < devs.size(); s_++) {
                if (s_ == d) continue;
                const Dev& S = devs[s_];
                const size_t lo = S.dz->aux_lo, cnt = S.dz->aux_n;
                if (!cnt) continue;
                for (int j = 0; j < k; j++) CG(cg_dev_copy_peer(D.ctx, (uint8_t*)D.wit[j] + lo * 32, S.ctx, (const uint8_t*)S.wit[j] + lo * 32, cnt * 32));
            }
        }
    }

    DistributedH run(const DeviceZKey& dz, const std::vector<Fr>& public_inputs) {
        const ZKey& z = *dz.z;
        const size_t num_inputs = z.n_public + 1, num_constraints = z.num_constraints;
        const Domain dom = groth16_domain(curve, z.pow, num_constraints, num_inputs);          // groth16.rs:150-153
        const size_t m = dom.m, nd = devs.size();
        const int party = drv.party();
        const int holder = drv.public_component();                                             // promote_to_trivial_shares: who holds a public value (fieldshare.rs:262-283)
        // :156-171 by rows: rows [lo_d, lo_d + n_d) of a and b on device d (zero past the constraints, the public inputs spliced in at
        // [num_constraints, num_constraints + num_inputs) of a)
        for (size_t d = 0; d < nd; d++) {
            Dev& D = devs[d];
            for (int j = 0; j < k; j++) { D.av[j] = dalloc(D, D.n * 32); D.bv[j] = dalloc(D, D.n * 32); }
            if (skip(d) || !D.n) continue;
            if (D.up_tk >= 0) CG(cg_copy_fence(D.ctx, D.up_tk));
            for (int j = 0; j < k; j++) { CG(cg_dev_memset_zero(D.ctx, D.av[j], D.n * 32)); CG(cg_dev_memset_zero(D.ctx, D.bv[j], D.n * 32)); }
            const DeviceMatrix* mt = D.dz->mat_rows;
            if (mt[0].rows) CG(cg_spmv_csr_dev(D.ctx, curve.id, mt[0].row_ptr, mt[0].col, mt[0].coeff, mt[0].rows, D.pub, (uint32_t)num_inputs, party, D.wit[0], D.wit[1], D.av[0], D.av[1]));
            if (mt[1].rows) CG(cg_spmv_csr_dev(D.ctx, curve.id, mt[1].row_ptr, mt[1].col, mt[1].coeff, mt[1].rows, D.pub, (uint32_t)num_inputs, party, D.wit[0], D.wit[1], D.bv[0], D.bv[1]));
            const size_t p0 = std::max(num_constraints, D.lo), p1 = std::min(num_constraints + num_inputs, D.lo + D.n);
            if (holder >= 0 && p0 < p1) CG(cg_dev_copy_peer(D.ctx, (uint8_t*)D.av[holder] + (p0 - D.lo) * 32, D.ctx, (const uint8_t*)D.pub + (p0 - num_constraints) * 32, (p1 - p0) * 32));
        }
        // whole vectors on their owners, gathered from the rows (vector index: a components 0..k-1, b components k..2k-1, c components 2k..3k-1)
        std::vector<void*> vec((size_t)2 * k + kc, nullptr);
        auto gather_vector = [&](int v, auto&& rows_of) {                                       // rows_of(d): device d's rows of vector v
            Dev& O = devs[owner(v)];
            vec[v] = dalloc(O, m * 32);
            if (skip(owner(v))) return;
            for (size_t d = 0; d < nd; d++) { Dev& D = devs[d]; if (D.n) CG(cg_dev_copy_peer(O.ctx, (uint8_t*)vec[v] + D.lo * 32, D.ctx, rows_of(d), D.n * 32)); }
        };
        for (int j = 0; j < k; j++) { gather_vector(j, [&](size_t d) { return devs[d].av[j]; }); gather_vector(k + j, [&](size_t d) { return devs[d].bv[j]; }); }
        // the pipelines of a and b on their owners (:175-188) under the first product's exchange (:174)
        mul_rows_begin_pipelines(vec, 0, 2 * k, dom);                                          // enqueued on the owners' streams
        mul_rows(m);                                                                           // :174 -> rows of c on every device
        // c: rows to the owners of its components, pipelines there (:194-200)
        std::vector<void*> crow_a(nd), crow_b(nd);
        for (size_t d = 0; d < nd; d++) { crow_a[d] = devs[d].prod; crow_b[d] = devs[d].recv; }
        for (int j = 0; j < kc; j++) gather_vector(2 * k + j, [&](size_t d) { return j == 0 ? crow_a[d] : crow_b[d]; });
        mul_rows_begin_pipelines(vec, 2 * k, 2 * k + kc, dom);
        // rows of the transformed a and b for the second product (:190): back into the row buffers the first product has read
        for (size_t d = 0; d < nd; d++) {
            Dev& D = devs[d];
            if (skip(d) || !D.n) continue;
            for (int j = 0; j < k; j++) {
                CG(cg_dev_copy_peer(D.ctx, D.av[j], devs[owner(j)].ctx, (const uint8_t*)vec[j] + D.lo * 32, D.n * 32));
                CG(cg_dev_copy_peer(D.ctx, D.bv[j], devs[owner(k + j)].ctx, (const uint8_t*)vec[k + j] + D.lo * 32, D.n * 32));
            }
        }
        mul_rows(m);                                                                           // :190 -> rows of a.b
        // h rows = a.b - c (:202), c's rows coming back from the owners of its components
        DistributedH out;
        for (size_t d = 0; d < nd; d++) {
            Dev& D = devs[d];
            ShareVec h; h.n = D.n; h.c[0] = D.prod; h.c[1] = kc == 2 ? D.recv : nullptr;
            for (int j = 0; j < kc; j++) {
                void* crow = dalloc(D, D.n * 32);
                if (skip(d) || !D.n) continue;
                CG(cg_dev_copy_peer(D.ctx, crow, devs[owner(2 * k + j)].ctx, (const uint8_t*)vec[2 * k + j] + D.lo * 32, D.n * 32));
                CG(cg_vec_sub_dev(D.ctx, curve.id, h.c[j], h.c[j], crow, D.n));
            }
            release(D, h.c[0]); if (h.c[1]) release(D, h.c[1]);                                // handed to the caller
            out.parts.push_back({D.ctx, h});
        }
        return out;
    }
    // iNTT -> coset shift -> NTT of vectors [v0, v1) on their owners' streams
    void mul_rows_begin_pipelines(std::vector<void*>& vec, int v0, int v1, const Domain& dom) {
        for (int v = v0; v < v1; v++) {
            const size_t o = owner(v);
            if (skip(o)) continue;
            void* p[1] = {vec[v]};
            CG(cg_ntt_coset_pair_dev(devs[o].ctx, curve.id, p, 1, dom.m, dom.omega.v, dom.coset_g.v));
        }
    }
};

}  // namespace cgh
