// witness_map_from_matrices (co-circom/co-groth16/src/groth16.rs:141-204) of ONE party over SEVERAL GPUs of its node:
//   * the six vector pipelines iNTT -> coset shift -> NTT (a.a, a.b, b.a, b.b, then c.a, c.b; groth16.rs:175-200) run one vector per device
//     ("owner" of the vector), whole vectors travelling device to device (cg_dev_copy_peer: xGMI between different GPUs);
//   * the two mul_vec calls (:174, :190; rep3.rs:650-670) are cut by ROWS: device d multiplies rows [lo_d, lo_d + n_d) — the same split as
//     its slice of h_query — with its slice of the masks uploaded over its own PCIe link, its slice of the local product downloaded
//     over its own link, and the previous party's slice uploaded to it; the host thread moves the messages in the reference's order
//     (one vector = chunks of at most 4 MiB in index order, whatever devices they come from: the peers need not know);
//   * h = a.b - c (:202) is formed where its rows are, and the h MSM of a device runs over its own table slice and its own rows: the
//     quotient vector is never gathered.
// The primary device (the driver's context) evaluates the constraints (:156-171) and hands out the rows and vectors.  Values are those
// of the single-device path bit for bit: the same kernels on the same numbers, only placed elsewhere.
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
        void* av[2] = {nullptr, nullptr}; void* bv[2] = {nullptr, nullptr};      // rows of a and b (components)
        void* prod = nullptr; void* recv = nullptr;                                // mul_vec result rows: local component, received component
        int32_t up_tk = -1;                                                        // last upload into this device
        void* bad = nullptr;                                                       // device counter: received elements that are not below the modulus
        std::vector<void*> owned;                                                  // device allocations to release at the end
    };
    std::vector<Dev> devs;
    const bool primary_only;                   // CGH_EMULATE_PRIMARY_ONLY: time the primary device's share (results are then wrong)
    std::vector<void*> pinned;                 // page-locked staging released at the end

    DistributedWitnessMap(HipDriver& d, const DeviceZKey& dz0, const MultiDevice& md)
        : drv(d), curve(d.curve), k(d.k()), additive(d.additive_h && d.mode == Mode::Rep3), kc(additive ? 1 : d.k()), primary_only(getenv("CGH_EMULATE_PRIMARY_ONLY") != nullptr) {
        Dev p; p.ctx = d.ctx; p.dz = &dz0; p.lo = dz0.h_lo; p.n = dz0.h_n; devs.push_back(p);
        for (const WorkerDevice& w : md.workers) { Dev x; x.ctx = w.chain ? w.chain : w.ctx; x.dz = w.dz; x.lo = w.dz->h_lo; x.n = w.dz->h_n; devs.push_back(x); }
    }
    static bool usable(const HipDriver& d, const DeviceZKey& dz0) {
        const bool off = getenv("CGH_NO_DISTRIBUTED_MAP") != nullptr;              // A/B knob (read per proof): keep the whole witness map on the primary device
        return !off && d.md && !d.md->workers.empty() && dz0.sliced && (d.mode == Mode::Plain || d.mode == Mode::Rep3);
    }
    ~DistributedWitnessMap() {
        for (Dev& d : devs) cg_ctx_sync(d.ctx);                                    // blocks of one device are read by the others' peer copies: all quiet first
        for (Dev& d : devs) for (void* p : d.owned) cg_dev_free(d.ctx, p);
        for (void* p : pinned) cg_host_free(p);
        for (void* p : host_tmp) free(p);
    }
    bool skip(size_t d) const { return primary_only && d != 0; }
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
        if (drv.rsrc && m >= drv.DEVICE_MASKS_MIN && !primary_only) {
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
    // rows [D.lo, D.lo + D.n) of whole vectors `src[j]` (on device `from`) into D's row buffers dst[j]
    void rows_to(Dev& D, void* const* dst, Dev& from, void* const* src) {
        for (int j = 0; j < k; j++) CG(cg_dev_copy_peer(D.ctx, dst[j], from.ctx, (const uint8_t*)src[j] + D.lo * 32, D.n * 32));
    }

    DistributedH run(const DeviceZKey& dz, const std::vector<Fr>& public_inputs, const ShareVec& private_witness) {
        const ZKey& z = *dz.z;
        const size_t num_inputs = z.n_public + 1, num_constraints = z.num_constraints;
        const Domain dom = groth16_domain(curve, z.pow, num_constraints, num_inputs);          // groth16.rs:150-153
        const size_t m = dom.m, nd = devs.size();
        Dev& P0 = devs[0];
        // :156-171 on the primary device
        ShareVec a = drv.evaluate_constraints(dz.mat[0], dz.pub_dev, (uint32_t)num_inputs, private_witness, m);
        ShareVec b = drv.evaluate_constraints(dz.mat[1], dz.pub_dev, (uint32_t)num_inputs, private_witness, m);
        for (int j = 0; j < k; j++) { P0.owned.push_back(a.c[j]); P0.owned.push_back(b.c[j]); }
        drv.clone_public_into(a, num_constraints, public_inputs, dz.pub_dev);
        // whole vectors to their owners (vector index: a components 0..k-1, b components k..2k-1, c components 2k..3k-1)
        std::vector<void*> vec((size_t)2 * k + kc, nullptr);
        for (int j = 0; j < k; j++) { vec[j] = a.c[j]; vec[k + j] = b.c[j]; }
        for (int v = 0; v < 2 * k; v++) {
            const size_t o = owner(v);
            if (o == 0) continue;
            void* dst = dalloc(devs[o], m * 32);
            CG(cg_dev_copy_peer(devs[o].ctx, dst, P0.ctx, vec[v], m * 32));
            vec[v] = dst;
        }
        // rows of a and b for the first product (:174), then the pipelines of a and b on their owners (:175-188) under the exchange
        for (size_t d = 0; d < nd; d++) {
            Dev& D = devs[d];
            if (d == 0) { for (int j = 0; j < k; j++) { D.av[j] = (uint8_t*)a.c[j] + D.lo * 32; D.bv[j] = (uint8_t*)b.c[j] + D.lo * 32; } continue; }
            for (int j = 0; j < k; j++) { D.av[j] = dalloc(D, D.n * 32); D.bv[j] = dalloc(D, D.n * 32); }
            rows_to(D, D.av, P0, a.c); rows_to(D, D.bv, P0, b.c);
        }
        // the primary's own rows are read in place by its product; its copies of a / b must not be transformed before that product ran:
        // vectors the primary owns are transformed in a private copy
        for (int v = 0; v < 2 * k; v++) if (owner(v) == 0) { void* cp = dalloc(P0, m * 32); CG(cg_dev_copy_peer(P0.ctx, cp, P0.ctx, vec[v], m * 32)); vec[v] = cp; }
        mul_rows_begin_pipelines(vec, 0, 2 * k, dom);                                          // enqueued on the owners' streams
        mul_rows(m);                                                                           // :174 -> rows of c on every device
        // c: rows to the owners of its components, pipelines there (:194-200)
        std::vector<void*> crow_a(nd), crow_b(nd);
        for (size_t d = 0; d < nd; d++) { crow_a[d] = devs[d].prod; crow_b[d] = devs[d].recv; }
        for (int j = 0; j < kc; j++) {
            Dev& O = devs[owner(2 * k + j)];
            vec[2 * k + j] = dalloc(O, m * 32);
            for (size_t d = 0; d < nd; d++) {
                Dev& D = devs[d];
                if (D.n) CG(cg_dev_copy_peer(O.ctx, (uint8_t*)vec[2 * k + j] + D.lo * 32, D.ctx, j == 0 ? crow_a[d] : crow_b[d], D.n * 32));
            }
        }
        mul_rows_begin_pipelines(vec, 2 * k, 2 * k + kc, dom);
        // rows of the transformed a and b for the second product (:190)
        for (size_t d = 0; d < nd; d++) {
            Dev& D = devs[d];
            if (d == 0) for (int j = 0; j < k; j++) { D.av[j] = dalloc(D, D.n * 32); D.bv[j] = dalloc(D, D.n * 32); }
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
                CG(cg_dev_copy_peer(D.ctx, crow, devs[owner(2 * k + j)].ctx, (const uint8_t*)vec[2 * k + j] + D.lo * 32, D.n * 32));
                if (!skip(d) && D.n) CG(cg_vec_sub_dev(D.ctx, curve.id, h.c[j], h.c[j], crow, D.n));
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
