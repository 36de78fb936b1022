// Sanitizer run (AddressSanitizer + UndefinedBehaviorSanitizer) of the two pieces of host C++ that parse untrusted files and juggle raw
// buffers: the CPU oracle (oracle/oracle_capi.cpp, test infrastructure) and the host mirror (collaborative-circom_amd/host/*.hpp and capi_*.cpp).
// Both are compiled INTO this binary with -fsanitize=address,undefined (tests/sanitize/Makefile); no GPU is touched: the host
// mirror's readers, JSON codecs and the secret-shared witness container run on the CPU, everything else is the oracle.
//   usage: san_main <golden dir> <scratch dir>
// Exit code 0 = every check passed and no sanitizer report (a report aborts the process).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>
#include "cogroth16_host.h"

extern "C" {
const char* orc_last_error();
void* orc_zkey_open(int curve, const char* path);
void orc_zkey_close(void* h);
int orc_zkey_info(void* h, size_t* info);
int orc_zkey_points(void* h, int which, uint64_t* out);
int orc_wtns_read(int curve, const char* path, uint64_t* out, size_t cap, size_t* n);
int orc_prove_plain(void* h, const uint64_t* full_witness, const uint64_t* r, const uint64_t* s, int threads, uint64_t* out_proof, double* seconds);
int orc_bench_ntt_selfcheck(int curve, int log_n, int threads);
int orc_make_synthetic(int curve, int log_m, uint64_t seed, const char* zkey_path, const char* wtns_path, int threads);
double orc_bench_rep3_party(int curve, int log_m, int threads, uint64_t seed, double* stage);
int orc_pairing_selfcheck(int curve, const uint64_t* scalar);
}

static int failures = 0;
#define EXPECT(cond, what) do { if (!(cond)) { printf("FAIL: %s\n", what); failures++; } else printf("ok: %s\n", what); } while (0)

static std::vector<uint8_t> slurp(const std::string& p) { std::ifstream f(p, std::ios::binary); return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>()); }
static void spit(const std::string& p, const std::vector<uint8_t>& b) { std::ofstream f(p, std::ios::binary); f.write((const char*)b.data(), (std::streamsize)b.size()); }

int main(int argc, char** argv) {
    if (argc < 3) { printf("usage: san_main <golden dir> <scratch dir>\n"); return 2; }
    const std::string golden = argv[1], tmp = argv[2];
    for (int curve = 0; curve < 2; curve++) {
        const std::string dir = golden + "/groth16/" + (curve == 0 ? "bn254" : "bls12_381") + "/poseidon/";
        const std::string zp = dir + "circuit.zkey", wp = dir + "witness.wtns";
        // ---- oracle: read, prove, (tiny) baseline workload
        void* z = orc_zkey_open(curve, zp.c_str());
        EXPECT(z != nullptr, "oracle opens the poseidon zkey");
        size_t info[8] = {0};
        EXPECT(orc_zkey_info(z, info) >= 0 && info[0] > 0, "oracle zkey info");
        size_t n = 0;
        orc_wtns_read(curve, wp.c_str(), nullptr, 0, &n);
        std::vector<uint64_t> w(n * 4);
        EXPECT(orc_wtns_read(curve, wp.c_str(), w.data(), n, &n) >= 0 && n == info[0], "oracle reads the witness");
        uint64_t r[4] = {5, 0, 0, 0}, s[4] = {7, 0, 0, 0};
        std::vector<uint64_t> proof(8 * (curve ? 6 : 4));
        EXPECT(orc_prove_plain(z, w.data(), r, s, 2, proof.data(), nullptr) >= 0, "oracle proves poseidon (plain driver)");
        orc_zkey_close(z);
        EXPECT(orc_bench_ntt_selfcheck(curve, 9, 3) == 1, "blocked multi-threaded transforms equal the plain ones");
        // ---- host mirror: readers and codecs (no GPU)
        size_t hinfo[7] = {0};
        EXPECT(cgh_zkey_info(curve, zp.c_str(), hinfo) == 0 && hinfo[0] == info[0], "host mirror zkey info agrees with the oracle");
        size_t hn = 0;
        EXPECT(cgh_read_wtns(curve, wp.c_str(), nullptr, 0, &hn) == 0 && hn == n, "host mirror witness size");
        std::vector<uint64_t> hw(hn * 4);
        EXPECT(cgh_read_wtns(curve, wp.c_str(), hw.data(), hn, &hn) == 0 && hw == w, "host mirror witness equals the oracle's");
        std::vector<char> js(8192);
        EXPECT(cgh_proof_to_json(curve, proof.data(), js.data(), js.size()) == 0, "proof -> JSON");
        std::vector<uint64_t> back(proof.size());
        EXPECT(cgh_proof_from_json(curve, js.data(), back.data()) == 0 && back == proof, "JSON -> proof round trip");
        EXPECT(cgh_proof_to_json(curve, proof.data(), js.data(), 16) != 0, "proof -> JSON into a short buffer is refused");
        const std::string sh = tmp + "/w.shared";
        EXPECT(cgh_shared_witness_write(curve, sh.c_str(), 0, w.data(), 2, w.data() + 8, w.data() + 8, n - 2) == 0, "shared witness written");
        size_t sizes[2] = {0, 0};
        EXPECT(cgh_shared_witness_read(curve, sh.c_str(), 0, sizes, nullptr, nullptr, nullptr) == 0 && sizes[0] == 2 && sizes[1] == n - 2, "shared witness sizes");
        std::vector<uint64_t> p(8), a((n - 2) * 4), b((n - 2) * 4);
        EXPECT(cgh_shared_witness_read(curve, sh.c_str(), 0, sizes, p.data(), a.data(), b.data()) == 0 && memcmp(a.data(), w.data() + 8, a.size() * 8) == 0, "shared witness round trip");
        // ---- malformed files must be refused, not read out of bounds
        std::vector<uint8_t> good = slurp(zp);
        for (size_t cut : {(size_t)8, (size_t)40, good.size() / 3, good.size() - 5}) {
            std::vector<uint8_t> t(good.begin(), good.begin() + cut);
            spit(tmp + "/trunc.zkey", t);
            EXPECT(cgh_zkey_info(curve, (tmp + "/trunc.zkey").c_str(), hinfo) != 0, "truncated zkey refused");
        }
        {   // first section length = 2^64 - 8: off + len wraps
            std::vector<uint8_t> t = good; const uint64_t huge = ~(uint64_t)0 - 7; memcpy(t.data() + 16, &huge, 8);
            spit(tmp + "/wrap.zkey", t);
            EXPECT(cgh_zkey_info(curve, (tmp + "/wrap.zkey").c_str(), hinfo) != 0, "wrapping section length refused");
        }
        {   // a matrix column index beyond n_vars (section 4 record: u32 matrix, u32 row, u32 signal, 32 B value)
            std::vector<uint8_t> t = good; size_t off = 12; bool done = false;
            while (off + 12 <= t.size() && !done) {
                uint32_t id; uint64_t len; memcpy(&id, t.data() + off, 4); memcpy(&len, t.data() + off + 4, 8);
                if (id == 4) { const uint32_t bad = 0x7fffffffu; memcpy(t.data() + off + 12 + 4 + 8, &bad, 4); done = true; }
                off += 12 + len;
            }
            spit(tmp + "/col.zkey", t);
            void* dummy = nullptr;
            EXPECT(done && cgh_session_open_ex(0, curve, (tmp + "/col.zkey").c_str(), 0, 1, &dummy) != 0, "column index beyond n_vars refused (or no GPU): no out-of-bounds read");
        }
        std::vector<uint8_t> wt = slurp(wp); wt.resize(wt.size() / 2); spit(tmp + "/trunc.wtns", wt);
        EXPECT(cgh_read_wtns(curve, (tmp + "/trunc.wtns").c_str(), hw.data(), hn, &hn) != 0, "truncated witness refused");
    }
    // ---- oracle: synthetic circuit generator and the baseline workload at a tiny size, pairing
    EXPECT(orc_make_synthetic(0, 6, 3, (tmp + "/s.zkey").c_str(), (tmp + "/s.wtns").c_str(), 3) >= 0, "oracle synthetic circuit 2^6");
    double st[4];
    EXPECT(orc_bench_rep3_party(0, 8, 3, 1, st) > 0, "CPU baseline workload 2^8");
    uint64_t k[4] = {12345, 0, 0, 0};
    EXPECT(orc_pairing_selfcheck(0, k) >= 0, "pairing self-check");
    printf("%d failure(s)\n", failures);
    return failures ? 1 : 0;
}
