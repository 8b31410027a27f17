// Host build of the phase-A kernel for sanitizers (tools only): each lane is run as an independent call,
// which is exact for phase A because lanes share nothing but disjoint LDS columns.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <thread>
#define __device__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(x)
#define __restrict__
#define __shared__
struct dim3s { unsigned x, y, z; };
static thread_local dim3s threadIdx, blockIdx;
static inline uint32_t __brev(uint32_t v) { uint32_t r = 0; for (int i = 0; i < 32; i++) r |= ((v >> i) & 1u) << (31 - i); return r; }
#define HOSTSIM 1
struct uint4 { uint32_t x, y, z, w; };
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
static inline unsigned long long __ballot(bool b) { return b ? 1ull : 0ull; }
static uint8_t *g_smem;
#define HIP_RUNTIME_STUB
#include <stdint.h>
// minimal copy of the shared declarations (dhts_common.h pulls hip headers)
struct BgzfTable { const uint64_t *coff; const uint32_t *clen; const uint32_t *isize; const uint64_t *uoff; int64_t n; };
#define DHTS_LIT_STRIDE 65536u
#define DHTS_TOK_STRIDE 22528u
#define DHTS_TOK_PURE 511u
struct InflateMeta { uint32_t ntok, nlit, outlen; int32_t status; };
#define DHTS_BLK_OK 0
#define DHTS_BLK_ERR_INFLATE (-3)
#define DHTS_BLK_ERR_CRC (-4)
#define DHTS_BLK_ERR_ISIZE (-5)
#define DHTS_COMMON_INCLUDED
#include "phaseA_extract.inc"

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    FILE *f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> d(n + 256, 0); if (fread(d.data(), 1, n, f) != (size_t)n) return 2; fclose(f);
    std::vector<uint64_t> coff; std::vector<uint32_t> clen, isz; std::vector<uint64_t> uoff;
    for (long p = 0; p < n;) { uint32_t bl = (d[p + 16] | (d[p + 17] << 8)) + 1; coff.push_back(p); clen.push_back(bl); isz.push_back(0); uoff.push_back(0); p += bl; }
    int64_t nb = coff.size();
    BgzfTable t{coff.data(), clen.data(), isz.data(), uoff.data(), nb};
    std::vector<uint8_t> lit((size_t)nb * DHTS_LIT_STRIDE + 8192); std::vector<uint32_t> tok((size_t)nb * DHTS_TOK_STRIDE + 64); std::vector<InflateMeta> meta(nb);
    const uint32_t nlo = getenv("NLO") ? (uint32_t)atoi(getenv("NLO")) : A_NLO_FAR;      // both LDS layouts are run by run.sh
    int bad = 0;
    for (int64_t wg = 0; wg * A_SL < nb; wg++) {
        std::vector<uint8_t> smem(A_LDS_BYTES_FOR(nlo), (uint8_t)(getenv("FILL") ? atoi(getenv("FILL")) : 0));  // exact size: ASAN catches any overrun
        g_smem = smem.data();
#ifdef SIM_THREADS
        std::vector<std::thread> th;
        for (int lane = 0; lane < A_SL; lane++) th.emplace_back([&, lane]() {
            blockIdx.x = (unsigned)wg; threadIdx.x = (unsigned)lane;
            bgzf_huff_decode(d.data(), t, 0, (int32_t)nb, lit.data(), tok.data(), meta.data(), nlo);
        });
        for (auto &x : th) x.join();
#else
        for (int lane = 0; lane < A_SL; lane++) {
            blockIdx.x = (unsigned)wg; threadIdx.x = (unsigned)lane;
            bgzf_huff_decode(d.data(), t, 0, (int32_t)nb, lit.data(), tok.data(), meta.data(), nlo);
        }
#endif
    }
    // replay the tokens (what phase B does) and check every block against its CRC32 / ISIZE trailer
    uint32_t crct[256];
    for (uint32_t k = 0; k < 256; k++) { uint32_t c = k; for (int j = 0; j < 8; j++) c = (c & 1u) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1); crct[k] = c; }
    int mism = 0;
    for (int64_t b = 0; b < nb; b++) {
        if (meta[b].status) { bad++; continue; }
        const uint8_t *L = lit.data() + (size_t)b * DHTS_LIT_STRIDE; const uint32_t *T = tok.data() + (size_t)b * DHTS_TOK_STRIDE;
        std::vector<uint8_t> out; out.reserve(65536);
        uint32_t lp = 0;
        for (uint32_t i = 0; i < meta[b].ntok; i++) {
            const uint32_t t = T[i], run = t >> 23;
            for (uint32_t k = 0; k < run; k++) out.push_back(L[lp++]);
            if (run != DHTS_TOK_PURE) {
                const uint32_t len = ((t >> 15) & 255u) + 3, dist = (t & 0x7fffu) + 1;
                if (dist > out.size()) { mism++; break; }
                for (uint32_t k = 0; k < len; k++) out.push_back(out[out.size() - dist]);
            }
        }
        while (lp < meta[b].nlit) out.push_back(L[lp++]);
        uint32_t c = 0xffffffffu; for (uint8_t x : out) c = crct[(c ^ x) & 0xff] ^ (c >> 8);
        c ^= 0xffffffffu;
        const uint8_t *tr = d.data() + coff[b] + clen[b] - 8;
        const uint32_t want_crc = tr[0] | (tr[1] << 8) | (tr[2] << 16) | ((uint32_t)tr[3] << 24), want_len = tr[4] | (tr[5] << 8) | (tr[6] << 16) | ((uint32_t)tr[7] << 24);
        if (out.size() != want_len || out.size() != meta[b].outlen || c != want_crc) mism++;
    }
    printf("blocks %lld failed %d mismatching %d (SL=%d LDS=%d)\n", (long long)nb, bad, mism, A_SL, (int)A_LDS_BYTES_FOR(nlo));
    return (bad || mism) ? 1 : 0;
}
