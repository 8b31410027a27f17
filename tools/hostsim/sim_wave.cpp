// Host build of the wave-per-block Huffman kernel (duckhts_amd/csrc/bgzf_huff_wave.hip) for sanitizers (test tooling).
// The kernel text is compiled as it stands with HOSTSIM_W: its wave-synchronous phases become loops over 64 lanes, its LDS an
// exact-size heap buffer (ASAN sees every overrun).  For every BGZF block of the input files the literal / token / meta output is
//   (1) replayed and checked against the block's CRC32 / ISIZE trailer, and
//   (2) compared word for word with the output of the lane-per-block kernel (phase A of bgzf_inflate.hip, host build of the same
//       kernel text: tools/hostsim/_gen/phaseA_extract.inc), including the error status of damaged blocks.
// usage: sim_wave [--flip N seed] file...
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
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
struct BgzfTable { const uint64_t *coff; const uint32_t *clen; const uint32_t *isize; const uint64_t *uoff; int64_t n; };
#define DHTS_LIT_STRIDE 65536u
#define DHTS_TOK_STRIDE 22528u
#define DHTS_TOK_PURE 511u
struct InflateMeta { uint32_t ntok, nlit, outlen; int32_t status; };
#define DHTS_BLK_OK 0
#define DHTS_BLK_ERR_INFLATE (-3)
#define DHTS_BLK_ERR_CRC (-4)
#define DHTS_BLK_ERR_ISIZE (-5)
#include "phaseA_extract.inc"
#define HOSTSIM_W 1
#include "../../../duckhts_amd/csrc/bgzf_huff_wave.hip"

static uint32_t crct[256];
static int replay(const std::vector<uint8_t> &d, uint64_t coff, uint32_t clen, const uint8_t *L, const uint32_t *T, const InflateMeta &m) {
    std::vector<uint8_t> out; out.reserve(65536);
    uint32_t lp = 0;
    for (uint32_t i = 0; i < m.ntok; i++) {
        const uint32_t t = T[i], run = t >> 23;
        for (uint32_t k = 0; k < run; k++) out.push_back(L[lp++]);
        if (run != DHTS_TOK_PURE) {
            const uint32_t len = ((t >> 15) & 255u) + 3, dist = (t & 0x7fffu) + 1;
            if (dist > out.size()) return 1;
            for (uint32_t k = 0; k < len; k++) out.push_back(out[out.size() - dist]);
        }
    }
    while (lp < m.nlit) out.push_back(L[lp++]);
    uint32_t c = 0xffffffffu; for (uint8_t x : out) c = crct[(c ^ x) & 0xff] ^ (c >> 8);
    c ^= 0xffffffffu;
    const uint8_t *tr = d.data() + coff + clen - 8;
    const uint32_t want_crc = tr[0] | (tr[1] << 8) | (tr[2] << 16) | ((uint32_t)tr[3] << 24), want_len = tr[4] | (tr[5] << 8) | (tr[6] << 16) | ((uint32_t)tr[7] << 24);
    return (out.size() != want_len || out.size() != m.outlen || c != want_crc) ? 1 : 0;
}

int main(int argc, char **argv) {
    for (uint32_t k = 0; k < 256; k++) { uint32_t c = k; for (int j = 0; j < 8; j++) c = (c & 1u) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1); crct[k] = c; }
    int flips = 0; unsigned seed = 1; int a = 1;
    if (argc > 3 && !strcmp(argv[1], "--flip")) { flips = atoi(argv[2]); seed = (unsigned)atoi(argv[3]); a = 4; }
    int rc = 0;
    for (; a < argc; a++) {
        FILE *f = fopen(argv[a], "rb"); if (!f) { fprintf(stderr, "cannot open %s\n", argv[a]); return 2; }
        fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
        std::vector<uint8_t> d(n + 256, 0); if (fread(d.data(), 1, n, f) != (size_t)n) return 2; fclose(f);
        std::vector<uint64_t> coff; std::vector<uint32_t> clen, isz; std::vector<uint64_t> uoff;
        for (long p = 0; p + 18 <= n;) { uint32_t bl = (d[p + 16] | (d[p + 17] << 8)) + 1; if (p + (long)bl > n) break; coff.push_back(p); clen.push_back(bl); isz.push_back(0); uoff.push_back(0); p += bl; }
        int64_t nb = coff.size();
        if (flips) {            // damage payload bits (never the 18-byte headers: the block table stays valid)
            srand(seed);
            for (int k = 0; k < flips; k++) { int64_t b = rand() % nb; if (clen[b] <= 26) continue; uint32_t o = 18 + rand() % (clen[b] - 26); d[coff[b] + o] ^= (uint8_t)(1u << (rand() % 8)); }
        }
        BgzfTable t{coff.data(), clen.data(), isz.data(), uoff.data(), nb};
        std::vector<uint8_t> lit((size_t)nb * DHTS_LIT_STRIDE + 8192), lit2((size_t)nb * DHTS_LIT_STRIDE + 8192);
        std::vector<uint32_t> tok((size_t)nb * DHTS_TOK_STRIDE + 64), tok2((size_t)nb * DHTS_TOK_STRIDE + 64);
        std::vector<InflateMeta> meta(nb), meta2(nb);
        // lane-per-block kernel (every symbol in LDS)
        for (int64_t wg = 0; wg * A_SL < nb; wg++) {
            std::vector<uint8_t> smem(A_LDS_BYTES_FOR(A_NLO_ALL), 0); g_smem = smem.data();
            for (int lane = 0; lane < A_SL; lane++) { blockIdx.x = (unsigned)wg; threadIdx.x = (unsigned)lane; bgzf_huff_decode(d.data(), t, 0, (int32_t)nb, lit.data(), tok.data(), meta.data(), A_NLO_ALL); }
        }
        // wave-per-block kernel (one persistent wave takes every block: its per-lane density memory carries from block to block)
        // SIM_WAVES=k: k persistent waves take the blocks round-robin, as the device's atomic counter would deal them
        const int nwaves = getenv("SIM_WAVES") ? atoi(getenv("SIM_WAVES")) : 1;
        std::vector<uint32_t> dens_all((size_t)64 * nwaves, 0u);
        for (int64_t b = 0; b < nb; b++) {
            uint32_t *dens_w = dens_all.data() + 64 * (size_t)(b % nwaves);
            // (exact-size LDS image and staging slices: ASAN sees every overrun; the fill pattern shows reads of unwritten bytes)
            std::vector<uint8_t> smem(HW_LDS_BYTES, (uint8_t)(getenv("FILL") ? atoi(getenv("FILL")) : 0xAB));
            std::vector<uint8_t> slit(HW_STAGE_LIT_BYTES, 0xCD); std::vector<uint32_t> stok(HW_STAGE_TOK_WORDS, 0xCDCDCDCDu);
            hw_block(smem.data(), b, d.data(), t, lit2.data() + (size_t)b * DHTS_LIT_STRIDE, tok2.data() + (size_t)b * DHTS_TOK_STRIDE, meta2[b], slit.data(), stok.data(), dens_w);
        }
        int bad = 0, mism = 0, differ = 0;
        for (int64_t b = 0; b < nb; b++) {
            const InflateMeta &m1 = meta[b], &m2 = meta2[b];
            if (m1.status != 0 && m2.status == 0 && flips) {
                // the wave kernel leaves the distance-beyond-the-output test to bgzf_lz_resolve: the replay (which makes that test) must fail
                if (!replay(d, coff[b], clen[b], lit2.data() + (size_t)b * DHTS_LIT_STRIDE, tok2.data() + (size_t)b * DHTS_TOK_STRIDE, m2)) { differ++; fprintf(stderr, "block %lld: lane kernel rejects it, the wave kernel's output replays cleanly\n", (long long)b); }
                else bad++;
                continue;
            }
            if ((m1.status != 0) != (m2.status != 0)) { differ++; if (differ < 5) fprintf(stderr, "block %lld: status lane %d wave %d\n", (long long)b, m1.status, m2.status); continue; }
            if (m2.status) { bad++; continue; }
            const uint8_t *L1 = lit.data() + (size_t)b * DHTS_LIT_STRIDE, *L2 = lit2.data() + (size_t)b * DHTS_LIT_STRIDE;
            const uint32_t *T1 = tok.data() + (size_t)b * DHTS_TOK_STRIDE, *T2 = tok2.data() + (size_t)b * DHTS_TOK_STRIDE;
            if (m1.ntok != m2.ntok || m1.nlit != m2.nlit || m1.outlen != m2.outlen || memcmp(L1, L2, m1.nlit) || memcmp(T1, T2, (size_t)m1.ntok * 4)) {
                differ++;
                if (differ < 5) {
                    fprintf(stderr, "block %lld: lane ntok %u nlit %u out %u | wave ntok %u nlit %u out %u\n", (long long)b, m1.ntok, m1.nlit, m1.outlen, m2.ntok, m2.nlit, m2.outlen);
                    for (uint32_t i = 0; i < m1.ntok && i < m2.ntok; i++) if (T1[i] != T2[i]) { fprintf(stderr, "  first token difference at %u: %08x vs %08x\n", i, T1[i], T2[i]); break; }
                    for (uint32_t i = 0; i < m1.nlit && i < m2.nlit; i++) if (L1[i] != L2[i]) { fprintf(stderr, "  first literal difference at %u\n", i); break; }
                }
            }
            if (!flips && replay(d, coff[b], clen[b], L2, T2, m2)) mism++;
        }
        printf("%s: blocks %lld failed %d mismatching %d differing-from-lane-kernel %d (LDS=%d)\n", argv[a], (long long)nb, bad, mism, differ, (int)HW_LDS_BYTES);
        if (mism || differ || (bad && !flips)) rc = 1;
    }
#ifdef HW_STATS
    printf("segments %llu; pass-1 rounds executed (by round index):", g_hw_stat_seg);
    for (int i = 0; i < 8; i++) printf(" %llu", g_hw_stat_p1[i]);
    printf("; lane-decodes in pass 1: %llu; sequential fallbacks %llu\n", g_hw_stat_dirty, g_hw_stat_fallback);
    for (int q = 0; q < 2; q++) printf("pass %d: loop iterations (4 units each) paid by the wave (max over lanes, summed over rounds) %llu; mean over 64 lanes %.1f -> lane utilisation %.3f\n", q, g_hw_stat_itmax[q], (double)g_hw_stat_itsum[q] / 64.0, g_hw_stat_itmax[q] ? (double)g_hw_stat_itsum[q] / 64.0 / (double)g_hw_stat_itmax[q] : 0.0);
    printf("rounds of segments with ranges >= 1024 bits:"); for (int i = 0; i < 8; i++) printf(" %llu", g_hw_stat_big[i]); printf("\n");
    printf("pass 0: %llu boundary proposals, %llu ended on an invalid code (nominal boundary proposed); round 1 found %llu wrong starts, %llu of them nominal\n", g_hw_stat_p0n, g_hw_stat_p0bad, g_hw_stat_wrong, g_hw_stat_wrong_nominal);
#endif
    return rc;
}
