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
    int bad = 0;
    for (int64_t wg = 0; wg * A_SL < nb; wg++) {
        std::vector<uint8_t> smem(A_LDS_BYTES, (uint8_t)(getenv("FILL") ? atoi(getenv("FILL")) : 0));  // exact size: ASAN catches any overrun
        g_smem = smem.data();
#ifdef SIM_THREADS
        std::vector<std::thread> th;
        for (int lane = 0; lane < A_SL; lane++) th.emplace_back([&, lane]() {
            blockIdx.x = (unsigned)wg; threadIdx.x = (unsigned)lane;
            bgzf_huff_decode(d.data(), t, 0, (int32_t)nb, lit.data(), tok.data(), meta.data());
        });
        for (auto &x : th) x.join();
#else
        for (int lane = 0; lane < A_SL; lane++) {
            blockIdx.x = (unsigned)wg; threadIdx.x = (unsigned)lane;
            bgzf_huff_decode(d.data(), t, 0, (int32_t)nb, lit.data(), tok.data(), meta.data());
        }
#endif
    }
    for (int64_t b = 0; b < nb; b++) if (meta[b].status) { bad++; }
    printf("blocks %lld bad %d (SL=%d LDS=%d)\n", (long long)nb, bad, A_SL, (int)A_LDS_BYTES);
    return bad ? 1 : 0;
}
