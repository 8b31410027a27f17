// bgzf_huff_wave.hip -- phase A of the BGZF inflate, ONE WAVE PER BGZF BLOCK (gfx950), written from RFC 1951.
//
// Replaces the Huffman-decoding half of htslib bgzf.c:762-824 (bgzf_uncompress / inflate_block over zlib's inflate).
// Produces exactly what bgzf_huff_decode (bgzf_inflate.hip, one LANE per block) produces -- the literal bytes, one u32 token per
// LZ77 match and the InflateMeta record that bgzf_lz_resolve consumes -- so the two kernels are interchangeable and each is the
// other's cross-check (tests/test_gpu_bam.py, tools/hostsim/sim_wave.cpp).
//
// Why a second kernel: with one lane per block a symbol costs ~250 wave instructions per 64 symbols, but a launch can never
// finish faster than ONE lane decodes ONE whole block (~22 k symbols, 13-15 ms), which is the floor of every small file, index
// window or LIMIT query.  Here the 64 lanes of a wave share one block:
//   * the code-length section of a dynamic block is decoded once, wave-uniformly (scalar unit + a 128-entry table held in two
//     VGPRs and read with v_readlane), and the canonical tables are built by all lanes: ranks by ballot/mbcnt, then a direct
//     lookup table of up to 2^12 u16 entries (literal/length) and 2^9 u32 entries (distance) in LDS, filled in code order so
//     that every lane writes a contiguous range of codes;
//   * the symbol stream [p0, end of payload) is cut into 64 equal bit ranges.  DEFLATE symbols are self-delimiting, so a decoder
//     started at a wrong bit falls into step with the true symbol sequence after a few dozen symbols: every lane first decodes
//     only the last SYNC_W bits before its right neighbour's range to PROPOSE where that neighbour starts (pass 0), then decodes
//     its own range from its proposed start, counting what it will emit (pass 1).  Lane 0 starts at the true position; lane i is
//     CONFIRMED when lane i-1 is confirmed and ends exactly where lane i started.  Lanes behind a broken link restart from their
//     predecessor's end until the chain holds: by induction the confirmed chain IS the serial decode -- nothing is probabilistic
//     about the result, only about how many rounds it takes (one, almost always);
//   * exclusive prefix sums over the per-lane counts give every lane its place in the literal and token streams, and pass 2
//     decodes the ranges once more, storing literals and tokens (same token format as the lane-per-block kernel: a literal run
//     that crosses lane boundaries is carried into the lane that holds the next match).
// A block costs ~2.2 decodes of each symbol at ~45 instructions per 64 symbols, a fraction of the canonical-arithmetic loop, and
// its latency is microseconds.  LDS per wave: 8 KB, so 20 waves share a CU (five per SIMD): the loop is a chain of dependent
// operations with two LDS lookups in it, and what hides that latency is the number of resident waves.
#ifndef BGZF_HUFF_WAVE_HIP
#define BGZF_HUFF_WAVE_HIP
#ifndef HOSTSIM_W
#include "dhts_common.h"
#endif

// ---- execution-model shim ------------------------------------------------------------------------------------------------------
// The kernel body is written as wave-synchronous phases: `W_LANES { ... }` is code every lane runs on its own state (PL(x)),
// everything outside is wave-uniform.  On the device W_LANES is empty and PL(x) is a register; the host simulation
// (tools/hostsim/sim_wave.cpp, ASAN/UBSAN) turns W_LANES into a loop over 64 lanes and PL(x) into x[lane], which is exact because
// lanes only communicate through LDS arrays and the three collectives below, always across a W_SYNC().
#ifdef HOSTSIM_W
#define W_LANES for (int lane = 0; lane < 64; lane++)
#define PL(x) x[lane]
#define PLD(type, x) type x[64]
#define W_SYNC() do { } while (0)
#define W_UNI(x) (x)
#define W_BALLOT(maskvar, expr) do { maskvar = 0; for (int lane = 0; lane < 64; lane++) if (expr) maskvar |= 1ull << lane; } while (0)
#define W_EXCL_SCAN(dst, src, total) do { uint32_t run_ = 0; for (int lane = 0; lane < 64; lane++) { const uint32_t v_ = src[lane]; dst[lane] = run_; run_ += v_; } total = run_; } while (0)
#define W_LANE_DECL
static inline uint32_t w_brev32(uint32_t v) { uint32_t r = 0; for (int i = 0; i < 32; i++) r |= ((v >> i) & 1u) << (31 - i); return r; }
static inline uint32_t w_popc64(uint64_t v) { return (uint32_t)__builtin_popcountll(v); }
static inline int w_ctz64(uint64_t v) { return v ? __builtin_ctzll(v) : 64; }
static inline int w_msb64(uint64_t v) { return 63 - __builtin_clzll(v); }
#define W_DEV static inline
#else
#define W_LANES
#define PL(x) x
#define PLD(type, x) type x
#define W_SYNC() __syncthreads()
#define W_UNI(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
#define W_BALLOT(maskvar, expr) do { maskvar = __ballot(expr); } while (0)
#define W_EXCL_SCAN(dst, src, total) do { const uint32_t v_ = (src); const uint32_t i_ = wave_incl_scan(v_, lane); dst = i_ - v_; total = RDLANE(i_, 63); } while (0)
#define W_LANE_DECL const int lane = threadIdx.x;
__device__ __forceinline__ uint32_t w_brev32(uint32_t v) { return __brev(v); }
__device__ __forceinline__ uint32_t w_popc64(uint64_t v) { return (uint32_t)__popcll(v); }
__device__ __forceinline__ int w_ctz64(uint64_t v) { return v ? __ffsll((unsigned long long)v) - 1 : 64; }
__device__ __forceinline__ int w_msb64(uint64_t v) { return 63 - __clzll((long long)v); }
#define W_DEV __device__ __forceinline__
#endif

// ---- geometry --------------------------------------------------------------------------------------------------------------------
#ifndef HW_RLL
#define HW_RLL 11u                 /* root bits of the literal/length table (u16 entries: 4 KB) */
#endif
#ifndef HW_RD
#define HW_RD 8u                   /* root bits of the distance table (u32 entries: 1 KB) */
#endif
#ifndef HW_SYNC_W
#define HW_SYNC_W 768u             /* bits decoded ahead of a range boundary to propose the neighbour's start */
#endif
#define HW_MIN_S 256u              /* a sub-stream is at least this many bits (short tails use fewer lanes) */
#define HW_STAGE 1024u             /* bytes of compressed data staged for the wave-uniform header reader */
// LDS image of one wave
#define HW_OFF_LL 0u                                     /* u16 [4096] */
#define HW_OFF_D (HW_OFF_LL + (2u << HW_RLL))            /* u32 [512]  */
#define HW_OFF_STAGE (HW_OFF_D + (4u << HW_RD))          /* u8 [1024 + 8] header bytes; the lane exchange arrays reuse this space once the header is read */
#define HW_OFF_LENS (HW_OFF_STAGE + 6u * 256u)           /* u8 [320] code lengths: literal/length then distance */
#define HW_OFF_SLL (HW_OFF_LENS + 320u)                  /* u16 [288] literal/length entries (without the code length) in canonical order */
#define HW_OFF_SD (HW_OFF_SLL + 576u)                    /* u32 [32]  distance entries in canonical order */
#define HW_OFF_TAB (HW_OFF_SD + 128u)                    /* u32 [2][3][16]: per alphabet limit15 / first / offs by code length */
#define HW_OFF_X HW_OFF_STAGE                            /* u32 [6][64] lane exchange arrays */
#define HW_LDS_BYTES (HW_OFF_TAB + 384u)
// exchange arrays
#define HX_START 0
#define HX_END 1
#define HX_FLAG 2
#define HX_A 3
#define HX_B 4
#define HX_C 5

// literal/length entry (u16): [3:0] code length (0 = no symbol: bit 4 set -> code longer than the root, else invalid), [6:4] extra bits,
//   bit 7 = length code, [15:8] literal byte or length base - 3; end of block = length code with extra-bit count 7 (0x00F0)
// distance entry (u32): [3:0] code length (0 as above), [7:4] extra bits, [31:16] base
// (the two formats share the positions of the code length and of the extra-bit count, so one extraction serves both alphabets)
#define HW_LONG 0x10u
#define HW_EOB 0x00F0u
W_DEV uint32_t hw_ll_entry(uint32_t sym) {
    if (sym < 256u) return sym << 8;
    if (sym == 256u) return HW_EOB;
    const uint32_t j = sym - 257u;
    if (j >= 29u) return 0xffffffffu;                      // 286, 287: never valid in a stream (RFC 1951 3.2.6)
    if (j == 28u) return 0x80u | (255u << 8);              // length 258, no extra bits
    if (j < 8u) return 0x80u | (j << 8);
    const uint32_t x = (j >> 2) - 1u;
    const uint32_t base = 3u + ((4u | (j & 3u)) << x);
    return 0x80u | (x << 4) | ((base - 3u) << 8);
}
W_DEV uint32_t hw_d_entry(uint32_t sym) {
    if (sym >= 30u) return 0xffffffffu;                    // 30, 31: never valid
    if (sym < 4u) return (sym + 1u) << 16;
    const uint32_t x = (sym >> 1) - 1u;
    const uint32_t base = 1u + ((2u | (sym & 1u)) << x);
    return (base << 16) | (x << 4);
}

#if defined(HOSTSIM_W) && defined(HW_STATS)
static unsigned long long g_hw_stat_p1[8], g_hw_stat_dirty, g_hw_stat_seg;
#endif
// -DHW_DIAG (device builds for tools/dbg/hw_diag.py): cycles per phase, summed over blocks by lane 0
#if defined(HW_DIAG) && !defined(HOSTSIM_W)
__device__ unsigned long long g_hw_diag[16];   // 0 header 1 tables 2 pass0 3 pass1 4 scans 5 pass2 6 total 7 blocks 8 segments 9 pass-1 rounds 10 stored
#define HWD_T(v) const unsigned long long v = clock64()
#define HWD_ADD(i, a, b) do { hwd[i] += (b) - (a); } while (0)
#define HWD_CNT(i, n) do { hwd[i] += (n); } while (0)
#else
#define HWD_T(v) do { } while (0)
#define HWD_ADD(i, a, b) do { } while (0)
#define HWD_CNT(i, n) do { } while (0)
#endif
// per-lane results of the counting pass
struct HwLane {
    uint32_t start, end;       // bit positions (relative to the aligned payload base): first unit / one past the last unit of this lane
    uint32_t flags;            // 1 = ended on the end-of-block symbol, 2 = invalid code / ran past the payload
    uint32_t nlit, nmatch, lead, tail, pint, outb;
};
#define HWF_EOB 1u
#define HWF_BAD 2u

// (hi:lo) >> n for n < 32
#ifdef HOSTSIM_W
static inline uint32_t hw_shr64lo(uint32_t hi, uint32_t lo, uint32_t n) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> n); }
#else
__device__ __forceinline__ uint32_t hw_shr64lo(uint32_t hi, uint32_t lo, uint32_t n) { return __builtin_amdgcn_alignbit(hi, lo, n); }
#endif

// code longer than the table's root (or no code at all): canonical arithmetic on the left-justified 15-bit prefix; 0 = invalid
W_DEV uint32_t hw_long_code(const uint8_t *smem, uint32_t bits, uint32_t mode, uint32_t root) {
    const uint32_t *tab = (const uint32_t *)(smem + HW_OFF_TAB) + 48u * mode;
    const uint32_t w15 = w_brev32(bits) >> 17;
    uint32_t L = root + 1u;
    while (L <= 15u && w15 >= tab[L]) L++;
    if (L > 15u) return 0u;
    const uint32_t si = tab[32 + L] + ((w15 >> (15u - L)) - tab[16 + L]);
    if (mode) { const uint32_t v = si < 32u ? ((const uint32_t *)(smem + HW_OFF_SD))[si] : 0xffffffffu; return v == 0xffffffffu ? 0u : (v | L); }
    const uint32_t v = si < 288u ? ((const uint16_t *)(smem + HW_OFF_SLL))[si] : 0xffffu;
    return v == 0xffffu ? 0u : (v | L);
}

// the four literal/length code lengths above the root (HW_RLL + 1 .. 15 when HW_RLL is 11): limits, first codes and offsets, wave-uniform
struct HwLong { uint32_t l0, l1, l2, l3, f0, f1, f2, f3, o0, o1, o2, o3; };
W_DEV uint32_t hw_long_ll(const uint8_t *smem, uint32_t bits, const HwLong q) {
    const uint32_t w15 = w_brev32(bits) >> 17;
    const uint32_t c0 = w15 >= q.l0, c1 = w15 >= q.l1, c2 = w15 >= q.l2;
    if (w15 >= q.l3) return 0u;
    const uint32_t L = HW_RLL + 1u + c0 + c1 + c2;
    const uint32_t f = c2 ? q.f3 : c1 ? q.f2 : c0 ? q.f1 : q.f0;
    const uint32_t o = c2 ? q.o3 : c1 ? q.o2 : c0 ? q.o1 : q.o0;
    const uint32_t si = o + ((w15 >> (15u - L)) - f);
    const uint32_t v = si < 288u ? ((const uint16_t *)(smem + HW_OFF_SLL))[si] : 0xffffu;
    return v == 0xffffu ? 0u : (v | L);
}

// One lane decodes the units that START in [start, stop) with the tables in LDS: ONE Huffman symbol of either alphabet per iteration
// (a wave always holds lanes in both states), written as straight-line selects: the only branches are the refill of the bit buffer,
// the rare long code, and -- in the emitting pass -- the stores.
// PASS 0: nothing is counted (proposal of the neighbour's start);  PASS 1: counts;  PASS 2: emits literals and tokens.
template <int PASS>
W_DEV void hw_span(const uint8_t *smem, const uint32_t *in32, uint32_t start, uint32_t stop, uint32_t limit_bits, uint32_t mask_ll, uint32_t mask_d, uint32_t rll, uint32_t rd, const HwLong lq,
                   HwLane &r, uint8_t *lit, uint32_t *tok, uint32_t lit_at, uint32_t tok_at, uint32_t out_at, uint32_t run_in, int32_t &status) {
    const uint16_t *lut_ll = (const uint16_t *)(smem + HW_OFF_LL);
    const uint32_t *lut_d = (const uint32_t *)(smem + HW_OFF_D);
    // input: the lane reads its range 16 bytes at a time (each 64-byte line is fetched by four loads instead of sixteen), one chunk
    // ahead of the chunk it is consuming, so a load has four refills (~16 symbols) to arrive before anything waits for it
    uint32_t lo, hi = 0, cnt, q0, q1, q2, left;
    uint4 nx;                                                 // the chunk in flight
    const uint32_t *qp;
    {
        const uint32_t w = start >> 5, o = start & 31u;
        uint4 c0; __builtin_memcpy(&c0, in32 + w, 16);
        lo = c0.x >> o; cnt = 32u - o; q0 = c0.y; q1 = c0.z; q2 = c0.w; left = 3;
        qp = in32 + w + 4; __builtin_memcpy(&nx, qp, 16);
    }
    uint32_t pos = start, mode = 0, want = 0, flags = 0;
    uint32_t nlit = 0, nmatch = 0, lead = 0, run = (PASS == 2) ? run_in : 0u, pint = 0, outb = 0;
    uint32_t litw = 0, l0 = 0, l1 = 0, l2 = 0;               // PASS 2: literal bytes not yet stored: (nlit >> 2) & 3 whole words, then the low (nlit & 3) bytes of litw
    uint32_t ntk = 0, t0 = 0, t1 = 0, t2 = 0;                // PASS 2: tokens of this lane so far; the last ntk & 3 are not yet stored
    bool live = pos < stop;
    while (live) {
        if (cnt < 32u) {
            if (left == 0u) { q0 = nx.x; q1 = nx.y; q2 = nx.z; const uint32_t q3 = nx.w; qp += 4; __builtin_memcpy(&nx, qp, 16); lo |= q0 << cnt; hi |= (q0 >> 1) >> (31u - cnt); q0 = q1; q1 = q2; q2 = q3; left = 3; }
            else { lo |= q0 << cnt; hi |= (q0 >> 1) >> (31u - cnt); q0 = q1; q1 = q2; left--; }
            cnt += 32u;
        }
        const uint32_t bits = lo;
        const uint32_t e0 = lut_ll[bits & mask_ll], e1 = lut_d[bits & mask_d];
        uint32_t e = mode ? e1 : e0;
        if ((e & 15u) == 0u) {
            // (a literal/length code above the root can only exist when the root is HW_RLL: four lengths, limits in registers)
            e = !(e & HW_LONG) ? 0u : mode ? hw_long_code(smem, bits, 1u, rd) : (HW_RLL == 11u && rll == HW_RLL) ? hw_long_ll(smem, bits, lq) : hw_long_code(smem, bits, 0u, rll);
            if ((e & 15u) == 0u) { flags |= HWF_BAD; e = 0x0001u; }
        }
        const uint32_t L = e & 15u;
        const uint32_t islen = mode ? 0u : (e >> 7) & 1u;
        const uint32_t xr = (e >> 4) & (mode ? 15u : 7u);
        const uint32_t iseob = (islen && xr == 7u) ? 1u : 0u;
        const uint32_t x = iseob ? 0u : xr;
        const uint32_t val = (mode ? (e >> 16) : ((e >> 8) & 255u) + 3u) + ((bits >> L) & ((1u << x) - 1u));
        const uint32_t use = L + x;
        const uint32_t islit = (mode | islen) ^ 1u;
        if (PASS == 2) {
            // literals and tokens leave the lane 16 bytes at a time (one store per 16 literals / 4 matches: stores are counted by the
            // same vmcnt as the input loads, and a scattered 4-byte store per lane costs the memory pipeline a transaction per lane)
            if (islit) {
                litw |= ((e >> 8) & 255u) << (8u * (nlit & 3u));
                if ((nlit & 3u) == 3u) {
                    const uint32_t qd = (nlit >> 2) & 3u;
                    if (qd == 0u) l0 = litw; else if (qd == 1u) l1 = litw; else if (qd == 2u) l2 = litw;
                    else { const uint4 v16 = make_uint4(l0, l1, l2, litw); __builtin_memcpy(lit + lit_at + nlit - 15u, &v16, 16); }
                    litw = 0;
                }
            }
            if (mode) {
                // the stream position of this match is out_at + outb: a distance may not reach before the start of the block
                if (val > out_at + outb) { status = DHTS_BLK_ERR_INFLATE; flags |= HWF_BAD; }
                else {
                    uint32_t tv = DHTS_TOK_PURE << 23;
                    for (;;) {
                        const bool pure = run >= DHTS_TOK_PURE;
                        if (!pure) tv = (run << 23) | ((want - 3u) << 15) | (val - 1u); else run -= DHTS_TOK_PURE;
                        const uint32_t qd = ntk & 3u;
                        if (qd == 0u) t0 = tv; else if (qd == 1u) t1 = tv; else if (qd == 2u) t2 = tv;
                        else { const uint4 v16 = make_uint4(t0, t1, t2, tv); __builtin_memcpy(tok + tok_at + ntk - 3u, &v16, 16); }
                        ntk++;
                        if (!pure) break;
                    }
                }
            }
        }
        if (PASS != 0) {
            if (PASS == 1 && mode && nmatch && run >= DHTS_TOK_PURE) pint += run / DHTS_TOK_PURE;
            lead = (mode && nmatch == 0u) ? run : lead;
            nlit += islit; outb += mode ? want : islit;
            nmatch += mode; run = mode ? 0u : run + islit;
        }
        want = islen ? val : want;
        mode = islen & (iseob ^ 1u);
        lo = hw_shr64lo(hi, lo, use); hi >>= use; cnt -= use; pos += use;
        flags |= iseob ? HWF_EOB : 0u;
        flags |= (pos > limit_bits + 64u) ? HWF_BAD : 0u;            // ran off the payload (a true stream never does)
        live = flags == 0u && (mode != 0u || pos < stop);
    }
    if (PASS == 2) {
        const uint32_t wq = (nlit >> 2) & 3u, base = lit_at + (nlit & ~15u);
        if (wq > 0u) __builtin_memcpy(lit + base, &l0, 4);
        if (wq > 1u) __builtin_memcpy(lit + base + 4u, &l1, 4);
        if (wq > 2u) __builtin_memcpy(lit + base + 8u, &l2, 4);
        for (uint32_t k = nlit & ~3u; k < nlit; k++) lit[lit_at + k] = (uint8_t)(litw >> (8u * (k & 3u)));
        const uint32_t tq = ntk & 3u, tb_ = tok_at + (ntk & ~3u);
        if (tq > 0u) tok[tb_] = t0;
        if (tq > 1u) tok[tb_ + 1u] = t1;
        if (tq > 2u) tok[tb_ + 2u] = t2;
    }
    r.end = pos; r.flags = flags;
    if (PASS == 1) { r.nlit = nlit; r.nmatch = nmatch; r.lead = nmatch ? lead : nlit; r.tail = nmatch ? run : 0u; r.pint = pint; r.outb = outb; }
}

// ---- wave-uniform header reader over the staged bytes ---------------------------------------------------------------------------
struct HwHdr { uint64_t buf; uint32_t cnt, widx; const uint32_t *words; };
W_DEV void hw_hdr_fill(HwHdr &h) { if (h.cnt <= 32u) { h.buf |= (uint64_t)W_UNI(h.words[h.widx]) << h.cnt; h.widx++; h.cnt += 32u; } }
W_DEV uint32_t hw_hdr_take(HwHdr &h, uint32_t n) { hw_hdr_fill(h); const uint32_t v = (uint32_t)h.buf & ((1u << n) - 1u); h.buf >>= n; h.cnt -= n; return v; }

#ifdef HOSTSIM_W
static void bgzf_huff_decode_wave_body(uint8_t *smem, int blk_in_grid, const uint8_t *comp, BgzfTable tab, int64_t blk0, int32_t nblk,
                                       uint8_t *lit_all, uint32_t *tok_all, InflateMeta *meta)
#else
extern "C" __global__ void __launch_bounds__(64)
bgzf_huff_decode_wave(const uint8_t *__restrict__ comp, BgzfTable tab, int64_t blk0, int32_t nblk,
                      uint8_t *__restrict__ lit_all, uint32_t *__restrict__ tok_all, InflateMeta *__restrict__ meta)
#endif
{
#ifndef HOSTSIM_W
    __shared__ __attribute__((aligned(16))) uint8_t smem[HW_LDS_BYTES];
    const int blk_in_grid = blockIdx.x;
#endif
    W_LANE_DECL
    if (blk_in_grid >= nblk) return;
    const int64_t s = blk_in_grid;
    const int64_t bi = blk0 + s;
    const uint32_t clen = tab.clen[bi];
    uint8_t *lit = lit_all + (size_t)s * DHTS_LIT_STRIDE;
    uint32_t *tok = tok_all + (size_t)s * DHTS_TOK_STRIDE;
    uint16_t *lut_ll = (uint16_t *)(smem + HW_OFF_LL);
    uint32_t *lut_d = (uint32_t *)(smem + HW_OFF_D);
    uint8_t *stage = smem + HW_OFF_STAGE;
    uint8_t *lens = smem + HW_OFF_LENS;
    uint16_t *sll = (uint16_t *)(smem + HW_OFF_SLL);
    uint32_t *sd = (uint32_t *)(smem + HW_OFF_SD);
    uint32_t *tb = (uint32_t *)(smem + HW_OFF_TAB);          // [0..15] ll limit15, [16..31] ll first, [32..47] ll offs, [48..] the same for distances
    uint32_t *xch = (uint32_t *)(smem + HW_OFF_X);

#if defined(HW_DIAG) && !defined(HOSTSIM_W)
    unsigned long long hwd[16]; for (int q_ = 0; q_ < 16; q_++) hwd[q_] = 0;
#endif
    HWD_T(t_begin);
    int32_t status = clen >= 26u ? 0 : DHTS_BLK_ERR_INFLATE;
    // bit positions are relative to the 4-byte aligned address at or below the first payload byte
    const uint64_t pay0 = tab.coff[bi] + 18u;
    const uint32_t *in32 = (const uint32_t *)(comp + (pay0 & ~(uint64_t)3));
    const uint32_t bit0 = (uint32_t)(pay0 & 3u) * 8u;
    const uint32_t limit_bits = bit0 + (clen >= 26u ? (clen - 26u) * 8u : 0u);   // one past the last payload bit
    uint32_t pos = bit0;                                     // wave-uniform stream position
    uint32_t nlit_tot = 0, ntok_tot = 0, outpos = 0, run = 0;
    bool last = (status != 0);

    while (!last && status == 0) {
        HWD_T(t_h0);
        // ---- stage the next HW_STAGE bytes for the uniform reader ----
        {
            // (bytes beyond the block's trailer are never needed: a lane whose piece starts there stages zeros instead of reading on)
            const uint32_t w0 = pos >> 5, wend = (limit_bits >> 5) + 3u;
            W_LANES {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (w0 + 4u * (uint32_t)lane < wend) __builtin_memcpy(&v, in32 + w0 + 4 * lane, 16);
                __builtin_memcpy(stage + 16 * lane, &v, 16);
            }
            W_SYNC();
        }
        HwHdr h; h.words = (const uint32_t *)stage; h.buf = 0; h.cnt = 0; h.widx = 0;
        { const uint32_t o = pos & 31u; hw_hdr_fill(h); h.buf >>= o; h.cnt -= o; }
        uint32_t hpos = pos;                                 // position of the next unread header bit
        last = hw_hdr_take(h, 1) != 0u;
        const uint32_t type = hw_hdr_take(h, 2);
        hpos += 3;
        if (type == 3u) { status = DHTS_BLK_ERR_INFLATE; break; }
        if (type == 0u) {
            // ---- stored block (RFC 1951 3.2.4): LEN bytes go to the literal stream ----
            hpos = (hpos + 7u) & ~7u;
            if (hpos + 32u > limit_bits) { status = DHTS_BLK_ERR_INFLATE; break; }
            const uint8_t *p8 = (const uint8_t *)in32 + (hpos >> 3);
            const uint32_t len = (uint32_t)p8[0] | ((uint32_t)p8[1] << 8), nlen = (uint32_t)p8[2] | ((uint32_t)p8[3] << 8);
            if ((len ^ 0xffffu) != nlen) { status = DHTS_BLK_ERR_INFLATE; break; }
            if (hpos + 32u + 8u * len > limit_bits || outpos + len > 65536u) { status = DHTS_BLK_ERR_INFLATE; break; }
            W_LANES { for (uint32_t k = (uint32_t)lane; k < len; k += 64u) lit[nlit_tot + k] = p8[4 + k]; }
            nlit_tot += len; outpos += len; run += len;
            const uint32_t q = run / DHTS_TOK_PURE;
            W_LANES { for (uint32_t k = (uint32_t)lane; k < q; k += 64u) tok[ntok_tot + k] = DHTS_TOK_PURE << 23; }
            ntok_tot += q; run -= q * DHTS_TOK_PURE;
            pos = hpos + 32u + 8u * len;
            continue;
        }
        // ---- code lengths (RFC 1951 3.2.6 / 3.2.7) ----
        uint32_t nl, nd;
        W_LANES { for (uint32_t k = (uint32_t)lane; k < 80u; k += 64u) ((uint32_t *)lens)[k] = 0u; }
        W_SYNC();
        if (type == 1u) {
            nl = 288; nd = 32;
            W_LANES { for (uint32_t k = (uint32_t)lane; k < 320u; k += 64u) lens[k] = (uint8_t)(k < 144u ? 8 : k < 256u ? 9 : k < 280u ? 7 : k < 288u ? 8 : 5); }
            W_SYNC();
        } else {
            nl = hw_hdr_take(h, 5) + 257u; nd = hw_hdr_take(h, 5) + 1u;
            const uint32_t nc = hw_hdr_take(h, 4) + 4u;
            hpos += 14;
            if (nl > 286u || nd > 30u) { status = DHTS_BLK_ERR_INFLATE; break; }
            // the 19 code-length code lengths, packed 3 bits each in the order of the RFC
            uint64_t clpack = 0;                              // 3 bits per symbol, symbol-indexed
            {
                const int ORD[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
#pragma unroll
                for (int i = 0; i < 19; i++) if ((uint32_t)i < nc) clpack |= (uint64_t)hw_hdr_take(h, 3) << (3 * ORD[i]);
                hpos += 3u * nc;
            }
            // canonical code of the code-length alphabet (max length 7) -> 128-entry direct table, entry = len | sym << 3, 0 = invalid.
            // Lane j builds entries j and j + 64 (two VGPRs on the device; the reader fetches an entry with v_readlane).
            uint32_t cfirst[8], ccount[8], coffs[8];
            {
                for (int L = 0; L < 8; L++) ccount[L] = 0;
                for (uint32_t i = 0; i < 19u; i++) ccount[(clpack >> (3u * i)) & 7u]++;
                uint32_t first = 0, offs = 0; int left = 1;
                for (int L = 1; L <= 7; L++) { left = (left << 1) - (int)ccount[L]; cfirst[L] = first; coffs[L] = offs; offs += ccount[L]; first = (first + ccount[L]) << 1; }
                if (left != 0) { status = DHTS_BLK_ERR_INFLATE; break; }      // the code-length code must be complete
            }
            PLD(uint32_t, cl0); PLD(uint32_t, cl1);
            W_LANES {
                for (int half = 0; half < 2; half++) {
                    const uint32_t idx = (uint32_t)lane + 64u * half;        // 7 stream bits, first code bit in bit 0
                    uint32_t ent = 0, code = 0;
                    for (uint32_t L = 1; L <= 7u && !ent; L++) {
                        code = (code << 1) | ((idx >> (L - 1u)) & 1u);
                        const uint32_t c = code - cfirst[L];
                        if (code >= cfirst[L] && c < ccount[L]) {
                            // the c-th symbol (in symbol order) among those of length L
                            uint32_t seen = 0;
                            for (uint32_t sy = 0; sy < 19u; sy++) if (((clpack >> (3u * sy)) & 7u) == L) { if (seen == c) { ent = L | (sy << 3); break; } seen++; }
                        }
                    }
                    if (half == 0) PL(cl0) = ent; else PL(cl1) = ent;
                }
            }
            (void)coffs;
#ifdef HOSTSIM_W
#define HW_CL_LOOKUP(i) ((i) < 64u ? cl0[(i)] : cl1[(i) - 64u])
#else
#define HW_CL_LOOKUP(i) ((i) < 64u ? (uint32_t)__builtin_amdgcn_readlane((int)cl0, (int)(i)) : (uint32_t)__builtin_amdgcn_readlane((int)cl1, (int)((i) - 64u)))
#endif
            const uint32_t total = nl + nd;
            uint32_t idx = 0, prev = 0;
            while (idx < total) {
                hw_hdr_fill(h);
                if (hpos > limit_bits) { status = DHTS_BLK_ERR_INFLATE; break; }
                const uint32_t ent = HW_CL_LOOKUP((uint32_t)h.buf & 127u);
                const uint32_t L = ent & 7u, sym = ent >> 3;
                if (L == 0u) { status = DHTS_BLK_ERR_INFLATE; break; }
                h.buf >>= L; h.cnt -= L; hpos += L;
                uint32_t val, rep;
                if (sym < 16u) { val = sym; rep = 1; }
                else if (sym == 16u) { if (idx == 0u) { status = DHTS_BLK_ERR_INFLATE; break; } val = prev; rep = 3u + hw_hdr_take(h, 2); hpos += 2; }
                else if (sym == 17u) { val = 0; rep = 3u + hw_hdr_take(h, 3); hpos += 3; }
                else { val = 0; rep = 11u + hw_hdr_take(h, 7); hpos += 7; }
                if (idx + rep > total) { status = DHTS_BLK_ERR_INFLATE; break; }
                prev = val;
                if (val != 0u) {
                    // distance lengths live behind the 288 literal/length slots
                    W_LANES { if ((uint32_t)lane < rep) { const uint32_t i = idx + (uint32_t)lane; lens[i < nl ? i : 288u + (i - nl)] = (uint8_t)val; } }
                }
                idx += rep;
            }
            if (status != 0) break;
            if ((uint32_t)(hpos - pos) > HW_STAGE * 8u - 64u) { status = DHTS_BLK_ERR_INFLATE; break; }     // (cannot happen: a header is < 4,600 bits)
            W_SYNC();
        }
        if (status == 0 && hpos > limit_bits) { status = DHTS_BLK_ERR_INFLATE; break; }
        HWD_T(t_h1); HWD_ADD(0, t_h0, t_h1);

        // ---- canonical tables ----
        // per alphabet: count by length (ballots), limits / first codes / offsets (uniform), rank of every symbol among the symbols of
        // its length (mbcnt) -> entries in canonical order, then the direct tables in code order.
        uint32_t rll = 0, rd = 0;
        for (int alpha = 0; alpha < 2 && status == 0; alpha++) {
            const uint32_t nsym = alpha ? 32u : 288u, base = alpha ? 288u : 0u, rounds = alpha ? 1u : 5u;
            uint32_t cnt[16]; for (int L = 0; L < 16; L++) cnt[L] = 0;
            for (uint32_t rr = 0; rr < rounds; rr++) {
                for (uint32_t L = 1; L <= 15u; L++) { uint64_t m; W_BALLOT(m, (64u * rr + (uint32_t)lane < nsym) && lens[base + 64u * rr + (uint32_t)lane] == L); cnt[L] += w_popc64(m); }
            }
            uint32_t nz = 0, maxlen = 0; int left = 1; uint32_t first = 0, offs = 0;
            uint32_t t_first[16], t_offs[16], t_lim[16];
            t_first[0] = t_offs[0] = t_lim[0] = 0;
            for (uint32_t L = 1; L <= 15u; L++) {
                left = (left << 1) - (int)cnt[L]; nz += cnt[L]; if (cnt[L]) maxlen = L;
                t_first[L] = first; t_offs[L] = offs; t_lim[L] = (first + cnt[L]) << (15u - L);
                offs += cnt[L]; first = (first + cnt[L]) << 1;
            }
            bool has_eob = true;
            if (!alpha) has_eob = lens[256] != 0;
            if (!alpha) { if (left < 0 || (left > 0 && nz != 1u) || !has_eob) { status = DHTS_BLK_ERR_INFLATE; break; } }
            else { if (left < 0 || (left > 0 && nz > 1u)) { status = DHTS_BLK_ERR_INFLATE; break; } }
            W_LANES { if (lane < 16) { tb[48 * alpha + lane] = t_lim[lane]; tb[48 * alpha + 16 + lane] = t_first[lane]; tb[48 * alpha + 32 + lane] = t_offs[lane]; } }
            // ranks -> canonical order
            uint32_t seen[16]; for (int L = 0; L < 16; L++) seen[L] = 0;
            for (uint32_t rr = 0; rr < rounds; rr++) {
                for (uint32_t L = 1; L <= 15u; L++) {
                    if (cnt[L] == 0u) continue;
                    uint64_t m; W_BALLOT(m, (64u * rr + (uint32_t)lane < nsym) && lens[base + 64u * rr + (uint32_t)lane] == L);
                    W_LANES {
                        if ((m >> lane) & 1ull) {
                            const uint32_t rank = seen[L] + w_popc64(m & ((1ull << lane) - 1ull));
                            const uint32_t sym = 64u * rr + (uint32_t)lane;
                            if (!alpha) sll[t_offs[L] + rank] = (uint16_t)hw_ll_entry(sym); else sd[t_offs[L] + rank] = hw_d_entry(sym);
                        }
                    }
                    seen[L] += w_popc64(m);
                }
            }
            W_SYNC();
            // direct table: R root bits; lane owns the codes w in [lane * chunk, (lane + 1) * chunk) (MSB-first prefixes), entry index = bit reversal
            const uint32_t R = maxlen < (alpha ? HW_RD : HW_RLL) ? (maxlen ? maxlen : 1u) : (alpha ? HW_RD : HW_RLL);
            if (alpha) rd = R; else rll = R;
            const uint32_t size = 1u << R, chunk = size >= 64u ? size >> 6 : 1u;
            W_LANES {
                if ((uint32_t)lane * chunk < size) {
                    uint32_t L = 1;
                    for (uint32_t k = 0; k < chunk; k++) {
                        const uint32_t w = (uint32_t)lane * chunk + k, w15 = w << (15u - R);
                        while (L <= R && w15 >= tb[48 * alpha + L]) L++;
                        uint32_t ent;
                        if (L <= R) {
                            const uint32_t si = tb[48 * alpha + 32 + L] + ((w15 >> (15u - L)) - tb[48 * alpha + 16 + L]);
                            if (!alpha) { const uint32_t v = sll[si]; ent = (v == 0xffffu) ? 0u : (v | L); }
                            else { const uint32_t v = sd[si]; ent = (v == 0xffffffffu) ? 0u : (v | L); }
                        } else ent = (w15 < tb[48 * alpha + 15]) ? HW_LONG : 0u;           // a longer code starts with this prefix / unused code space
                        const uint32_t ix = w_brev32(w) >> (32u - R);
                        if (!alpha) lut_ll[ix] = (uint16_t)ent; else lut_d[ix] = ent;
                    }
                }
            }
            W_SYNC();
        }
        if (status != 0) break;
        const uint32_t mask_ll = (1u << rll) - 1u, mask_d = (1u << rd) - 1u;
        HwLong lq;
        {
            const uint32_t a0 = HW_RLL + 1u < 15u ? HW_RLL + 1u : 15u, a1 = HW_RLL + 2u < 15u ? HW_RLL + 2u : 15u, a2 = HW_RLL + 3u < 15u ? HW_RLL + 3u : 15u;
            lq.l0 = W_UNI(tb[a0]); lq.l1 = W_UNI(tb[a1]); lq.l2 = W_UNI(tb[a2]); lq.l3 = W_UNI(tb[15]);
            lq.f0 = W_UNI(tb[16 + a0]); lq.f1 = W_UNI(tb[16 + a1]); lq.f2 = W_UNI(tb[16 + a2]); lq.f3 = W_UNI(tb[16 + 15]);
            lq.o0 = W_UNI(tb[32 + a0]); lq.o1 = W_UNI(tb[32 + a1]); lq.o2 = W_UNI(tb[32 + a2]); lq.o3 = W_UNI(tb[32 + 15]);
        }
        HWD_T(t_h2); HWD_ADD(1, t_h1, t_h2);

        // ---- symbols: segments of up to 64 bit ranges until the end-of-block symbol ----
        uint32_t p0 = hpos;
        bool eob_seen = false;
        while (!eob_seen && status == 0) {
            if (p0 > limit_bits) { status = DHTS_BLK_ERR_INFLATE; break; }
            const uint32_t span = limit_bits - p0;
            uint32_t S = (span + 63u) / 64u; if (S < HW_MIN_S) S = HW_MIN_S;
            uint32_t nlanes = (span + S - 1u) / S; if (nlanes < 1u) nlanes = 1u;            // <= 64
#if defined(HOSTSIM_W) && defined(HW_STATS)
            g_hw_stat_seg++;
#endif
            // pass 0: propose the start of lane i + 1 from the last HW_SYNC_W bits of range i
            HWD_T(t_s0); HWD_CNT(8, 1);
            PLD(HwLane, ln);
            W_LANES {
                PL(ln).start = p0 + (uint32_t)lane * S; PL(ln).end = 0; PL(ln).flags = 0;
                xch[HX_START * 64 + lane] = p0 + (uint32_t)lane * S;
            }
            W_SYNC();
            W_LANES {
                if ((uint32_t)lane + 1u < nlanes) {
                    const uint32_t bnd = p0 + ((uint32_t)lane + 1u) * S;
                    const uint32_t from = bnd - p0 > HW_SYNC_W + (uint32_t)lane * S ? bnd - HW_SYNC_W : p0 + (uint32_t)lane * S;
                    HwLane tmp; int32_t st_ = 0;
                    hw_span<0>(smem, in32, from, bnd, limit_bits, mask_ll, mask_d, rll, rd, lq, tmp, nullptr, nullptr, 0, 0, 0, 0, st_);
                    if (tmp.flags == 0u) xch[HX_START * 64 + lane + 1] = tmp.end;
                }
            }
            W_SYNC();
            W_LANES { PL(ln).start = xch[HX_START * 64 + lane]; }
            HWD_T(t_s1); HWD_ADD(2, t_s0, t_s1);
            // pass 1 until the chain of confirmed lanes reaches the end-of-block symbol or the last lane
            uint64_t dirty = nlanes >= 64u ? ~0ull : ((1ull << nlanes) - 1ull);
            uint32_t n_ok = 0;                               // lanes 0 .. n_ok-1 are confirmed and belong to the segment
            for (int guard = 0; guard < 66; guard++) {
                HWD_CNT(9, 1);
#if defined(HOSTSIM_W) && defined(HW_STATS)
                g_hw_stat_p1[guard < 7 ? guard : 7]++; g_hw_stat_dirty += (unsigned long long)w_popc64(dirty);
#endif
                W_LANES {
                    if ((dirty >> lane) & 1ull) {
                        int32_t st_ = 0;
                        const uint32_t bnd = (uint32_t)lane + 1u < nlanes ? p0 + ((uint32_t)lane + 1u) * S : 0xffffffffu;   // the last lane runs to the end-of-block symbol
                        hw_span<1>(smem, in32, PL(ln).start, bnd, limit_bits, mask_ll, mask_d, rll, rd, lq, PL(ln), nullptr, nullptr, 0, 0, 0, 0, st_);
                        // a range whose first unit starts at or beyond its boundary holds nothing: it ends where it starts
                    }
                    xch[HX_END * 64 + lane] = PL(ln).end; xch[HX_FLAG * 64 + lane] = PL(ln).flags;
                }
                W_SYNC();
                // link i: lane i starts where lane i-1 ended (and lane i-1 went on: no end-of-block, no error)
                uint64_t linked, stop;
                W_BALLOT(linked, lane == 0 || ((uint32_t)lane < nlanes && xch[HX_FLAG * 64 + lane - 1] == 0u && xch[HX_END * 64 + lane - 1] == PL(ln).start));
                W_BALLOT(stop, (uint32_t)lane < nlanes && PL(ln).flags != 0u);
                const uint32_t k_conf = (uint32_t)w_ctz64(~linked);                     // lanes 0 .. k_conf-1 are confirmed
                const uint64_t conf_mask = k_conf >= 64u ? ~0ull : ((1ull << k_conf) - 1ull);
                if (stop & conf_mask) { n_ok = (uint32_t)w_ctz64(stop & conf_mask) + 1u; break; }      // the first confirmed lane that stopped ends the segment
                if (k_conf >= nlanes) { n_ok = nlanes; break; }
                // restart every lane behind a broken link from its predecessor's end (the first of them becomes confirmed next round)
                uint64_t nd_;
                W_BALLOT(nd_, (uint32_t)lane >= k_conf && (uint32_t)lane < nlanes && xch[HX_FLAG * 64 + lane - 1] == 0u && xch[HX_END * 64 + lane - 1] != PL(ln).start);
                W_LANES { if ((nd_ >> lane) & 1ull) PL(ln).start = xch[HX_END * 64 + lane - 1]; }
                dirty = nd_;
                if (dirty == 0ull) { status = DHTS_BLK_ERR_INFLATE; break; }            // (cannot happen: lane k_conf's link is broken, so it is dirty)
            }
            if (status != 0) break;
            if (n_ok == 0u) { status = DHTS_BLK_ERR_INFLATE; break; }
            uint32_t last_flags, seg_end;
            HWD_T(t_s2); HWD_ADD(3, t_s1, t_s2);
            W_SYNC();
            last_flags = xch[HX_FLAG * 64 + n_ok - 1]; seg_end = xch[HX_END * 64 + n_ok - 1];
            if (last_flags & HWF_BAD) { status = DHTS_BLK_ERR_INFLATE; break; }
            eob_seen = (last_flags & HWF_EOB) != 0u;
            if (seg_end > limit_bits) { status = DHTS_BLK_ERR_INFLATE; break; }
            if (!eob_seen && seg_end >= limit_bits) { status = DHTS_BLK_ERR_INFLATE; break; }     // the payload ended without an end-of-block symbol
            // ---- places: exclusive sums over the lanes of the segment ----
            PLD(uint32_t, v_nlit); PLD(uint32_t, v_out); PLD(uint32_t, v_ntok); PLD(uint32_t, o_lit); PLD(uint32_t, o_out); PLD(uint32_t, o_tok); PLD(uint32_t, carry);
            uint32_t tot_lit, tot_out, tot_tok;
            W_LANES { const bool in = (uint32_t)lane < n_ok; PL(v_nlit) = in ? PL(ln).nlit : 0u; PL(v_out) = in ? PL(ln).outb : 0u; }
            W_EXCL_SCAN(o_lit, v_nlit, tot_lit);
            W_EXCL_SCAN(o_out, v_out, tot_out);
            if (outpos + tot_out > 65536u) { status = DHTS_BLK_ERR_INFLATE; break; }
            uint64_t has_match;
            W_BALLOT(has_match, (uint32_t)lane < n_ok && PL(ln).nmatch != 0u);
            W_LANES { xch[HX_A * 64 + lane] = PL(o_lit); xch[HX_B * 64 + lane] = (uint32_t)lane < n_ok ? PL(ln).tail : 0u; xch[HX_C * 64 + lane] = PL(v_nlit); }
            W_SYNC();
            W_LANES {
                // literals that precede this lane's first unit and follow the last match before it (or the start of the block)
                const uint64_t lower = has_match & ((1ull << lane) - 1ull);
                uint32_t c;
                if (lower == 0ull) c = run + PL(o_lit);
                else { const int j = w_msb64(lower); c = xch[HX_B * 64 + j] + PL(o_lit) - (xch[HX_A * 64 + j] + xch[HX_C * 64 + j]); }
                PL(carry) = c;
                PL(v_ntok) = ((uint32_t)lane < n_ok && PL(ln).nmatch != 0u) ? PL(ln).nmatch + PL(ln).pint + (c + PL(ln).lead) / DHTS_TOK_PURE : 0u;
            }
            W_EXCL_SCAN(o_tok, v_ntok, tot_tok);
            HWD_T(t_s3); HWD_ADD(4, t_s2, t_s3);
            // ---- pass 2: emit ----
            W_LANES {
                if ((uint32_t)lane < n_ok) {
                    HwLane chk; int32_t st_ = 0;
                    const uint32_t bnd = (uint32_t)lane + 1u < nlanes ? p0 + ((uint32_t)lane + 1u) * S : 0xffffffffu;
                    hw_span<2>(smem, in32, PL(ln).start, bnd, limit_bits, mask_ll, mask_d, rll, rd, lq, chk, lit, tok, nlit_tot + PL(o_lit), ntok_tot + PL(o_tok), outpos + PL(o_out), PL(carry), st_);
                    xch[HX_FLAG * 64 + lane] = (st_ != 0 || chk.end != PL(ln).end) ? 1u : 0u;
                } else xch[HX_FLAG * 64 + lane] = 0u;
            }
            W_SYNC();
            uint64_t bad2;
            W_BALLOT(bad2, xch[HX_FLAG * 64 + lane] != 0u);
            if (bad2) { status = DHTS_BLK_ERR_INFLATE; break; }
            // the run that is still open: literals after the last match of the segment
            if (has_match) { const int j = w_msb64(has_match); run = xch[HX_B * 64 + j] + tot_lit - (xch[HX_A * 64 + j] + xch[HX_C * 64 + j]); }
            else run += tot_lit;
            nlit_tot += tot_lit; ntok_tot += tot_tok; outpos += tot_out;
            {
                const uint32_t q = run / DHTS_TOK_PURE;
                W_LANES { for (uint32_t k = (uint32_t)lane; k < q; k += 64u) tok[ntok_tot + k] = DHTS_TOK_PURE << 23; }
                ntok_tot += q; run -= q * DHTS_TOK_PURE;
            }
            p0 = seg_end;
            W_SYNC();
            HWD_T(t_s4); HWD_ADD(5, t_s3, t_s4);
        }
        pos = p0;
    }
    if (status == 0 && pos > limit_bits) status = DHTS_BLK_ERR_INFLATE;
    InflateMeta m; m.ntok = ntok_tot; m.nlit = nlit_tot; m.outlen = outpos; m.status = status;
    W_LANES { if (lane == 0) meta[s] = m; }
#if defined(HW_DIAG) && !defined(HOSTSIM_W)
    { HWD_T(t_end); HWD_ADD(6, t_begin, t_end); HWD_CNT(7, 1); if (lane == 0) for (int q_ = 0; q_ < 16; q_++) if (hwd[q_]) atomicAdd(&g_hw_diag[q_], hwd[q_]); }
#endif
}
#endif
