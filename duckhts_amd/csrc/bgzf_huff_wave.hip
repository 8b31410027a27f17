// bgzf_huff_wave.hip -- phase A of the BGZF inflate, ONE WAVE PER BGZF BLOCK (gfx950), written from RFC 1951.
//
// Replaces the Huffman-decoding half of htslib bgzf.c:762-824 (bgzf_uncompress / inflate_block over zlib's inflate).
// Produces what bgzf_lz_resolve consumes: the literal bytes, one u32 token per LZ77 match and the InflateMeta record of every block
// (the same words the one-lane-per-block kernel of bgzf_inflate.hip produces: each is the other's cross-check in the tests and in
// tools/hostsim/sim_wave.cpp).
//
// Round 3 structure (every symbol is decoded ~1.3 times instead of 3.4):
//   * the kernel is PERSISTENT: a workgroup (one wave) takes BGZF blocks from an atomic counter until none is left, and owns a staging
//     area in global memory (64 lanes x {352 tokens, 1 KiB of literals}) that it reuses for every block, so it stays in the caches;
//   * the code-length section of a dynamic block is decoded once, wave-uniformly, and both alphabets become direct lookup tables in LDS
//     (u16 literal/length entries under an 11-bit root, u32 distance entries under an 8-bit root; longer codes by canonical arithmetic);
//   * the symbol stream [p0, end of payload) is cut into 64 equal bit ranges.  DEFLATE symbols are self-delimiting, so a decoder started
//     at a wrong bit falls into step with the true sequence after a few dozen symbols: PASS 0 decodes only the last HW_SYNC_W bits in
//     front of every range boundary (a loop that tracks nothing but the bit position) and PROPOSES where the next range starts;
//   * PASS 1 decodes every range from its proposed start and EMITS WHILE IT COUNTS: literal bytes and tokens go through small per-lane
//     rings in LDS (one unconditional ds_write each per symbol; the cursor only advances when the symbol was a literal / a distance) and
//     leave for the lane's staging slice in 16-byte stores.  Lane 0 starts at the true position; lane i is CONFIRMED when lane i-1 is
//     confirmed and ended exactly where lane i started.  Lanes behind a broken link start again from their predecessor's end (a few
//     percent of the 64-range segments need that second round): by induction the confirmed chain IS the serial decode;
//   * wave scans over the per-lane counts place every lane's literals and tokens in the block's scratch slot, and every lane copies its
//     own staging slice there (16 bytes per step); the literal run that crosses lane boundaries is added to the first match token of
//     the lane that holds the next match (plus "511 literals, no match" tokens when it overflows the 9-bit field);
//   * a lane whose staging slice is too small (never on real data: a slice holds 4x the average) sends the segment through lane 0
//     alone, which then appends directly to the block's slot.
// Distances are checked against the output position by bgzf_lz_resolve (the first kernel that knows absolute positions).
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
#define W_READLANE(x, i) ((x)[(i)])
#define W_WRITELANE(x, i, v) do { (x)[(i)] = (v); } while (0)
#define W_INCL_SCAN_MAX(dst, src) do { uint32_t run_ = 0; for (int lane = 0; lane < 64; lane++) { if (src[lane] > run_) run_ = src[lane]; dst[lane] = run_; } } while (0)
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
#define W_READLANE(x, i) ((uint32_t)__builtin_amdgcn_readlane((int)(x), (int)(i)))
#define W_WRITELANE(x, i, v) do { (x) = (lane == (int)(i)) ? (uint32_t)(v) : (x); } while (0)
#define W_INCL_SCAN_MAX(dst, src) do { dst = wave_incl_scan_max(src); } while (0)
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
// staging slice of one lane in global memory (per workgroup: 64 slices of each)
#define HW_LANE_TOK 352u           /* tokens  (64 x 352 = DHTS_TOK_STRIDE) */
#define HW_LANE_LIT 1024u          /* literal bytes */
#define HW_STAGE_TOK_WORDS (64u * HW_LANE_TOK)
#define HW_STAGE_LIT_BYTES (64u * HW_LANE_LIT)
// per-lane rings in LDS: literal bytes (32 per lane, stride 36), tokens [8][64]
#define HW_LRING_BYTES 2304u
#define HW_TRING_BYTES 2048u
// LDS image of one wave
#define HW_OFF_LL 0u                                     /* u16 [2^HW_RLL] */
#define HW_SUB_ENTRIES 512u                              /* second-level literal/length entries (u16) */
#define HW_OFF_SUB (HW_OFF_LL + (2u << HW_RLL))          /* u16 [HW_SUB_ENTRIES] */
#define HW_OFF_D (HW_OFF_SUB + 2u * HW_SUB_ENTRIES)      /* u32 [2^HW_RD]  */
#define HW_OFF_SLL (HW_OFF_D + (4u << HW_RD))            /* u16 [288] literal/length entries (without the code length) in canonical order */
#define HW_OFF_SD (HW_OFF_SLL + 576u)                    /* u32 [32]  distance entries in canonical order */
#define HW_OFF_TAB (HW_OFF_SD + 128u)                    /* u32 [2][3][16]: per alphabet limit15 / first / offs by code length */
#define HW_OFF_LRING (HW_OFF_TAB + 384u)                 /* literal ring; while the header is read: u8 [1024 + 8] staged header bytes */
#define HW_OFF_TRING (HW_OFF_LRING + HW_LRING_BYTES)     /* token ring;   while the tables are built: u8 [320] code lengths */
#define HW_OFF_IRING (HW_OFF_TRING + HW_TRING_BYTES)      /* input ring: 64 bytes per lane; between two decoding passes: */
#define HW_OFF_X HW_OFF_IRING                            /*   u32 [6][64] lane exchange arrays (never touched while a lane decodes) */
#define HW_OFF_HX (HW_OFF_TRING + 512u)                  /* u32 [64] exchange array of the header reader (the input ring's space holds its position table) */
#define HW_OFF_STAGE HW_OFF_LRING
#define HW_OFF_LENS HW_OFF_TRING
#define HW_LDS_BYTES (HW_OFF_IRING + 4096u)
// exchange arrays
#define HX_START 0
#define HX_END 1
#define HX_FLAG 2
#define HX_A 3
#define HX_B 4
#define HX_C 5

// literal/length entry (u16): [3:0] code length (0 = no symbol: bit 4 set -> code longer than the root, else invalid), [6:4] extra bits,
//   bit 7 = length code, [15:8] literal byte or length base - 3; end of block = length code with extra-bit count 7 (0x00F0)
// distance entry (u32): [3:0] code length (0 as above), [7:4] extra bits, [31:16] base - 1
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
    if (sym < 4u) return sym << 16;
    const uint32_t x = (sym >> 1) - 1u;
    const uint32_t base = 1u + ((2u | (sym & 1u)) << x);
    return ((base - 1u) << 16) | (x << 4);
}

#if defined(HOSTSIM_W) && defined(HW_STATS)
static unsigned long long g_hw_stat_p1[8], g_hw_stat_dirty, g_hw_stat_seg, g_hw_stat_fallback, g_hw_stat_p0bad, g_hw_stat_p0n, g_hw_stat_wrong, g_hw_stat_wrong_nominal;
static unsigned long long g_hw_stat_big[8];
// balance of the lanes of a pass: loop iterations (4 units each) of every lane; the wave pays the maximum
static unsigned long long g_hw_it[64], g_hw_stat_itmax[2], g_hw_stat_itsum[2], g_hw_stat_units[2];
#endif
// -DHW_DIAG (device builds for tools/dbg/hw_diag.py): cycles per phase, summed over blocks by lane 0
#if defined(HW_DIAG) && !defined(HOSTSIM_W)
__device__ unsigned long long g_hw_diag[16];   // 0 header 1 tables 2 pass0 3 pass1 4 scans 5 copy 6 total 7 blocks 8 segments 9 pass-1 rounds 10 fallbacks
#define HWD_T(v) const unsigned long long v = clock64()
#define HWD_ADD(i, a, b) do { hwd[i] += (b) - (a); } while (0)
#define HWD_CNT(i, n) do { hwd[i] += (n); } while (0)
#else
#define HWD_T(v) do { } while (0)
#define HWD_ADD(i, a, b) do { } while (0)
#define HWD_CNT(i, n) do { } while (0)
#endif
// per-lane results of the decoding pass
struct HwLane {
    uint32_t start, end;       // bit positions (relative to the aligned payload base): first unit / one past the last unit of this lane
    uint32_t flags;            // HWF_*
    uint32_t nlit, ntok;       // literal bytes / tokens (matches + the lane's own "511 literals" tokens) in the lane's slice
    uint32_t run, outb;        // literals since the lane's last token; inflated bytes of the lane's units
};
#define HWF_EOB 1u             /* ended on the end-of-block symbol */
#define HWF_BAD 2u             /* invalid code / ran past the payload */
#define HWF_OVF 4u             /* the lane's staging slice is full (the counts stay exact) */

// (hi:lo) >> n for n < 32
#ifdef HOSTSIM_W
static inline uint32_t hw_shr64lo(uint32_t hi, uint32_t lo, uint32_t n) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> n); }
static inline uint32_t hw_bfe(uint32_t v, uint32_t off, uint32_t width) { return width ? (v >> off) & (0xffffffffu >> (32u - width)) : 0u; }
#else
__device__ __forceinline__ uint32_t hw_shr64lo(uint32_t hi, uint32_t lo, uint32_t n) { return __builtin_amdgcn_alignbit(hi, lo, n); }
__device__ __forceinline__ uint32_t hw_bfe(uint32_t v, uint32_t off, uint32_t width) { return __builtin_amdgcn_ubfe(v, off, width); }
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

// ---- the lane's output streams: LDS rings -> 16-byte stores into the lane's slice ----------------------------------------------------
// literal ring: 32 bytes per lane at a stride of 36 bytes (nine banks: the 32 lanes of a group hit 32 different banks when they are at the
// same ring position); token ring: [8][64] words (every lane owns one bank)
#define HW_LRING_AT(smem, lane, k) ((smem) + HW_OFF_LRING + (uint32_t)(lane) * 36u + ((k) & 31u))
#define HW_TRING_AT(smem, lane, k) ((uint32_t *)((smem) + HW_OFF_TRING) + ((k) & 7u) * 64u + (uint32_t)(lane))
W_DEV uint4 hw_lring_chunk(const uint8_t *smem, int lane, uint32_t k) {     // the 16 literal bytes [k, k + 16), k a multiple of 16
    const uint32_t *p = (const uint32_t *)(smem + HW_OFF_LRING + (uint32_t)lane * 36u + (k & 16u));
    return make_uint4(p[0], p[1], p[2], p[3]);
}
W_DEV uint4 hw_tring_chunk(const uint8_t *smem, int lane, uint32_t k) {     // the 4 tokens [k, k + 4), k a multiple of 4
    const uint32_t *p = (const uint32_t *)(smem + HW_OFF_TRING) + (k & 4u) * 64u + (uint32_t)lane;
    return make_uint4(p[0], p[64], p[128], p[192]);
}

// input ring: 64 bytes per lane, [16 words][64 lanes]
#define HW_IRING_AT(smem, lane, k) ((uint32_t *)((smem) + HW_OFF_IRING) + ((k) & 15u) * 64u + (uint32_t)(lane))
// The next 16 input bytes of a lane travel global memory -> four VGPRs -> LDS ring WITHOUT the compiler's knowledge (a load the compiler
// sees is waited for where its result is copied, i.e. right behind the load: every iteration of the loop then costs a memory latency):
// HW_LOAD16 issues the load, HW_LOAD16_WAIT is the wait; both sit in ONE straight-line loop body (no back edge between them, so the four
// registers cannot be moved in between) with four units of decoding in the middle.
#ifdef HOSTSIM_W
typedef struct { uint32_t x, y, z, w; } hw_u32x4;
#define HW_LOAD16(dst, ptr) do { uint32_t t_[4]; __builtin_memcpy(t_, (ptr), 16); (dst).x = t_[0]; (dst).y = t_[1]; (dst).z = t_[2]; (dst).w = t_[3]; } while (0)
#define HW_LOAD16_WAIT(dst) do { } while (0)
#define HW_ANY(c) (c)
#else
typedef uint32_t hw_u32x4 __attribute__((ext_vector_type(4)));
#define HW_LOAD16(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory")
#define HW_LOAD16_WAIT(dst) asm volatile("s_waitcnt vmcnt(0)" : "+v"(dst) : : "memory")
// (a wave-level test in front of a rare per-lane case -- __ballot(c) != 0 -- was measured: the compiler then emits both the scalar
//  branch and the execution-mask branch, 4 % more instructions in all)
#define HW_ANY(c) (c)
#endif

// One lane decodes the units that START in [start, stop) with the tables in LDS: ONE UNIT per step -- a literal, or a length with
// its distance (the distance half runs under the execution mask of the lanes that decoded a length) -- so a range boundary always lies
// between units and no lane carries an alphabet state.  Straight-line arithmetic (masks instead of selects); the branches are the refill
// of the bit buffer (from the lane's input ring in LDS), the rare code above the table's root, the end-of-block symbol, a literal run
// of 511 or more in front of a match.  The loop body is four units between two "stage points": the top one requests the lane's next
// 16 input bytes, the bottom one parks them in the input ring and moves full 16-byte pieces of the output rings to the lane's slice.
// PASS 0: nothing is counted or emitted (proposal of the neighbour's start);  PASS 1: counts and emits into the lane's slice.
template <int PASS>
W_DEV void hw_span(uint8_t *smem, const uint32_t *in32, uint32_t start, uint32_t stop, uint32_t limit_bits, uint32_t mask_ll, uint32_t mask_d, uint32_t rll, uint32_t rd, uint32_t subbits,
                   int lane, HwLane &r, uint8_t *lit_out, uint32_t *tok_out, uint32_t lit_cap, uint32_t tok_cap, uint32_t run_in) {
    const uint16_t *lut_ll = (const uint16_t *)(smem + HW_OFF_LL);
    const uint32_t *lut_d = (const uint32_t *)(smem + HW_OFF_D);
    // input: words [.., wcommit) of the lane's stream are in the ring (word k at ring position k & 15); widx = the next word to enter the
    // bit buffer, nw = that word (read one refill ahead, so the LDS latency overlaps the decoding of 32 bits)
    uint32_t lo, hi = 0, cnt, widx, nw, gnext, wcommit;
    {
        const uint32_t w = start >> 5, o = start & 31u, g0 = w >> 2;
        for (uint32_t g = g0; g < g0 + 4u; g++) {
            uint4 c; __builtin_memcpy(&c, in32 + 4u * g, 16);
            *HW_IRING_AT(smem, lane, 4u * g) = c.x; *HW_IRING_AT(smem, lane, 4u * g + 1u) = c.y; *HW_IRING_AT(smem, lane, 4u * g + 2u) = c.z; *HW_IRING_AT(smem, lane, 4u * g + 3u) = c.w;
        }
        gnext = g0 + 4u; wcommit = 4u * gnext;
        lo = *HW_IRING_AT(smem, lane, w) >> o; cnt = 32u - o; widx = w + 1u; nw = *HW_IRING_AT(smem, lane, widx);
    }
    // (the bit buffer holds fewer than 32 bits here, all of them in lo: hi is zero)
#define HW_REFILL() do {                                                                                                                              \
        if (cnt < 32u) { lo |= nw << cnt; hi = (nw >> 1) >> (31u - cnt); cnt += 32u; widx++; nw = *HW_IRING_AT(smem, lane, widx); }                       \
    } while (0)
#define HW_TAKE(n_) do { const uint32_t u_ = (n_); lo = hw_shr64lo(hi, lo, u_); hi >>= u_; cnt -= u_; pos += u_; } while (0)
#define HW_PUSH_TOK(v_) do { *HW_TRING_AT(smem, lane, ntok) = (v_); ntok++; } while (0)
#ifdef HW_EXP_NOEMIT            /* knock-out experiment: what the 16-byte stores into the lane's slice cost; the output is invalid */
#define HW_EXP_CAP(c_) 0u
#else
#define HW_EXP_CAP(c_) (c_)
#endif
#define HW_FLUSH_TOK() do {                                                                                                                             \
        if (tfl + 4u <= HW_EXP_CAP(tok_cap)) { const uint4 v_ = hw_tring_chunk(smem, lane, tfl); __builtin_memcpy(tok_out + tfl, &v_, 16); } else flags |= HWF_OVF;  \
        tfl += 4u;                                                                                                                                        \
    } while (0)
#define HW_FLUSH_LIT() do {                                                                                                                             \
        if (lfl + 16u <= HW_EXP_CAP(lit_cap)) { const uint4 v_ = hw_lring_chunk(smem, lane, lfl); __builtin_memcpy(lit_out + lfl, &v_, 16); } else flags |= HWF_OVF; \
        lfl += 16u;                                                                                                                                       \
    } while (0)
    uint32_t pos = start, flags = 0;
    uint32_t nlit = 0, ntok = 0, run = (PASS == 1) ? run_in : 0u, mbytes = 0;
    uint32_t lfl = 0, tfl = 0;                               // literal bytes / tokens already stored in the slice (multiples of 16 / 4)
    // a lane that decodes garbage (wrong start, damaged stream) stops at the end of the payload at the latest; a lane that has seen the
    // end-of-block symbol or an invalid code stops by pulling its limit down to zero
    uint32_t stopv = stop < limit_bits + 64u ? stop : limit_bits + 64u;
    // one literal / length symbol: entry, fields, the literal ring (pass 1), the bits taken, the end of the block
#define HW_LL() do {                                                                                                                                  \
            HW_REFILL();                                                                                                                                  \
            e = lut_ll[lo & mask_ll];                                                                                                                     \
            if (HW_ANY((e & 15u) == 0u)) {                                                                                                                \
                if ((e & 15u) == 0u) {                                                                                                                    \
                    /* a code longer than the root: second-level table (number in the entry, index = the next stream bits), else canonical arithmetic */\
                    if (e & 0x20u) e = ((const uint16_t *)(smem + HW_OFF_SUB))[((e >> 6) << subbits) + hw_bfe(lo, rll, subbits)];                          \
                    else e = (e & HW_LONG) ? hw_long_code(smem, lo, 0u, rll) : 0u;                                                                        \
                    if ((e & 15u) == 0u) { flags |= HWF_BAD; stopv = 0; e = 0x0001u; }                                                                    \
                }                                                                                                                                         \
            }                                                                                                                                             \
            const uint32_t L = e & 15u, x = (e >> 4) & 7u;                                                                                                \
            ext = hw_bfe(lo, L, x);                               /* (the end-of-block entry asks for 7 bits: given back below) */                         \
            islen_m = (uint32_t)((int32_t)(e << 24) >> 31);       /* all ones for a length code / end of block */                                         \
            if (PASS == 1) {                                                                                                                              \
                /* the literal ring is written every time; its cursor moves only when the symbol was a literal */                                         \
                *HW_LRING_AT(smem, lane, nlit) = (uint8_t)(e >> 8);                                                                                       \
                nlit += 1u + islen_m; run += 1u + islen_m;                                                                                                \
            }                                                                                                                                             \
            HW_TAKE(L + x);                                                                                                                               \
            iseob = (e & 0xf0u) == 0xf0u;                                                                                                                 \
            if (HW_ANY(iseob)) { if (iseob) { flags |= HWF_EOB; stopv = 0; pos -= 7u; } }                                                                 \
    } while (0)
    // ONE step: a literal / length symbol, then the distance of the lanes that hold a length.  -DHW_LL2=1 (round 4, measured and not used:
    // 9.10 against 9.01 ms per 65,536 blocks, profiles/r04/huff_two_symbols_per_step.txt) gives the lanes whose first symbol was a literal a
    // SECOND literal / length symbol before the distance half (two thirds of the symbols are literals, and a lane that has decoded one idles
    // through the distance half): 40 % fewer steps, no less time -- the wave waits on its dependent LDS look-ups, not on its instruction count.
    // Every piece ends on a unit boundary, so a range boundary lies between units either way.
#ifndef HW_LL2
#define HW_LL2 0
#endif
#define HW_UNIT() do {                                                                                                                                \
        if (pos < stopv) {                                                                                                                                \
            uint32_t e, ext, islen_m; bool iseob;                                                                                                         \
            HW_LL();                                                                                                                                      \
            if (HW_LL2 && islen_m == 0u && pos < stopv) HW_LL();                                                                                          \
            if (islen_m != 0u && !iseob) {                                                                                                                \
                const uint32_t want3 = (e >> 8) + ext;          /* length - 3 */                                                                          \
                HW_REFILL();                                                                                                                              \
                uint32_t d = lut_d[lo & mask_d];                                                                                                          \
                if (HW_ANY((d & 15u) == 0u)) {                                                                                                            \
                    if ((d & 15u) == 0u) {                                                                                                                \
                        d = !(d & HW_LONG) ? 0u : hw_long_code(smem, lo, 1u, rd);                                                                         \
                        if ((d & 15u) == 0u) { flags |= HWF_BAD; stopv = 0; d = 0x0001u; }                                                                \
                    }                                                                                                                                     \
                }                                                                                                                                         \
                const uint32_t L2 = d & 15u, x2 = (d >> 4) & 15u;                                                                                         \
                const uint32_t dist1 = (d >> 16) + hw_bfe(lo, L2, x2);      /* distance - 1 */                                                            \
                HW_TAKE(L2 + x2);                                                                                                                         \
                if (PASS == 1) {                                                                                                                          \
                    if (HW_ANY(run >= DHTS_TOK_PURE)) {                                                                                                   \
                        while (run >= DHTS_TOK_PURE) {             /* 511 literals or more in front of this match: "511 literals, no match" tokens */     \
                            HW_PUSH_TOK(DHTS_TOK_PURE << 23); run -= DHTS_TOK_PURE;                                                                       \
                            if (ntok - tfl >= 4u) HW_FLUSH_TOK();                                                                                         \
                        }                                                                                                                                 \
                    }                                                                                                                                     \
                    HW_PUSH_TOK((run << 23) | (want3 << 15) | dist1);                                                                                     \
                    run = 0; mbytes += want3 + 3u;                                                                                                        \
                }                                                                                                                                         \
            }                                                                                                                                             \
        }                                                                                                                                                 \
    } while (0)
#if defined(HOSTSIM_W) && defined(HW_STATS)
    g_hw_it[lane] = 0;
#endif
#ifdef HW_PER_SYMBOL           /* (not the default: measured slower, see below) */
    // ONE SYMBOL per step, the same instructions for both alphabets (round 4).  A lane is either in front of a literal / length code or -- `ind`
    // -- in front of the distance code of the length it has just read; the two cases differ in the table they index, in two field widths
    // and in what the value means, all of which are selects, not branches.  A unit-per-step loop made every lane walk through the distance
    // half in every step because SOME lane of 64 always holds a length (a third of the units are matches), although two thirds of the
    // lanes had nothing to do there.  A range still ends between units: a lane that has read a length goes on to its distance whatever
    // `stopv` says.  MEASURED AND NOT USED (profiles/r04/pmc_sq_per_unit_vs_per_symbol.txt, huff_per_unit_vs_per_symbol.txt): the
    // symbol step costs ~93 wave-instructions as compiled, a unit step ~110, and a unit is 1.33 symbols: 80.0 k instead of 70.8 k
    // wave-instructions per BGZF block, 10.1 instead of 9.0 ms per 65,536 blocks.  It would need < 83 per step to pay.  Kept because it is
    // exact (host simulation, word for word the lane kernel's output) and is the shape a hand-scheduled version would start from.
    // (`ind` is kept as 0 / 1 in a vector register and the bookkeeping is arithmetic on it: a bool makes the compiler juggle lane masks on the
    //  scalar unit, and scalar instructions take the same issue slots)
    uint32_t ind = 0, want3p = 0;                             // want3p = (length - 3) + 3 of the pending match
#define HW_SYM() do {                                                                                                                                 \
        if (pos < stopv || ind != 0u) {                                                                                                                   \
            HW_REFILL();                                                                                                                                  \
            const uint32_t a_ll = (lo & mask_ll) << 1, a_d = HW_OFF_D + ((lo & mask_d) << 2);                                                              \
            uint32_t e; __builtin_memcpy(&e, smem + (ind ? a_d : a_ll), 4);   /* (a literal/length entry is the low half: the fields below never look higher) */ \
            if (HW_ANY((e & 15u) == 0u)) {                                                                                                                \
                if ((e & 15u) == 0u) {                                                                                                                    \
                    if (ind) e = !(e & HW_LONG) ? 0u : hw_long_code(smem, lo, 1u, rd);                                                                    \
                    else if (e & 0x20u) e = ((const uint16_t *)(smem + HW_OFF_SUB))[(((e & 0xffffu) >> 6) << subbits) + hw_bfe(lo, rll, subbits)];          \
                    else e = (e & HW_LONG) ? hw_long_code(smem, lo, 0u, rll) : 0u;                                                                        \
                    if ((e & 15u) == 0u) { flags |= HWF_BAD; stopv = 0; e = 0x0001u; ind = 0u; }                                                          \
                }                                                                                                                                         \
            }                                                                                                                                             \
            const uint32_t L = e & 15u, x = hw_bfe(e, 4u, 3u + ind);                                                                                      \
            const uint32_t k8 = 8u << ind, val = hw_bfe(e, k8, k8) + hw_bfe(lo, L, x);   /* literal byte | length - 3 | distance - 1 (an end-of-block entry asks for 7 bits: given back below) */ \
            const uint32_t lenb = hw_bfe(e, 7u, 1u) & ~ind;            /* 1: a length code (or the end of the block) */                                     \
            const uint32_t litv = (lenb | ind) ^ 1u;                   /* 1: a literal */                                                                  \
            HW_TAKE(L + x);                                                                                                                               \
            if (PASS == 1) {                                                                                                                              \
                /* both rings are written every time; a cursor moves only when the symbol was what the ring holds */                                      \
                *HW_LRING_AT(smem, lane, nlit) = (uint8_t)val;                                                                                            \
                nlit += litv; run += litv;                                                                                                                \
                *HW_TRING_AT(smem, lane, ntok) = (run << 23) | ((want3p - 3u) << 15) | val;                                                               \
                mbytes += ind * want3p;                                                                                                                   \
            }                                                                                                                                             \
            /* the rare cases in one test: the end of the block, and (pass 1) a match behind 511 literals or more, whose token was not right */           \
            if (HW_ANY((lenb != 0u && (e & 0x70u) == 0x70u) || (PASS == 1 && ind != 0u && run >= DHTS_TOK_PURE))) {                                        \
                if (lenb != 0u && (e & 0x70u) == 0x70u) { flags |= HWF_EOB; stopv = 0; pos -= 7u; }                                                       \
                else if (PASS == 1 && ind != 0u && run >= DHTS_TOK_PURE) {                                                                                \
                    while (run >= DHTS_TOK_PURE) {                 /* "511 literals, no match" tokens in front of the match's own */                      \
                        HW_PUSH_TOK(DHTS_TOK_PURE << 23); run -= DHTS_TOK_PURE;                                                                           \
                        if (ntok - tfl >= 4u) HW_FLUSH_TOK();                                                                                             \
                    }                                                                                                                                     \
                    *HW_TRING_AT(smem, lane, ntok) = (run << 23) | ((want3p - 3u) << 15) | val;                                                           \
                }                                                                                                                                         \
            }                                                                                                                                             \
            if (PASS == 1) { ntok += ind; run &= ind - 1u; }                                                                                              \
            want3p = lenb ? val + 3u : want3p;                                                                                                            \
            ind = lenb & (flags ^ HWF_EOB);                            /* (bit 0 of flags is HWF_EOB: a lane that has just seen it has no distance to read) */ \
        }                                                                                                                                                 \
    } while (0)
    hw_u32x4 inq; inq.x = 0; inq.y = 0; inq.z = 0; inq.w = 0;
    while (pos < stopv || ind != 0u) {
#if defined(HOSTSIM_W) && defined(HW_STATS)
        g_hw_it[lane]++;
#endif
        // stage points as in the per-unit loop below: six symbols take at most 21 bytes
        if (wcommit - widx < 8u) {
            const uint32_t *gp = in32 + 4u * gnext; hw_u32x4 now; HW_LOAD16(now, gp); HW_LOAD16_WAIT(now);
            *HW_IRING_AT(smem, lane, 4u * gnext) = now.x; *HW_IRING_AT(smem, lane, 4u * gnext + 1u) = now.y; *HW_IRING_AT(smem, lane, 4u * gnext + 2u) = now.z; *HW_IRING_AT(smem, lane, 4u * gnext + 3u) = now.w;
            gnext++; wcommit += 4u;
        }
        const bool req = gnext - (widx >> 2) < 4u;
        if (req) { const uint32_t *gp = in32 + 4u * gnext; HW_LOAD16(inq, gp); }
        HW_SYM(); HW_SYM(); HW_SYM(); HW_SYM(); HW_SYM(); HW_SYM();
        HW_LOAD16_WAIT(inq);
        if (req) {
            *HW_IRING_AT(smem, lane, 4u * gnext) = inq.x; *HW_IRING_AT(smem, lane, 4u * gnext + 1u) = inq.y; *HW_IRING_AT(smem, lane, 4u * gnext + 2u) = inq.z; *HW_IRING_AT(smem, lane, 4u * gnext + 3u) = inq.w;
            gnext++; wcommit += 4u;
        }
        if (PASS == 1) {
            // at most three new tokens and six new literals since the last stage point (a match with a very long literal run flushes for itself)
            if (ntok - tfl >= 4u) HW_FLUSH_TOK();
            if (nlit - lfl >= 16u) HW_FLUSH_LIT();
        }
    }
#undef HW_SYM
#else
    hw_u32x4 inq; inq.x = 0; inq.y = 0; inq.z = 0; inq.w = 0;
    while (pos < stopv) {
#if defined(HOSTSIM_W) && defined(HW_STATS)
        g_hw_it[lane]++;
#endif
        // stage point: four units take at most 24 bytes (6 words) and the refill reads one word ahead: with fewer than 8 words in the ring
        // beyond widx the lane fetches its next 16 bytes and waits for them (a rare burst of long matches); otherwise it requests
        // them when the ring has a free slot (the chunks [widx / 4, gnext) are in it or on their way) and collects them four units later
        if (wcommit - widx < 8u) {
            const uint32_t *gp = in32 + 4u * gnext; hw_u32x4 now; HW_LOAD16(now, gp); HW_LOAD16_WAIT(now);
            *HW_IRING_AT(smem, lane, 4u * gnext) = now.x; *HW_IRING_AT(smem, lane, 4u * gnext + 1u) = now.y; *HW_IRING_AT(smem, lane, 4u * gnext + 2u) = now.z; *HW_IRING_AT(smem, lane, 4u * gnext + 3u) = now.w;
            gnext++; wcommit += 4u;
        }
        const bool req = gnext - (widx >> 2) < 4u;
        if (req) { const uint32_t *gp = in32 + 4u * gnext; HW_LOAD16(inq, gp); }
#if HW_LL2
        HW_UNIT(); HW_UNIT(); HW_UNIT();                     // (three steps of at most 20 + 20 + 28 bits: 26 bytes, within the 28 the ring holds ahead)
#else
        HW_UNIT(); HW_UNIT(); HW_UNIT(); HW_UNIT();
#endif
        // stage point: the requested bytes have arrived (the load is four units old; so are the stores of the previous stage point)
        HW_LOAD16_WAIT(inq);
        if (req) {
            *HW_IRING_AT(smem, lane, 4u * gnext) = inq.x; *HW_IRING_AT(smem, lane, 4u * gnext + 1u) = inq.y; *HW_IRING_AT(smem, lane, 4u * gnext + 2u) = inq.z; *HW_IRING_AT(smem, lane, 4u * gnext + 3u) = inq.w;
            gnext++; wcommit += 4u;
        }
        if (PASS == 1) {
            // at most four new entries per ring since the last stage point (a match with a very long literal run flushes for itself)
            if (ntok - tfl >= 4u) HW_FLUSH_TOK();
            if (nlit - lfl >= 16u) HW_FLUSH_LIT();
        }
    }
#endif
    if (!(flags & HWF_EOB) && pos > limit_bits) flags |= HWF_BAD;            // ran off the payload (a true stream ends with its end-of-block symbol)
    if (PASS == 1) {
        // what is still in the rings: whole pieces first, then single entries (nothing is ever stored beyond the lane's counts)
        while (ntok - tfl >= 4u) HW_FLUSH_TOK();
        for (; tfl < ntok; tfl++) { if (tfl < tok_cap) tok_out[tfl] = *HW_TRING_AT(smem, lane, tfl); else flags |= HWF_OVF; }
        while (nlit - lfl >= 16u) HW_FLUSH_LIT();
        for (; lfl < nlit; lfl++) { if (lfl < lit_cap) lit_out[lfl] = *HW_LRING_AT(smem, lane, lfl); else flags |= HWF_OVF; }
        r.nlit = nlit; r.ntok = ntok; r.run = run; r.outb = nlit + mbytes;
    }
    r.end = pos; r.flags = flags;
#undef HW_UNIT
#undef HW_REFILL
#undef HW_TAKE
#undef HW_PUSH_TOK
#undef HW_FLUSH_TOK
#undef HW_FLUSH_LIT
}

// ---- wave-uniform header reader over the staged bytes ---------------------------------------------------------------------------
struct HwHdr { uint64_t buf; uint32_t cnt, widx; const uint32_t *words; };
W_DEV void hw_hdr_fill(HwHdr &h) { if (h.cnt <= 32u) { h.buf |= (uint64_t)W_UNI(h.words[h.widx]) << h.cnt; h.widx++; h.cnt += 32u; } }
W_DEV uint32_t hw_hdr_take(HwHdr &h, uint32_t n) { hw_hdr_fill(h); const uint32_t v = (uint32_t)h.buf & ((1u << n) - 1u); h.buf >>= n; h.cnt -= n; return v; }

// One BGZF block: block `bi` of the table, decoded by the calling wave into `lit` / `tok` (room for DHTS_LIT_STRIDE bytes / DHTS_TOK_STRIDE
// tokens) and described by `mres`.  `slit` / `stok` are the workgroup's staging slices.
#ifdef HOSTSIM_W
static void hw_block(uint8_t *smem, int64_t bi, const uint8_t *comp, BgzfTable tab,
                     uint8_t *lit, uint32_t *tok, InflateMeta &mres, uint8_t *slit, uint32_t *stok, uint32_t *dens)
#else
__device__ __forceinline__ void hw_block(uint8_t *smem, int64_t bi, const uint8_t *__restrict__ comp, BgzfTable tab,
                                         uint8_t *__restrict__ lit, uint32_t *__restrict__ tok, InflateMeta &mres,
                                         uint8_t *__restrict__ slit, uint32_t *__restrict__ stok, unsigned long long *hwd, uint32_t &dens)
#endif
{
    W_LANE_DECL
    const uint32_t clen = tab.clen[bi];
    uint16_t *lut_ll = (uint16_t *)(smem + HW_OFF_LL);
    uint32_t *lut_d = (uint32_t *)(smem + HW_OFF_D);
    uint8_t *stage = smem + HW_OFF_STAGE;
    uint8_t *lens = smem + HW_OFF_LENS;
    uint16_t *sll = (uint16_t *)(smem + HW_OFF_SLL);
    uint32_t *sd = (uint32_t *)(smem + HW_OFF_SD);
    uint32_t *tb = (uint32_t *)(smem + HW_OFF_TAB);          // [0..15] ll limit15, [16..31] ll first, [32..47] ll offs, [48..] the same for distances
    uint32_t *xch = (uint32_t *)(smem + HW_OFF_X);
#ifndef HOSTSIM_W
    (void)hwd;
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
        HWD_T(t_hs); HWD_ADD(11, t_h0, t_hs);
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
            HWD_T(t_hc0);
            // lane s < 19 owns code-length symbol s: its rank among the symbols of its length (ballot per length) gives its canonical code;
            // then 19 uniform steps drop every symbol into the entries whose low bits are its (bit-reversed) code
            PLD(uint32_t, cl0); PLD(uint32_t, cl1); PLD(uint32_t, clen_s); PLD(uint32_t, crev_s);
            W_LANES { PL(clen_s) = lane < 19 ? (uint32_t)(clpack >> (3u * (uint32_t)lane)) & 7u : 0u; PL(crev_s) = 0; PL(cl0) = 0; PL(cl1) = 0; }
            for (uint32_t L = 1; L <= 7u; L++) {
                if (ccount[L] == 0u) continue;
                uint64_t m; W_BALLOT(m, PL(clen_s) == L);
                W_LANES { if (PL(clen_s) == L) PL(crev_s) = w_brev32(cfirst[L] + w_popc64(m & ((1ull << lane) - 1ull))) >> (32u - L); }
            }
            for (int sy = 0; sy < 19; sy++) {
                const uint32_t L = W_READLANE(clen_s, sy), rv = W_READLANE(crev_s, sy);
                if (L == 0u) continue;
                const uint32_t msk = (1u << L) - 1u, ent = L | ((uint32_t)sy << 3);
                W_LANES { if (((uint32_t)lane & msk) == rv) PL(cl0) = ent; if ((((uint32_t)lane + 64u) & msk) == rv) PL(cl1) = ent; }
            }
            (void)coffs;
            // The code-length symbols (RFC 1951 3.2.7) are decoded at EVERY bit position of a 1,024-bit window at once (16 positions per
            // lane): how far the symbol at that position reaches, how many lengths it stands for and which value.  A symbol is at most
            // 14 bits long, so the chain of positions where symbols really start touches every lane's 16 positions: each lane works out
            // where the chain leaves its positions for each of the 16 offsets at which it can come in (a 16 x 4-bit map, built backwards),
            // 64 scalar steps push the chain's true entry offset through the maps, and then every lane walks only its own few symbols:
            // once to count the lengths they stand for (wave scans give each lane its first index and the value a "repeat previous"
            // symbol copies), once to write them.
            uint16_t *cltab = (uint16_t *)(smem + HW_OFF_SUB);               // u16 [128]: the direct table of the code-length code (the second-level table's space)
            uint32_t *arr = (uint32_t *)(smem + HW_OFF_IRING);               // u32 [1024]: per position: advance | count << 4 | value << 12 (0xff: the previous length)
            W_LANES { cltab[lane] = (uint16_t)PL(cl0); cltab[64 + lane] = (uint16_t)PL(cl1); }
            HWD_T(t_hc1); HWD_ADD(12, t_hc0, t_hc1);
            const uint32_t total = nl + nd;
            const uint32_t w0s = pos >> 5;                                   // the staged bytes start at this word
            uint32_t idx0 = 0, prev0 = 16u, wbase = hpos;                    // lengths written so far; the last one (16: there is none); the window's first position (a symbol starts there)
            bool hdr_done = false;
            while (status == 0 && !hdr_done) {
                W_SYNC();
                W_LANES {
                    const uint32_t *stw = (const uint32_t *)stage;
                    for (uint32_t j = 0; j < 16u; j++) {
                        const uint32_t qq = j * 64u + (uint32_t)lane, bp = wbase + qq, wi = (bp >> 5) - w0s;
                        uint32_t packed = 0;                                 // advance 0: no symbol can be read here
                        if (wi + 1u < HW_STAGE / 4u) {
                            const uint32_t v = hw_shr64lo(stw[wi + 1u], stw[wi], bp & 31u);
                            const uint32_t ent = cltab[v & 127u], L = ent & 7u, sym = ent >> 3;
                            const uint32_t xb = sym < 16u ? 0u : sym == 16u ? 2u : sym == 17u ? 3u : 7u;
                            const uint32_t ev = (v >> L) & ((1u << xb) - 1u);
                            const uint32_t rep = sym < 16u ? 1u : sym == 18u ? 11u + ev : 3u + ev;
                            const uint32_t val = sym < 16u ? sym : sym == 16u ? 0xffu : 0u;
                            if (L != 0u) packed = (L + xb) | (rep << 4) | (val << 12);
                        }
                        arr[qq] = packed;
                    }
                }
                W_SYNC();
                HWD_T(t_hw0);
                // the lane's map: offset of the chain's first position in the lane's range -> the same for the next lane (15: the chain ends
                // on an invalid code).  Built from offset 15 down: a symbol either leaves the range or lands on an offset already done.
                PLD(uint32_t, mlo); PLD(uint32_t, mhi); PLD(uint32_t, ent_o);
                W_LANES {
                    uint64_t m = 0;
                    for (int o = 15; o >= 0; o--) {
                        const uint32_t a = arr[16u * (uint32_t)lane + (uint32_t)o] & 15u, k = (uint32_t)o + a;
                        const uint32_t ex = a == 0u ? 15u : k >= 16u ? k - 16u : (uint32_t)(m >> (4u * k)) & 15u;
                        m |= (uint64_t)ex << (4u * (uint32_t)o);
                    }
                    // (an entry offset of 15 can only be the previous lane's "chain ended": a symbol is at most 14 bits long)
                    m |= 15ull << 60;
                    PL(mlo) = (uint32_t)m; PL(mhi) = (uint32_t)(m >> 32); PL(ent_o) = 0;
                }
                uint32_t e_run = 0;                                          // lane 0 starts on the window's first position
                for (int i = 0; i < 64; i++) {
                    W_WRITELANE(ent_o, i, e_run);
                    const uint32_t lo_ = W_READLANE(mlo, i), hi_ = W_READLANE(mhi, i);
                    e_run = ((e_run < 8u ? lo_ >> (4u * e_run) : hi_ >> (4u * (e_run - 8u))) & 15u);
                }
                // first walk: how many lengths the lane's symbols stand for, and the last value they define (16: none -- only "repeat previous")
                PLD(uint32_t, nout); PLD(uint32_t, lastv); PLD(uint32_t, o_out);
                W_LANES {
                    uint32_t o = PL(ent_o), n = 0, lv = 16u;
                    if (o == 15u) o = 16u;                                   // the chain ended in front of this lane
                    while (o < 16u) {
                        const uint32_t f = arr[16u * (uint32_t)lane + o], a = f & 15u;
                        if (a == 0u) break;
                        n += (f >> 4) & 255u;
                        const uint32_t v = (f >> 12) & 255u;
                        lv = v == 0xffu ? lv : v;
                        o += a;
                    }
                    PL(nout) = n; PL(lastv) = lv;
                }
                uint32_t tot_out;
                W_EXCL_SCAN(o_out, nout, tot_out);
                // the value a leading "repeat previous" copies: the last value defined by an earlier lane (or by the previous window)
                PLD(uint32_t, key); PLD(uint32_t, kmax);
                W_LANES { PL(key) = PL(lastv) < 16u ? (((uint32_t)lane + 1u) << 8) | PL(lastv) : 0u; }
                W_INCL_SCAN_MAX(kmax, key);
                uint32_t *hx = (uint32_t *)(smem + HW_OFF_HX);
                W_LANES { hx[lane] = PL(kmax); }
                W_SYNC();
                // second walk: write the lengths; the lane that meets index `total` knows where the header ends
                PLD(uint32_t, res);                                          // 0 nothing, 1 error, 0x100 | offset: the header ends at this offset of the lane's range
                W_LANES {
                    uint32_t o = PL(ent_o), idx = idx0 + PL(o_out), r_ = 0;
                    const uint32_t kin = lane == 0 ? 0u : hx[lane - 1];
                    uint32_t pv = kin ? (kin & 255u) : prev0;
                    if (PL(ent_o) == 15u) { if (idx < total) r_ = 1u; }       // the chain ended on an invalid code before the last length
                    else if (idx <= total) {
                        while (o < 16u) {
                            if (idx >= total) { r_ = 0x100u | o; break; }
                            const uint32_t f = arr[16u * (uint32_t)lane + o], a = f & 15u, rep = (f >> 4) & 255u;
                            if (a == 0u) { r_ = 1u; break; }
                            uint32_t v = (f >> 12) & 255u;
                            if (v == 0xffu) { if (pv >= 16u) { r_ = 1u; break; } v = pv; }      // nothing to repeat (RFC 1951 3.2.7)
                            if (idx + rep > total) { r_ = 1u; break; }
                            // distance lengths live behind the 288 literal/length slots (a value other than zero is repeated six times at most)
                            if (v != 0u) for (uint32_t rr = 0; rr < rep; rr++) { const uint32_t i_ = idx + rr; lens[i_ < nl ? i_ : 288u + (i_ - nl)] = (uint8_t)v; }
                            pv = v; idx += rep; o += a;
                        }
                    }
                    PL(res) = r_;
                }
                uint64_t m_err, m_end;
                W_BALLOT(m_err, PL(res) == 1u);
                W_BALLOT(m_end, PL(res) >= 0x100u);
                { HWD_T(t_hw1); HWD_ADD(14, t_hw0, t_hw1); }
                // (lanes behind the one that meets `total` see indices >= total at once: the first of them all is the one that counts)
                if (m_end) {
                    const int le = w_ctz64(m_end);
                    if (m_err & ((1ull << le) - 1ull)) { status = DHTS_BLK_ERR_INFLATE; break; }
                    hpos = wbase + 16u * (uint32_t)le + (W_READLANE(res, le) & 255u);
                    hdr_done = true;
                } else {
                    if (m_err) { status = DHTS_BLK_ERR_INFLATE; break; }
                    // the window ends before the last length: go on where the chain leaves it
                    if (e_run == 15u) { status = DHTS_BLK_ERR_INFLATE; break; }
                    idx0 += tot_out;
                    const uint32_t klast = W_READLANE(kmax, 63);
                    prev0 = klast ? (klast & 255u) : prev0;
                    wbase += 1024u + e_run;
                    if (idx0 == total) { hpos = wbase; hdr_done = true; }
                    else if (idx0 > total || (uint32_t)(wbase - pos) > HW_STAGE * 8u - 64u) { status = DHTS_BLK_ERR_INFLATE; break; }
                }
            }
            if (status != 0) break;
            if ((uint32_t)(hpos - pos) > HW_STAGE * 8u - 64u) { status = DHTS_BLK_ERR_INFLATE; break; }     // (cannot happen: a header is < 4,600 bits)
            W_SYNC();
        }
        if (status == 0 && hpos > limit_bits) { status = DHTS_BLK_ERR_INFLATE; break; }
        HWD_T(t_h1); HWD_ADD(0, t_h0, t_h1);

        // ---- canonical tables ----
        // per alphabet: every lane holds K consecutive symbols; the counts by code length travel as 10-bit fields of five words through
        // five wave scans (totals -> limits / first codes / offsets, prefixes -> rank of every symbol among the symbols of its length)
        // -> entries in canonical order, then the direct table in code order; literal/length codes longer than the root get a second-level
        // table of 2^(longest - root) entries per root-bit prefix.
        uint32_t rll = 0, rd = 0, subbits = 0;
        for (int alpha = 0; alpha < 2 && status == 0; alpha++) {
            const uint32_t nsym = alpha ? 32u : 288u, base = alpha ? 288u : 0u, K = alpha ? 1u : 5u;
            PLD(uint32_t, lp); PLD(uint32_t, c0); PLD(uint32_t, c1); PLD(uint32_t, c2); PLD(uint32_t, c3); PLD(uint32_t, c4);
            PLD(uint32_t, e0); PLD(uint32_t, e1); PLD(uint32_t, e2); PLD(uint32_t, e3); PLD(uint32_t, e4);
            uint32_t tt[5];
            W_LANES {
                uint32_t packed = 0, c[5] = {0, 0, 0, 0, 0};
                for (uint32_t j = 0; j < K; j++) {
                    const uint32_t sy = K * (uint32_t)lane + j;
                    const uint32_t L = sy < nsym ? lens[base + sy] : 0u;
                    packed |= L << (4u * j);
                    if (L != 0u) { for (uint32_t w = 0; w < 5u; w++) { const uint32_t t = L - 1u - 3u * w; if (t < 3u) c[w] += 1u << (10u * t); } }
                }
                PL(lp) = packed; PL(c0) = c[0]; PL(c1) = c[1]; PL(c2) = c[2]; PL(c3) = c[3]; PL(c4) = c[4];
            }
            W_EXCL_SCAN(e0, c0, tt[0]); W_EXCL_SCAN(e1, c1, tt[1]); W_EXCL_SCAN(e2, c2, tt[2]); W_EXCL_SCAN(e3, c3, tt[3]); W_EXCL_SCAN(e4, c4, tt[4]);
            uint32_t cnt[16]; cnt[0] = 0;
            for (uint32_t L = 1; L <= 15u; L++) cnt[L] = (tt[(L - 1u) / 3u] >> (10u * ((L - 1u) % 3u))) & 1023u;
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
            W_SYNC();
            // ranks -> canonical order
            W_LANES {
                const uint32_t ex[5] = {PL(e0), PL(e1), PL(e2), PL(e3), PL(e4)};
                for (uint32_t j = 0; j < K; j++) {
                    const uint32_t L = (PL(lp) >> (4u * j)) & 15u;
                    if (L != 0u) {
                        uint32_t rank = (ex[(L - 1u) / 3u] >> (10u * ((L - 1u) % 3u))) & 1023u;
                        for (uint32_t j2 = 0; j2 < j; j2++) rank += ((PL(lp) >> (4u * j2)) & 15u) == L ? 1u : 0u;
                        const uint32_t sy = K * (uint32_t)lane + j, at = tb[48 * alpha + 32 + L] + rank;
                        if (!alpha) sll[at] = (uint16_t)hw_ll_entry(sy); else sd[at] = hw_d_entry(sy);
                    }
                }
            }
            W_SYNC();
            // direct table: R root bits; lane owns the codes w in [lane * chunk, (lane + 1) * chunk) (MSB-first prefixes), entry index = bit
            // reversal.  Spans of lanes and of symbols are aligned powers of two: a lane's span is one symbol's, or made of whole symbols.
            const uint32_t R = maxlen < (alpha ? HW_RD : HW_RLL) ? (maxlen ? maxlen : 1u) : (alpha ? HW_RD : HW_RLL);
            if (alpha) rd = R; else rll = R;
            const uint32_t size = 1u << R, chunk = size >= 64u ? size >> 6 : 1u;
            // second level (literal/length only): the prefixes w_long .. of codes longer than the root, 2^sb entries each
            const uint32_t w_long = t_lim[R] >> (15u - R);
            uint32_t sb = 0, npref = 0;
            if (!alpha && maxlen > R) {
                npref = ((t_lim[15] + (1u << (15u - R)) - 1u) >> (15u - R)) - w_long;
                sb = maxlen - R;
                if (sb > 4u || (npref << sb) > HW_SUB_ENTRIES) sb = 0;       // (too many long codes for the second level: canonical arithmetic per symbol)
                subbits = sb;
            }
            W_LANES {
                if ((uint32_t)lane * chunk < size) {
                    uint32_t w = (uint32_t)lane * chunk, L = 1;
                    const uint32_t wend = w + chunk;
                    uint32_t lim = tb[48 * alpha + 1], fst = tb[48 * alpha + 16 + 1], off = tb[48 * alpha + 32 + 1];
                    while (w < wend) {
                        const uint32_t w15 = w << (15u - R);
                        while (L <= R && w15 >= lim) { L++; if (L <= R) { lim = tb[48 * alpha + L]; fst = tb[48 * alpha + 16 + L]; off = tb[48 * alpha + 32 + L]; } }
                        uint32_t ent, n;
                        if (L <= R) {
                            const uint32_t si = off + ((w15 >> (15u - L)) - fst);
                            if (!alpha) { const uint32_t v = sll[si]; ent = (v == 0xffffu) ? 0u : (v | L); }
                            else { const uint32_t v = sd[si]; ent = (v == 0xffffffffu) ? 0u : (v | L); }
                            n = 1u << (R - L); if (n > wend - w) n = wend - w;
                        } else {
                            // a longer code starts with this prefix (second-level table number w - w_long, or canonical arithmetic) / unused code space
                            ent = (w15 < tb[48 * alpha + 15]) ? (sb ? (0x30u | ((w - w_long) << 6)) : HW_LONG) : 0u;
                            n = 1u;
                        }
                        for (uint32_t k = 0; k < n; k++) {
                            const uint32_t ix = w_brev32(w + k) >> (32u - R);
                            if (!alpha) lut_ll[ix] = (uint16_t)ent; else lut_d[ix] = ent;
                        }
                        w += n;
                    }
                }
            }
            if (sb) {
                uint16_t *sub = (uint16_t *)(smem + HW_OFF_SUB);
                const uint32_t nent = npref << sb, M = maxlen;
                W_LANES {
                    for (uint32_t t = (uint32_t)lane; t < nent; t += 64u) {
                        // entry t: prefix w_long + (t >> sb), followed by the sb stream bits (t & mask), i.e. their reversal in code order
                        const uint32_t w = w_long + (t >> sb), sbits = t & ((1u << sb) - 1u);
                        const uint32_t cm = (w << sb) | (w_brev32(sbits) >> (32u - sb)), w15 = cm << (15u - M);
                        uint32_t L = R + 1u;
                        while (L <= M && w15 >= tb[L]) L++;
                        uint32_t ent = 0;
                        if (L <= M) { const uint32_t si = tb[32 + L] + ((w15 >> (15u - L)) - tb[16 + L]); const uint32_t v = si < 288u ? sll[si] : 0xffffu; ent = (v == 0xffffu) ? 0u : (v | L); }
                        sub[t] = (uint16_t)ent;
                    }
                }
            }
            W_SYNC();
        }
        if (status != 0) break;
        const uint32_t mask_ll = (1u << rll) - 1u, mask_d = (1u << rd) - 1u;
        HWD_T(t_h2); HWD_ADD(1, t_h1, t_h2);
#ifdef HW_EXP_STOP_AFTER_TABLES          /* instruction-count experiment (tools/dbg/pmc_sq.sh): header and tables only, output invalid */
        status = DHTS_BLK_ERR_INFLATE; break;
#endif

        // ---- symbols: segments of up to 64 bit ranges until the end-of-block symbol ----
        uint32_t p0 = hpos;
        bool eob_seen = false;
        bool first_seg = outpos == 0u && nlit_tot == 0u && ntok_tot == 0u;      // the block's first DEFLATE block: its window starts empty
        bool force_even = false;
        while (!eob_seen && status == 0) {
            if (p0 > limit_bits) { status = DHTS_BLK_ERR_INFLATE; break; }
            const uint32_t span = limit_bits - p0;
            uint32_t S = (span + 63u) / 64u; if (S < HW_MIN_S) S = HW_MIN_S;
            uint32_t nlanes = (span + S - 1u) / S; if (nlanes < 1u) nlanes = 1u;            // <= 64
#if defined(HOSTSIM_W) && defined(HW_STATS)
            g_hw_stat_seg++;
#endif
            // ---- where the ranges are cut ----
            // Equal BIT ranges are not equal work: a BGZF block is compressed on its own, so its first DEFLATE block starts with an empty
            // window -- mostly literals, i.e. many short units per bit -- and the first lanes of an equal cut decode a third more units than
            // the rest while the wave waits for them (host simulation on the bench data: 90 / 85 / 81 / 76 loop iterations in lanes 0..3
            // against 64 in the rest).  Every lane therefore remembers how many units per bit ITS range of the previous blocks held
            // (`dens`, units per 4,096 bits, a running mean over the blocks this wave has decoded) and the first segment of a block is cut
            // so that the expected units are equal: lane i gets bits in proportion to 1 / dens_i.  Only the cut moves; what is decoded
            // does not depend on it.  Later segments (the small trailing DEFLATE blocks: full window) are cut evenly.
            PLD(uint32_t, cut0); PLD(uint32_t, cut1);                 // the lane's range [cut0, cut1) (nominal; cut1 of the last lane is never used as a stop)
            PLD(uint32_t, syncw);                                     // bits decoded in front of cut1 for the proposal: HW_SYNC_W at the mean density, fewer where units are short
            const bool profiled = first_seg && !force_even && S >= 1024u && nlanes == 64u;
            bool used_profile = false;
            {
                PLD(uint32_t, wgt); PLD(uint32_t, cw); uint32_t wtot;
                // (lanes that have not seen a range of their own yet -- the ones behind the first block's end-of-block symbol -- take the mean)
                uint64_t have; W_BALLOT(have, PL(dens) != 0u);
                const bool use = profiled && w_popc64(have) >= 32u;
                W_LANES { PL(wgt) = (use && PL(dens) != 0u) ? (1u << 22) / (PL(dens) < 64u ? 64u : PL(dens)) : 0u; }
                W_EXCL_SCAN(cw, wgt, wtot);
                const uint32_t wmean = use ? wtot / w_popc64(have) : 1024u;
                // (no range more than 1.5 times the even one: the lanes' staging slices hold four times the average, as before)
                W_LANES { if (!use || PL(dens) == 0u) PL(wgt) = wmean; else PL(wgt) = PL(wgt) > wmean + wmean / 2u ? wmean + wmean / 2u : PL(wgt) < wmean / 2u ? wmean / 2u : PL(wgt); }
                W_EXCL_SCAN(cw, wgt, wtot);
                used_profile = use;
                W_LANES {
                    PL(cut0) = use ? p0 + (uint32_t)(((uint64_t)span * PL(cw)) / wtot) : p0 + (uint32_t)lane * S;
                    uint32_t sw = use ? (uint32_t)(((uint64_t)HW_SYNC_W * PL(wgt)) / wmean) : HW_SYNC_W;
                    PL(syncw) = sw < HW_SYNC_W / 2u ? HW_SYNC_W / 2u : sw > HW_SYNC_W + HW_SYNC_W / 4u ? HW_SYNC_W + HW_SYNC_W / 4u : sw;
                }
                W_SYNC();
                W_LANES { xch[HX_A * 64 + lane] = PL(cut0); }
                W_SYNC();
                W_LANES { PL(cut1) = lane < 63 ? xch[HX_A * 64 + lane + 1] : p0 + 64u * S; if (!use) PL(cut1) = p0 + ((uint32_t)lane + 1u) * S; }
                W_SYNC();
            }
            // pass 0: propose the start of lane i + 1 from the last HW_SYNC_W bits of range i
            HWD_T(t_s0); HWD_CNT(8, 1);
            PLD(HwLane, ln); PLD(uint32_t, prop);
            W_LANES {
                PL(ln).start = PL(cut0); PL(ln).end = 0; PL(ln).flags = 0;
                PL(ln).nlit = 0; PL(ln).ntok = 0; PL(ln).run = 0; PL(ln).outb = 0;
                PL(prop) = PL(cut1);                                // what lane i proposes for lane i + 1: the nominal boundary unless its decoder finds better
            }
            W_SYNC();
            W_LANES {
                if ((uint32_t)lane + 1u < nlanes) {
                    const uint32_t bnd = PL(cut1);
                    const uint32_t from = bnd - PL(cut0) > PL(syncw) ? bnd - PL(syncw) : PL(cut0);
                    HwLane tmp;
                    hw_span<0>(smem, in32, from, bnd, limit_bits, mask_ll, mask_d, rll, rd, subbits, lane, tmp, nullptr, nullptr, 0, 0, 0);
                    if (tmp.flags == 0u) PL(prop) = tmp.end;
#if defined(HOSTSIM_W) && defined(HW_STATS)
                    g_hw_stat_p0n++; if (tmp.flags != 0u) g_hw_stat_p0bad++;
#endif
                }
            }
#if defined(HOSTSIM_W) && defined(HW_STATS)
            { unsigned long long mx = 0; for (int q = 0; q + 1 < (int)nlanes; q++) { g_hw_stat_itsum[0] += g_hw_it[q]; if (g_hw_it[q] > mx) mx = g_hw_it[q]; } g_hw_stat_itmax[0] += mx; for (int q = 0; q < 64; q++) g_hw_it[q] = 0; }
#endif
            W_SYNC();
            W_LANES { xch[HX_START * 64 + lane] = PL(prop); }
            W_SYNC();
            W_LANES { if (lane > 0) PL(ln).start = xch[HX_START * 64 + lane - 1]; }
            W_SYNC();
            HWD_T(t_s1); HWD_ADD(2, t_s0, t_s1);
            // pass 1 until the chain of confirmed lanes reaches the end-of-block symbol or the last lane
            uint64_t dirty = nlanes >= 64u ? ~0ull : ((1ull << nlanes) - 1ull);
            uint32_t n_ok = 0;                               // lanes 0 .. n_ok-1 are confirmed and belong to the segment
            for (int guard = 0; guard < 66; guard++) {
                HWD_CNT(9, 1);
#if defined(HOSTSIM_W) && defined(HW_STATS)
                g_hw_stat_p1[guard < 7 ? guard : 7]++; g_hw_stat_dirty += (unsigned long long)w_popc64(dirty);
                if (S >= 1024u) g_hw_stat_big[guard < 7 ? guard : 7]++;
#endif
                W_LANES {
                    if ((dirty >> lane) & 1ull) {
                        const uint32_t bnd = (uint32_t)lane + 1u < nlanes ? PL(cut1) : 0xffffffffu;   // the last lane runs to the end-of-block symbol
                        // (lane 0 continues the literal run that is open at the start of the segment)
                        hw_span<1>(smem, in32, PL(ln).start, bnd, limit_bits, mask_ll, mask_d, rll, rd, subbits, lane, PL(ln),
                                   slit + (uint32_t)lane * HW_LANE_LIT, stok + (uint32_t)lane * HW_LANE_TOK, HW_LANE_LIT, HW_LANE_TOK, lane == 0 ? run : 0u);
                        // a range whose first unit starts at or beyond its boundary holds nothing: it ends where it starts
                    }
                }
#if defined(HOSTSIM_W) && defined(HW_STATS)
                { unsigned long long mx = 0; for (int q = 0; q < 64; q++) if ((dirty >> q) & 1ull) { g_hw_stat_itsum[1] += g_hw_it[q]; if (g_hw_it[q] > mx) mx = g_hw_it[q]; } g_hw_stat_itmax[1] += mx;
                  if (getenv("HW_DUMP_IT") && guard == 0 && S >= 1024u && g_hw_stat_seg < 12) { printf("seg S=%u:", S); for (int q = 0; q < 64; q++) printf(" %llu", g_hw_it[q]); printf("\n"); }
                  for (int q = 0; q < 64; q++) g_hw_it[q] = 0; }
#endif
                W_SYNC();
                W_LANES { xch[HX_END * 64 + lane] = PL(ln).end; xch[HX_FLAG * 64 + lane] = PL(ln).flags; }
                W_SYNC();
                // link i: lane i starts where lane i-1 ended (and lane i-1 went on: no end-of-block, no error)
                uint64_t linked, stop;
                W_BALLOT(linked, lane == 0 || ((uint32_t)lane < nlanes && (xch[HX_FLAG * 64 + lane - 1] & (HWF_EOB | HWF_BAD)) == 0u && xch[HX_END * 64 + lane - 1] == PL(ln).start));
                W_BALLOT(stop, (uint32_t)lane < nlanes && (PL(ln).flags & (HWF_EOB | HWF_BAD)) != 0u);
                const uint32_t k_conf = (uint32_t)w_ctz64(~linked);                     // lanes 0 .. k_conf-1 are confirmed
                const uint64_t conf_mask = k_conf >= 64u ? ~0ull : ((1ull << k_conf) - 1ull);
                if (stop & conf_mask) { n_ok = (uint32_t)w_ctz64(stop & conf_mask) + 1u; break; }      // the first confirmed lane that stopped ends the segment
                if (k_conf >= nlanes) { n_ok = nlanes; break; }
                // restart every lane behind a broken link from its predecessor's end (the first of them becomes confirmed next round)
                uint64_t nd_;
                W_BALLOT(nd_, (uint32_t)lane >= k_conf && (uint32_t)lane < nlanes && (xch[HX_FLAG * 64 + lane - 1] & (HWF_EOB | HWF_BAD)) == 0u && xch[HX_END * 64 + lane - 1] != PL(ln).start);
#if defined(HOSTSIM_W) && defined(HW_STATS)
                if (guard == 0) for (int lane = 1; lane < 64; lane++) if ((nd_ >> lane) & 1ull) { g_hw_stat_wrong++; if (PL(ln).start == PL(cut0)) g_hw_stat_wrong_nominal++; }
#endif
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
            uint64_t ovf;
            W_BALLOT(ovf, (uint32_t)lane < n_ok && (PL(ln).flags & HWF_OVF) != 0u);
            if (ovf && used_profile) { force_even = true; continue; }       // (an uneven cut overfilled a slice: the segment again with the even cut, nothing has been placed yet)
            if (ovf) {
                // ---- a staging slice was too small: lane 0 decodes the whole segment and appends to the block's slot itself ----
                HWD_CNT(10, 1);
#if defined(HOSTSIM_W) && defined(HW_STATS)
                g_hw_stat_fallback++;
#endif
                const uint32_t lcap = nlit_tot < DHTS_LIT_STRIDE ? DHTS_LIT_STRIDE - nlit_tot : 0u, tcap = ntok_tot + 64u < DHTS_TOK_STRIDE ? DHTS_TOK_STRIDE - 64u - ntok_tot : 0u;
                W_LANES {
                    if (lane == 0) hw_span<1>(smem, in32, p0, 0xffffffffu, limit_bits, mask_ll, mask_d, rll, rd, subbits, lane, PL(ln), lit + nlit_tot, tok + ntok_tot, lcap, tcap, run);
                }
                W_SYNC();
                W_LANES { if (lane == 0) { xch[HX_END * 64] = PL(ln).end; xch[HX_FLAG * 64] = PL(ln).flags; xch[HX_A * 64] = PL(ln).nlit; xch[HX_B * 64] = PL(ln).ntok; xch[HX_C * 64] = PL(ln).run; xch[HX_START * 64] = PL(ln).outb; } }
                W_SYNC();
                const uint32_t f0 = xch[HX_FLAG * 64];
                if (f0 != HWF_EOB || xch[HX_END * 64] != seg_end) { status = DHTS_BLK_ERR_INFLATE; break; }      // (an overflow here means more output than a block may hold)
                if (outpos + xch[HX_START * 64] > 65536u) { status = DHTS_BLK_ERR_INFLATE; break; }
                nlit_tot += xch[HX_A * 64]; ntok_tot += xch[HX_B * 64]; run = xch[HX_C * 64]; outpos += xch[HX_START * 64];
                p0 = seg_end; first_seg = false;
                W_SYNC();
                HWD_T(t_sf); HWD_ADD(5, t_s2, t_sf);
                continue;
            }
            if (profiled) {
                // what the lanes' ranges held: units per 4,096 bits, folded into the running mean (lanes beyond the end-of-block symbol keep theirs)
                W_LANES {
                    if ((uint32_t)lane < n_ok && PL(ln).end > PL(ln).start) {
                        const uint32_t bits = PL(ln).end - PL(ln).start, units = PL(ln).nlit + PL(ln).ntok;
                        const uint32_t d = (uint32_t)(((uint64_t)units << 12) / bits) + 1u;
                        PL(dens) = PL(dens) ? (3u * PL(dens) + d + 2u) >> 2 : d;
                    }
                }
            }
            first_seg = false;
            // ---- places: exclusive sums over the lanes of the segment ----
            PLD(uint32_t, v_nlit); PLD(uint32_t, v_out); PLD(uint32_t, v_run); PLD(uint32_t, v_ntok); PLD(uint32_t, o_lit); PLD(uint32_t, o_out); PLD(uint32_t, o_run); PLD(uint32_t, o_tok);
            PLD(uint32_t, plead); PLD(uint32_t, extra); PLD(uint32_t, newrun);
            uint32_t tot_lit, tot_out, tot_run, tot_tok;
            W_LANES { const bool in = (uint32_t)lane < n_ok; PL(v_nlit) = in ? PL(ln).nlit : 0u; PL(v_out) = in ? PL(ln).outb : 0u; PL(v_run) = in ? PL(ln).run : 0u; }
            W_EXCL_SCAN(o_lit, v_nlit, tot_lit);
            W_EXCL_SCAN(o_out, v_out, tot_out);
            W_EXCL_SCAN(o_run, v_run, tot_run);
            (void)o_out;
            if (outpos + tot_out > 65536u) { status = DHTS_BLK_ERR_INFLATE; break; }
            // the lane's first match token (behind its own "511 literals" tokens, if it has any): read back from the slice
            W_LANES {
                uint32_t k = 0; const uint32_t n = (uint32_t)lane < n_ok ? PL(ln).ntok : 0u;
                const uint32_t *st = stok + (uint32_t)lane * HW_LANE_TOK;
                while (k < n && st[k] == (DHTS_TOK_PURE << 23)) k++;
                PL(plead) = k;
                PL(newrun) = k < n ? st[k] >> 23 : 0u;       // (for now: the token's own literal run)
            }
            uint64_t has_match;
            W_BALLOT(has_match, (uint32_t)lane < n_ok && PL(plead) < PL(ln).ntok);
            W_LANES { xch[HX_A * 64 + lane] = PL(o_run); }
            W_SYNC();
            W_LANES {
                // literals that precede this lane's first unit and follow the last token before it: the runs left open by the lanes
                // from the last one with a match (lane 0's run includes what the previous segment left open)
                const uint64_t lower = has_match & ((1ull << lane) - 1ull);
                const uint32_t c = lower == 0ull ? PL(o_run) : PL(o_run) - xch[HX_A * 64 + w_msb64(lower)];
                const bool hm = ((has_match >> lane) & 1ull) != 0ull;
                const uint32_t t = c + PL(newrun);
                PL(extra) = hm ? t / DHTS_TOK_PURE : 0u;
                PL(newrun) = t % DHTS_TOK_PURE;
                PL(v_ntok) = (uint32_t)lane < n_ok ? PL(ln).ntok + PL(extra) : 0u;
            }
            W_EXCL_SCAN(o_tok, v_ntok, tot_tok);
            // the run that is still open behind the segment's last token
            if (has_match) run = tot_run - xch[HX_A * 64 + w_msb64(has_match)]; else run = tot_run;
            const uint32_t q_end = run / DHTS_TOK_PURE;
            if (ntok_tot + tot_tok + q_end + 64u > DHTS_TOK_STRIDE || nlit_tot + tot_lit > DHTS_LIT_STRIDE) { status = DHTS_BLK_ERR_INFLATE; break; }     // (more than a valid block can hold)
            HWD_T(t_s3); HWD_ADD(4, t_s2, t_s3);
            // ---- every lane moves its slice to its place in the block's slot ----
            W_LANES {
#ifdef HW_EXP_NOSLICE           /* knock-out experiment: what the slice -> block area copies cost; the output is invalid */
                if (false) {
#else
                if ((uint32_t)lane < n_ok) {
#endif
                    const uint32_t *st = stok + (uint32_t)lane * HW_LANE_TOK;
                    uint32_t *dt = tok + ntok_tot + PL(o_tok);
                    for (uint32_t k = 0; k < PL(extra); k++) dt[k] = DHTS_TOK_PURE << 23;
                    dt += PL(extra);
                    const uint32_t n = PL(ln).ntok;
                    uint32_t k = 0;
                    // (four 16-byte pieces in flight: the slice was written by this lane a moment ago and comes from the caches)
                    for (; k + 16u <= n; k += 16u) { uint4 v[4]; for (int u = 0; u < 4; u++) __builtin_memcpy(&v[u], st + k + 4 * u, 16); for (int u = 0; u < 4; u++) __builtin_memcpy(dt + k + 4 * u, &v[u], 16); }
                    for (; k + 4u <= n; k += 4u) { uint4 v; __builtin_memcpy(&v, st + k, 16); __builtin_memcpy(dt + k, &v, 16); }
                    for (; k < n; k++) dt[k] = st[k];
                    const uint8_t *sl = slit + (uint32_t)lane * HW_LANE_LIT;
                    uint8_t *dl = lit + nlit_tot + PL(o_lit);
                    const uint32_t m = PL(ln).nlit;
                    uint32_t j = 0;
                    for (; j + 64u <= m; j += 64u) { uint4 v[4]; for (int u = 0; u < 4; u++) __builtin_memcpy(&v[u], sl + j + 16 * u, 16); for (int u = 0; u < 4; u++) __builtin_memcpy(dl + j + 16 * u, &v[u], 16); }
                    for (; j + 16u <= m; j += 16u) { uint4 v; __builtin_memcpy(&v, sl + j, 16); __builtin_memcpy(dl + j, &v, 16); }
                    for (; j < m; j++) dl[j] = sl[j];
                }
            }
            W_SYNC();
            W_LANES {
                // the first match token takes the literals that the lanes in front of it left open
                if ((uint32_t)lane < n_ok && PL(plead) < PL(ln).ntok) {
                    const uint32_t t0 = stok[(uint32_t)lane * HW_LANE_TOK + PL(plead)];
                    tok[ntok_tot + PL(o_tok) + PL(extra) + PL(plead)] = (t0 & 0x007fffffu) | (PL(newrun) << 23);
                }
            }
            nlit_tot += tot_lit; ntok_tot += tot_tok; outpos += tot_out;
            {
                W_LANES { for (uint32_t k = (uint32_t)lane; k < q_end; k += 64u) tok[ntok_tot + k] = DHTS_TOK_PURE << 23; }
                ntok_tot += q_end; run -= q_end * DHTS_TOK_PURE;
            }
            p0 = seg_end;
            W_SYNC();
            HWD_T(t_s4); HWD_ADD(5, t_s3, t_s4);
        }
        pos = p0;
    }
    if (status == 0 && pos > limit_bits) status = DHTS_BLK_ERR_INFLATE;
    mres.ntok = ntok_tot; mres.nlit = nlit_tot; mres.outlen = outpos; mres.status = status;
    W_SYNC();
    { HWD_T(t_end); HWD_ADD(6, t_begin, t_end); HWD_CNT(7, 1); }
}

#ifndef HOSTSIM_W
// The launches: `grid` workgroups of one wave; workgroup g starts with block g and then takes blocks from *counter (set to `grid` by the
// host before the launch) until the range is exhausted -- every wave reaches the exit test after each block.
//
// bgzf_huff_decode_wave: phase A.  A block is assembled in the workgroup's own area (`wg_lit` / `wg_tok`, reused for every block the
// workgroup takes, i.e. cache-resident), then exactly the room it needs -- its literal bytes rounded up to 16 + 4 bytes per token -- is
// taken from `pool` with one atomic add and the block moves there in coalesced 16-byte pieces: the scratch a later bgzf_lz_resolve launch
// reads is packed (34 KB per block of a 30x BAM instead of a 152 KiB slot) and is written in whole lines.  blk_off[b] = the block's
// offset in the pool.  A block that finds the pool exhausted is marked DHTS_BLK_ERR_SCRATCH (the host repeats the range with more room).
// With pool == nullptr the blocks go to fixed slots of lit_all / tok_all (the format of the one-lane-per-block kernel: cross-check path).
extern "C" __global__ void __launch_bounds__(64)
bgzf_huff_decode_wave(const uint8_t *__restrict__ comp, BgzfTable tab, int64_t blk0, int32_t nblk,
                      uint8_t *__restrict__ lit_all, uint32_t *__restrict__ tok_all, InflateMeta *__restrict__ meta,
                      uint8_t *__restrict__ stage_lit, uint32_t *__restrict__ stage_tok, uint32_t *__restrict__ counter,
                      uint8_t *__restrict__ pool, unsigned long long pool_cap, unsigned long long *__restrict__ pool_used, unsigned long long *__restrict__ blk_off,
                      uint8_t *__restrict__ wg_lit, uint32_t *__restrict__ wg_tok) {
    __shared__ __attribute__((aligned(16))) uint8_t smem[HW_LDS_BYTES];
    uint8_t *slit = stage_lit + (size_t)blockIdx.x * HW_STAGE_LIT_BYTES;
    uint32_t *stok = stage_tok + (size_t)blockIdx.x * HW_STAGE_TOK_WORDS;
#if defined(HW_DIAG)
    unsigned long long hwd[16]; for (int q_ = 0; q_ < 16; q_++) hwd[q_] = 0;
#else
    unsigned long long *hwd = nullptr;
#endif
    const int lane = threadIdx.x;
    uint32_t b = blockIdx.x;
    uint32_t dens = 0;                                       // the lane's units per 4,096 bits in the blocks decoded so far (hw_block: where the ranges are cut)
    while (b < (uint32_t)nblk) {
        InflateMeta m;
        if (pool == nullptr) {
            hw_block(smem, blk0 + (int64_t)b, comp, tab, lit_all + (size_t)b * DHTS_LIT_STRIDE, tok_all + (size_t)b * DHTS_TOK_STRIDE, m, slit, stok, hwd, dens);
        } else {
            uint8_t *wl = wg_lit + (size_t)blockIdx.x * (DHTS_LIT_STRIDE + 64u);
            uint32_t *wt = wg_tok + (size_t)blockIdx.x * DHTS_TOK_STRIDE;
            hw_block(smem, blk0 + (int64_t)b, comp, tab, wl, wt, m, slit, stok, hwd, dens);
            HWD_T(t_p0);
            if (m.status == 0) {
                const uint32_t lbytes = (m.nlit + 15u) & ~15u, need = lbytes + 4u * m.ntok;
                unsigned long long off = 0;
                if (lane == 0) off = atomicAdd(pool_used, (unsigned long long)((need + 63u) & ~63u));
                off = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(off >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)off);
                if (off + need > pool_cap) m.status = DHTS_BLK_ERR_SCRATCH;
                else {
                    uint8_t *dl = pool + off; uint8_t *dt = dl + lbytes;
                    // (eight 16-byte pieces per lane in flight: the area was written a moment ago and comes from the caches, but every piece
                    //  is still a round trip)
                    const uint32_t tbytes = 4u * m.ntok;
#ifdef HW_EXP_NOPOOL            /* knock-out experiment (tools/dbg/time_huff.py): what the move to the pool costs; the output is invalid */
                    for (int part = 2; part < 2; part++) {
#else
                    for (int part = 0; part < 2; part++) {
#endif
                        const uint8_t *sp = part ? (const uint8_t *)wt : wl; uint8_t *dp = part ? dt : dl;
                        const uint32_t nbytes = part ? (tbytes & ~15u) : lbytes;
                        uint32_t k = 16u * (uint32_t)lane;
                        for (; k + 7u * 1024u < nbytes; k += 8u * 1024u) {
                            uint4 v[8];
                            for (int u = 0; u < 8; u++) __builtin_memcpy(&v[u], sp + k + 1024u * (uint32_t)u, 16);
                            for (int u = 0; u < 8; u++) __builtin_memcpy(dp + k + 1024u * (uint32_t)u, &v[u], 16);
                        }
                        for (; k < nbytes; k += 1024u) { uint4 v; __builtin_memcpy(&v, sp + k, 16); __builtin_memcpy(dp + k, &v, 16); }
                    }
                    if (lane < (int)(m.ntok & 3u)) ((uint32_t *)dt)[(m.ntok & ~3u) + (uint32_t)lane] = wt[(m.ntok & ~3u) + (uint32_t)lane];
                    if (lane == 0) blk_off[b] = off;
                }
            }
            __syncthreads();                                   // the copies have read the workgroup's area before the next block overwrites it
            HWD_T(t_p1); HWD_ADD(15, t_p0, t_p1);
        }
        if (lane == 0) meta[b] = m;
        uint32_t nb_ = 0;
        if (lane == 0) nb_ = atomicAdd(counter, 1u);
        b = (uint32_t)__builtin_amdgcn_readfirstlane((int)nb_);
    }
#if defined(HW_DIAG)
    if (threadIdx.x == 0) for (int q_ = 0; q_ < 16; q_++) if (hwd[q_]) atomicAdd(&g_hw_diag[q_], hwd[q_]);
#endif
}

// bgzf_inflate_fused (DHTS_INFLATE=fused; measured slower than the two launches: 202 ms against 178 ms per 92 M-record step, because phase B
// then runs at phase A's ten waves per CU instead of sixteen).  The wave that decoded a block's Huffman symbols resolves its LZ77 copies right away
// (lz_block, bgzf_inflate.hip): literals and tokens never leave the workgroup's own 152 KiB area (`wg_lit` / `wg_tok`, reused for every
// block the workgroup takes, i.e. cache-resident), so the inflate stage reads the compressed block and writes the inflated one -- there
// is no per-block scratch.  The two phases use the same LDS (phase A's tables are dead when phase B builds its window).
#define HWF_LDS_BYTES (HW_LDS_BYTES > B_LDS_BYTES ? HW_LDS_BYTES : B_LDS_BYTES)
extern "C" __global__ void __launch_bounds__(64)
bgzf_inflate_fused(const uint8_t *__restrict__ comp, BgzfTable tab, int64_t blk0, int32_t nblk,
                   uint8_t *__restrict__ wg_lit, uint32_t *__restrict__ wg_tok, uint8_t *__restrict__ stage_lit, uint32_t *__restrict__ stage_tok,
                   uint32_t *__restrict__ counter, uint8_t *__restrict__ out, uint64_t out_base, int32_t *__restrict__ blk_status) {
    __shared__ __attribute__((aligned(16))) uint8_t smem[HWF_LDS_BYTES];
    uint8_t *slit = stage_lit + (size_t)blockIdx.x * HW_STAGE_LIT_BYTES;
    uint32_t *stok = stage_tok + (size_t)blockIdx.x * HW_STAGE_TOK_WORDS;
    uint8_t *lit = wg_lit + (size_t)blockIdx.x * (DHTS_LIT_STRIDE + 64u);
    uint32_t *tok = wg_tok + (size_t)blockIdx.x * DHTS_TOK_STRIDE;
#if defined(HW_DIAG)
    unsigned long long hwd[16]; for (int q_ = 0; q_ < 16; q_++) hwd[q_] = 0;
#else
    unsigned long long *hwd = nullptr;
#endif
    uint32_t b = blockIdx.x;
    uint32_t dens = 0;
    while (b < (uint32_t)nblk) {
        InflateMeta m;
        hw_block(smem, blk0 + (int64_t)b, comp, tab, lit, tok, m, slit, stok, hwd, dens);
        __syncthreads();                                       // the block's tokens and literals are stored; phase A's LDS is free
        lz_load_crc_tables((uint32_t *)(smem + B_CRCT));
        __syncthreads();
        lz_block(smem + B_WIN, (uint32_t *)(smem + B_CRCT), smem + B_RING, comp, tab, blk0 + (int64_t)b, m, lit, tok, out, out_base, blk_status);
        __syncthreads();
        uint32_t nb_ = 0;
        if (threadIdx.x == 0) nb_ = atomicAdd(counter, 1u);
        b = (uint32_t)__builtin_amdgcn_readfirstlane((int)nb_);
    }
#if defined(HW_DIAG)
    if (threadIdx.x == 0) for (int q_ = 0; q_ < 16; q_++) if (hwd[q_]) atomicAdd(&g_hw_diag[q_], hwd[q_]);
#endif
}
#endif
#endif
