// bgzf_inflate.hip -- BGZF block inflate for MI355X (gfx950), written from RFC 1951/1952.
//
// Replaces htslib bgzf.c:762-824 (bgzf_uncompress/inflate_block over zlib) and the CRC
// compare at bgzf.c:793-801.  Two kernels, both integer/bit work:
//
//  phase A  bgzf_huff_decode : ONE LANE PER BGZF BLOCK (64 independent DEFLATE streams per
//           wave64).  Each lane walks its stream with canonical-code arithmetic (no big
//           LUTs): the 15 left-justified code limits of BOTH alphabets live in the two 16-bit
//           halves of 15 VGPRs, the canonical base values in 15 more (telescoped deltas), the
//           sorted symbol lists in LDS ([entry][lane] layout, 20,480 B per wave => 8 waves per
//           CU: the 196 shortest literal/length codes, the rest are fetched through L2 when they occur; the tables are built in two passes over the code-length stream so that no
//           per-symbol length array has to be kept).  One symbol per lane per iteration from whichever alphabet the lane expects.
//           Output is append-only: literal bytes and one 32-bit token per LZ77 match.  No
//           loads depend on earlier stores, so lanes never stall on the LZ77 window.
//  phase B  bgzf_lz_resolve  : ONE WAVE PER BGZF BLOCK.  Tokens are consumed 64 at a time
//           (coalesced), wave prefix sums give every token's destination, literals are
//           placed lane-parallel, matches are replayed in dependency rounds in a 4 KiB LDS
//           ring (older sources are read back from the block's own flushed output), every
//           half ring is CRC-32'd from LDS (slice-by-4, per-lane pieces combined with
//           x^(8n) mod P) and flushed to HBM with 16-byte coalesced stores.
#include "dhts_common.h"

// ------------------------------------------------------------------------------------
// phase A
// ------------------------------------------------------------------------------------
#ifndef A_LDS_ROUND
#define A_LDS_ROUND 64
#endif
#ifndef A_SL
#define A_SL 64                          /* streams (active lanes) per wave */
#endif
#ifndef A_ST
#define A_ST A_SL                           /* LDS row stride in lanes */
#endif
// LDS of one wave, [row][lane] (row stride A_ST lanes), for `nlo` sorted literal/length entries kept in LDS:
//   u8  [nlo]          sorted literal/length symbols, low 8 bits           offset 0
//   u32 [(nlo+31)/32]  bit 8 of the same, one bit per entry                A_OFF_HI(nlo)
//   u8  [32]           sorted distance symbols                             A_OFF_DSYM(nlo)
//   u32 [16]           64-byte input window                                A_OFF_WIN(nlo); while the tables are built it holds
//                      u16 [15] literal/length counts -> running offsets of code lengths 1..15, u8 [15] distance ditto, u8 [19] CL symbols
// Two layouts are launched (kernel argument `nlo`):
//   A_NLO_ALL  288: every sorted symbol in LDS, 26,880 B = 420 B per stream, SIX waves per CU.  For launches that fit the machine
//                   at six waves per CU (<= 98,304 blocks): nothing leaves LDS, nothing waits on L2.
//   A_NLO_FAR  196: 20,480 B = 320 B per stream, EIGHT waves per CU (all 160 KB): the longest codes' symbols sit in the slack of
//                   the stream's own token slot in HBM; a lane that decodes one fetches it and handles it one iteration later.
//                   Worth 5-6 % on long launches (two waves per SIMD hide the fetch); on a lone wave the fetch latency is exposed
//                   (a 1 GB BCF became 25 % slower), hence the first layout for small launches.
#define A_NLO_ALL 288u
#define A_NLO_FAR 196u
#define A_HI_ROWS(nlo) (((nlo) + 31u) >> 5)
#define A_OFF_HI(nlo) ((nlo) * A_ST)
#define A_OFF_DSYM(nlo) (A_OFF_HI(nlo) + A_HI_ROWS(nlo) * A_ST * 4u)
#define A_OFF_WIN(nlo) (A_OFF_DSYM(nlo) + 32u * A_ST)
#define A_LDS_BYTES_FOR(nlo) (A_OFF_WIN(nlo) + 16u * A_ST * 4u)
#define A_LDS_BYTES A_LDS_BYTES_FOR(A_NLO_ALL)       /* the larger of the two: what the kernel attribute must allow */
#define A_FAR_SLOT (DHTS_TOK_STRIDE - 64u) /* the far symbols (u16 [92]) sit in the slack at the end of the stream's own token slot */
static_assert(288u - A_NLO_FAR <= 128u, "far table must fit 64 dwords");

struct BitR {
    const uint8_t *p;   // stream base (deflate payload start)
    uint32_t pos;       // bytes of the stream already moved into buf (excludes the two look-ahead words)
    uint32_t lim;       // bytes that may be touched (payload + trailer)
    uint64_t buf;
    uint32_t cnt;
    uint32_t w0, w1;    // two 32-bit words fetched ahead: the global-load latency overlaps 64 bits of decoding
};

__device__ __forceinline__ uint32_t ld32_guard(const uint8_t *p, uint32_t pos, uint32_t lim) {
    if (pos + 4 <= lim) { uint32_t v; __builtin_memcpy(&v, p + pos, 4); return v; }
    uint32_t v = 0;
    for (int k = 0; k < 4; k++) if (pos + k < lim) v |= (uint32_t)p[pos + k] << (8 * k);
    return v;
}
__device__ __forceinline__ void br_init(BitR &b) {
    b.pos = 0; b.buf = 0; b.cnt = 0;
    b.w0 = ld32_guard(b.p, 0, b.lim); b.w1 = ld32_guard(b.p, 4, b.lim);
}
__device__ __forceinline__ void br_refill(BitR &b) {
    if (b.cnt <= 32) {
        b.buf |= (uint64_t)b.w0 << b.cnt; b.pos += 4; b.cnt += 32;
        b.w0 = b.w1; b.w1 = ld32_guard(b.p, b.pos + 4, b.lim);
    }
}
__device__ __forceinline__ uint32_t br_take(BitR &b, uint32_t n) {   // n <= 32, caller refilled
    uint32_t v = (uint32_t)b.buf & (uint32_t)((1ull << n) - 1);
    b.buf >>= n; b.cnt -= n;
    return v;
}

// packed 16-bit helpers (v_pk_*): two independent u16 lanes per VGPR, no carries between the halves
#ifdef HOSTSIM
static inline uint32_t pk_subsat_u16(uint32_t a, uint32_t b) { uint32_t lo = (a & 0xffff) > (b & 0xffff) ? (a & 0xffff) - (b & 0xffff) : 0, hi = (a >> 16) > (b >> 16) ? (a >> 16) - (b >> 16) : 0; return lo | (hi << 16); }
static inline uint32_t pk_min_u16(uint32_t a, uint32_t b) { uint32_t lo = (a & 0xffff) < (b & 0xffff) ? (a & 0xffff) : (b & 0xffff), hi = (a >> 16) < (b >> 16) ? (a >> 16) : (b >> 16); return lo | (hi << 16); }
static inline uint32_t pk_add_u16(uint32_t a, uint32_t b) { return ((a + b) & 0xffff) | ((((a >> 16) + (b >> 16)) & 0xffff) << 16); }
static inline uint32_t pk_mad_u16(uint32_t a, uint32_t b, uint32_t c) { return (((a & 0xffff) * (b & 0xffff) + (c & 0xffff)) & 0xffff) | ((((a >> 16) * (b >> 16) + (c >> 16)) & 0xffff) << 16); }
#else
__device__ __forceinline__ uint32_t pk_mad_u16(uint32_t a, uint32_t b, uint32_t c) { uint32_t d; asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ uint32_t pk_subsat_u16(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_min_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ uint32_t pk_add_u16(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_add_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
#endif

// 64-bit logical right shift by n < 32: v_alignbit_b32 + v_lshrrev_b32 (full rate) instead of v_lshrrev_b64 (quarter rate)
__device__ __forceinline__ uint64_t shr64_small(uint64_t v, uint32_t n) {
#ifdef HOSTSIM
    return v >> n;
#else
    const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    return ((uint64_t)(hi >> n) << 32) | __builtin_amdgcn_alignbit(hi, lo, n);
#endif
}

// 15 left-justified limits: lim[L-1] = (first_code[L] + count[L]) << (15 - L)
struct Limits { uint32_t v[15]; };

__device__ __forceinline__ uint32_t code_len(const Limits &lm, uint32_t w15) {
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 15; i++) c += (w15 >= lm.v[i]) ? 1u : 0u;
    return c + 1;   // 16 => invalid code
}

// builds limits + canonical base values (registers) from cnt[L-1] (LDS); leaves cnt[L-1] = first sorted index of length L
// for the placement pass.  Returns the Kraft remainder `left` (0 complete, >0 incomplete, <0 over-subscribed).
template <typename T>
__device__ __forceinline__ int build_limits(Limits &lm, T *cnt, int lane, uint32_t *bs) {
    uint32_t first = 0, offs = 0; int left = 1;
#pragma unroll
    for (int L = 1; L <= 15; L++) {
        const uint32_t c = cnt[(L - 1) * A_ST + lane];
        left = (left << 1) - (int)c;
        bs[L] = (offs - first) & 0xffffu;
        lm.v[L - 1] = (first + c) << (15 - L);
        cnt[(L - 1) * A_ST + lane] = (T)offs;
        offs += c;
        first = (first + c) << 1;
    }
    return left;
}

#ifdef DHTS_DIAG
__device__ unsigned long long g_diagA[8];
#endif
// The body is instantiated twice (see the two LDS layouts above): FAR = false keeps every sorted symbol in LDS and compiles the
// far-symbol path away; FAR = true is the eight-waves-per-CU layout.
template <bool FAR>
__device__ __forceinline__ void huff_decode_body(uint8_t *smem, const uint8_t *__restrict__ comp, BgzfTable tab, int64_t blk0, int32_t nblk,
                                                 uint8_t *__restrict__ lit_all, uint32_t *__restrict__ tok_all, InflateMeta *__restrict__ meta) {
    constexpr uint32_t A_NLO = FAR ? A_NLO_FAR : A_NLO_ALL;
    constexpr uint32_t off_dsym = A_OFF_DSYM(A_NLO), off_win = A_OFF_WIN(A_NLO), hi_rows = A_HI_ROWS(A_NLO);
    uint8_t *lsym_lo = smem;
    uint32_t *lsym_hi = (uint32_t *)(smem + A_OFF_HI(A_NLO));
    uint8_t *dsym = smem + off_dsym;
    uint16_t *cntl = (uint16_t *)(smem + off_win);
    uint8_t *cntd = smem + off_win + 15 * A_ST * 2;
    uint8_t *clsym = smem + off_win + 15 * A_ST * 2 + 15 * A_ST;

    const int lane = threadIdx.x;
    const int64_t s = (int64_t)blockIdx.x * A_SL + lane;
    if (s >= nblk) return;
    const int64_t bi = blk0 + s;
    const uint32_t clen = tab.clen[bi];
    uint8_t *lit = lit_all + (size_t)s * DHTS_LIT_STRIDE;
    uint32_t *tok = tok_all + (size_t)s * DHTS_TOK_STRIDE;
    uint16_t *far_tab = (uint16_t *)(tok + A_FAR_SLOT);

    BitR br;
    br.p = comp + tab.coff[bi] + 18;
    br.lim = clen >= 18 ? clen - 18 : 0;          // payload + 8-byte trailer are readable
    br_init(br);
    const uint32_t payload_bits = clen >= 26 ? (clen - 26) * 8 : 0;

    uint32_t nlit = 0, ntok = 0, outpos = 0, run = 0, litbuf = 0;
    // Output staging: literals and tokens leave a lane 16 bytes at a time (aligned global_store_dwordx4).  A 4-byte store per
    // lane per dword costs the memory pipeline one cache-line transaction per active lane; 16-byte pieces quarter that.
    uint32_t lq0 = 0, lq1 = 0, lq2 = 0, tq0 = 0, tq1 = 0, tq2 = 0;
#ifndef A_EXP_NOSTORE
#define A_ST16(ptr_, a_, b_, c_, d_) do { const uint4 v16_ = make_uint4(a_, b_, c_, d_); *(uint4 *)(ptr_) = v16_; } while (0)
#else
#define A_ST16(ptr_, a_, b_, c_, d_) do { } while (0)
#endif
#define PUSH_LIT(byte_)                                                                              \
    do {                                                                                             \
        litbuf |= (uint32_t)(byte_) << (8u * (nlit & 3u)); nlit++;                                   \
        if ((nlit & 3u) == 0u) {                                                                     \
            const uint32_t q_ = (nlit >> 2) & 3u;                                                    \
            if (q_ == 1u) lq0 = litbuf; else if (q_ == 2u) lq1 = litbuf; else if (q_ == 3u) lq2 = litbuf; \
            else A_ST16(lit + nlit - 16, lq0, lq1, lq2, litbuf);                                     \
            litbuf = 0;                                                                              \
        }                                                                                            \
    } while (0)
#define PUSH_TOK(val_)                                                                               \
    do {                                                                                             \
        const uint32_t q_ = ntok & 3u, tv_ = (val_); ntok++;                                         \
        if (q_ == 0u) tq0 = tv_; else if (q_ == 1u) tq1 = tv_; else if (q_ == 2u) tq2 = tv_;         \
        else A_ST16(tok + ntok - 4, tq0, tq1, tq2, tv_);                                             \
    } while (0)
#ifdef DHTS_DIAG
    unsigned long long dA_t0 = clock64(), dA_sym = 0, dA_it = 0, dA_build = 0;
#endif
    int status = clen >= 26 ? 0 : DHTS_BLK_ERR_INFLATE;
    bool last = (status != 0);
    Limits ll, dl;

#define EMIT_LIT(byte_)                                                                              \
    do {                                                                                             \
        if (outpos >= 65536u) { status = DHTS_BLK_ERR_INFLATE; }                                     \
        else {                                                                                       \
            PUSH_LIT(byte_); outpos++;                                                               \
            if (++run == DHTS_TOK_PURE) { PUSH_TOK(DHTS_TOK_PURE << 23); run = 0; }                  \
        }                                                                                            \
    } while (0)

    while (!last && status == 0) {
        br_refill(br);
        last = br_take(br, 1) != 0;
        uint32_t type = br_take(br, 2);
        if (type == 0) {
            // ---- stored block ----
            br_take(br, br.cnt & 7);                         // to byte boundary
            br_refill(br);
            uint32_t len = br_take(br, 16);
            br_refill(br);
            uint32_t nlen = br_take(br, 16);
            if ((len ^ 0xffffu) != nlen) { status = DHTS_BLK_ERR_INFLATE; break; }
            for (uint32_t k = 0; k < len && status == 0; k++) {
                br_refill(br);
                uint32_t b = br_take(br, 8);
                EMIT_LIT(b);
            }
            if (br.pos * 8 - br.cnt > payload_bits) status = DHTS_BLK_ERR_INFLATE;
            continue;
        }
        if (type == 3) { status = DHTS_BLK_ERR_INFLATE; break; }

        // ---- code lengths -> canonical tables, in TWO passes over the code-length stream (RFC 1951 3.2.6 / 3.2.7) ----
        // Pass 0 only counts the symbols of every code length; the limits / bases follow from the counts; pass 1 decodes the same
        // bits again and drops every symbol into its sorted slot.  Nothing per-symbol is kept in between, which is what lets the
        // whole per-stream LDS state fit 420 bytes.
        uint32_t nl, nd;
        uint32_t climit[7], cbase[7];
#pragma unroll
        for (int i = 0; i < 7; i++) { climit[i] = 0; cbase[i] = 0; }
        if (type == 1) {
            nl = 288; nd = 32;                               // 32 five-bit distance codes; 30/31 are rejected on use
        } else {
            br_refill(br);
            nl = br_take(br, 5) + 257; nd = br_take(br, 5) + 1;
            const uint32_t nc = br_take(br, 4) + 4;
            if (nl > 286 || nd > 30) { status = DHTS_BLK_ERR_INFLATE; break; }
            uint32_t cl[19];
#pragma unroll
            for (int i = 0; i < 19; i++) cl[i] = 0;
            {
                // order 16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15 -- static indices keep cl[] in VGPRs
                const int ORD[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
#pragma unroll
                for (int i = 0; i < 19; i++) {
                    if ((i & 7) == 0) br_refill(br);
                    uint32_t v = (uint32_t)i < nc ? br_take(br, 3) : 0;
                    cl[ORD[i]] = v;
                }
            }
            // canonical code for the CL alphabet (max length 7)
            uint32_t ccount[8];
#pragma unroll
            for (int L = 0; L < 8; L++) ccount[L] = 0;
#pragma unroll
            for (int i = 0; i < 19; i++) {
#pragma unroll
                for (int L = 1; L < 8; L++) ccount[L] += (cl[i] == (uint32_t)L) ? 1u : 0u;
            }
            uint32_t coffs[8];
            {
                uint32_t first = 0, offs = 0; int left = 1;
#pragma unroll
                for (int L = 1; L <= 7; L++) {
                    left = (left << 1) - (int)ccount[L];
                    cbase[L - 1] = offs - first;
                    climit[L - 1] = (first + ccount[L]) << (7 - L);
                    coffs[L] = offs;
                    offs += ccount[L];
                    first = (first + ccount[L]) << 1;
                }
                if (left != 0) { status = DHTS_BLK_ERR_INFLATE; break; }   // CL code must be complete
            }
#pragma unroll
            for (int i = 0; i < 19; i++) {
#pragma unroll
                for (int L = 1; L < 8; L++) {
                    if (cl[i] == (uint32_t)L) { clsym[coffs[L] * A_ST + lane] = (uint8_t)i; coffs[L]++; }
                }
            }
        }
#pragma unroll
        for (int L = 0; L < 15; L++) { cntl[L * A_ST + lane] = 0; cntd[L * A_ST + lane] = 0; }
        uint32_t nz_l = 0, nz_d = 0, bsl[16], bsd[16];
        bool has_eob = false;
        const BitR br0 = br;                                 // start of the code-length stream
        const uint32_t total = nl + nd;
#pragma unroll 1
        for (int pass = 0; pass < 2 && status == 0; pass++) {
            if (pass == 1) {
                int left = build_limits(ll, cntl, lane, bsl);
                if (left < 0 || (left > 0 && nz_l != 1) || !has_eob) { status = DHTS_BLK_ERR_INFLATE; break; }
                left = build_limits(dl, cntd, lane, bsd);
                if (left < 0 || (left > 0 && nz_d > 1)) { status = DHTS_BLK_ERR_INFLATE; break; }
                for (uint32_t k = 0; k < hi_rows; k++) lsym_hi[k * A_ST + lane] = 0;
                br = br0;
            }
            uint32_t idx = 0, prev = 0;
            while (idx < total) {
                uint32_t val, rep;
                if (type == 1) {
                    const uint32_t stop = idx < 144u ? 144u : idx < 256u ? 256u : idx < 280u ? 280u : idx < 288u ? 288u : 320u;
                    val = idx < 144u ? 8u : idx < 256u ? 9u : idx < 280u ? 7u : idx < 288u ? 8u : 5u;
                    rep = stop - idx;
                } else {
                    br_refill(br);
                    const uint32_t w7 = __brev((uint32_t)br.buf) >> 25;
                    uint32_t c = 0;
#pragma unroll
                    for (int i = 0; i < 7; i++) c += (w7 >= climit[i]) ? 1u : 0u;
                    if (c >= 7) { status = DHTS_BLK_ERR_INFLATE; break; }
                    const uint32_t CL = c + 1;
                    uint32_t cb = 0;
#pragma unroll
                    for (int i = 0; i < 7; i++) cb = (c == (uint32_t)i) ? cbase[i] : cb;
                    uint32_t ci = cb + (w7 >> (7 - CL)); if (ci > 18) ci = 18;
                    const uint32_t sym = clsym[ci * A_ST + lane];
                    br_take(br, CL);
                    if (sym < 16) { val = sym; rep = 1; }
                    else if (sym == 16) { if (idx == 0) { status = DHTS_BLK_ERR_INFLATE; break; } val = prev; rep = 3 + br_take(br, 2); }
                    else if (sym == 17) { val = 0; rep = 3 + br_take(br, 3); }
                    else { val = 0; rep = 11 + br_take(br, 7); }
                    if (idx + rep > total) { status = DHTS_BLK_ERR_INFLATE; break; }
                    prev = val;
                }
                if (val == 0) { idx += rep; continue; }
                for (uint32_t k = 0; k < rep; k++) {
                    const uint32_t i = idx++;
                    if (i < nl) {
                        const uint32_t o = cntl[(val - 1) * A_ST + lane]; cntl[(val - 1) * A_ST + lane] = (uint16_t)(o + 1);
                        if (pass == 0) { nz_l++; has_eob |= (i == 256u); }
                        else {
                            const uint32_t oc = o > 287u ? 287u : o;
                            if (oc < A_NLO) {
                                lsym_lo[oc * A_ST + lane] = (uint8_t)i;
                                if (i & 256) lsym_hi[(oc >> 5) * A_ST + lane] |= 1u << (oc & 31);
                            } else far_tab[oc - A_NLO] = (uint16_t)i;          // the longest codes: looked up through L2 when they occur
                        }
                    } else {
                        const uint32_t o = cntd[(val - 1) * A_ST + lane]; cntd[(val - 1) * A_ST + lane] = (uint8_t)(o + 1);
                        if (pass == 0) nz_d++;
                        else dsym[(o & 31) * A_ST + lane] = (uint8_t)(i - nl);
                    }
                }
            }
        }
        if (status != 0) break;

#ifdef DHTS_DIAG
        unsigned long long dA_s0 = clock64();
#endif
#ifndef HOSTSIM
        __threadfence();                                      // the far-table stores are in L2 before the loop may read them
#endif
        // ---- symbol loop ----
        // ONE Huffman symbol per lane per iteration, from whichever alphabet the lane expects (mode 0 = literal/length,
        // 1 = distance): a wave always holds lanes in both states, so separate literal and distance paths would both execute every
        // iteration.  Code length = 16 - #{i : w < limit[i]}; the limits of BOTH alphabets sit in the two halves of 15 VGPRs and are
        // compared with packed 16-bit saturating arithmetic (no VCC/SGPR round trips).  Malformed-stream tests only accumulate
        // into `bad` (indices are clamped), so the body has three short divergent regions: literal store, length, distance + token.
        // Input: per-lane 64-byte LDS ring of four 16-byte granules, topped up one granule per period of 4 symbols (<= 28 bits per
        // symbol, so a lane moves <= 14 bytes per period): every compressed byte is requested from memory once.
        {
            uint32_t *winA = (uint32_t *)(smem + off_win);
            const uint8_t *sp = br.p;
            // Input ring: granule G = stream bytes [16G, 16G+16) lives in window words 4(G&3)..4(G&3)+3.  Primed with four granules;
            // afterwards a lane fetches ONE granule (16 bytes) per period, and only while it has less than 40 bytes ahead of its read
            // position: a period consumes <= 14 bytes and looks <= 8 bytes ahead, so >= 24 bytes ahead at the start of every period
            // is enough, and a granule is overwritten only when its bytes lie behind the read position (16*G - pos < 40 <= 45).
            uint32_t gnext = (br.pos >> 4) + 4u, gpend = 0; bool fpend = false;
            uint4 R0;
            {
                uint4 q0, q1, q2, q3; const uint8_t *g0p = sp + ((br.pos >> 4) << 4);
                __builtin_memcpy(&q0, g0p, 16); __builtin_memcpy(&q1, g0p + 16, 16); __builtin_memcpy(&q2, g0p + 32, 16); __builtin_memcpy(&q3, g0p + 48, 16);
                const uint32_t s0 = ((br.pos >> 4) & 3u) * 4u, s1 = (s0 + 4u) & 15u, s2 = (s0 + 8u) & 15u, s3 = (s0 + 12u) & 15u;
                winA[(s0 + 0) * A_ST + lane] = q0.x; winA[(s0 + 1) * A_ST + lane] = q0.y; winA[(s0 + 2) * A_ST + lane] = q0.z; winA[(s0 + 3) * A_ST + lane] = q0.w;
                winA[(s1 + 0) * A_ST + lane] = q1.x; winA[(s1 + 1) * A_ST + lane] = q1.y; winA[(s1 + 2) * A_ST + lane] = q1.z; winA[(s1 + 3) * A_ST + lane] = q1.w;
                winA[(s2 + 0) * A_ST + lane] = q2.x; winA[(s2 + 1) * A_ST + lane] = q2.y; winA[(s2 + 2) * A_ST + lane] = q2.z; winA[(s2 + 3) * A_ST + lane] = q2.w;
                winA[(s3 + 0) * A_ST + lane] = q3.x; winA[(s3 + 1) * A_ST + lane] = q3.y; winA[(s3 + 2) * A_ST + lane] = q3.z; winA[(s3 + 3) * A_ST + lane] = q3.w;
            }
            // P[i]: limits of code length i+1 (literal/length | distance << 16).  DB[i]: telescoped deltas of the canonical
            // base table, so that sum_i [w < limit_i] * DB[i] = base[L] (mod 2^12) without an LDS lookup.
            uint32_t P[15], DB[15];
#pragma unroll
            for (int i = 0; i < 15; i++) {
                P[i] = ll.v[i] | (dl.v[i] << 16);
                // 12 bits of base delta above a 4-bit "1": one multiply-add per limit accumulates base (mod 4096) AND the count
                const uint32_t dl_ = (i < 14 ? (bsl[i + 1] - bsl[i + 2]) : bsl[15]) & 0xfffu;
                const uint32_t dd_ = (i < 14 ? (bsd[i + 1] - bsd[i + 2]) : bsd[15]) & 0xfffu;
                DB[i] = ((dl_ << 4) | 1u) | (((dd_ << 4) | 1u) << 16);
            }
            bool live = true, pending = false; uint32_t mode = 0, want = 0, bad = 0, nw = 0, pend_sym = 0;
            nw = winA[((br.pos >> 2) & 15u) * A_ST + lane];                   // next unread word, always one ahead
            while (__ballot(live) != 0ull) {
                // park the granule fetched during the previous period, then fetch the next one if this lane is running low
                if (fpend) {
                    const uint32_t sl = (gpend & 3u) * 4u;
                    winA[(sl + 0) * A_ST + lane] = R0.x; winA[(sl + 1) * A_ST + lane] = R0.y; winA[(sl + 2) * A_ST + lane] = R0.z; winA[(sl + 3) * A_ST + lane] = R0.w;
                    fpend = false;
                }
                if (live && 16u * gnext - br.pos < 40u) { __builtin_memcpy(&R0, sp + 16u * gnext, 16); gpend = gnext; gnext++; fpend = true; }
#pragma unroll 1
                for (int sub = 0; sub < 4; sub++) {
#ifdef DHTS_DIAG
                    dA_it++;
#endif
                    if (live) {
                        if (br.cnt <= 32) {
                            br.buf |= (uint64_t)nw << br.cnt; br.pos += 4; br.cnt += 32;
                            nw = winA[((br.pos >> 2) & 15u) * A_ST + lane];                    // consumed one refill later: latency hidden
                        }
                        const uint32_t w = __brev((uint32_t)br.buf) >> 17;
                        const uint32_t ww = w | (w << 16);
                        uint32_t m[15];
#pragma unroll
                        for (int i = 0; i < 15; i++) m[i] = pk_subsat_u16(P[i], ww);
#pragma unroll
                        for (int i = 0; i < 15; i++) m[i] = pk_min_u16(m[i], 0x00010001u);         // [w < limit_i] per half
                        uint32_t b0 = 0, b1 = 0, b2 = 0;
#pragma unroll
                        for (int i = 0; i < 15; i += 3) { b0 = pk_mad_u16(m[i], DB[i], b0); b1 = pk_mad_u16(m[i + 1], DB[i + 1], b1); b2 = pk_mad_u16(m[i + 2], DB[i + 2], b2); }
                        const uint32_t accB = pk_add_u16(pk_add_u16(b0, b1), b2);
                        const uint32_t accM = mode ? (accB >> 16) : (accB & 0xffffu);
                        const uint32_t clt = accM & 15u;                                     // #{limits > w}
                        // a lane whose previous symbol lives in the far table (sorted index >= A_NLO: the longest literal/length
                        // codes) spent that iteration fetching it: its code bits are already consumed, this iteration only handles it
                        const bool pend = FAR && pending;
                        bad |= (!pend && clt == 0) ? 1u : 0u;
                        uint32_t L = 16u - clt; L = L > 15u ? 15u : L;
                        uint32_t o = ((accM >> 4) + (w >> (15u - L))) & 0xfffu;
                        const uint32_t omax = mode ? 31u : 287u;
                        bad |= (!pend && o > omax) ? 1u : 0u;
                        o = o > omax ? omax : o;
                        const bool far = FAR && !pend && !mode && o >= A_NLO;
                        const uint32_t ol = (FAR && !mode && o >= A_NLO) ? A_NLO - 1u : o;      // (LDS index of a far / pending lane is a dummy)
                        const uint32_t sb = smem[(mode ? off_dsym : 0u) + ol * A_ST + lane];
                        const uint32_t hw = lsym_hi[(ol >> 5) * A_ST + lane];              // both reads in flight together
                        uint32_t sym = sb | (((hw >> (ol & 31u)) << 8) & (mode ? 0u : 0x100u));
                        if (pend) { sym = pend_sym; L = 0; pending = false; }
                        if (far) {
#ifdef HOSTSIM
                            pend_sym = far_tab[o - A_NLO];
#else
                            pend_sym = __hip_atomic_load(far_tab + (o - A_NLO), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
                            pending = true;
                        }
                        br.buf = shr64_small(br.buf, L); br.cnt -= L;
                        if (far) {
                            // symbol handled next iteration
                        } else if (!mode && sym < 256u) {
                            // literal
                            bad |= (outpos >= 65536u) ? 1u : 0u;
                            PUSH_LIT(sym); outpos++;
                            if (++run == DHTS_TOK_PURE) { PUSH_TOK(DHTS_TOK_PURE << 23); run = 0; }
                        } else if (!mode && sym == 256u) {
                            live = false;
                        } else {
                            // length (mode 0) or distance (mode 1): value = base(j) + extra bits, same arithmetic with k = 2 / 1
                            const uint32_t j = mode ? sym : sym - 257u;
                            bad |= (j >= (mode ? 30u : 29u)) ? 1u : 0u;
                            const uint32_t k = mode ? 1u : 2u;
                            const int32_t e = (int32_t)(j >> k) - 1;
                            uint32_t eb = e < 0 ? 0u : (uint32_t)e;
                            eb = eb > 13u ? 13u : eb;
                            uint32_t val = e < 0 ? j : ((((1u << k) | (j & ((1u << k) - 1u))) << eb));
                            val += mode ? 1u : 3u;
                            if (!mode && j == 28u) { val = 258u; eb = 0; }
                            val += (uint32_t)br.buf & ((1u << eb) - 1u);
                            br.buf = shr64_small(br.buf, eb); br.cnt -= eb;
                            if (!mode) { want = val; mode = 1; }
                            else {
                                bad |= (val > outpos || outpos + want > 65536u) ? 1u : 0u;
                                if (!bad) { PUSH_TOK((run << 23) | ((want - 3u) << 15) | (val - 1u)); run = 0; outpos += want; }
                                mode = 0;
                            }
                        }
                        bad |= (br.pos * 8u - br.cnt > payload_bits + 64u) ? 1u : 0u;   // ran off the payload
                        if (bad) { status = DHTS_BLK_ERR_INFLATE; live = false; }
                    }
                }
            }
            // back to the plain reader for the next block header: re-prime its two look-ahead words
            br.w0 = ld32_guard(br.p, br.pos, br.lim); br.w1 = ld32_guard(br.p, br.pos + 4, br.lim);
        }
#ifdef DHTS_DIAG
        dA_sym += clock64() - dA_s0;
#endif
        if (status == 0 && br.pos * 8 - br.cnt > payload_bits) status = DHTS_BLK_ERR_INFLATE;
    }
#ifdef DHTS_DIAG
    if (lane == 0) { atomicAdd(&g_diagA[0], dA_sym); atomicAdd(&g_diagA[1], dA_it); atomicAdd(&g_diagA[3], clock64() - dA_t0); atomicAdd(&g_diagA[4], 1ull); }
#endif
    // (a block that failed is never read by phase B; its open group could lie one byte past the 64 KiB literal slot)
    if (status == 0 && (nlit & 15u)) {                  // the open 16-byte group: complete dwords, then the partial one
        const uint32_t cq = (nlit >> 2) & 3u;
        A_ST16(lit + (nlit & ~15u), cq == 0u ? litbuf : lq0, cq == 1u ? litbuf : lq1, cq == 2u ? litbuf : lq2, litbuf);
    }
    if (status == 0 && (ntok & 3u)) A_ST16(tok + (ntok & ~3u), tq0, tq1, tq2, 0u);
    InflateMeta m; m.ntok = ntok; m.nlit = nlit; m.outlen = outpos; m.status = status;
    meta[s] = m;
#undef EMIT_LIT
#undef PUSH_LIT
#undef PUSH_TOK
#undef A_ST16
}

extern "C" __global__ void __launch_bounds__(64)
bgzf_huff_decode(const uint8_t *__restrict__ comp, BgzfTable tab, int64_t blk0, int32_t nblk,
                 uint8_t *__restrict__ lit_all, uint32_t *__restrict__ tok_all, InflateMeta *__restrict__ meta, uint32_t nlo) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    if (nlo == A_NLO_ALL) huff_decode_body<false>(smem, comp, tab, blk0, nblk, lit_all, tok_all, meta);
    else huff_decode_body<true>(smem, comp, tab, blk0, nblk, lit_all, tok_all, meta);
}

// ------------------------------------------------------------------------------------
// phase B
// ------------------------------------------------------------------------------------
// wave64 inclusive scans on the DPP network (row_shr 1/2/4/8, row_bcast15 -> rows 1,3, row_bcast31 -> rows 2,3):
// six VALU-rate steps instead of six ds_bpermute round trips.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ uint32_t dpp0(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWMASK, 0xf, false); }
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
    (void)lane;
    v += dpp0<0x111, 0xf>(v); v += dpp0<0x112, 0xf>(v); v += dpp0<0x114, 0xf>(v); v += dpp0<0x118, 0xf>(v);
    v += dpp0<0x142, 0xa>(v); v += dpp0<0x143, 0xc>(v);
    return v;
}
__device__ __forceinline__ uint32_t wave_incl_scan_or(uint32_t v) {
    v |= dpp0<0x111, 0xf>(v); v |= dpp0<0x112, 0xf>(v); v |= dpp0<0x114, 0xf>(v); v |= dpp0<0x118, 0xf>(v);
    v |= dpp0<0x142, 0xa>(v); v |= dpp0<0x143, 0xc>(v);
    return v;
}
__device__ __forceinline__ uint32_t wave_incl_scan_max(uint32_t v) {      // (lanes without a source read 0: values are unsigned)
    v = max(v, dpp0<0x111, 0xf>(v)); v = max(v, dpp0<0x112, 0xf>(v)); v = max(v, dpp0<0x114, 0xf>(v)); v = max(v, dpp0<0x118, 0xf>(v));
    v = max(v, dpp0<0x142, 0xa>(v)); v = max(v, dpp0<0x143, 0xc>(v));
    return v;
}
__device__ __forceinline__ uint32_t wave_shr1(uint32_t v) { return dpp0<0x138, 0xf>(v); }   // lane i <- lane i-1, lane 0 <- 0
#define RDLANE(v, i) ((uint32_t)__builtin_amdgcn_readlane((int)(v), (int)(i)))

// multiply two polynomials mod the reflected CRC-32 polynomial (bit 31 = x^0)
__device__ __forceinline__ uint32_t crc_mulmod(uint32_t a, uint32_t b) {
    uint32_t p = 0;
#pragma unroll 8
    for (int i = 0; i < 32; i++) {
        p ^= (b & 0x80000000u) ? a : 0u;
        a = (a >> 1) ^ ((a & 1u) ? 0xEDB88320u : 0u);
        b <<= 1;
    }
    return p;
}

// Unaligned 2/4/8-byte LDS accesses are legal on gfx950 in the HSA alignment mode (verified on hardware:
// tools/dbg/lds_unaligned.hip); the build passes +unaligned-ds-access so these memcpys become single DS ops.
__device__ __forceinline__ uint64_t lds_ld64(const uint8_t *p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
__device__ __forceinline__ void lds_st_n(uint8_t *p, uint64_t v, uint32_t n) {   // stores exactly min(n, 8) bytes
    if (n >= 8) { __builtin_memcpy(p, &v, 8); return; }
    if (n & 4) { uint32_t x = (uint32_t)v; __builtin_memcpy(p, &x, 4); p += 4; v >>= 32; }
    if (n & 2) { uint16_t x = (uint16_t)v; __builtin_memcpy(p, &x, 2); p += 2; v >>= 16; }
    if (n & 1) *p = (uint8_t)v;
}

// Window ring: a power-of-two ring of 2^B_RING_LOG2 bytes, flushed progressively in half-ring chunks: it holds the unflushed tail
// (< half a ring + one batch span) plus the most recent history.  A batch places its literals before its matches, so it may run at
// most BR_SPAN bytes ahead of the oldest byte one of its own matches can still need.  Matches whose source is older than the ring
// read the bytes back from the block's own output in HBM (such sources always lie below `flushed`, see the static_assert).
// 4 KiB ring: 10,240 B of LDS per block, i.e. sixteen workgroups per CU, so every SIMD holds four waves that fill each other's
// issue slots (measured per 16,384-block launch: 6.1 ms with a full 34.5 KB DEFLATE window in LDS, 4.45 ms with a 16 KiB ring,
// 3.5 ms with 8 KiB; with the far loads issued together 3.17 ms at 8 KiB and 2.48 ms at 4 KiB).
#ifndef B_RING_LOG2
#define B_RING_LOG2 12
#endif
#define BR_R (1u << B_RING_LOG2)
#define BR_FLUSH (BR_R / 2u)
#define BR_PIECE (BR_FLUSH / 64u)        /* bytes of a flush chunk CRC'd by one lane */
#ifndef BR_SPAN
#define BR_SPAN 1536u
#endif
static_assert(BR_R >= BR_FLUSH + BR_SPAN + 265u, "far sources must always be flushed");
#define B_WIN 0
#define B_CRCT (BR_R)                    /* u32 [4][256] slice-by-4 tables */
#define B_RING (B_CRCT + 4096)           /* u8 [2048] literal staging ring */
#define B_LDS_BYTES (B_RING + 2048)
#define B_NULLTOK 0xffffffffu
// knock-out experiments (tools/dbg/time_lz.py; the output is wrong on purpose): -DB_EXP_NOCRC, -DB_EXP_NOLIT, -DB_EXP_NOFAR, -DB_EXP_NOROUNDS,
// -DB_EXP_NOREPLAY, -DB_EXP_NOSTORE leave out one part of the kernel each, so that its cost can be read off the launch time
#ifdef B_EXP_NOSTORE
#define B_EXP_STORE(p_, v_) do { asm volatile("" :: "v"((v_).x), "v"((v_).y), "v"((v_).z), "v"((v_).w)); } while (0)
#else
#define B_EXP_STORE(p_, v_) __builtin_memcpy((p_), &(v_), 16)
#endif
#ifdef B_EXP_NOCRC
#define B_EXP_CRC_BYTES 0u
#else
#define B_EXP_CRC_BYTES BR_PIECE
#endif
#ifndef B_SEQ_T
#define B_SEQ_T 6u                        /* pending matches at or below which the rounds of a batch give way to the in-order replay */
#endif
#ifdef DHTS_DIAG
__device__ unsigned long long g_diag[8];   // batches, rounds, easy, hard, lit_iters, long_lit
#define DIAG_ADD(i, v) do { dcnt[i] += (unsigned long long)(v); } while (0)
#define DIAG_T(var) unsigned long long var = clock64()
__device__ unsigned long long g_diagt[8];
#define DIAG_TADD(i, a, b) do { dacc[i] += (b) - (a); } while (0)
#define DIAG_DECL unsigned long long dacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dcnt[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define DIAG_FLUSH do { if (lane == 0) for (int q_ = 0; q_ < 8; q_++) { atomicAdd(&g_diagt[q_], dacc[q_]); atomicAdd(&g_diag[q_], dcnt[q_]); } } while (0)
#else
#define DIAG_DECL do {} while (0)
#define DIAG_FLUSH do {} while (0)
#define DIAG_T(var) do {} while (0)
#define DIAG_TADD(i, a, b) do {} while (0)
#define DIAG_ADD(i, v) do {} while (0)
#endif


__device__ __forceinline__ uint32_t ridx(uint32_t p) { return p & (BR_R - 1u); }
__device__ __forceinline__ uint64_t win_ld64(const uint8_t *win, uint32_t p) {
    const uint32_t i = ridx(p);
    if (i + 8 <= BR_R) return lds_ld64(win + i);
    uint64_t v = 0;
    for (uint32_t k = 0; k < 8; k++) v |= (uint64_t)win[ridx(p + k)] << (8 * k);
    return v;
}
__device__ __forceinline__ void win_st_n(uint8_t *win, uint32_t p, uint64_t v, uint32_t n) {
    const uint32_t i = ridx(p), m = n < 8 ? n : 8;
    if (i + m <= BR_R) lds_st_n(win + i, v, n);
    else for (uint32_t k = 0; k < m; k++) win[ridx(p + k)] = (uint8_t)(v >> (8 * k));
}
// x^(8*nbytes) mod P (reflected)
__device__ __forceinline__ uint32_t crc_xpow8(uint32_t nbytes) {
    uint32_t pw = 0x80000000u, sq = 0x00800000u;
    while (nbytes) { if (nbytes & 1) pw = crc_mulmod(pw, sq); sq = crc_mulmod(sq, sq); nbytes >>= 1; }
    return pw;
}

// Block-independent CRC constants, computed once per context by crc_const_init:
//   [0,64)  K_lane = x^(8*BR_PIECE*(63-lane)),  [64] x^(8*BR_FLUSH),  [65,82) x^(8*2^k) for k < 17,  [96,1120) slice-by-4 tables,
//   [1120,1248) products of every nibble position with x^(8*(BR_FLUSH-BR_PIECE))
#define CRCC_K 0
#define CRCC_XF 64
#define CRCC_XP2 65
#define CRCC_TAB 96
#define CRCC_MK (96 + 1024)              /* u32 [8][16]: (nibble << 4j) * x^(8*(BR_FLUSH-BR_PIECE)): a lane's CRC state carried over the other lanes' pieces */
__device__ uint32_t g_crcc[96 + 1024 + 128];
extern "C" __global__ void __launch_bounds__(64) crc_const_init() {
    const uint32_t lane = threadIdx.x;
    g_crcc[CRCC_K + lane] = crc_xpow8(BR_PIECE * (63u - lane));
    if (lane == 0) g_crcc[CRCC_XF] = crc_xpow8(BR_FLUSH);
    if (lane < 17) g_crcc[CRCC_XP2 + lane] = crc_xpow8(1u << lane);
    {
        const uint32_t kadv = crc_xpow8(BR_FLUSH - BR_PIECE);
        for (uint32_t e = lane; e < 128u; e += 64u) g_crcc[CRCC_MK + e] = crc_mulmod((e & 15u) << (4u * (e >> 4)), kadv);
    }
    for (uint32_t k = lane; k < 256; k += 64) {
        uint32_t t[4], c = k;
        for (int j = 0; j < 8; j++) c = (c & 1u) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
        t[0] = c;
        for (int q = 1; q < 4; q++) {            // t[q][k] = CRC of byte k followed by q zero bytes
            uint32_t z = t[q - 1] & 0xffu, zc = z;
            for (int j = 0; j < 8; j++) zc = (zc & 1u) ? (0xEDB88320u ^ (zc >> 1)) : (zc >> 1);
            t[q] = zc ^ (t[q - 1] >> 8);
        }
        for (int q = 0; q < 4; q++) g_crcc[CRCC_TAB + 256 * q + k] = t[q];
    }
}
// x^(8*nbytes) mod P from the table of squares: one multiplication per set bit of nbytes (< 2^17)
__device__ __forceinline__ uint32_t crc_xpow8_tab(uint32_t nbytes) {
    uint32_t pw = 0x80000000u;
    for (uint32_t k = 0; nbytes; k++, nbytes >>= 1) if (nbytes & 1u) pw = crc_mulmod(pw, g_crcc[CRCC_XP2 + k]);
    return pw;
}

// One BGZF block: the tokens / literals `tok` / `lit` described by `m` become the block's bytes at out + (uoff[bi] - out_base); the CRC-32 and
// ISIZE of the trailer are checked (blk_status[bi]).  Called by bgzf_lz_resolve (one workgroup per block, scratch slots written by a
// phase-A launch) and by bgzf_inflate_fused (bgzf_huff_wave.hip: the wave that decoded the block resolves it right away).
// The waves of a workgroup never exchange data (each resolves a block of its own in its own window; the CRC tables are read-only), and the
// LDS operations of ONE wave execute in program order: what a phase boundary needs is that the compiler keeps that order.
#define LZ_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
// slice-by-4 tables: 4 KiB copied from the per-device constants (by every wave of the workgroup: the same words)
__device__ __forceinline__ void lz_load_crc_tables(uint32_t *crct) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int q = 0; q < 4; q++) *(uint4 *)(crct + q * 256 + lane * 4) = *(const uint4 *)(g_crcc + CRCC_TAB + q * 256 + lane * 4);
}
__device__ __forceinline__ void lz_block(uint8_t *win, uint32_t *crct, uint8_t *ring, const uint8_t *__restrict__ comp, BgzfTable tab, int64_t bi, const InflateMeta m,
                                         const uint8_t *__restrict__ lit, const uint32_t *__restrict__ tok,
                                         uint8_t *__restrict__ out, uint64_t out_base, int32_t *__restrict__ blk_status) {
    const int lane = threadIdx.x & 63;
    const uint32_t isize = tab.isize[bi];
    const uint32_t clen = tab.clen[bi];
    DIAG_DECL;
    DIAG_T(t_begin);

    DIAG_T(t_tab);
    DIAG_TADD(0, t_begin, t_tab);

    int st = m.status;
    if (st == 0 && m.outlen != isize) st = DHTS_BLK_ERR_ISIZE;   // htslib does not test ISIZE; we flag it
    if (st != 0) { if (lane == 0) blk_status[bi] = st; return; }

    uint8_t *dstp = out + (tab.uoff[bi] - out_base);
    uint32_t outpos = 0, litpos = 0, flushed = 0, crc_acc = 0;
    // constants of the flush-chunk CRC: the lane's rows of the advance table (entries lane and 64 + lane)
    const uint32_t mk_lo = g_crcc[CRCC_MK + lane], mk_hi = g_crcc[CRCC_MK + 64 + lane];

    // Flush the chunk [flushed, flushed + BR_FLUSH): store it with 1 KiB coalesced wave stores and run its bytes through the CRC.
    // Every lane carries ONE CRC state across all flushes of the block (crc_acc): its BR_PIECE-byte piece of this chunk continues the
    // state it left behind its piece of the previous chunk, advanced over the BR_FLUSH - BR_PIECE bytes of the other lanes in between
    // (a multiplication by the constant x^(8*(BR_FLUSH-BR_PIECE)): linear, so eight nibble look-ups in the 128-entry table that the
    // wave holds in two registers, fetched with ds_bpermute).  The 64 states are combined once, behind the block's last full chunk.
    // BR_R is a multiple of BR_PIECE, so neither a lane's CRC piece nor a 16-byte store unit wraps in the ring.
#define CRC_ADVANCE(r_) ({                                                                                                  \
        const uint32_t a_ = (r_); uint32_t m_ = 0;                                                                          \
        _Pragma("unroll") for (uint32_t j_ = 0; j_ < 4; j_++) {                                                            \
            m_ ^= (uint32_t)__builtin_amdgcn_ds_bpermute((int)((((a_ >> (4u * j_)) & 15u) + 16u * j_) << 2), (int)mk_lo);      \
            m_ ^= (uint32_t)__builtin_amdgcn_ds_bpermute((int)((((a_ >> (16u + 4u * j_)) & 15u) + 16u * j_) << 2), (int)mk_hi); \
        }                                                                                                                   \
        m_; })
#define FLUSH_CHUNK() do {                                                                                              \
        const uint32_t pi_ = ridx(flushed + (uint32_t)lane * BR_PIECE);                                                 \
        uint32_t c_ = CRC_ADVANCE(crc_acc);                                                                             \
        if (flushed == 0 && lane == 0) c_ = 0xffffffffu;                         /* the stream's first byte */            \
        for (uint32_t q_ = 0; q_ < B_EXP_CRC_BYTES; q_ += 4) {                                                               \
            uint32_t v_ = *(const uint32_t *)(win + pi_ + q_) ^ c_;                                                     \
            c_ = crct[768 + (v_ & 0xff)] ^ crct[512 + ((v_ >> 8) & 0xff)] ^ crct[256 + ((v_ >> 16) & 0xff)] ^ crct[v_ >> 24]; \
        }                                                                                                               \
        crc_acc = c_;                                                                                                   \
        for (uint32_t k_ = 0; k_ < BR_FLUSH; k_ += 1024) {                                                              \
            const uint32_t p_ = flushed + k_ + (uint32_t)lane * 16u;                                                    \
            uint4 v4_ = *(const uint4 *)(win + ridx(p_));                                                               \
            B_EXP_STORE(dstp + p_, v4_);                                                                                \
        }                                                                                                               \
        __builtin_amdgcn_s_waitcnt(0x0f70);                   /* vmcnt(0): the chunk is in L2 before any far-match read-back */ \
        flushed += BR_FLUSH;                                                                                            \
    } while (0)

    // literal staging ring: literal bytes [.., stage_hi) are in LDS at (abs & 2047); filled 1 KiB at a time, one piece in flight
    uint32_t stage_hi = 0; bool pend = false; uint4 pv = make_uint4(0, 0, 0, 0);
#define STAGE_ISSUE() do { if (!pend && stage_hi < m.nlit && (int32_t)(stage_hi - litpos) <= 1024) { __builtin_memcpy(&pv, lit + stage_hi + lane * 16, 16); pend = true; } } while (0)
#define STAGE_COMMIT() do { if (pend) { *(uint4 *)(ring + ((stage_hi + lane * 16) & 2047u)) = pv; stage_hi += 1024u; pend = false; } } while (0)
    STAGE_ISSUE(); STAGE_COMMIT(); STAGE_ISSUE(); STAGE_COMMIT();
    uint32_t tnext = (lane < (int)m.ntok) ? tok[lane] : B_NULLTOK;

    for (uint32_t t0 = 0; t0 < m.ntok; t0 += 64) {
        const uint32_t t = tnext;
        tnext = (t0 + 64 + lane < m.ntok) ? tok[t0 + 64 + lane] : B_NULLTOK;     // in flight while this batch is resolved
        STAGE_ISSUE();
        uint32_t lrun = (t == B_NULLTOK) ? 0 : (t >> 23);
        const bool pure = (lrun == DHTS_TOK_PURE) || (t == B_NULLTOK);
        const uint32_t mlen = pure ? 0 : ((t >> 15) & 255u) + 3;
        const uint32_t mdist = (t & 0x7fffu) + 1;
        // one scan for both sums: a token advances the output by at most 511 + 258 bytes, 64 of them by less than 2^16
        const uint32_t both_i = wave_incl_scan(((lrun + mlen) << 16) | lrun, lane);
        const uint32_t adv_i = both_i >> 16, lit_i = both_i & 0xffffu;
        const uint32_t both_t = RDLANE(both_i, 63), tot_adv = both_t >> 16, tot_lit = both_t & 0xffffu;
        const uint32_t dst = outpos + adv_i - (lrun + mlen);       // where this token's literals start
        const uint32_t lsrc = litpos + lit_i - lrun;               // absolute index of its first literal byte
        // a distance may not reach in front of the block's first byte (RFC 1951 3.2.3): the wave kernel of phase A cannot make this
        // test (its lanes do not know their output position while they decode), so it is made here, before any source is touched
        if (__ballot(mlen > 0 && mdist > dst + lrun) != 0ull) { if (lane == 0) blk_status[bi] = DHTS_BLK_ERR_INFLATE; return; }
        DIAG_T(t_a);
        if (tot_adv <= BR_SPAN && tot_lit <= 1024u) {
            DIAG_ADD(0, 1);
            // ---- far matches: request their bytes now, the literal pass below hides the read-back latency ----
            // sources below rlo are overwritten in the ring by this batch's own output: they come from HBM (always < flushed;
            // the block's own output is always readable 40 bytes past a far source: ms + 40 < bend)
            const uint32_t md = dst + lrun, ms = md - mdist, mspan = mlen < mdist ? mlen : mdist;
            const uint32_t bend = outpos + tot_adv, rlo = bend > BR_R ? bend - BR_R : 0u;
#ifdef B_EXP_NOFAR
            const bool farm = false;
#else
            const bool farm = mlen > 0 && ms < rlo;
#endif
            uint64_t fv0 = 0, fv1 = 0, fv2 = 0, fv3 = 0;
            if (farm) {
                const uint8_t *g = dstp + ms;
                uint64_t a[2]; __builtin_memcpy(a, g, 16); fv0 = a[0]; fv1 = a[1];
                if (mlen > 16u) { uint64_t b[2]; __builtin_memcpy(b, g + 16, 16); fv2 = b[0]; fv3 = b[1]; }
            }
            // ---- literals ----
            while (litpos + tot_lit > stage_hi && stage_hi < m.nlit) { STAGE_ISSUE(); STAGE_COMMIT(); }
            LZ_SYNC();
            // the first 8 bytes of every run lane-parallel: one (usually unaligned) 64-bit ring read, then an exact-length store; the few
            // runs that are longer (3 % on BAM data) are finished one at a time by the whole wave, a byte per lane -- a second and third
            // lane-parallel step would cost the whole wave a full iteration each for one or two lanes' bytes
            DIAG_ADD(4, 1);
#ifdef B_EXP_NOLIT
            if (false) {
#else
            if (lrun) {
#endif
                const uint32_t ri = lsrc & 2047u;
                uint64_t v;
                if (ri <= 2040u) v = lds_ld64(ring + ri);
                else { v = 0; for (uint32_t k = 0; k < 8; k++) v |= (uint64_t)ring[(ri + k) & 2047u] << (8 * k); }
                win_st_n(win, dst, v, lrun);
            }
#ifdef B_EXP_NOLIT
            uint64_t LL = 0;
#else
            uint64_t LL = __ballot(lrun > 8u);
#endif
            while (LL) {
                const int i = __ffsll((unsigned long long)LL) - 1; LL &= LL - 1;
                const uint32_t d0 = RDLANE(dst, i) + 8u, s0 = RDLANE(lsrc, i) + 8u, n0 = RDLANE(lrun, i) - 8u;
                for (uint32_t k = lane; k < n0; k += 64) win[ridx(d0 + k)] = ring[(s0 + k) & 2047u];
            }
            LZ_SYNC();
            DIAG_T(t_b);
            DIAG_TADD(1, t_a, t_b);
            // ---- matches ----
            // A pending match is ready when its source overlaps no destination of an EARLIER pending match (all literals of
            // the batch are already placed).  Destinations/sources are rasterised into 64 cells covering the batch's output
            // span; an exclusive prefix-OR over lanes gives each lane the cells still owed by earlier matches.  The earliest
            // pending match always sees an empty set, so every round makes progress; cell granularity only delays.
            uint32_t sh = 4; while ((tot_adv >> sh) > 63u) sh++;
            uint64_t dmask = 0, smask = 0;
            if (farm) {
                const uint32_t rdm = ridx(md);
                if (mlen <= 32u && rdm + 40u <= BR_R) {
                    // the usual case: whole 8-byte pieces out of the registers, then the exact tail
                    uint8_t *dp = win + rdm;
                    const uint32_t full = mlen >> 3, tail = mlen & 7u;
                    if (full > 0u) __builtin_memcpy(dp, &fv0, 8);
                    if (full > 1u) __builtin_memcpy(dp + 8, &fv1, 8);
                    if (full > 2u) __builtin_memcpy(dp + 16, &fv2, 8);
                    if (full > 3u) __builtin_memcpy(dp + 24, &fv3, 8);
                    if (tail) { const uint64_t vt = full == 0u ? fv0 : full == 1u ? fv1 : full == 2u ? fv2 : fv3; lds_st_n(dp + 8u * full, vt, tail); }
                } else {
                    const uint8_t *g = dstp + ms;
                    win_st_n(win, md, fv0, mlen);
                    if (mlen > 8u) win_st_n(win, md + 8, fv1, mlen - 8u);
                    if (mlen > 16u) win_st_n(win, md + 16, fv2, mlen - 16u);
                    if (mlen > 24u) win_st_n(win, md + 24, fv3, mlen - 24u);
                    for (uint32_t c = 32; c < mlen; c += 8) { uint64_t v; __builtin_memcpy(&v, g + c, 8); win_st_n(win, md + c, v, mlen - c); }
                }
            }
            if (mlen > 0) {
                const uint32_t lo = (md - outpos) >> sh, hi = (md - outpos + mlen - 1) >> sh;
                dmask = ((~0ull) >> (63u - hi)) & ((~0ull) << lo);
                if (!farm && ms + mspan > outpos) {
                    const uint32_t slo = (ms > outpos ? ms - outpos : 0u) >> sh, shi = (ms + mspan - 1 - outpos) >> sh;
                    smask = ((~0ull) >> (63u - shi)) & ((~0ull) << slo);
                }
            }
#ifdef B_EXP_NOROUNDS
            uint64_t P = 0;
#else
            uint64_t P = __ballot(mlen > 0 && !farm);
#endif
            DIAG_T(t_b2);
            DIAG_TADD(7, t_b, t_b2);
            LZ_SYNC();                                   // far copies are in the ring before anybody reads them
            DIAG_ADD(6, __popcll(P)); DIAG_ADD(5, __popcll(__ballot(farm)));
            // Parallel rounds while they pay: on BAM data 84 % of a batch's matches are ready in the first round, 11 % in the second,
            // and the dependency chains of the rest would cost a full round for one or two copies each.  As soon as B_SEQ_T or fewer
            // matches are pending (or a round found nothing it could copy) the rest is replayed one match at a time in stream order
            // -- always ready by construction, no readiness test -- by the whole wave (one byte per lane).
            bool more = P != 0ull;
            while (more) {
                DIAG_ADD(1, 1);
                const bool pending = (P >> lane) & 1ull;
                const uint64_t e = pending ? dmask : 0ull;
                const uint32_t el = wave_shr1(wave_incl_scan_or((uint32_t)e)), eh = wave_shr1(wave_incl_scan_or((uint32_t)(e >> 32)));
                const uint64_t owed = ((uint64_t)eh << 32) | el;                   // exclusive prefix-OR over earlier lanes
                const bool ready = pending && ((smask & owed) == 0ull);
                // lane-parallel, 8 bytes per step with unaligned 64-bit LDS accesses: byte runs (dist 1), non-overlapping copies, and
                // overlapping copies with dist >= 8 (a chunk never reads what it writes; chunks go in order).  Matches whose source or
                // destination would wrap in the ring, and long ones, stay pending for the sequential replay.
                const uint32_t rs = ridx(ms), rd = ridx(md);
                const bool easy = ready && (mlen <= 32u) && (mdist >= 8u || mdist >= mlen || mdist == 1u) && (rs + 40u <= BR_R) && (rd + 40u <= BR_R);
                if (easy) {
                    const uint8_t *sp = win + rs; uint8_t *dp = win + rd;
                    const uint32_t full = mlen >> 3, tail = mlen & 7u;
                    if (mdist == 1u) {
                        const uint64_t rep = (uint64_t)sp[0] * 0x0101010101010101ull;
#pragma unroll
                        for (uint32_t c = 0; c < 4; c++) if (c < full) __builtin_memcpy(dp + 8 * c, &rep, 8);
                        if (tail) lds_st_n(dp + 8 * full, rep, tail);
                    } else {
#pragma unroll
                        for (uint32_t c = 0; c < 4; c++) if (c < full) { const uint64_t v = lds_ld64(sp + 8 * c); __builtin_memcpy(dp + 8 * c, &v, 8); }
                        if (tail) lds_st_n(dp + 8 * full, lds_ld64(sp + 8 * full), tail);
                    }
                }
                const uint64_t E = __ballot(easy);
                DIAG_ADD(2, __popcll(E));
                LZ_SYNC();
                P &= ~E;
                more = E != 0ull && (uint32_t)__popcll(P) > B_SEQ_T;
            }
            DIAG_ADD(3, __popcll(P));
#ifdef B_EXP_NOREPLAY
            P = 0;
#endif
            while (P) {
                const int i = __ffsll((unsigned long long)P) - 1; P &= P - 1;
                const uint32_t d0 = RDLANE(md, i), l0 = RDLANE(mlen, i), di = RDLANE(mdist, i);
                const uint32_t src0 = d0 - di;
                if (l0 <= di) { for (uint32_t k = lane; k < l0; k += 64) win[ridx(d0 + k)] = win[ridx(src0 + k)]; }
                else if (di == 1u) { const uint8_t v = win[ridx(src0)]; for (uint32_t k = lane; k < l0; k += 64) win[ridx(d0 + k)] = v; }
                else { for (uint32_t k = lane; k < l0; k += 64) win[ridx(d0 + k)] = win[ridx(src0 + (k % di))]; }
            }
            LZ_SYNC();
            DIAG_T(t_c);
            DIAG_TADD(2, t_b, t_c);
            outpos += tot_adv; litpos += tot_lit;
            while (outpos - flushed >= BR_FLUSH) FLUSH_CHUNK();
        } else {
            // ---- oversized batch (long literal runs / long matches): strict stream order, one token at a time ----
            DIAG_ADD(7, 1);
            for (int i = 0; i < 64; i++) {
                const uint32_t lr = RDLANE(lrun, i), ml = RDLANE(mlen, i), di = RDLANE(mdist, i);
                for (uint32_t k = lane; k < lr; k += 64) win[ridx(outpos + k)] = lit[litpos + k];
                outpos += lr; litpos += lr;
                LZ_SYNC();
                if (ml) {
                    const uint32_t src0 = outpos - di;
                    const uint32_t thr = outpos + ml > BR_R ? outpos + ml - BR_R : 0u;      // older bytes are read back from HBM
                    if (ml <= di) { for (uint32_t k = lane; k < ml; k += 64) { const uint32_t sx = src0 + k; win[ridx(outpos + k)] = sx < thr ? dstp[sx] : win[ridx(sx)]; } }
                    else { for (uint32_t k = lane; k < ml; k += 64) win[ridx(outpos + k)] = win[ridx(src0 + (k % di))]; }          // overlapping: dist < 258, never far
                    outpos += ml;
                    LZ_SYNC();
                }
                while (outpos - flushed >= BR_FLUSH) FLUSH_CHUNK();
            }
            pend = false; stage_hi = litpos & ~1023u;          // restart the literal ring behind the literals consumed here
        }
        STAGE_COMMIT();
    }
    DIAG_T(t_loop);
    DIAG_TADD(3, t_tab, t_loop);
    // trailing literals
    {
        uint32_t rem = m.nlit - litpos;
        while (rem) {
            const uint32_t n = rem < BR_FLUSH / 2u ? rem : BR_FLUSH / 2u;
            for (uint32_t k = lane; k < n; k += 64) win[ridx(outpos + k)] = lit[litpos + k];
            outpos += n; litpos += n; rem -= n;
            LZ_SYNC();
            while (outpos - flushed >= BR_FLUSH) FLUSH_CHUNK();
        }
    }
    LZ_SYNC();
    if (outpos != m.outlen) { if (lane == 0) blk_status[bi] = DHTS_BLK_ERR_INFLATE; return; }

    DIAG_T(t_crc0);
    // ---- tail [flushed, outlen): CRC per lane + combine, then fold into crc_run ----
    const uint32_t n = m.outlen - flushed;                     // < BR_FLUSH
    const uint32_t chunk = ((n + 63) / 64 + 3) & ~3u;          // multiple of 4; flushed is a multiple of 4, so dwords never wrap
    uint32_t beg = lane * chunk; if (beg > n) beg = n;
    uint32_t end = beg + chunk; if (end > n) end = n;
    // the flushed chunks: a lane's state is followed by the BR_PIECE*(63-lane) bytes of the lanes behind it in the last chunk
    uint32_t crc_run = 0;
    if (flushed != 0) {
        crc_run = crc_mulmod(crc_acc, g_crcc[CRCC_K + lane]);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) crc_run ^= __shfl_xor(crc_run, d, 64);
    }
    uint32_t c = lane == 0 ? (flushed == 0 ? 0xffffffffu : crc_run) : 0u;      // lane 0 continues the running CRC
    uint32_t q = beg;
    for (; q + 4 <= end; q += 4) {
        uint32_t v = *(const uint32_t *)(win + ridx(flushed + q)) ^ c;
        c = crct[768 + (v & 0xff)] ^ crct[512 + ((v >> 8) & 0xff)] ^ crct[256 + ((v >> 16) & 0xff)] ^ crct[v >> 24];
    }
    for (; q < end; q++) c = crct[(c ^ win[ridx(flushed + q)]) & 0xff] ^ (c >> 8);
    c = crc_mulmod(c, crc_xpow8_tab(n - end));
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c ^= __shfl_xor(c, d, 64);
    c ^= 0xffffffffu;
    uint32_t want; __builtin_memcpy(&want, comp + tab.coff[bi] + clen - 8, 4);
    if (c != want) st = DHTS_BLK_ERR_CRC;
    if (lane == 0) blk_status[bi] = st;
    DIAG_T(t_crc1);
    DIAG_TADD(4, t_crc0, t_crc1);
    // a CRC mismatch ends the stream at this block: the host cuts the inflated stream here, the bytes already stored are ignored
    for (uint32_t k = lane * 16; k < n; k += 1024) {
        const uint32_t p = flushed + k;
        if (k + 16 <= n) { uint4 v4 = *(const uint4 *)(win + ridx(p)); __builtin_memcpy(dstp + p, &v4, 16); }
        else for (uint32_t j = k; j < n; j++) dstp[flushed + j] = win[ridx(flushed + j)];
    }
    DIAG_T(t_end);
    DIAG_TADD(5, t_crc1, t_end);
    DIAG_TADD(6, t_begin, t_end);
    DIAG_FLUSH;
}

// B_NW waves per workgroup, one BGZF block each: the waves share the 4 KiB of CRC tables, so a wave costs 6 KiB of LDS instead of 10
// (4 KiB ring window + 2 KiB literal staging) and a CU holds 20 waves instead of 16.  Measured per 92 M-record step: one wave per workgroup
// 175.4 ms, four 174.2 ms, eight (24 waves per CU) 182.3 ms -- the kernel is bound by the SIMDs' instruction issue (VALU and SALU together,
// about 3 cycles per wave-instruction), not by latency, so more resident waves only take issue slots from the record stage beside it.
#ifndef B_NW
#define B_NW 4
#endif
#define B_WAVE_LDS (BR_R + 2048u)
#define B_LDS_BYTES_NW (4096u + B_NW * B_WAVE_LDS)
extern "C" __global__ void __launch_bounds__(64 * B_NW)
bgzf_lz_resolve(const uint8_t *__restrict__ comp, BgzfTable tab, int64_t blk0, int32_t nblk,
                const uint8_t *__restrict__ lit_all, const uint32_t *__restrict__ tok_all,
                const InflateMeta *__restrict__ meta, int64_t scratch_b0, uint8_t *__restrict__ out, uint64_t out_base,
                int32_t *__restrict__ blk_status, const unsigned long long *__restrict__ blk_off) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t *crct = (uint32_t *)smem;
    const uint32_t wave = threadIdx.x >> 6;
    uint8_t *win = smem + 4096u + wave * B_WAVE_LDS, *ring = win + BR_R;
    if (wave == 0) lz_load_crc_tables(crct);
    __syncthreads();                                       // the only workgroup-wide barrier: every wave reaches it
    const int64_t k = (int64_t)blockIdx.x * B_NW + wave;
    if (k >= nblk) return;
    const int64_t bi = blk0 + k;
    const int64_t s = bi - scratch_b0;            // slot in the phase-A scratch (phase A may run ahead over a larger block range)
    const InflateMeta m = meta[s];
    if (blk_off != nullptr) {
        // packed scratch (bgzf_huff_decode_wave): literals at the block's offset of the pool `lit_all`, tokens behind them
        const uint8_t *lp = lit_all + (m.status == 0 ? blk_off[s] : 0ull);
        lz_block(win, crct, ring, comp, tab, bi, m, lp, (const uint32_t *)(lp + ((m.nlit + 15u) & ~15u)), out, out_base, blk_status);
    } else lz_block(win, crct, ring, comp, tab, bi, m, lit_all + (size_t)s * DHTS_LIT_STRIDE, tok_all + (size_t)s * DHTS_TOK_STRIDE, out, out_base, blk_status);
}

// Blocks whose decoded length differs from the ISIZE field the block table was built on (phase A has decoded them, status 0): htslib never
// reads ISIZE (bgzf.c:793-801 checks the CRC only), so such a block is valid there -- the host re-places the blocks from the decoded lengths
// (dhts_bgzf.inc: isize_repair).  cnt[0] = how many.
extern "C" __global__ void __launch_bounds__(256)
bgzf_len_check(const InflateMeta *__restrict__ meta, const uint32_t *__restrict__ isize, int64_t b0, int32_t n, uint32_t *__restrict__ cnt) {
    const int32_t k = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x);
    const bool bad = k < n && meta[k].status == 0 && meta[k].outlen != isize[b0 + k];
    const unsigned long long m = __ballot(bad);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(cnt, (uint32_t)__popcll(m));
}

// First damaged block of a batch, found on the device so that the host need not read the whole status array (and need not wait for it
// before it queues the record stage): res[0] = index (relative to blk0) of the first block whose status is not 0, res[1] = that status,
// res[2] = index of the first block marked DHTS_BLK_ERR_SCRATCH; 0xffffffff = none.  One workgroup.
extern "C" __global__ void __launch_bounds__(1024)
bgzf_first_bad_block(const int32_t *__restrict__ blk_status, int64_t blk0, int32_t nblk, uint32_t *__restrict__ res) {
    __shared__ uint32_t s_bad, s_scr;
    if (threadIdx.x == 0) { s_bad = 0xffffffffu; s_scr = 0xffffffffu; }
    __syncthreads();
    uint32_t bad = 0xffffffffu, scr = 0xffffffffu;
    for (int32_t k = (int32_t)threadIdx.x; k < nblk; k += 1024) {
        const int32_t v = blk_status[blk0 + k];
        if (v != 0 && (uint32_t)k < bad) bad = (uint32_t)k;
        if (v == DHTS_BLK_ERR_SCRATCH && (uint32_t)k < scr) scr = (uint32_t)k;
    }
    if (bad != 0xffffffffu) atomicMin(&s_bad, bad);
    if (scr != 0xffffffffu) atomicMin(&s_scr, scr);
    __syncthreads();
    if (threadIdx.x == 0) { res[0] = s_bad; res[1] = s_bad != 0xffffffffu ? (uint32_t)blk_status[blk0 + s_bad] : 0u; res[2] = s_scr; res[3] = 0u; }
}
