// bgzf_deflate.hip -- the write side of BGZF: raw bytes -> BGZF blocks on the device (src/bgzip.c -> bgzf_write /
// bgzf_compress, htslib bgzf.c:509-620: every 0xff00 input bytes become one gzip member with the BC extra field, CRC-32 and ISIZE).
//
// One wave per BGZF block, three stages in one kernel:
//   1. LZ77 parse.  64 consecutive input positions per step; lane i hashes the 4 bytes at its position into a table of 2^12 buckets of the two
//      last positions in LDS (read the candidates, then insert) and verifies both byte for byte (8 bytes per compare, up to 258); the longer
//      wins.  The parse of the step is a walk over the lanes' step lengths (a wave-uniform loop over v_readlane, one iteration per token) with
//      zlib's one-position lazy evaluation: a match gives way to a literal when the next position holds a longer one (not at level 1).  Tokens (literal, or length + distance) go to a per-block scratch in HBM; their symbols are counted in LDS.
//   2. Code construction.  Code lengths from the counts: len = ceil(log2(total / count)) capped at 15 (a Shannon code: satisfies Kraft by
//      construction), then codes are shortened, most frequent symbol first, until the code is complete (Kraft sum exactly 1: inflate
//      rejects incomplete literal/length codes).  The same for the 19-symbol code-length alphabet (cap 7).  Code lengths are sent with zero
//      runs (symbols 17 / 18), no repeat-previous.  The exact sizes of the dynamic block, the fixed-code block (RFC 1951 3.2.6) and the
//      stored block are known before a bit is written: the smallest wins.
//   3. Bit packing.  64 tokens per step: each lane looks its token's codes up in LDS (<= 48 bits), a wave scan of the bit counts gives the
//      positions, ds_or packs them into a 1 KiB ring in LDS whose complete words leave as one coalesced store per step.
// The CRC-32 of the input is computed by the same wave (64 pieces, slice-by-4, combined with x^(8n) multipliers: the constants of
// bgzf_inflate.hip).  Output: one 65,536-byte slot per block + its size; bgzf_pack_blocks copies the slots to their final offsets.
//
// The bytes differ from zlib's for the same input (any valid DEFLATE stream is a valid answer; the reference's own output depends on the zlib
// / libdeflate it was linked with); what is tested is that every reader gives the input back and that the container is BGZF.
#pragma once
#include "dhts_common.h"

#define DFL_IN 65280u                    /* BGZF_BLOCK_SIZE 0xff00: input bytes per block (bgzf.c:66) */
#define DFL_SLOT 65536u                  /* BGZF_MAX_BLOCK_SIZE: a block, header and trailer included, never exceeds it */
#define DFL_HASH_BITS 13
#define DFL_NONE 0xffffu
#define DFL_RING_WORDS 256u
// LDS: hash table 16 KiB (reused for the code tables after the parse) | ring 1 KiB | CRC tables 4 KiB | counts: 288 + 32 + 20 words
#define DFL_OFF_RING ((1u << DFL_HASH_BITS) * 2u)
#define DFL_OFF_CRCT (DFL_OFF_RING + DFL_RING_WORDS * 4u)
#define DFL_OFF_CNT (DFL_OFF_CRCT + 4096u)
#define DFL_LDS_BYTES (DFL_OFF_CNT + (288u + 32u + 32u) * 4u)

__device__ const uint8_t g_dfl_clord[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};   // order of the code-length code lengths (RFC 1951 3.2.7)
__device__ __forceinline__ uint32_t dfl_rev(uint32_t v, uint32_t nbits) { return nbits ? __brev(v) >> (32u - nbits) : 0u; }
// length 3..258 -> code index 0..28, extra bit count, extra value (RFC 1951 3.2.5)
__device__ __forceinline__ void dfl_len_code(uint32_t len, uint32_t &lc, uint32_t &eb, uint32_t &ev) {
    const uint32_t l = len - 3u;
    if (len == 258u) { lc = 28; eb = 0; ev = 0; }
    else if (l < 8u) { lc = l; eb = 0; ev = 0; }
    else { const uint32_t hb = 31u - (uint32_t)__builtin_clz(l); eb = hb - 2u; lc = (eb << 2) + ((l >> eb) & 3u) + 4u; ev = l & ((1u << eb) - 1u); }
}
__device__ __forceinline__ void dfl_dist_code(uint32_t d /* distance - 1 */, uint32_t &dc, uint32_t &eb, uint32_t &ev) {
    if (d < 4u) { dc = d; eb = 0; ev = 0; }
    else { const uint32_t hb = 31u - (uint32_t)__builtin_clz(d); eb = hb - 1u; dc = (eb << 1) + ((d >> eb) & 1u) + 2u; ev = d & ((1u << eb) - 1u); }
}
__device__ __forceinline__ uint32_t dfl_fixed_len(uint32_t s) { return s < 144u ? 8u : s < 256u ? 9u : s < 280u ? 7u : 8u; }

// Code lengths for n symbols with counts cnt[] (LDS), written to len[] (LDS, bytes); at most maxbits.  Single lane (lane 0) for the serial
// parts; n <= 288.  Symbols with count 0 get length 0.  With fewer than two used symbols a second one is given length 1 (a complete code
// needs two leaves; zlib does the same in build_tree).  Returns nothing; the caller syncs.
__device__ void dfl_build_lengths(const uint32_t *cnt, uint8_t *len, uint32_t n, uint32_t maxbits, int lane, uint32_t *ord /* LDS scratch n words */) {
    // total and per-symbol Shannon length, in parallel
    uint32_t part = 0, used = 0;
    for (uint32_t s = lane; s < n; s += 64) { part += cnt[s]; used += cnt[s] ? 1u : 0u; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { part += __shfl_xor(part, d, 64); used += __shfl_xor(used, d, 64); }
    const uint32_t total = part;
    for (uint32_t s = lane; s < n; s += 64) {
        const uint32_t f = cnt[s];
        uint32_t L = 0;
        if (f) { L = 1; while (L < maxbits && ((uint64_t)f << L) < (uint64_t)total) L++; }
        len[s] = (uint8_t)L;
        // rank by descending count (ties: lower symbol first): the order in which codes are shortened
        uint32_t r = 0;
        for (uint32_t t = 0; t < n; t++) { const uint32_t g = cnt[t]; r += (g > f || (g == f && t < s)) ? 1u : 0u; }
        ord[r] = s;
    }
    __syncthreads();
    if (lane == 0) {
        if (used < 2u) {                                       // one (or no) used symbol: pair it with a dummy so that the code is complete
            uint32_t a = n; for (uint32_t s = 0; s < n; s++) if (cnt[s]) { a = s; break; }
            for (uint32_t s = 0; s < n; s++) len[s] = 0;
            if (a == n) { len[0] = 1; len[1] = 1; } else { len[a] = 1; len[a == 0 ? 1 : 0] = 1; }
        } else {
            const uint32_t one = 1u << maxbits;
            uint32_t K = 0;
            for (uint32_t s = 0; s < n; s++) if (len[s]) K += one >> len[s];
            // the cap can oversubscribe the code (many symbols rarer than 2^-maxbits): lengthen the rarest short codes
            for (int32_t r = (int32_t)used - 1; K > one && r >= 0; r--) {
                const uint32_t s = ord[r];
                while (K > one && len[s] < maxbits) { K -= one >> (len[s] + 1u); len[s]++; }
            }
            // complete the code: shorten, most frequent first, while it fits; the longest code can always be shortened while K < one
            while (K < one) {
                for (uint32_t r = 0; r < used && K < one; r++) {
                    const uint32_t s = ord[r];
                    while (len[s] > 1u && K + (one >> len[s]) <= one) { K += one >> len[s]; len[s]--; }
                }
            }
        }
    }
    __syncthreads();
}
// canonical codes (RFC 1951 3.2.2) for len[0..n), bit-reversed for LSB-first packing: tab[s] = code | len << 16
__device__ void dfl_assign_codes(const uint8_t *len, uint32_t *tab, uint32_t n, int lane) {
    if (lane == 0) {
        uint32_t blc[16]; for (int i = 0; i < 16; i++) blc[i] = 0;
        for (uint32_t s = 0; s < n; s++) blc[len[s]]++;
        blc[0] = 0;
        uint32_t nxt[16], code = 0;
        for (int b = 1; b < 16; b++) { code = (code + blc[b - 1]) << 1; nxt[b] = code; }
        for (uint32_t s = 0; s < n; s++) { const uint32_t L = len[s]; tab[s] = L ? (dfl_rev(nxt[L]++, L) | (L << 16)) : 0u; }
    }
    __syncthreads();
}

extern "C" __global__ void __launch_bounds__(64)
bgzf_deflate_blocks(const uint8_t *__restrict__ in, uint64_t n_in, int64_t nblk, int level, uint8_t *__restrict__ slots, uint32_t *__restrict__ sizes, uint32_t *__restrict__ tok_all) {
    extern __shared__ __attribute__((aligned(16))) uint8_t dsm[];
    uint16_t *tab = (uint16_t *)dsm;
    uint32_t *ring = (uint32_t *)(dsm + DFL_OFF_RING);
    uint32_t *crct = (uint32_t *)(dsm + DFL_OFF_CRCT);
    uint32_t *cnt_l = (uint32_t *)(dsm + DFL_OFF_CNT), *cnt_d = cnt_l + 288, *cnt_c = cnt_d + 32;
    // after the parse the hash table's room holds: code tables (288 + 32 + 20 words), code lengths (288 + 32 + 20 bytes), the code-length
    // sequence (<= 320 entries of u16: symbol | extra << 8), the rank scratch
    uint32_t *ct_l = (uint32_t *)dsm, *ct_d = ct_l + 288, *ct_c = ct_d + 32;
    uint8_t *ln_l = dsm + 1536, *ln_d = ln_l + 288, *ln_c = ln_d + 32;
    uint16_t *clseq = (uint16_t *)(dsm + 2048);
    uint32_t *ord = (uint32_t *)(dsm + 4096);
    const int lane = threadIdx.x;
    const int64_t bi = blockIdx.x;
    if (bi >= nblk) return;
    const uint64_t off = (uint64_t)bi * DFL_IN;
    const uint32_t n = (uint32_t)(n_in - off < DFL_IN ? n_in - off : DFL_IN);
    const uint8_t *src = in + off;
    uint8_t *slot = slots + (uint64_t)bi * DFL_SLOT;
    uint8_t *pay = slot + 18;
    uint32_t *tok = tok_all + (uint64_t)bi * DFL_IN;

    for (uint32_t k = lane; k < (1u << DFL_HASH_BITS); k += 64) tab[k] = DFL_NONE;
    for (uint32_t k = lane; k < DFL_RING_WORDS; k += 64) ring[k] = 0;
    for (uint32_t k = lane; k < 288u + 32u + 32u; k += 64) cnt_l[k] = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) *(uint4 *)(crct + q * 256 + lane * 4) = *(const uint4 *)(g_crcc + CRCC_TAB + q * 256 + lane * 4);
    __syncthreads();

    // ---- CRC-32 of the input: lane's piece (a multiple of 4 bytes), then x^(8 * bytes behind the piece) ----
    uint32_t crc;
    {
        const uint32_t piece = (((n + 63u) >> 6) + 3u) & ~3u;
        const uint32_t b = lane * piece < n ? lane * piece : n, e = b + piece < n ? b + piece : n;
        uint32_t c = lane == 0 ? 0xffffffffu : 0u, q = b;
        for (; q + 4 <= e; q += 4) {
            uint32_t v; __builtin_memcpy(&v, src + q, 4); v ^= c;
            c = crct[768 + (v & 0xff)] ^ crct[512 + ((v >> 8) & 0xff)] ^ crct[256 + ((v >> 16) & 0xff)] ^ crct[v >> 24];
        }
        for (; q < e; q++) c = crct[(c ^ src[q]) & 0xff] ^ (c >> 8);
        c = crc_mulmod(c, crc_xpow8_tab(n - e));
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) c ^= __shfl_xor(c, d, 64);
        crc = c ^ 0xffffffffu;
    }

    bool stored = level == 0 || n == 0;
    uint32_t bitpos = 0;
    if (!stored) {
        // ---- 1. parse ----
        uint32_t ntok = 0, skip = 0, extra_bits = 0;                  // extra_bits: length + distance extra bits of all matches (lane-partial)
        for (uint32_t base = 0; base < n; base += 64) {
            const uint32_t p = base + lane;
            const bool hashable = p + 4 <= n;
            uint32_t v = 0;
            if (p < n) __builtin_memcpy(&v, src + p, 4);          // (the input buffer is padded: reading up to 3 bytes past n is safe)
            // buckets of two: [0] the most recent position with this hash, [1] the one before
            const uint32_t h = ((v * 2654435761u) >> (32 - DFL_HASH_BITS + 1)) * 2u;
            const uint32_t c0 = hashable ? tab[h] : DFL_NONE, c1 = hashable ? tab[h + 1] : DFL_NONE;
            __syncthreads();                                       // every lane has its candidates before the step's positions are entered
            if (hashable) { tab[h + 1] = (uint16_t)c0; tab[h] = (uint16_t)p; }      // (lanes of one bucket write the same [1]; one of them wins [0])
            uint32_t len = 0, cand = DFL_NONE;
            const uint32_t maxl = n - p < 258u ? n - p : 258u;
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const uint32_t cc = k ? c1 : c0;
                if (cc == DFL_NONE || p - cc > 32768u || (k && cc == c0)) continue;
                uint32_t cv; __builtin_memcpy(&cv, src + cc, 4);
                if (cv != v) continue;
                uint32_t l = 4;
                while (l < maxl) {
                    uint64_t a, b; __builtin_memcpy(&a, src + p + l, 8); __builtin_memcpy(&b, src + cc + l, 8);
                    const uint64_t x = a ^ b;
                    if (x) { l += (uint32_t)(__builtin_ctzll(x) >> 3); break; }
                    l += 8;
                }
                l = l > maxl ? maxl : l;
                if (l > len) { len = l; cand = cc; }                // (the nearer candidate wins a tie: shorter distance code)
            }
            const uint32_t step = len >= 4u ? len : 1u;
            // greedy parse with one step of look-ahead (zlib's lazy evaluation, deflate.c deflate_slow): a match gives way to a literal when the
            // next position holds a longer one.  A walk over the step lengths from the first uncovered position, wave-uniform.
            const uint32_t step_next = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)step, 0x130, 0xf, 0xf, false);   // wave_shl:1 -> lane i gets lane i+1 (0 for lane 63)
            const uint32_t eff = (step > 1u && step_next > step && level != 1) ? 1u : step;
            uint64_t sel = 0; uint32_t q = skip;
            const uint32_t lim = n - base < 64u ? n - base : 64u;
            while (q < lim) { sel |= 1ull << q; q += (uint32_t)__builtin_amdgcn_readlane((int)eff, (int)q); }
            skip = q - lim;                                        // (only meaningful when lim == 64; the last step ends the loop)
            if ((sel >> lane) & 1ull) {
                const uint32_t idx = ntok + (uint32_t)__popcll(sel & ((1ull << lane) - 1ull));
                if (eff == 1u) { tok[idx] = v & 0xffu; atomicAdd(&cnt_l[v & 0xffu], 1u); }
                else {
                    uint32_t lc, leb, lev, dc, deb, dev;
                    dfl_len_code(len, lc, leb, lev); dfl_dist_code(p - cand - 1u, dc, deb, dev);
                    tok[idx] = 0x80000000u | ((len - 3u) << 16) | (p - cand - 1u);
                    atomicAdd(&cnt_l[257u + lc], 1u); atomicAdd(&cnt_d[dc], 1u);
                    extra_bits += leb + deb;
                }
            }
            ntok += (uint32_t)__popcll(sel);
        }
        if (lane == 0) cnt_l[256] = 1;                                // end of block
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) extra_bits += __shfl_xor(extra_bits, d, 64);
        __syncthreads();

        // ---- 2. codes (the hash table is no longer needed: its room holds the tables) ----
        dfl_build_lengths(cnt_l, ln_l, 286, 15, lane, ord);
        dfl_build_lengths(cnt_d, ln_d, 30, 15, lane, ord);
        // code-length sequence: HLIT + HDIST lengths, zero runs as 17 (3-10) / 18 (11-138)
        uint32_t hlit = 286, hdist = 30, nseq = 0;
        if (lane == 0) {
            while (hlit > 257u && ln_l[hlit - 1] == 0) hlit--;
            while (hdist > 1u && ln_d[hdist - 1] == 0) hdist--;
            const uint32_t tot = hlit + hdist;
            for (uint32_t i = 0; i < tot;) {
                const uint32_t L = i < hlit ? ln_l[i] : ln_d[i - hlit];
                if (L == 0) {
                    uint32_t r = 1; while (i + r < tot && r < 138u && (i + r < hlit ? ln_l[i + r] : ln_d[i + r - hlit]) == 0) r++;
                    if (r >= 11u) { clseq[nseq++] = (uint16_t)(18u | ((r - 11u) << 8)); cnt_c[18]++; i += r; continue; }
                    if (r >= 3u) { clseq[nseq++] = (uint16_t)(17u | ((r - 3u) << 8)); cnt_c[17]++; i += r; continue; }
                }
                clseq[nseq++] = (uint16_t)L; cnt_c[L]++; i++;
            }
        }
        hlit = (uint32_t)__builtin_amdgcn_readfirstlane((int)hlit); hdist = (uint32_t)__builtin_amdgcn_readfirstlane((int)hdist); nseq = (uint32_t)__builtin_amdgcn_readfirstlane((int)nseq);
        __syncthreads();
        dfl_build_lengths(cnt_c, ln_c, 19, 7, lane, ord);
        // sizes in bits
        uint32_t dyn_bits = 0, fix_bits = 0;
        for (uint32_t s = lane; s < 286u; s += 64) { dyn_bits += cnt_l[s] * ln_l[s]; fix_bits += cnt_l[s] * dfl_fixed_len(s); }
        for (uint32_t s = lane; s < 30u; s += 64) { dyn_bits += cnt_d[s] * ln_d[s]; fix_bits += cnt_d[s] * 5u; }
        for (uint32_t s = lane; s < 19u; s += 64) dyn_bits += cnt_c[s] * (ln_c[s] + (s == 17u ? 3u : s == 18u ? 7u : 0u));
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { dyn_bits += __shfl_xor(dyn_bits, d, 64); fix_bits += __shfl_xor(fix_bits, d, 64); }
        const uint8_t *clord = g_dfl_clord;
        uint32_t hclen = 19; while (hclen > 4u && ln_c[clord[hclen - 1]] == 0) hclen--;
        dyn_bits += 3u + 5u + 5u + 4u + 3u * hclen + extra_bits; fix_bits += 3u + extra_bits;
        const bool dynamic = dyn_bits < fix_bits;
        const uint32_t best = dynamic ? dyn_bits : fix_bits;
        if (((best + 7u) >> 3) >= n + 5u) stored = true;
        else {
            // tables for the chosen code
            if (dynamic) { dfl_assign_codes(ln_l, ct_l, 286, lane); dfl_assign_codes(ln_d, ct_d, 30, lane); dfl_assign_codes(ln_c, ct_c, 19, lane); }
            else {
                for (uint32_t s = lane; s < 288u; s += 64) {
                    const uint32_t L = dfl_fixed_len(s), c = s < 144u ? 0x30u + s : s < 256u ? 0x190u + (s - 144u) : s < 280u ? s - 256u : 0xc0u + (s - 280u);
                    ct_l[s] = dfl_rev(c, L) | (L << 16);
                }
                for (uint32_t s = lane; s < 30u; s += 64) ct_d[s] = dfl_rev(s, 5) | (5u << 16);
                __syncthreads();
            }
            // ---- 3. header, then the tokens ----
            uint32_t flushed = 0;
            if (lane == 0) {
                uint64_t acc = 0; uint32_t na = 0, w = 0;
                auto put = [&](uint32_t v, uint32_t nb) { acc |= (uint64_t)v << na; na += nb; if (na >= 32u) { ring[w++] = (uint32_t)acc; acc >>= 32; na -= 32u; } };
                put(1u, 1); put(dynamic ? 2u : 1u, 2);
                if (dynamic) {
                    put(hlit - 257u, 5); put(hdist - 1u, 5); put(hclen - 4u, 4);
                    for (uint32_t k = 0; k < hclen; k++) put(ln_c[clord[k]], 3);
                    for (uint32_t k = 0; k < nseq; k++) {
                        const uint32_t s = clseq[k] & 0xffu, x = clseq[k] >> 8, e = ct_c[s];
                        put(e & 0xffffu, e >> 16);
                        if (s == 17u) put(x, 3); else if (s == 18u) put(x, 7);
                    }
                }
                if (na) ring[w] = (uint32_t)acc;
                bitpos = w * 32u + na;
            }
            bitpos = (uint32_t)__builtin_amdgcn_readfirstlane((int)bitpos);
            __syncthreads();
            // (the header is < 256 words: 17 + 57 + 316 * (7 + 7) bits at most = 4,498 bits = 141 words)
            {
                const uint32_t full = bitpos >> 5;
                for (uint32_t wi = lane; wi < full; wi += 64) { const uint32_t rv = ring[wi]; __builtin_memcpy(pay + (uint64_t)wi * 4u, &rv, 4); }
                __syncthreads();
                for (uint32_t wi = lane; wi < full; wi += 64) ring[wi] = 0;
                flushed = full;
                // the partial word sits at ring[full]; the ring index of word wi is wi & 255, and full < 256: nothing to move
                __syncthreads();
            }
            for (uint32_t t0 = 0; t0 <= ntok; t0 += 64) {                 // token ntok is the end-of-block symbol
                const uint32_t ti = t0 + lane;
                uint64_t code = 0; uint32_t nb = 0;
                if (ti < ntok) {
                    const uint32_t t = tok[ti];
                    if (!(t & 0x80000000u)) { const uint32_t e = ct_l[t & 0xffu]; code = e & 0xffffu; nb = e >> 16; }
                    else {
                        const uint32_t len = ((t >> 16) & 0xffu) + 3u, d = t & 0x7fffu;
                        uint32_t lc, leb, lev, dc, deb, dev;
                        dfl_len_code(len, lc, leb, lev); dfl_dist_code(d, dc, deb, dev);
                        const uint32_t e = ct_l[257u + lc], f = ct_d[dc];
                        code = e & 0xffffu; nb = e >> 16;
                        code |= (uint64_t)lev << nb; nb += leb;
                        code |= (uint64_t)(f & 0xffffu) << nb; nb += f >> 16;
                        code |= (uint64_t)dev << nb; nb += deb;
                    }
                } else if (ti == ntok) { const uint32_t e = ct_l[256]; code = e & 0xffffu; nb = e >> 16; }
                const uint32_t incl = wave_incl_scan(nb, lane);
                const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                if (nb) {
                    const uint32_t bp = bitpos + incl - nb, w = bp >> 5, sh = bp & 31u;
                    const uint64_t lo = code << sh;                                    // nb <= 48, sh <= 31: up to 79 bits -> three words
                    atomicOr(&ring[w & (DFL_RING_WORDS - 1u)], (uint32_t)lo);
                    if (sh + nb > 32u) atomicOr(&ring[(w + 1u) & (DFL_RING_WORDS - 1u)], (uint32_t)(lo >> 32));
                    if (sh + nb > 64u) atomicOr(&ring[(w + 2u) & (DFL_RING_WORDS - 1u)], (uint32_t)(code >> (64u - sh)));
                }
                bitpos += total;
                __syncthreads();
                const uint32_t full = bitpos >> 5;                                      // <= 96 complete words per step
                for (uint32_t wi = flushed + lane; wi < full; wi += 64) {
                    const uint32_t rv = ring[wi & (DFL_RING_WORDS - 1u)];
                    __builtin_memcpy(pay + (uint64_t)wi * 4u, &rv, 4);
                    ring[wi & (DFL_RING_WORDS - 1u)] = 0;
                }
                flushed = full;
                __syncthreads();
            }
            if (lane == 0 && (bitpos & 31u)) { const uint32_t rv = ring[flushed & (DFL_RING_WORDS - 1u)]; __builtin_memcpy(pay + (uint64_t)flushed * 4u, &rv, 4); }
        }
    }
    uint32_t plen;
    if (stored) {
        // stored block: 01, LEN, NLEN, bytes
        if (lane == 0) { pay[0] = 1; pay[1] = (uint8_t)n; pay[2] = (uint8_t)(n >> 8); pay[3] = (uint8_t)~n; pay[4] = (uint8_t)(~n >> 8); }
        for (uint32_t k = lane * 16u; k < n; k += 1024u) {
            if (k + 16u <= n) { uint4 x; __builtin_memcpy(&x, src + k, 16); __builtin_memcpy(pay + 5 + k, &x, 16); }
            else for (uint32_t j = k; j < n; j++) pay[5 + j] = src[j];
        }
        plen = 5u + n;
    } else plen = (bitpos + 7u) >> 3;
    __syncthreads();
    if (lane == 0) {
        const uint32_t total = 18u + plen + 8u;
        const uint8_t hd[18] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, (uint8_t)((total - 1u) & 0xff), (uint8_t)((total - 1u) >> 8)};
        for (int k = 0; k < 18; k++) slot[k] = hd[k];
        uint8_t *tr = pay + plen;
        for (int k = 0; k < 4; k++) { tr[k] = (uint8_t)(crc >> (8 * k)); tr[4 + k] = (uint8_t)(n >> (8 * k)); }
        sizes[bi] = total;
    }
}

// slots -> the file: block bi's bytes to out + offs[bi]
extern "C" __global__ void __launch_bounds__(64)
bgzf_pack_blocks(const uint8_t *__restrict__ slots, const uint32_t *__restrict__ sizes, const uint64_t *__restrict__ offs, int64_t nblk, uint8_t *__restrict__ out) {
    const int64_t bi = blockIdx.x;
    if (bi >= nblk) return;
    const uint8_t *s = slots + (uint64_t)bi * DFL_SLOT; uint8_t *d = out + offs[bi];
    const uint32_t n = sizes[bi];
    for (uint32_t k = threadIdx.x * 16u; k < n; k += 1024u) {
        if (k + 16u <= n) { uint4 x = *(const uint4 *)(s + k); __builtin_memcpy(d + k, &x, 16); }
        else for (uint32_t j = k; j < n; j++) d[j] = s[j];
    }
}
