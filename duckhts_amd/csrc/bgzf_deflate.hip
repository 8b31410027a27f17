// bgzf_deflate.hip -- the write side of BGZF: raw bytes -> BGZF blocks on the device (src/bgzip.c -> bgzf_write /
// bgzf_compress, htslib bgzf.c:509-620: every 0xff00 input bytes become one gzip member with the BC extra field, CRC-32 and ISIZE).
//
// One wave per BGZF block.  The DEFLATE stream of a block is ONE block with the fixed Huffman code (RFC 1951 3.2.6), so that match finding
// and bit packing fuse into a single pass with no token scratch in HBM:
//   * 64 consecutive input positions per step; lane i hashes the 4 bytes at its position into a 2^13-entry table of last positions in LDS
//     (read the candidate, then insert), verifies the candidate byte for byte (8 bytes per compare, up to 258) -- greedy, one candidate;
//   * the greedy parse of the step is a walk over the lanes' step lengths (a wave-uniform loop over v_readlane, one iteration per token);
//   * every chosen lane builds its token's bits (<= 31: 7-8 bit length code + <= 5 extra + 5 bit distance code + <= 13 extra, or an 8-9 bit
//     literal), a wave scan of the bit counts gives the positions, ds_or packs them into a 512-byte ring in LDS whose complete words leave
//     as one coalesced store per step.
// A block whose coded size reaches its stored size is written as a stored block (also level 0).  The CRC-32 of the input is computed by the
// same wave (64 pieces, slice-by-4, combined with x^(8n) multipliers: the constants of bgzf_inflate.hip).  Output: one 65,536-byte slot per
// block + its size; bgzf_pack_blocks copies the slots to their final offsets (exclusive scan of the sizes).
//
// The bytes differ from zlib's for the same input (any valid DEFLATE stream is a valid answer; the reference's own output depends on the zlib
// / libdeflate it was linked with); what is tested is that every reader gives the input back and that the container is BGZF.
#pragma once
#include "dhts_common.h"

#define DFL_IN 65280u                    /* BGZF_BLOCK_SIZE 0xff00: input bytes per block (bgzf.c:66) */
#define DFL_SLOT 65536u                  /* BGZF_MAX_BLOCK_SIZE: a block, header and trailer included, never exceeds it */
#define DFL_HASH_BITS 13
#define DFL_NONE 0xffffu
#define DFL_RING_WORDS 128u
#define DFL_LDS_BYTES ((1u << DFL_HASH_BITS) * 2u + DFL_RING_WORDS * 4u + 4096u)

__device__ __forceinline__ uint32_t dfl_rev(uint32_t v, uint32_t nbits) { return __brev(v) >> (32u - nbits); }

extern "C" __global__ void __launch_bounds__(64)
bgzf_deflate_blocks(const uint8_t *__restrict__ in, uint64_t n_in, int64_t nblk, int level, uint8_t *__restrict__ slots, uint32_t *__restrict__ sizes) {
    extern __shared__ __attribute__((aligned(16))) uint8_t dsm[];
    uint16_t *tab = (uint16_t *)dsm;
    uint32_t *ring = (uint32_t *)(dsm + (1u << DFL_HASH_BITS) * 2u);
    uint32_t *crct = (uint32_t *)(dsm + (1u << DFL_HASH_BITS) * 2u + DFL_RING_WORDS * 4u);
    const int lane = threadIdx.x;
    const int64_t bi = blockIdx.x;
    if (bi >= nblk) return;
    const uint64_t off = (uint64_t)bi * DFL_IN;
    const uint32_t n = (uint32_t)(n_in - off < DFL_IN ? n_in - off : DFL_IN);
    const uint8_t *src = in + off;
    uint8_t *slot = slots + (uint64_t)bi * DFL_SLOT;
    uint8_t *pay = slot + 18;

    for (uint32_t k = lane; k < (1u << DFL_HASH_BITS); k += 64) tab[k] = DFL_NONE;
    for (uint32_t k = lane; k < DFL_RING_WORDS; k += 64) ring[k] = 0;
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

    // ---- one fixed-Huffman DEFLATE block ----
    bool stored = level == 0 || n == 0;
    uint32_t bitpos = 3, flushed = 0;                 // bits written / ring words already stored to `pay`
    if (!stored) {
        if (lane == 0) ring[0] = 3u;                  // BFINAL = 1, BTYPE = 01
        uint32_t skip = 0;                            // positions at the start of the step that the previous step's last match covers
        for (uint32_t base = 0; base < n; base += 64) {
            const uint32_t p = base + lane;
            const bool hashable = p + 4 <= n;
            uint32_t v = 0;
            if (p < n) __builtin_memcpy(&v, src + p, 4);          // (the input buffer is padded: reading up to 3 bytes past n is safe)
            const uint32_t h = (v * 2654435761u) >> (32 - DFL_HASH_BITS);
            const uint32_t cand = hashable ? tab[h] : DFL_NONE;
            __syncthreads();                                       // every lane has its candidate before the step's positions are entered
            if (hashable) tab[h] = (uint16_t)p;
            uint32_t len = 0;
            if (cand != DFL_NONE && p - cand <= 32768u) {
                uint32_t cv; __builtin_memcpy(&cv, src + cand, 4);
                if (cv == v) {
                    const uint32_t maxl = n - p < 258u ? n - p : 258u;
                    len = 4;
                    while (len < maxl) {
                        uint64_t a, b; __builtin_memcpy(&a, src + p + len, 8); __builtin_memcpy(&b, src + cand + len, 8);
                        const uint64_t x = a ^ b;
                        if (x) { len += (uint32_t)(__builtin_ctzll(x) >> 3); break; }
                        len += 8;
                    }
                    len = len > maxl ? maxl : len;
                }
            }
            const uint32_t step = len >= 4u ? len : 1u;
            // greedy parse: walk the step lengths from the first uncovered position
            uint64_t sel = 0; uint32_t q = skip;
            const uint32_t lim = n - base < 64u ? n - base : 64u;
            while (q < lim) { sel |= 1ull << q; q += (uint32_t)__builtin_amdgcn_readlane((int)step, (int)q); }
            skip = q - lim;                                        // (only meaningful when lim == 64; the last step ends the loop)
            const bool mine = (sel >> lane) & 1ull;
            uint32_t code = 0, nb = 0;
            if (mine) {
                if (step == 1u) {
                    const uint32_t b = v & 0xffu;
                    if (b < 144u) { code = dfl_rev(0x30u + b, 8); nb = 8; } else { code = dfl_rev(0x190u + (b - 144u), 9); nb = 9; }
                } else {
                    // length: symbol 257 + lc, eb extra bits (RFC 1951 3.2.5)
                    const uint32_t l = len - 3u;
                    uint32_t lc, leb, lev;
                    if (len == 258u) { lc = 28; leb = 0; lev = 0; }
                    else if (l < 8u) { lc = l; leb = 0; lev = 0; }
                    else { const uint32_t hb = 31u - (uint32_t)__builtin_clz(l); leb = hb - 2u; lc = (leb << 2) + ((l >> leb) & 3u) + 4u; lev = l & ((1u << leb) - 1u); }
                    if (lc < 23u) { code = dfl_rev(lc + 1u, 7); nb = 7; }                   // symbols 257..279: 7-bit codes 0000001..
                    else { code = dfl_rev(0xc0u + (lc - 23u), 8); nb = 8; }                 // symbols 280..285: 8-bit codes 11000000..
                    code |= lev << nb; nb += leb;
                    const uint32_t d = p - cand - 1u;
                    uint32_t dc, deb, dev;
                    if (d < 4u) { dc = d; deb = 0; dev = 0; }
                    else { const uint32_t hb = 31u - (uint32_t)__builtin_clz(d); deb = hb - 1u; dc = (deb << 1) + ((d >> deb) & 1u) + 2u; dev = d & ((1u << deb) - 1u); }
                    code |= dfl_rev(dc, 5) << nb; nb += 5;
                    code |= dev << nb; nb += deb;
                }
            }
            const uint32_t incl = wave_incl_scan(nb, lane);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            if (mine) {
                const uint32_t bp = bitpos + incl - nb, w = (bp >> 5) & (DFL_RING_WORDS - 1u), sh = bp & 31u;
                atomicOr(&ring[w], code << sh);
                if (sh + nb > 32u) atomicOr(&ring[(w + 1u) & (DFL_RING_WORDS - 1u)], code >> (32u - sh));
            }
            bitpos += total;
            __syncthreads();
            // complete words leave the ring (<= 62 per step)
            const uint32_t full = bitpos >> 5;
            if (flushed + lane < full) {
                const uint32_t wi = flushed + lane, rv = ring[wi & (DFL_RING_WORDS - 1u)];
                __builtin_memcpy(pay + (uint64_t)wi * 4u, &rv, 4);
                ring[wi & (DFL_RING_WORDS - 1u)] = 0;
            }
            flushed = full;
            __syncthreads();
            if ((bitpos >> 3) + 16u > n + 5u) { stored = true; break; }          // no gain over a stored block (wave-uniform)
        }
        if (!stored) {
            bitpos += 7;                                                           // end of block: symbol 256 = 0000000
            __syncthreads();
            const uint32_t words = (bitpos + 31u) >> 5;
            for (uint32_t wi = flushed + lane; wi < words; wi += 64) { const uint32_t rv = ring[wi & (DFL_RING_WORDS - 1u)]; __builtin_memcpy(pay + (uint64_t)wi * 4u, &rv, 4); }
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
