// gzip_serial.hip -- plain (non-BGZF) gzip members inflated on the device (gfx950); included by dhts_api.hip.
//
// Replaces, for inputs that are gzip but not BGZF (a .vcf.gz written by gzip(1) instead of bgzip):
//   bgzf_read_block's gzip branch: inflate_gzip_block + check_header      htslib bgzf.c:829-893, 1128-1130  (zlib's inflate over the members)
//   zlib's member loop: header (RFC 1952 2.3), DEFLATE (RFC 1951), CRC-32 + ISIZE trailer, the next member where another header follows
//
// A DEFLATE stream without BGZF's 64 KiB framing has one bit position that is known -- its start -- so this is a SERIAL decoder: one lane
// walks the stream (the review's item 7c: correctness before speed).  What keeps it from being slower than it has to be: the literal/length
// and distance codes are looked up in LDS tables (10 / 9 bits, longer codes by the canonical count walk), the last 32 KiB of output live in
// an LDS ring that matches copy from, so the output in HBM is only ever written (and, beyond the capacity given, only counted: one launch
// sizes and fills a buffer whose size was guessed right, a second one follows when it was not).  CRC-32 and ISIZE of every member are
// checked afterwards, in parallel (gz_crc_ranges; the host folds the pieces).  A stream that ends early (the bind context stages only the
// head of a file) gives what it had: `truncated` says so.
#pragma once

struct GzMember { unsigned long long out_beg, out_end; uint32_t crc, isize; };
struct GzResult {
    unsigned long long out_len;          // bytes the stream inflates to (also when they did not all fit)
    unsigned long long in_used;          // bytes of input consumed (whole members)
    uint32_t n_members, status;          // status: 0 ok, 1 truncated input (what was decoded stands), 2 invalid data, 3 more members than the list holds
    uint32_t pad[2];
};
#define GZ_IN_R 16384u                 /* input ring (LDS): the I/O wave keeps it filled ahead of the decoder */
#define GZ_OUT_R 65536u                /* output ring (LDS): the 32 KiB window behind the decoder + what the I/O wave has not flushed yet */
#define GZ_CH 4096u                    /* bytes the I/O wave moves per step: 64 bytes per lane */
#define GZ_LL_BITS 10
#define GZ_D_BITS 9

// what the two waves tell each other (LDS; relaxed / acquire / release at workgroup scope)
struct GzShared { unsigned long long in_avail, in_cons, out_pos, out_flushed, final_pos; uint32_t done, pad; };
#define GZ_LD(x) __hip_atomic_load(&(x), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)
#define GZ_ST(x, v) __hip_atomic_store(&(x), (v), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP)

// the decoder's view of the input: bytes [0, n) of the file through the LDS ring
struct GzIn {
    const uint8_t *ring; GzShared *sh; unsigned long long n, have;       // have: the in_avail last seen
    __device__ __forceinline__ void need(unsigned long long upto) {      // bytes [.., upto) must be in the ring (upto <= n)
        while (have < upto) { have = GZ_LD(sh->in_avail); if (have < upto) __builtin_amdgcn_s_sleep(2); }
    }
    __device__ __forceinline__ uint8_t byte(unsigned long long q) { need(q + 1); return ring[q & (GZ_IN_R - 1u)]; }
};
struct GzBits {
    GzIn *src; unsigned long long n, p;              // p: next byte to load
    unsigned long long buf; uint32_t cnt;            // bits behind the end of the input read as zeros; past_end() tells when one was consumed
    __device__ __forceinline__ void fill() {
        if (p + 8 <= n && ((p & (GZ_IN_R - 1u)) + 8 <= GZ_IN_R)) {
            // one 8-byte read: the bits above `cnt` that do not make a whole byte are the next byte's own low bits -- the next refill ORs the same
            // values onto them
            src->need(p + 8);
            unsigned long long w; __builtin_memcpy(&w, src->ring + (p & (GZ_IN_R - 1u)), 8);
            buf |= w << cnt; const uint32_t adv = (63u - cnt) >> 3; p += adv; cnt += adv * 8u;
        } else while (cnt <= 56) { if (p < n) buf |= (unsigned long long)src->byte(p) << cnt; p++; cnt += 8; }
        GZ_ST(src->sh->in_cons, p < n ? p : n);
    }
    __device__ __forceinline__ uint32_t peek(uint32_t k) { return (uint32_t)(buf & ((1ull << k) - 1ull)); }
    __device__ __forceinline__ void drop(uint32_t k) { buf >>= k; cnt -= k; }
    __device__ __forceinline__ uint32_t take(uint32_t k) { if (cnt < k) fill(); const uint32_t v = peek(k); drop(k); return v; }
    __device__ __forceinline__ bool past_end() const { return p > n + (cnt >> 3); }          // bits consumed lie beyond the input
    __device__ __forceinline__ unsigned long long byte_pos() const { return p - (cnt >> 3); }  // (after align())
    __device__ __forceinline__ void align() { drop(cnt & 7u); }
    __device__ __forceinline__ void seek(unsigned long long q) { p = q; buf = 0; cnt = 0; GZ_ST(src->sh->in_cons, q < n ? q : n); }
};

// canonical Huffman code (RFC 1951 3.2.2) of `n` symbols with lengths len[]: count[l], symbols in code order, and a primary table of
// 2^BITS entries (sym << 4 | len; 0 = longer code or none) indexed by the next bits as they come (LSB first = reversed code)
template <int BITS>
__device__ __forceinline__ bool gz_build(const uint8_t *len, int n, uint16_t *count, uint16_t *symbol, uint16_t *fast) {
    for (int l = 0; l <= 15; l++) count[l] = 0;
    for (int s = 0; s < n; s++) count[len[s]]++;
    for (int i = 0; i < (1 << BITS); i++) fast[i] = 0;
    if (count[0] == n) return true;                                                                  // no codes: legal for distances
    int left = 1;
    for (int l = 1; l <= 15; l++) { left <<= 1; left -= count[l]; if (left < 0) return false; }      // over-subscribed
    uint16_t offs[16]; offs[1] = 0;
    for (int l = 1; l < 15; l++) offs[l + 1] = offs[l] + count[l];
    for (int s = 0; s < n; s++) if (len[s]) symbol[offs[len[s]]++] = (uint16_t)s;
    uint32_t code = 0; int idx = 0;
    for (int l = 1; l <= BITS; l++) {
        for (int k = 0; k < count[l]; k++, code++, idx++) {
            const uint32_t r = __builtin_bitreverse32(code) >> (32 - l);
            const uint16_t e = (uint16_t)(symbol[idx] << 4 | l);
            for (uint32_t i = r; i < (1u << BITS); i += 1u << l) fast[i] = e;
        }
        code <<= 1;
    }
    return true;                                      // (an incomplete code is let through, as zlib lets a single-code distance tree through; a code word without a symbol fails when it is met)
}
template <int BITS>
__device__ __forceinline__ int gz_decode(GzBits &b, const uint16_t *count, const uint16_t *symbol, const uint16_t *fast) {
    if (b.cnt < 15) b.fill();
    const uint16_t e = fast[b.peek(BITS)];
    if (e) { b.drop(e & 15u); return e >> 4; }
    // canonical walk, a bit at a time (codes longer than the primary table)
    int code = 0, first = 0, index = 0; unsigned long long bits = b.buf;
    for (int l = 1; l <= 15; l++) {
        code |= (int)(bits & 1ull); bits >>= 1;
        const int c = count[l];
        if (code - c < first) { b.drop((uint32_t)l); return symbol[index + (code - first)]; }
        index += c; first += c; first <<= 1; code <<= 1;
    }
    return -1;
}
__device__ const uint8_t gz_clorder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// One workgroup of two waves.  Wave 0, lane 0: the decoder -- it touches LDS only (input ring, tables, output ring).  Wave 1: input and output:
// it keeps the input ring filled ahead of the decoder (4 KiB a step, coalesced) and writes what the decoder has produced to `out` (bytes
// behind out_cap are dropped: the launch then only measures).  A lane that waits sleeps; nothing else synchronises the two.
extern "C" __global__ void __launch_bounds__(128)
gz_inflate_serial(const uint8_t *__restrict__ in, unsigned long long in_len, uint8_t *__restrict__ out, unsigned long long out_cap, GzMember *__restrict__ members, uint32_t members_cap, GzResult *__restrict__ res) {
    __shared__ __attribute__((aligned(16))) uint8_t in_ring[GZ_IN_R];
    __shared__ __attribute__((aligned(16))) uint8_t ring[GZ_OUT_R];
    __shared__ uint16_t ll_fast[1 << GZ_LL_BITS], d_fast[1 << GZ_D_BITS], ll_count[16], d_count[16], ll_sym[288], d_sym[32], cl_fast[128], cl_count[16], cl_sym[19];
    __shared__ uint8_t lens[320];
    __shared__ GzShared sh;
    if (threadIdx.x == 0) { sh.in_avail = 0; sh.in_cons = 0; sh.out_pos = 0; sh.out_flushed = 0; sh.final_pos = 0; sh.done = 0; }
    __syncthreads();
    if (threadIdx.x >= 64) {
        // ---- the I/O wave
        const uint32_t lane = threadIdx.x - 64u;
        unsigned long long avail = 0, flushed = 0;
        for (;;) {
            bool worked = false;
            // input: the next 4 KiB when the ring has room for them (16 bytes of slack: the decoder's bit buffer may still hold bytes it has read)
            const unsigned long long cons = GZ_LD(sh.in_cons);
            if (avail < in_len && avail + GZ_CH <= (cons > 16 ? cons - 16 : 0) + GZ_IN_R) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const unsigned long long q = avail + (unsigned long long)k * 1024u + lane * 16u;
                    uint4 v = make_uint4(0, 0, 0, 0);
                    if (q + 16 <= in_len) v = *(const uint4 *)(in + q);
                    else if (q < in_len) { uint8_t t[16]; for (int j = 0; j < 16; j++) t[j] = q + j < in_len ? in[q + j] : 0; __builtin_memcpy(&v, t, 16); }
                    *(uint4 *)(in_ring + (q & (GZ_IN_R - 1u))) = v;
                }
                avail += GZ_CH;
                GZ_ST(sh.in_avail, avail < in_len ? avail : in_len);
                worked = true;
            }
            // output: whole 4 KiB pieces while the decoder runs, the rest when it has finished
            const uint32_t done = GZ_LD(sh.done);
            const unsigned long long produced = done ? GZ_LD(sh.final_pos) : GZ_LD(sh.out_pos);
            if (flushed + GZ_CH <= produced) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const unsigned long long q = flushed + (unsigned long long)k * 1024u + lane * 16u;
                    const uint4 v = *(const uint4 *)(ring + (q & (GZ_OUT_R - 1u)));
                    if (q + 16 <= out_cap) *(uint4 *)(out + q) = v;
                    else if (q < out_cap) { uint8_t t[16]; __builtin_memcpy(t, &v, 16); for (int j = 0; j < 16 && q + j < out_cap; j++) out[q + j] = t[j]; }
                }
                flushed += GZ_CH;
                GZ_ST(sh.out_flushed, flushed);
                worked = true;
            } else if (done) {
                for (unsigned long long q = flushed + lane; q < produced; q += 64) if (q < out_cap) out[q] = ring[q & (GZ_OUT_R - 1u)];
                break;
            }
            if (!worked) __builtin_amdgcn_s_sleep(4);
        }
        return;
    }
    if (threadIdx.x != 0) return;
    // ---- the decoder
    const unsigned long long t_clk0 = clock64(), t_rt0 = wall_clock64();
    GzIn src; src.ring = in_ring; src.sh = &sh; src.n = in_len; src.have = 0;
    GzBits b; b.src = &src; b.n = in_len; b.p = 0; b.buf = 0; b.cnt = 0;
    unsigned long long pos = 0, pub = 0, flushed = 0;       // output position; the one last published; out_flushed as last seen
    uint32_t n_mem = 0, status = 0; unsigned long long used = 0;
    auto room = [&](uint32_t w) {                           // the ring takes w more bytes without touching what has not been written out
        while (pos + w - flushed > GZ_OUT_R) { GZ_ST(sh.out_pos, pos); pub = pos; flushed = GZ_LD(sh.out_flushed); if (pos + w - flushed > GZ_OUT_R) __builtin_amdgcn_s_sleep(2); }
    };
    auto publish = [&]() { if (pos - pub >= 1024) { GZ_ST(sh.out_pos, pos); pub = pos; } };
    auto emit = [&](uint8_t v) { ring[pos & (GZ_OUT_R - 1u)] = v; pos++; };
    for (;;) {
        // ---- member header (RFC 1952): ID1 ID2 CM FLG MTIME(4) XFL OS [XLEN + extra] [name\0] [comment\0] [HCRC]
        b.align();
        const unsigned long long h0 = b.byte_pos();
        if (h0 >= in_len) break;                                                          // clean end
        if (h0 + 10 > in_len) { status = n_mem ? 0u : 1u; break; }                        // (bytes behind the last member, too few for a header: in_used tells the host, which ends the stream in an error as zlib does)
        if (src.byte(h0) != 0x1f || src.byte(h0 + 1) != 0x8b) { if (!n_mem) status = 2; break; }      // bytes behind a member that are not a member: in_used tells the host
        if (src.byte(h0 + 2) != 8 || (src.byte(h0 + 3) & 0xe0)) { status = 2; break; }
        const uint32_t flg = src.byte(h0 + 3);
        unsigned long long q = h0 + 10; bool cut = false;
        b.seek(q);                                                                        // (lets the I/O wave move on while the variable fields are walked)
        if (flg & 4) { if (q + 2 > in_len) cut = true; else { const uint32_t xl = src.byte(q) | (uint32_t)src.byte(q + 1) << 8; q += 2 + xl; } }
        if (!cut && q < in_len) b.seek(q);
        if (!cut && (flg & 8)) { while (q < in_len && src.byte(q)) { q++; if ((q & 1023u) == 0) b.seek(q); } q++; }
        if (!cut && q < in_len) b.seek(q);
        if (!cut && (flg & 16)) { while (q < in_len && src.byte(q)) { q++; if ((q & 1023u) == 0) b.seek(q); } q++; }
        if (!cut && (flg & 2)) q += 2;
        if (cut || q > in_len) { status = 1; break; }
        b.seek(q);
        const unsigned long long m_beg = pos;
        // ---- DEFLATE blocks
        bool bad = false, trunc = false;
        for (;;) {
            const uint32_t bfinal = b.take(1), btype = b.take(2);
            if (b.past_end()) { trunc = true; break; }
            if (btype == 0) {                                                             // stored
                b.align();
                if (b.cnt < 32) b.fill();
                const uint32_t len = b.take(16), nlen = b.take(16);
                if (b.past_end()) { trunc = true; break; }
                if ((len ^ 0xffffu) != nlen) { bad = true; break; }
                unsigned long long sp = b.byte_pos();
                const bool cut_short = sp + len > in_len;                                  // the file ends inside the block: what is there is output
                const unsigned long long se = cut_short ? in_len : sp + len;
                b.seek(sp);
                for (; sp < se; sp++) { room(1); emit(src.byte(sp)); if ((sp & 255u) == 0) { b.seek(sp); publish(); } }
                b.seek(se);
                if (cut_short) { trunc = true; break; }
            } else if (btype == 3) { bad = true; break; }
            else {
                if (btype == 1) {                                                         // fixed codes
                    for (int s = 0; s < 144; s++) lens[s] = 8;
                    for (int s = 144; s < 256; s++) lens[s] = 9;
                    for (int s = 256; s < 280; s++) lens[s] = 7;
                    for (int s = 280; s < 288; s++) lens[s] = 8;
                    gz_build<GZ_LL_BITS>(lens, 288, ll_count, ll_sym, ll_fast);
                    for (int s = 0; s < 30; s++) lens[s] = 5;
                    gz_build<GZ_D_BITS>(lens, 30, d_count, d_sym, d_fast);
                } else {                                                                  // dynamic codes
                    const uint32_t hlit = b.take(5) + 257, hdist = b.take(5) + 1, hclen = b.take(4) + 4;
                    if (hlit > 286 || hdist > 30) { bad = true; break; }
                    uint8_t cl[19]; for (int k = 0; k < 19; k++) cl[k] = 0;
                    for (uint32_t k = 0; k < hclen; k++) cl[gz_clorder[k]] = (uint8_t)b.take(3);
                    if (b.past_end()) { trunc = true; break; }
                    if (!gz_build<7>(cl, 19, cl_count, cl_sym, cl_fast)) { bad = true; break; }
                    uint32_t k = 0;
                    while (k < hlit + hdist) {
                        const int s = gz_decode<7>(b, cl_count, cl_sym, cl_fast);
                        if (s < 0) { bad = true; break; }
                        if (s < 16) lens[k++] = (uint8_t)s;
                        else {
                            uint32_t rep, v = 0;
                            if (s == 16) { if (k == 0) { bad = true; break; } v = lens[k - 1]; rep = 3 + b.take(2); }
                            else if (s == 17) rep = 3 + b.take(3);
                            else rep = 11 + b.take(7);
                            if (k + rep > hlit + hdist) { bad = true; break; }
                            while (rep--) lens[k++] = (uint8_t)v;
                        }
                        if (b.past_end()) break;
                    }
                    if (bad) break;
                    if (b.past_end()) { trunc = true; break; }
                    if (lens[256] == 0) { bad = true; break; }                            // no end-of-block code
                    if (!gz_build<GZ_LL_BITS>(lens, (int)hlit, ll_count, ll_sym, ll_fast)) { bad = true; break; }
                    uint8_t dl[32]; for (uint32_t i = 0; i < hdist; i++) dl[i] = lens[hlit + i];
                    if (!gz_build<GZ_D_BITS>(dl, (int)hdist, d_count, d_sym, d_fast)) { bad = true; break; }
                }
                // ---- symbols
                for (;;) {
                    room(272);
                    publish();
                    const int s = gz_decode<GZ_LL_BITS>(b, ll_count, ll_sym, ll_fast);
                    if (b.past_end()) { trunc = true; break; }
                    if (s < 0) { bad = true; break; }
                    if (s < 256) { emit((uint8_t)s); continue; }
                    if (s == 256) break;
                    if (s > 285) { bad = true; break; }
                    // length and distance: base and extra bits by arithmetic (RFC 1951 3.2.5)
                    const uint32_t li = (uint32_t)s - 257u;
                    uint32_t len;
                    if (li < 8) len = 3 + li; else if (li == 28) len = 258; else { const uint32_t e = (li >> 2) - 1; len = 3 + ((4 + (li & 3)) << e) + b.take(e); }
                    const int ds = gz_decode<GZ_D_BITS>(b, d_count, d_sym, d_fast);
                    if (ds < 0 || ds > 29) { if (b.past_end()) trunc = true; else bad = true; break; }
                    uint32_t dist;
                    if (ds < 4) dist = 1 + (uint32_t)ds; else { const uint32_t e = ((uint32_t)ds >> 1) - 1; dist = 1 + ((2 + ((uint32_t)ds & 1)) << e) + b.take(e); }
                    if (b.past_end()) { trunc = true; break; }
                    if ((unsigned long long)dist > pos - m_beg) { bad = true; break; }     // (a member's window starts with the member)
                    // the copy: eight bytes a step while source and target do not overlap and neither wraps in the ring, else a byte a step
                    while (len >= 8 && dist >= 8) {
                        const uint32_t si = (uint32_t)((pos - dist) & (GZ_OUT_R - 1u)), ti = (uint32_t)(pos & (GZ_OUT_R - 1u));
                        if (si + 8 > GZ_OUT_R || ti + 8 > GZ_OUT_R) break;
                        unsigned long long v; __builtin_memcpy(&v, ring + si, 8); __builtin_memcpy(ring + ti, &v, 8);
                        pos += 8; len -= 8;
                    }
                    for (; len; len--) emit(ring[(pos - dist) & (GZ_OUT_R - 1u)]);
                }
                if (bad || trunc) break;
            }
            if (bfinal) break;
        }
        if (bad) { status = 2; break; }
        if (trunc) { status = 1; break; }
        // ---- trailer: CRC-32, ISIZE
        b.align();
        const unsigned long long t0 = b.byte_pos();
        if (t0 + 8 > in_len) { status = 1; break; }
        if (n_mem < members_cap) {
            GzMember m; m.out_beg = m_beg; m.out_end = pos;
            m.crc = src.byte(t0) | (uint32_t)src.byte(t0 + 1) << 8 | (uint32_t)src.byte(t0 + 2) << 16 | (uint32_t)src.byte(t0 + 3) << 24;
            m.isize = src.byte(t0 + 4) | (uint32_t)src.byte(t0 + 5) << 8 | (uint32_t)src.byte(t0 + 6) << 16 | (uint32_t)src.byte(t0 + 7) << 24;
            members[n_mem] = m;
        } else status = 3;
        n_mem++;
        b.seek(t0 + 8);
        used = t0 + 8;
        if (status) break;
    }
    GZ_ST(sh.final_pos, pos);
    GZ_ST(sh.done, 1u);
    res->out_len = pos; res->in_used = used; res->n_members = n_mem; res->status = status;
    { const unsigned long long dc = clock64() - t_clk0, dr = wall_clock64() - t_rt0; res->pad[0] = (uint32_t)(dc >> 10); res->pad[1] = (uint32_t)(dr >> 10); }     // shader cycles / 1024, 100 MHz ticks / 1024 (DHTS_TRACE: what the lone decoder wave cost)
}

// CRC-32 state (from 0, no final xor) of out[off[i], off[i] + len[i]): a lane per range, byte-wise through the first table of g_crcc
extern "C" __global__ void __launch_bounds__(256)
gz_crc_ranges(const uint8_t *__restrict__ out, const unsigned long long *__restrict__ off, const uint32_t *__restrict__ len, uint32_t n, uint32_t *__restrict__ state) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t *p = out + off[i]; const uint32_t l = len[i];
    const uint32_t *t = g_crcc + CRCC_TAB;
    uint32_t c = 0, k = 0;
    for (; k + 4 <= l; k += 4) {
        uint32_t w; __builtin_memcpy(&w, p + k, 4);
        c ^= w;
        c = t[768 + (c & 0xffu)] ^ t[512 + ((c >> 8) & 0xffu)] ^ t[256 + ((c >> 16) & 0xffu)] ^ t[c >> 24];
    }
    for (; k < l; k++) c = t[(c ^ p[k]) & 0xffu] ^ (c >> 8);
    state[i] = c;
}
