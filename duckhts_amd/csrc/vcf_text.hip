// vcf_text.hip -- VCF TEXT lines -> BCF2 records on the device (gfx950), so that read_bcf's record stage (bcf_rec_check, bcf_cells) serves
// text input unchanged.  SURVEY.md 8(f) item 3, first half; included by dhts_api.hip.
//
// Replaces, for sites-only files (no sample columns):
//   vcf_read / hts_getline                       htslib vcf.c:4170-4176          -> vcf_line_count / vcf_line_fill (line index of a batch)
//   vcf_parse (CHROM .. INFO)                    vcf.c:3987-4165                 -> vcf_encode<false> (sizes, undefined names) + vcf_encode<true>
//   vcf_parse_filter, vcf_parse_info             vcf.c:3763-3816, 3818-3985
//   hts_str2uint / hts_str2int / hts_str2dbl     textutils_internal.h:218-428    (the fast path of hts_str2dbl is one IEEE division; what it
//                                                                                 hands to strtod -- exponents, > 14 digits, inf / nan / hex --
//                                                                                 and QUAL's atof outside that form go to the host as patches)
//   bcf_enc_size / bcf_enc_int1 / bcf_enc_vchar  htslib/vcf.h, vcf.c:2834-2980   (integers are written as int32 vectors: every width decodes to
//                                                                                 the same values in the getters)
// Names a record uses without a header definition (contig, FILTER, INFO key) are reported to the host, which adds htslib's dummy definitions
// (fix_chromosome vcf.c:3744-3761, the "Dummy" lines of vcf_parse_filter / vcf_parse_info) in order of first appearance and runs the batch
// again.  One lane per line; two passes (measure, then write behind an exclusive scan of the record lengths).
#pragma once

struct VcfDictDev { const uint32_t *off; const uint8_t *bytes; const int32_t *id; const uint8_t *ityp; const uint8_t *ftyp; int32_t n; const uint32_t *hash; uint32_t hmask; };   // sorted by name (byte order); ityp / ftyp: INFO / FORMAT type of the id, 15 = none
struct VcfUndef { uint32_t line, pos, len, cls; };        // cls: 0 contig, 1 FILTER, 2 INFO key, 3 FORMAT key; 4: a FORMAT Float the host has to convert AND check (strtod must stop at the
                                                          // end of the token, else "Invalid character"); pos = offset of the name / token in the batch text
struct VcfPatch { uint32_t pos, len, dst, kind; };        // kind 0: (float)atof(token) -> f32 at dst; 1: hts_str2dbl(token) -> f32 (or missing) at dst; 2: the same, 0.0 when it fails (FORMAT)
struct VcfArgs {
    const uint8_t *u; const uint32_t *line_off; int64_t nlines; uint64_t text_end;   // line i = u[line_off[i], line_off[i+1] - 1) (the last one ends at text_end when it has no newline)
    int32_t last_open;                                      // 1: the last line has no terminating newline
    uint32_t lds_budget;                                    // bytes of LDS a workgroup may stage its lines in (0: parse from HBM)
    int32_t n_smp, v44;                                     // samples of the header; header version >= VCFv4.4 (a leading '/' or '|' in GT is a phasing prefix)
    VcfDictDev ctg, ids;
    uint32_t *rec_len; const uint32_t *rec_off; uint8_t *out;
    unsigned long long *first_bad;
    uint32_t *counters;                                     // [0] undefined names, [1] patches
    VcfUndef *undef; uint32_t undef_cap; VcfPatch *patch; uint32_t patch_cap;
    // projection pushdown into the encoder: bit `id` of info_keep set = a projected column reads INFO key `id` (nullptr: every key is kept).
    // A key outside the set is looked up (names without a definition are reported whatever is projected) and counted towards the
    // 65535-entry limit, but its value is neither parsed nor written and the record's n_info counts the kept fields only; nothing in an
    // INFO value can fail a line (vcf_parse_info turns what it cannot convert into a missing value), so the rows are the same.
    // info_none: no INFO key is kept at all (the write pass skips INFO).
    const uint32_t *info_keep; int32_t info_none;
    const uint32_t *fmt_keep;                               // bit `id` set = a projected column reads FORMAT key `id` (nullptr: every key is kept): another key's values are validated by the
                                                            // measure pass and left out of the record (as a repeated key's are, without the validation, in htslib and here)
    int32_t fmt_none;                                       // no FORMAT column is projected: the sample columns are validated by the measure pass (a line that fails there ends the scan, whatever is
                                                            // projected) and left out of the records: the write pass does not look at them
    int32_t *pos_hi;                                        // [line] bits 32.. of the record's 0-based position (hts_pos_t is 64 bits wide for text, vcf.c:4052-4063)
    uint32_t *endsv;                                        // wave kernel: [2 * line] where the values of the line's first END= / SVLEN= fields begin (0xffffffff: none), measure pass -> write pass
};

// Line index of a batch: chunks of 4 KiB (one workgroup, 16 bytes per lane, coalesced); newline count per chunk, an exclusive scan over
// the chunks on the host side (run_scan), then every newline's successor position lands at its rank.
#define VCF_CHUNK 4096u
__device__ __forceinline__ uint32_t vcf_nl_mask16(const uint8_t *__restrict__ u, uint64_t p, uint64_t ulen) {      // bit k set: u[p + k] == '\n'
    uint32_t m = 0;
    if (p + 16 <= ulen && (p & 15) == 0) {
        const uint4 v = *(const uint4 *)(u + p);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t x = w[q] ^ 0x0a0a0a0au;                              // zero bytes where the byte is '\n'
#pragma unroll
            for (int b = 0; b < 4; b++) if (((x >> (8 * b)) & 0xffu) == 0) m |= 1u << (4 * q + b);
        }
    } else for (uint32_t k = 0; k < 16 && p + k < ulen; k++) if (u[p + k] == '\n') m |= 1u << k;
    return m;
}
extern "C" __global__ void __launch_bounds__(256)
vcf_line_count(const uint8_t *__restrict__ u, uint64_t start, uint64_t ulen, uint32_t *__restrict__ cnt, int64_t nchunks) {
    __shared__ uint32_t wsum[4];
    const int64_t k = blockIdx.x;
    const uint64_t base = (start & ~(uint64_t)15) + (uint64_t)k * VCF_CHUNK, p = base + threadIdx.x * 16u;
    uint32_t m = p < ulen ? vcf_nl_mask16(u, p, ulen) : 0u;
    if (p < start) m = p + 16 <= start ? 0u : m & ~((1u << (start - p)) - 1u);       // bytes in front of the first line do not count
    uint32_t n = __popc(m);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) n += __shfl_xor(n, d, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) cnt[k] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
extern "C" __global__ void __launch_bounds__(256)
vcf_line_fill(const uint8_t *__restrict__ u, uint64_t start, uint64_t ulen, const uint32_t *__restrict__ base_of, uint32_t *__restrict__ line_off, int64_t nchunks) {
    __shared__ uint32_t wsum[4];
    const int64_t k = blockIdx.x;
    if (k == 0 && threadIdx.x == 0) line_off[0] = (uint32_t)start;
    const uint64_t base = (start & ~(uint64_t)15) + (uint64_t)k * VCF_CHUNK, p = base + threadIdx.x * 16u;
    uint32_t m = p < ulen ? vcf_nl_mask16(u, p, ulen) : 0u;
    if (p < start) m = p + 16 <= start ? 0u : m & ~((1u << (start - p)) - 1u);
    const uint32_t n = __popc(m);
    uint32_t incl = n;                                                          // inclusive scan inside the wave, then across the four waves
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(incl, d, 64); if ((int)(threadIdx.x & 63) >= d) incl += t; }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t rank = base_of[k] + incl - n;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) rank += wsum[w];
    while (m) { const uint32_t b = __ffs(m) - 1; m &= m - 1; line_off[++rank] = (uint32_t)(p + b + 1); }
}

// Name -> its position in the sorted table.  A parser's lookups are chains of dependent loads, so the chain is kept short: an open-addressing
// table over FNV-1a of the name (built by the host next to the sorted arrays; slot = position + 1, 0 = free) finds the one candidate, and the
// bytes are compared eight at a time (eight loads in flight, one wait).
__host__ __device__ __forceinline__ uint32_t vcf_name_hash(const uint8_t *s, uint32_t l) { uint32_t h = 2166136261u; for (uint32_t i = 0; i < l; i++) h = (h ^ s[i]) * 16777619u; return h; }
__device__ __forceinline__ bool vcf_bytes_equal(const uint8_t *m, const uint8_t *s, uint32_t l) {
    uint32_t i = 0;
    for (; i + 8 <= l; i += 8) {
        uint64_t x = 0, y = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) { x |= (uint64_t)m[i + k] << (8 * k); y |= (uint64_t)s[i + k] << (8 * k); }
        if (x != y) return false;
    }
    uint64_t x = 0, y = 0;                                   // up to seven bytes are left (a 32-bit accumulator dropped the first of them: soak seeds 3037 / 3086)
    for (; i < l; i++) { x = (x << 8) | m[i]; y = (y << 8) | s[i]; }
    return x == y;
}
__device__ __forceinline__ int vcf_dict_find(const VcfDictDev &d, const uint8_t *s, uint32_t l) {
    if (d.n <= 0) return -1;
    for (uint32_t h = vcf_name_hash(s, l) & d.hmask;; h = (h + 1) & d.hmask) {
        const uint32_t e = d.hash[h];
        if (!e) return -1;
        const uint32_t o0 = d.off[e - 1], o1 = d.off[e];
        if (o1 - o0 == l && vcf_bytes_equal(d.bytes + o0, s, l)) return (int)(e - 1);
    }
}

// byte sink of one record: counts in the measure pass, stores in the write pass
template <bool WRITE> struct VcfSink {
    uint8_t *p; uint32_t n;
    __device__ __forceinline__ void b(uint8_t v) { if (WRITE) p[n] = v; n++; }
    __device__ __forceinline__ void w32(uint32_t v) { if (WRITE) __builtin_memcpy(p + n, &v, 4); n += 4; }                    // (unaligned dword store)
    __device__ __forceinline__ void bytes(const uint8_t *s, uint32_t l) {
        if (WRITE) {                                        // byte reads (the source may be the LDS copy behind a generic pointer), 8-byte unaligned stores
            uint32_t i = 0;
            for (; i + 8 <= l; i += 8) { uint64_t v = 0; for (int k = 0; k < 8; k++) v |= (uint64_t)s[i + k] << (8 * k); __builtin_memcpy(p + n + i, &v, 8); }
            for (; i < l; i++) p[n + i] = s[i];
        }
        n += l;
    }
    __device__ __forceinline__ void size(uint32_t cnt, uint32_t type) {                               // bcf_enc_size
        if (cnt < 15) { b((uint8_t)(cnt << 4 | type)); return; }
        b((uint8_t)(0xF0 | type));
        if (cnt < 128) { b(0x11); b((uint8_t)cnt); }
        else if (cnt < 32768) { b(0x12); b((uint8_t)cnt); b((uint8_t)(cnt >> 8)); }
        else { b(0x13); w32(cnt); }
    }
    __device__ __forceinline__ void key(int32_t x) {                                                  // bcf_enc_int1 of a dictionary id
        if (x <= 127) { b(0x11); b((uint8_t)x); }
        else if (x <= 32767) { b(0x12); b((uint8_t)x); b((uint8_t)(x >> 8)); }
        else { b(0x13); w32((uint32_t)x); }
    }
    __device__ __forceinline__ void vchar(const uint8_t *s, uint32_t l) { size(l, 7); bytes(s, l); }
};

__device__ __forceinline__ bool vcf_isspace(uint8_t c) { return c == ' ' || (c >= 9 && c <= 13); }

// Decimal text -> double, correctly rounded, for what hts_str2dbl / strtod / atof all agree on: [ws][+-](digits[.digits] | .digits)[e[+-]digits]
// with at most 15 significant digits and a power of ten that keeps both operands of ONE IEEE multiplication or division exact (Clinger's
// fast path: w < 2^53, 10^k <= 10^22).  hts_str2dbl's own fast path (n / 10^k, textutils_internal.h:354-428) is the k <= 14 slice of this;
// a correctly rounded result is unique, so the values are the same whoever computes them.  0 = converted (*val, *end = characters the
// number takes, as strtod's end pointer), 1 = something else (inf / nan / hex, more digits, far exponents, nothing numeric): the host
// converts that token.
__device__ __forceinline__ int vcf_str2dbl_fast(const uint8_t *s, uint32_t l, double *val, uint32_t *end) {
    uint32_t v = 0; bool neg = false;
    while (v < l && vcf_isspace(s[v])) v++;
    if (v < l && s[v] == '-') { neg = true; v++; } else if (v < l && s[v] == '+') v++;
    const uint8_t c0 = v < l ? s[v] : 0, c1 = v + 1 < l ? s[v + 1] : 0;
    if (!((c0 >= '0' && c0 <= '9') || (c0 == '.' && c1 >= '0' && c1 <= '9'))) return 1;
    if (c0 == '0' && (c1 == 'x' || c1 == 'X')) return 1;
    uint64_t n = 0; int nd = 0, fd = 0; bool seen_nz = false;
    for (; v < l && s[v] >= '0' && s[v] <= '9'; v++) { if (s[v] != '0' || seen_nz) { seen_nz = true; if (++nd > 15) return 1; n = n * 10 + (s[v] - '0'); } }
    if (v < l && s[v] == '.') {
        v++;
        for (; v < l && s[v] >= '0' && s[v] <= '9'; v++) { fd++; if (s[v] != '0' || seen_nz) { seen_nz = true; if (++nd > 15) return 1; n = n * 10 + (s[v] - '0'); } }
    }
    int ex = 0;
    if (v < l && (s[v] == 'e' || s[v] == 'E')) {
        uint32_t w = v + 1; bool eneg = false;
        if (w < l && (s[w] == '-' || s[w] == '+')) { eneg = s[w] == '-'; w++; }
        if (w < l && s[w] >= '0' && s[w] <= '9') {                                // (an 'e' without digits is not part of the number)
            int e = 0;
            for (; w < l && s[w] >= '0' && s[w] <= '9'; w++) if (e < 100000) e = e * 10 + (s[w] - '0');
            ex = eneg ? -e : e; v = w;
        }
    }
    double d;
    if (n == 0) d = 0.0;
    else {
        int e10 = ex - fd;
        while (e10 > 22 && nd < 15) { n *= 10; nd++; e10--; }                     // (digits to spare: the significand stays below 10^15 < 2^53)
        if (e10 > 22 || e10 < -22) return 1;
        double p10 = 1.0; for (int i = 0, m = e10 < 0 ? -e10 : e10; i < m; i++) p10 *= 10.0;      // exact up to 10^22
        d = e10 < 0 ? __ddiv_rn((double)n, p10) : __dmul_rn((double)n, p10);
    }
    *val = neg ? -d : d; *end = v;
    return 0;
}

// ---- the interval of a VCF line as tabix sees it (tbx_parse1, htslib tbx.c:96-312): what region queries on text are tested against --------
// strtoll(s, &e, base) over u[p, e): white space, sign, base 0 = 0x.. hex / 0.. octal / decimal; saturating.  *adv = characters consumed.
__device__ __forceinline__ long long vcf_strtoll(const uint8_t *u, uint32_t p, uint32_t e, int base, uint32_t *adv) {
    uint32_t v = p;
    while (v < e && vcf_isspace(u[v])) v++;
    bool neg = false;
    if (v < e && (u[v] == '-' || u[v] == '+')) { neg = u[v] == '-'; v++; }
    int b = base;
    if (b == 0) {
        if (v + 2 < e + 0 && u[v] == '0' && (u[v + 1] == 'x' || u[v + 1] == 'X') && v + 2 < e && ((u[v + 2] >= '0' && u[v + 2] <= '9') || ((u[v + 2] | 32) >= 'a' && (u[v + 2] | 32) <= 'f'))) { b = 16; v += 2; }
        else if (v < e && u[v] == '0') b = 8;
        else b = 10;
    }
    unsigned long long n = 0; bool any = false, over = false; const unsigned long long lim = neg ? 0x8000000000000000ull : 0x7fffffffffffffffull;
    for (; v < e; v++) {
        const uint8_t c = u[v]; int d = (c >= '0' && c <= '9') ? c - '0' : ((c | 32) >= 'a' && (c | 32) <= 'z') ? (c | 32) - 'a' + 10 : 99;
        if (d >= b) break;
        any = true;
        if (over) continue;
        if (n > (lim - (unsigned)d) / (unsigned)b) { over = true; n = lim; } else n = n * (unsigned)b + (unsigned)d;
    }
    if (adv) *adv = any ? v - p : 0;
    return neg ? (long long)(0ull - n) : (long long)n;
}
__device__ __forceinline__ uint32_t vcf_find(const uint8_t *u, uint32_t a, uint32_t b, const char *pat, uint32_t pl) {      // strstr over u[a, b)
    for (uint32_t i = a; i + pl <= b; i++) { uint32_t k = 0; while (k < pl && u[i + k] == (uint8_t)pat[k]) k++; if (k == pl) return i; }
    return 0xffffffffu;
}
__device__ __forceinline__ bool vcf_svlen_alt(const uint8_t *u, uint32_t a, uint32_t l) {                                    // svlen_on_ref_for_vcf_alt
    if (l < 5 || u[a] != '<') return false;
    if (u[a + 4] != '>' && u[a + 4] != ':') return false;
    const uint8_t c1 = u[a + 1], c2 = u[a + 2], c3 = u[a + 3];
    if (!((c1 == 'C' && c2 == 'N' && c3 == 'V') || (c1 == 'D' && c2 == 'E' && c3 == 'L') || (c1 == 'D' && c2 == 'U' && c3 == 'P') || (c1 == 'I' && c2 == 'N' && c3 == 'V'))) return false;
    return u[a + l - 1] == '>';
}
// end of the line's interval; beg = POS - 1 clamped at 0; [r0,r1) REF, [a0,a1) ALT, [i0,i1) INFO, [x0,x1) FORMAT + samples (x0 == x1: none)
// found = true: pe_in / ps_in are where the values of the first INFO field that starts with "END=" / "SVLEN=" begin (0xffffffff: none), found by
// the caller (the wave kernel sees every field anyway); false: the INFO string is walked here
__device__ __forceinline__ long long vcf_tabix_end(const uint8_t *u, long long beg, uint32_t r0, uint32_t r1, uint32_t a0, uint32_t a1, uint32_t i0, uint32_t i1, uint32_t x0, uint32_t x1,
                                                   const bool found = false, const uint32_t pe_in = 0xffffffffu, const uint32_t ps_in = 0xffffffffu) {
    if (beg < 0) beg = 0;
    const long long reflen = (long long)(r1 - r0);
    long long end = 1, svlen = 0, fmtlen = 0;
    if (reflen > 0) end = beg + reflen;
    // ALT alleles numbered from 1; which of the first 64 are <DEL>/<DUP>/<CNV>/<INV> (more symbolic alleles than that: ignored)
    unsigned long long svmask = 0; int alcnt = 1; bool getlen = false;
    for (uint32_t s0 = a0;;) {
        uint32_t t = s0; while (t < a1 && u[t] != ',') t++;
        ++alcnt;
        if (vcf_svlen_alt(u, s0, t - s0)) { if (alcnt - 1 < 64) svmask |= 1ull << (alcnt - 1); }
        else if ((t - s0 == 3 && u[s0] == '<' && u[s0 + 1] == '*' && u[s0 + 2] == '>') || (t - s0 == 9 && vcf_find(u, s0, t, "<NON_REF>", 9) == s0)) getlen = true;
        if (t >= a1 || alcnt >= 65536) break;
        s0 = t + 1;
    }
    // strstr(info, "END=") at the start of INFO, else strstr(info, ";END=") (tbx.c:213-233) = the first ';'-separated field that starts
    // with "END="; the same for "SVLEN=".  One walk over the fields, the ';' found eight bytes at a time.  SVLEN can only matter when an
    // allele is symbolic or REF is empty (its contribution is otherwise 1 <= reflen).
    const bool want_sv = svmask != 0 || reflen < 1;
    uint32_t pe = found ? pe_in : 0xffffffffu, ps = (found && want_sv) ? ps_in : 0xffffffffu;
    for (uint32_t f = i0; !found && f < i1;) {
        if (pe == 0xffffffffu && f + 4 <= i1 && u[f] == 'E' && u[f + 1] == 'N' && u[f + 2] == 'D' && u[f + 3] == '=') pe = f + 4;
        else if (want_sv && ps == 0xffffffffu && f + 6 <= i1 && u[f] == 'S' && u[f + 1] == 'V' && u[f + 2] == 'L' && u[f + 3] == 'E' && u[f + 4] == 'N' && u[f + 5] == '=') ps = f + 6;
        if (pe != 0xffffffffu && (ps != 0xffffffffu || !want_sv)) break;
        uint32_t t = f;
        for (; t + 8 <= i1; t += 8) {
            uint64_t v; __builtin_memcpy(&v, u + t, 8); v ^= 0x3b3b3b3b3b3b3b3bull;
            const uint64_t z = (v - 0x0101010101010101ull) & ~v & 0x8080808080808080ull;
            if (z) { t += (uint32_t)(__builtin_ctzll(z) >> 3); goto semi; }
        }
        while (t < i1 && u[t] != ';') t++;
    semi:
        f = t + 1;
    }
    uint32_t s = pe;
    if (s != 0xffffffffu && !(s < i1 && u[s] == '.')) { const long long v = vcf_strtoll(u, s, i1, 0, nullptr); if (v > beg) end = v; }
    s = ps;
    for (int d = 1; s != 0xffffffffu && d < alcnt; ++d) {
        uint32_t t = s; while (t < i1 && u[t] != ',') t++;
        long long tmp = 1;
        if (d < 64 && ((svmask >> d) & 1ull)) { tmp = vcf_strtoll(u, s, i1, 10, nullptr); if (tmp < 0) tmp = -tmp; }
        if (svlen < tmp) svlen = tmp;
        s = t < i1 ? t + 1 : 0xffffffffu;
    }
    if (getlen && x1 > x0) {                                                    // FORMAT/LEN of the samples (gVCF blocks)
        uint32_t fq = x0; while (fq < x1 && u[fq] != '\t') fq++;
        int lenpos = -1, pos = 0;
        for (uint32_t a = x0;; pos++) {
            uint32_t b2 = a; while (b2 < fq && u[b2] != ':') b2++;
            if (b2 - a == 3 && u[a] == 'L' && u[a + 1] == 'E' && u[a + 2] == 'N') { lenpos = pos; break; }
            if (b2 >= fq) break;
            a = b2 + 1;
        }
        for (uint32_t sm = fq < x1 ? fq + 1 : x1; lenpos >= 0 && sm <= x1;) {
            uint32_t se = sm; while (se < x1 && u[se] != '\t') se++;
            uint32_t f = sm; long long tmp = 0;
            for (int d = 0; d <= lenpos; ++d) {
                if (d == lenpos) { tmp = vcf_strtoll(u, f, se, 10, nullptr); break; }
                uint32_t c = f; while (c < se && u[c] != ':') c++;
                if (c >= se) break;
                f = c + 1;
            }
            if (fmtlen < tmp) fmtlen = tmp;
            if (se >= x1) break;
            sm = se + 1;
        }
    }
    long long m = reflen; if (svlen > m) m = svlen; if (fmtlen > m) m = fmtlen;
    if (end < beg + m) end = beg + m;
    return end;
}

// ---- tabix_index for the line formats other than VCF (tbx_parse1 with the generic / UCSC / SAM presets, tbx.c:96-178, 300-312) --------------------
// One lane per line: flag 1 = a meta line (first byte == meta) -- skipped by the indexer --, 2 = the line does not parse, 0 = interval
// [beg, end) on the sequence named by u[name_off, name_off + name_len); same = the name equals the previous line's (so that the host looks at
// a name only where it changes).
struct TbxLine { uint32_t name_off, name_len; long long beg, end; uint32_t flag, same; };
struct TbxConf { int32_t preset, sc, bc, ec, meta, skip; };
extern "C" __global__ void __launch_bounds__(256)
tabix_intervals(const uint8_t *__restrict__ u, const uint32_t *__restrict__ line_off, int64_t nlines, uint64_t text_end, int32_t last_open, TbxConf cf, TbxLine *__restrict__ out) {
    const int64_t li = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= nlines) return;
    const uint32_t l0 = line_off[li];
    uint32_t l1 = (li + 1 < nlines || !last_open) ? line_off[li + 1] - 1u : (uint32_t)text_end;
    if (l1 > l0 && u[l1 - 1] == '\r') l1--;                                      // bgzf_getline drops a CR in front of the newline (bgzf.c:2328)
    TbxLine r; r.name_off = l0; r.name_len = 0; r.beg = -1; r.end = -1; r.flag = 0; r.same = 0;
    if (l1 > l0 && u[l0] == (uint8_t)cf.meta) { r.flag = 1; out[li] = r; return; }
    bool have_name = false, bad = false;
    int id = 1; uint32_t b = l0;
    for (uint32_t i = l0; i <= l1 && !bad; i++) {
        if (i < l1 && u[i] != '\t' && u[i] != 0) continue;                       // (a NUL ends the line for tbx_parse1's C strings)
        if (id == cf.sc) { r.name_off = b; r.name_len = i - b; have_name = true; }
        else if (id == cf.bc) {
            uint32_t adv = 0; r.beg = vcf_strtoll(u, b, i, 0, &adv);
            if (cf.bc <= cf.ec) r.end = r.beg;
            if (adv == 0) { bad = true; break; }
            if (!(cf.preset & 0x10000)) --r.beg; else if (cf.bc <= cf.ec) ++r.end;
            if (r.beg < 0) r.beg = 0;
            if (r.end < 1) r.end = 1;
        } else if ((cf.preset & 0xffff) == 0) {
            if (id == cf.ec) { uint32_t adv = 0; r.end = vcf_strtoll(u, b, i, 0, &adv); if (adv == 0) { bad = true; break; } }
        } else if ((cf.preset & 0xffff) == 1 && id == 6) {                       // SAM: reference length of the CIGAR
            long long l = 0;
            for (uint32_t s = b; s < i;) {
                uint32_t adv = 0; const long long x = vcf_strtoll(u, s, i, 10, &adv);
                const uint32_t t = s + adv; const uint8_t op = t < i ? (u[t] & 0xdf) : 0;
                if (op == 'M' || op == 'D' || op == 'N') l += (int)x;
                s = t + 1;
            }
            if (l == 0) l = 1;
            r.end = r.beg + l;
        }
        if (i < l1 && u[i] == 0) break;
        b = i + 1; ++id;
    }
    if (bad || !have_name || r.beg < 0 || r.end < 0) r.flag = 2;
    if (r.flag == 0 && li > 0) {
        // the previous line's name: parsed again (cheap: the name column is among the first)
        const uint32_t p0 = line_off[li - 1]; uint32_t p1 = line_off[li] - 1u; if (p1 > p0 && u[p1 - 1] == '\r') p1--;
        int pid = 1; uint32_t pb = p0, pn0 = 0, pn1 = 0; bool got = false;
        if (!(p1 > p0 && u[p0] == (uint8_t)cf.meta))
            for (uint32_t i = p0; i <= p1; i++) { if (i < p1 && u[i] != '\t' && u[i] != 0) continue; if (pid == cf.sc) { pn0 = pb; pn1 = i; got = true; break; } if (i < p1 && u[i] == 0) break; pb = i + 1; ++pid; }
        if (got && pn1 - pn0 == r.name_len) { uint32_t k = 0; while (k < r.name_len && u[pn0 + k] == u[r.name_off + k]) k++; r.same = k == r.name_len ? 1u : 0u; }
    }
    out[li] = r;
}

// ---- read_bed rows for the overlap join (src/interval_udf.c:141-157 is_meta_bed_line / count_tab_fields, 159-181 get_field_span, 127-139
// parse_int64_span_local, 330-342 next_bed_line) ---------------------------------------------------------------------------------------------
// One lane per line.  flag 1 = not a row (empty, '#', "track", "browser"), 2 = fewer than 3 tab-delimited fields (read_bed raises), else bit 2 /
// bit 3 = start / end is NULL (the field is empty or strtoll does not consume all of it); chrom = u[name_off, +name_len); same as in TbxLine.
extern "C" __global__ void __launch_bounds__(256)
bed_intervals(const uint8_t *__restrict__ u, const uint32_t *__restrict__ line_off, int64_t nlines, uint64_t text_end, int32_t last_open, TbxLine *__restrict__ out) {
    const int64_t li = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= nlines) return;
    const uint32_t l0 = line_off[li];
    uint32_t l1 = (li + 1 < nlines || !last_open) ? line_off[li + 1] - 1u : (uint32_t)text_end;
    if (l1 > l0 && u[l1 - 1] == '\r') l1--;
    TbxLine r; r.name_off = l0; r.name_len = 0; r.beg = 0; r.end = 0; r.flag = 0; r.same = 0;
    auto starts = [&](uint32_t a, uint32_t b, const char *w, uint32_t n) { if (b - a < n) return false; for (uint32_t k = 0; k < n; k++) if (u[a + k] != (uint8_t)w[k]) return false; return true; };
    if (l1 == l0 || u[l0] == 0 || u[l0] == '#' || starts(l0, l1, "track", 5) || starts(l0, l1, "browser", 7)) { r.flag = 1; out[li] = r; return; }
    int id = 0; uint32_t b = l0;
    for (uint32_t i = l0; i <= l1; i++) {
        const bool stop = i == l1 || u[i] == 0;                                     // (the reference walks a C string)
        if (!stop && u[i] != '\t') continue;
        if (id == 0) { r.name_off = b; r.name_len = i - b; }
        else if (id <= 2) {
            uint32_t adv = 0; const long long v = vcf_strtoll(u, b, i, 10, &adv);
            const bool ok = i > b && adv == i - b;
            if (id == 1) { r.beg = v; if (!ok) r.flag |= 4u; } else { r.end = v; if (!ok) r.flag |= 8u; }
        }
        ++id; b = i + 1;
        if (stop) break;
    }
    if (id < 3) r.flag = 2;
    else if (li > 0) {
        const uint32_t p0 = line_off[li - 1]; uint32_t p1 = line_off[li] - 1u;
        uint32_t k = 0; while (k < r.name_len && p0 + k < p1 && u[p0 + k] == u[l0 + k]) k++;
        r.same = (k == r.name_len && p0 + k < p1 && u[p0 + k] == '\t') ? 1u : 0u;
    }
    out[li] = r;
}

// One INFO field u[key, fend) -- the text between two ';' --: key[=value] (vcf_parse_info, vcf.c:3744-3985).  Returns 0 for an empty key (the
// field is skipped and does not count), 1 for a field that was measured / written, 2 for a field outside the projection (VcfArgs::info_keep).  QUIET: a measure pass that leaves no record of undefined names (the write pass of the wave
// kernel measures again to place its fields).
// DEFER: a string value longer than VCF_DEFER_MIN is given its place but not copied; (*d_src, *d_dst, *d_len) tell the caller what is left to do
// (the wave kernel copies such a value with all its lanes).
#define VCF_DEFER_MIN 96u
// The key of a field with its first 32 bytes in registers (REGKEY: u is the stream itself, which is padded, so the four 8-byte loads -- in
// flight together -- may run past the field): the '=' by a zero-byte test on the words, the hash and the comparison with the table's name
// from the registers.  A walk byte by byte is a chain of dependent loads, forty of them for a 20-character key; this is four chains of
// one.  Returns false when the key is longer than the 32 bytes (the caller walks).
__device__ __forceinline__ bool vcf_key_regs(const VcfDictDev &d, const uint8_t *u, const uint32_t key, const uint32_t fend, uint32_t *kend_out, int *k_out) {
    uint64_t w[4];
    __builtin_memcpy(w, u + key, 32);
    const uint32_t flen = fend - key;
    uint32_t pe = 32;
#pragma unroll
    for (int q = 3; q >= 0; q--) {
        const uint64_t x = w[q] ^ 0x3d3d3d3d3d3d3d3dull, z = (x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull;      // (the lowest flagged byte is exact)
        if (z) pe = (uint32_t)(8 * q) + (uint32_t)(__builtin_ctzll(z) >> 3);
    }
    if (pe >= 32 && flen > 32) return false;
    const uint32_t L = pe < flen ? pe : flen;
    *kend_out = key + L; *k_out = -1;
    if (L == 0 || d.n <= 0) return true;
    uint64_t msk[4];
#pragma unroll
    for (int q = 0; q < 4; q++) { const int nb = (int)L - 8 * q; msk[q] = nb >= 8 ? ~0ull : nb <= 0 ? 0ull : ((1ull << (8 * nb)) - 1ull); w[q] &= msk[q]; }
    uint32_t h = 2166136261u;
    for (uint32_t i = 0; i < L; i++) h = (h ^ (uint32_t)((w[i >> 3] >> (8 * (i & 7))) & 0xffu)) * 16777619u;
    for (h &= d.hmask;; h = (h + 1) & d.hmask) {
        const uint32_t e = d.hash[h];
        if (!e) return true;
        const uint32_t o0 = d.off[e - 1], o1 = d.off[e];
        if (o1 - o0 != L) continue;
        uint64_t m[4];
        __builtin_memcpy(m, d.bytes + o0, 32);                                     // (the table's bytes have 64 bytes of slack behind them)
        if ((((m[0] & msk[0]) ^ w[0]) | ((m[1] & msk[1]) ^ w[1]) | ((m[2] & msk[2]) ^ w[2]) | ((m[3] & msk[3]) ^ w[3])) == 0) { *k_out = (int)(e - 1); return true; }
    }
}
template <bool WRITE, bool QUIET, bool DEFER = false, bool REGKEY = false>
__device__ __forceinline__ int vcf_info_field(const VcfArgs &a, const int64_t li, const uint8_t *u, const uint32_t bias, const uint32_t key, const uint32_t fend, VcfSink<WRITE> &o,
                                              uint32_t *d_src = nullptr, uint32_t *d_dst = nullptr, uint32_t *d_len = nullptr) {
    uint32_t kend = key; int k = -1;
    if (!REGKEY || !vcf_key_regs(a.ids, u, key, fend, &kend, &k)) {
        kend = key; while (kend < fend && u[kend] != '=') kend++;
        if (kend > key) k = vcf_dict_find(a.ids, u + key, kend - key);
    }
    if (kend == key) return 0;
    const uint32_t val = kend < fend ? kend + 1 : 0xffffffffu, end = fend;
    int ht = 3; int32_t id = 0;
    if (k < 0 || a.ids.ityp[k] == 15) { if (!WRITE && !QUIET) { const uint32_t q = atomicAdd(&a.counters[0], 1u); if (q < a.undef_cap) a.undef[q] = {(uint32_t)li, key + bias, kend - key, 2u}; } }
    else { ht = a.ids.ityp[k]; id = a.ids.id[k]; }
    if (a.info_keep && !((a.info_keep[(uint32_t)id >> 5] >> ((uint32_t)id & 31u)) & 1u)) return 2;      // (a name without a definition has id 0 here and a definition of its own in the next round)
    o.key(id);
    if (val == 0xffffffffu) o.b(0x00);
    else if (ht == 0 || ht == 3) {
        if (DEFER && WRITE && end - val > VCF_DEFER_MIN) { o.size(end - val, 7); *d_src = val; *d_dst = o.n; *d_len = end - val; o.n += end - val; }
        else o.vchar(u + val, end - val);
    } else {
        uint32_t n_val = 1; for (uint32_t t = val; t < end; t++) n_val += u[t] == ',';
        if (ht == 1 && n_val == 1) o.b(0x13); else o.size(n_val, ht == 1 ? 3 : 5);
        uint32_t t = val;
        for (uint32_t i = 0; i < n_val; i++, t++) {
            uint32_t te = t, w;
            if (ht == 1) {                                                               // hts_str2int, 64 bits
                bool neg = false, over = false; uint64_t n = 0, limit = (1ull << 63) - 1;
                if (te < end && u[te] == '-') { limit++; neg = true; te++; } else if (te < end && u[te] == '+') te++;
                for (; te < end && u[te] >= '0' && u[te] <= '9'; te++) { const uint32_t d = u[te] - '0'; if (over) continue; if (n < limit / 10 || (n == limit / 10 && d <= limit % 10)) n = n * 10 + d; else over = true; }
                const int64_t v = neg ? (int64_t)(0 - n) : (int64_t)n;
                w = (te == t || over || v < -2147483640ll || v > 2147483647ll) ? 0x80000000u : (uint32_t)(int32_t)v;
            } else {
                uint32_t tok_end = t; while (tok_end < end && u[tok_end] != ',') tok_end++;
                double d; uint32_t e;
                if (vcf_str2dbl_fast(u + t, tok_end - t, &d, &e) == 0) { w = __float_as_uint(__double2float_rn(d)); te = t + e; }
                else { w = 0x7F800001u; te = tok_end; if (WRITE) { const uint32_t q = atomicAdd(&a.counters[1], 1u); if (q < a.patch_cap) a.patch[q] = {t + bias, tok_end - t, a.rec_off[li] + o.n, 1u}; } }
            }
            o.w32(w);
            for (t = te; t < end && u[t] != ','; t++) {}
        }
    }
    return 1;
}

// ---- tokens the host has to look at (undefined names, numbers in strtod's wider forms): gathered into one buffer for one copy each way ----------
// ent = VcfUndef / VcfPatch records (four words each); pos_w / len_w = which words hold the token's position and length; tok_off = where each
// token goes in `out` (an exclusive scan of the lengths, made by the host from the same records)
extern "C" __global__ void __launch_bounds__(256)
vcf_gather_tokens(const uint8_t *__restrict__ u, const uint32_t *__restrict__ ent, int pos_w, int len_w, const uint32_t *__restrict__ tok_off, uint32_t n, uint8_t *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = ent[4u * i + (uint32_t)pos_w], l = ent[4u * i + (uint32_t)len_w], o = tok_off[i];
    for (uint32_t k = 0; k < l; k++) out[o + k] = u[p + k];
}
// the converted numbers go back: word i to out[patch[i].dst .. +4) (records are byte-packed: no alignment)
extern "C" __global__ void __launch_bounds__(256)
vcf_scatter_words(uint8_t *__restrict__ out, const VcfPatch *__restrict__ patch, const uint32_t *__restrict__ bits, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t d = patch[i].dst, w = bits[i];
    out[d] = (uint8_t)w; out[d + 1] = (uint8_t)(w >> 8); out[d + 2] = (uint8_t)(w >> 16); out[d + 3] = (uint8_t)(w >> 24);
}

#define VCF_LDS_BYTES 40960u
#define VCF_ENC_THREADS 64
#define VCF_MAXF 16                                           // FORMAT keys of a line in the lane-per-line encoder (more: the batch goes to the wave encoder)
#define VCF_MAXF_WAVE 255                                     // htslib's MAX_N_FMT (vcf.c:3134)
struct VcfFmtLds { int32_t key[VCF_MAXF_WAVE + 1]; uint32_t mx_l[VCF_MAXF_WAVE + 1], mx_m[VCF_MAXF_WAVE + 1], mx_g[VCF_MAXF_WAVE + 1], fsz[VCF_MAXF_WAVE + 1], fat[VCF_MAXF_WAVE + 1]; uint8_t ht[VCF_MAXF_WAVE + 1], flg[VCF_MAXF_WAVE + 1]; };
#define VCF_WSEP 2048u                                        // ';' positions a wave collects before it parses the fields between them (LDS, 8 KB)
struct VcfFmtLds;
template <bool WRITE, bool WAVE, bool SMP> __device__ __forceinline__ void vcf_encode_line(const VcfArgs &a, const int64_t li, const uint8_t *u, const uint32_t bias, uint32_t *sep, VcfFmtLds *fl);

// ---- a wave per line (lines too long for 64 of them to share the LDS staging: gnomAD-style INFO of hundreds of keys) ----------------------------
// first position in [from, to) that holds ch, else `to`: 16 bytes per lane, a ballot per KiB (the stream is padded, so whole 16-byte words are read)
__device__ __forceinline__ uint32_t vcf_eq_mask16(const uint8_t *u, uint32_t p, uint32_t from, uint32_t to, uint32_t pat) {       // bit k: u[p + k] == ch, p + k in [from, to)
    const uint4 v = *(const uint4 *)(u + p);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t m = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t x = w[q] ^ pat;
        const uint32_t z = ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu);               // 0x80 exactly where the byte is zero
        m |= (((z >> 7) & 1u) | ((z >> 14) & 2u) | ((z >> 21) & 4u) | ((z >> 28) & 8u)) << (4 * q);
    }
    if (p < from) m &= 0xffffu << (from - p);
    if (p + 16 > to) m &= to > p ? (0xffffu >> (p + 16 - to)) : 0u;
    return m;
}
__device__ __forceinline__ uint32_t vcf_wave_find(const uint8_t *u, uint32_t from, uint32_t to, uint8_t ch) {
    const uint32_t lane = threadIdx.x & 63u, pat = 0x01010101u * ch;
    for (uint32_t base = from & ~15u; base < to; base += 1024u) {
        const uint32_t p = base + lane * 16u;
        const uint32_t m = p < to ? vcf_eq_mask16(u, p, from, to, pat) : 0u;
        const unsigned long long hit = __ballot(m != 0);
        if (hit) { const int l = __builtin_ctzll(hit); const uint32_t ml = (uint32_t)__shfl((int)m, l); return base + (uint32_t)l * 16u + (uint32_t)__builtin_ctz(ml); }
    }
    return to;
}
__device__ __forceinline__ uint32_t vcf_wave_excl_scan(uint32_t v, uint32_t *total) {
    const int lane = threadIdx.x & 63;
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)x, d); if (lane >= d) x += y; }
    *total = (uint32_t)__shfl((int)x, 63);
    return x - v;
}
// The line stays where it lies (HBM / L2): staging it in LDS was measured and bought nothing -- the parse is a chain of dependent byte reads
// either way, and what hides their latency is the number of waves per CU, which 8 KB of LDS per wave leaves at twenty.
// SMP: the file has sample columns.  The sites-only instantiations carry none of the FORMAT code, whose per-key arrays would otherwise take
// the kernel to 300 registers and one wave per SIMD.
template <bool WRITE, bool SMP>
__global__ void __launch_bounds__(64) vcf_encode_wave(VcfArgs a) {
    __shared__ uint32_t sep[VCF_WSEP];
    __shared__ uint32_t fl_words[SMP ? (sizeof(VcfFmtLds) + 3) / 4 : 1];                             // (the sites-only kernels do not pay for the tables)
    VcfFmtLds *fl = (VcfFmtLds *)fl_words;
    const int64_t li = blockIdx.x;
    if (li < a.nlines) vcf_encode_line<WRITE, true, SMP>(a, li, a.u, 0u, sep, fl);
}

// The lines of a workgroup are consecutive in the text: their span is staged in LDS with coalesced 16-byte loads and parsed from there (a
// lane walks its line byte by byte); a span that does not fit is parsed from HBM.  Two separate calls, so that the LDS copy is reached
// through LDS instructions: a pointer that may be either kind becomes a flat access, and flat accesses to LDS fault on this system.
template <bool WRITE, bool SMP>
__global__ void __launch_bounds__(VCF_ENC_THREADS) vcf_encode(VcfArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t vcf_lds[];
    const int64_t li0 = (int64_t)blockIdx.x * blockDim.x, li = li0 + threadIdx.x;
    const int64_t liN = li0 + blockDim.x < a.nlines ? li0 + blockDim.x : a.nlines;
    const uint32_t s0 = a.line_off[li0] & ~15u;
    const uint32_t s1 = (liN == a.nlines && a.last_open) ? (uint32_t)a.text_end : a.line_off[liN];
    const bool staged = a.lds_budget >= 16u && s1 - s0 <= a.lds_budget - 16u;
    if (staged) {
        for (uint32_t q = threadIdx.x * 16u; s0 + q < s1; q += blockDim.x * 16u) *(uint4 *)(vcf_lds + q) = *(const uint4 *)(a.u + s0 + q);   // (the stream is padded: reading up to 15 bytes past s1 is safe)
        __syncthreads();
        if (li < a.nlines) vcf_encode_line<WRITE, false, SMP>(a, li, vcf_lds, s0, nullptr, nullptr);
    } else if (li < a.nlines) vcf_encode_line<WRITE, false, SMP>(a, li, a.u, 0u, nullptr, nullptr);
}

// every position below is relative to `bias` (the start of the staged span, or 0): u[] is either the LDS copy or the stream itself
// WAVE: the 64 lanes of a wave share one line.  What is short (the first seven columns, the FORMAT keys) every lane computes alike -- stores
// then hit one address, and `lead` keeps the records of undefined names and patches single --; what is long is shared out: the searches for
// the line's NUL and INFO's end (vcf_wave_find), and INFO, whose ';' are collected 16 bytes per lane and whose fields are then parsed one per
// lane, measured, placed by a scan of the sizes and written.
// ---- the sample columns (vcf_parse_format vcf.c:3686-3742; steps 3137-3684), one sample at a time: the serial encoder walks them in a row, the wave encoder a sample per lane ----
// max3 of one sample column u[r, e) (e: its tab, or the line's end): the widths of its fields into mx_l / mx_m / mx_g (ATOMIC: LDS tables shared
// by the lanes of a wave).  false: more fields than FORMAT names ("Incorrect number of FORMAT fields").
template <bool ATOMIC>
__device__ __forceinline__ bool vcf_fmt_widths(const uint8_t *u, uint32_t r, const uint32_t e, const int n_fmt, const uint8_t *flg, uint32_t *mx_l, uint32_t *mx_m, uint32_t *mx_g) {
    auto up = [&](uint32_t *p, uint32_t v) { if (ATOMIC) { if (*p < v) atomicMax(p, v); } else if (*p < v) *p = v; };
    int j = 0; uint32_t r_start = r, m = 1, g = 1;
    for (;;) {
        while (r < e && u[r] != ',' && u[r] != '/' && u[r] != ':' && u[r] != '|') r++;
        const uint8_t ch = r < e ? u[r] : 0;
        if (ch == ',') m++;
        else if (ch == '|' || ch == '/') { if (flg[j] & 1) g++; }
        else {
            const uint32_t l = r - r_start; r_start = r;                          // (from the second field on the ':' in front counts, as in htslib)
            up(mx_m + j, m); up(mx_l + j, l); if (flg[j] & 1) up(mx_g + j, g);
            m = g = 1;
            if (ch == ':') { j++; if (j >= n_fmt) return false; }
            else break;
        }
        if (r >= e) break;
        r++;
    }
    return true;
}
// fill5 of one sample column u[t, end) of a line that ends at lend: every field validated (measure pass) and stored at its place fat[field] + fsz[field] * m (write pass).
// Returns true when the column is an error.
template <bool WRITE>
__device__ __forceinline__ bool vcf_fmt_fill_sample(const VcfArgs &a, const int64_t li, const uint8_t *u, const uint32_t bias, uint32_t t, const uint32_t end, const uint32_t lend, const uint32_t m, const int n_fmt,
                                                    const uint8_t *ht, const uint8_t *flg, const uint32_t *fsz, const uint32_t *fat, uint8_t *op) {
    bool bad = false;
    int j = 0;
    while (t < lend) {                                                            // (lend: the line's end -- the walk looks at a field behind a ':' even when the column ends there, unless the line does too)
        const int z = j++;
        const uint32_t at = fat[z] + fsz[z] * m;
        auto put32 = [&](uint32_t idx, uint32_t v) { if (WRITE) __builtin_memcpy(op + at + 4 * idx, &v, 4); };
        auto ch = [&](uint32_t p) -> uint8_t { return p < end ? u[p] : (uint8_t)0; };                                // a sample column is a C string (end: its tab, or the line's end)
        if ((flg[z] & 2) || (WRITE && (flg[z] & 4))) { while (ch(t) != ':' && ch(t)) t++; }                 // a repeated key; a key outside the projection (the measure pass has validated its values)
        else if (ht[z] == 3 && (flg[z] & 1)) {                                        // GT: ([/|])?val([/|]val)*, val = digits or '.'
            uint32_t is_phased = 0, maxv = 0, x0 = 0; bool unreadable = false; int l = 0, ploidy = 0, anyunphased = 0, prfx = 0, unknown1 = 0;
            if (a.v44 && (ch(t) == '|' || ch(t) == '/')) { is_phased = ch(t) == '|'; t++; prfx = 1; }
            for (;; ++t) {
                ploidy++;
                uint32_t xv;
                if (ch(t) == '.') { ++t; xv = is_phased; if (l == 0) unknown1 = 1; }
                else {
                    const uint32_t tt = t; uint64_t n = 0;
                    if (ch(t) == '+') t++;
                    while (ch(t) >= '0' && ch(t) <= '9') n = n * 10 + (uint64_t)(ch(t++) - '0');
                    const uint32_t val = (uint32_t)n;
                    unreadable |= tt == t;
                    if (maxv < val) maxv = val;
                    xv = (val + 1) << 1 | is_phased;
                }
                if (l == 0) x0 = xv; else put32((uint32_t)l, xv);
                l++;
                anyunphased |= (ploidy != 1) && !is_phased;
                is_phased = ch(t) == '|';
                if (ch(t) != '|' && ch(t) != '/') break;
            }
            if (!prfx) { if (ploidy == 1) { if (!unknown1) x0 |= 1; } else x0 |= anyunphased ? 0u : 1u; }
            if (maxv > (0x7fffffffu >> 1) - 1 || unreadable) { bad = true; break; }
            put32(0, x0);
            for (; (uint32_t)l < fsz[z] >> 2; ++l) put32((uint32_t)l, 0x80000001u);
        } else if (ht[z] == 3) {
            uint32_t l = 0;
            for (; ch(t) != ':' && ch(t); ++t, ++l) if (WRITE) op[at + l] = u[t];
            if (WRITE) for (; l < fsz[z]; l++) op[at + l] = 0;
        } else if (ht[z] == 1) {
            uint32_t l = 0;
            for (;; ++t) {
                if (ch(t) == '.') { put32(l++, 0x80000000u); ++t; }
                else {
                    uint32_t te = t; bool neg = false, over = false; uint64_t n = 0, limit = (1ull << 63) - 1;
                    if (ch(te) == '-') { limit++; neg = true; te++; } else if (ch(te) == '+') te++;
                    for (; ch(te) >= '0' && ch(te) <= '9'; te++) { const uint32_t d = ch(te) - '0'; if (over) continue; if (n < limit / 10 || (n == limit / 10 && d <= limit % 10)) n = n * 10 + d; else over = true; }
                    const int64_t v = neg ? (int64_t)(0 - n) : (int64_t)n;
                    put32(l++, (te == t || over || v < -2147483640ll || v > 2147483647ll) ? 0x80000000u : (uint32_t)(int32_t)v);
                    t = te;
                }
                if (ch(t) != ',') break;
            }
            for (; l < fsz[z] >> 2; ++l) put32(l, 0x80000001u);
        } else {
            uint32_t l = 0;
            for (;; ++t) {
                const uint8_t c1 = ch(t + 1);
                if (ch(t) == '.' && !(c1 >= '0' && c1 <= '9')) { put32(l++, 0x7F800001u); ++t; }
                else {
                    uint32_t tok_end = t; while (ch(tok_end) && ch(tok_end) != ',' && ch(tok_end) != ':') tok_end++;
                    double d; uint32_t e;
                    if (tok_end == t) put32(l++, 0u);                               // an empty value: strtod converts nothing and returns 0.0
                    else if (vcf_str2dbl_fast(u + t, tok_end - t, &d, &e) == 0) { put32(l++, __float_as_uint(__double2float_rn(d))); t += e; }
                    else {
                        // strtod's forms: the host converts (write pass) and checks that the number ends where the token ends (measure pass)
                        if (!WRITE) { const uint32_t q = atomicAdd(&a.counters[0], 1u); if (q < a.undef_cap) a.undef[q] = {(uint32_t)li, t + bias, tok_end - t, 4u}; }
                        else { const uint32_t q = atomicAdd(&a.counters[1], 1u); if (q < a.patch_cap) a.patch[q] = {t + bias, tok_end - t, a.rec_off[li] + at + 4 * l, 2u}; }
                        put32(l++, 0u);
                        t = tok_end;
                    }
                }
                if (ch(t) != ',') break;
            }
            for (; l < fsz[z] >> 2; ++l) put32(l, 0x7F800002u);
        }
        if (ch(t) == 0) break;
        else if (ch(t) == ':') t++;
        else { bad = true; break; }                                                   // "Invalid character"
    }
    if (bad) return true;
    for (; j < n_fmt; ++j) {                                                          // trailing fields the sample leaves out
        if (flg[j] & 6) continue;
        const uint32_t at = fat[j] + fsz[j] * m;
        if (!WRITE) continue;
        if (ht[j] == 3 && !(flg[j] & 1)) { for (uint32_t l = 0; l < fsz[j]; l++) op[at + l] = l == 0 ? '.' : 0; }
        else for (uint32_t l = 0; l < fsz[j] >> 2; l++) { const uint32_t w = l == 0 ? (ht[j] == 2 ? 0x7F800001u : 0x80000000u) : (ht[j] == 2 ? 0x7F800002u : 0x80000001u); __builtin_memcpy(op + at + 4 * l, &w, 4); }
    }
    return false;
}

template <bool WRITE, bool WAVE, bool SMP>
__device__ __forceinline__ void vcf_encode_line(const VcfArgs &a, const int64_t li, const uint8_t *u, const uint32_t bias, uint32_t *sep, VcfFmtLds *fl) {
    const bool lead = !WAVE || (threadIdx.x & 63u) == 0;
    const uint32_t l0 = a.line_off[li] - bias;
    uint32_t l1 = ((li + 1 == a.nlines && a.last_open) ? (uint32_t)a.text_end : a.line_off[li + 1] - 1) - bias;
    if (l1 > l0 && u[l1 - 1] == '\r') l1--;                                                          // KS_SEP_LINE drops the carriage return
    if (WAVE) l1 = vcf_wave_find(u, l0, l1, 0);
    else { uint32_t e = l0; while (e < l1 && u[e]) e++; l1 = e; }                                    // the parser works on a C string
    VcfSink<WRITE> o; o.p = WRITE ? a.out + a.rec_off[li] : nullptr; o.n = 0;
    bool bad = false;
    // the eight mandatory columns (kstrtok on '\t': empty tokens count); a ninth one is not looked at (no samples: vcf_parse_format returns at once)
    uint32_t fs[8], fe[8]; int nf = 0;
    {
        uint32_t p = l0;
        for (;;) {
            fs[nf] = p;
            if (WAVE && nf == 7) p = vcf_wave_find(u, p, l1, '\t'); else while (p < l1 && u[p] != '\t') p++;
            fe[nf] = p; nf++;
            if (p >= l1 || nf == 8) break;
            p++;
        }
    }
    uint32_t w_pe = 0xffffffffu, w_ps = 0xffffffffu;                                                  // WAVE: where the values of END= / SVLEN= begin
    if (nf < 8) bad = true;
    int32_t rid = 0; int64_t pos = 0; uint32_t n_allele = 1, n_info = 0, n_info_all = 0, qbits = 0x7F800001u; int32_t rlen = 0;      // n_info: fields kept (the record's count); n_info_all: fields met (the limit's)
    if (!bad) {
        // CHROM
        const int k = vcf_dict_find(a.ctg, u + fs[0], fe[0] - fs[0]);
        if (k < 0) { if (!WRITE && lead) { const uint32_t q = atomicAdd(&a.counters[0], 1u); if (q < a.undef_cap) a.undef[q] = {(uint32_t)li, fs[0] + bias, fe[0] - fs[0], 0u}; } }
        else rid = a.ctg.id[k];
        // POS: hts_str2uint(.., 62 bits), the whole token
        {
            uint32_t v = fs[1]; uint64_t n = 0; bool over = false; const uint64_t limit = (1ull << 62) - 1;
            if (v < fe[1] && u[v] == '+') v++;
            for (; v < fe[1] && u[v] >= '0' && u[v] <= '9'; v++) { const uint32_t d = u[v] - '0'; if (over) continue; if (n < limit / 10 || (n == limit / 10 && d <= limit % 10)) n = n * 10 + d; else over = true; }
            if (over || v != fe[1]) bad = true;
            pos = (int64_t)n - 1;
        }
    }
    if (!bad) {
        o.n = 32;                                                                                    // lengths + core, written last
        // ID
        if (fe[2] - fs[2] == 1 && u[fs[2]] == '.') o.size(0, 7); else o.vchar(u + fs[2], fe[2] - fs[2]);
        // REF, ALT
        o.vchar(u + fs[3], fe[3] - fs[3]);
        // rlen: pos + rlen = the END the tabix iterator tests regions with (text carries no rlen; see vcf_tabix_end)
        if (WRITE && !WAVE) rlen = (int32_t)(vcf_tabix_end(u, pos, fs[3], fe[3], fs[4], fe[4], fs[7], fe[7], fe[7] < l1 ? fe[7] + 1 : l1, l1) - pos);
        if (!(fe[4] - fs[4] == 1 && u[fs[4]] == '.')) {
            uint32_t t = fs[4];
            for (uint32_t r = fs[4];; r++) {
                if (r == fe[4] || u[r] == ',') { if (n_allele == 65535) { bad = true; break; } o.vchar(u + t, r - t); t = r + 1; n_allele++; }
                if (r == fe[4]) break;
            }
        }
    }
    if (!bad) {
        // QUAL: (float)atof
        if (!(fe[5] - fs[5] == 1 && u[fs[5]] == '.')) {
            double d; uint32_t e;
            if (vcf_str2dbl_fast(u + fs[5], fe[5] - fs[5], &d, &e) == 0) { const float f = __double2float_rn(d); qbits = __float_as_uint(f); }
            else { qbits = 0; if (WRITE && lead) { const uint32_t q = atomicAdd(&a.counters[1], 1u); if (q < a.patch_cap) a.patch[q] = {fs[5] + bias, fe[5] - fs[5], a.rec_off[li] + 20u, 0u}; } }
        }
        // FILTER
        if (fe[6] - fs[6] == 1 && u[fs[6]] == '.') o.b(0x00);
        else {
            uint32_t e6 = fe[6]; if (e6 > fs[6] && u[e6 - 1] == ';') e6--;                             // one trailing ';' is dropped
            uint32_t n_flt = 1; for (uint32_t r = fs[6]; r < e6; r++) n_flt += u[r] == ';';
            o.size(n_flt, 3);
            uint32_t t = fs[6];
            for (uint32_t r = fs[6];; r++) if (r == e6 || u[r] == ';') {
                const int k = vcf_dict_find(a.ids, u + t, r - t);
                if (k < 0) { if (!WRITE && lead) { const uint32_t q = atomicAdd(&a.counters[0], 1u); if (q < a.undef_cap) a.undef[q] = {(uint32_t)li, t + bias, r - t, 1u}; } o.w32(0); }
                else o.w32((uint32_t)a.ids.id[k]);
                t = r + 1;
                if (r == e6) break;
            }
        }
        // INFO
        if (!(fe[7] - fs[7] == 1 && u[fs[7]] == '.')) {
            uint32_t e7 = fe[7]; if (e7 > fs[7] && u[e7 - 1] == ';') e7--;
            if (WAVE && WRITE && a.info_none) {                                                      // nothing of INFO is projected: the measure pass has seen every field (and left END= / SVLEN=)
                w_pe = a.endsv[2 * li]; w_ps = a.endsv[2 * li + 1];
            } else if (WAVE) {
                const uint32_t lane = threadIdx.x & 63u;
                uint32_t seg = fs[7], nsep = 0;                                                       // seg: where the first field of the collected stretch begins
                for (uint32_t base = fs[7] & ~15u;; base += 1024u) {
                    const bool last = base + 1024u >= e7;
                    if (base < e7) {                                                                  // the ';' of this KiB, in order, behind those already collected
                        const uint32_t p = base + lane * 16u;
                        uint32_t m = p < e7 ? vcf_eq_mask16(u, p, fs[7], e7, 0x3b3b3b3bu) : 0u, tot;
                        uint32_t at = nsep + vcf_wave_excl_scan((uint32_t)__builtin_popcount(m), &tot);
                        for (; m; m &= m - 1) sep[at++] = p + (uint32_t)__builtin_ctz(m);
                        nsep += tot;
                    }
                    if (last) { if (lane == 0) sep[nsep] = e7; nsep++; }                              // the last field ends where INFO does
                    if (!last && nsep + 1025u <= VCF_WSEP) continue;
                    __syncthreads();
                    for (uint32_t k0 = 0; k0 < nsep; k0 += 64u) {                                     // a field per lane
                        const uint32_t k = k0 + lane; const bool have = k < nsep;
                        const uint32_t key = have ? (k ? sep[k - 1] + 1u : seg) : 0u, fend = have ? sep[k] : 0u;
                        VcfSink<false> ms; ms.p = nullptr; ms.n = 0;
                        int cnt = 0;
                        if (have) cnt = WRITE ? vcf_info_field<false, true, false, true>(a, li, u, bias, key, fend, ms) : vcf_info_field<false, false, false, true>(a, li, u, bias, key, fend, ms);
                        uint32_t tot_n, tot_c;
                        // one scan for both counts: fields met (low half) and fields kept (high half), at most 64 of either
                        const uint32_t off = vcf_wave_excl_scan(ms.n, &tot_n), both = vcf_wave_excl_scan((cnt ? 1u : 0u) | (cnt == 1 ? 0x10000u : 0u), &tot_c);
                        if (__ballot(have && n_info_all + (both & 0xffffu) == 65535u)) bad = true;
                        if (WRITE) {
                            uint32_t d_src = 0, d_dst = 0, d_len = 0;
                            if (have && !bad && cnt == 1) { VcfSink<WRITE> ws; ws.p = o.p; ws.n = o.n + off; vcf_info_field<WRITE, false, true, true>(a, li, u, bias, key, fend, ws, &d_src, &d_dst, &d_len); }
                            for (unsigned long long big = __ballot(d_len != 0); big; big &= big - 1) {      // long strings: all lanes copy, a byte each per step
                                const int l = __builtin_ctzll(big);
                                const uint32_t src = (uint32_t)__shfl((int)d_src, l), dst = (uint32_t)__shfl((int)d_dst, l), len = (uint32_t)__shfl((int)d_len, l);
                                uint32_t i = lane * 8u;                                              // eight bytes per lane and step (unaligned 8-byte loads and stores), the tail a byte per lane
                                for (; i + 8u <= len; i += 512u) { uint64_t v; __builtin_memcpy(&v, u + src + i, 8); __builtin_memcpy(o.p + dst + i, &v, 8); }
                                i = (len & ~7u) + lane;
                                if (i < len) o.p[dst + i] = u[src + i];
                            }
                        }
                        if (!WRITE || !a.endsv) {                                                     // the first fields that begin with END= / SVLEN= (vcf_tabix_end)
                            const bool is_e = have && fend - key >= 4 && u[key] == 'E' && u[key + 1] == 'N' && u[key + 2] == 'D' && u[key + 3] == '=';
                            const bool is_s = have && fend - key >= 6 && u[key] == 'S' && u[key + 1] == 'V' && u[key + 2] == 'L' && u[key + 3] == 'E' && u[key + 4] == 'N' && u[key + 5] == '=';
                            const unsigned long long be = __ballot(is_e), bs = __ballot(is_s);
                            if (be && w_pe == 0xffffffffu) w_pe = (uint32_t)__shfl((int)key, __builtin_ctzll(be)) + 4u;
                            if (bs && w_ps == 0xffffffffu) w_ps = (uint32_t)__shfl((int)key, __builtin_ctzll(bs)) + 6u;
                        }
                        o.n += tot_n; n_info += tot_c >> 16; n_info_all += tot_c & 0xffffu;
                        if (bad) break;
                    }
                    seg = sep[nsep - 1] + 1u;
                    __syncthreads();
                    nsep = 0;
                    if (last || bad) break;
                }
                if (!WRITE && a.endsv && lane == 0) { a.endsv[2 * li] = w_pe; a.endsv[2 * li + 1] = w_ps; }
                if (WRITE && a.endsv) { w_pe = a.endsv[2 * li]; w_ps = a.endsv[2 * li + 1]; }
            } else if (!(WRITE && a.info_none))                                                      // (nothing of INFO projected: the measure pass has seen every field)
            for (uint32_t key = fs[7];;) {                                                           // fields between ';' (kstrtok)
                uint32_t fend = key; while (fend < e7 && u[fend] != ';') fend++;
                if (n_info_all == 65535) { bad = true; break; }
                const int got = vcf_info_field<WRITE, false>(a, li, u, bias, key, fend, o);
                n_info += got == 1; n_info_all += got != 0;
                if (fend >= e7) break;
                key = fend + 1;
            }
        }
    }
    if (WAVE && WRITE && !bad) rlen = (int32_t)(vcf_tabix_end(u, pos, fs[3], fe[3], fs[4], fe[4], fs[7], fe[7], fe[7] < l1 ? fe[7] + 1 : l1, l1, true, w_pe, w_ps) - pos);
    // FORMAT + sample columns (vcf_parse_format vcf.c:3686-3742; steps 3137-3684): per-sample text A:B:C becomes per-field arrays.  The tables of
    // the FORMAT keys (up to htslib's 255, MAX_N_FMT) live in LDS for the wave encoder -- which then takes the sample columns a sample per lane --
    // and in private arrays of VCF_MAXF for the lane-per-line encoder, which hands a line with more keys to the wave encoder (counters[2]).
    uint32_t n_fmt_kept = 0, n_sample = 0; const uint32_t indiv0 = o.n;
    if (SMP && !bad && a.n_smp > 0 && fe[7] < l1 && !(WRITE && a.fmt_none)) {
        const uint32_t fp = fe[7] + 1; uint32_t fq = fp; while (fq < l1 && u[fq] != '\t') fq++;
        if (fq >= l1) bad = true;                                                                     // "FORMAT column with no sample columns"
        else if (fq - fp == 1 && u[fp] == '.') n_sample = (uint32_t)a.n_smp;                          // FORMAT ".": nothing to parse, the sample columns are not looked at
        else {
            constexpr int MAXF = WAVE ? VCF_MAXF_WAVE : VCF_MAXF, NP = WAVE ? 1 : VCF_MAXF;
            int32_t key_p[NP]; uint8_t ht_p[NP], flg_p[NP]; uint32_t mx_l_p[NP], mx_m_p[NP], mx_g_p[NP], fsz_p[NP], fat_p[NP];      // flg: 1 = GT, 2 = dropped duplicate
            int32_t *key = WAVE ? fl->key : key_p; uint8_t *ht = WAVE ? fl->ht : ht_p, *flg = WAVE ? fl->flg : flg_p;
            uint32_t *mx_l = WAVE ? fl->mx_l : mx_l_p, *mx_m = WAVE ? fl->mx_m : mx_m_p, *mx_g = WAVE ? fl->mx_g : mx_g_p, *fsz = WAVE ? fl->fsz : fsz_p, *fat = WAVE ? fl->fat : fat_p;
            const uint32_t lane = WAVE ? (threadIdx.x & 63u) : 0u;
            // dict2: the keys (every lane of a wave alike: the stores hit one address with one value)
            int n_fmt = 0;
            for (uint32_t t = fp;;) {
                uint32_t c = t; while (c < fq && u[c] != ':') c++;
                if (n_fmt >= MAXF) { if (!WAVE) a.counters[2] = 1u; bad = true; break; }              // (the lane-per-line encoder: the batch goes to the wave encoder; there: htslib's own limit)
                const int k = vcf_dict_find(a.ids, u + t, c - t);
                key[n_fmt] = 0; ht[n_fmt] = 3;
                if (k < 0 || a.ids.ftyp[k] == 15) {
                    if (c - t == 1 && u[t] == '.') { bad = true; break; }                            // "Invalid FORMAT tag name '.'"
                    if (!WRITE && lead) { const uint32_t q = atomicAdd(&a.counters[0], 1u); if (q < a.undef_cap) a.undef[q] = {(uint32_t)li, t + bias, c - t, 3u}; }
                } else { key[n_fmt] = a.ids.id[k]; ht[n_fmt] = a.ids.ftyp[k]; }
                flg[n_fmt] = (c - t == 2 && u[t] == 'G' && u[t + 1] == 'T') ? 1 : 0;
                if (a.fmt_keep && !((a.fmt_keep[(uint32_t)key[n_fmt] >> 5] >> ((uint32_t)key[n_fmt] & 31u)) & 1u)) flg[n_fmt] |= 4;      // flg: 4 = outside the projection
                mx_l[n_fmt] = mx_m[n_fmt] = mx_g[n_fmt] = 0;
                n_fmt++;
                if (c >= fq) break;
                t = c + 1;
            }
            if (WAVE) __syncthreads();
            const uint32_t body = fq + 1, end = l1;
            // The sample columns of a wave: the tabs of [body, end) collected 16 bytes per lane into `sep` (a stretch at a time), then a column per
            // lane.  A column that starts at the line's end does not count (the serial walk stops there); columns behind the header's samples are
            // not looked at.  fn(column's start, its end, its number) returns true on an error.
            auto wave_columns = [&](auto &&fn) -> uint32_t {
                uint32_t seg = body, nsep = 0, done = 0;
                for (uint32_t base = body & ~15u;; base += 1024u) {
                    const bool last = base + 1024u >= end;
                    if (base < end) {
                        const uint32_t p = base + lane * 16u;
                        uint32_t m = p < end ? vcf_eq_mask16(u, p, body, end, 0x09090909u) : 0u, tot;
                        uint32_t at = nsep + vcf_wave_excl_scan((uint32_t)__builtin_popcount(m), &tot);
                        for (; m; m &= m - 1) sep[at++] = p + (uint32_t)__builtin_ctz(m);
                        nsep += tot;
                    }
                    if (last) { if (lane == 0) sep[nsep] = end; nsep++; }
                    if (!last && nsep + 1025u <= VCF_WSEP) continue;
                    __syncthreads();
                    uint32_t ncol = nsep;
                    if (last && (nsep > 1 ? sep[nsep - 2] + 1u : seg) >= end) ncol--;               // the column that would start at the line's end
                    if (ncol > (uint32_t)a.n_smp - done) ncol = (uint32_t)a.n_smp - done;
                    for (uint32_t k0 = 0; k0 < ncol && !bad; k0 += 64u) {
                        const uint32_t k = k0 + lane; const bool have = k < ncol;
                        bool err = false;
                        if (have) err = fn(k ? sep[k - 1] + 1u : seg, sep[k], done + k);
                        if (__ballot(err)) bad = true;
                    }
                    done += ncol;
                    seg = sep[nsep - 1] + 1u;
                    __syncthreads();
                    nsep = 0;
                    if (last || bad || done >= (uint32_t)a.n_smp) break;
                }
                return done;
            };
            // max3: widths of every field over the samples
            if (!bad) {
                if (WAVE) n_sample = wave_columns([&](uint32_t s, uint32_t e, uint32_t) { return !vcf_fmt_widths<true>(u, s, e, n_fmt, flg, mx_l, mx_m, mx_g); });
                else {
                    uint32_t r = body;
                    while (r < end && !bad) {
                        uint32_t e = r; while (e < end && u[e] != '\t') e++;
                        if (!vcf_fmt_widths<false>(u, r, e, n_fmt, flg, mx_l, mx_m, mx_g)) bad = true;
                        n_sample++;
                        if (n_sample == (uint32_t)a.n_smp) break;
                        r = e + 1;
                    }
                }
            }
            // alloc4: slot sizes, duplicates, and where every field's array sits in the indiv block
            if (WAVE) __syncthreads();
            if (!bad && (!WAVE || lane == 0)) {
                for (int j = 0; j < n_fmt; j++) {
                    if (!mx_m[j]) mx_m[j] = 1;
                    if (ht[j] == 3) fsz[j] = (flg[j] & 1) ? mx_g[j] << 2 : mx_l[j];
                    else if (ht[j] == 1 || ht[j] == 2) fsz[j] = mx_m[j] << 2;
                    else { bad = true; break; }                                                       // a FORMAT Flag: "currently not supported"
                }
                for (int i = 1; i < n_fmt && !bad; i++) for (int j = 0; j < i; j++) if (!(flg[j] & 2) && key[i] == key[j] && ht[i] != 15) { flg[i] |= 2; break; }
            }
            if (WAVE) { __syncthreads(); bad = __shfl((int)bad, 0) != 0; }
            if (!bad && n_sample != (uint32_t)a.n_smp) bad = true;                                   // check7 (fill5 errors found below are errors either way)
            if (!bad) {
                for (int j = 0; j < n_fmt; j++) {                                                     // (a wave: every lane alike; lane 0 keeps the places)
                    if (flg[j] & 6) continue;
                    n_fmt_kept++;
                    o.key(key[j]);
                    if (ht[j] == 3 && !(flg[j] & 1)) o.size(fsz[j], 7); else o.size(fsz[j] >> 2, ht[j] == 2 ? 5 : 3);
                    if (!WAVE || lane == 0) fat[j] = o.n;
                    o.n += fsz[j] * n_sample;
                }
                if (WAVE) __syncthreads();
                // fill5: every sample, field by field (validated in the measure pass, stored in the write pass)
                if (WAVE) (void)wave_columns([&](uint32_t s, uint32_t e, uint32_t m) { return vcf_fmt_fill_sample<WRITE>(a, li, u, bias, s, e, end, m, n_fmt, ht, flg, fsz, fat, o.p); });
                else {
                    uint32_t t = body, m = 0;
                    while (t < end && !bad) {
                        if (m == (uint32_t)a.n_smp) break;
                        uint32_t e = t; while (e < end && u[e] != '\t') e++;
                        if (vcf_fmt_fill_sample<WRITE>(a, li, u, bias, t, e, end, m, n_fmt, ht, flg, fsz, fat, o.p)) { bad = true; break; }
                        m++; t = e + 1;
                    }
                }
                if (!bad && n_sample == 0) { o.n = indiv0; n_fmt_kept = 0; }
            }
        }
    }
    if (SMP && a.fmt_none && !bad) { o.n = indiv0; n_fmt_kept = 0; n_sample = 0; }                      // (validated above in the measure pass; the record carries no sample data)
    if (bad) { atomicMin(a.first_bad, (unsigned long long)li); if (!WRITE) a.rec_len[li] = 0; return; }
    if (!WRITE) { a.rec_len[li] = o.n; return; }
    const uint32_t total = o.n;
    o.n = 0;
    o.w32(indiv0 - 8); o.w32(total - indiv0);                                                        // l_shared (core + shared block), l_indiv
    o.w32((uint32_t)rid); o.w32((uint32_t)(uint64_t)pos); o.w32((uint32_t)rlen);                      // (the core holds the position's low word; text keeps 64-bit positions: the high word goes beside the records)
    if (a.pos_hi) a.pos_hi[li] = (int32_t)(pos >> 32);
    o.w32(qbits);
    o.w32(n_info | (n_allele << 16)); o.w32((n_sample & 0xffffffu) | (n_fmt_kept << 24));
}
