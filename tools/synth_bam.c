/*
 * synth_bam.c -- deterministic synthetic BGZF BAM generator.
 *
 * Workload tooling, not part of the scan path: bench.py and the tests use it to
 * make WGS-shaped inputs.  Compressor = zlib level 6 raw deflate (statically
 * linked), 65280-byte BGZF payloads cut without regard to record boundaries
 * (records straddle blocks), header flushed into its own block, optional
 * 28-byte EOF block.
 *
 * Record model: coordinate-sorted, paired 150 bp; 25 @SQ (GRCh38 lengths), one
 * @RG (ID:rg1 SM:NA00001); QNAME SYN:<run>:<tile>:<x>:<y>; FLAG 94 % from
 * {99,147,83,163} + unmapped/dup/secondary/supplementary mix; MAPQ skewed to
 * 60; CIGAR 85 % 150M, 10 % soft-clipped, 5 % with I/D; SEQ uniform ACGT with
 * 0.1 % N; QUAL 4-bin Markov runs; aux NM:C MD:Z AS:C XS:C RG:Z.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#define NREF 25
static const char *REF_NAME[NREF] = {"chr1","chr2","chr3","chr4","chr5","chr6","chr7","chr8","chr9","chr10","chr11","chr12",
    "chr13","chr14","chr15","chr16","chr17","chr18","chr19","chr20","chr21","chr22","chrX","chrY","chrM"};
static const uint32_t REF_LEN[NREF] = {248956422,242193529,198295559,190214555,181538259,170805979,159345973,145138636,
    138394717,133797422,135086622,133275309,114364328,107043718,101991189,90338345,83257441,80373285,58617616,64444167,
    46709983,50818468,156040895,57227415,16569};

typedef struct { uint64_t s; } rng_t;
static inline uint64_t mix64(uint64_t z) { z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
static inline uint64_t rnext(rng_t *r) { r->s += 0x9E3779B97F4A7C15ull; uint64_t z = r->s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

static inline void put32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }

static int reg2bin(int64_t beg, int64_t end) {
    --end;
    if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

/* global record index -> (tid, pos): monotone by construction */
typedef struct { uint64_t total_n; uint64_t seed; uint64_t genome; uint64_t cum[NREF + 1]; } model_t;

static void model_init(model_t *m, uint64_t total_n, uint64_t seed) {
    m->total_n = total_n; m->seed = seed; m->cum[0] = 0;
    for (int i = 0; i < NREF; i++) m->cum[i + 1] = m->cum[i] + (REF_LEN[i] > 400 ? REF_LEN[i] - 400 : 1);
    m->genome = m->cum[NREF];
}
static void model_pos(const model_t *m, uint64_t idx, int *tid, int64_t *pos) {
    /* evenly spaced anchor + jitter smaller than the spacing keeps the order */
    __uint128_t a = (__uint128_t)idx * m->genome / m->total_n;
    __uint128_t b = (__uint128_t)(idx + 1) * m->genome / m->total_n;
    uint64_t lo = (uint64_t)a, span = (uint64_t)(b - a);
    uint64_t g = lo + (span ? mix64(m->seed ^ (idx * 0xD1B54A32D192ED03ull)) % span : 0);
    int t = 0;
    while (t + 1 < NREF && g >= m->cum[t + 1]) t++;
    *tid = t; *pos = (int64_t)(g - m->cum[t]);
}

static int fmt_u(char *p, uint64_t v) { char t[24]; int n = 0; do { t[n++] = (char)('0' + v % 10); v /= 10; } while (v); for (int i = 0; i < n; i++) p[i] = t[n - 1 - i]; return n; }

/* one record -> bytes at dst; returns length incl. the 4-byte block_size */
static size_t gen_record(const model_t *m, uint64_t idx, uint8_t *dst) {
    /* pairing: within each run of 20 records, record j<10 pairs with j+10 */
    uint64_t blk = idx / 20, j = idx % 20;
    int first = j < 10;
    uint64_t mate_idx = first ? idx + 10 : idx - 10;
    if (mate_idx >= m->total_n) mate_idx = idx;
    uint64_t pair_id = blk * 10 + (j % 10);
    rng_t pr = { m->seed ^ (pair_id * 0xA24BAED4963EE407ull) };      /* shared by the two mates */
    rng_t r = { m->seed ^ (idx * 0x9FB21C651E98DF25ull) ^ 0x5555 };
    int tid, mtid; int64_t pos, mpos;
    model_pos(m, idx, &tid, &pos);
    model_pos(m, mate_idx, &mtid, &mpos);

    /* name: SYN:<run>:<tile>:<x>:<y>, 20..34 chars */
    char name[48]; int nl = 0;
    memcpy(name, "SYN:", 4); nl = 4;
    uint64_t a = rnext(&pr);
    nl += fmt_u(name + nl, 100000 + a % 900000000ull % (a & 1 ? 900000ull : 900000000ull)); name[nl++] = ':';
    nl += fmt_u(name + nl, 1000 + (a >> 32) % 90000); name[nl++] = ':';
    uint64_t b = rnext(&pr);
    nl += fmt_u(name + nl, 1000 + b % 99000); name[nl++] = ':';
    nl += fmt_u(name + nl, 1000 + (b >> 32) % 990000);
    while (nl < 20) name[nl++] = '0';
    name[nl] = 0;
    int l_qname = nl + 1;

    /* flags */
    uint64_t f = rnext(&pr);
    int fwd_first = f & 1;
    int flag = first ? (fwd_first ? 99 : 83) : (fwd_first ? 147 : 163);
    uint64_t q = rnext(&r);
    int unmapped = 0;
    unsigned cls = (unsigned)(q % 1000);
    if (cls < 15) { unmapped = 1; flag = (flag & ~2) | 4; }
    else if (cls < 35) flag |= 0x400;
    else if (cls < 48) flag |= 0x100;
    else if (cls < 60) flag |= 0x800;
    int mapq = unmapped ? 0 : (((q >> 16) % 100) < 80 ? 60 : (int)((q >> 24) % 60));

    /* cigar */
    uint32_t cig[5]; int ncig = 0; int64_t rlen = 150; int nm = 0;
    char md[40]; int mdl = 0;
    if (!unmapped) {
        unsigned c = (unsigned)((q >> 32) % 100);
        if (c < 85) { cig[ncig++] = 150u << 4; }
        else if (c < 95) {
            unsigned s = 1 + (unsigned)((q >> 40) % 60);
            if ((q >> 50) & 1) { cig[ncig++] = (s << 4) | 4; cig[ncig++] = ((150 - s) << 4); }
            else { cig[ncig++] = ((150 - s) << 4); cig[ncig++] = (s << 4) | 4; }
            rlen = 150 - s;
        } else {
            unsigned x = 10 + (unsigned)((q >> 40) % 100), k = 1 + (unsigned)((q >> 52) % 6);
            if ((q >> 58) & 1) { cig[ncig++] = x << 4; cig[ncig++] = (k << 4) | 1; cig[ncig++] = (150 - x - k) << 4; rlen = 150 - k; }
            else { cig[ncig++] = x << 4; cig[ncig++] = (k << 4) | 2; cig[ncig++] = (150 - x) << 4; rlen = 150 + k; }
            nm = (int)k;
        }
        /* MD: mostly "<len>", sometimes one mismatch */
        uint64_t mm = rnext(&r);
        int mlen = (int)(rlen > 150 ? 150 : rlen);
        if (mm % 10 < 7) mdl = fmt_u(md, (uint64_t)mlen);
        else { int at = (int)((mm >> 8) % (uint64_t)(mlen - 1)); mdl = fmt_u(md, (uint64_t)at); md[mdl++] = "ACGT"[(mm >> 20) & 3]; mdl += fmt_u(md + mdl, (uint64_t)(mlen - at - 1)); nm++; }
    } else { mdl = 0; }
    md[mdl] = 0;

    int l_seq = 150;
    size_t body = 32 + (size_t)l_qname + 4 * (size_t)ncig + 75 + 150;
    size_t auxlen = unmapped ? (3 + 4) : (4 + 3 + (size_t)mdl + 1 + 4 + 4 + 3 + 4);
    size_t block_len = body + auxlen;
    uint8_t *p = dst;
    put32(p, (uint32_t)block_len); p += 4;
    put32(p, (uint32_t)tid); put32(p + 4, (uint32_t)pos);
    int bin = reg2bin(pos, pos + (unmapped ? 1 : rlen));
    put32(p + 8, ((uint32_t)bin << 16) | ((uint32_t)mapq << 8) | (uint32_t)l_qname);
    put32(p + 12, ((uint32_t)flag << 16) | (uint32_t)ncig);
    put32(p + 16, (uint32_t)l_seq);
    put32(p + 20, (uint32_t)mtid); put32(p + 24, (uint32_t)mpos);
    int64_t tl = 0;
    if (mtid == tid && mate_idx != idx) tl = first ? (mpos + 150 - pos) : -(pos + 150 - mpos);
    put32(p + 28, (uint32_t)(int32_t)tl);
    p += 32;
    memcpy(p, name, (size_t)l_qname); p += l_qname;
    for (int k = 0; k < ncig; k++) { put32(p, cig[k]); p += 4; }
    /* seq: 2 bases per byte, uniform ACGT, 0.1 % N */
    for (int k = 0; k < 75; k += 8) {
        uint64_t w = rnext(&r), n = rnext(&r);
        for (int t = 0; t < 8 && k + t < 75; t++) {
            unsigned hi = 1u << ((w >> (4 * t)) & 3), lo = 1u << ((w >> (4 * t + 2)) & 3);
            if (((n >> (8 * t)) & 0xff) == 0 && ((w >> 40) & 3) == 0) hi = 15;
            p[k + t] = (uint8_t)((hi << 4) | lo);
        }
    }
    p += 75;
    /* qual: 4-bin Markov runs */
    {
        static const uint8_t QB[4] = {2, 11, 25, 37};
        unsigned st = 3; uint64_t w = 0; int have = 0;
        for (int k = 0; k < 150; k++) {
            if (!have) { w = rnext(&r); have = 8; }
            unsigned u = (unsigned)(w & 0xff); w >>= 8; have--;
            if (u < 26) st = (u & 3);                              /* ~10 % chance to re-draw the bin */
            else if (u < 40 && st > 0 && k > 100) st--;            /* tail degradation */
            p[k] = QB[st];
        }
        p += 150;
    }
    /* aux */
    if (!unmapped) {
        p[0] = 'N'; p[1] = 'M'; p[2] = 'C'; p[3] = (uint8_t)nm; p += 4;
        p[0] = 'M'; p[1] = 'D'; p[2] = 'Z'; memcpy(p + 3, md, (size_t)mdl + 1); p += 3 + mdl + 1;
        p[0] = 'A'; p[1] = 'S'; p[2] = 'C'; p[3] = (uint8_t)(150 - 5 * nm - (int)(q >> 60)); p += 4;
        p[0] = 'X'; p[1] = 'S'; p[2] = 'C'; p[3] = (uint8_t)((q >> 44) % 100); p += 4;
    }
    p[0] = 'R'; p[1] = 'G'; p[2] = 'Z'; memcpy(p + 3, "rg1", 4); p += 7;
    return (size_t)(p - dst);
}

static size_t gen_header(uint8_t *dst) {
    char text[4096]; int tl = 0;
    tl += sprintf(text + tl, "@HD\tVN:1.6\tSO:coordinate\n");
    for (int i = 0; i < NREF; i++) tl += sprintf(text + tl, "@SQ\tSN:%s\tLN:%u\n", REF_NAME[i], REF_LEN[i]);
    tl += sprintf(text + tl, "@RG\tID:rg1\tSM:NA00001\tPL:SYNTH\n");
    uint8_t *p = dst;
    memcpy(p, "BAM\1", 4); p += 4;
    put32(p, (uint32_t)tl); p += 4;
    memcpy(p, text, (size_t)tl); p += tl;
    put32(p, NREF); p += 4;
    for (int i = 0; i < NREF; i++) {
        size_t nl = strlen(REF_NAME[i]) + 1;
        put32(p, (uint32_t)nl); p += 4; memcpy(p, REF_NAME[i], nl); p += nl;
        put32(p, REF_LEN[i]); p += 4;
    }
    return (size_t)(p - dst);
}

/* ---- BGZF block compression ---- */
static size_t bgzf_block(const uint8_t *src, size_t slen, uint8_t *dst, int level) {
    static const uint8_t HDR[16] = {0x1f,0x8b,0x08,0x04,0,0,0,0,0,0xff,0x06,0,0x42,0x43,0x02,0};
    z_stream zs; memset(&zs, 0, sizeof(zs));
    deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
    zs.next_in = (Bytef *)src; zs.avail_in = (uInt)slen;
    zs.next_out = dst + 18; zs.avail_out = 65536 - 18 - 8;
    int r = deflate(&zs, Z_FINISH);
    size_t clen = zs.total_out;
    deflateEnd(&zs);
    if (r != Z_STREAM_END) {                                          /* incompressible: store */
        memset(&zs, 0, sizeof(zs));
        deflateInit2(&zs, 0, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
        zs.next_in = (Bytef *)src; zs.avail_in = (uInt)slen; zs.next_out = dst + 18; zs.avail_out = 65536 - 18 - 8;
        deflate(&zs, Z_FINISH); clen = zs.total_out; deflateEnd(&zs);
    }
    memcpy(dst, HDR, 16);
    size_t total = 18 + clen + 8;
    dst[16] = (uint8_t)((total - 1) & 0xff); dst[17] = (uint8_t)((total - 1) >> 8);
    uint32_t crc = (uint32_t)crc32(crc32(0L, NULL, 0), src, (uInt)slen);
    put32(dst + 18 + clen, crc); put32(dst + 18 + clen + 4, (uint32_t)slen);
    return total;
}

typedef struct {
    const model_t *m; uint64_t i0, i1; uint8_t *buf; size_t len;
} gen_job_t;
static void *gen_thread(void *a) {
    gen_job_t *j = (gen_job_t *)a;
    size_t cap = (size_t)(j->i1 - j->i0) * 420 + 1024;
    j->buf = (uint8_t *)malloc(cap); j->len = 0;
    for (uint64_t i = j->i0; i < j->i1; i++) j->len += gen_record(j->m, i, j->buf + j->len);
    return NULL;
}
typedef struct {
    const uint8_t *raw; size_t raw_len; size_t hdr_len; size_t payload; int level;
    size_t n_chunks; size_t c0, c1; uint8_t *out; size_t *clen;
} cmp_job_t;
static void chunk_range(const cmp_job_t *j, size_t c, size_t *beg, size_t *end) {
    /* chunk 0 = header alone (if any), the rest fixed payload cuts of the record stream */
    if (j->hdr_len) {
        if (c == 0) { *beg = 0; *end = j->hdr_len; return; }
        *beg = j->hdr_len + (c - 1) * j->payload;
    } else *beg = c * j->payload;
    *end = *beg + j->payload; if (*end > j->raw_len) *end = j->raw_len;
}
static void *cmp_thread(void *a) {
    cmp_job_t *j = (cmp_job_t *)a;
    for (size_t c = j->c0; c < j->c1; c++) {
        size_t b, e; chunk_range(j, c, &b, &e);
        j->clen[c] = bgzf_block(j->raw + b, e - b, j->out + c * 65536, j->level);
    }
    return NULL;
}

/*
 * Generates records [rec0, rec0+n) of a conceptual file of total_n records.
 * with_header: emit the BAM header block first; with_eof: append the EOF block.
 * Returns bytes written to out (0 on error / insufficient capacity).
 * stats[0] = uncompressed bytes, stats[1] = number of BGZF blocks.
 */
size_t synth_bam_segment(uint64_t seed, uint64_t total_n, uint64_t rec0, uint64_t n, int with_header, int with_eof,
                         int level, int payload, int threads, uint8_t *out, size_t out_cap, uint64_t *stats) {
    static const uint8_t EOFB[28] = {0x1f,0x8b,0x08,0x04,0,0,0,0,0,0xff,0x06,0,0x42,0x43,0x02,0,0x1b,0,0x03,0,0,0,0,0,0,0,0,0};
    model_t m; model_init(&m, total_n, seed);
    if (threads < 1) threads = 1; if (threads > 256) threads = 256;
    if (payload <= 0 || payload > 65280) payload = 65280;
    gen_job_t gj[256]; pthread_t th[256];
    for (int t = 0; t < threads; t++) {
        gj[t].m = &m; gj[t].i0 = rec0 + n * (uint64_t)t / (uint64_t)threads; gj[t].i1 = rec0 + n * (uint64_t)(t + 1) / (uint64_t)threads;
        pthread_create(&th[t], NULL, gen_thread, &gj[t]);
    }
    size_t raw_len = 0; uint8_t hdr[8192]; size_t hl = with_header ? gen_header(hdr) : 0;
    for (int t = 0; t < threads; t++) { pthread_join(th[t], NULL); raw_len += gj[t].len; }
    raw_len += hl;
    uint8_t *raw = (uint8_t *)malloc(raw_len + 16); size_t o = 0;
    memcpy(raw, hdr, hl); o = hl;
    for (int t = 0; t < threads; t++) { memcpy(raw + o, gj[t].buf, gj[t].len); o += gj[t].len; free(gj[t].buf); }
    size_t rec_bytes = raw_len - hl;
    size_t n_chunks = (hl ? 1 : 0) + (rec_bytes + (size_t)payload - 1) / (size_t)payload;
    uint8_t *cbuf = (uint8_t *)malloc(n_chunks * 65536 + 64); size_t *clen = (size_t *)calloc(n_chunks + 1, sizeof(size_t));
    cmp_job_t cj[256];
    for (int t = 0; t < threads; t++) {
        cj[t].raw = raw; cj[t].raw_len = raw_len; cj[t].hdr_len = hl; cj[t].payload = (size_t)payload; cj[t].level = level;
        cj[t].n_chunks = n_chunks; cj[t].c0 = n_chunks * (size_t)t / (size_t)threads; cj[t].c1 = n_chunks * (size_t)(t + 1) / (size_t)threads;
        cj[t].out = cbuf; cj[t].clen = clen;
        pthread_create(&th[t], NULL, cmp_thread, &cj[t]);
    }
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    size_t total = 0;
    for (size_t c = 0; c < n_chunks; c++) total += clen[c];
    if (with_eof) total += 28;
    if (total > out_cap) { free(raw); free(cbuf); free(clen); return 0; }
    o = 0;
    for (size_t c = 0; c < n_chunks; c++) { memcpy(out + o, cbuf + c * 65536, clen[c]); o += clen[c]; }
    if (with_eof) { memcpy(out + o, EOFB, 28); o += 28; }
    if (stats) { stats[0] = raw_len; stats[1] = n_chunks + (with_eof ? 1 : 0); }
    free(raw); free(cbuf); free(clen);
    return o;
}
