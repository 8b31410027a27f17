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

typedef size_t (*gen_fn)(const model_t *, uint64_t, uint8_t *);
typedef struct {
    const model_t *m; uint64_t i0, i1; uint8_t *buf; size_t len; gen_fn gen; size_t rec_cap;
} gen_job_t;
static void *gen_thread(void *a) {
    gen_job_t *j = (gen_job_t *)a;
    size_t cap = (size_t)(j->i1 - j->i0) * j->rec_cap + 1024;
    j->buf = (uint8_t *)malloc(cap); j->len = 0;
    for (uint64_t i = j->i0; i < j->i1; i++) j->len += j->gen(j->m, i, j->buf + j->len);
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
static size_t segment_impl(gen_fn gen, size_t rec_cap, const uint8_t *hdr, size_t hl, uint64_t seed, uint64_t total_n, uint64_t rec0, uint64_t n, int with_eof,
                           int level, int payload, int threads, uint8_t *out, size_t out_cap, uint64_t *stats) {
    static const uint8_t EOFB[28] = {0x1f,0x8b,0x08,0x04,0,0,0,0,0,0xff,0x06,0,0x42,0x43,0x02,0,0x1b,0,0x03,0,0,0,0,0,0,0,0,0};
    model_t m; model_init(&m, total_n, seed);
    if (threads < 1) threads = 1; if (threads > 256) threads = 256;
    if (payload <= 0 || payload > 65280) payload = 65280;
    gen_job_t gj[256]; pthread_t th[256];
    for (int t = 0; t < threads; t++) {
        gj[t].m = &m; gj[t].gen = gen; gj[t].rec_cap = rec_cap; gj[t].i0 = rec0 + n * (uint64_t)t / (uint64_t)threads; gj[t].i1 = rec0 + n * (uint64_t)(t + 1) / (uint64_t)threads;
        pthread_create(&th[t], NULL, gen_thread, &gj[t]);
    }
    size_t raw_len = 0;
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

size_t synth_bam_segment(uint64_t seed, uint64_t total_n, uint64_t rec0, uint64_t n, int with_header, int with_eof,
                         int level, int payload, int threads, uint8_t *out, size_t out_cap, uint64_t *stats) {
    uint8_t hdr[8192]; size_t hl = with_header ? gen_header(hdr) : 0;
    return segment_impl(gen_record, 420, hdr, hl, seed, total_n, rec0, n, with_eof, level, payload, threads, out, out_cap, stats);
}

/* =====================================================================================================================
 * Synthetic BCF: VCFv4.2 header, 25 contigs, FILTER q10/s50, INFO DP:1:Integer AF:A:Float
 * AC:A:Integer AN:1:Integer MQ:1:Float DB:0:Flag SB:4:Integer ANN_S:1:String, FORMAT GT GQ:1 DP:1 AD:R PL:G (Integer)
 * GL:G (Float) for 16 samples; 90 % biallelic SNV, 8 % indel, 2 % tri-allelic; ~3 % missing values, 1 % missing QUAL;
 * FILTER PASS / q10 / q10+s50; integer vectors use the narrowest of int8/int16/int32 that fits the record's values.
 * ===================================================================================================================== */
#define BCF_NS 16
static const char *BCF_SAMPLES[BCF_NS] = {"NA00001","NA00002","NA00003","NA00004","NA00005","NA00006","NA00007","NA00008",
    "NA00009","NA00010","NA00011","NA00012","NA00013","NA00014","NA00015","NA00016"};
enum { K_PASS = 0, K_Q10, K_S50, K_DP, K_AF, K_AC, K_AN, K_MQ, K_DB, K_SB, K_ANN, K_GT, K_GQ, K_AD, K_PL, K_GL };

static size_t gen_bcf_header(uint8_t *dst) {
    char *p = (char *)dst + 9; char *q = p;
    q += sprintf(q, "##fileformat=VCFv4.2\n##FILTER=<ID=PASS,Description=\"All filters passed\">\n");
    for (int i = 0; i < NREF; i++) q += sprintf(q, "##contig=<ID=%s,length=%u>\n", REF_NAME[i], REF_LEN[i]);
    q += sprintf(q, "##FILTER=<ID=q10,Description=\"Quality below 10\">\n##FILTER=<ID=s50,Description=\"Less than 50%% of samples have data\">\n");
    q += sprintf(q, "##INFO=<ID=DP,Number=1,Type=Integer,Description=\"Total Depth\">\n##INFO=<ID=AF,Number=A,Type=Float,Description=\"Allele Frequency\">\n");
    q += sprintf(q, "##INFO=<ID=AC,Number=A,Type=Integer,Description=\"Allele count\">\n##INFO=<ID=AN,Number=1,Type=Integer,Description=\"Allele number\">\n");
    q += sprintf(q, "##INFO=<ID=MQ,Number=1,Type=Float,Description=\"RMS mapping quality\">\n##INFO=<ID=DB,Number=0,Type=Flag,Description=\"dbSNP membership\">\n");
    q += sprintf(q, "##INFO=<ID=SB,Number=4,Type=Integer,Description=\"Strand counts\">\n##INFO=<ID=ANN_S,Number=1,Type=String,Description=\"Short annotation\">\n");
    q += sprintf(q, "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n##FORMAT=<ID=GQ,Number=1,Type=Integer,Description=\"Genotype Quality\">\n");
    q += sprintf(q, "##FORMAT=<ID=DP,Number=1,Type=Integer,Description=\"Read Depth\">\n##FORMAT=<ID=AD,Number=R,Type=Integer,Description=\"Allelic depths\">\n");
    q += sprintf(q, "##FORMAT=<ID=PL,Number=G,Type=Integer,Description=\"Phred-scaled likelihoods\">\n##FORMAT=<ID=GL,Number=G,Type=Float,Description=\"Genotype likelihoods\">\n");
    q += sprintf(q, "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT");
    for (int i = 0; i < BCF_NS; i++) q += sprintf(q, "\t%s", BCF_SAMPLES[i]);
    *q++ = '\n'; *q++ = 0;
    size_t l_text = (size_t)(q - p);
    memcpy(dst, "BCF\2\2", 5); put32(dst + 5, (uint32_t)l_text);
    return 9 + l_text;
}

static inline uint8_t *put_desc(uint8_t *p, int n, int t) {
    if (n < 15) { *p++ = (uint8_t)((n << 4) | t); return p; }
    *p++ = (uint8_t)(0xF0 | t);
    if (n < 128) { *p++ = 0x11; *p++ = (uint8_t)n; } else { *p++ = 0x12; *p++ = (uint8_t)n; *p++ = (uint8_t)(n >> 8); }
    return p;
}
/* integer vector with the narrowest width; v == INT32_MIN encodes "missing" */
static uint8_t *put_ints(uint8_t *p, const int32_t *v, int n, int per_desc) {
    int w = 1;
    for (int i = 0; i < n; i++) { if (v[i] == INT32_MIN) continue; if (v[i] < -120 || v[i] > 127) w = w < 2 ? 2 : w; if (v[i] < -32000 || v[i] > 32767) w = 4; }
    p = put_desc(p, per_desc, w == 1 ? 1 : w == 2 ? 2 : 3);
    for (int i = 0; i < n; i++) {
        int miss = v[i] == INT32_MIN;
        if (w == 1) *p++ = miss ? 0x80 : (uint8_t)v[i];
        else if (w == 2) { uint16_t x = miss ? 0x8000 : (uint16_t)v[i]; *p++ = (uint8_t)x; *p++ = (uint8_t)(x >> 8); }
        else { put32(p, miss ? 0x80000000u : (uint32_t)v[i]); p += 4; }
    }
    return p;
}
static inline uint8_t *put_key(uint8_t *p, int k) { *p++ = 0x11; *p++ = (uint8_t)k; return p; }
static inline uint8_t *put_f32(uint8_t *p, float f) { uint32_t b; memcpy(&b, &f, 4); put32(p, b); return p + 4; }

static size_t gen_bcf_record(const model_t *m, uint64_t idx, uint8_t *dst) {
    rng_t r; r.s = mix64(m->seed * 0x2545F4914F6CDD1Dull + idx);
    int tid; int64_t pos; model_pos(m, idx, &tid, &pos);
    static const char B[4] = {'A', 'C', 'G', 'T'};
    uint64_t kind = rnext(&r) % 100;
    int n_allele = kind < 98 ? 2 : 3;
    char al[3][16]; int all[3];
    if (kind < 90) { int a = (int)(rnext(&r) & 3), b = (a + 1 + (int)(rnext(&r) % 3)) & 3; al[0][0] = B[a]; all[0] = 1; al[1][0] = B[b]; all[1] = 1; }
    else if (kind < 98) {
        int L = 2 + (int)(rnext(&r) % 9); for (int i = 0; i < L; i++) al[0][i] = B[rnext(&r) & 3]; all[0] = L; al[1][0] = al[0][0]; all[1] = 1;
        if (rnext(&r) & 1) { char t[16]; memcpy(t, al[0], 16); memcpy(al[0], al[1], 16); memcpy(al[1], t, 16); int x = all[0]; all[0] = all[1]; all[1] = x; }
    } else { int a = (int)(rnext(&r) & 3); al[0][0] = B[a]; all[0] = 1; al[1][0] = B[(a + 1) & 3]; all[1] = 1; al[2][0] = B[a]; al[2][1] = B[(a + 2) & 3]; all[2] = 2; }
    int G = n_allele * (n_allele + 1) / 2;
    uint8_t *sh = dst + 32, *p = sh;
    if (rnext(&r) % 10 < 3) { char id[16]; int l = 2 + fmt_u(id + 2, 1000 + rnext(&r) % 900000000ull); id[0] = 'r'; id[1] = 's'; p = put_desc(p, l, 7); memcpy(p, id, (size_t)l); p += l; }
    else *p++ = 0x07;
    for (int a = 0; a < n_allele; a++) { p = put_desc(p, all[a], 7); memcpy(p, al[a], (size_t)all[a]); p += all[a]; }
    uint64_t fl = rnext(&r) % 100;
    if (fl < 80) { *p++ = 0x11; *p++ = K_PASS; } else if (fl < 92) { *p++ = 0x11; *p++ = K_Q10; } else { *p++ = 0x21; *p++ = K_Q10; *p++ = K_S50; }
    int n_info = 0; int32_t iv[8];
    const int miss3 = 3;
    int32_t dp_tot = (int32_t)(50 + rnext(&r) % 900); if (rnext(&r) % 64 == 0) dp_tot += 40000;
    iv[0] = (rnext(&r) % 100 < (uint64_t)miss3) ? INT32_MIN : dp_tot;
    p = put_key(p, K_DP); p = put_ints(p, iv, 1, 1); n_info++;
    p = put_key(p, K_AF); p = put_desc(p, n_allele - 1, 5);
    for (int a = 1; a < n_allele; a++) { if (rnext(&r) % 100 < (uint64_t)miss3) { put32(p, 0x7F800001u); p += 4; } else p = put_f32(p, (float)(rnext(&r) % 10000) / 10000.0f); }
    n_info++;
    for (int a = 1; a < n_allele; a++) iv[a - 1] = (int32_t)(1 + rnext(&r) % 31);
    p = put_key(p, K_AC); p = put_ints(p, iv, n_allele - 1, n_allele - 1); n_info++;
    iv[0] = 2 * BCF_NS; p = put_key(p, K_AN); p = put_ints(p, iv, 1, 1); n_info++;
    p = put_key(p, K_MQ); p = put_desc(p, 1, 5); p = put_f32(p, 20.0f + (float)(rnext(&r) % 4000) / 100.0f); n_info++;
    if (rnext(&r) % 10 < 3) { p = put_key(p, K_DB); *p++ = 0x00; n_info++; }
    for (int k = 0; k < 4; k++) iv[k] = (int32_t)(rnext(&r) % 300);
    p = put_key(p, K_SB); p = put_ints(p, iv, 4, 4); n_info++;
    { static const char *ANN[6] = {"missense_variant", "synonymous_variant", "intron_variant", "intergenic_region", "stop_gained", "splice_region_variant&intron_variant"};
      const char *a = ANN[rnext(&r) % 6]; int l = (int)strlen(a); p = put_key(p, K_ANN); p = put_desc(p, l, 7); memcpy(p, a, (size_t)l); p += l; n_info++; }
    size_t l_shared = (size_t)(p - sh);
    /* ---- individual data ---- */
    uint8_t *in = p;
    int32_t gq[BCF_NS], dp[BCF_NS], ad[BCF_NS * 3], pl[BCF_NS * 6]; float gl[BCF_NS * 6]; uint8_t gt[BCF_NS * 2];
    for (int s = 0; s < BCF_NS; s++) {
        uint64_t x = rnext(&r);
        int a0 = (int)(x % (uint64_t)n_allele), a1 = (int)((x >> 8) % (uint64_t)n_allele); if (a0 > a1) { int t = a0; a0 = a1; a1 = t; }
        int ph = (int)((x >> 16) & 1);
        int missing = ((x >> 20) % 100) < 2;
        gt[2 * s] = missing ? 0 : (uint8_t)(((a0 + 1) << 1));
        gt[2 * s + 1] = missing ? 0 : (uint8_t)(((a1 + 1) << 1) | ph);
        gq[s] = ((x >> 28) % 100 < (uint64_t)miss3) ? INT32_MIN : (int32_t)((x >> 36) % 100);
        int d = (int)(5 + (x >> 44) % 60); if ((x >> 52) % 97 == 0) d += 200;
        dp[s] = ((x >> 56) % 100 < (uint64_t)miss3) ? INT32_MIN : d;
        uint64_t y = rnext(&r);
        int rem = d;
        for (int a = 0; a < n_allele; a++) { int v = a == n_allele - 1 ? rem : (int)(y % (uint64_t)(rem + 1)); y >>= 8; ad[s * n_allele + a] = v; rem -= v; }
        int best = (int)(rnext(&r) % (uint64_t)G);
        for (int g = 0; g < G; g++) { int v = g == best ? 0 : (int)(10 + rnext(&r) % 1500); pl[s * G + g] = v; gl[s * G + g] = -(float)v / 10.0f; }
    }
    p = put_key(p, K_GT); p = put_desc(p, 2, 1); memcpy(p, gt, BCF_NS * 2); p += BCF_NS * 2;
    p = put_key(p, K_GQ); p = put_ints(p, gq, BCF_NS, 1);
    p = put_key(p, K_DP); p = put_ints(p, dp, BCF_NS, 1);
    p = put_key(p, K_AD); p = put_ints(p, ad, BCF_NS * n_allele, n_allele);
    p = put_key(p, K_PL); p = put_ints(p, pl, BCF_NS * G, G);
    p = put_key(p, K_GL); p = put_desc(p, G, 5); for (int i = 0; i < BCF_NS * G; i++) p = put_f32(p, gl[i]);
    size_t l_indiv = (size_t)(p - in);
    put32(dst, (uint32_t)(24 + l_shared)); put32(dst + 4, (uint32_t)l_indiv);
    put32(dst + 8, (uint32_t)tid); put32(dst + 12, (uint32_t)pos); put32(dst + 16, (uint32_t)all[0]);
    if (rnext(&r) % 100 == 0) put32(dst + 20, 0x7F800001u); else { float qf = (float)(rnext(&r) % 50000) / 10.0f; uint32_t b; memcpy(&b, &qf, 4); put32(dst + 20, b); }
    dst[24] = (uint8_t)n_info; dst[25] = 0; dst[26] = (uint8_t)n_allele; dst[27] = 0;
    put32(dst + 28, (uint32_t)BCF_NS | (6u << 24));
    return 32 + l_shared + l_indiv;
}

size_t synth_bcf_segment(uint64_t seed, uint64_t total_n, uint64_t rec0, uint64_t n, int with_header, int with_eof,
                         int level, int payload, int threads, uint8_t *out, size_t out_cap, uint64_t *stats) {
    uint8_t hdr[16384]; size_t hl = with_header ? gen_bcf_header(hdr) : 0;
    return segment_impl(gen_bcf_record, 1400, hdr, hl, seed, total_n, rec0, n, with_eof, level, payload, threads, out, out_cap, stats);
}
