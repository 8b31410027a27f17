/*
 * dhts_oracle.c -- CPU restatement of the DuckHTS read_bam scan path
 * (BGZF framing -> DEFLATE -> CRC-32 -> BAM record decode -> 13 core columns).
 *
 * TEST INFRASTRUCTURE ONLY (see dhts_oracle.h).  Every function cites the
 * reference file:line whose behaviour it restates; paths are relative to
 * /root/reference, "htslib/" = third_party/htslib/ (htslib 1.23).
 *
 * DEFLATE / CRC-32 live in the reference's third-party dependency zlib
 * (system zlib, unpinned; call sites htslib/bgzf.c:775-793).  They are
 * restated here from the published formats RFC 1951 / RFC 1952, not from zlib
 * source, and pinned in tests against CPython's zlib module.
 */
#include "dhts_oracle.h"
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ========================================================================
 * CRC-32 (RFC 1952 section 8; reference call site htslib/bgzf.c:793)
 * ======================================================================== */
static uint32_t crc_tab[256];
static int crc_tab_ready = 0;
static void crc_init(void) {
    for (uint32_t n = 0; n < 256; n++) {
        uint32_t c = n;
        for (int k = 0; k < 8; k++) c = (c & 1) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
        crc_tab[n] = c;
    }
    crc_tab_ready = 1;
}
uint32_t orc_crc32(uint32_t crc, const uint8_t *p, size_t n) {
    if (!crc_tab_ready) crc_init();
    uint32_t c = crc ^ 0xFFFFFFFFu;
    for (size_t i = 0; i < n; i++) c = crc_tab[(c ^ p[i]) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

/* ========================================================================
 * DEFLATE decoder (RFC 1951).  Canonical-code decode by the "count / first
 * code per length" method of RFC 1951 section 3.2.2.
 * ======================================================================== */
typedef struct {
    const uint8_t *in; size_t inlen, inpos;
    uint64_t bitbuf; int bitcnt;
    uint8_t *out; size_t outcap, outpos;
} inf_t;

typedef struct { uint16_t count[16]; uint16_t symbol[288]; } huff_t;

static int inf_bits(inf_t *s, int need, uint32_t *val) {
    while (s->bitcnt < need) {
        if (s->inpos >= s->inlen) return -1;            /* input exhausted */
        s->bitbuf |= (uint64_t)s->in[s->inpos++] << s->bitcnt;
        s->bitcnt += 8;
    }
    *val = (uint32_t)(s->bitbuf & ((1ull << need) - 1));
    s->bitbuf >>= need; s->bitcnt -= need;
    return 0;
}

static int huff_build(huff_t *h, const uint8_t *lens, int n) {
    uint16_t offs[16];
    memset(h->count, 0, sizeof(h->count));
    for (int i = 0; i < n; i++) h->count[lens[i]]++;
    if (h->count[0] == n) return 0;                     /* no codes: complete but unusable */
    int left = 1;
    for (int len = 1; len <= 15; len++) {
        left <<= 1; left -= h->count[len];
        if (left < 0) return -1;                        /* over-subscribed */
    }
    offs[1] = 0;
    for (int len = 1; len < 15; len++) offs[len + 1] = offs[len] + h->count[len];
    for (int i = 0; i < n; i++) if (lens[i]) h->symbol[offs[lens[i]]++] = (uint16_t)i;
    return left;                                        /* >0: incomplete */
}

static int huff_decode(inf_t *s, const huff_t *h) {
    int code = 0, first = 0, index = 0;
    for (int len = 1; len <= 15; len++) {
        uint32_t b;
        if (inf_bits(s, 1, &b) < 0) return -1;
        code |= (int)b;
        int count = h->count[len];
        if (code - count < first) return h->symbol[index + (code - first)];
        index += count; first += count; first <<= 1; code <<= 1;
    }
    return -2;                                          /* ran out of codes */
}

static const uint16_t LEN_BASE[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
static const uint16_t LEN_EXTRA[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
static const uint16_t DIST_BASE[30] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
static const uint16_t DIST_EXTRA[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};

static int inf_codes(inf_t *s, const huff_t *lc, const huff_t *dc) {
    for (;;) {
        int sym = huff_decode(s, lc);
        if (sym < 0) return -1;
        if (sym < 256) {
            if (s->outpos >= s->outcap) return -2;
            s->out[s->outpos++] = (uint8_t)sym;
        } else if (sym == 256) {
            return 0;
        } else {
            sym -= 257;
            if (sym >= 29) return -1;
            uint32_t eb;
            if (inf_bits(s, LEN_EXTRA[sym], &eb) < 0) return -1;
            int len = LEN_BASE[sym] + (int)eb;
            int ds = huff_decode(s, dc);
            if (ds < 0 || ds >= 30) return -1;
            if (inf_bits(s, DIST_EXTRA[ds], &eb) < 0) return -1;
            size_t dist = DIST_BASE[ds] + eb;
            if (dist > s->outpos) return -1;            /* distance too far back */
            if (s->outpos + (size_t)len > s->outcap) return -2;
            for (int i = 0; i < len; i++) { s->out[s->outpos] = s->out[s->outpos - dist]; s->outpos++; }
        }
    }
}

int orc_inflate_raw(const uint8_t *src, size_t slen, uint8_t *dst, size_t dcap, size_t *dlen) {
    inf_t s = { src, slen, 0, 0, 0, dst, dcap, 0 };
    static const uint8_t CL_ORDER[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};
    uint32_t last, type;
    do {
        if (inf_bits(&s, 1, &last) < 0 || inf_bits(&s, 2, &type) < 0) return -1;
        if (type == 0) {                                /* stored */
            s.bitbuf = 0; s.bitcnt = 0;                 /* discard to byte boundary */
            if (s.inpos + 4 > s.inlen) return -1;
            unsigned len = s.in[s.inpos] | (s.in[s.inpos + 1] << 8);
            unsigned nlen = s.in[s.inpos + 2] | (s.in[s.inpos + 3] << 8);
            s.inpos += 4;
            if ((len ^ 0xFFFF) != nlen) return -1;
            if (s.inpos + len > s.inlen) return -1;
            if (s.outpos + len > s.outcap) return -2;
            memcpy(s.out + s.outpos, s.in + s.inpos, len);
            s.inpos += len; s.outpos += len;
        } else if (type == 1) {                         /* fixed codes, RFC 1951 3.2.6 */
            uint8_t lens[288 + 30]; huff_t lc, dc; int i;
            for (i = 0; i < 144; i++) lens[i] = 8;
            for (; i < 256; i++) lens[i] = 9;
            for (; i < 280; i++) lens[i] = 7;
            for (; i < 288; i++) lens[i] = 8;
            huff_build(&lc, lens, 288);
            for (i = 0; i < 30; i++) lens[i] = 5;
            huff_build(&dc, lens, 30);
            int r = inf_codes(&s, &lc, &dc);
            if (r < 0) return r;
        } else if (type == 2) {                         /* dynamic codes, RFC 1951 3.2.7 */
            uint32_t nlen, ndist, ncode, v;
            uint8_t lens[320]; huff_t lc, dc, cl;
            if (inf_bits(&s, 5, &nlen) < 0 || inf_bits(&s, 5, &ndist) < 0 || inf_bits(&s, 4, &ncode) < 0) return -1;
            nlen += 257; ndist += 1; ncode += 4;
            if (nlen > 286 || ndist > 30) return -1;
            memset(lens, 0, 19);
            for (uint32_t i = 0; i < ncode; i++) { if (inf_bits(&s, 3, &v) < 0) return -1; lens[CL_ORDER[i]] = (uint8_t)v; }
            if (huff_build(&cl, lens, 19) != 0) return -1;   /* must be complete */
            uint32_t idx = 0;
            while (idx < nlen + ndist) {
                int sym = huff_decode(&s, &cl);
                if (sym < 0) return -1;
                if (sym < 16) lens[idx++] = (uint8_t)sym;
                else {
                    uint8_t rep_val = 0; uint32_t rep;
                    if (sym == 16) {
                        if (idx == 0) return -1;
                        rep_val = lens[idx - 1];
                        if (inf_bits(&s, 2, &rep) < 0) return -1; rep += 3;
                    } else if (sym == 17) { if (inf_bits(&s, 3, &rep) < 0) return -1; rep += 3; }
                    else { if (inf_bits(&s, 7, &rep) < 0) return -1; rep += 11; }
                    if (idx + rep > nlen + ndist) return -1;
                    while (rep--) lens[idx++] = rep_val;
                }
            }
            if (lens[256] == 0) return -1;              /* no end-of-block code */
            int r = huff_build(&lc, lens, (int)nlen);
            if (r < 0 || (r > 0 && nlen - lc.count[0] != 1)) return -1;
            r = huff_build(&dc, lens + nlen, (int)ndist);
            if (r < 0 || (r > 0 && ndist - dc.count[0] != 1)) return -1;
            r = inf_codes(&s, &lc, &dc);
            if (r < 0) return r;
        } else return -1;
    } while (!last);
    *dlen = s.outpos;
    return 0;
}

/* ---- optional system-zlib inflate for the cpu_baseline timing leg -------------------------
 * The reference inflates through system zlib (htslib/bgzf.c:775-793).  For an honest host
 * baseline the timing path may use the same library (dlopen libz.so.1) instead of the
 * readable-but-slow RFC restatement above; parity tests never use this path.            */
typedef struct { const unsigned char *next_in; unsigned avail_in; unsigned long total_in; unsigned char *next_out; unsigned avail_out;
                 unsigned long total_out; const char *msg; void *state; void *zalloc; void *zfree; void *opaque; int data_type;
                 unsigned long adler; unsigned long reserved; } orc_zstream;
static int (*z_inflateInit2_)(orc_zstream *, int, const char *, int);
static int (*z_inflate)(orc_zstream *, int);
static int (*z_inflateEnd)(orc_zstream *);
static unsigned long (*z_crc32)(unsigned long, const unsigned char *, unsigned);
static int z_ready = 0;
static int zlib_load(void) {
    if (z_ready) return z_ready > 0;
    void *h = dlopen("libz.so.1", RTLD_NOW);
    if (!h) { z_ready = -1; return 0; }
    z_inflateInit2_ = (int (*)(orc_zstream *, int, const char *, int))dlsym(h, "inflateInit2_");
    z_inflate = (int (*)(orc_zstream *, int))dlsym(h, "inflate");
    z_inflateEnd = (int (*)(orc_zstream *))dlsym(h, "inflateEnd");
    z_crc32 = (unsigned long (*)(unsigned long, const unsigned char *, unsigned))dlsym(h, "crc32");
    z_ready = (z_inflateInit2_ && z_inflate && z_inflateEnd && z_crc32) ? 1 : -1;
    return z_ready > 0;
}
static int use_zlib = 0;
int orc_use_system_zlib(int on) { use_zlib = on && zlib_load(); return use_zlib; }
static int zlib_inflate_raw(const uint8_t *src, size_t slen, uint8_t *dst, size_t dcap, size_t *dlen) {
    orc_zstream zs; memset(&zs, 0, sizeof(zs));
    zs.next_in = src; zs.avail_in = (unsigned)slen; zs.next_out = dst; zs.avail_out = (unsigned)dcap;
    if (z_inflateInit2_(&zs, -15, "1.2.11", (int)sizeof(zs)) != 0) return -1;
    int r = z_inflate(&zs, 4 /* Z_FINISH */);
    z_inflateEnd(&zs);
    if (r != 1 /* Z_STREAM_END */) return -1;
    *dlen = dcap - zs.avail_out;
    return 0;
}

/* ========================================================================
 * BGZF framing
 * ======================================================================== */
/* htslib/bgzf.c:896-903 check_header: 0 = BGZF, -1 = plain gzip, -2 = not gzip */
static int bgzf_check_header(const uint8_t *h) {
    if (h[0] != 31 || h[1] != 139 || h[2] != 8) return -2;
    return ((h[3] & 4) != 0 && (h[10] | (h[11] << 8)) == 6 && h[12] == 'B' && h[13] == 'C'
            && (h[14] | (h[15] << 8)) == 2) ? 0 : -1;
}

static const uint8_t BGZF_EOF[28] = {0x1f,0x8b,0x08,0x04,0,0,0,0,0,0xff,0x06,0,0x42,0x43,0x02,0,0x1b,0,0x03,0,0,0,0,0,0,0,0,0};

/* htslib/bgzf.c:1004-1239 bgzf_read_block (single-threaded arm) +
 * inflate_block 808-824 + bgzf_uncompress 762-805: walk the BSIZE chain,
 * inflate every block into a 64 KiB bound, compare CRC-32 (ISIZE is NOT
 * checked by htslib), skip empty blocks, stop the stream at the first error. */
int orc_bgzf_inflate_all(const uint8_t *file, size_t flen, orc_bgzf_t *out) {
    memset(out, 0, sizeof(*out));
    size_t cap = flen * 4 + 65536, nb_cap = 1024;
    out->data = (uint8_t *)malloc(cap);
    out->coff = (int64_t *)malloc(nb_cap * sizeof(int64_t));
    out->clen = (int32_t *)malloc(nb_cap * sizeof(int32_t));
    out->ulen = (int32_t *)malloc(nb_cap * sizeof(int32_t));
    size_t pos = 0;
    out->has_eof_marker = (flen >= 28 && memcmp(file + flen - 28, BGZF_EOF, 28) == 0);
    while (pos < flen) {
        if (flen - pos < 18) { out->status = -1; break; }             /* "Failed to read" header */
        int hc = bgzf_check_header(file + pos);
        if (hc != 0) { out->status = -1; break; }                      /* invalid / plain gzip: out of scope */
        int block_length = (file[pos + 16] | (file[pos + 17] << 8)) + 1;
        if (block_length < 18) { out->status = -1; break; }           /* bgzf.c:1199-1205 */
        if (pos + (size_t)block_length > flen) { out->status = -2; break; } /* short read */
        if (block_length < 26) { out->status = -3; break; }           /* no room for trailer: inflate fails */
        if (out->len + 65536 > cap) { cap = cap * 2 + 65536; out->data = (uint8_t *)realloc(out->data, cap); }
        size_t dlen = 0;
        int r = use_zlib ? zlib_inflate_raw(file + pos + 18, (size_t)block_length - 18, out->data + out->len, 65536, &dlen)
                         : orc_inflate_raw(file + pos + 18, (size_t)block_length - 18, out->data + out->len, 65536, &dlen);
        if (r < 0) { out->status = -3; break; }                       /* BGZF_ERR_ZLIB */
        const uint8_t *t = file + pos + block_length - 8;
        uint32_t crc = t[0] | (t[1] << 8) | (t[2] << 16) | ((uint32_t)t[3] << 24);
        uint32_t have = use_zlib ? (uint32_t)z_crc32(z_crc32(0, NULL, 0), out->data + out->len, (unsigned)dlen) : orc_crc32(0, out->data + out->len, dlen);
        if (have != crc) { out->status = -4; break; }                 /* BGZF_ERR_CRC */
        if ((size_t)out->n_blocks == nb_cap) {
            nb_cap *= 2;
            out->coff = (int64_t *)realloc(out->coff, nb_cap * sizeof(int64_t));
            out->clen = (int32_t *)realloc(out->clen, nb_cap * sizeof(int32_t));
            out->ulen = (int32_t *)realloc(out->ulen, nb_cap * sizeof(int32_t));
        }
        out->coff[out->n_blocks] = (int64_t)pos; out->clen[out->n_blocks] = block_length; out->ulen[out->n_blocks] = (int32_t)dlen;
        out->n_blocks++;
        out->len += dlen;                                              /* empty blocks contribute nothing (bgzf.c:1225-1227) */
        pos += (size_t)block_length;
    }
    return out->status;
}

void orc_bgzf_free(orc_bgzf_t *b) { free(b->data); free(b->coff); free(b->clen); free(b->ulen); memset(b, 0, sizeof(*b)); }

/* ========================================================================
 * column helpers
 * ======================================================================== */
static void sc_push(orc_strcol_t *c, const void *p, size_t len, int valid) {
    if (c->n + 1 >= c->cap_n) {
        c->cap_n = c->cap_n ? c->cap_n * 2 : 1024;
        c->off = (uint64_t *)realloc(c->off, (c->cap_n + 1) * sizeof(uint64_t));
        c->valid = (uint8_t *)realloc(c->valid, c->cap_n);
        if (c->n == 0) c->off[0] = 0;
    }
    if (c->nbytes + len + 1 > c->cap_bytes) {
        c->cap_bytes = (c->cap_bytes + len + 1) * 2;
        c->bytes = (uint8_t *)realloc(c->bytes, c->cap_bytes);
    }
    if (len) memcpy(c->bytes + c->nbytes, p, len);
    c->nbytes += len;
    c->valid[c->n] = (uint8_t)valid;
    c->n++;
    c->off[c->n] = c->nbytes;
}
static void sc_push_cstr(orc_strcol_t *c, const char *s) { sc_push(c, s, strlen(s), 1); }
static void sc_free(orc_strcol_t *c) { free(c->off); free(c->bytes); free(c->valid); memset(c, 0, sizeof(*c)); }

static inline uint32_t le32(const uint8_t *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }

/* ========================================================================
 * SAM text header -> @RG ID -> SM dictionary
 * (htslib/header.c:995-1075 parse_lines, 830-893 parse_noncomment_line,
 *  271-318 RG hash: first ID wins, 2282-2312 sam_hdr_find_tag_id: first SM
 *  tag, value must be non-empty).  Any malformed line fails the whole
 *  dictionary => every SAMPLE_ID NULL (find_tag_id returns -2).
 * ======================================================================== */
typedef struct { char *id; char *sm; } rg_ent_t;
typedef struct { rg_ent_t *e; int n; int ok; } rg_map_t;

static int is_alpha(char c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'); }

static void rg_map_build(rg_map_t *m, const char *text, size_t len) {
    memset(m, 0, sizeof(*m)); m->ok = 1;
    if (len < 3) { if (len != 0 && text[0] != '\0') m->ok = 0; return; }
    size_t i = 0;
    while (i < len - 3 && text[i] != '\0') {
        const char *h = text + i; size_t rem = len - i;
        if (h[0] != '@' || !is_alpha(h[1]) || !is_alpha(h[2])) { m->ok = 0; return; }
        if (rem < 3 || h[3] == '\n') { m->ok = 0; return; }
        int is_rg = (h[1] == 'R' && h[2] == 'G');
        int is_co = (h[1] == 'C' && h[2] == 'O');
        size_t j = 3;
        const char *id = NULL, *sm = NULL; size_t idl = 0, sml = 0; int sm_seen = 0;
        if (is_co) {
            if (rem == 3 || h[3] != '\t') { m->ok = 0; return; }
            for (j = 4; j < rem && h[j] != '\0' && h[j] != '\n'; j++) ;
        } else {
            do {
                if (j == rem || h[j] != '\t') { m->ok = 0; return; }
                size_t k = ++j;
                while (k < rem && h[k] != '\0' && h[k] != '\n' && h[k] != '\t') k++;
                if (k - j < 3 || h[j + 2] != ':') { m->ok = 0; return; }
                if (is_rg && h[j] == 'I' && h[j + 1] == 'D' && !id) { id = h + j + 3; idl = k - j - 3; }
                if (is_rg && h[j] == 'S' && h[j + 1] == 'M' && !sm_seen) { sm = h + j + 3; sml = k - j - 3; sm_seen = 1; }
                j = k;
            } while (j < rem && h[j] != '\0' && h[j] != '\n');
        }
        if (is_rg) {
            if (!id) { m->ok = 0; return; }             /* "@RG line with no ID" */
            int dup = 0;
            for (int q = 0; q < m->n; q++) if (strlen(m->e[q].id) == idl && memcmp(m->e[q].id, id, idl) == 0) dup = 1;
            if (!dup) {
                m->e = (rg_ent_t *)realloc(m->e, (m->n + 1) * sizeof(rg_ent_t));
                m->e[m->n].id = strndup(id, idl);
                m->e[m->n].sm = (sm && sml > 0) ? strndup(sm, sml) : NULL;
                m->n++;
            }
        }
        i += j + 1;
    }
}
static void rg_map_free(rg_map_t *m) { for (int i = 0; i < m->n; i++) { free(m->e[i].id); free(m->e[i].sm); } free(m->e); }

/* ========================================================================
 * BAM aux walk (htslib/sam.c:732-748 aux_type2size, 4785-4809 skip_aux,
 * 4811-4855 bam_aux_first/next/get)
 * ======================================================================== */
static int aux_type2size(uint8_t t) {
    switch (t) {
    case 'A': case 'c': case 'C': return 1;
    case 's': case 'S': return 2;
    case 'i': case 'I': case 'f': return 4;
    case 'd': return 8;
    case 'Z': case 'H': case 'B': return t;
    default: return 0;
    }
}
static const uint8_t *skip_aux(const uint8_t *s, const uint8_t *end) {
    if (s >= end) return end;
    int size = aux_type2size(*s); ++s;
    switch (size) {
    case 'Z': case 'H': { const uint8_t *z = (const uint8_t *)memchr(s, 0, (size_t)(end - s)); return z ? z + 1 : end; }
    case 'B': {
        if (end - s < 5) return NULL;
        size = aux_type2size(*s); ++s;
        uint32_t n = le32(s); s += 4;
        if (size == 0 || (uint64_t)(end - s) < (uint64_t)size * n) return NULL;
        return s + (size_t)size * n;
    }
    case 0: return NULL;
    default: if (end - s < size) return NULL; return s + size;
    }
}
/* returns pointer to the type byte of tag, NULL if absent; *bad=1 on corrupt aux */
static const uint8_t *aux_get(const uint8_t *aux, const uint8_t *end, const char tag[2], int *bad) {
    *bad = 0;
    if (end - aux <= 2) return NULL;
    const uint8_t *s = aux + 2;
    while (s) {
        if (s[-2] == (uint8_t)tag[0] && s[-1] == (uint8_t)tag[1]) {
            const uint8_t *e = skip_aux(s, end);
            if (!e) { *bad = 1; return NULL; }
            if ((*s == 'Z' || *s == 'H') && *(e - 1) != '\0') { *bad = 1; return NULL; }
            return s;
        }
        const uint8_t *next = skip_aux(s, end);
        if (!next) { *bad = 1; return NULL; }
        if (end - next <= 2) return NULL;
        s = next + 2;
    }
    return NULL;
}

/* ========================================================================
 * read_bam: header + record loop + column writers
 * ======================================================================== */
#define GROW(ptr, type, n, cap) do { if ((n) >= (cap)) { (cap) = (cap) ? (cap) * 2 : 4096; \
    b->flag = (uint16_t *)realloc(b->flag, (cap) * sizeof(uint16_t)); b->pos = (int64_t *)realloc(b->pos, (cap) * 8); \
    b->mapq = (int32_t *)realloc(b->mapq, (cap) * 4); b->pnext = (int64_t *)realloc(b->pnext, (cap) * 8); \
    b->tlen = (int64_t *)realloc(b->tlen, (cap) * 8); b->tid = (int32_t *)realloc(b->tid, (cap) * 4); \
    b->mtid = (int32_t *)realloc(b->mtid, (cap) * 4); b->rec_off = (int64_t *)realloc(b->rec_off, (cap) * 8); } } while (0)

static const char CIGAR_CH[] = "MIDNSHP=XB??????";       /* htslib/sam.h:112 BAM_CIGAR_STR + '?' (sam.h:131) */
static const char NT16[] = "=ACMGRSVTWYHKDBN";           /* htslib/hts.c:260 seq_nt16_str */
/* bam_cigar_type table 0x3C1A7 (htslib/sam.h:139-148): bit0 consumes query, bit1 consumes ref */
#define CIGAR_TYPE(op) ((0x3C1A7 >> ((op) << 1)) & 3)

/* per-row hook for the standard_tags table: receives the aux area as bam_aux_get sees it (CG spliced out after a long-CIGAR swap) */
static void (*g_aux_hook)(const uint8_t *aux, const uint8_t *end, void *ud) = NULL;
static void *g_aux_ud = NULL;

static void (*g_avail_wait)(size_t need) = NULL;    /* set only by orc_bam_scan_count_mt */
static int bam_decode_stream(const uint8_t *u, size_t ulen, orc_bam_t *b) {
    size_t p = 0;
    /* ---- header: htslib/sam.c:229-342 bam_hdr_read ---- */
    if (ulen < 4 || memcmp(u, "BAM\1", 4) != 0) return -10;
    p = 4;
    if (p + 4 > ulen) return -10;
    uint32_t l_text = le32(u + p); p += 4;
    if (p + l_text > ulen) return -10;
    b->text = (char *)malloc((size_t)l_text + 1); memcpy(b->text, u + p, l_text); b->text[l_text] = 0; b->l_text = l_text;
    p += l_text;
    if (p + 4 > ulen) return -10;
    int32_t n_ref = (int32_t)le32(u + p); p += 4;
    if (n_ref < 0) return -10;
    b->n_ref = n_ref;
    b->ref_len = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_ref > 0 ? n_ref : 1));
    for (int32_t i = 0; i < n_ref; i++) {
        if (p + 4 > ulen) return -10;
        int32_t l_name = (int32_t)le32(u + p); p += 4;
        if (l_name <= 0) return -10;
        if (p + (size_t)l_name + 4 > ulen) return -10;
        /* the name is used as a C string (sam_hdr_tid2name, header.c:2412): stop at first NUL */
        size_t nl = strnlen((const char *)u + p, (size_t)l_name);
        sc_push(&b->ref_names, u + p, nl, 1);
        p += (size_t)l_name;
        b->ref_len[i] = (int32_t)le32(u + p); p += 4;
    }
    b->first_rec_off = (int64_t)p;

    rg_map_t rgm; rg_map_build(&rgm, b->text, b->l_text);
    size_t cap = 0; int status = 0;
    char *tmp = NULL; size_t tmp_cap = 0;

    /* ---- record loop: src/bam_reader.c:747-1035 over htslib/sam.c:779-855 bam_read1 ---- */
    for (;;) {
        if (g_avail_wait) {                                           /* timing leg only: the bytes are being inflated by other threads */
            if (p + 4 <= ulen) { g_avail_wait(p + 4); size_t e_ = p + 4 + (size_t)(le32(u + p) & 0x7fffffffu); g_avail_wait(e_ < ulen ? e_ : ulen); }
        }
        if (p == ulen) { status = 0; break; }                        /* ret == -1: normal EOF */
        if (ulen - p < 4) { status = -2; break; }                    /* truncated */
        int32_t block_len = (int32_t)le32(u + p);
        if (block_len < 32) { status = -4; break; }
        if (ulen - p - 4 < 32) { status = -3; break; }
        const uint8_t *x = u + p + 4;
        int32_t tid = (int32_t)le32(x), pos = (int32_t)le32(x + 4);
        uint32_t x2 = le32(x + 8), x3 = le32(x + 12);
        uint32_t mapq = (x2 >> 8) & 0xff, l_qname = x2 & 0xff;
        uint32_t flag = x3 >> 16, n_cigar = x3 & 0xffff;
        int32_t l_qseq = (int32_t)le32(x + 16), mtid = (int32_t)le32(x + 20), mpos = (int32_t)le32(x + 24), isize = (int32_t)le32(x + 28);
        uint64_t body = (uint64_t)block_len - 32;
        if (l_qseq < 0 || l_qname < 1) { status = -4; break; }
        if (((uint64_t)n_cigar << 2) + l_qname + (((uint64_t)l_qseq + 1) >> 1) + (uint64_t)l_qseq > body) { status = -4; break; }
        if ((uint64_t)(ulen - p - 36) < body) { status = -4; break; } /* short read of the variable part */
        const uint8_t *d = x + 32, *dend = d + body;
        const uint8_t *qname = d;
        const uint8_t *cig = d + l_qname;
        const uint8_t *seq = cig + 4 * (size_t)n_cigar;
        const uint8_t *qual = seq + (((size_t)l_qseq + 1) >> 1);
        const uint8_t *aux = qual + l_qseq;

        /* long-CIGAR swap: htslib/sam.c:675-730 bam_tag2cigar */
        const uint8_t *real_cig = cig; uint32_t real_ncig = n_cigar;
        const uint8_t *cg_tag_start = NULL, *cg_tag_end = NULL;
        if (n_cigar > 0 && le32(cig) == (4u | ((uint32_t)l_qseq << 4)) && tid >= 0 && pos >= 0) {
            int bad; const uint8_t *CG = aux_get(aux, dend, "CG", &bad);
            if (!CG && bad) { status = -4; break; }
            if (CG && CG[0] == 'B' && (CG[1] == 'I' || CG[1] == 'i')) {
                uint32_t cgl = le32(CG + 2);
                if (cgl >= n_cigar && cgl < (1u << 29)) {
                    real_cig = CG + 6; real_ncig = cgl;
                    cg_tag_start = CG - 2; cg_tag_end = CG + 6 + 4 * (size_t)cgl;
                }
            }
        }
        /* CIGAR/qlen sanity: htslib/sam.c:842-852 */
        if (real_ncig > 0) {
            int64_t qlen = 0;
            for (uint32_t k = 0; k < real_ncig; k++) { uint32_t c = le32(real_cig + 4 * k); if (CIGAR_TYPE(c & 0xf) & 1) qlen += c >> 4; }
            if (l_qseq > 0 && !(flag & 4) && qlen != l_qseq) { status = -4; break; }
        }
        /* header range check: htslib/sam.c:4124-4134 sam_read1_bam */
        if (tid >= n_ref || tid < -1 || mtid >= n_ref || mtid < -1) { status = -3; break; }

        GROW(b, x, (size_t)b->n_rows, cap);
        int64_t r = b->n_rows;
        b->rec_off[r] = (int64_t)p;
        /* QNAME: C string up to first NUL (bam_reader.c:785-790; qname NUL fix-up sam.c:758-773,827-829) */
        sc_push(&b->qname, qname, strnlen((const char *)qname, l_qname), 1);
        b->flag[r] = (uint16_t)flag;                                  /* bam_reader.c:792-796 */
        b->tid[r] = tid; b->mtid[r] = mtid;
        if (tid >= 0) sc_push(&b->rname, b->ref_names.bytes + b->ref_names.off[tid], b->ref_names.off[tid + 1] - b->ref_names.off[tid], 1);
        else sc_push_cstr(&b->rname, "*");                            /* bam_reader.c:798-805 */
        b->pos[r] = (int64_t)pos + 1;                                 /* :807-811 */
        b->mapq[r] = (int32_t)mapq;                                   /* :813-817 */
        /* CIGAR text: bam_reader.c:819-834 + cigar_to_kstring 375-383 */
        if (real_ncig > 0) {
            size_t need = (size_t)real_ncig * 11 + 1;
            if (need > tmp_cap) { tmp_cap = need * 2; tmp = (char *)realloc(tmp, tmp_cap); }
            size_t l = 0;
            for (uint32_t k = 0; k < real_ncig; k++) { uint32_t c = le32(real_cig + 4 * k); l += (size_t)sprintf(tmp + l, "%d", (int)(c >> 4)); tmp[l++] = CIGAR_CH[c & 0xf]; }
            sc_push(&b->cigar, tmp, l, 1);
        } else sc_push_cstr(&b->cigar, "*");
        if (mtid >= 0) sc_push(&b->rnext, b->ref_names.bytes + b->ref_names.off[mtid], b->ref_names.off[mtid + 1] - b->ref_names.off[mtid], 1);
        else sc_push_cstr(&b->rnext, "*");                            /* :836-843 (name, never '=') */
        b->pnext[r] = (int64_t)mpos + 1;                              /* :845-849 */
        b->tlen[r] = (int64_t)isize;                                  /* :851-855 */
        /* SEQ :857-866 + seq_to_string 390-394; assigned via strlen API (no NUL can occur) */
        if ((size_t)l_qseq + 1 > tmp_cap) { tmp_cap = ((size_t)l_qseq + 1) * 2; tmp = (char *)realloc(tmp, tmp_cap); }
        if (l_qseq > 0) {
            for (int32_t k = 0; k < l_qseq; k++) tmp[k] = NT16[(seq[k >> 1] >> ((~k & 1) << 2)) & 0xf];
            sc_push(&b->seq, tmp, (size_t)l_qseq, 1);
        } else sc_push_cstr(&b->seq, "*");
        /* QUAL :868-877 + qual_to_string 400-404; NUL-terminated assign => truncates at byte 223 (+33 == 0) */
        if (l_qseq > 0 && qual[0] != 255) {
            size_t ql = 0;
            for (int32_t k = 0; k < l_qseq; k++) { uint8_t c = (uint8_t)(qual[k] + 33); if (c == 0) break; tmp[ql++] = (char)c; }
            sc_push(&b->qual, tmp, ql, 1);
        } else sc_push_cstr(&b->qual, "*");
        /* READ_GROUP_ID / SAMPLE_ID :879-918.  After a CG swap the CG tag is no longer part of aux (sam.c:716-720). */
        {
            const uint8_t *rg = NULL; int bad = 0;
            if (cg_tag_start) {
                /* walk the aux area with the CG tag spliced out */
                size_t l1 = (size_t)(cg_tag_start - aux), l2 = (size_t)(dend - cg_tag_end);
                uint8_t *sp = (uint8_t *)malloc(l1 + l2 + 1);
                memcpy(sp, aux, l1); memcpy(sp + l1, cg_tag_end, l2);
                const uint8_t *g = aux_get(sp, sp + l1 + l2, "RG", &bad);
                if (g && (g[0] == 'Z' || g[0] == 'H')) {
                    const char *z = (const char *)g + 1; size_t zl = strlen(z);
                    sc_push(&b->rg, z, zl, 1);
                    const char *sm = NULL;
                    if (rgm.ok) for (int q = 0; q < rgm.n; q++) if (strlen(rgm.e[q].id) == zl && memcmp(rgm.e[q].id, z, zl) == 0) { sm = rgm.e[q].sm; break; }
                    if (sm) sc_push_cstr(&b->sample, sm); else sc_push(&b->sample, "", 0, 0);
                } else { sc_push(&b->rg, "", 0, 0); sc_push(&b->sample, "", 0, 0); }
                free(sp);
            } else {
                rg = aux_get(aux, dend, "RG", &bad);
                if (rg && (rg[0] == 'Z' || rg[0] == 'H')) {             /* bam_aux2Z sam.c:5134-5141 */
                    const char *z = (const char *)rg + 1; size_t zl = strlen(z);
                    sc_push(&b->rg, z, zl, 1);
                    const char *sm = NULL;
                    if (rgm.ok) for (int q = 0; q < rgm.n; q++) if (strlen(rgm.e[q].id) == zl && memcmp(rgm.e[q].id, z, zl) == 0) { sm = rgm.e[q].sm; break; }
                    if (sm) sc_push_cstr(&b->sample, sm); else sc_push(&b->sample, "", 0, 0);
                } else { sc_push(&b->rg, "", 0, 0); sc_push(&b->sample, "", 0, 0); }
            }
        }
        if (g_aux_hook) {
            if (cg_tag_start) {
                size_t l1 = (size_t)(cg_tag_start - aux), l2 = (size_t)(dend - cg_tag_end);
                uint8_t *sp = (uint8_t *)malloc(l1 + l2 + 1);
                memcpy(sp, aux, l1); memcpy(sp + l1, cg_tag_end, l2);
                g_aux_hook(sp, sp + l1 + l2, g_aux_ud);
                free(sp);
            } else g_aux_hook(aux, dend, g_aux_ud);
        }
        b->n_rows++;
        p += 4 + (size_t)block_len;
    }
    free(tmp);
    rg_map_free(&rgm);
    b->status = status;
    return status;
}

int orc_bam_read(const uint8_t *file, size_t flen, orc_bam_t *out) {
    memset(out, 0, sizeof(*out));
    orc_bgzf_t z;
    orc_bgzf_inflate_all(file, flen, &z);
    /* a BGZF-level error ends the byte stream where it occurred; records fully
     * before it are still returned and the straddling read fails (bgzf.c:1241-1291). */
    int st = bam_decode_stream(z.data, z.len, out);
    if (st == 0 && z.status < 0) out->status = z.status * 100;
    orc_bgzf_free(&z);
    return out->status;
}

int orc_bam_read_path(const char *path, orc_bam_t *out) {
    FILE *f = fopen(path, "rb");
    if (!f) { memset(out, 0, sizeof(*out)); out->status = -1000; return -1000; }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t *buf = (uint8_t *)malloc((size_t)n + 1);
    if (fread(buf, 1, (size_t)n, f) != (size_t)n) { fclose(f); free(buf); memset(out, 0, sizeof(*out)); out->status = -1000; return -1000; }
    fclose(f);
    int r = orc_bam_read(buf, (size_t)n, out);
    free(buf);
    return r;
}

void orc_bam_free(orc_bam_t *b) {
    free(b->flag); free(b->pos); free(b->mapq); free(b->pnext); free(b->tlen); free(b->tid); free(b->mtid); free(b->rec_off);
    sc_free(&b->qname); sc_free(&b->rname); sc_free(&b->cigar); sc_free(&b->rnext); sc_free(&b->seq); sc_free(&b->qual);
    sc_free(&b->rg); sc_free(&b->sample); sc_free(&b->ref_names); free(b->ref_len); free(b->text);
    memset(b, 0, sizeof(*b));
}

int64_t orc_bam_scan_count(const uint8_t *file, size_t flen, int *status) {
    orc_bam_t b; orc_bam_read(file, flen, &b);
    int64_t n = b.n_rows; if (status) *status = b.status;
    orc_bam_free(&b);
    return n;
}


/* Column digests of a scan (bench.py's in-run parity sample): CRC-32 (RFC 1952) of each column's canonical bytes, so that a multi-
 * million-row sample can be compared without materialising Python objects.  Layout of out[32]:
 *   0 n_rows, 1 status, 2 FLAG u16[], 3 tid i32[] (RNAME ids), 4 POS i64[], 5 MAPQ i32[], 6 mtid i32[] (RNEXT ids), 7 PNEXT i64[],
 *   8 TLEN i64[]; string columns as (lengths u32[], bytes): 9,10 QNAME  11,12 CIGAR  13,14 SEQ  15,16 QUAL;
 *   17,18,19 READ_GROUP_ID validity u8[], lengths, bytes;  20,21,22 SAMPLE_ID likewise. */
static uint32_t dg_crc(const void *p, size_t n) {
    uint32_t c = 0; const uint8_t *q = (const uint8_t *)p;
    while (n) { size_t k = n > (1u << 30) ? (1u << 30) : n; c = orc_crc32(c, q, k); q += k; n -= k; }
    return c;
}
static void dg_str(const orc_strcol_t *c, size_t n, uint32_t *len_crc, uint32_t *byte_crc, uint32_t *valid_crc) {
    uint32_t *l = (uint32_t *)malloc((n ? n : 1) * 4);
    for (size_t i = 0; i < n; i++) l[i] = (uint32_t)(c->off[i + 1] - c->off[i]);
    *len_crc = dg_crc(l, n * 4); *byte_crc = dg_crc(c->bytes, (size_t)c->off[n]);
    if (valid_crc) *valid_crc = dg_crc(c->valid, n);
    free(l);
}
int orc_bam_digest(const uint8_t *file, size_t flen, uint32_t *out) {
    orc_bam_t b; orc_bam_read(file, flen, &b);
    size_t n = (size_t)b.n_rows;
    memset(out, 0, 32 * 4);
    out[0] = (uint32_t)n; out[1] = (uint32_t)b.status;
    out[2] = dg_crc(b.flag, n * 2); out[3] = dg_crc(b.tid, n * 4); out[4] = dg_crc(b.pos, n * 8); out[5] = dg_crc(b.mapq, n * 4);
    out[6] = dg_crc(b.mtid, n * 4); out[7] = dg_crc(b.pnext, n * 8); out[8] = dg_crc(b.tlen, n * 8);
    dg_str(&b.qname, n, &out[9], &out[10], NULL); dg_str(&b.cigar, n, &out[11], &out[12], NULL);
    dg_str(&b.seq, n, &out[13], &out[14], NULL); dg_str(&b.qual, n, &out[15], &out[16], NULL);
    dg_str(&b.rg, n, &out[18], &out[19], &out[17]); dg_str(&b.sample, n, &out[21], &out[22], &out[20]);
    int st = b.status;
    orc_bam_free(&b);
    return st;
}

/* ------------------------------------------------------------------------
 * Timing leg only (bench.py cpu_baseline): the same scan with the BGZF blocks inflated by `n_threads` worker threads while ONE thread
 * decodes records and materialises the columns -- the shape of the reference's read path, which opens its file with
 * hts_set_threads(fp, 2) (src/bam_reader.c:625 -> hts.c:1922-1932 -> bgzf_mt, bgzf.c:1781-1800) and scans on one DuckDB thread when
 * there is no index (bam_reader.c:582-585).  Blocks are placed by their ISIZE fields; if any block disagrees (or fails) the routine
 * falls back to the sequential restatement, so the result is always the oracle's.
 * ------------------------------------------------------------------------ */
#include <pthread.h>
#include <sched.h>
typedef struct {
    const uint8_t *file; const size_t *coff; const uint32_t *clen; const uint64_t *uoff; size_t nb; uint8_t *out;
    size_t ngroups; volatile int *done; volatile long next; volatile int failed;
} mt_job_t;
#define MT_GROUP 16
static mt_job_t *g_mt = NULL; static size_t g_mt_front_group = 0; static size_t g_mt_avail = 0;
static void *mt_worker(void *arg) {
    mt_job_t *j = (mt_job_t *)arg;
    for (;;) {
        long g = __atomic_fetch_add(&j->next, 1, __ATOMIC_RELAXED);
        if ((size_t)g >= j->ngroups || j->failed) break;
        size_t b0 = (size_t)g * MT_GROUP, b1 = b0 + MT_GROUP < j->nb ? b0 + MT_GROUP : j->nb;
        for (size_t b = b0; b < b1; b++) {
            size_t want = (size_t)(j->uoff[b + 1] - j->uoff[b]), dlen = 0;
            const uint8_t *blk = j->file + j->coff[b];
            int r = use_zlib ? zlib_inflate_raw(blk + 18, (size_t)j->clen[b] - 18, j->out + j->uoff[b], want, &dlen)
                             : orc_inflate_raw(blk + 18, (size_t)j->clen[b] - 18, j->out + j->uoff[b], want, &dlen);
            const uint8_t *t = blk + j->clen[b] - 8;
            uint32_t crc = t[0] | (t[1] << 8) | (t[2] << 16) | ((uint32_t)t[3] << 24);
            uint32_t have = use_zlib ? (uint32_t)z_crc32(z_crc32(0, NULL, 0), j->out + j->uoff[b], (unsigned)dlen) : orc_crc32(0, j->out + j->uoff[b], dlen);
            if (r < 0 || dlen != want || have != crc) { j->failed = 1; break; }
        }
        __atomic_store_n(&j->done[g], 1, __ATOMIC_RELEASE);
    }
    return NULL;
}
static void mt_wait(size_t need) {
    mt_job_t *j = g_mt;
    while (g_mt_avail < need && g_mt_front_group < j->ngroups && !j->failed) {
        if (__atomic_load_n(&j->done[g_mt_front_group], __ATOMIC_ACQUIRE)) {
            size_t b1 = (g_mt_front_group + 1) * MT_GROUP < j->nb ? (g_mt_front_group + 1) * MT_GROUP : j->nb;
            g_mt_avail = (size_t)j->uoff[b1]; g_mt_front_group++;
        } else sched_yield();
    }
}
int64_t orc_bam_scan_count_mt(const uint8_t *file, size_t flen, int n_threads, int *status) {
    if (n_threads < 1) return orc_bam_scan_count(file, flen, status);
    /* BSIZE chain (bgzf.c:1155-1236) */
    size_t cap = 1024, nb = 0, pos = 0; int clean = 1;
    size_t *coff = (size_t *)malloc(cap * sizeof(size_t)); uint32_t *clen = (uint32_t *)malloc(cap * 4); uint64_t *uoff = (uint64_t *)malloc((cap + 1) * 8);
    uoff[0] = 0;
    while (pos < flen) {
        if (flen - pos < 18 || bgzf_check_header(file + pos) != 0) { clean = 0; break; }
        size_t bl = (size_t)(file[pos + 16] | (file[pos + 17] << 8)) + 1;
        if (bl < 26 || pos + bl > flen) { clean = 0; break; }
        if (nb == cap) { cap *= 2; coff = (size_t *)realloc(coff, cap * sizeof(size_t)); clen = (uint32_t *)realloc(clen, cap * 4); uoff = (uint64_t *)realloc(uoff, (cap + 1) * 8); }
        const uint8_t *t = file + pos + bl - 4;
        uint32_t isz = t[0] | (t[1] << 8) | (t[2] << 16) | ((uint32_t)t[3] << 24);
        if (isz > 65536) { clean = 0; break; }
        coff[nb] = pos; clen[nb] = (uint32_t)bl; uoff[nb + 1] = uoff[nb] + isz; nb++;
        pos += bl;
    }
    int64_t n = -1;
    if (clean && nb > 0) {
        mt_job_t j; memset(&j, 0, sizeof(j));
        j.file = file; j.coff = coff; j.clen = clen; j.uoff = uoff; j.nb = nb; j.out = (uint8_t *)malloc((size_t)uoff[nb] + 64);
        j.ngroups = (nb + MT_GROUP - 1) / MT_GROUP; j.done = (volatile int *)calloc(j.ngroups, sizeof(int));
        pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
        for (int t = 0; t < n_threads; t++) pthread_create(&th[t], NULL, mt_worker, &j);
        g_mt = &j; g_mt_front_group = 0; g_mt_avail = 0;
        mt_wait((size_t)uoff[nb] < (1u << 20) ? (size_t)uoff[nb] : (1u << 20));       /* the header (bench files: a few KB) */
        orc_bam_t b; memset(&b, 0, sizeof(b));
        g_avail_wait = mt_wait;
        int st = bam_decode_stream(j.out, (size_t)uoff[nb], &b);
        g_avail_wait = NULL;
        for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
        if (!j.failed) { n = b.n_rows; if (status) *status = st == 0 ? b.status : st; }
        orc_bam_free(&b);
        free(th); free((void *)j.done); free(j.out); g_mt = NULL;
    }
    free(coff); free(clen); free(uoff);
    if (n < 0) return orc_bam_scan_count(file, flen, status);
    return n;
}

/* ========================================================================
 * standard_tags := true  (src/bam_reader.c:54-70 table, 88-104 types, 920-966 writers over bam_aux_get sam.c:4834-4855,
 * bam_aux2i 5020-5034 (non-integer types yield 0), bam_aux2A 5118-5126 (non-'A' yields NUL => ""), bam_aux2Z 5134-5141
 * (non Z/H => NULL), bam_auxB_len 5143-5150 (non-'B' => 0 elements), bam_auxB2i 5152-5160, bam_auxB2f 5162-5171)
 * ======================================================================== */
#include "orc_cols.h"
typedef struct { const char *tag; char type, subtype; } std_tag_t;
static const std_tag_t STD_TAGS[] = {
    {"AM",'i',0},{"AS",'i',0},{"BC",'Z',0},{"BQ",'Z',0},{"BZ",'Z',0},{"CB",'Z',0},{"CC",'Z',0},{"CG",'B','I'},{"CM",'i',0},{"CO",'Z',0},{"CP",'i',0},{"CQ",'Z',0},
    {"CR",'Z',0},{"CS",'Z',0},{"CT",'Z',0},{"CY",'Z',0},{"E2",'Z',0},{"FI",'i',0},{"FS",'Z',0},{"FZ",'B','S'},{"H0",'i',0},{"H1",'i',0},{"H2",'i',0},{"HI",'i',0},
    {"IH",'i',0},{"LB",'Z',0},{"MC",'Z',0},{"MD",'Z',0},{"MI",'Z',0},{"ML",'B','C'},{"MM",'Z',0},{"MN",'i',0},{"MQ",'i',0},{"NH",'i',0},{"NM",'i',0},{"OA",'Z',0},
    {"OC",'Z',0},{"OP",'i',0},{"OQ",'Z',0},{"OX",'Z',0},{"PG",'Z',0},{"PQ",'i',0},{"PT",'Z',0},{"PU",'Z',0},{"Q2",'Z',0},{"QT",'Z',0},{"QX",'Z',0},{"R2",'Z',0},
    {"RG",'Z',0},{"RX",'Z',0},{"SA",'Z',0},{"SM",'i',0},{"TC",'i',0},{"TS",'A',0},{"U2",'Z',0},{"UQ",'i',0},{NULL,0,0}};
#define N_STD_TAGS 56

static int64_t aux_int_val(uint8_t type, const uint8_t *s, uint32_t idx, int *ok) {       /* get_int_aux_val sam.c:4997-5018 */
    *ok = 1;
    switch (type) {
    case 'c': return (int8_t)s[idx];
    case 'C': return s[idx];
    case 's': return (int16_t)(s[2 * idx] | s[2 * idx + 1] << 8);
    case 'S': return (uint16_t)(s[2 * idx] | s[2 * idx + 1] << 8);
    case 'i': return (int32_t)le32(s + 4 * (size_t)idx);
    case 'I': return (uint32_t)le32(s + 4 * (size_t)idx);
    default: *ok = 0; return 0;
    }
}

static void std_tags_row(const uint8_t *aux, const uint8_t *end, void *ud) {
    col_t *cols = (col_t *)ud;
    for (int t = 0; t < N_STD_TAGS; t++) {
        col_t *c = &cols[t];
        int bad; const uint8_t *a = aux_get(aux, end, STD_TAGS[t].tag, &bad);
        if (!a) { col_null(c); continue; }
        int ok;
        switch (STD_TAGS[t].type) {
        case 'A': { char ch = a[0] == 'A' ? (char)a[1] : 0; col_str(c, &ch, ch ? 1 : 0); break; }     /* assigned through the NUL-terminated API */
        case 'Z': if (a[0] == 'Z' || a[0] == 'H') col_cstr(c, (const char *)a + 1); else col_null(c); break;
        case 'i': { int64_t v = aux_int_val(a[0], a + 1, 0, &ok); col_fixed(c, (uint64_t)v, 1); break; }
        case 'B': {
            uint32_t len = a[0] == 'B' ? le32(a + 2) : 0;
            list_begin(c);
            for (uint32_t i = 0; i < len; i++) {
                if (a[1] == 'f' || a[1] == 'd') {                        /* bam_assign_list_double writes doubles into the BIGINT child */
                    double d = 0.0;
                    if (a[1] == 'f') { uint32_t bits = le32(a + 6 + 4 * (size_t)i); float f; memcpy(&f, &bits, 4); d = f; }
                    uint64_t b; memcpy(&b, &d, 8); list_fixed(c, b);
                } else list_fixed(c, (uint64_t)aux_int_val(a[1], a + 6, i, &ok));
            }
            list_end(c);
            break;
        }
        }
    }
}

/* canonical blob (layout of orc_bcf_read) of the 56 standard-tag columns of every row orc_bam_read would return */
int orc_bam_read_std_tags(const uint8_t *file, size_t flen, uint8_t **blob, size_t *blob_len) {
    col_t cols[N_STD_TAGS];
    for (int t = 0; t < N_STD_TAGS; t++)
        col_init(&cols[t], STD_TAGS[t].tag, STD_TAGS[t].type == 'i' || STD_TAGS[t].type == 'B' ? T_BIGINT : T_VARCHAR, STD_TAGS[t].type == 'B');
    g_aux_hook = std_tags_row; g_aux_ud = cols;
    orc_bam_t b; int st = orc_bam_read(file, flen, &b);
    g_aux_hook = NULL; g_aux_ud = NULL;
    buf_t o = {0};
    uint32_t nc = N_STD_TAGS; uint64_t nr = (uint64_t)b.n_rows, fr = (uint64_t)b.first_rec_off; int32_t s32 = st; uint32_t ns = 0;
    buf_push(&o, &nc, 4); buf_push(&o, &nr, 8); buf_push(&o, &s32, 4); buf_push(&o, &fr, 8); buf_push(&o, &ns, 4);
    for (int t = 0; t < N_STD_TAGS; t++) { ser_col(&o, &cols[t], b.n_rows); col_free(&cols[t]); }
    uint64_t zero = 0; buf_push(&o, &zero, 8);
    *blob = o.p; *blob_len = o.n;
    orc_bam_free(&b);
    return st;
}


/* ========================================================================
 * auxiliary_tags := true  (src/bam_reader.c:967-1027: bam_aux_first / bam_aux_next walk sam.c:4811-4832, tags that are standard
 * columns excluded when standard_tags is on; values rendered by bam_aux_to_string bam_reader.c:140-183 and assigned through the
 * NUL-terminated API).  Blob = two LIST(VARCHAR) columns AUX_KEYS / AUX_VALUES sharing list offsets; the MAP cell is NULL when the
 * row lists no tag.  A tag whose value is corrupt ends the walk and is listed with an empty value (the reference renders bytes
 * beyond the field there).
 * ======================================================================== */
typedef struct { col_t *keys, *vals; int excl_std; } auxmap_ud;
static void aux_map_row(const uint8_t *aux, const uint8_t *end, void *ud_) {
    auxmap_ud *ud = (auxmap_ud *)ud_;
    col_t *K = ud->keys, *V = ud->vals;
    uint64_t start = K->child_n; int n = 0;
    if (end - aux > 2) {
        const uint8_t *s = aux + 2;
        for (;;) {
            int skip = 0;
            if (ud->excl_std) for (int t = 0; t < N_STD_TAGS; t++) if (s[-2] == (uint8_t)STD_TAGS[t].tag[0] && s[-1] == (uint8_t)STD_TAGS[t].tag[1]) { skip = 1; break; }
            const uint8_t *e = skip_aux(s, end);
            if (!skip) {
                char key[3] = { (char)s[-2], (char)s[-1], 0 };
                list_str(K, key, strlen(key));
                char tmp[64]; buf_t v = {0};
                if (e) {
                    int ok; uint8_t ty = s[0];
                    if (ty == 'A') { char ch = (char)s[1]; buf_push(&v, &ch, 1); }
                    else if (ty == 'c' || ty == 'C' || ty == 's' || ty == 'S' || ty == 'i' || ty == 'I') { int l = snprintf(tmp, sizeof tmp, "%lld", (long long)aux_int_val(ty, s + 1, 0, &ok)); buf_push(&v, tmp, (size_t)l); }
                    else if (ty == 'f') { uint32_t b = le32(s + 1); float f; memcpy(&f, &b, 4); int l = snprintf(tmp, sizeof tmp, "%g", (double)f); buf_push(&v, tmp, (size_t)l); }
                    else if (ty == 'd') { double d; memcpy(&d, s + 1, 8); int l = snprintf(tmp, sizeof tmp, "%g", d); buf_push(&v, tmp, (size_t)l); }
                    else if (ty == 'Z' || ty == 'H') { const uint8_t *z = s + 1; size_t l = 0; while (z + l < e && z[l]) l++; buf_push(&v, z, l); }
                    else if (ty == 'B') {
                        uint8_t sub = s[1]; uint32_t len = le32(s + 2);
                        buf_push(&v, &sub, 1);
                        for (uint32_t i = 0; i < len; i++) {
                            int l;
                            if (sub == 'f') { uint32_t b = le32(s + 6 + 4 * (size_t)i); float f; memcpy(&f, &b, 4); l = snprintf(tmp, sizeof tmp, ",%g", (double)f); }
                            else if (sub == 'd') l = snprintf(tmp, sizeof tmp, ",%g", 0.0);
                            else l = snprintf(tmp, sizeof tmp, ",%lld", (long long)aux_int_val(sub, s + 6, i, &ok));
                            buf_push(&v, tmp, (size_t)l);
                        }
                    }
                }
                size_t vl = 0; while (vl < v.n && v.p[vl]) vl++;                       /* assigned as a C string */
                list_str(V, v.p ? (const char *)v.p : "", vl);
                free(v.p);
                n++;
            }
            if (!e) break;
            if (end - e <= 2) break;
            s = e + 2;
        }
    }
    g_list_start = start;
    if (n) { list_end(K); g_list_start = start; list_end(V); }
    else { col_null(K); col_null(V); }
}

int orc_bam_read_aux_map(const uint8_t *file, size_t flen, int exclude_standard, uint8_t **blob, size_t *blob_len) {
    col_t cols[2];
    col_init(&cols[0], "AUX_KEYS", T_VARCHAR, 1); col_init(&cols[1], "AUX_VALUES", T_VARCHAR, 1);
    auxmap_ud ud = { &cols[0], &cols[1], exclude_standard };
    g_aux_hook = aux_map_row; g_aux_ud = &ud;
    orc_bam_t b; int st = orc_bam_read(file, flen, &b);
    g_aux_hook = NULL; g_aux_ud = NULL;
    buf_t o = {0};
    uint32_t nc = 2; uint64_t nr = (uint64_t)b.n_rows, fr = (uint64_t)b.first_rec_off; int32_t s32 = st; uint32_t ns = 0;
    buf_push(&o, &nc, 4); buf_push(&o, &nr, 8); buf_push(&o, &s32, 4); buf_push(&o, &fr, 8); buf_push(&o, &ns, 4);
    for (int t = 0; t < 2; t++) { ser_col(&o, &cols[t], b.n_rows); col_free(&cols[t]); }
    uint64_t zero = 0; buf_push(&o, &zero, 8);
    *blob = o.p; *blob_len = o.n;
    orc_bam_free(&b);
    return st;
}
