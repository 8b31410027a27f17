/*
 * bcf_oracle.c -- CPU restatement of the read_bcf scan path (TEST INFRASTRUCTURE ONLY, see dhts_oracle.h).
 *
 * Follows, by reading (nothing is copied; htslib = third_party/htslib in the reference tree):
 *   header      bcf_hdr_read vcf.c:1710-1769, bcf_hdr_parse 1410-1489, bcf_hdr_parse_line 653-789,
 *               bcf_hdr_register_hrec 831-1024 (+ bcf_hdr_set_idx 796-828), bcf_hdr_parse_sample_line 286-314
 *   framing     bcf_read1_core vcf.c:1874-1911, bcf_record_check 2040-2212, updatephasing 1985-2029
 *   decode      bcf_unpack vcf.c:4234-4302, bcf_fmt_array 3036-3079, bcf_get_info_values 6056-6138,
 *               bcf_get_format_string 6140-6177, bcf_get_format_values 6179-6248
 *   schema      src/bcf_reader.c:540-760 + src/include/vcf_types.h (spec corrections, is_list rule)
 *   writers     src/bcf_reader.c:1381-1982 (core, INFO, FORMAT wide/tidy, GT text)
 *
 * Mode restated: sequential scan (no index), i.e. every record in file order until EOF or the first bad record
 * (bcf_reader.c:1319-1349: ret < 0 ends the scan silently).
 *   VEP_*       src/vep_parser.c:33-182 (tag detection, "Format: a|b|c" field list, type inference), 207-326 (record parse),
 *               src/bcf_reader.c:582-603 (columns behind FILTER), 1370-1373 + 1463-1541 (LIST per transcript, NULL elements)
 *
 * Parity pinning: tests/golden/vcf_file.bcf against its upstream text form vcf_file.vcf and the expectations of
 * test/sql/duckhts.test:28-84 (see tests/test_oracle_bcf.py).
 */
#include "dhts_oracle.h"
#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>

#include "orc_cols.h"

/* ------------------------------------------------------------------ header ------------------------------------- */
enum { HL_FLT = 0, HL_INFO, HL_FMT, HL_CTG, HL_STR, HL_GEN };
enum { HT_FLAG = 0, HT_INT, HT_REAL, HT_STR };
enum { VL_FIXED = 0, VL_VAR, VL_A, VL_G, VL_R, VL_P, VL_LA, VL_LG, VL_LR, VL_M };

typedef struct { char *key; int has[3]; int type[3]; int vl[3]; char *info_desc; } dict_ent;      /* key == NULL: hole; info_desc: Description of the INFO line */
typedef struct {
    dict_ent *ids; int n_ids;
    char **ctg; int n_ctg;
    char **smp; int n_smp;
    int version;
} hdr_t;

static int dict_find(const hdr_t *h, const char *key)
{
    for (int i = 0; i < h->n_ids; i++) if (h->ids[i].key && !strcmp(h->ids[i].key, key)) return i;
    return -1;
}
static int ctg_find(const hdr_t *h, const char *key)
{
    for (int i = 0; i < h->n_ctg; i++) if (h->ctg[i] && !strcmp(h->ctg[i], key)) return i;
    return -1;
}
static char *dupn(const char *s, size_t n) { char *r = (char *)malloc(n + 1); memcpy(r, s, n); r[n] = 0; return r; }

typedef struct { char *key; char *value; int nkeys; char **keys, **vals; } hrec_t;
static void hrec_free(hrec_t *r)
{
    if (!r) return;
    for (int i = 0; i < r->nkeys; i++) { free(r->keys[i]); free(r->vals[i]); }
    free(r->keys); free(r->vals); free(r->key); free(r->value); free(r);
}
static int is_escaped(const char *min, const char *str) { int n = 0; while (--str >= min && *str == '\\') n++; return n % 2; }

/* vcf.c:653-789.  Returns the record (or NULL) and the line length in *len (0: not a ## line, <0 fatal). */
static hrec_t *parse_line(const char *line, int *len)
{
    const char *p = line;
    if (p[0] != '#' || p[1] != '#') { *len = 0; return NULL; }
    p += 2;
    const char *q = p;
    while (*q && *q != '=' && *q != '\n') q++;
    ptrdiff_t n = q - p;
    hrec_t *r = NULL;
    if (*q != '=' || !n) goto malformed;
    r = (hrec_t *)calloc(1, sizeof *r);
    r->key = dupn(p, (size_t)n);
    p = ++q;
    if (*p != '<') {
        while (*q && *q != '\n') q++;
        r->value = dupn(p, (size_t)(q - p));
        *len = (int)(q - line) + (*q ? 1 : 0);
        return r;
    }
    int nopen = 1;
    while (*q && *q != '\n' && nopen > 0) {
        p = ++q;
        while (*q && *q == ' ') { p++; q++; }
        if (p == q && *q && (isalpha((unsigned char)*q) || *q == '_')) {
            q++;
            while (*q && (isalnum((unsigned char)*q) || *q == '_' || *q == '.')) q++;
        }
        n = q - p;
        int m = 0;
        while (*q && *q == ' ') { q++; m++; }
        if (*q != '=' || !n) goto malformed;
        r->keys = (char **)realloc(r->keys, sizeof(char *) * (size_t)(r->nkeys + 1));
        r->vals = (char **)realloc(r->vals, sizeof(char *) * (size_t)(r->nkeys + 1));
        r->keys[r->nkeys] = dupn(p, (size_t)(q - p - m));
        r->vals[r->nkeys] = NULL;
        r->nkeys++;
        p = ++q;
        while (*q && *q == ' ') { p++; q++; }
        int quoted = 0; char ending = 0;
        if (*p == '"') { quoted = 1; ending = '"'; p++; }
        else if (*p == '[') { quoted = 1; ending = ']'; }
        if (quoted) q++;
        while (*q && *q != '\n') {
            if (quoted) { if (*q == ending && !is_escaped(p, q)) break; }
            else {
                if (*q == '<') nopen++;
                if (*q == '>') nopen--;
                if (!nopen) break;
                if (*q == ',' && nopen == 1) break;
            }
            q++;
        }
        const char *e = q;
        if (quoted && ending == ']') {
            if (*q == ending) { e++; q++; quoted = 0; }
            else { hrec_free(r); *len = -1; return NULL; }
        }
        while (e > p && e[-1] == ' ') e--;
        r->vals[r->nkeys - 1] = dupn(p, (size_t)(e - p));
        if (quoted && *q == ending) q++;
        if (*q == '>') { if (nopen) nopen--; q++; }
    }
    while (*q && *q != '\n') q++;
    *len = (int)(q - line) + (*q ? 1 : 0);
    return r;
malformed:
    while (*q && *q != '\n') q++;
    *len = (int)(q - line) + (*q ? 1 : 0);
    hrec_free(r);
    return NULL;
}

static int hrec_find_ci(const hrec_t *r, const char *key)
{
    for (int i = 0; i < r->nkeys; i++) if (!strcasecmp(key, r->keys[i])) return i;
    return -1;
}

static int set_idx_ctg(hdr_t *h, int idx, char *name)
{
    if (idx == -1) idx = h->n_ctg;
    else if (idx < h->n_ctg && h->ctg[idx]) return -1;                        /* conflicting IDX */
    if (idx >= h->n_ctg) {
        h->ctg = (char **)realloc(h->ctg, sizeof(char *) * (size_t)(idx + 1));
        for (int i = h->n_ctg; i <= idx; i++) h->ctg[i] = NULL;
        h->n_ctg = idx + 1;
    }
    h->ctg[idx] = name;
    return 0;
}
static int set_idx_id(hdr_t *h, int idx, char *name, int *out)
{
    if (idx == -1) idx = h->n_ids;
    else if (idx < h->n_ids && h->ids[idx].key) return -1;
    if (idx >= h->n_ids) {
        h->ids = (dict_ent *)realloc(h->ids, sizeof(dict_ent) * (size_t)(idx + 1));
        memset(h->ids + h->n_ids, 0, sizeof(dict_ent) * (size_t)(idx + 1 - h->n_ids));
        h->n_ids = idx + 1;
    }
    h->ids[idx].key = name;
    *out = idx;
    return 0;
}

static int parse_idx(const char *s, int *idx)                                 /* vcf.c:873-886 / 918-927 */
{
    char *end; long v = strtol(s, &end, 10);
    if (*end || v < 0 || v >= 2147483646L) return -1;
    *idx = (int)v; return 0;
}

/* vcf.c:831-1024; returns <0 on fatal error */
static int register_hrec(hdr_t *h, const hrec_t *r)
{
    int hl;
    if (!strcmp(r->key, "contig")) hl = HL_CTG;
    else if (!strcmp(r->key, "INFO")) hl = HL_INFO;
    else if (!strcmp(r->key, "FILTER")) hl = HL_FLT;
    else if (!strcmp(r->key, "FORMAT")) hl = HL_FMT;
    else return 0;
    if (r->value) return 0;                                                   /* generic "##INFO=foo" line: nkeys == 0 */
    if (hl == HL_CTG) {
        int i = hrec_find_ci(r, "length");
        if (i >= 0) { char *end; long long len = strtoll(r->vals[i], &end, 10); if (end == r->vals[i] || len < 0) return 0; }
        i = hrec_find_ci(r, "ID");
        if (i < 0) return 0;
        if (ctg_find(h, r->vals[i]) >= 0) return 0;
        int idx = -1, k = hrec_find_ci(r, "IDX");
        if (k != -1 && parse_idx(r->vals[k], &idx) < 0) return 0;
        char *name = strdup(r->vals[i]);
        if (set_idx_ctg(h, idx, name) < 0) { free(name); return -1; }
        return 1;
    }
    const char *id = NULL; int type = -1, var = -1, num = -1, idx = -1;
    for (int i = 0; i < r->nkeys; i++) {
        if (!strcmp(r->keys[i], "ID")) id = r->vals[i];
        else if (!strcmp(r->keys[i], "IDX")) { if (parse_idx(r->vals[i], &idx) < 0) return 0; }
        else if (!strcmp(r->keys[i], "Type")) {
            const char *v = r->vals[i];
            if (!strcmp(v, "Integer")) type = HT_INT; else if (!strcmp(v, "Float")) type = HT_REAL;
            else if (!strcmp(v, "Flag")) type = HT_FLAG; else type = HT_STR;   /* String, Character, unknown */
        } else if (!strcmp(r->keys[i], "Number")) {
            const char *v = r->vals[i]; int is_fmt = hl == HL_FMT;
            if (!strcmp(v, "A")) var = VL_A; else if (!strcmp(v, "R")) var = VL_R; else if (!strcmp(v, "G")) var = VL_G;
            else if (!strcmp(v, ".")) var = VL_VAR;
            else if (is_fmt && !strcmp(v, "P")) var = VL_P; else if (is_fmt && !strcmp(v, "LA")) var = VL_LA;
            else if (is_fmt && !strcmp(v, "LR")) var = VL_LR; else if (is_fmt && !strcmp(v, "LG")) var = VL_LG;
            else if (is_fmt && !strcmp(v, "M")) var = VL_M;
            else if (sscanf(v, "%d", &num) == 1) var = VL_FIXED;
            if (var != VL_FIXED) num = 0xfffff;
        }
    }
    if (hl == HL_INFO || hl == HL_FMT) {
        if (type == -1) type = HT_STR;
        if (var == -1) var = VL_VAR;
        if (type == HT_FLAG && (var != VL_FIXED || num != 0)) { var = VL_FIXED; num = 0; }
    }
    if (!id) return 0;
    int k = dict_find(h, id);
    if (k < 0) {
        char *name = strdup(id);
        if (set_idx_id(h, idx, name, &k) < 0) { free(name); return -1; }
    } else if (h->ids[k].has[hl]) return 0;
    h->ids[k].has[hl] = 1; h->ids[k].type[hl] = type & 0xf; h->ids[k].vl[hl] = var & 0xf;
    if (hl == HL_INFO) for (int i = 0; i < r->nkeys; i++) if (!strcmp(r->keys[i], "Description")) { h->ids[k].info_desc = strdup(r->vals[i]); break; }   /* what vep_schema_parse reads off the hrec */
    return 1;
}

static void hdr_free(hdr_t *h)
{
    for (int i = 0; i < h->n_ids; i++) { free(h->ids[i].key); free(h->ids[i].info_desc); }
    for (int i = 0; i < h->n_ctg; i++) free(h->ctg[i]);
    for (int i = 0; i < h->n_smp; i++) free(h->smp[i]);
    free(h->ids); free(h->ctg); free(h->smp);
    memset(h, 0, sizeof *h);
}

static int parse_version(const char *v)                                        /* vcf.c:145-192 */
{
    const char *major = strstr(v, "VCFv");
    if (!major) return 4002000;
    major += 4;
    const char *minor = strchr(major, '.');
    if (!minor) return 4002000;
    return (int)(strtol(major, NULL, 10) * 1000000 + strtol(minor + 1, NULL, 10) * 1000);
}

static int hdr_parse(hdr_t *h, const char *txt)
{
    memset(h, 0, sizeof *h);
    h->version = 0;
    int len;
    const char *p = txt;
    hrec_t *r = parse_line(p, &len);                                          /* first line is added twice, harmlessly */
    if (r) {
        if (r->value && !strcmp(r->key, "fileformat")) h->version = parse_version(r->value);
        if (register_hrec(h, r) < 0) { hrec_free(r); return -1; }
        hrec_free(r);
    }
    r = parse_line("##FILTER=<ID=PASS,Description=\"All filters passed\">", &len);
    register_hrec(h, r); hrec_free(r);
    int done = 0;
    do {
        while ((r = parse_line(p, &len)) != NULL) {
            if (r->value && !h->version && !strcmp(r->key, "fileformat")) h->version = parse_version(r->value);
            int rc = register_hrec(h, r);
            hrec_free(r);
            if (rc < 0) return -1;
            p += len;
        }
        if (len < 0) return -1;
        if (len > 0) { p += len; continue; }
        if (strncmp("#CHROM\t", p, 7) && strncmp("#CHROM ", p, 7)) {
            const char *eol = strchr(p, '\n');
            if (eol) p = eol + 1; else done = -1;
        } else done = 1;
    } while (!done);
    if (done < 0) return -1;
    if (!h->version) h->version = 4002000;
    const char *mand = "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO";
    if (strncmp(p, mand, strlen(mand))) return -1;
    const char *beg = p + strlen(mand);
    if (!*beg || *beg == '\n') return 0;
    if (strncmp(beg, "\tFORMAT\t", 8)) return -1;
    beg += 8;
    while (*beg) {
        const char *end = beg;
        while (*end && *end != '\t' && *end != '\n') end++;
        size_t l = (size_t)(end - beg);
        const char *ss = beg;
        while (*ss && isspace((unsigned char)*ss) && (size_t)(ss - beg) < l) ss++;
        if (!*ss || (size_t)(ss - beg) == l) return -1;                       /* empty sample name */
        char *nm = dupn(beg, l);
        for (int i = 0; i < h->n_smp; i++) if (!strcmp(h->smp[i], nm)) { free(nm); return -1; }   /* duplicate */
        h->smp = (char **)realloc(h->smp, sizeof(char *) * (size_t)(h->n_smp + 1));
        h->smp[h->n_smp++] = nm;
        if (!*end || *end == '\n') break;
        beg = end + 1;
    }
    return 0;
}

/* ------------------------------------------------------------------ schema (bcf_reader.c:540-760) --------------- */
typedef struct { const char *name; int vl; int type; } spec_t;
/* the VCF-spec reserved keys the reader corrects (vcf_types.h:46-93): name -> Number class */
static const spec_t FMT_SPEC[] = { {"AD", VL_R, HT_INT}, {"ADF", VL_R, HT_INT}, {"ADR", VL_R, HT_INT}, {"EC", VL_A, HT_INT}, {"GL", VL_G, HT_REAL},
    {"GP", VL_G, HT_REAL}, {"PL", VL_G, HT_INT}, {"PP", VL_G, HT_INT}, {"DP", VL_FIXED, HT_INT}, {"LEN", VL_FIXED, HT_INT}, {"FT", VL_FIXED, HT_STR},
    {"GQ", VL_FIXED, HT_INT}, {"GT", VL_FIXED, HT_STR}, {"HQ", VL_FIXED, HT_INT}, {"MQ", VL_FIXED, HT_INT}, {"PQ", VL_FIXED, HT_INT},
    {"PS", VL_FIXED, HT_INT}, {NULL, 0, 0} };
static const spec_t INFO_SPEC[] = { {"AD", VL_R, HT_INT}, {"ADF", VL_R, HT_INT}, {"ADR", VL_R, HT_INT}, {"AC", VL_A, HT_INT}, {"AF", VL_A, HT_REAL},
    {"CIGAR", VL_A, HT_STR}, {"AA", VL_FIXED, HT_STR}, {"AN", VL_FIXED, HT_INT}, {"BQ", VL_FIXED, HT_REAL}, {"DB", VL_FIXED, HT_FLAG},
    {"DP", VL_FIXED, HT_INT}, {"END", VL_FIXED, HT_INT}, {"H2", VL_FIXED, HT_FLAG}, {"H3", VL_FIXED, HT_FLAG}, {"MQ", VL_FIXED, HT_REAL},
    {"MQ0", VL_FIXED, HT_INT}, {"NS", VL_FIXED, HT_INT}, {"SB", VL_FIXED, HT_INT}, {"SOMATIC", VL_FIXED, HT_FLAG}, {"VALIDATED", VL_FIXED, HT_FLAG},
    {"1000G", VL_FIXED, HT_FLAG}, {NULL, 0, 0} };

static int corrected_vl(const spec_t *tab, const char *name, int vl)           /* vcf_types.h:119-205 */
{
    for (; tab->name; tab++) if (!strcmp(tab->name, name)) {
        int bad = tab->vl == VL_FIXED ? (vl != VL_FIXED) : (vl != tab->vl && vl != VL_VAR);
        return bad ? tab->vl : vl;
    }
    return vl;
}

typedef struct { char *name; int id, htype, is_list; } field_t;
static int duck_type(int ht) { return ht == HT_FLAG ? T_BOOLEAN : ht == HT_INT ? T_INTEGER : ht == HT_REAL ? T_FLOAT : T_VARCHAR; }

/* ------------------------------------------------------------------ typed values -------------------------------- */
static const int TYPE_SHIFT[16] = { 0, 0, 1, 2, 3, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
static int32_t rd_i16(const uint8_t *p) { return (int16_t)(p[0] | p[1] << 8); }
static int32_t rd_i32(const uint8_t *p) { return (int32_t)((uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24); }
static uint32_t rd_u32(const uint8_t *p) { return (uint32_t)rd_i32(p); }

static int dec_int1_safe(const uint8_t *p, const uint8_t *end, const uint8_t **q, int32_t *val)    /* vcf.c:1918-1949 */
{
    if (end - p < 2) return -1;
    int t = *p++ & 0xf;
    if (t == 1) *val = (int8_t)*p++;
    else {
        if (end - p < (1 << TYPE_SHIFT[t])) return -1;
        if (t == 2) { *val = rd_i16(p); p += 2; }
        else if (t == 3) { *val = rd_i32(p); p += 4; }
        else return -1;
    }
    *q = p; return 0;
}
static int dec_size_safe(const uint8_t *p, const uint8_t *end, const uint8_t **q, int *num, int *type)   /* vcf.c:1951-1963 */
{
    if (p >= end) return -1;
    *type = *p & 0xf;
    if (*p >> 4 != 15) { *q = p + 1; *num = *p >> 4; return 0; }
    int r = dec_int1_safe(p + 1, end, q, num);
    if (r) return r;
    return *num >= 0 ? 0 : -1;
}
static int32_t dec_int1(const uint8_t *p, int type, const uint8_t **q)         /* htslib/vcf.h bcf_dec_int1 */
{
    if (type == 1) { *q = p + 1; return (int8_t)*p; }
    if (type == 2) { *q = p + 2; return rd_i16(p); }
    *q = p + 4; return rd_i32(p);
}

typedef struct { int key, type, n, ns; const uint8_t *p; } ent_t;                   /* one INFO / FORMAT entry of a record */

/* widen one stored element to the 32-bit word the getters produce (vcf.c:6096-6131, 6221-6244) */
static int elem_word(const uint8_t *p, int type, int j, uint32_t *w, int *is_end, int *is_missing)
{
    *is_end = *is_missing = 0;
    if (type == 1) { int8_t v = (int8_t)p[j]; *is_end = v == -127; *is_missing = v == -128; *w = *is_missing ? 0x80000000u : (uint32_t)(int32_t)v; return 0; }
    if (type == 2) { int32_t v = rd_i16(p + 2 * j); *is_end = v == -32767; *is_missing = v == -32768; *w = *is_missing ? 0x80000000u : (uint32_t)v; return 0; }
    if (type == 3) { int32_t v = rd_i32(p + 4 * j); *is_end = v == -2147483647; *is_missing = v == (-2147483647 - 1); *w = (uint32_t)v; return 0; }
    if (type == 5) { uint32_t v = rd_u32(p + 4 * j); *is_end = v == 0x7F800002u; *is_missing = v == 0x7F800001u; *w = v; return 0; }
    return -1;
}

/* ------------------------------------------------------------------ the scan ------------------------------------ */
#define ORC_BCF_EOPEN (-100)
#define ORC_BCF_EHDR (-101)

/* ---- VEP / CSQ / BCSQ / ANN (src/vep_parser.c) ---- */
typedef struct { char *name; int htype; } vepf_t;
static int vep_infer(const char *n)                                            /* vep_infer_type :70-90 */
{
    if (!strcmp(n, "DISTANCE") || !strcmp(n, "STRAND") || !strcmp(n, "TSL") || !strcmp(n, "GENE_PHENO") || !strcmp(n, "HGVS_OFFSET") ||
        strstr(n, "MOTIF_POS") == n) return HT_INT;
    if (!strcmp(n, "Consequence") || !strcmp(n, "FLAGS") || !strcmp(n, "CLIN_SIG")) return HT_STR;
    if (strstr(n, "_AF") || strstr(n, "AF_") || strstr(n, "MOTIF_SCORE_CHANGE") || strstr(n, "SpliceAI_pred_DS_") == n) return HT_REAL;   /* (the *_AF names of :82-84 all contain "_AF") */
    return HT_STR;
}

typedef struct {
    hdr_t h;
    int n_info, n_fmt, tidy, ncol, fmt_default;
    field_t *info, *fmt;
    col_t *col;
    int c_info0, c_sample, c_fmt0;
    int n_vep, vep_id; vepf_t *vep;           /* VEP_* columns sit at 7 .. 7 + n_vep - 1; vep_id = dictionary id of the tag */
    int gt_id;
    int64_t n_rows, n_recs;
    buf_t rec_rid, rec_pos, rec_rlen;      /* per emitted record: contig id, 0-based pos, rlen (region-oracle inputs) */
    const int32_t *pos_hi; size_t n_pos_hi;   /* VCF text: bits 32.. of every record's 0-based position, in record order (hts_pos_t is 64 bits wide: vcf.c:4052-4063) */
} scan_t;

static void scan_free(scan_t *s)
{
    for (int i = 0; i < s->ncol; i++) col_free(&s->col[i]);
    free(s->col);
    for (int i = 0; i < s->n_info; i++) free(s->info[i].name);
    for (int i = 0; i < s->n_fmt; i++) free(s->fmt[i].name);
    for (int i = 0; i < s->n_vep; i++) free(s->vep[i].name);
    free(s->info); free(s->fmt); free(s->vep);
    free(s->rec_rid.p); free(s->rec_pos.p); free(s->rec_rlen.p);
    hdr_free(&s->h);
}

static int build_schema(scan_t *s)
{
    hdr_t *h = &s->h;
    /* vep_detect_tag :102-119 (first of CSQ, BCSQ, ANN, VEP, vep that is an INFO tag) + vep_schema_parse :125-182 */
    static const char *veptags[] = { "CSQ", "BCSQ", "ANN", "VEP", "vep", NULL };
    s->vep_id = -1;
    for (int i = 0; veptags[i]; i++) { int id = dict_find(h, veptags[i]); if (id >= 0 && h->ids[id].has[HL_INFO]) { s->vep_id = id; break; } }
    if (s->vep_id >= 0) {
        const char *desc = h->ids[s->vep_id].info_desc, *fmt = desc ? strstr(desc, "Format: ") : NULL;
        int nf = 0;
        if (fmt) {                                                             /* parse_format_string :33-45 */
            fmt += 8;
            const char *end = strchr(fmt, '"'); if (!end) end = fmt + strlen(fmt);
            nf = 1; for (const char *q = fmt; q < end; q++) if (*q == '|') nf++;
        }
        if (!fmt || nf > 256) s->vep_id = -1;
        else {                                                                 /* split_format_fields :47-68 */
            s->vep = (vepf_t *)calloc((size_t)nf, sizeof(vepf_t));
            const char *st = fmt;
            for (const char *q = fmt;; q++) if (*q == '|' || *q == 0 || *q == '"') {
                s->vep[s->n_vep].name = dupn(st, (size_t)(q - st)); s->vep[s->n_vep].htype = vep_infer(s->vep[s->n_vep].name); s->n_vep++;
                if (*q == 0 || *q == '"') break;
                st = q + 1;
            }
        }
    }
    for (int i = 0; i < h->n_ids; i++) if (h->ids[i].key && h->ids[i].has[HL_INFO]) s->n_info++;
    s->info = (field_t *)calloc((size_t)s->n_info + 1, sizeof(field_t));
    for (int i = 0, k = 0; i < h->n_ids; i++) if (h->ids[i].key && h->ids[i].has[HL_INFO]) {
        field_t *f = &s->info[k++];
        f->name = strdup(h->ids[i].key); f->id = i; f->htype = h->ids[i].type[HL_INFO];
        f->is_list = corrected_vl(INFO_SPEC, f->name, h->ids[i].vl[HL_INFO]) != VL_FIXED;
    }
    if (h->n_smp > 0) {
        for (int i = 0; i < h->n_ids; i++) if (h->ids[i].key && h->ids[i].has[HL_FMT]) s->n_fmt++;
        if (s->n_fmt == 0) {                                                   /* default GT column: bcf_reader.c:683-692 */
            s->n_fmt = 1; s->fmt_default = 1;
            s->fmt = (field_t *)calloc(1, sizeof(field_t));
            s->fmt[0].name = strdup("GT"); s->fmt[0].htype = HT_STR; s->fmt[0].id = -1;
        } else {
            s->fmt = (field_t *)calloc((size_t)s->n_fmt, sizeof(field_t));
            for (int i = 0, k = 0; i < h->n_ids; i++) if (h->ids[i].key && h->ids[i].has[HL_FMT]) {
                field_t *f = &s->fmt[k++];
                f->name = strdup(h->ids[i].key); f->id = i; f->htype = h->ids[i].type[HL_FMT];
                f->is_list = corrected_vl(FMT_SPEC, f->name, h->ids[i].vl[HL_FMT]) != VL_FIXED;
            }
        }
    }
    int nf = h->n_smp > 0 ? (s->tidy ? 1 + s->n_fmt : h->n_smp * s->n_fmt) : 0;
    s->ncol = 7 + s->n_vep + s->n_info + nf;
    s->col = (col_t *)calloc((size_t)s->ncol, sizeof(col_t));
    col_init(&s->col[0], "CHROM", T_VARCHAR, 0); col_init(&s->col[1], "POS", T_BIGINT, 0); col_init(&s->col[2], "ID", T_VARCHAR, 0);
    col_init(&s->col[3], "REF", T_VARCHAR, 0); col_init(&s->col[4], "ALT", T_VARCHAR, 1); col_init(&s->col[5], "QUAL", T_DOUBLE, 0);
    col_init(&s->col[6], "FILTER", T_VARCHAR, 1);
    int c = 7; char nm[640];
    for (int v = 0; v < s->n_vep; v++) {                                       /* bcf_reader.c:587-600: char col_name[256] */
        snprintf(nm, 256, "VEP_%s", s->vep[v].name);
        col_init(&s->col[c], nm, duck_type(s->vep[v].htype), 1); s->col[c].has_cvalid = 1; c++;
    }
    s->c_info0 = c;
    for (int i = 0; i < s->n_info; i++) { snprintf(nm, sizeof nm, "INFO_%s", s->info[i].name); col_init(&s->col[c++], nm, duck_type(s->info[i].htype), s->info[i].is_list); }
    s->c_sample = -1; s->c_fmt0 = c;
    if (h->n_smp > 0) {
        if (s->tidy) {
            s->c_sample = c; col_init(&s->col[c++], "SAMPLE_ID", T_VARCHAR, 0); s->c_fmt0 = c;
            for (int f = 0; f < s->n_fmt; f++) { snprintf(nm, sizeof nm, "FORMAT_%s", s->fmt[f].name); col_init(&s->col[c++], nm, duck_type(s->fmt[f].htype), s->fmt[f].is_list); }
        } else {
            for (int sm = 0; sm < h->n_smp; sm++) for (int f = 0; f < s->n_fmt; f++) {
                snprintf(nm, sizeof nm, "FORMAT_%s_%s", s->fmt[f].name, h->smp[sm]);
                col_init(&s->col[c++], nm, duck_type(s->fmt[f].htype), s->fmt[f].is_list);
            }
        }
    }
    s->gt_id = dict_find(h, "GT");
    return 0;
}

/* bcf_fmt_array for CHAR data (vcf.c:3036-3056): n == 0 -> ".", else bytes up to the first NUL */
static size_t char_len(const uint8_t *p, int n) { const uint8_t *z = (const uint8_t *)memchr(p, 0, (size_t)n); return z ? (size_t)(z - p) : (size_t)n; }

static void emit_numeric(col_t *c, int htype, int is_list, const uint32_t *w, int n)   /* bcf_reader.c:1576-1683 / 1769-1889 */
{
    uint32_t miss = htype == HT_INT ? 0x80000000u : 0x7F800001u, vend = htype == HT_INT ? 0x80000001u : 0x7F800002u;
    if (is_list) {
        list_begin(c);
        for (int v = 0; v < n; v++) if (w[v] != miss && w[v] != vend) list_fixed(c, w[v]);
        list_end(c);
    } else if (w[0] != miss) col_fixed(c, w[0], 1);
    else col_null(c);
}

/* one FORMAT cell (sample smp of field f); `e` is the record's entry or NULL when the tag is absent */
static void emit_format(scan_t *s, col_t *c, const field_t *f, const ent_t *e, int smp)
{
    /* Parity domain: a record whose n_sample is smaller than the header's makes the reference read past the vector
     * (the getters iterate bcf_hdr_nsamples, vcf.c:6215); here such cells are NULL. */
    if (e && smp >= e->ns) e = NULL;
    if ((f->htype == HT_INT || f->htype == HT_REAL) && !strcmp(f->name, "GT")) { col_null(c); return; }   /* tag "GT" must be declared String: vcf.c:6183-6187 returns -2 */
    if (f->htype == HT_INT || f->htype == HT_REAL) {
        /* bcf_get_format_values: >0 values only when present with n > 0; stored CHAR/NULL make htslib exit(1): out of domain */
        if (!e || e->n <= 0 || !(e->type == 1 || e->type == 2 || e->type == 3 || e->type == 5)) { col_null(c); return; }
        uint32_t *w = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)e->n);
        const uint8_t *p = e->p + (size_t)smp * ((size_t)e->n << TYPE_SHIFT[e->type]);
        uint32_t vend = e->type == 5 ? 0x7F800002u : 0x80000001u, miss = e->type == 5 ? 0x7F800001u : 0x80000000u;
        int j = 0;
        for (; j < e->n; j++) {
            uint32_t x; int ie, im; elem_word(p, e->type, j, &x, &ie, &im);
            if (im) w[j] = miss; else if (ie) break; else w[j] = x;
        }
        for (; j < e->n; j++) w[j] = vend;
        emit_numeric(c, f->htype, f->is_list, w, e->n);
        free(w);
        return;
    }
    if (f->htype == HT_FLAG) { col_null(c); return; }                          /* FORMAT Flag: no getter branch matches a BOOLEAN column sensibly; the reference takes the string path */
    if (!strcmp(f->name, "GT")) {                                             /* bcf_reader.c:1893-1957 */
        int ok = s->gt_id >= 0 && s->h.ids[s->gt_id].has[HL_FMT] && s->h.ids[s->gt_id].type[HL_FMT] == HT_STR;
        if (!ok || !e || e->n <= 0 || !(e->type == 1 || e->type == 2 || e->type == 3 || e->type == 5)) { col_null(c); return; }
        char buf[64]; buf_t out = { 0 };
        const uint8_t *p = e->p + (size_t)smp * ((size_t)e->n << TYPE_SHIFT[e->type]);
        uint32_t miss = e->type == 5 ? 0x7F800001u : 0x80000000u;
        for (int j = 0; j < e->n; j++) {
            uint32_t x; int ie, im; elem_word(p, e->type, j, &x, &ie, &im);
            if (im) x = miss; else if (ie) break;
            int32_t v = (int32_t)x;
            if (v == -2147483647) break;                                       /* a widened value equal to int32 vector_end */
            if (j > 0) buf_u8(&out, (v & 1) ? '|' : '/');
            if ((v >> 1) == 0) buf_u8(&out, '.');
            else { int l = snprintf(buf, sizeof buf, "%d", (v >> 1) - 1); buf_push(&out, buf, (size_t)l); }
        }
        if (out.n) col_str(c, out.p, out.n); else col_null(c);
        free(out.p);
        return;
    }
    /* other string FORMAT fields: bcf_get_format_string copies fmt->n BYTES per sample at stride fmt->n (vcf.c:6166-6174) */
    /* a Number=. string FORMAT field binds as LIST(VARCHAR) but the reference assigns a plain string into the list vector
     * (bcf_reader.c:1971-1972): undefined there, NULL here. */
    if (!e || f->is_list) { col_null(c); return; }
    const uint8_t *p = e->p + (size_t)smp * (size_t)e->n;
    col_str(c, p, e->n > 0 ? char_len(p, e->n) : 0);
}

/* One record's VEP_* cells.  vep_record_parse_bcf :317-326 (bcf_get_info_string: the tag must be declared String, present, len > 0),
 * vep_record_parse :286-315 (strtok_r on ','), parse_single_transcript :243-284 ('|' split, trim, "" / "." missing),
 * vep_parse_int / vep_parse_float :207-235; cells as bcf_reader.c:1463-1541 writes them; in tidy mode only a record's first sample
 * row carries the annotation (bcf_reader.c:1370-1373). */
static void emit_vep(scan_t *s, col_t *vc, const ent_t *info, int n_info, int rep)
{
    const ent_t *e = NULL;
    for (int i = 0; i < n_info; i++) if (info[i].key == s->vep_id) { e = &info[i]; break; }
    char *str = NULL;
    if (rep == 0 && e && e->n > 0 && s->h.ids[s->vep_id].type[HL_INFO] == HT_STR) {
        str = (char *)malloc((size_t)e->n + 1); memcpy(str, e->p, (size_t)e->n); str[e->n] = 0;         /* vcf.c:6071-6081 */
    }
    /* transcripts: the non-empty ','-separated pieces */
    char **tr = NULL; int ntr = 0;
    if (str && *str) {
        tr = (char **)malloc(sizeof(char *) * (strlen(str) + 1));
        char *save = NULL;
        for (char *t = strtok_r(str, ",", &save); t; t = strtok_r(NULL, ",", &save)) tr[ntr++] = t;
    }
    if (ntr == 0) { for (int v = 0; v < s->n_vep; v++) col_null(&vc[v]); free(tr); free(str); return; }
    /* tokens of every transcript: tok[t * n_vep + v] (NULL = missing) */
    char **tok = (char **)calloc((size_t)ntr * (size_t)s->n_vep, sizeof(char *));
    for (int t = 0; t < ntr; t++) {
        char *token = tr[t]; int fi = 0;
        while (fi < s->n_vep) {
            char *np = strchr(token, '|');
            if (np) *np = 0;
            while (*token && isspace((unsigned char)*token)) token++;
            char *end = token + strlen(token) - 1;
            while (end > token && isspace((unsigned char)*end)) *end-- = 0;
            if (*token && strcmp(token, ".") != 0) tok[(size_t)t * (size_t)s->n_vep + (size_t)fi] = token;
            fi++;
            if (!np) break;
            token = np + 1;
        }
    }
    for (int v = 0; v < s->n_vep; v++) {
        col_t *cc = &vc[v];
        list_begin(cc);
        for (int t = 0; t < ntr; t++) {
            const char *tk = tok[(size_t)t * (size_t)s->n_vep + (size_t)v];
            if (s->vep[v].htype == HT_INT) {
                int32_t iv = INT32_MIN;
                if (tk) { char *endp; long val = strtol(tk, &endp, 10); if (!(endp == tk || *endp)) iv = (int32_t)val; }
                list_fixed(cc, tk ? (uint64_t)(uint32_t)iv : 0);
            } else if (s->vep[v].htype == HT_REAL) {
                float fv = NAN;
                if (tk) { char *endp; double val = strtod(tk, &endp); if (!(endp == tk || *endp)) fv = (float)val; }
                uint32_t b; memcpy(&b, &fv, 4);
                list_fixed(cc, tk ? b : 0);
            } else list_str(cc, tk ? tk : "", tk ? strlen(tk) : 0);
            list_cvalid(cc, tk != NULL);
        }
        list_end(cc);
    }
    free(tok); free(tr); free(str);
}

static int scan_records(scan_t *s, const uint8_t *u, size_t ulen, size_t pos, int materialise)
{
    hdr_t *h = &s->h;
    ent_t *info = NULL, *fmt = NULL; int cap_info = 0, cap_fmt = 0;
    int status = 0;
    const uint8_t **al = NULL; int *aln = NULL; int cap_al = 0;
    int32_t *flt = NULL; int cap_flt = 0;
    for (;;) {
        /* bcf_read1_core */
        if (pos == ulen) break;
        if (ulen - pos < 32) { status = -2; break; }
        const uint8_t *x = u + pos;
        uint32_t shared_len = rd_u32(x), indiv_len = rd_u32(x + 4);
        if (shared_len < 24) { status = -2; break; }
        shared_len -= 24;
        int32_t rid = rd_i32(x + 8);
        int64_t rpos = rd_u32(x + 12); if (rpos == 0xFFFFFFFFll) rpos = -1;
        if (s->pos_hi && (size_t)s->n_recs < s->n_pos_hi) rpos = (int64_t)(((uint64_t)(uint32_t)s->pos_hi[s->n_recs] << 32) | rd_u32(x + 12));
        uint32_t qbits = rd_u32(x + 20);
        int n_info = x[24] | x[25] << 8, n_allele = x[26] | x[27] << 8;
        uint32_t n_sample = rd_u32(x + 28) & 0xffffff; int n_fmt = x[31];
        if ((!indiv_len || !n_sample) && n_fmt) n_fmt = 0;
        if ((uint64_t)ulen - pos - 32 < (uint64_t)shared_len + indiv_len) { status = -2; break; }
        const uint8_t *sh = x + 32, *she = sh + shared_len, *in = she, *ine = in + indiv_len;
        /* bcf_record_check */
        int err = 0, num, type; const uint8_t *p = sh; size_t bytes;
        if (rid < 0 || rid >= h->n_ctg || !h->ctg[rid]) err = 1;
        if (dec_size_safe(p, she, &p, &num, &type)) { status = -2; break; }
        if (type != 7) err = 1;
        const uint8_t *idp = p; int idn = num;
        bytes = (size_t)num << TYPE_SHIFT[type];
        if ((size_t)(she - p) < bytes) { status = -2; break; }
        p += bytes;
        if (n_allele < 1) err = 1;
        if (n_allele > cap_al) { cap_al = n_allele + 8; al = (const uint8_t **)realloc(al, sizeof(*al) * (size_t)cap_al); aln = (int *)realloc(aln, sizeof(int) * (size_t)cap_al); }
        int bad = 0;
        for (int i = 0; i < n_allele; i++) {
            if (dec_size_safe(p, she, &p, &num, &type)) { bad = 1; break; }
            if (type != 7) err = 1;
            al[i] = p; aln[i] = num;
            bytes = (size_t)num << TYPE_SHIFT[type];
            if ((size_t)(she - p) < bytes) { bad = 1; break; }
            p += bytes;
        }
        if (bad) { status = -2; break; }
        if (dec_size_safe(p, she, &p, &num, &type)) { status = -2; break; }
        int n_flt = 0;
        if (num > 0) {
            bytes = (size_t)num << TYPE_SHIFT[type];
            if ((size_t)(she - p) < bytes) { status = -2; break; }
            if (!(type == 1 || type == 2 || type == 3)) { err = 1; p += bytes; }
            else {
                if (num > cap_flt) { cap_flt = num + 8; flt = (int32_t *)realloc(flt, sizeof(int32_t) * (size_t)cap_flt); }
                for (int i = 0; i < num; i++) {
                    int32_t key = dec_int1(p, type, &p);
                    if (key < 0 || key >= h->n_ids || !h->ids[key].key) err = 1;
                    flt[i] = key;
                }
                n_flt = num;
            }
        }
        if (n_info > cap_info) { cap_info = n_info + 8; info = (ent_t *)realloc(info, sizeof(ent_t) * (size_t)cap_info); }
        for (int i = 0; i < n_info && !bad; i++) {
            int32_t key = -1;
            if (dec_int1_safe(p, she, &p, &key)) { bad = 1; break; }
            if (key < 0 || key >= h->n_ids || !h->ids[key].key) err = 1;
            if (dec_size_safe(p, she, &p, &num, &type)) { bad = 1; break; }
            if (!(type == 0 || type == 1 || type == 2 || type == 3 || type == 5 || type == 7) || (type == 0 && num > 0)) err = 1;
            bytes = (size_t)num << TYPE_SHIFT[type];
            if ((size_t)(she - p) < bytes) { bad = 1; break; }
            info[i].key = key; info[i].type = type; info[i].n = num; info[i].p = p; info[i].ns = 0;
            p += bytes;
        }
        if (bad) { status = -2; break; }
        p = in;
        if (n_fmt > cap_fmt) { cap_fmt = n_fmt + 8; fmt = (ent_t *)realloc(fmt, sizeof(ent_t) * (size_t)cap_fmt); }
        int gt_entry = -1;
        for (int i = 0; i < n_fmt && !bad; i++) {
            int32_t key = -1;
            if (dec_int1_safe(p, ine, &p, &key)) { bad = 1; break; }
            if (key < 0 || key >= h->n_ids || !h->ids[key].key) err = 1;
            if (dec_size_safe(p, ine, &p, &num, &type)) { bad = 1; break; }
            if (!(type == 0 || type == 1 || type == 2 || type == 3 || type == 5 || type == 7) || (type == 0 && num > 0)) err = 1;
            bytes = ((size_t)num << TYPE_SHIFT[type]) * n_sample;
            if ((size_t)(ine - p) < bytes) { bad = 1; break; }                 /* bad_indiv, or updatephasing's own bounds failure: both reject */
            fmt[i].key = key; fmt[i].type = type; fmt[i].n = num; fmt[i].p = p; fmt[i].ns = (int)n_sample;
            if (h->version < 4004000 && s->gt_id >= 0 && key == s->gt_id && gt_entry < 0 && !err) gt_entry = i;
            p += bytes;
        }
        if (bad || err) { status = -2; break; }
        s->n_recs++;
        { int64_t rl = rd_i32(x + 16); if (rl < 0) rl = aln[0] > 0 ? (int64_t)char_len(al[0], aln[0]) : 0;
          uint64_t a = (uint64_t)(int64_t)rid, b = (uint64_t)rpos, c2 = (uint64_t)rl; buf_u64(&s->rec_rid, a); buf_u64(&s->rec_pos, b); buf_u64(&s->rec_rlen, c2); }
        pos += 32 + (size_t)shared_len + indiv_len;
        if (!materialise) { s->n_rows += s->tidy && h->n_smp > 0 ? h->n_smp : 1; continue; }

        /* updatephasing (vcf.c:1985-2029) on a private copy of the GT vector of pre-4.4 files */
        uint8_t *gtcopy = NULL;
        if (gt_entry >= 0) {
            ent_t *e = &fmt[gt_entry];
            size_t inc = (size_t)1 << TYPE_SHIFT[e->type], tot = (size_t)e->n * inc * n_sample;
            gtcopy = (uint8_t *)malloc(tot + 1); memcpy(gtcopy, e->p, tot);
            uint8_t *g = gtcopy;
            for (uint32_t j = 0; j < n_sample && e->n > 0; j++, g += (size_t)e->n * inc) {
                if (e->n == 1) { if (*g) *g |= 1; }
                else if (e->n == 2) *g |= (g[inc] & 1);
                else { uint8_t all = 1; for (int k = 1; k < e->n; k++) all &= g[inc * (size_t)k]; *g |= all; }
            }
            e->p = gtcopy;
        }

        int reps = s->tidy && h->n_smp > 0 ? h->n_smp : 1;
        for (int rep = 0; rep < reps; rep++) {
            col_t *c = s->col;
            col_cstr(&c[0], h->ctg[rid]);
            col_fixed(&c[1], (uint64_t)(rpos + 1), 1);
            { const char *ids = idn ? (const char *)idp : "."; size_t l = idn ? char_len(idp, idn) : 1;
              if (l == 1 && ids[0] == '.') col_null(&c[2]); else col_str(&c[2], ids, l); }
            if (aln[0]) col_str(&c[3], al[0], char_len(al[0], aln[0])); else col_cstr(&c[3], ".");
            list_begin(&c[4]);
            for (int a = 1; a < n_allele; a++) { if (aln[a]) list_str(&c[4], al[a], char_len(al[a], aln[a])); else list_str(&c[4], ".", 1); }
            list_end(&c[4]);
            if (qbits == 0x7F800001u) col_fixed(&c[5], 0, 0);                  /* NULL with payload 0.0: bcf_reader.c:1427-1434 */
            else { float f; memcpy(&f, &qbits, 4); double d = f; uint64_t b; memcpy(&b, &d, 8); col_fixed(&c[5], b, 1); }
            list_begin(&c[6]);
            if (n_flt == 0) list_str(&c[6], "PASS", 4);
            else for (int f = 0; f < n_flt; f++) list_str(&c[6], h->ids[flt[f]].key, strlen(h->ids[flt[f]].key));
            list_end(&c[6]);
            if (s->n_vep) emit_vep(s, c + 7, info, n_info, rep);
            for (int k = 0; k < s->n_info; k++) {
                const field_t *f = &s->info[k]; col_t *cc = &c[s->c_info0 + k];
                const ent_t *e = NULL;
                for (int i = 0; i < n_info; i++) if (info[i].key == f->id) { e = &info[i]; break; }
                if (f->htype == HT_FLAG) { col_fixed(cc, e ? 1 : 0, 1); continue; }
                if (f->htype == HT_STR) {                                      /* bcf_reader.c:1686-1731 */
                    if (!e || e->n <= 0) { col_null(cc); continue; }
                    size_t l = char_len(e->p, e->n);
                    if (l == 1 && e->p[0] == '.') { col_null(cc); continue; }
                    if (!f->is_list) { col_str(cc, e->p, l); continue; }
                    /* process_comma_separated_list bcf_reader.c:996-1061: split on ',', empty tokens kept */
                    list_begin(cc);
                    size_t st = 0;
                    for (size_t i = 0; i < l; i++) if (e->p[i] == ',') { list_str(cc, e->p + st, i - st); st = i + 1; }
                    if (l > st) list_str(cc, e->p + st, l - st);                 /* the last token only when non-empty */
                    list_end(cc);
                    continue;
                }
                if (!e || !(e->type == 1 || e->type == 2 || e->type == 3 || e->type == 5)) { col_null(cc); continue; }
                uint32_t *w = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(e->n + 1));
                int j = 0;
                for (; j < e->n; j++) { uint32_t xw; int ie, im; elem_word(e->p, e->type, j, &xw, &ie, &im); if (ie) break; w[j] = im ? (e->type == 5 ? 0x7F800001u : 0x80000000u) : xw; }
                if (j > 0) emit_numeric(cc, f->htype, f->is_list, w, j); else col_null(cc);
                free(w);
            }
            if (h->n_smp > 0) {
                if (s->tidy) {
                    col_cstr(&c[s->c_sample], h->smp[rep]);
                    for (int k = 0; k < s->n_fmt; k++) {
                        const ent_t *e = NULL;
                        for (int i = 0; i < n_fmt; i++) if (fmt[i].key == s->fmt[k].id && s->fmt[k].id >= 0) { e = &fmt[i]; break; }
                        emit_format(s, &c[s->c_fmt0 + k], &s->fmt[k], e, rep);
                    }
                } else {
                    for (int sm = 0; sm < h->n_smp; sm++) for (int k = 0; k < s->n_fmt; k++) {
                        const ent_t *e = NULL;
                        for (int i = 0; i < n_fmt; i++) if (fmt[i].key == s->fmt[k].id && s->fmt[k].id >= 0) { e = &fmt[i]; break; }
                        emit_format(s, &c[s->c_fmt0 + sm * s->n_fmt + k], &s->fmt[k], e, sm);
                    }
                }
            }
            s->n_rows++;
        }
        free(gtcopy);
    }
    free(info); free(fmt); free(al); free(aln); free(flt);
    return status;
}

/* Canonical blob: u32 ncol, u64 nrows, i32 status, u64 first_rec_uoff, u32 n_samples; per column: u16 name_len, name, u8 type, u8 is_list,
 * valid[nrows]; scalar fixed: nrows x u64 raw bits | scalar varchar: (nrows+1) x u64 offsets + bytes | list: nrows x (u64 off, u64 len),
 * u64 child_n, child payload in the scalar encoding.  Trailer: u64 n_rec, then i64 rid[n_rec], i64 pos0[n_rec], i64 rlen[n_rec]. */
/* ------------------------------------------------------------------ VCF text input -------------------------------- */
/* read_bcf on a text VCF (plain or BGZF-compressed): the reference reads it through the same htsFile -- vcf_hdr_read vcf.c:2594-2680 for the
 * header, vcf_read -> vcf_parse vcf.c:3987-4165 (vcf_parse_filter 3763-3816, vcf_parse_info 3818-3985) for every line -- into the same
 * bcf1_t the binary reader fills, and the column writers do not know the difference.  Restated here as "text line -> BCF2 record bytes",
 * after which scan_records() above runs unchanged.  Number parsers: hts_str2uint / hts_str2int / hts_str2dbl textutils_internal.h:218-428.
 * Names a record uses without a header definition are added on the fly with dummy definitions, as htslib does (fix_chromosome 3744-3761 and
 * the "Dummy" lines of vcf_parse_filter / vcf_parse_info); the columns were bound before, so only the dictionaries grow.
 * Records whose POS does not fit the BCF2 core (>= 2^31 - 1) end the scan (the reference keeps 64-bit positions in memory). */
#define ORC_BCF_ESAMPLES (-103)

static void enc_typed_int(buf_t *o, int64_t x)                                 /* bcf_enc_int1 (htslib/vcf.h): smallest width; missing as int8 missing */
{
    if (x == (int32_t)0x80000000) { buf_u8(o, 0x11); buf_u8(o, 0x80); return; }
    if (x <= 127 && x > -121) { buf_u8(o, 0x11); buf_u8(o, (uint8_t)(int8_t)x); }
    else if (x <= 32767 && x > -32761) { int16_t v = (int16_t)x; buf_u8(o, 0x12); buf_push(o, &v, 2); }
    else { int32_t v = (int32_t)x; buf_u8(o, 0x13); buf_push(o, &v, 4); }
}
static void enc_size(buf_t *o, int64_t n, int type)                            /* bcf_enc_size */
{
    if (n < 15) buf_u8(o, (uint8_t)(n << 4 | type));
    else {
        buf_u8(o, (uint8_t)(15 << 4 | type));
        if (n < 128) { buf_u8(o, 0x11); buf_u8(o, (uint8_t)n); }
        else if (n < 32768) { int16_t v = (int16_t)n; buf_u8(o, 0x12); buf_push(o, &v, 2); }
        else { int32_t v = (int32_t)n; buf_u8(o, 0x13); buf_push(o, &v, 4); }
    }
}
static void enc_vchar(buf_t *o, size_t l, const char *a) { enc_size(o, (int64_t)l, 7); buf_push(o, a, l); }
static void enc_vint32(buf_t *o, int n, const int32_t *a)                      /* (any width decodes to the same values: int32 throughout) */
{
    if (n <= 0) { buf_u8(o, 0x00); return; }                                   /* bcf_enc_vint n == 0: typed NULL */
    enc_size(o, n, 3); buf_push(o, a, (size_t)n * 4);
}

/* hts_str2uint(in, &end, bits, &failed): optional '+', digits; saturates at 2^bits - 1 with *failed = 1 */
static uint64_t str2uint(const char *in, const char **end, int bits, int *failed)
{
    const unsigned char *v = (const unsigned char *)in; uint64_t n = 0, limit = (bits < 64 ? (1ULL << bits) : 0) - 1; int over = 0;
    if (*v == '+') v++;
    for (; *v >= '0' && *v <= '9'; v++) { unsigned d = *v - '0'; if (over) continue; if (n < limit / 10 || (n == limit / 10 && d <= limit % 10)) n = n * 10 + d; else { n = limit; over = 1; } }
    if (over) *failed = 1;
    *end = (const char *)v; return n;
}
/* hts_str2int(in, &end, 64, &failed): sign, digits; the end pointer moves past a lone sign */
static int64_t str2int64(const char *in, const char **end, int *failed)
{
    const unsigned char *v = (const unsigned char *)in; uint64_t n = 0, limit = (1ULL << 63) - 1; int neg = 0, over = 0;
    if (*v == '-') { limit++; neg = 1; v++; } else if (*v == '+') v++;
    for (; *v >= '0' && *v <= '9'; v++) { unsigned d = *v - '0'; if (over) continue; if (n < limit / 10 || (n == limit / 10 && d <= limit % 10)) n = n * 10 + d; else { n = limit; over = 1; } }
    if (over) *failed = 1;
    *end = (const char *)v; return neg ? (int64_t)(0 - n) : (int64_t)n;
}
/* hts_str2dbl: [+-]?digits[.digits] with at most 14 digits after the leading zeros -> n / 10^k in double arithmetic; everything else strtod */
static double str2dbl(const char *in, const char **end, int *failed)
{
    static const double D[] = { 1, 1, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20 };
    const unsigned char *v = (const unsigned char *)in; uint64_t n = 0; int max_len = 15, neg = 0, point = -1; char *e;
    while (isspace(*v)) v++;
    if (*v == '-') { neg = 1; v++; } else if (*v == '+') v++;
    if (!((*v >= '1' && *v <= '9') || (*v == '0' && v[1] != 'x' && v[1] != 'X'))) { double d = strtod(in, &e); *end = e; if (e == in) *failed = 1; return d; }
    while (*v == '0') ++v;
    const unsigned char *start = v;
    while (--max_len && *v >= '0' && *v <= '9') n = n * 10 + *v++ - '0';
    if (max_len && *v == '.') { point = (int)(v - start); v++; while (--max_len && *v >= '0' && *v <= '9') n = n * 10 + *v++ - '0'; }
    if (point < 0) point = (int)(v - start);
    if (!max_len || *v == 'e' || *v == 'E') { double d = strtod(in, &e); *end = e; if (e == in) *failed = 1; return d; }
    *end = (const char *)v;
    double d = (double)n / D[v - start - point];
    return neg ? -d : d;
}

static int hdr_add_dummy(hdr_t *h, const char *fmt, const char *name)          /* bcf_hdr_parse_line + bcf_hdr_add_hrec of a generated line */
{
    size_t l = strlen(fmt) + strlen(name) + 8; char *line = (char *)malloc(l);
    snprintf(line, l, fmt, name);
    int len; hrec_t *r = parse_line(line, &len);
    free(line);
    if (!r) return -1;
    int rc = register_hrec(h, r);
    hrec_free(r);
    return rc < 0 ? -1 : 0;
}

/* The END of a VCF line as the tabix iterator sees it (tbx_parse1, htslib tbx.c:96-312, VCF preset) -- the reference serves
 * read_bcf(region := ...) on text through tbx_itr_next, whose overlap test uses this interval, not get_rlen's: beg = POS - 1 (>= 0),
 * end = beg + max(len(REF), SVLEN of <DEL>/<DUP>/<CNV>/<INV> alleles, FORMAT/LEN of gVCF blocks), or INFO/END when that is larger. */
static int svlen_alt(const char *alt, size_t sz)                               /* svlen_on_ref_for_vcf_alt, hts_internal.h:181-197 */
{
    if (sz < 5 || alt[0] != '<') return 0;
    if (alt[4] != '>' && alt[4] != ':') return 0;
    if (memcmp(alt, "<CNV", 4) && memcmp(alt, "<DEL", 4) && memcmp(alt, "<DUP", 4) && memcmp(alt, "<INV", 4)) return 0;
    return alt[sz - 1] == '>';
}
static int64_t tbx_vcf_end(int64_t beg, const char *ref, const char *alt, const char *info, const char *rest)
{
    int64_t end = 1, reflen = (int64_t)strlen(ref), svlen = 0, fmtlen = 0;
    if (beg < 0) beg = 0;
    if (reflen > 0) end = beg + reflen;
    uint8_t svl[8192]; memset(svl, 0, sizeof svl);
    int alcnt = 1, use_svlen = 0, getlen = 0;
    for (const char *s = alt;;) {                                              /* every ALT allele (also of "."), numbered from 1 */
        const char *t = strchr(s, ',');
        size_t l = t ? (size_t)(t - s) : strlen(s);
        ++alcnt;
        if (svlen_alt(s, l)) { svl[(alcnt - 1) >> 3] |= (uint8_t)(1 << ((alcnt - 1) & 7)); use_svlen = 1; }
        else if ((l == 3 && !memcmp(s, "<*>", 3)) || (l == 9 && !memcmp(s, "<NON_REF>", 9))) getlen = 1;
        if (!t || alcnt >= 65536) break;
        s = t + 1;
    }
    const char *s = strstr(info, "END=");
    if (s == info) s += 4; else if (s) { s = strstr(info, ";END="); if (s) s += 5; }
    if (s && *s != '.') { char *e; long long v = strtoll(s, &e, 0); if (v > beg) end = v; }
    s = strstr(info, "SVLEN=");
    if (s == info) s += 6; else if (s) { s = strstr(info, ";SVLEN="); if (s) s += 7; }
    for (int d = 1; s && d < alcnt; ++d) {
        const char *t = strchr(s, ','); long long tmp = 1;
        if (use_svlen && (svl[d >> 3] & (1 << (d & 7)))) { tmp = atoll(s); if (tmp < 0) tmp = -tmp; }
        if (svlen < tmp) svlen = tmp;
        s = t ? t + 1 : NULL;
    }
    if (getlen && rest) {                                                      /* FORMAT/LEN of the samples */
        const char *fq = strchr(rest, '\t'); size_t fl = fq ? (size_t)(fq - rest) : strlen(rest);
        int lenpos = -1, pos = 0;
        for (size_t a = 0;; pos++) {
            size_t b2 = a; while (b2 < fl && rest[b2] != ':') b2++;
            if (b2 - a == 3 && !memcmp(rest + a, "LEN", 3)) { lenpos = pos; break; }
            if (b2 >= fl) break;
            a = b2 + 1;
        }
        for (const char *sm = fq ? fq + 1 : NULL; sm && lenpos >= 0;) {
            const char *se = strchr(sm, '\t'); const char *f = sm; int d = 0; long long tmp = 0;
            for (; d <= lenpos; ++d) {
                if (d == lenpos) { tmp = atoll(f); break; }
                const char *c = (const char *)memchr(f, ':', se ? (size_t)(se - f) : strlen(f));
                if (!c) break;
                f = c + 1;
            }
            if (fmtlen < tmp) fmtlen = tmp;
            sm = se ? se + 1 : NULL;
        }
    }
    int64_t m = reflen; if (svlen > m) m = svlen; if (fmtlen > m) m = fmtlen;
    if (end < beg + m) end = beg + m;
    return end;
}

/* FORMAT column + sample columns -> the indiv block (vcf_parse_format vcf.c:3686-3742 and its seven steps 3137-3684).
 * rest = "FORMAT\tsample1\tsample2..." (NUL-terminated, modified in place).  Integers and genotypes are written as int32 vectors. */
typedef struct { int key, ht, is_gt, max_l, max_m, max_g, size, skip; uint8_t *buf; } fmtaux_t;
static int vcf_format_to_bcf(hdr_t *h, char *rest, buf_t *in, int *n_fmt_out, int *n_sample_out)
{
    const int nsamples = h->n_smp;
    char *q = strchr(rest, '\t');                                             /* end of the FORMAT column */
    if (!q) return -1;                                                        /* "FORMAT column with no sample columns" */
    *q = 0;
    *n_fmt_out = 0; *n_sample_out = 0;
    if (rest[0] == '.' && rest[1] == 0) { *n_sample_out = nsamples; return 0; }   /* FORMAT ".": no fields, the sample columns are not looked at */
    fmtaux_t fmt[255]; int n_fmt = 0;
    for (char *t = rest;;) {                                                  /* vcf_parse_format_dict2 */
        char *c = strchr(t, ':'); if (c) *c = 0;
        if (n_fmt >= 255) return -1;
        int k = dict_find(h, t);
        if (k < 0 || !h->ids[k].has[HL_FMT]) {
            if (t[0] == '.' && t[1] == 0) return -1;
            if (hdr_add_dummy(h, "##FORMAT=<ID=%s,Number=1,Type=String,Description=\"Dummy\">", t) < 0 || (k = dict_find(h, t)) < 0 || !h->ids[k].has[HL_FMT]) return -1;
        }
        fmtaux_t *f = &fmt[n_fmt++]; memset(f, 0, sizeof *f);
        f->key = k; f->ht = h->ids[k].type[HL_FMT]; f->is_gt = !strcmp(t, "GT");
        if (!c) break;
        t = c + 1;
    }
    char *body = q + 1; const char *end = body + strlen(body);
    /* vcf_parse_format_max3: widths of every field over the samples; tabs become NULs */
    int n_sample = 0;
    {
        char *r = body; int l = 0, m = 1, g = 1, j;
        while (r < end) {
            j = 0; fmtaux_t *f = fmt; char *r_start = r;
            for (;;) {
                while (*r != 0 && *r != '\t' && *r != ',' && *r != '/' && *r != ':' && *r != '|') r++;
                if (*r == ',') m++;
                else if (*r == '|' || *r == '/') { if (f->is_gt) g++; }
                else {
                    if (*r == '\t') *r = 0;
                    l = (int)(r - r_start); r_start = r;
                    if (f->max_m < m) f->max_m = m;
                    if (f->max_l < l) f->max_l = l;
                    if (f->is_gt && f->max_g < g) f->max_g = g;
                    l = 0; m = g = 1;
                    if (*r == ':') { j++; f++; if (j >= n_fmt) return -1; }     /* "Incorrect number of FORMAT fields" */
                    else break;
                }
                if (r >= end) break;
                r++;
            }
            n_sample++;
            if (n_sample == nsamples) break;
            r++;
        }
    }
    /* vcf_parse_format_alloc4 */
    for (int j = 0; j < n_fmt; j++) {
        fmtaux_t *f = &fmt[j];
        if (!f->max_m) f->max_m = 1;
        if (f->ht == HT_STR) f->size = f->is_gt ? f->max_g << 2 : f->max_l;
        else if (f->ht == HT_REAL || f->ht == HT_INT) f->size = f->max_m << 2;
        else { for (int k = 0; k < j; k++) free(fmt[k].buf); return -1; }      /* "The format type ... is currently not supported" (Flag) */
        f->buf = (uint8_t *)calloc((size_t)n_sample * (size_t)f->size + 8, 1);
    }
    for (int i = 1; i < n_fmt; i++) for (int j = 0; j < i; j++) if (!fmt[j].skip && fmt[i].key == fmt[j].key) { fmt[i].skip = 1; break; }   /* duplicate tags: the later one is dropped */
    /* vcf_parse_format_fill5 */
    int rc = 0;
    {
        const char *t = body; int m = 0;
        const int v44 = h->version >= 4004000;
        while (t < end && rc == 0) {
            if (m == nsamples) break;
            int j = 0;
            while (t < end) {
                fmtaux_t *z = &fmt[j++];
                if (z->skip) { while (*t != ':' && *t) t++; }
                else if (z->ht == HT_STR && z->is_gt) {
                    uint32_t *x = (uint32_t *)(z->buf + (size_t)z->size * (size_t)m);
                    uint32_t is_phased = 0, unreadable = 0, max = 0; int l, ploidy = 0, anyunphased = 0, phasingprfx = 0, unknown1 = 0;
                    if (v44 && (*t == '|' || *t == '/')) { is_phased = *t++ == '|'; phasingprfx = 1; }
                    for (l = 0;; ++t) {
                        ploidy++;
                        if (*t == '.') { ++t; x[l++] = is_phased; if (l == 1) unknown1 = 1; }
                        else {
                            const char *tt = t; uint64_t n = 0;
                            if (*t == '+') t++;
                            while (*t >= '0' && *t <= '9') n = n * 10 + (uint64_t)(*t++ - '0');          /* (hts_str2uint with a 506-bit limit: wraps, never saturates) */
                            uint32_t val = (uint32_t)n;
                            unreadable |= tt == t;
                            if (max < val) max = val;
                            x[l++] = (val + 1) << 1 | is_phased;
                        }
                        anyunphased |= (ploidy != 1) && !is_phased;
                        is_phased = (*t == '|');
                        if (*t != '|' && *t != '/') break;
                    }
                    if (!phasingprfx) { if (ploidy == 1) { if (!unknown1) x[0] |= 1; } else x[0] |= anyunphased ? 0 : 1; }
                    if (max > (0x7fffffffu >> 1) - 1 || unreadable) { rc = -1; break; }
                    if (!l) x[l++] = 0;
                    for (; l < z->size >> 2; ++l) x[l] = 0x80000001u;
                } else if (z->ht == HT_STR) {
                    char *x = (char *)z->buf + (size_t)z->size * (size_t)m; int l;
                    for (l = 0; *t != ':' && *t; ++t) x[l++] = *t;
                } else if (z->ht == HT_INT) {
                    int32_t *x = (int32_t *)(z->buf + (size_t)z->size * (size_t)m); int l;
                    for (l = 0;; ++t) {
                        if (*t == '.') { x[l++] = (int32_t)0x80000000; ++t; }
                        else {
                            int over = 0; const char *te; int64_t v = str2int64(t, &te, &over);
                            if (te == t || over || v < -2147483640LL || v > 2147483647LL) v = (int32_t)0x80000000;
                            x[l++] = (int32_t)v; t = te;
                        }
                        if (*t != ',') break;
                    }
                    if (!l) x[l++] = (int32_t)0x80000000;
                    for (; l < z->size >> 2; ++l) x[l] = (int32_t)0x80000001;
                } else {
                    uint32_t *x = (uint32_t *)(z->buf + (size_t)z->size * (size_t)m); int l;
                    for (l = 0;; ++t) {
                        if (*t == '.' && !(t[1] >= '0' && t[1] <= '9')) { x[l++] = 0x7F800001u; ++t; }
                        else { int over = 0; const char *te; float fv = (float)str2dbl(t, &te, &over); memcpy(&x[l++], &fv, 4); t = te; }   /* (a failed conversion stores 0.0) */
                        if (*t != ',') break;
                    }
                    if (!l) x[l++] = 0x7F800001u;
                    for (; l < z->size >> 2; ++l) x[l] = 0x7F800002u;
                }
                if (*t == 0) break;
                else if (*t == ':') t++;
                else { rc = -1; break; }                                      /* "Invalid character" */
            }
            if (rc) break;
            for (; j < n_fmt; ++j) {                                          /* trailing fields the sample leaves out */
                fmtaux_t *z = &fmt[j];
                if (z->skip) continue;
                if (z->ht == HT_STR && !z->is_gt) { char *x = (char *)z->buf + (size_t)z->size * (size_t)m; if (z->size) x[0] = '.'; }
                else {
                    uint32_t *x = (uint32_t *)(z->buf + (size_t)z->size * (size_t)m);
                    if (z->size) x[0] = z->ht == HT_REAL ? 0x7F800001u : 0x80000000u;
                    for (int l = 1; l < z->size >> 2; ++l) x[l] = z->ht == HT_REAL ? 0x7F800002u : 0x80000001u;
                }
            }
            m++; t++;
        }
    }
    /* vcf_parse_format_gt6 + check7 */
    int kept = 0;
    if (rc == 0 && n_sample != nsamples) rc = -1;                              /* "Number of columns ... does not match the number of samples" */
    if (rc == 0 && n_sample > 0) for (int i = 0; i < n_fmt; i++) {
        fmtaux_t *z = &fmt[i];
        if (z->skip) continue;
        kept++;
        enc_typed_int(in, z->key);
        if (z->ht == HT_STR && !z->is_gt) { enc_size(in, z->size, 7); buf_push(in, z->buf, (size_t)z->size * (size_t)n_sample); }
        else { enc_size(in, z->size >> 2, z->ht == HT_REAL ? 5 : 3); buf_push(in, z->buf, (size_t)z->size * (size_t)n_sample); }
    }
    for (int j = 0; j < n_fmt; j++) free(fmt[j].buf);
    *n_fmt_out = kept; *n_sample_out = n_sample;
    return rc;
}

/* one line -> one BCF2 record appended to `out`; < 0: the line is an error (the scan ends before it) */
static int vcf_line_to_bcf(hdr_t *h, char *line, buf_t *out, buf_t *pos_hi)
{
    char *f[8]; int nf = 0; char *p = line; char *rest = NULL;                /* rest: FORMAT column and what follows it (NULL when there are only eight columns) */
    for (;;) {                                                                /* kstrtok on '\t': empty tokens count, all eight are required */
        f[nf++] = p;
        char *t = strchr(p, '\t');
        if (!t) break;
        *t = 0; p = t + 1;
        if (nf == 8) { rest = p; break; }
    }
    if (nf < 8) return -1;
    buf_t sh = { 0 };
    /* CHROM */
    int rid = ctg_find(h, f[0]);
    if (rid < 0) { if (hdr_add_dummy(h, "##contig=<ID=%s>", f[0]) < 0 || (rid = ctg_find(h, f[0])) < 0) { free(sh.p); return -1; } }
    /* POS */
    int failed = 0; const char *e;
    uint64_t pos1 = str2uint(f[1], &e, 62, &failed);
    if (failed || *e) { free(sh.p); return -1; }
    int64_t pos = (int64_t)pos1 - 1;
    /* ID, REF, ALT */
    if (strcmp(f[2], ".")) enc_vchar(&sh, strlen(f[2]), f[2]); else enc_size(&sh, 0, 7);
    enc_vchar(&sh, strlen(f[3]), f[3]);
    int n_allele = 1; int32_t rlen = (int32_t)strlen(f[3]);
    if (strcmp(f[4], ".")) {
        char *t = f[4];
        for (char *r = f[4];; ++r) if (*r == ',' || *r == 0) {
            if (n_allele == 65535) { free(sh.p); return -1; }
            enc_vchar(&sh, (size_t)(r - t), t); t = r + 1; ++n_allele;
            if (*r == 0) break;
        }
    }
    /* rlen: what makes pos + rlen the END the tabix iterator tests regions with (text has no stored rlen; get_rlen's value is not observable) */
    rlen = (int32_t)(tbx_vcf_end(pos, f[3], f[4], f[7], rest) - pos);
    /* QUAL: atof */
    uint32_t qbits = 0x7F800001u;
    if (strcmp(f[5], ".")) { float q = (float)atof(f[5]); memcpy(&qbits, &q, 4); }
    /* FILTER */
    if (strcmp(f[6], ".")) {
        size_t l = strlen(f[6]);
        if (l && f[6][l - 1] == ';') f[6][l - 1] = 0;
        int n_flt = 1; for (char *r = f[6]; *r; ++r) if (*r == ';') ++n_flt;
        int32_t *a = (int32_t *)malloc(sizeof(int32_t) * (size_t)n_flt); int i = 0;
        for (char *t = f[6];;) {
            char *sc = strchr(t, ';'); if (sc) *sc = 0;
            int k = dict_find(h, t);
            if (k < 0) { if (hdr_add_dummy(h, "##FILTER=<ID=%s,Description=\"Dummy\">", t) < 0 || (k = dict_find(h, t)) < 0) { free(a); free(sh.p); return -1; } }
            a[i++] = k;
            if (!sc) break;
            t = sc + 1;
        }
        enc_vint32(&sh, n_flt, a);
        free(a);
    } else buf_u8(&sh, 0x00);
    /* INFO */
    int n_info = 0;
    if (strcmp(f[7], ".")) {
        size_t l = strlen(f[7]);
        if (l && f[7][l - 1] == ';') f[7][l - 1] = 0;
        char *r, *key;
        for (r = key = f[7];; ++r) {
            while (*r != ';' && *r != '=' && *r != 0) r++;
            if (n_info == 65535) { free(sh.p); return -1; }
            char *val = NULL, *end; int c = *r; *r = 0;
            if (c == '=') { val = r + 1; for (end = val; *end != ';' && *end != 0; ++end); c = *end; *end = 0; } else end = r;
            if (!*key) { if (c == 0) break; r = end; key = r + 1; continue; }
            int k = dict_find(h, key);
            if (k < 0 || !h->ids[k].has[HL_INFO]) {
                if (hdr_add_dummy(h, "##INFO=<ID=%s,Number=1,Type=String,Description=\"Dummy\">", key) < 0 || (k = dict_find(h, key)) < 0 || !h->ids[k].has[HL_INFO]) { free(sh.p); return -1; }
            }
            int ht = h->ids[k].type[HL_INFO];
            ++n_info;
            enc_typed_int(&sh, k);
            if (!val) buf_u8(&sh, 0x00);
            else if (ht == HT_FLAG || ht == HT_STR) enc_vchar(&sh, (size_t)(end - val), val);
            else {
                int n_val = 1; for (char *t = val; *t; ++t) if (*t == ',') ++n_val;
                int32_t *a = (int32_t *)malloc(sizeof(int32_t) * (size_t)n_val);
                const char *t = val, *te;
                for (int i = 0; i < n_val; ++i, ++t) {
                    int over = 0;
                    if (ht == HT_INT) {
                        int64_t v = str2int64(t, &te, &over);
                        if (te == t) v = (int32_t)0x80000000;
                        else if (over || v < -2147483640LL || v > 2147483647LL) v = (int32_t)0x80000000;
                        a[i] = (int32_t)v;
                    } else {
                        float fv = (float)str2dbl(t, &te, &over);
                        uint32_t b; memcpy(&b, &fv, 4);
                        if (te == t || over) b = 0x7F800001u;
                        a[i] = (int32_t)b;
                    }
                    for (t = te; *t && *t != ','; t++);
                }
                if (ht == HT_INT) { if (n_val == 1) enc_typed_int(&sh, a[0]); else enc_vint32(&sh, n_val, a); }
                else { enc_size(&sh, n_val, 5); buf_push(&sh, a, (size_t)n_val * 4); }
                free(a);
            }
            if (c == 0) break;
            r = end; key = r + 1;
        }
    }
    buf_t in = { 0 }; int n_fmt = 0, n_sample = 0;
    if (rest && h->n_smp > 0 && vcf_format_to_bcf(h, rest, &in, &n_fmt, &n_sample) < 0) { free(sh.p); free(in.p); return -1; }
    uint32_t l_shared = 24 + (uint32_t)sh.n, l_indiv = (uint32_t)in.n, w;
    buf_push(out, &l_shared, 4); buf_push(out, &l_indiv, 4);
    int32_t i32 = rid; buf_push(out, &i32, 4); i32 = (int32_t)(uint32_t)(uint64_t)pos; buf_push(out, &i32, 4); buf_push(out, &rlen, 4); buf_push(out, &qbits, 4);      /* the core takes the low word */
    { int32_t hi = (int32_t)(pos >> 32); buf_push(pos_hi, &hi, 4); }
    w = (uint32_t)n_info | ((uint32_t)n_allele << 16); buf_push(out, &w, 4);
    w = ((uint32_t)n_sample & 0xffffff) | ((uint32_t)n_fmt << 24); buf_push(out, &w, 4);
    buf_push(out, sh.p, sh.n); buf_push(out, in.p, in.n);
    free(sh.p); free(in.p);
    return 0;
}

/* text stream -> header (parsed into s->h, schema built) + BCF2 record bytes; returns ORC_BCF_* or 0; *bad = 1 when a line failed */
static int vcf_text_load(scan_t *s, const uint8_t *u, size_t ulen, buf_t *recs, buf_t *pos_hi, int *bad)
{
    buf_t txt = { 0 }; size_t pos = 0; int have_sample_line = 0;
    char *line = NULL;
    #define NEXT_LINE(ok) do { ok = pos < ulen; if (ok) { const uint8_t *nl = (const uint8_t *)memchr(u + pos, '\n', ulen - pos); size_t e = nl ? (size_t)(nl - u) : ulen, l = e - pos; \
        free(line); line = dupn((const char *)u + pos, l); if (l && line[l - 1] == '\r') line[l - 1] = 0; pos = nl ? e + 1 : ulen; } } while (0)
    int ok;
    for (;;) {                                                                /* vcf_hdr_read */
        NEXT_LINE(ok);
        if (!ok) break;
        if (!line[0]) continue;
        if (line[0] != '#') { free(line); free(txt.p); return ORC_BCF_EHDR; }  /* "No sample line" */
        buf_push(&txt, line, strlen(line)); buf_u8(&txt, '\n');
        if (line[1] != '#') { have_sample_line = 1; break; }
    }
    if (!txt.n) { free(line); free(txt.p); return ORC_BCF_EHDR; }
    (void)have_sample_line;
    buf_u8(&txt, 0);
    if (hdr_parse(&s->h, (const char *)txt.p) < 0) { free(line); free(txt.p); return ORC_BCF_EHDR; }
    free(txt.p);
    int rc = build_schema(s);
    if (rc < 0) { free(line); return rc; }
    for (;;) {
        NEXT_LINE(ok);
        if (!ok) break;
        if (vcf_line_to_bcf(&s->h, line, recs, pos_hi) < 0) { *bad = 1; break; }
    }
    free(line);
    #undef NEXT_LINE
    return 0;
}

int orc_bcf_read(const uint8_t *file, size_t flen, int tidy, int materialise, uint8_t **blob, size_t *blob_len, int64_t *n_rows)
{
    orc_bgzf_t bz;
    if (blob) { *blob = NULL; *blob_len = 0; }
    if (n_rows) *n_rows = 0;
    /* any content that is not a readable BCF2.2 header ends in "Failed to read BCF/VCF header" (bcf_reader.c:505); hts_open itself
     * only fails for a missing file (ORC_BCF_EOPEN is kept for that case at the surface) */
    const int is_bgzf = flen >= 18 && file[0] == 31 && file[1] == 139 && file[2] == 8 && (file[3] & 4);
    memset(&bz, 0, sizeof bz);
    if (is_bgzf && orc_bgzf_inflate_all(file, flen, &bz) < 0 && bz.len == 0) { orc_bgzf_free(&bz); return ORC_BCF_EHDR; }
    const uint8_t *u = is_bgzf ? bz.data : file; size_t ulen = is_bgzf ? bz.len : flen;
    if (ulen >= 16 && !memcmp(u, "##fileformat=VCF", 16)) {                     /* text VCF (hts_detect_format) */
        scan_t s; memset(&s, 0, sizeof s); s.tidy = tidy;
        buf_t recs = { 0 }, phi = { 0 }; int bad = 0;
        int rc = vcf_text_load(&s, u, ulen, &recs, &phi, &bad);
        if (rc < 0) { free(recs.p); free(phi.p); scan_free(&s); orc_bgzf_free(&bz); return rc; }
        s.pos_hi = (const int32_t *)phi.p; s.n_pos_hi = phi.n / 4;
        int status = scan_records(&s, recs.p, recs.n, 0, materialise);
        s.pos_hi = NULL;
        if (status == 0 && bad) status = -2;
        if (status == 0 && is_bgzf && bz.status < 0) status = bz.status;
        if (n_rows) *n_rows = s.n_rows;
        if (blob && materialise) {
            buf_t o = { 0 };
            uint32_t nc = (uint32_t)s.ncol; uint64_t nr = (uint64_t)s.n_rows, fr = 0; int32_t st = status; uint32_t ns = (uint32_t)s.h.n_smp;
            buf_push(&o, &nc, 4); buf_push(&o, &nr, 8); buf_push(&o, &st, 4); buf_push(&o, &fr, 8); buf_push(&o, &ns, 4);
            for (int i = 0; i < s.ncol; i++) ser_col(&o, &s.col[i], s.n_rows);
            { uint64_t nrec = (uint64_t)s.n_recs; buf_push(&o, &nrec, 8); buf_push(&o, s.rec_rid.p, s.rec_rid.n); buf_push(&o, s.rec_pos.p, s.rec_pos.n); buf_push(&o, s.rec_rlen.p, s.rec_rlen.n); }
            *blob = o.p; *blob_len = o.n;
        }
        free(recs.p); free(phi.p); scan_free(&s); orc_bgzf_free(&bz);
        return status;
    }
    if (!is_bgzf || ulen < 9 || memcmp(u, "BCF\2\2", 5)) { orc_bgzf_free(&bz); return ORC_BCF_EHDR; }
    size_t hlen = rd_u32(u + 5);
    if (ulen - 9 < hlen) { orc_bgzf_free(&bz); return ORC_BCF_EHDR; }
    char *txt = dupn((const char *)u + 9, hlen);
    scan_t s; memset(&s, 0, sizeof s); s.tidy = tidy;
    if (hdr_parse(&s.h, txt) < 0) { free(txt); hdr_free(&s.h); orc_bgzf_free(&bz); return ORC_BCF_EHDR; }
    free(txt);
    int rc = build_schema(&s);
    if (rc < 0) { scan_free(&s); orc_bgzf_free(&bz); return rc; }
    int status = scan_records(&s, u, ulen, 9 + hlen, materialise);
    if (status == 0 && bz.status < 0) status = bz.status;                     /* the byte stream itself ended on a BGZF error */
    if (n_rows) *n_rows = s.n_rows;
    if (blob && materialise) {
        buf_t o = { 0 };
        uint32_t nc = (uint32_t)s.ncol; uint64_t nr = (uint64_t)s.n_rows, fr = 9 + hlen; int32_t st = status; uint32_t ns = (uint32_t)s.h.n_smp;
        buf_push(&o, &nc, 4); buf_push(&o, &nr, 8); buf_push(&o, &st, 4); buf_push(&o, &fr, 8); buf_push(&o, &ns, 4);
        for (int i = 0; i < s.ncol; i++) ser_col(&o, &s.col[i], s.n_rows);
        { uint64_t nrec = (uint64_t)s.n_recs; buf_push(&o, &nrec, 8); buf_push(&o, s.rec_rid.p, s.rec_rid.n); buf_push(&o, s.rec_pos.p, s.rec_pos.n); buf_push(&o, s.rec_rlen.p, s.rec_rlen.n); }
        *blob = o.p; *blob_len = o.n;
    }
    scan_free(&s); orc_bgzf_free(&bz);
    return status;
}

void orc_free(void *p) { free(p); }
