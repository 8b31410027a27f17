/*
 * dhts_oracle.h -- CPU restatement of the DuckHTS read_bam / read_bcf scan path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ may be imported, linked or
 * executed by the product (duckhts_amd/); only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it, as the checker.
 *
 * Parity pinning: this restatement is pinned against the reference's own
 * fixtures (tests/golden/: range.bam + duckhts.test expectations, range.out /
 * range.out2, bgzf_boundaries{1,2,3}.bam <-> ce#1.sam, no_hdr_sq_1.bam <->
 * no_hdr_sq_1.expected.sam, vcf_file.bcf <-> vcf_file.vcf) and, for the
 * third-party DEFLATE/CRC-32 arithmetic (system zlib, version unpinned by the
 * reference: CMakeLists.txt:130, vcpkg.json:6), against CPython's zlib module.
 * The reference itself (htslib 1.23 + src/bam_reader.c) is NOT buildable here
 * under the round rules: htslib needs generated config.h / version.h.
 */
#ifndef DHTS_ORACLE_H
#define DHTS_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- DEFLATE (RFC 1951) + CRC-32 (RFC 1952) ------------------------------ */
/* returns 0 on success (final block reached), <0 on malformed/overflow.      */
int orc_inflate_raw(const uint8_t *src, size_t slen, uint8_t *dst, size_t dcap, size_t *dlen);
uint32_t orc_crc32(uint32_t crc, const uint8_t *p, size_t n);

/* ---- BGZF container (htslib bgzf.c:896-903, 1004-1239) ------------------- */
typedef struct {
    uint8_t *data;      /* concatenated inflated payload of all good blocks   */
    size_t len;
    int64_t n_blocks;   /* blocks consumed (incl. empty)                       */
    int status;         /* 0 = clean EOF, <0 = stream ended by an error        */
    int has_eof_marker; /* last block is the 28-byte EOF block                 */
    /* per-block table (for kernel-level parity tests)                        */
    int64_t *coff;      /* compressed offset of block i                        */
    int32_t *clen;      /* BSIZE+1                                             */
    int32_t *ulen;      /* inflated length                                     */
} orc_bgzf_t;

int orc_bgzf_inflate_all(const uint8_t *file, size_t flen, orc_bgzf_t *out);
void orc_bgzf_free(orc_bgzf_t *b);

/* ---- column containers ---------------------------------------------------- */
typedef struct {
    uint64_t *off;      /* n+1 offsets into bytes                               */
    uint8_t *bytes;
    uint8_t *valid;     /* n bytes, 1 = valid, 0 = NULL                         */
    size_t n, cap_n, nbytes, cap_bytes;
} orc_strcol_t;

/* read_bam result: the 13 core columns of src/bam_reader.c:514-526 in
 * sequential (no index) mode, all rows of the file in file order.          */
typedef struct {
    int64_t n_rows;
    int status;           /* 0 clean EOF; <0 first error (rows before it kept, bam_reader.c:754-766) */
    int32_t n_ref;
    uint16_t *flag;       /* USMALLINT */
    int64_t *pos;         /* BIGINT pos+1 */
    int32_t *mapq;        /* INTEGER */
    int64_t *pnext;       /* BIGINT mpos+1 */
    int64_t *tlen;        /* BIGINT */
    int32_t *tid, *mtid;  /* dictionary ids behind RNAME / RNEXT */
    int64_t *rec_off;     /* offset of each record's block_size word in the inflated stream */
    orc_strcol_t qname, rname, cigar, rnext, seq, qual, rg, sample;
    /* header dictionary */
    orc_strcol_t ref_names;
    int32_t *ref_len;
    char *text; size_t l_text;
    int64_t first_rec_off;  /* inflated offset of the first alignment record */
} orc_bam_t;

int orc_bam_read(const uint8_t *file, size_t flen, orc_bam_t *out);
int orc_bam_read_path(const char *path, orc_bam_t *out);
void orc_bam_free(orc_bam_t *b);

/* timing helper for bench.py's cpu_baseline ("port"): decodes the whole file
 * (inflate + crc + record decode + 13-column materialisation), returns rows. */
int64_t orc_bam_scan_count(const uint8_t *file, size_t flen, int *status);
/* CRC-32 digests of the 13 columns of a whole-file scan (layout: dhts_oracle.c), out[32] */
int orc_bam_digest(const uint8_t *file, size_t flen, uint32_t *out);
/* timing leg only: n_threads BGZF inflate workers + one scan thread (the reference opens with hts_set_threads(fp, 2), src/bam_reader.c:625) */
int64_t orc_bam_scan_count_mt(const uint8_t *file, size_t flen, int n_threads, int *status);
/* timing leg only: inflate + crc32 through system zlib (the reference's own dependency) instead of the RFC restatement */
int orc_use_system_zlib(int on);

/* read_bam(standard_tags := true): the 56 typed tag columns (src/bam_reader.c:54-70, 920-966) as a canonical column blob */
int orc_bam_read_std_tags(const uint8_t *file, size_t flen, uint8_t **blob, size_t *blob_len);

/* read_bam(auxiliary_tags := true): keys and rendered values of the non-standard tags (src/bam_reader.c:967-1027, 140-183) */
int orc_bam_read_aux_map(const uint8_t *file, size_t flen, int exclude_standard, uint8_t **blob, size_t *blob_len);

/* ---- read_bcf (bcf_oracle.c) ---------------------------------------------- */
/* Sequential read_bcf scan of a whole BCF file; `blob` receives the canonical serialisation of every schema column
 * (layout documented above orc_bcf_read in bcf_oracle.c; free with orc_free).  materialise = 0 only frames and
 * validates records (row count).  Returns 0 clean EOF, -2 stream ended at a bad record (rows before it kept),
 * -100 not a BGZF/BCF file, -101 header unreadable. */
int orc_bcf_read(const uint8_t *file, size_t flen, int tidy, int materialise, uint8_t **blob, size_t *blob_len, int64_t *n_rows);
void orc_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
