/*
 * duckhts_amd.h -- thin C ABI between host code and the MI355X (gfx950) HIP scan path.
 *
 * This is the "inner" boundary named by the north star ("host C code calls hand-written HIP
 * kernels through a thin C-ABI").  The reference has no such seam of its own: its scan
 * callbacks call htslib directly.  Each entry point below names the reference interface whose
 * work it takes over (paths relative to the reference tree; htslib/ = third_party/htslib/):
 *
 *   dhts_open_* / dhts_bgzf_index   <- hts_open/bgzf_open + the BSIZE chain walk inside
 *                                      bgzf_read_block              htslib/bgzf.c:1155-1236
 *   dhts_bgzf_inflate_to_host       <- inflate_block/bgzf_uncompress + CRC-32 compare
 *                                                                    htslib/bgzf.c:762-824
 *   dhts_bam_open / dhts_bam_header <- sam_hdr_read -> bam_hdr_read  htslib/sam.c:229-342
 *                                      (+ @RG ID->SM dictionary: header.c:271-318, 2282-2312)
 *   dhts_bam_next_batch             <- the loop body of bam_read_function
 *                                      src/bam_reader.c:747-1035 over bam_read1 sam.c:779-855,
 *                                      sam_read1_bam sam.c:4124-4134, bam_aux_get sam.c:4834-4855
 *
 * The "outer" drop-in boundary (DuckDB C-API extension entry point duckhts_init_c_api and the
 * read_bam bind/init/local_init/scan callbacks, src/duckhts.c:48-93, src/bam_reader.c:1044-1068)
 * is declared in duckhts_extension.h and implemented on top of this ABI.
 *
 * All pointers are plain; no torch / HIP types cross the boundary.  Every function fails
 * loudly (non-zero return + dhts_error()) when no MI355X device / code object is available:
 * there is no CPU fallback.
 */
#ifndef DUCKHTS_AMD_H
#define DUCKHTS_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DHTS_ABI_VERSION 1

typedef struct dhts_ctx dhts_ctx;

/* read_bam core column ids, same order as src/bam_reader.c:187-202 */
enum {
    DHTS_BAM_QNAME = 0, DHTS_BAM_FLAG, DHTS_BAM_RNAME, DHTS_BAM_POS, DHTS_BAM_MAPQ, DHTS_BAM_CIGAR, DHTS_BAM_RNEXT,
    DHTS_BAM_PNEXT, DHTS_BAM_TLEN, DHTS_BAM_SEQ, DHTS_BAM_QUAL, DHTS_BAM_READ_GROUP_ID, DHTS_BAM_SAMPLE_ID,
    DHTS_BAM_CORE_COUNT
};
#define DHTS_BAM_ALL_COLS ((1u << DHTS_BAM_CORE_COUNT) - 1u)

/* variable-width column in device memory: row i = bytes[off[i] .. off[i]+len[i])            */
typedef struct {
    const uint32_t *off;     /* n+1 offsets (prefix sum of reserved widths)                    */
    const uint32_t *len;     /* n actual lengths (<= reserved)                                 */
    const uint8_t *bytes;
    uint64_t nbytes;         /* off[n]                                                          */
} dhts_strcol;

/* One projected column of a batch; DEVICE pointers, valid until the next call on the context.
 * scalar fixed-width : fixed[n_rows] in the native width of `type` (BOOLEAN 1 byte, INTEGER/FLOAT/ids 4, BIGINT/DOUBLE 8)
 * scalar VARCHAR     : off[n_rows+1] byte offsets into bytes
 * LIST               : off[n_rows+1] child offsets; children in child_fixed[child_n] (4-byte words) or, for
 *                      LIST(VARCHAR) in plain encoding, child_off[child_n+1] byte offsets into bytes                   */
typedef struct dhts_col {
    int32_t col;             /* schema column id                                                            */
    int32_t child_width;     /* bytes per child_fixed word: 0/4 = 4 (read_bcf), 8 = BIGINT children (read_bam tag lists)     */
    const uint8_t *valid;    /* n_rows bytes, 1 = valid                                                     */
    const void *fixed;
    const uint32_t *off;
    const uint8_t *bytes; uint64_t nbytes;
    const uint32_t *child_fixed; const uint32_t *child_off; uint64_t child_n;
    const uint8_t *child_valid; /* child_n bytes, 0 = NULL element; NULL pointer = every element valid (only VEP_* columns have NULL elements) */
} dhts_col;
typedef dhts_col dhts_bcf_col;

/* auxiliary_tags := true (src/bam_reader.c:967-1027): per row the tags that are not standard-tag columns, in record order, as TYPED
 * entries; the caller renders the value text (bam_aux_to_string bam_reader.c:140-183: %lld / %g / "subtype,v,v..."), because %g is
 * floating-point formatting.  kind: 0 int64, 1 f64 bits, 2 string bytes, 3 one char, 4 int64 array, 5 f64-bits array, 6 corrupt value
 * (the reference reads past the field there; rendered empty).  Device pointers.                                                  */
typedef struct {
    const uint8_t *valid;        /* n_rows: 0 = no entries (the MAP cell is NULL)                          */
    const uint32_t *off;         /* n_rows+1 entry offsets                                                 */
    uint64_t n_ent;
    const uint16_t *key;         /* two raw tag bytes, low byte first                                      */
    const uint8_t *kind, *sub;   /* sub = B-array subtype character                                        */
    const uint32_t *pay_off;     /* n_ent+1 payload offsets                                                */
    const uint8_t *payload; uint64_t payload_bytes;
} dhts_aux_map;

/* One decoded batch.  All pointers are DEVICE pointers owned by the context and stay valid
 * until the next dhts_bam_next_batch / dhts_bam_rewind / dhts_destroy on that context.        */
typedef struct {
    int64_t n_rows;
    int32_t status;          /* 0 = more data may follow, 1 = end of stream reached cleanly,
                                <0 = the stream ended on an error after these rows (the
                                reference ends the scan silently: bam_reader.c:754-766)        */
    int32_t seq_packed;      /* 1: seq.bytes holds the file's 4-bit base codes, seq.len the number of bases (dhts_bam_set_seq_packed) */
    const uint16_t *flag;    /* FLAG  USMALLINT */
    const int64_t *pos;      /* POS   BIGINT (1-based) */
    const int32_t *mapq;     /* MAPQ  INTEGER */
    const int64_t *pnext;    /* PNEXT BIGINT */
    const int64_t *tlen;     /* TLEN  BIGINT */
    const int32_t *tid;      /* dictionary id behind RNAME (-1 => "*") */
    const int32_t *mtid;     /* dictionary id behind RNEXT (-1 => "*") */
    const int32_t *rg_idx;   /* dictionary id behind SAMPLE_ID (index into header @RG table, -1 => NULL) */
    const uint64_t *rg_valid;/* validity words of READ_GROUP_ID (bit = 1 valid) */
    dhts_strcol qname, cigar, seq, qual, rg;
    uint64_t first_rec_uoff; /* absolute inflated-stream offset of the first row's record        */
    uint64_t end_uoff;       /* absolute inflated-stream offset just past the last row's record  */
    int32_t n_tag_cols;      /* standard-tag columns selected by dhts_bam_set_tag_columns         */
    int32_t qual_bits;       /* host batches only (dhts_bam_batch_fetch*, after dhts_bam_set_qual_packed): 0 = qual.bytes are the characters; 2 / 4 =
                              * qual.bytes is a 16-byte symbol table followed by the heap's characters as 2- / 4-bit codes, little end first:
                              * character k of the heap = table[(stream[k * bits / 8] >> (k * bits % 8)) & mask]; off / len count characters */
    const dhts_aux_map *aux_map; /* NULL unless dhts_bam_set_aux_map enabled it                       */
    const dhts_col *tag_cols;/* host array; BIGINT scalars in fixed (8 B), VARCHAR in off/bytes, LIST(BIGINT) in off/child_fixed (8 B words) */
    const uint32_t *ov_off;  /* overlap join (dhts_bam_set_overlap_intervals): n_rows+1 offsets into ov_ids, NULL when off */
    const uint32_t *ov_ids;  /* ids (positions in the caller's interval arrays) of the intervals overlapping each row     */
    uint64_t n_ov;
} dhts_bam_batch;

/* Host copy of the BAM header dictionaries (pointers owned by the context).                   */
typedef struct {
    int32_t n_ref;
    const char *const *ref_name;   /* NUL-terminated, as sam_hdr_tid2name would return          */
    const uint32_t *ref_len;
    const char *text; uint32_t l_text;
    int32_t n_rg;                  /* distinct @RG IDs, header order                             */
    const char *const *rg_id;
    const char *const *rg_sm;      /* NULL when the @RG has no (non-empty) SM                    */
    uint64_t first_rec_uoff;       /* inflated offset of the first alignment record              */
} dhts_bam_header;

/* kernel ids for dhts_kernel_time */
enum { DHTS_K_SIGSCAN = 0, DHTS_K_HUFF, DHTS_K_LZ, DHTS_K_TILES, DHTS_K_CORE, DHTS_K_SCAN, DHTS_K_STRINGS,
       DHTS_K_BCF_CHECK, DHTS_K_BCF_MEASURE, DHTS_K_BCF_WRITE, DHTS_K_COUNT };

int dhts_abi_version(void);
int dhts_device_count(void);                       /* number of visible HIP devices (0 => nothing will work) */
dhts_ctx *dhts_create(int device_id);              /* NULL if the device / code object is unavailable */
void dhts_destroy(dhts_ctx *);
const char *dhts_error(const dhts_ctx *);          /* last error message ("" if none) */

/* ---- input: compressed bytes become resident in HBM -------------------------------------- */
int dhts_open_path(dhts_ctx *, const char *path);                  /* pread (reader threads) -> pinned -> HBM, whole file */
/* bytes [off, off+len) of the file (len = 0: to its end) become the resident bytes: a rank of a multi-GPU scan stages only its
 * own block range plus a halo; a bind that only needs the header stages the first megabytes (hts_open + the reads under
 * bgzf_read_block, htslib/bgzf.c:1004-1239, restricted to a window of the file) */
int dhts_open_path_range(dhts_ctx *, const char *path, uint64_t off, uint64_t len);
/* Staging that overlaps the scan (the analogue of bgzf_mt_reader running ahead of the consumer, htslib/bgzf.c:1598-1738):
 * dhts_open_path_async starts reader threads and returns; dhts_stage_wait blocks until at least min_bytes contiguous bytes of the file
 * are resident (or staging has finished: *done) and returns that count; dhts_bgzf_index_staged builds -- later extends -- the block table
 * over the resident prefix; dhts_bam_next_batch then serves the blocks known so far and reports status 0 ("more may follow") until the
 * table covers the whole file; dhts_blocks_ahead = blocks in the table that the scan has not consumed yet. */
int dhts_open_path_async(dhts_ctx *, const char *path);
int64_t dhts_stage_wait(dhts_ctx *, uint64_t min_bytes, int *done);
int64_t dhts_bgzf_index_staged(dhts_ctx *);
int64_t dhts_blocks_ahead(const dhts_ctx *);
/* one rank's share of ONE file: resident bytes = the header blocks file[0, header_bytes) followed by the rank's own
 * window of whole BGZF blocks plus a 4 MiB halo; cut points t_r = header_bytes + (size - header_bytes) * r / world.  Follow with
 * dhts_bgzf_index, dhts_bam_open and dhts_bam_set_file_shard(rank, world).  header_bytes: dhts_bam_header_bytes of a context that has
 * opened (the head of) the file.  dhts_voffset turns a position of the inflated stream into a BGZF virtual offset (bgzf_tell,
 * htslib/bgzf.h): adjacent ranks hand off end == first-record as virtual offsets. */
int dhts_open_path_shard(dhts_ctx *, const char *path, int rank, int world, uint64_t header_bytes);
/* A region query stages only what it needs (htslib seeks to each index chunk, hts.c:4320-4607): dhts_bam_region_segments -- on a context
 * that holds the header (dhts_bam_open) and the regions (dhts_bam_set_regions) -- turns the index into file byte ranges: beg[k] = offset
 * of a window's first BGZF block, end[k] = offset of its LAST block (~0 = to the end of the file), *count = -1 when the whole file is
 * needed; dhts_open_path_segments stages the header blocks file[0, header_bytes) and those windows (whole blocks, merged where they
 * touch).  dhts_bgzf_index / dhts_bam_open / dhts_bam_set_regions / dhts_bam_load_index follow as for a whole file; virtual offsets
 * (dhts_voffset, the index) stay those of the FILE. */
int dhts_bam_region_segments(dhts_ctx *, const void *index_bytes, uint64_t n, uint64_t *beg, uint64_t *end, int64_t cap, int64_t *count);
int dhts_open_path_segments(dhts_ctx *, const char *path, uint64_t header_bytes, const uint64_t *beg, const uint64_t *end, int64_t n);
/* the cut itself, host only (no device): rank r stages file[win_begin, win_end) and owns the blocks that start in [win_begin, own_end) */
int dhts_shard_window(const char *path, int rank, int world, uint64_t header_bytes, uint64_t *win_begin, uint64_t *win_end, uint64_t *own_end);
int dhts_bam_set_file_shard(dhts_ctx *, int rank, int world);
uint64_t dhts_bam_header_bytes(const dhts_ctx *);
uint64_t dhts_voffset(const dhts_ctx *, uint64_t uoff);
int dhts_open_host(dhts_ctx *, const void *bytes, uint64_t n);     /* copy host bytes -> HBM */
int dhts_open_tiled(dhts_ctx *, const void *head, uint64_t n_head, const void *body, uint64_t n_body, int reps,
                    const void *tail, uint64_t n_tail);            /* HBM = head + body x reps + tail (benchmark helper) */
uint64_t dhts_resident_bytes(const dhts_ctx *);

/* ---- BGZF -------------------------------------------------------------------------------- */
int64_t dhts_bgzf_index(dhts_ctx *);               /* block discovery on resident bytes; returns n_blocks or <0 */
int dhts_bgzf_table(const dhts_ctx *, uint64_t *coff, uint32_t *clen, uint32_t *isize, int64_t cap);  /* host copies */
/* inflate blocks [blk0, blk0+nblk) and copy the concatenated payload to host (parity tests) */
int64_t dhts_bgzf_inflate_to_host(dhts_ctx *, int64_t blk0, int64_t nblk, uint8_t *out, uint64_t cap, int32_t *blk_status);

/* ---- read_bam ------------------------------------------------------------------------------ */
int dhts_bam_open(dhts_ctx *);                                     /* header + dictionaries; positions the scan at the first record */
int dhts_bam_header_get(const dhts_ctx *, dhts_bam_header *out);
int dhts_bam_set_shard(dhts_ctx *, int rank, int world);           /* scan only this rank's BGZF block range */
int dhts_bam_set_block_range(dhts_ctx *, int64_t b0, int64_t b1, int speculative_start);   /* explicit shard: blocks [b0,b1) */
/* the shard cut itself (host arithmetic only): blocks [*b0,*b1) of rank, balanced by compressed bytes */
int dhts_shard_cut(const uint64_t *coff, int64_t n_blocks, uint64_t comp_len, int rank, int world, int64_t *b0, int64_t *b1);
/* region := 'chr:beg-end,...' (sam_itr_regarray, htslib sam.c:1763-1790 over hts_reglist_create region.c:177-260, hts_parse_region
 * hts.c:3995-4150, the predicate of hts_itr_multi_next hts.c:4575-4592 + bam_endpos sam.c:668-673).  Rows are filtered on the device.
 * Returns 0, 1 when no region names a known reference (reference: "No reads found for region(s): ..."), <0 on error.          */
int dhts_bam_set_regions(dhts_ctx *, const char *regions);
/* BAI or CSI bytes (hts_idx_load: hts.c:2920-3055; a BGZF-compressed CSI is inflated on the device): narrows the scan window to
 * the chunks of the bins the regions touch (reg2bins hts.c:3142-3213 + BAI linear index); optional for exactness, call after
 * dhts_bam_set_regions / dhts_bcf_set_region.                                                                                    */
int dhts_bam_load_index(dhts_ctx *, const void *index_bytes, uint64_t n);
/* the scan range the regions + index produced: number of disjoint windows (the merged chunk list of hts_itr_multi_bam, hts.c:3597-3739;
 * chunks closer than DHTS_WINDOW_GAP_MB, default 32, are scanned as one window) and the BGZF blocks they cover */
int dhts_scan_window_stats(const dhts_ctx *, int64_t *n_windows, int64_t *n_blocks);
/* BCF: CSI bytes, narrows the window as above (optional).  bgzipped VCF TEXT: TBI, or CSI with a tabix header (tbx_index_load3, tbx.c:552-597)
 * -- REQUIRED for a region other than ".": the region names a sequence of the INDEX (tbx_itr_querys -> tbx_name2id), so it is resolved here;
 * the index's sequences missing from the header become ##contig lines as in vcf_hdr_read (vcf.c:2649-2668).  Rows are decided on the
 * device by the interval tbx_parse1 gives a line (tbx.c:96-312: REF length, SVLEN of <DEL>/<DUP>/<CNV>/<INV>, FORMAT/LEN, INFO/END).
 * Returns 0, 1 = the index does not know the region's sequence (no iterator: the reference skips the region), <0 on error.            */
int dhts_bcf_load_index(dhts_ctx *, const void *index_bytes, uint64_t n);
/* What a region query of read_bcf has to stage instead of the whole file (the reference seeks to the index chunks): the compressed bytes of
 * the header blocks, and the union of the index windows of every region of 'a,b,...' as file byte ranges for dhts_open_path_segments
 * (same conventions as dhts_bam_region_segments; *count = -1: stage the whole file).  Both on a context that holds the header. */
uint64_t dhts_bcf_header_bytes(const dhts_ctx *);
int dhts_bcf_region_segments(dhts_ctx *, const char *regions, const void *index_bytes, uint64_t n, uint64_t *beg, uint64_t *end, int64_t cap, int64_t *count);
/* standard_tags := true (src/bam_reader.c:54-70, 920-966): the reference's 56-entry tag table, in its order */
int dhts_bam_std_tag_count(void);
int dhts_bam_std_tag_info(int idx, char name[3], char *type, char *subtype);   /* type: 'i' BIGINT, 'Z'/'A' VARCHAR, 'B' LIST(BIGINT) */
int dhts_bam_set_tag_columns(dhts_ctx *, const int32_t *std_tag_ids, int32_t n);  /* tag columns materialised by the next batches (default none) */
int dhts_bam_set_aux_map(dhts_ctx *, int enable, int exclude_standard_tags);        /* AUXILIARY_TAGS entries in the next batches */
int dhts_bam_rewind(dhts_ctx *);
/* BAI writer (the reference: src/hts_index_builder.c over sam_index_build, htslib sam.c:989-1027 +
 * hts_idx_push / hts_idx_finish / idx_save_core hts.c:2315-2818).  One whole-file scan of the open BAM; returns the size of the
 * index (bytes of a .bai file) kept in the context, or <0 (unsorted input, a record beyond 2^29, ... as in hts_idx_push).       */
int64_t dhts_bam_build_index(dhts_ctx *);
int64_t dhts_bam_build_index_csi(dhts_ctx *, int min_shift);   /* sam_index_build3(fn, fnidx, min_shift): > 0 writes a CSI (depth from the longest @SQ), <= 0 the BAI */
int dhts_bam_index_bytes(dhts_ctx *, uint8_t *out, uint64_t cap);
/* Interval overlap join on the scan (SURVEY 8(f) item 1 / BASELINE config 5).  The reference vendors cgranges
 * (third_party/cgranges, cr_add / cr_index / cr_overlap cgranges.c:255-297) as the model for joining read_bam rows with
 * read_bed intervals (src/interval_udf.c:344-426) but registers no SQL function for it yet; this entry point is what such a
 * function would bind.  Intervals are (tid, beg, end) half-open, 0-based, tid = index in the BAM header's reference table
 * (intervals with tid outside it never match).  Every following batch carries, per row, the ids of the intervals i with
 * beg_i < read_end && read_beg < end_i on the read's contig (read interval = [pos, bam_endpos)), i.e. cr_overlap's answer as
 * a set.  n = 0 switches the join off.  Host pointers; copied.                                                              */
int dhts_bam_set_overlap_intervals(dhts_ctx *, const int32_t *tid, const int64_t *beg, const int64_t *end, int64_t n);
/* The intervals from BED text, i.e. the rows read_bed would return for it (src/interval_udf.c:330-426: next_bed_line skips empty lines and
 * those that begin with '#', "track" or "browser", 141-147; a line with fewer than 3 tab-delimited fields is read_bed's error, 358-365;
 * start / end through strtoll over the whole field, else NULL, 127-139).  The device splits the text into lines and parses chrom / start /
 * end of each (bed_intervals, vcf_text.hip); chrom names are mapped to the BAM header's reference ids.  Interval id = row number of
 * read_bed; rows with a NULL start or end, or a chrom the header does not have, never match.  Returns the number of rows, < 0 on error.
 * _path: a plain or BGZF / gzip-compressed BED file (what hts_open + hts_getline accept, 330-337).                                   */
int64_t dhts_bam_set_overlap_bed(dhts_ctx *, const uint8_t *text, uint64_t n);
int64_t dhts_bam_set_overlap_bed_path(dhts_ctx *, const char *path);
/* Next batch of rows (<= max_blocks BGZF blocks of input; 0 = default 16,384, at most 24,576).  colmask = projection pushdown, bit i
 * = read_bam core column i (DHTS_BAM_*): the fixed-width columns are always produced; the string heaps (QNAME, CIGAR, SEQ, QUAL,
 * READ_GROUP_ID) of columns that are not projected are not written, and with none of them projected the string pass is skipped
 * (what the reference does per column in its writer switch, src/bam_reader.c:783-918).                                            */
int dhts_bam_next_batch(dhts_ctx *, int64_t max_blocks, uint32_t colmask, dhts_bam_batch *out);

/* ---- read_bcf ------------------------------------------------------------------------------
 *   dhts_bcf_open                   <- bcf_hdr_read -> bcf_hdr_parse      htslib/vcf.c:1710-1769, 1410-1489, 831-1024
 *                                      + schema construction              src/bcf_reader.c:540-760, src/include/vcf_types.h
 *   dhts_bcf_next_batch             <- the loop body of bcf_read_function src/bcf_reader.c:1155-2049 over bcf_read
 *                                      vcf.c:2255-2265 (bcf_read1_core 1874-1911, bcf_record_check 2040-2212),
 *                                      bcf_unpack 4234-4302 and the bcf_get_info_* / bcf_get_format_* getters 6056-6248
 * VEP_* columns (a CSQ / BCSQ / ANN / VEP / vep INFO tag in the header, src/vep_parser.c:100-182, src/bcf_reader.c:582-603,
 * 1463-1541): LIST columns with one element per transcript and NULL elements (dhts_bcf_col.child_valid) for missing fields. */

/* element types of a read_bcf column (values of DUCKDB_TYPE_* in duckdb.h) */
enum { DHTS_T_BOOLEAN = 1, DHTS_T_INTEGER = 4, DHTS_T_BIGINT = 5, DHTS_T_FLOAT = 10, DHTS_T_DOUBLE = 11, DHTS_T_VARCHAR = 17 };
/* how a column's payload is coded on the device */
enum { DHTS_ENC_PLAIN = 0,      /* payload is the value itself                                              */
       DHTS_ENC_CONTIG = 1,     /* int32 contig id   -> dhts_bcf_info.contig_name[id]   (CHROM)               */
       DHTS_ENC_DICT = 2,       /* int32 dictionary id -> dhts_bcf_info.dict_name[id], -1 = "PASS" (FILTER)    */
       DHTS_ENC_SAMPLE = 3,     /* int32 sample index -> dhts_bcf_info.sample_name[i]   (SAMPLE_ID)           */
       DHTS_ENC_FLOAT_TEXT = 4 };/* LIST(FLOAT) whose elements arrive as their decimal TEXT (child_off / bytes, like a LIST(VARCHAR)):
                                   the consumer converts every non-NULL element with (float)strtod(text, &end), NaN unless the whole
                                   text converts (vep_parse_float src/vep_parser.c:222-235).  Used by the Float fields of VEP_* columns;
                                   host text conversion, like the %g of AUXILIARY_TAGS. */

typedef struct {
    const char *name;        /* result column name, as duckdb_bind_add_result_column receives it           */
    int32_t type;            /* DHTS_T_* of the element                                                     */
    int32_t is_list;         /* LIST(type)                                                                  */
    int32_t encoding;        /* DHTS_ENC_*                                                                  */
    int32_t reserved;
} dhts_bcf_colinfo;

typedef struct {
    int32_t n_cols; const dhts_bcf_colinfo *cols;
    int32_t n_contigs; const char *const *contig_name;      /* NULL entries = holes left by IDX= numbering    */
    int32_t n_dict; const char *const *dict_name;           /* FILTER/INFO/FORMAT dictionary (BCF_DT_ID)      */
    int32_t n_samples; const char *const *sample_name;
    int32_t tidy;                                           /* rows are (record, sample) pairs                 */
    uint64_t first_rec_uoff;                                /* inflated offset of the first record             */
} dhts_bcf_info;


typedef struct {
    int64_t n_rows;
    int32_t status;          /* same convention as dhts_bam_batch.status                                    */
    int32_t n_cols;          /* projected columns, in projection order                                      */
    const dhts_bcf_col *cols;/* host array of n_cols descriptors (device pointers inside)                   */
    uint64_t first_rec_uoff, end_uoff;
} dhts_bcf_batch;

int dhts_bcf_open(dhts_ctx *, int tidy_format);                      /* header + dictionaries + schema; positions the scan at the first record */
int dhts_bcf_info_get(const dhts_ctx *, dhts_bcf_info *out);
/* QUAL over PCIe by the batch's own alphabet: when on, dhts_bam_batch_fetch / _fetch_begin look at which characters the batch's QUAL heap holds
 * (one pass on the device) and ship 2 bits per character when there are at most 4, 4 bits when at most 16 (binned base qualities: current Illumina
 * instruments write 4-8 distinct values), the characters themselves otherwise; dhts_bam_batch.qual_bits says which.  The batch in HBM is unchanged. */
void dhts_bam_set_qual_packed(dhts_ctx *, int on);
void dhts_bam_set_seq_packed(dhts_ctx *, int on);                    /* SEQ stays 4 bits per base in the batch: (l + 1) / 2 bytes per row, high nibble first, "=ACMGRSVTWYHKDBN"; len = bases, 0 = "*" */
void dhts_set_super_blocks(dhts_ctx *, int64_t n_blocks);             /* phase A look-ahead (default 524,288 blocks = 67 GB of scratch for a 10 GB file); the table functions use 196,608 */
int dhts_bcf_is_text(const dhts_ctx *);                              /* after dhts_bcf_open: 0 binary BCF, 1 bgzipped VCF text, 2 plain VCF text (also: VCF text inside plain, non-BGZF gzip -- inflated by the serial device decoder at open, bgzf.c:828-905) */
int dhts_bcf_set_projection(dhts_ctx *, const int32_t *col_ids, int32_t n);   /* default: every schema column */
int dhts_bcf_set_block_range(dhts_ctx *, int64_t b0, int64_t b1, int speculative_start);
/* ONE region of read_bcf(region := 'a,b,...'): the reference chains single-region iterators in the order given (src/bcf_reader.c:
 * 1327-1345; overlapping regions repeat rows).  bcf_itr_querys + the overlap test of hts_itr_next (hts.c:4287-4300) over bcf_readrec
 * (vcf.c:2267-2276).  0 = set, 1 = no iterator for this region (unknown contig: skipped by the reference), NULL/"" clears.
 * On VCF text the name is resolved by dhts_bcf_load_index (see there), which must follow before the scan.                          */
int dhts_bcf_set_region(dhts_ctx *, const char *region);
int dhts_bcf_rewind(dhts_ctx *);
int dhts_bcf_next_batch(dhts_ctx *, int64_t max_blocks, dhts_bcf_batch *out);

/* ---- batches in host memory ---------------------------------------------------------------------
 * dhts_host_alloc: pinned (page-locked, portable) host memory from a process-wide pool; dhts_host_free returns it to the pool.
 * dhts_bam_batch_fetch copies the projected core columns (colmask as in dhts_bam_next_batch) of `b` into `dst` -- every copy
 * queued, one wait -- and fills `out` = `b` with HOST pointers for those columns (NULL for unprojected ones); tag columns, the
 * auxiliary map and the overlap lists keep their device pointers.  dhts_bam_batch_host_bytes = the room `dst` needs.
 * This is the seam a DuckDB scan callback fills DataChunks from (src/bam_reader.c:783-918 reads the same values out of bam1_t). */
/* NUMA placement of the host side of a device (multi-GPU hosts have several sockets): the node a device hangs off (hipDeviceGetPCIBusId ->
 * /sys/bus/pci/devices/<id>/numa_node; -1 unknown), and binding of the CALLING thread -- and of the threads it starts afterwards: a
 * producer's staging readers -- to that node's CPUs.  Pinned buffers (dhts_host_alloc) remember the node of the thread that allocated them
 * and are handed to threads of the same node first.  0 bound, 1 nothing to do (no NUMA information, DHTS_NUMA=0), -1 failed.
 * The reference's reader threads (hts_set_threads, src/bam_reader.c:584, 625) are placed by the OS. */
int dhts_device_numa_node(int device);
int dhts_bind_thread_to_node(int node);
int dhts_bind_thread_near_device(int device);
void *dhts_host_alloc(uint64_t nbytes);
void dhts_host_free(void *p);
/* Device buffers of destroyed contexts and freed pinned buffers are kept in process-wide pools (a context per query would otherwise pay
 * hipMalloc / hipHostMalloc of gigabytes each time); this returns every idle pooled buffer to the driver. */
void dhts_release_pools(void);
/* 1 when the context's file bytes were taken over from an earlier context of this process instead of being read and copied again: a file
 * staged WHOLE stays in HBM (in the device pool, tagged with device / inode / size / mtime / ctime) after its context is destroyed, until
 * the pool needs the room (least recently used first) or dhts_release_pools.  DHTS_FILE_CACHE=0 disables it. */
int dhts_resident_from_cache(const dhts_ctx *);
int dhts_device_mem_info(int device, uint64_t *free_bytes, uint64_t *total_bytes);   /* hipMemGetInfo after a device synchronise */
/* CSI writer (src/hts_index_builder.c -> bcf_index_build3; htslib vcf.c:4657-4688, hts.c hts_idx_push / hts_idx_finish / idx_save_core):
 * one scan of the open BCF; the bytes (dhts_bam_index_bytes) are the UNCOMPRESSED index, dhts_bgzf_wrap (host only) turns raw bytes into
 * a valid BGZF file -- stored DEFLATE blocks + the EOF block -- which is what a .csi on disk is. */
/* tabix index of a bgzipped line format other than VCF (tbx_index_build3 with tbx_conf_bed / _gff / _sam or custom columns, tbx.c:437-541):
 * preset = TBX_GENERIC 0 | TBX_SAM 1 (| TBX_UCSC 0x10000 for 0-based half-open coordinates), sc / bc / ec = 1-based sequence, begin and end
 * columns, meta_char / line_skip = lines the indexer passes over.  The context needs dhts_open_path + dhts_bgzf_index only.  The device finds
 * the lines and their intervals (tbx_parse1), the host numbers the names in order of first appearance.  min_shift <= 0: TBI, else CSI.
 * Returns the size of the (uncompressed) index, fetched with dhts_bam_index_bytes. */
int64_t dhts_tabix_build_index(dhts_ctx *, int preset, int sc, int bc, int ec, int meta_char, int line_skip, int min_shift);
/* bgzip / bgunzip (src/bgzip.c:88-293 -> bgzf_write / bgzf_read, htslib bgzf.c): DEFLATE *compression* on the device, one wave per BGZF block
 * (0xff00 input bytes each; fixed-Huffman code, hash-table match finder; a block that does not shrink is stored), CRC-32 and ISIZE included,
 * EOF block appended.  level 0 stores; -1 and 1..9 all select the one compressing setting.  The bytes are not zlib's bytes (no two DEFLATE
 * implementations agree); every reader returns the input.
 * dhts_bgzf_compress: host buffer -> host buffer; returns the size, or an upper bound when out is NULL / cap is below that bound.
 * dhts_bgzip_file / dhts_bgunzip_file: 0, -2 cannot open input (or gzip without a block to read), -3 cannot open output, -4 read / block error, -5 write error.
 * bgunzip reads what bgzf_read reads: BGZF, plain gzip (any content; serial device decoder), and files that are not gzip at all (handed through). */
int64_t dhts_bgzf_compress(dhts_ctx *, const void *raw, uint64_t n, int level, void *out, uint64_t cap);
int dhts_bgzip_file(dhts_ctx *, const char *in_path, const char *out_path, int level, int64_t *bytes_in, int64_t *bytes_out);
int dhts_bgunzip_file(dhts_ctx *, const char *in_path, const char *out_path, int64_t *bytes_in, int64_t *bytes_out);
int64_t dhts_bcf_build_index(dhts_ctx *, int min_shift);          /* min_shift <= 0: 14, the default of bcf_index_build */
/* On bgzipped VCF text the same call is the tabix writer (tbx_index_build3 with tbx_conf_vcf, tbx.c:437-510): min_shift <= 0 writes a TBI
 * (14 / 5 levels), > 0 a CSI whose depth follows the ##contig lengths (hts_adjust_csi_settings) and whose aux block is the tabix header;
 * sequence ids in the order of first appearance, intervals by the tbx_parse1 rule above. */
int64_t dhts_bgzf_wrap(const void *raw, uint64_t n, void *out, uint64_t cap);   /* returns the size needed / written */
/* read_bcf: the room the LAST batch of the context needs, and its read-back: out_cols[b->n_cols] = b->cols with HOST pointers (four queued
 * copies -- validity, fixed payloads, offsets, children / bytes -- and one wait) */
uint64_t dhts_bcf_batch_host_bytes(const dhts_ctx *);
int dhts_bcf_batch_fetch(dhts_ctx *, const dhts_bcf_batch *b, void *dst, uint64_t cap, dhts_bcf_col *out_cols);
int dhts_bcf_batch_fetch_begin(dhts_ctx *, const dhts_bcf_batch *dev_batch, void *dst, uint64_t cap, dhts_bcf_col *host_cols, int slot);   /* overlapped, as dhts_bam_batch_fetch_begin */
int dhts_bcf_batch_fetch_wait(dhts_ctx *, int slot);
uint64_t dhts_bam_batch_host_bytes(const dhts_bam_batch *b, uint32_t colmask);
int dhts_bam_batch_fetch(dhts_ctx *, const dhts_bam_batch *b, uint32_t colmask, void *dst, uint64_t cap, dhts_bam_batch *out);
/* the same read-back overlapped with the next batch: the columns are gathered into one of two device snapshots (slot 0 / 1) and leave for
 * the host as ONE copy on a separate stream; _begin returns at once (out points into dst), _wait(slot) when the bytes have landed.  Call
 * dhts_bam_next_batch in between: PCIe and the scan of the next batch then run side by side. */
int dhts_bam_batch_fetch_begin(dhts_ctx *, const dhts_bam_batch *dev_batch, uint32_t colmask, void *dst, uint64_t cap, dhts_bam_batch *host_batch, int slot);
int dhts_bam_batch_fetch_wait(dhts_ctx *, int slot);

/* ---- utilities ------------------------------------------------------------------------------ */
int dhts_memcpy_d2h(dhts_ctx *, void *dst, const void *src_dev, uint64_t n);
int dhts_sync(dhts_ctx *);
/* accumulated device time of one kernel family, measured with HIP events on the context's stream */
double dhts_kernel_time_ms(const dhts_ctx *, int kernel_id, int64_t *launches);
void dhts_kernel_time_reset(dhts_ctx *);
void dhts_set_timing(dhts_ctx *, int enabled);

#ifdef __cplusplus
}
#endif
#endif
