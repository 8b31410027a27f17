// bam_tags.hip -- read_bam(standard_tags := true): the 56 typed SAM-tag columns on MI355X.
//
// Replaces src/bam_reader.c:920-966 over htslib bam_aux_get (sam.c:4834-4855), bam_aux2i / 2A / 2Z (5109-5141),
// bam_auxB_len / bam_auxB2i / bam_auxB2f (5143-5171).  Integer/byte work only:
//   bam_tag_dir   one lane per row: ONE walk of the aux area leaves, per requested tag, the offset of its type byte (first
//                 occurrence wins; a tag reached through / stored as a corrupt value reads as absent, exactly where bam_aux_get
//                 would return NULL), with the CG tag spliced out after a long-CIGAR swap (sam.c:716-720);
//   bam_tag_cells one lane per (row, tag column): validity, BIGINT payloads, byte / child counts; after the matrix scan the
//                 same kernel writes strings and list children in place.
#pragma once

struct TagColDev {
    int32_t kind;                 // 'i', 'Z', 'A', 'B' (column kind from the standard-tag table, src/bam_reader.c:54-70)
    int32_t slot;                 // directory slot
    int32_t sa_cnt, sa_bytes;     // rows of the count matrix or -1
    uint8_t *valid; int64_t *fixed; uint8_t *bytes; uint64_t *child;
};
#define TAG_SEEN_NULL 0xffffffffu

extern "C" __global__ void __launch_bounds__(256)
bam_tag_dir(BamStream st, const uint32_t *rec_off, int64_t nrows, const uint16_t *codes, int ncodes, uint32_t *dir, uint32_t stride) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= nrows) return;
    GSrc s; s.g = st.u;
    const uint64_t o = rec_off[row];
    for (int k = 0; k < ncodes; k++) dir[(size_t)k * stride + row] = 0;
    RecInfo r;
    if (rec_check_t(st, s, o, r, true) != REC_OK) return;
    const uint64_t end = o + 4ull + r.block_len;
    const uint64_t aux = o + 36 + r.l_qname + 4ull * r.n_cigar + (((uint64_t)r.l_seq + 1) >> 1) + (uint64_t)r.l_seq;
    const uint64_t sb = r.cg_beg, se = r.cg_end;
    if ((end - aux) - (se - sb) <= 2) return;
    uint64_t p = aux;
    if (p == sb) p = se;
    p += 2;
    for (;;) {
        const uint64_t e = aux_skip_t(s, p, end);
        if (e == NONE64) break;                               // corrupt from here on: this and every later tag read as absent
        const uint8_t ty = s.u8(p);
        const uint16_t code = (uint16_t)(s.u8(p - 2) | (s.u8(p - 1) << 8));
        for (int k = 0; k < ncodes; k++) if (codes[k] == code) {
            const size_t q = (size_t)k * stride + row;
            if (dir[q] == 0) dir[q] = ((ty == 'Z' || ty == 'H') && s.u8(e - 1) != 0) ? TAG_SEEN_NULL : (uint32_t)(p - o);
            break;
        }
        uint64_t nx = e;
        if (nx == sb) nx = se;
        if (end - nx <= 2) break;
        p = nx + 2;
    }
}

struct TagCellArgs { const uint32_t *rec_off; const uint32_t *dir; uint32_t stride; int64_t nrows; uint32_t *lens; const uint32_t *offs; uint32_t ostride; const TagColDev *cols; };

__device__ __forceinline__ int64_t tag_int_val(const uint8_t *u, uint8_t type, uint64_t p, uint32_t idx) {     // get_int_aux_val sam.c:5093-5107
    switch (type) {
    case 'c': return (int8_t)u[p + idx];
    case 'C': return u[p + idx];
    case 's': return (int16_t)b_u16(u + p + 2ull * idx);
    case 'S': return (int64_t)b_u16(u + p + 2ull * idx);
    case 'i': return (int32_t)ldu32(u + p + 4ull * idx);
    case 'I': return (int64_t)ldu32(u + p + 4ull * idx);
    default: return 0;
    }
}

template <bool WRITE>
__global__ void __launch_bounds__(256) bam_tag_cells(BamStream st, TagCellArgs a) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= a.nrows) return;
    const TagColDev cd = a.cols[blockIdx.y];
    if (WRITE && cd.sa_cnt < 0 && cd.sa_bytes < 0) return;
    const uint8_t *u = st.u;
    const uint64_t o = a.rec_off[row];
    const uint32_t d = a.dir[(size_t)cd.slot * a.stride + row];
    bool valid = d != 0 && d != TAG_SEEN_NULL;
    uint32_t cnt = 0, nbytes = 0;
    if (valid) {
        const uint64_t p = o + d; const uint8_t ty = u[p];
        if (cd.kind == 'i') { if (!WRITE) cd.fixed[row] = tag_int_val(u, ty, p + 1, 0); }
        else if (cd.kind == 'A') {                              // bam_aux2A, assigned through the NUL-terminated API
            const uint8_t ch = ty == 'A' ? u[p + 1] : 0;
            nbytes = ch ? 1 : 0;
            if (WRITE && ch) cd.bytes[a.offs[(size_t)cd.sa_bytes * a.ostride + row]] = ch;
        } else if (cd.kind == 'Z') {
            if (ty == 'Z' || ty == 'H') {
                uint32_t l = 0; while (u[p + 1 + l]) l++;
                nbytes = l;
                if (WRITE) { const uint32_t b0 = a.offs[(size_t)cd.sa_bytes * a.ostride + row]; for (uint32_t k = 0; k < l; k++) cd.bytes[b0 + k] = u[p + 1 + k]; }
            } else valid = false;
        } else {                                                // 'B': list of BIGINT (double bit patterns when the stored subtype is f/d)
            const uint32_t len = ty == 'B' ? ldu32(u + p + 2) : 0;
            cnt = len;
            if (WRITE && len) {
                const uint32_t c0 = a.offs[(size_t)cd.sa_cnt * a.ostride + row]; const uint8_t sub = u[p + 1];
                for (uint32_t i = 0; i < len; i++) {
                    uint64_t w;
                    if (sub == 'f') w = f32_bits_to_f64_bits(ldu32(u + p + 6 + 4ull * i));
                    else if (sub == 'd') w = 0;
                    else w = (uint64_t)tag_int_val(u, sub, p + 6, i);
                    cd.child[c0 + i] = w;
                }
            }
        }
    }
    if (!WRITE) {
        cd.valid[row] = valid ? 1 : 0;
        if (!valid && cd.kind == 'i') cd.fixed[row] = 0;
        if (cd.sa_cnt >= 0) a.lens[(size_t)cd.sa_cnt * a.ostride + row] = valid ? cnt : 0;
        if (cd.sa_bytes >= 0) a.lens[(size_t)cd.sa_bytes * a.ostride + row] = valid ? nbytes : 0;
    }
}

// ---- AUXILIARY_TAGS (src/bam_reader.c:967-1027 over bam_aux_first / bam_aux_next sam.c:4811-4832) ------------------------------
// One lane per row lists every aux tag that is not excluded (the standard-tag set when standard_tags is on), in record
// order, as TYPED entries: key (two raw bytes), kind, B subtype, and a payload (int64 / f64 bits / string bytes / widened array
// elements).  The value TEXT (`%lld`, `%g`, `subtype,v,v...`: bam_aux_to_string bam_reader.c:140-183) is rendered from these
// entries on the host: `%g` is floating-point formatting (SURVEY 8(f) item 2 keeps it host-side).  The walk stops where
// bam_aux_next would (corrupt value): that last tag is still listed, with kind AUXK_CORRUPT and no payload (the reference
// renders memory beyond the value there: undefined, empty here).
enum { AUXK_INT = 0, AUXK_FLT, AUXK_STR, AUXK_CHR, AUXK_BINT, AUXK_BFLT, AUXK_CORRUPT };
struct AuxMapDev {
    const uint16_t *excl; int32_t n_excl;
    uint8_t *valid; uint32_t *lens_ent, *lens_pay; const uint32_t *off_ent, *off_pay;
    uint16_t *key; uint8_t *kind, *sub; uint32_t *pay_off; uint8_t *payload;
};

__device__ __forceinline__ void put_u64(uint8_t *p, uint64_t v) { __builtin_memcpy(p, &v, 8); }

template <bool WRITE>
__global__ void __launch_bounds__(256) bam_aux_list(BamStream st, const uint32_t *rec_off, int64_t nrows, AuxMapDev a) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= nrows) return;
    GSrc s; s.g = st.u; const uint8_t *u = st.u;
    const uint64_t o = rec_off[row];
    uint32_t ne = 0, np = 0;
    uint32_t e0 = 0, p0 = 0;
    if (WRITE) { e0 = a.off_ent[row]; p0 = a.off_pay[row]; }
    RecInfo r;
    if (rec_check_t(st, s, o, r, true) == REC_OK) {
        const uint64_t end = o + 4ull + r.block_len;
        const uint64_t aux = o + 36 + r.l_qname + 4ull * r.n_cigar + (((uint64_t)r.l_seq + 1) >> 1) + (uint64_t)r.l_seq;
        const uint64_t sb = r.cg_beg, se = r.cg_end;
        if ((end - aux) - (se - sb) > 2) {
            uint64_t p = aux; if (p == sb) p = se; p += 2;
            for (;;) {
                const uint16_t code = (uint16_t)(s.u8(p - 2) | (s.u8(p - 1) << 8));
                bool skip = false;
                for (int k = 0; k < a.n_excl; k++) if (a.excl[k] == code) { skip = true; break; }
                const uint64_t e = aux_skip_t(s, p, end);
                if (!skip) {
                    const uint8_t ty = u[p];
                    uint32_t kind = AUXK_CORRUPT, plen = 0; uint8_t sub = 0;
                    if (e != NONE64) {
                        if (ty == 'A') { kind = AUXK_CHR; plen = 1; }
                        else if (ty == 'c' || ty == 'C' || ty == 's' || ty == 'S' || ty == 'i' || ty == 'I') { kind = AUXK_INT; plen = 8; }
                        else if (ty == 'f' || ty == 'd') { kind = AUXK_FLT; plen = 8; }
                        else if (ty == 'Z' || ty == 'H') { kind = AUXK_STR; uint32_t l = 0; while (p + 1 + l < e && u[p + 1 + l]) l++; plen = l; }
                        else { sub = u[p + 1]; kind = (sub == 'f' || sub == 'd') ? AUXK_BFLT : AUXK_BINT; plen = 8u * ldu32(u + p + 2); }
                    }
                    if (WRITE) {
                        const uint32_t ei = e0 + ne; uint8_t *dp = a.payload + p0 + np;
                        a.key[ei] = code; a.kind[ei] = (uint8_t)kind; a.sub[ei] = sub; a.pay_off[ei] = p0 + np;
                        if (kind == AUXK_CHR) dp[0] = u[p + 1];
                        else if (kind == AUXK_INT) put_u64(dp, (uint64_t)tag_int_val(u, ty, p + 1, 0));
                        else if (kind == AUXK_FLT) { uint64_t b; if (ty == 'd') __builtin_memcpy(&b, u + p + 1, 8); else b = f32_bits_to_f64_bits(ldu32(u + p + 1)); put_u64(dp, b); }
                        else if (kind == AUXK_STR) { for (uint32_t k = 0; k < plen; k++) dp[k] = u[p + 1 + k]; }
                        else if (kind == AUXK_BINT) { const uint32_t n = plen >> 3; for (uint32_t i = 0; i < n; i++) put_u64(dp + 8ull * i, (uint64_t)tag_int_val(u, sub, p + 6, i)); }
                        else if (kind == AUXK_BFLT) { const uint32_t n = plen >> 3; for (uint32_t i = 0; i < n; i++) put_u64(dp + 8ull * i, sub == 'f' ? f32_bits_to_f64_bits(ldu32(u + p + 6 + 4ull * i)) : 0ull); }
                    }
                    ne++; np += plen;
                }
                if (e == NONE64) break;
                uint64_t nx = e; if (nx == sb) nx = se;
                if (end - nx <= 2) break;
                p = nx + 2;
            }
        }
    }
    if (!WRITE) { a.lens_ent[row] = ne; a.lens_pay[row] = np; a.valid[row] = ne ? 1 : 0; }
    else if (row == nrows - 1) a.pay_off[a.off_ent[nrows]] = a.off_pay[nrows];
}
