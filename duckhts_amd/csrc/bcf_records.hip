// bcf_records.hip -- BCF record boundary discovery, validation and typed INFO/FORMAT column unpack for MI355X (gfx950).
//
// Replaces, for the read_bcf scan (src/bcf_reader.c:1155-2049):
//   htslib vcf.c:1874-1911 bcf_read1_core (framing), 2040-2212 bcf_record_check (+ updatephasing 1985-2029),
//   4234-4302 bcf_unpack, 3036-3079 bcf_fmt_array (ID / allele text), 6056-6138 bcf_get_info_values,
//   6140-6177 bcf_get_format_string, 6179-6248 bcf_get_format_values, and the column writers
//   src/bcf_reader.c:1381-1462 (core), 1542-1733 (INFO), 1738-1982 (FORMAT wide/tidy, GT text), 996-1061 (comma lists).
//
// Stages (all integer/byte work, HBM-bound):
//   1. bcf_tile_scan / bcf_tile_fix: the record chain (next = cur + 8 + l_shared + l_indiv) is cut into 8 KiB tiles; one
//      wave per tile speculates the first record inside its tile, walks to the tile end, and continuity is proven
//      tile-to-tile exactly as for BAM (bam_records.hip) -- a wrong guess only costs a repair round.
//   2. bcf_rec_check: one lane per record runs every bcf_record_check predicate and leaves a small directory
//      (offsets of the allele / FILTER vectors and of each schema INFO / FORMAT field inside the record).
//   3. bcf_cells<false>: one lane per (row, projected column) evaluates the cell through the directory: validity,
//      fixed-width payloads, and the child / byte counts of variable-width cells.
//   4. matrix prefix sums over those counts, then bcf_cells<true> writes list children and string bytes in place.
#pragma once
#include "dhts_common.h"

struct BcfStream {
    const uint8_t *u; uint64_t ulen;
    int32_t n_ctg, n_ids, n_smp, final_batch;
    int32_t n_info_f, n_fmt_f;
    const uint8_t *ctg_ok;        // n_ctg: contig id present in the header dictionary
    const uint8_t *id_ok;         // n_ids: dictionary id present
    const int16_t *info_slot;     // n_ids: schema INFO field index of a dictionary id, -1 if none
    const int16_t *fmt_slot;      // n_ids: schema FORMAT field index, -1 if none
    const int32_t *pos_hi;        // VCF text: bits 32.. of every record's 0-based position (htslib keeps 64-bit positions for text, vcf.c:4052-4063; the BCF2 core in the record holds the low word); nullptr for BCF
};
// 0-based position of record `rec` at u + o (BCF: the core's word, 0xffffffff = -1, vcf.c:1895-1896)
__device__ __forceinline__ int64_t bcf_pos64(const BcfStream &st, const uint8_t *u, uint64_t o, int64_t rec) {
    uint32_t lo; __builtin_memcpy(&lo, u + o + 12, 4);
    if (st.pos_hi) return (int64_t)(((uint64_t)(uint32_t)st.pos_hi[rec] << 32) | lo);
    return lo == 0xffffffffu ? -1 : (int64_t)lo;
}

__device__ __constant__ uint8_t BCF_SHIFT[16] = {0, 0, 1, 2, 3, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};     // vcf.c:91

__device__ __forceinline__ uint32_t b_u16(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
__device__ __forceinline__ int32_t b_i16(const uint8_t *p) { return (int32_t)(int16_t)b_u16(p); }

// one hop of the chain: the tests bcf_read1_core / bcf_record_check make on the 32-byte core alone
template <class S> __device__ __forceinline__ int bcf_hop(const BcfStream &st, const S &s, uint64_t o, uint64_t &sz, bool spec) {
    if (st.ulen - o < 32) return REC_INCOMPLETE;
    const uint32_t l_shared = s.u32(o), l_indiv = s.u32(o + 4);
    if (l_shared < 24) return REC_INVALID;                                   // vcf.c:1885
    const int32_t rid = (int32_t)s.u32(o + 8);
    if (rid < 0 || rid >= st.n_ctg) return REC_INVALID;                      // vcf.c:2069-2073 (dictionary holes are tested in bcf_rec_check)
    if ((s.u32(o + 24) >> 16) < 1) return REC_INVALID;                       // n_allele >= 1, vcf.c:2088-2092
    if (spec) {                                                              // speculation filter only: never decides validity
        if (l_shared > (1u << 28) || l_indiv > (1u << 30)) return REC_INVALID;
        const uint32_t x = s.u32(o + 28);
        if ((x >> 24) != 0 && (x & 0xffffff) != 0 && (x & 0xffffff) != (uint32_t)st.n_smp) return REC_INVALID;
    }
    sz = 8ull + l_shared + l_indiv;
    if (st.ulen - o < sz) return REC_INCOMPLETE;
    return REC_OK;
}

__device__ void bcf_tile_walk(const BcfStream &st, uint64_t start, uint64_t tile_end, uint64_t &end_next, uint32_t &count, int &err) {
    GSrc gs; gs.g = st.u;
    uint64_t o = start; uint32_t c = 0; err = 0;
    while (o < tile_end) {
        uint64_t sz = 0;
        const int rc = bcf_hop(st, gs, o, sz, false);
        if (rc == REC_INCOMPLETE) { if (st.final_batch && o < st.ulen) err = 1; break; }   // short read = error (vcf.c:1879-1882, 1908-1909)
        if (rc == REC_INVALID) { err = 1; break; }
        c++; o += sz;
    }
    end_next = o; count = c;
}

extern "C" __global__ void __launch_bounds__(64)
bcf_tile_scan(BcfStream st, uint64_t start0, int64_t ntiles, TileOut out, uint64_t spec_from) {
    __shared__ __attribute__((aligned(16))) uint8_t buf[TL_TILE + TL_HALO];
    const int lane = threadIdx.x;
    const int64_t t = blockIdx.x;
    if (t >= ntiles) return;
    const uint64_t tb = (uint64_t)t * TL_TILE;
    uint64_t te = tb + TL_TILE; if (te > st.ulen) te = st.ulen;
    BamStream bs; bs.u = st.u; bs.ulen = st.ulen; bs.n_ref = 0; bs.final_batch = st.final_batch;
    LSrc s; tile_stage(bs, tb, buf, s, lane);
    PSrc ls; ls.l = buf; ls.base = tb;
    GSrc gs; gs.g = st.u;
    uint64_t first = NONE64;
    if (t == 0 && start0 != NONE64) first = start0;
    else {
        const uint64_t lim = (t == 0) ? st.ulen : te;
        const uint64_t from = (t == 0) ? spec_from : 0;        // (a retry after a false start resumes behind the failed candidate)
        for (uint64_t base = (t == 0) ? (from & ~63ull) : tb; base < lim; base += 64) {
            const uint64_t o = base + (uint64_t)lane;
            bool ok = false;
            if (o < lim && o >= from) {
                uint64_t sz = 0;
                const bool inw = (o - tb) + 32u <= (uint64_t)s.len;
                const int rc0 = inw ? bcf_hop(st, ls, o, sz, true) : bcf_hop(st, gs, o, sz, true);
                if (rc0 == REC_OK) {
                    ok = true;
                    uint64_t o2 = o + sz;
                    for (int k = 0; k < 2 && ok; k++) {
                        uint64_t s2 = 0;
                        const bool inw2 = (o2 >= tb) && (o2 - tb) + 32u <= (uint64_t)s.len;
                        const int rc = inw2 ? bcf_hop(st, ls, o2, s2, true) : bcf_hop(st, gs, o2, s2, true);
                        if (rc == REC_INVALID) ok = false;
                        else if (rc == REC_INCOMPLETE) break;
                        else o2 += s2;
                    }
                }
            }
            const uint64_t m = __ballot(ok);
            if (m) { first = base + (uint64_t)(__ffsll((unsigned long long)m) - 1); break; }
        }
    }
    uint64_t en = NONE64; uint32_t cnt = 0; int err = 0;
    if (first != NONE64 && first < te) {
        uint64_t o = first;
        while (o < te) {
            uint64_t sz = 0;
            const int rc = ((o - tb) + 32 <= (uint64_t)s.len) ? bcf_hop(st, ls, o, sz, false) : bcf_hop(st, gs, o, sz, false);
            if (rc == REC_INCOMPLETE) { if (st.final_batch && o < st.ulen) err = 1; break; }
            if (rc == REC_INVALID) { err = 1; break; }
            cnt++; o += sz;
        }
        en = o;
    } else if (first != NONE64) en = first;
    if (lane == 0) {
        out.first[t] = (first != NONE64 && first < te) ? first : NONE64;
        out.end_next[t] = en; out.count[t] = cnt; out.err[t] = err;
    }
}

// out-of-place continuity proof + repair round (same protocol as bam_tile_fix)
extern "C" __global__ void __launch_bounds__(256)
bcf_tile_fix(BcfStream st, uint32_t tile_bytes, int64_t ntiles, TileOut in, TileOut out, uint32_t *nfixed) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntiles) return;
    uint64_t f = in.first[t], en = in.end_next[t]; uint32_t cnt = in.count[t]; int32_t err = in.err[t];
    if (t >= 1) {
        const uint64_t tb = (uint64_t)t * tile_bytes; uint64_t te = tb + tile_bytes; if (te > st.ulen) te = st.ulen;
        const uint64_t p = in.end_next[t - 1];
        bool pred_ok = (p != NONE64) && !in.err[t - 1];
        if (pred_ok && t >= 2) pred_ok = tile_consistent(in, t - 1, tb);
        if (pred_ok && !tile_consistent(in, t, te)) {
            if (p >= te) { f = NONE64; cnt = 0; err = 0; en = p; }
            else { int e; bcf_tile_walk(st, p, te, en, cnt, e); f = p; err = e; }
            atomicAdd(nfixed, 1u);
        }
    }
    out.first[t] = f; out.end_next[t] = en; out.count[t] = cnt; out.err[t] = err;
}

extern "C" __global__ void bcf_tile_fix_seq(BcfStream st, uint32_t tile_bytes, int64_t ntiles, TileOut out) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    for (int64_t t = 1; t < ntiles; t++) {
        if (out.err[t - 1]) return;
        uint64_t p = out.end_next[t - 1];
        uint64_t tb = (uint64_t)t * tile_bytes, te = tb + tile_bytes; if (te > st.ulen) te = st.ulen;
        if (p >= te) { out.first[t] = NONE64; out.count[t] = 0; out.err[t] = 0; out.end_next[t] = p; continue; }
        if (out.first[t] == p) continue;
        uint64_t en; uint32_t cnt; int err;
        bcf_tile_walk(st, p, te, en, cnt, err);
        out.first[t] = p; out.count[t] = cnt; out.err[t] = err; out.end_next[t] = en;
    }
}

// record offsets by record id (one lane per tile; <= 256 records per 8 KiB tile)
extern "C" __global__ void __launch_bounds__(256)
bcf_tile_offsets(BcfStream st, uint32_t tile_bytes, int64_t ntiles, TileOut out, const uint32_t *rowbase, const uint64_t *res, int64_t nrec, uint32_t *rec_off) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntiles || (uint64_t)t > res[3]) return;
    uint64_t o = out.first[t]; if (o == NONE64) return;
    uint32_t n = out.count[t], row = rowbase[t];
    for (uint32_t k = 0; k < n && (int64_t)(row + k) < nrec; k++) { rec_off[row + k] = (uint32_t)o; o += 8ull + ldu32(st.u + o) + ldu32(st.u + o + 4); }
}

// ---- typed values (BCF2 typed-value codec, htslib/vcf.h:1586-1690) ------------------------------------------------------
__device__ __forceinline__ bool bcf_dec_int1_safe(const uint8_t *u, uint64_t &p, uint64_t end, int32_t &val) {   // vcf.c:1918-1949
    if (end - p < 2) return false;
    const int t = u[p] & 0xf; uint64_t q = p + 1;
    if (t == 1) { val = (int8_t)u[q]; q += 1; }
    else {
        if (end - q < (1ull << BCF_SHIFT[t])) return false;
        if (t == 2) { val = b_i16(u + q); q += 2; }
        else if (t == 3) { val = (int32_t)ldu32(u + q); q += 4; }
        else return false;
    }
    p = q; return true;
}
__device__ __forceinline__ bool bcf_dec_size_safe(const uint8_t *u, uint64_t &p, uint64_t end, int &num, int &type) {   // vcf.c:1951-1963
    if (p >= end) return false;
    type = u[p] & 0xf;
    if ((u[p] >> 4) != 15) { num = u[p] >> 4; p += 1; return true; }
    uint64_t q = p + 1; int32_t v;
    if (!bcf_dec_int1_safe(u, q, end, v)) return false;
    if (v < 0) return false;
    num = v; p = q; return true;
}
// unchecked variant for vectors bcf_rec_check has already validated
__device__ __forceinline__ void bcf_dec_size(const uint8_t *u, uint64_t &p, int &num, int &type) {
    type = u[p] & 0xf;
    if ((u[p] >> 4) != 15) { num = u[p] >> 4; p += 1; return; }
    const int t = u[p + 1] & 0xf;
    if (t == 1) { num = (int8_t)u[p + 2]; p += 3; }
    else if (t == 2) { num = b_i16(u + p + 2); p += 4; }
    else { num = (int32_t)ldu32(u + p + 2); p += 6; }
}

// One lane per record: bcf_record_check (vcf.c:2040-2212) + the per-record directory.
// dir is slot-major: dir[slot * stride + rec]; slot 0 = first allele descriptor, 1 = FILTER descriptor, then one slot per
// schema INFO field and per schema FORMAT field (offset of the value descriptor relative to the record start, 0 = absent;
// the first occurrence of a key wins, vcf.c:6064-6066 / 6192-6194).
extern "C" __global__ void __launch_bounds__(256)
bcf_rec_check(BcfStream st, const uint32_t *rec_off, int64_t nrec, uint32_t *dir, uint32_t stride, unsigned long long *bad_rec) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrec) return;
    const uint8_t *u = st.u;
    const uint64_t o = rec_off[r];
    const int D = 2 + st.n_info_f + st.n_fmt_f;
    for (int k = 0; k < D; k++) dir[(size_t)k * stride + r] = 0;
    const uint32_t l_shared = ldu32(u + o) - 24, l_indiv = ldu32(u + o + 4);
    const int32_t rid = (int32_t)ldu32(u + o + 8);
    const uint32_t n_info = b_u16(u + o + 24), n_allele = b_u16(u + o + 26);
    const uint32_t n_sample = ldu32(u + o + 28) & 0xffffff; uint32_t n_fmt = u[o + 31];
    if ((!l_indiv || !n_sample) && n_fmt) n_fmt = 0;                          // vcf.c:1906
    const uint64_t she = o + 32 + l_shared, ine = she + l_indiv;
    bool err = false, bad = false;
    if (rid < 0 || rid >= st.n_ctg || !st.ctg_ok[rid]) err = true;
    uint64_t p = o + 32; int num = 0, type = 0;
    do {
        if (!bcf_dec_size_safe(u, p, she, num, type)) { bad = true; break; }            // ID
        if (type != 7) err = true;
        uint64_t bytes = (uint64_t)num << BCF_SHIFT[type];
        if (she - p < bytes) { bad = true; break; }
        p += bytes;
        if (n_allele < 1) err = true;
        for (uint32_t i = 0; i < n_allele; i++) {
            if (i == 0) dir[r] = (uint32_t)(p - o);
            if (!bcf_dec_size_safe(u, p, she, num, type)) { bad = true; break; }
            if (type != 7) err = true;
            bytes = (uint64_t)num << BCF_SHIFT[type];
            if (she - p < bytes) { bad = true; break; }
            p += bytes;
        }
        if (bad) break;
        dir[(size_t)stride + r] = (uint32_t)(p - o);
        if (!bcf_dec_size_safe(u, p, she, num, type)) { bad = true; break; }            // FILTER
        if (num > 0) {
            bytes = (uint64_t)num << BCF_SHIFT[type];
            if (she - p < bytes) { bad = true; break; }
            if (!(type == 1 || type == 2 || type == 3)) { err = true; p += bytes; }
            else for (int i = 0; i < num; i++) {
                int32_t key = type == 1 ? (int32_t)(int8_t)u[p] : type == 2 ? b_i16(u + p) : (int32_t)ldu32(u + p);
                p += 1ull << BCF_SHIFT[type];
                if (key < 0 || key >= st.n_ids || !st.id_ok[key]) err = true;
            }
        }
        for (uint32_t i = 0; i < n_info; i++) {
            int32_t key = -1;
            if (!bcf_dec_int1_safe(u, p, she, key)) { bad = true; break; }
            const bool kok = !(key < 0 || key >= st.n_ids || !st.id_ok[key]);
            if (!kok) err = true;
            const uint32_t d = (uint32_t)(p - o);
            if (!bcf_dec_size_safe(u, p, she, num, type)) { bad = true; break; }
            if (!(type == 0 || type == 1 || type == 2 || type == 3 || type == 5 || type == 7) || (type == 0 && num > 0)) err = true;
            bytes = (uint64_t)num << BCF_SHIFT[type];
            if (she - p < bytes) { bad = true; break; }
            if (kok) { const int sl = st.info_slot[key]; if (sl >= 0 && dir[(size_t)(2 + sl) * stride + r] == 0) dir[(size_t)(2 + sl) * stride + r] = d; }
            p += bytes;
        }
        if (bad) break;
        p = she;
        for (uint32_t i = 0; i < n_fmt; i++) {
            int32_t key = -1;
            if (!bcf_dec_int1_safe(u, p, ine, key)) { bad = true; break; }
            const bool kok = !(key < 0 || key >= st.n_ids || !st.id_ok[key]);
            if (!kok) err = true;
            const uint32_t d = (uint32_t)(p - o);
            if (!bcf_dec_size_safe(u, p, ine, num, type)) { bad = true; break; }
            if (!(type == 0 || type == 1 || type == 2 || type == 3 || type == 5 || type == 7) || (type == 0 && num > 0)) err = true;
            bytes = ((uint64_t)num << BCF_SHIFT[type]) * n_sample;               // also updatephasing's own bounds test (vcf.c:1989-1991)
            if (ine - p < bytes) { bad = true; break; }
            if (kok) { const int sl = st.fmt_slot[key]; if (sl >= 0) { const size_t q = (size_t)(2 + st.n_info_f + sl) * stride + r; if (dir[q] == 0) dir[q] = d; } }
            p += bytes;
        }
    } while (0);
    if (bad || err) atomicMin(bad_rec, (unsigned long long)r);
}

// ---- cells -------------------------------------------------------------------------------------------------------------------
enum { BK_CHROM = 0, BK_POS, BK_ID, BK_REF, BK_ALT, BK_QUAL, BK_FILTER, BK_INFO, BK_SAMPLE_ID, BK_FORMAT, BK_VEP };
enum { BF_NULL_ALWAYS = 1, BF_GT = 2, BF_GT_FIX = 4 };

struct BcfColDev {
    int32_t kind, htype, is_list, slot, sample, flags;
    int32_t sa_cnt, sa_bytes;     // rows of the count matrix (children / bytes) or -1
    uint8_t *valid;               // nrows bytes
    void *fixed;                  // nrows x native width (scalar fixed-width columns, dictionary-coded columns)
    uint8_t *bytes;               // VARCHAR bytes (scalar) or child bytes (list of VARCHAR)
    uint32_t *child_off;          // list of VARCHAR: child_n + 1 byte offsets
    uint32_t *child_fixed;        // list of 4-byte words (INTEGER, FLOAT bits, dictionary ids)
    uint8_t *child_valid;         // BK_VEP: one byte per child, 0 = NULL element (a missing field of a transcript)
    int32_t vep_field, pad_;      // BK_VEP: index of the field inside a transcript
};

struct BcfCellArgs {
    const uint32_t *rec_off; const uint32_t *dir; uint32_t stride;
    int64_t nrows; int32_t tidy, n_smp;
    uint32_t *lens; const uint32_t *offs; uint32_t ostride;
    const BcfColDev *cols;
    const uint32_t *sel;          // region filter: compacted list of kept record ids (nullptr = every record)
    uint32_t ncols;
    uint32_t n_vep; const uint32_t *vep_cols;   // n_vep > 0: the BK_VEP columns (indices into cols) are evaluated by bcf_vep_wave, a wave per row and column; bcf_cells leaves them alone
};

__device__ __forceinline__ uint32_t cstr_len(const uint8_t *p, uint32_t n) { uint32_t l = 0; while (l < n && p[l]) l++; return l; }

// widen one stored element (getter BRANCH macros, vcf.c:6096-6131 / 6221-6244)
__device__ __forceinline__ void bcf_elem(const uint8_t *p, int type, int j, uint32_t &w, bool &is_end, bool &is_miss) {
    if (type == 1) { const int32_t v = (int8_t)p[j]; is_end = v == -127; is_miss = v == -128; w = (uint32_t)v; }
    else if (type == 2) { const int32_t v = b_i16(p + 2 * j); is_end = v == -32767; is_miss = v == -32768; w = (uint32_t)v; }
    else if (type == 3) { const uint32_t v = ldu32(p + 4 * j); is_end = v == 0x80000001u; is_miss = v == 0x80000000u; w = v; }
    else { const uint32_t v = ldu32(p + 4 * j); is_end = v == 0x7F800002u; is_miss = v == 0x7F800001u; w = v; }
}
__device__ __forceinline__ bool bcf_numeric_type(int t) { return t == 1 || t == 2 || t == 3 || t == 5; }

__device__ __forceinline__ uint64_t f32_bits_to_f64_bits(uint32_t b) {      // exact widening, NaN payload kept and quieted like cvtss2sd
    const uint64_t sign = (uint64_t)(b >> 31) << 63; const uint32_t e = (b >> 23) & 0xff; uint32_t m = b & 0x7fffff;
    if (e == 0xff) return sign | (0x7ffull << 52) | ((uint64_t)m << 29) | (m ? (1ull << 51) : 0);
    if (e == 0) {
        if (!m) return sign;
        const int sh = __clz(m) - 8;                                          // normalise the denormal
        m = (m << sh) & 0x7fffff;
        return sign | ((uint64_t)(1 - sh - 127 + 1023) << 52) | ((uint64_t)m << 29);
    }
    return sign | ((uint64_t)(e - 127 + 1023) << 52) | ((uint64_t)m << 29);
}

__device__ __forceinline__ uint32_t ndigits10(uint32_t a) { return ndigits(a) + (a >= 1000000000u); }
__device__ __forceinline__ uint32_t dec_len_i32(int32_t v) {
    uint32_t a = v < 0 ? (uint32_t)(-(int64_t)v) : (uint32_t)v;
    return ndigits10(a) + (v < 0);
}

template <bool WRITE>
__global__ void __launch_bounds__(256) bcf_cells(BcfStream st, BcfCellArgs a) {
    // 1-D grid, column index fastest: the blocks that evaluate the same rows for different columns are dispatched back to back,
    // so a record's cache lines are fetched from HBM once and served from L2 for the other columns
    const uint32_t colb = blockIdx.x % a.ncols;
    const int64_t row = (int64_t)(blockIdx.x / a.ncols) * blockDim.x + threadIdx.x;
    if (row >= a.nrows) return;
    const BcfColDev cd = a.cols[colb];
    if (WRITE && cd.sa_cnt < 0 && cd.sa_bytes < 0) return;
    if (cd.kind == BK_VEP && a.n_vep) return;
    const uint8_t *u = st.u;
    int64_t rec = a.tidy ? row / a.n_smp : row;
    if (a.sel) rec = a.sel[rec];
    const int smp = cd.sample >= 0 ? cd.sample : (a.tidy ? (int)(row % a.n_smp) : 0);
    const uint64_t o = a.rec_off[rec];
    uint32_t cnt = 0, nbytes = 0; bool valid = true;
    uint32_t cbase = 0, bbase = 0;
    if (WRITE) {
        if (cd.sa_cnt >= 0) cbase = a.offs[(size_t)cd.sa_cnt * a.ostride + row];
        if (cd.sa_bytes >= 0) bbase = a.offs[(size_t)cd.sa_bytes * a.ostride + row];
    }
    auto put_str = [&](const uint8_t *p, uint32_t l) {                        // scalar VARCHAR payload
        if (WRITE) for (uint32_t k = 0; k < l; k++) cd.bytes[bbase + k] = p[k];
        nbytes = l;
    };
    auto put_child_str = [&](const uint8_t *p, uint32_t l) {
        if (WRITE) { cd.child_off[cbase + cnt] = bbase + nbytes; for (uint32_t k = 0; k < l; k++) cd.bytes[bbase + nbytes + k] = p[k]; }
        cnt++; nbytes += l;
    };
    auto put_child_w = [&](uint32_t w) { if (WRITE) cd.child_fixed[cbase + cnt] = w; cnt++; };
    auto c_isspace = [](uint8_t ch) { return ch == ' ' || (ch >= 9 && ch <= 13); };
    const uint8_t dot = '.';

    switch (cd.kind) {
    case BK_CHROM: if (!WRITE) ((int32_t *)cd.fixed)[row] = (int32_t)ldu32(u + o + 8); break;
    case BK_POS: if (!WRITE) ((int64_t *)cd.fixed)[row] = bcf_pos64(st, u, o, rec) + 1; break;   // vcf.c:1895-1896
    case BK_QUAL: if (!WRITE) {
            const uint32_t b = ldu32(u + o + 20);
            if (b == 0x7F800001u) { valid = false; ((uint64_t *)cd.fixed)[row] = 0; }           // NULL, payload 0.0 (bcf_reader.c:1427-1434)
            else ((uint64_t *)cd.fixed)[row] = f32_bits_to_f64_bits(b);
        } break;
    case BK_SAMPLE_ID: if (!WRITE) ((int32_t *)cd.fixed)[row] = smp; break;
    case BK_ID: {
            uint64_t p = o + 32; int n, t; bcf_dec_size(u, p, n, t);
            const uint32_t l = n > 0 ? cstr_len(u + p, (uint32_t)n) : 0;
            if (n <= 0 || (l == 1 && u[p] == '.')) valid = false;               // "." (bcf_reader.c:1393-1401)
            else put_str(u + p, l);
        } break;
    case BK_REF: {
            uint64_t p = o + a.dir[rec]; int n, t; bcf_dec_size(u, p, n, t);
            if (n > 0) put_str(u + p, cstr_len(u + p, (uint32_t)n)); else put_str(&dot, 1);     // vcf.c:3040-3042
        } break;
    case BK_ALT: {
            const uint32_t n_allele = b_u16(u + o + 26);
            uint64_t p = o + a.dir[rec];
            for (uint32_t i = 0; i < n_allele; i++) {
                int n, t; bcf_dec_size(u, p, n, t);
                if (i > 0) { if (n > 0) put_child_str(u + p, cstr_len(u + p, (uint32_t)n)); else put_child_str(&dot, 1); }
                p += (uint64_t)n;
            }
        } break;
    case BK_FILTER: {
            uint64_t p = o + a.dir[(size_t)a.stride + rec]; int n, t; bcf_dec_size(u, p, n, t);
            if (n <= 0) put_child_w(0xffffffffu);                              // no filters => literal "PASS" (bcf_reader.c:1443-1447)
            else for (int i = 0; i < n; i++) put_child_w((uint32_t)(t == 1 ? (int32_t)(int8_t)u[p + i] : t == 2 ? b_i16(u + p + 2 * i) : (int32_t)ldu32(u + p + 4 * i)));
        } break;
    case BK_VEP: {
            // One field of every transcript of the annotation tag (CSQ / BCSQ / ANN / ...).  vep_record_parse_bcf (src/vep_parser.c:317-326)
            // takes the tag through bcf_get_info_string -- nothing unless the tag is declared String, present and non-empty (vcf.c:6056-6081);
            // vep_record_parse (:286-315) walks the non-empty ','-separated pieces (strtok_r), parse_single_transcript (:243-284) the
            // '|'-separated fields of a piece: white space trimmed, "" and "." missing; a record without any transcript is a NULL row
            // and so, in tidy mode, is every sample row of a record but the first (src/bcf_reader.c:1370-1373, 1463-1541).
            const uint32_t d = a.dir[(size_t)(2 + cd.slot) * a.stride + rec];
            if ((cd.flags & BF_NULL_ALWAYS) || (a.tidy && row % a.n_smp != 0) || !d) { valid = false; break; }
            uint64_t p = o + d; int n, t; bcf_dec_size(u, p, n, t);
            if (n <= 0) { valid = false; break; }
            const uint32_t l = cstr_len(u + p, (uint32_t)n);
            const uint8_t *s = u + p;
            uint32_t i = 0;
            while (i < l) {
                if (s[i] == ',') { i++; continue; }
                uint32_t e = i; while (e < l && s[e] != ',') e++;              // transcript = s[i, e)
                uint32_t f0 = i; bool have = true;
                for (int k = 0; k < cd.vep_field; k++) {                       // skip to the field; fewer fields than asked for = missing
                    while (f0 < e && s[f0] != '|') f0++;
                    if (f0 >= e) { have = false; break; }
                    f0++;
                }
                uint32_t f1 = f0;
                if (have) { while (f1 < e && s[f1] != '|') f1++; while (f0 < f1 && c_isspace(s[f0])) f0++; while (f1 > f0 + 1 && c_isspace(s[f1 - 1])) f1--; }
                const uint32_t tl = have ? f1 - f0 : 0;
                const bool miss = tl == 0 || (tl == 1 && s[f0] == '.');
                if (WRITE) cd.child_valid[cbase + cnt] = miss ? 0 : 1;
                if (cd.htype == 1) {                                           // Integer: (int32_t)strtol(token, &end, 10), INT32_MIN unless the whole token is a number (vep_parse_int :207-220)
                    uint32_t w = 0;
                    if (!miss) {
                        uint32_t q = f0; const bool neg = s[q] == '-'; if (s[q] == '-' || s[q] == '+') q++;
                        uint64_t acc = 0; bool sat = false, ok = q < f1;
                        for (; q < f1; q++) {
                            const uint32_t dg = (uint32_t)s[q] - '0'; if (dg > 9) { ok = false; break; }
                            if (acc > (0x7fffffffffffffffull - dg) / 10) sat = true; else acc = acc * 10 + dg;
                        }
                        if (!ok) w = 0x80000000u;
                        else if (sat || (neg ? acc > 0x8000000000000000ull : acc > 0x7fffffffffffffffull)) w = neg ? 0u : 0xffffffffu;   // LONG_MIN / LONG_MAX, truncated
                        else w = (uint32_t)(neg ? (uint64_t)0 - acc : acc);
                    }
                    put_child_w(w);
                } else put_child_str(s + f0, miss ? 0 : tl);                   // String; Float travels as its text (DHTS_ENC_FLOAT_TEXT)
                i = e;
            }
            if (cnt == 0) valid = false;
        } break;
    case BK_INFO: {
            const uint32_t d = a.dir[(size_t)(2 + cd.slot) * a.stride + rec];
            if (cd.htype == 0) { if (!WRITE) ((uint8_t *)cd.fixed)[row] = d != 0; break; }      // Flag: present or not, never NULL
            if (!d) { valid = false; break; }
            uint64_t p = o + d; int n, t; bcf_dec_size(u, p, n, t);
            if (cd.htype == 3) {                                               // String: info->len BYTES, C-string semantics (vcf.c:6071-6081)
                if (n <= 0) { valid = false; break; }
                const uint32_t l = cstr_len(u + p, (uint32_t)n);
                if (l == 1 && u[p] == '.') { valid = false; break; }
                if (!cd.is_list) { put_str(u + p, l); break; }
                uint32_t stt = 0;                                              // comma split; the last token only if non-empty (bcf_reader.c:1018-1057)
                for (uint32_t i = 0; i < l; i++) if (u[p + i] == ',') { put_child_str(u + p + stt, i - stt); stt = i + 1; }
                if (l > stt) put_child_str(u + p + stt, l - stt);
                break;
            }
            if (!bcf_numeric_type(t)) { valid = false; break; }
            const uint32_t miss_h = cd.htype == 1 ? 0x80000000u : 0x7F800001u, vend_h = cd.htype == 1 ? 0x80000001u : 0x7F800002u;
            const uint32_t miss_s = t == 5 ? 0x7F800001u : 0x80000000u;
            int j = 0; uint32_t w0 = 0;
            for (; j < n; j++) {                                               // the getter stops at the first vector_end (vcf.c:6105)
                uint32_t w; bool ie, im; bcf_elem(u + p, t, j, w, ie, im);
                if (ie) break;
                if (im) w = miss_s;
                if (j == 0) w0 = w;
                if (cd.is_list && w != miss_h && w != vend_h) put_child_w(w);
            }
            if (j == 0) { valid = false; cnt = 0; break; }
            if (!cd.is_list) { if (w0 == miss_h) valid = false; else if (!WRITE) ((uint32_t *)cd.fixed)[row] = w0; }
        } break;
    case BK_FORMAT: {
            const uint32_t d = (cd.flags & BF_NULL_ALWAYS) ? 0 : a.dir[(size_t)(2 + st.n_info_f + cd.slot) * a.stride + rec];
            const uint32_t n_sample = ldu32(u + o + 28) & 0xffffff;
            if (!d || (uint32_t)smp >= n_sample) { valid = false; break; }
            uint64_t p = o + d; int n, t; bcf_dec_size(u, p, n, t);
            if (cd.htype == 3 && !(cd.flags & BF_GT)) {                        // plain string: n BYTES per sample at stride n (vcf.c:6166-6174)
                const uint8_t *q = u + p + (uint64_t)smp * (uint32_t)(n > 0 ? n : 0);
                put_str(q, n > 0 ? cstr_len(q, (uint32_t)n) : 0);
                break;
            }
            if (n <= 0 || !bcf_numeric_type(t)) { valid = false; break; }
            const uint8_t *q = u + p + (uint64_t)smp * ((uint64_t)n << BCF_SHIFT[t]);
            const uint32_t miss_s = t == 5 ? 0x7F800001u : 0x80000000u, vend_s = t == 5 ? 0x7F800002u : 0x80000001u;
            if (cd.flags & BF_GT) {                                            // GT text (bcf_reader.c:1904-1957)
                uint8_t tmp[12];
                for (int j = 0; j < n; j++) {
                    uint32_t w; bool ie, im; bcf_elem(q, t, j, w, ie, im);
                    if (j == 0 && (cd.flags & BF_GT_FIX)) {                    // updatephasing on pre-4.4 files (vcf.c:1985-2029): low byte of the first allele
                        const int inc = 1 << BCF_SHIFT[t];
                        uint8_t lo = q[0];
                        if (n == 1) { if (lo) lo |= 1; }
                        else if (n == 2) lo |= (q[inc] & 1);
                        else { uint8_t all = 1; for (int k = 1; k < n; k++) all &= q[inc * k]; lo |= all; }
                        uint8_t e4[4] = {lo, 0, 0, 0};
                        for (int k = 1; k < inc; k++) e4[k] = q[k];
                        bcf_elem(e4, t, 0, w, ie, im);
                    }
                    if (im) w = miss_s; else if (ie) break;
                    const int32_t v = (int32_t)w;
                    if (v == (int32_t)0x80000001u) break;
                    if (j > 0) { const uint8_t sep = (v & 1) ? '|' : '/'; if (WRITE) cd.bytes[bbase + nbytes] = sep; nbytes++; }
                    if ((v >> 1) == 0) { if (WRITE) cd.bytes[bbase + nbytes] = '.'; nbytes++; }
                    else {
                        const int32_t al = (v >> 1) - 1;
                        const uint32_t l = dec_len_i32(al);
                        if (WRITE) {
                            uint32_t mag = al < 0 ? (uint32_t)(-(int64_t)al) : (uint32_t)al;
                            for (uint32_t k = 0; k < l; k++) { tmp[l - 1 - k] = (uint8_t)('0' + mag % 10); mag /= 10; }
                            if (al < 0) tmp[0] = '-';
                            for (uint32_t k = 0; k < l; k++) cd.bytes[bbase + nbytes + k] = tmp[k];
                        }
                        nbytes += l;
                    }
                }
                if (nbytes == 0) valid = false;
                break;
            }
            const uint32_t miss_h = cd.htype == 1 ? 0x80000000u : 0x7F800001u, vend_h = cd.htype == 1 ? 0x80000001u : 0x7F800002u;
            bool ended = false; uint32_t w0 = 0;
            for (int j = 0; j < n; j++) {                                      // per-sample widen + vector_end padding (vcf.c:6221-6236)
                uint32_t w = vend_s;
                if (!ended) { bool ie, im; bcf_elem(q, t, j, w, ie, im); if (im) w = miss_s; else if (ie) { ended = true; w = vend_s; } }
                if (j == 0) w0 = w;
                if (cd.is_list) { if (w != miss_h && w != vend_h) put_child_w(w); }
                else break;
            }
            if (!cd.is_list) { if (w0 == miss_h) valid = false; else if (!WRITE) ((uint32_t *)cd.fixed)[row] = w0; }
        } break;
    }
    if (!WRITE) {
        cd.valid[row] = valid ? 1 : 0;
        if (!valid) {
            cnt = 0; nbytes = 0;
            if (cd.fixed && cd.kind != BK_QUAL) {
                if (cd.kind == BK_POS) ((int64_t *)cd.fixed)[row] = 0;
                else if (cd.htype == 0 && (cd.kind == BK_INFO || cd.kind == BK_FORMAT)) ((uint8_t *)cd.fixed)[row] = 0;   // BOOLEAN columns are one byte wide (a 4-byte store here ran into the next column's payload)
                else ((uint32_t *)cd.fixed)[row] = 0;
            }
        }
        if (cd.sa_cnt >= 0) a.lens[(size_t)cd.sa_cnt * a.ostride + row] = cnt;
        if (cd.sa_bytes >= 0) a.lens[(size_t)cd.sa_bytes * a.ostride + row] = nbytes;
    } else if (row == a.nrows - 1 && cd.child_off && cd.sa_cnt >= 0 && cd.sa_bytes >= 0) {
        // closing offset of the child string table
        cd.child_off[a.offs[(size_t)cd.sa_cnt * a.ostride + a.nrows]] = a.offs[(size_t)cd.sa_bytes * a.ostride + a.nrows];
    }
}

// ---- VEP_<field> columns of long annotation strings: a wave per (row, column) -----------------------------------------------------------------
// The BK_VEP case of bcf_cells walks the annotation string with one lane: a chain of dependent byte reads as long as the string (gnomAD:
// ~9 KB, forty transcripts), one row per lane and 15,000 rows per batch -- fewer workgroups than the chip has CUs.  Here the wave finds the
// ',' between transcripts 16 bytes per lane (positions collected in LDS, in order), then every lane takes one transcript: the '|' in front of
// the wanted field is found by counting them eight bytes at a time, the field is trimmed / converted as in bcf_cells, and the elements are
// placed by a scan over the wave.  Same rules (src/vep_parser.c:207-326), same outputs.
#define VEP_WSEP 2048u
__device__ __forceinline__ uint64_t vep_eq8(const uint8_t *s, uint32_t p, uint32_t end, uint64_t pat) {       // 0x80 in byte k: s[p + k] == pattern byte, p + k < end (reads 8 bytes: the stream is padded)
    uint64_t v; __builtin_memcpy(&v, s + p, 8);
    const uint64_t x = v ^ pat;
    uint64_t z = ~(((x & 0x7f7f7f7f7f7f7f7full) + 0x7f7f7f7f7f7f7f7full) | x | 0x7f7f7f7f7f7f7f7full);
    if (p + 8 > end) z &= end > p ? (~0ull >> (8 * (p + 8 - end))) : 0ull;
    return z;
}
template <bool WRITE>
__global__ void __launch_bounds__(64) bcf_vep_wave(BcfStream st, BcfCellArgs a) {
    __shared__ uint32_t sep[VEP_WSEP];
    const uint32_t lane = threadIdx.x;
    const uint32_t colb = a.vep_cols[blockIdx.x % a.n_vep];
    const int64_t row = (int64_t)(blockIdx.x / a.n_vep);
    if (row >= a.nrows) return;
    const BcfColDev cd = a.cols[colb];
    const uint8_t *u = st.u;
    int64_t rec = a.tidy ? row / a.n_smp : row;
    if (a.sel) rec = a.sel[rec];
    const uint64_t o = a.rec_off[rec];
    uint32_t cnt = 0, nbytes = 0; bool valid = true;
    uint32_t cbase = 0, bbase = 0;
    if (WRITE) {
        if (cd.sa_cnt >= 0) cbase = a.offs[(size_t)cd.sa_cnt * a.ostride + row];
        if (cd.sa_bytes >= 0) bbase = a.offs[(size_t)cd.sa_bytes * a.ostride + row];
    }
    auto c_isspace = [](uint8_t ch) { return ch == ' ' || (ch >= 9 && ch <= 13); };
    const uint32_t d = a.dir[(size_t)(2 + cd.slot) * a.stride + rec];
    int n = 0, t = 0; uint64_t p = o + d;
    if ((cd.flags & BF_NULL_ALWAYS) || (a.tidy && row % a.n_smp != 0) || !d) valid = false;
    else { bcf_dec_size(u, p, n, t); if (n <= 0) valid = false; }
    if (valid) {
        const uint8_t *s = u + p;
        // the C string's length: the first NUL, 16 bytes per lane
        uint32_t l = (uint32_t)n;
        for (uint32_t base = 0; base < (uint32_t)n; base += 1024u) {
            const uint32_t q = base + lane * 16u;
            const uint64_t z = q < (uint32_t)n ? (vep_eq8(s, q, (uint32_t)n, 0) | 0) : 0ull, z2 = q + 8 < (uint32_t)n ? vep_eq8(s, q + 8, (uint32_t)n, 0) : 0ull;
            const uint32_t first = z ? q + (uint32_t)(__builtin_ctzll(z) >> 3) : z2 ? q + 8u + (uint32_t)(__builtin_ctzll(z2) >> 3) : 0xffffffffu;
            const unsigned long long hit = __ballot(first != 0xffffffffu);
            if (hit) { l = (uint32_t)__shfl((int)first, __builtin_ctzll(hit)); break; }
        }
        uint32_t seg = 0, nsep = 0;                                           // seg: where the first transcript of the collected stretch begins
        for (uint32_t base = 0;; base += 1024u) {
            const bool last = base + 1024u >= l;
            if (base < l) {
                const uint32_t q = base + lane * 16u;
                const uint64_t z = q < l ? vep_eq8(s, q, l, 0x2c2c2c2c2c2c2c2cull) : 0ull, z2 = q + 8 < l ? vep_eq8(s, q + 8, l, 0x2c2c2c2c2c2c2c2cull) : 0ull;
                uint32_t m = 0;                                               // bit k: s[q + k] == ','
#pragma unroll
                for (int k = 0; k < 8; k++) m |= (uint32_t)((z >> (8 * k + 7)) & 1ull) << k | (uint32_t)((z2 >> (8 * k + 7)) & 1ull) << (8 + k);
                uint32_t x = (uint32_t)__builtin_popcount(m), inc = x;
#pragma unroll
                for (int dd = 1; dd < 64; dd <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)inc, dd); if ((int)lane >= dd) inc += y; }
                uint32_t at = nsep + inc - x;
                for (; m; m &= m - 1) sep[at++] = q + (uint32_t)__builtin_ctz(m);
                nsep += (uint32_t)__shfl((int)inc, 63);
            }
            if (last) { if (lane == 0) sep[nsep] = l; nsep++; }                // the last transcript ends where the string does
            if (!last && nsep + 1025u <= VEP_WSEP) continue;
            __syncthreads();
            for (uint32_t k0 = 0; k0 < nsep; k0 += 64u) {                      // a transcript per lane; empty pieces do not count (strtok_r)
                const uint32_t k = k0 + lane;
                const uint32_t i0 = k < nsep ? (k ? sep[k - 1] + 1u : seg) : 0u, e = k < nsep ? sep[k] : 0u;
                const bool piece = k < nsep && e > i0;
                uint32_t f0 = i0, f1 = i0; bool have = piece;
                if (piece) {
                    int need = cd.vep_field;                                   // '|' to pass; fewer fields than asked for = missing
                    while (need > 0 && f0 < e) {
                        uint64_t z = vep_eq8(s, f0, e, 0x7c7c7c7c7c7c7c7cull);
                        const int c8 = __builtin_popcountll(z);
                        if (c8 < need) { need -= c8; f0 += 8; continue; }
                        for (int j = 1; j < need; j++) z &= z - 1;
                        f0 += (uint32_t)(__builtin_ctzll(z) >> 3) + 1u; need = 0;
                    }
                    if (need > 0) have = false;
                    else {
                        if (f0 > e) f0 = e;
                        f1 = f0;
                        for (;;) { if (f1 >= e) { f1 = e; break; } const uint64_t z = vep_eq8(s, f1, e, 0x7c7c7c7c7c7c7c7cull); if (z) { f1 += (uint32_t)(__builtin_ctzll(z) >> 3); break; } f1 += 8; }
                        while (f0 < f1 && c_isspace(s[f0])) f0++;
                        while (f1 > f0 + 1 && c_isspace(s[f1 - 1])) f1--;
                    }
                }
                const uint32_t tl = have ? f1 - f0 : 0;
                const bool miss = tl == 0 || (tl == 1 && s[f0] == '.');
                const uint32_t my_b = (piece && cd.htype != 1 && !miss) ? tl : 0u, my_c = piece ? 1u : 0u;
                uint32_t ic = my_c, ib = my_b;
#pragma unroll
                for (int dd = 1; dd < 64; dd <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)ic, dd), y2 = (uint32_t)__shfl_up((int)ib, dd); if ((int)lane >= dd) { ic += y; ib += y2; } }
                const uint32_t tot_c = (uint32_t)__shfl((int)ic, 63), tot_b = (uint32_t)__shfl((int)ib, 63);
                if (WRITE && piece) {
                    const uint32_t ci = cbase + cnt + ic - my_c, bo = bbase + nbytes + ib - my_b;
                    cd.child_valid[ci] = miss ? 0 : 1;
                    if (cd.htype == 1) {                                       // Integer: (int32_t)strtol(token, &end, 10), INT32_MIN unless the whole token is a number (vep_parse_int :207-220)
                        uint32_t w = 0;
                        if (!miss) {
                            uint32_t q = f0; const bool neg = s[q] == '-'; if (s[q] == '-' || s[q] == '+') q++;
                            uint64_t acc = 0; bool sat = false, ok = q < f1;
                            for (; q < f1; q++) {
                                const uint32_t dg = (uint32_t)s[q] - '0'; if (dg > 9) { ok = false; break; }
                                if (acc > (0x7fffffffffffffffull - dg) / 10) sat = true; else acc = acc * 10 + dg;
                            }
                            if (!ok) w = 0x80000000u;
                            else if (sat || (neg ? acc > 0x8000000000000000ull : acc > 0x7fffffffffffffffull)) w = neg ? 0u : 0xffffffffu;
                            else w = (uint32_t)(neg ? (uint64_t)0 - acc : acc);
                        }
                        cd.child_fixed[ci] = w;
                    } else {
                        cd.child_off[ci] = bo;
                        for (uint32_t j = 0; j < my_b; j++) cd.bytes[bo + j] = s[f0 + j];
                    }
                }
                cnt += tot_c; nbytes += tot_b;
            }
            seg = sep[nsep - 1] + 1u;
            __syncthreads();
            nsep = 0;
            if (last) break;
        }
        if (cnt == 0) valid = false;
    }
    if (lane != 0) return;
    if (!WRITE) {
        cd.valid[row] = valid ? 1 : 0;
        if (!valid) { cnt = 0; nbytes = 0; }
        if (cd.sa_cnt >= 0) a.lens[(size_t)cd.sa_cnt * a.ostride + row] = cnt;
        if (cd.sa_bytes >= 0) a.lens[(size_t)cd.sa_bytes * a.ostride + row] = nbytes;
    } else if (row == a.nrows - 1 && cd.child_off && cd.sa_cnt >= 0 && cd.sa_bytes >= 0) {
        cd.child_off[a.offs[(size_t)cd.sa_cnt * a.ostride + a.nrows]] = a.offs[(size_t)cd.sa_bytes * a.ostride + a.nrows];
    }
}

// rid, pos, rlen of every record (the three int32 behind the two length words) and the position's high word, packed for the index writer
extern "C" __global__ void __launch_bounds__(256)
bcf_index_rows(const uint8_t *__restrict__ u, const uint32_t *__restrict__ rec_off, int64_t nrec, const int32_t *__restrict__ pos_hi, uint32_t *__restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrec) return;
    const uint64_t o = rec_off[r];
    out[4 * r] = ldu32(u + o + 8); out[4 * r + 1] = ldu32(u + o + 12); out[4 * r + 2] = ldu32(u + o + 16);
    out[4 * r + 3] = pos_hi ? (uint32_t)pos_hi[r] : (out[4 * r + 1] == 0xffffffffu ? 0xffffffffu : 0u);          // bits 32.. of the position (text), the sign of BCF's -1
}

// ---- region predicate (bcf_itr_querys -> hts_itr_next hts.c:4287-4300 over bcf_readrec vcf.c:2267-2276) ----------------------
// keep[r] = rid == tid && end > beg_q && end_q > beg, beg = pos, end = pos + rlen (a negative rlen falls back to the REF allele
// length, the first choice of get_rlen vcf.c:6440-6560); all != 0 keeps every record (region ".").
extern "C" __global__ void __launch_bounds__(256)
bcf_region_keep(BcfStream st, const uint32_t *rec_off, const uint32_t *dir, int64_t nrec, int32_t tid, int64_t qbeg, int64_t qend, int32_t all, uint32_t *keep) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrec) return;
    const uint8_t *u = st.u; const uint64_t o = rec_off[r];
    uint32_t k = 0;
    if (all) k = 1;
    else if ((int32_t)ldu32(u + o + 8) == tid) {
        const int64_t beg = bcf_pos64(st, u, o, r);
        int64_t rlen = (int32_t)ldu32(u + o + 16);
        if (rlen < 0) { rlen = 0; const uint32_t d = dir[r]; if (d) { uint64_t p = o + d; int n, t; bcf_dec_size(u, p, n, t); rlen = n > 0 ? (int64_t)cstr_len(u + p, (uint32_t)n) : 0; } }
        const int64_t end = beg + rlen;
        if (end > qbeg && qend > beg) k = 1;
    }
    keep[r] = k;
}
extern "C" __global__ void __launch_bounds__(256)
bcf_select(const uint32_t *map, int64_t nrec, uint32_t *sel) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < nrec && map[r + 1] > map[r]) sel[map[r]] = (uint32_t)r;
}

// ---- matrix exclusive scan: rows of `lens` (narr x stride) -> rows of `offs`, off[n] = total -------------------------------
struct MScanArgs { const uint32_t *in; uint32_t *out; uint64_t *partial; uint64_t *total; uint32_t stride; int64_t n; int64_t nparts; };

extern "C" __global__ void __launch_bounds__(256) mscan_reduce(MScanArgs a) {
    __shared__ uint32_t sh[256];
    const uint32_t *in = a.in + (size_t)blockIdx.y * a.stride;
    int64_t base = (int64_t)blockIdx.x * SCAN_ITEMS;
    uint32_t s = 0;
    for (int k = 0; k < 16; k++) { int64_t i = base + k * 256 + threadIdx.x; if (i < a.n) s += in[i]; }
    sh[threadIdx.x] = s; __syncthreads();
    for (int d = 128; d > 0; d >>= 1) { if ((int)threadIdx.x < d) sh[threadIdx.x] += sh[threadIdx.x + d]; __syncthreads(); }
    if (threadIdx.x == 0) a.partial[(size_t)blockIdx.y * a.nparts + blockIdx.x] = sh[0];
}
extern "C" __global__ void __launch_bounds__(1024) mscan_partials(MScanArgs a) {
    __shared__ uint64_t sh[1024]; __shared__ uint64_t carry;
    uint64_t *p = a.partial + (size_t)blockIdx.x * a.nparts;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < a.nparts; base += 1024) {
        int64_t i = base + threadIdx.x;
        uint64_t v = i < a.nparts ? p[i] : 0;
        sh[threadIdx.x] = v; __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) { uint64_t t = ((int)threadIdx.x >= d) ? sh[threadIdx.x - d] : 0; __syncthreads(); sh[threadIdx.x] += t; __syncthreads(); }
        if (i < a.nparts) p[i] = carry + sh[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += sh[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) a.total[blockIdx.x] = carry;
}
extern "C" __global__ void __launch_bounds__(256) mscan_apply(MScanArgs a) {
    __shared__ uint32_t sh[256];
    const uint32_t *in = a.in + (size_t)blockIdx.y * a.stride; uint32_t *out = a.out + (size_t)blockIdx.y * a.stride;
    int64_t base = (int64_t)blockIdx.x * SCAN_ITEMS + (int64_t)threadIdx.x * 16;
    uint32_t v[16]; uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) { int64_t i = base + k; v[k] = i < a.n ? in[i] : 0; s += v[k]; }
    sh[threadIdx.x] = s; __syncthreads();
    for (int d = 1; d < 256; d <<= 1) { uint32_t t = ((int)threadIdx.x >= d) ? sh[threadIdx.x - d] : 0; __syncthreads(); sh[threadIdx.x] += t; __syncthreads(); }
    uint64_t run = a.partial[(size_t)blockIdx.y * a.nparts + blockIdx.x] + sh[threadIdx.x] - s;
#pragma unroll
    for (int k = 0; k < 16; k++) { int64_t i = base + k; if (i <= a.n) out[i] = (uint32_t)run; run += v[k]; }
}
