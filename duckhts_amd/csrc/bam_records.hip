// bam_records.hip -- BAM record boundary discovery + column unpack for MI355X (gfx950).
//
// Replaces, for the read_bam scan (src/bam_reader.c:722-1038):
//   htslib sam.c:779-855 bam_read1 (framing + validation), sam.c:675-730 bam_tag2cigar,
//   sam.c:4124-4134 sam_read1_bam (tid range), sam.c:4785-4855 aux walk (RG lookup), and the
//   column writers bam_reader.c:785-918 (QNAME..SAMPLE_ID).
//
// The inflated BAM stream is a linked list (next = cur + 4 + block_size).  It is cut into
// fixed TILE-byte tiles; bam_tiles_lds.hip speculates the first record start inside each tile
// (same predicates bam_read1 enforces, chained 3 deep) and walks to the tile end.  The fix-up
// kernels here prove continuity tile-to-tile (end of tile t-1's chain == start of tile t) and
// re-walk only the tiles whose speculation was wrong, so the result is exact.  Row ids come
// from a prefix sum of per-tile counts; string columns are written with a length pass +
// prefix sums + a 16-lanes-per-record write pass.  Integer/byte work only.
#include "dhts_common.h"

#define REC_OK 0
#define REC_INVALID 1
#define REC_INCOMPLETE 2
#define NONE64 0xffffffffffffffffull

struct BamStream {
    const uint8_t *u;      // inflated bytes (this batch, carry included)
    uint64_t ulen;         // valid bytes
    int32_t n_ref;
    int32_t final_batch;   // 1: no more data follows (an incomplete tail is a truncation)
    int32_t seq_packed;    // 1: the SEQ heap keeps the file's 4-bit codes ((l_seq + 1) / 2 bytes per row; the consumer expands them)
    int32_t want_rg;       // 0: neither READ_GROUP_ID nor SAMPLE_ID is projected: the walk over the auxiliary fields to RG is skipped
};

__device__ __forceinline__ uint32_t ldu32(const uint8_t *p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }

// bam_cigar_type (htslib/sam.h:139-148): bit0 = consumes query
#define CIG_QUERY(op) ((0x3C1A7u >> ((op) << 1)) & 1u)

__device__ __forceinline__ int aux_size(uint8_t t) {
    switch (t) {
    case 'A': case 'c': case 'C': return 1;
    case 's': case 'S': return 2;
    case 'i': case 'I': case 'f': return 4;
    case 'd': return 8;
    default: return 0;
    }
}
struct RecInfo {
    uint32_t block_len, l_qname, n_cigar, flag, mapq;
    int32_t tid, pos, l_seq, mtid, mpos, tlen;
    uint64_t cig_off;      // absolute offset of the effective CIGAR (CG-swapped if needed)
    uint32_t n_cigar_eff;
    uint64_t cg_beg, cg_end;   // spliced-out CG tag, or (0,0)
};

// (the bam_read1 checks themselves live in bam_tiles_lds.hip: rec_check_t / rec_hop)
struct TileOut {
    uint64_t *first;     // first record start in tile (absolute offset in this batch buffer) or NONE64
    uint64_t *end_next;  // where the chain leaves the tile (start of the first record at/after tile end, or stuck position)
    uint32_t *count;     // records starting in this tile
    int32_t *err;        // 0, or 1 = chain stopped on an invalid record at end_next
};

// chain walk used by the repair kernels: same core-only tests as rec_hop in bam_tiles_lds.hip (the CIGAR/qlen test and the
// CG swap are checked per row by bam_tile_unpack)
__device__ void tile_walk(const BamStream &st, uint64_t start, uint64_t tile_end, uint64_t &end_next, uint32_t &count, int &err) {
    uint64_t o = start; uint32_t c = 0; err = 0;
    const uint8_t *u = st.u;
    while (o < tile_end) {
        int rc = REC_OK; uint32_t bl = 0;
        if (st.ulen - o < 4) rc = REC_INCOMPLETE;
        else {
            const int32_t b = (int32_t)ldu32(u + o);
            if (b < 32) rc = REC_INVALID;
            else if (st.ulen - o - 4 < 32) rc = REC_INCOMPLETE;
            else {
                const int32_t tid = (int32_t)ldu32(u + o + 4), mtid = (int32_t)ldu32(u + o + 24), l_seq = (int32_t)ldu32(u + o + 20);
                const uint32_t l_qname = ldu32(u + o + 12) & 0xff, n_cigar = ldu32(u + o + 16) & 0xffff;
                const uint64_t body = (uint64_t)(uint32_t)b - 32;
                if (l_seq < 0 || l_qname < 1) rc = REC_INVALID;
                else if (((uint64_t)n_cigar << 2) + l_qname + (((uint64_t)l_seq + 1) >> 1) + (uint64_t)l_seq > body) rc = REC_INVALID;
                else if (st.ulen - o - 36 < body) rc = REC_INCOMPLETE;
                else if (tid >= st.n_ref || tid < -1 || mtid >= st.n_ref || mtid < -1) rc = REC_INVALID;
                bl = (uint32_t)b;
            }
        }
        if (rc == REC_INCOMPLETE) { if (st.final_batch && o < st.ulen) err = 1; break; }   // truncated tail = read error (sam.c:790-791,833-835)
        if (rc == REC_INVALID) { err = 1; break; }
        c++; o += 4ull + bl;
    }
    end_next = o; count = c;
}

// One round of continuity proof + repair, OUT OF PLACE (reads `in`, writes `out` for every tile; the host swaps), so a round
// never observes half-updated neighbours.  A tile is re-walked only when its predecessor is itself consistent with ITS
// predecessor: a mis-speculated tile is repaired first and its (correctly speculated) successors simply wait one round,
// instead of being re-walked from a wrong position.  nfixed counts the tiles changed this round.
__device__ __forceinline__ bool tile_consistent(const TileOut &in, int64_t t, uint64_t te_t) {
    // is tile t consistent with the chain exit of tile t-1?   (t >= 1)
    const uint64_t p = in.end_next[t - 1];
    if (p == NONE64 || in.err[t - 1]) return false;
    if (p >= te_t) return in.first[t] == NONE64 && in.end_next[t] == p && in.count[t] == 0;
    return in.first[t] == p;
}
extern "C" __global__ void __launch_bounds__(256)
bam_tile_fix(BamStream st, uint32_t tile_bytes, int64_t ntiles, TileOut in, TileOut out, uint32_t *nfixed) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntiles) return;
    uint64_t f = in.first[t], en = in.end_next[t]; uint32_t cnt = in.count[t]; int32_t err = in.err[t];
    if (t >= 1) {
        const uint64_t tb = (uint64_t)t * tile_bytes; uint64_t te = tb + tile_bytes; if (te > st.ulen) te = st.ulen;
        const uint64_t p = in.end_next[t - 1];
        bool pred_ok = (p != NONE64) && !in.err[t - 1];
        if (pred_ok && t >= 2) pred_ok = tile_consistent(in, t - 1, tb);      // tile_end(t-1) == tile_begin(t)
        if (pred_ok && !tile_consistent(in, t, te)) {
            if (p >= te) { f = NONE64; cnt = 0; err = 0; en = p; }          // tile lies inside a record that started earlier
            else { int e; tile_walk(st, p, te, en, cnt, e); f = p; err = e; }
            atomicAdd(nfixed, 1u);
        }
    }
    out.first[t] = f; out.end_next[t] = en; out.count[t] = cnt; out.err[t] = err;
}

// Sequential fallback (pathological inputs only): one thread proves/repairs every tile in order.
extern "C" __global__ void bam_tile_fix_seq(BamStream st, uint32_t tile_bytes, int64_t ntiles, TileOut out) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    for (int64_t t = 1; t < ntiles; t++) {
        if (out.err[t - 1]) return;
        uint64_t p = out.end_next[t - 1];
        uint64_t tb = (uint64_t)t * tile_bytes, te = tb + tile_bytes; if (te > st.ulen) te = st.ulen;
        if (p >= te) { out.first[t] = NONE64; out.count[t] = 0; out.err[t] = 0; out.end_next[t] = p; continue; }
        if (out.first[t] == p) continue;
        uint64_t en; uint32_t cnt; int err;
        tile_walk(st, p, te, en, cnt, err);
        out.first[t] = p; out.count[t] = cnt; out.err[t] = err; out.end_next[t] = en;
    }
}

// Single-workgroup finalize: first error tile E, exclusive prefix of counts, totals (rows of tiles after E are never used).
// res[0] = total rows (tiles 0..E), res[1] = stream position after the last good record (carry start),
// res[2] = 1 if the chain stopped on an error, res[3] = E (or ntiles)
// 16 consecutive tiles per thread, DPP wave scans, one LDS exchange per 16,384 tiles.
extern "C" __global__ void __launch_bounds__(1024)
bam_tile_finalize(int64_t ntiles, TileOut out, uint32_t *rowbase, uint64_t *res) {
    __shared__ uint32_t shw[16];
    __shared__ unsigned long long shE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) shE = (unsigned long long)ntiles;
    __syncthreads();
    uint32_t carry = 0;
    unsigned long long myE = (unsigned long long)ntiles;
    for (int64_t base = 0; base < ntiles; base += 16384) {
        const int64_t t0 = base + (int64_t)tid * 16;
        uint32_t v[16], e[16], sum = 0;
        if (t0 + 16 <= ntiles) {                      // 4 + 4 aligned 16-byte loads, all in flight together
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint4 a = *(const uint4 *)(out.count + t0 + 4 * q), b = *(const uint4 *)((const uint32_t *)out.err + t0 + 4 * q);
                v[4 * q] = a.x; v[4 * q + 1] = a.y; v[4 * q + 2] = a.z; v[4 * q + 3] = a.w;
                e[4 * q] = b.x; e[4 * q + 1] = b.y; e[4 * q + 2] = b.z; e[4 * q + 3] = b.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; k++) { const int64_t t = t0 + k; v[k] = t < ntiles ? out.count[t] : 0u; e[k] = t < ntiles ? (uint32_t)out.err[t] : 0u; }
        }
#pragma unroll
        for (int k = 15; k >= 0; k--) if (e[k] && (unsigned long long)(t0 + k) < myE) myE = (unsigned long long)(t0 + k);
#pragma unroll
        for (int k = 0; k < 16; k++) sum += v[k];
        const uint32_t incl = wave_incl_scan(sum, lane);
        if (lane == 63) shw[wave] = incl;
        __syncthreads();
        uint32_t woff = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) { const uint32_t x = shw[w]; total += x; if (w < wave) woff += x; }
        uint32_t run = carry + woff + incl - sum;
        if (t0 + 16 <= ntiles) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                uint4 o4; o4.x = run; run += v[4 * q]; o4.y = run; run += v[4 * q + 1]; o4.z = run; run += v[4 * q + 2]; o4.w = run; run += v[4 * q + 3];
                *(uint4 *)(rowbase + t0 + 4 * q) = o4;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; k++) { const int64_t t = t0 + k; if (t < ntiles) rowbase[t] = run; run += v[k]; }
        }
        carry += total;
        __syncthreads();
    }
    if (myE < (unsigned long long)ntiles) atomicMin(&shE, myE);
    __threadfence_block();
    __syncthreads();
    if (tid == 0) {
        const int64_t E = (int64_t)shE;
        const int64_t last = (E < ntiles) ? E : ntiles - 1;
        res[0] = (E < ntiles) ? (uint64_t)rowbase[E] + out.count[E] : (uint64_t)carry;
        res[1] = out.end_next[last];
        res[2] = (E < ntiles) ? 1 : 0;
        res[3] = (uint64_t)E;
    }
}

// second walk: record offsets by row id
extern "C" __global__ void __launch_bounds__(256)
bam_tile_offsets(BamStream st, uint32_t tile_bytes, int64_t ntiles, TileOut out, const uint32_t *rowbase, const uint64_t *res,
                 uint32_t *rec_off) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntiles || (uint64_t)t > res[3]) return;
    uint64_t o = out.first[t]; if (o == NONE64) return;
    uint32_t n = out.count[t], row = rowbase[t];
    for (uint32_t k = 0; k < n; k++) { rec_off[row + k] = (uint32_t)o; o += 4ull + ldu32(st.u + o); }
}

// ------------------------------------------------------------------------------------
// column unpack
// ------------------------------------------------------------------------------------
struct BamDict {
    int32_t n_rg;
    const uint32_t *rg_off;     // n_rg+1 offsets into rg_bytes (@RG ID strings, header order, first ID wins)
    const uint8_t *rg_bytes;
    int32_t n_bytes;            // rg_off[n_rg]
};

struct BamCols {
    // fixed-width, DuckDB physical types (src/bam_reader.c:514-526)
    uint16_t *flag; int64_t *pos; int32_t *mapq; int64_t *pnext; int64_t *tlen;
    int32_t *tid, *mtid;        // dictionary ids behind RNAME / RNEXT (names resolved by sam_hdr_tid2name on the host mirror)
    int32_t *rg_idx;            // dictionary id behind SAMPLE_ID (-1 = NULL)
    uint64_t *rg_valid;         // validity words of READ_GROUP_ID
    // var-width: reserved length per row (scanned into offsets) and actual length
    uint32_t *len_qname, *len_cigar, *len_seq, *len_qual, *len_rg;
    // per-row scratch for the write pass
    uint32_t *cig_rel;          // effective CIGAR offset relative to the record start
    uint32_t *ncig_eff;
    uint32_t *rg_rel;           // RG value offset relative to the record start
};

__device__ __forceinline__ uint32_t ndigits(uint32_t v) {
    return 1 + (v >= 10) + (v >= 100) + (v >= 1000) + (v >= 10000) + (v >= 100000) + (v >= 1000000) + (v >= 10000000) + (v >= 100000000);
}

// ---- generic multi-array exclusive scan (u32 in; u32 or u64 out), 3 launches for up to 8 arrays at once ----
#define SCAN_ITEMS 4096     /* per workgroup: 256 threads x 16 */
struct ScanArgs { const uint32_t *in[8]; uint32_t *out32[8]; uint64_t *out64[8]; uint64_t *partial[8]; uint64_t *total[8]; int64_t n; int narr;
                  const unsigned long long *n_dev; };   // n_dev: the element count lives on the device (at most n, the launch's size): a batch whose row count the host has not seen
__device__ __forceinline__ int64_t scan_n(const ScanArgs &a) { if (!a.n_dev) return a.n; const int64_t d = (int64_t)*a.n_dev; return d < a.n ? d : a.n; }

extern "C" __global__ void __launch_bounds__(256) scan_reduce(ScanArgs a) {
    __shared__ uint32_t sh[256];
    const int arr = blockIdx.y; const uint32_t *in = a.in[arr];
    int64_t base = (int64_t)blockIdx.x * SCAN_ITEMS;
    const int64_t n = scan_n(a);
    uint32_t s = 0;
    for (int k = 0; k < 16; k++) { int64_t i = base + k * 256 + threadIdx.x; if (i < n) s += in[i]; }
    sh[threadIdx.x] = s; __syncthreads();
    for (int d = 128; d > 0; d >>= 1) { if ((int)threadIdx.x < d) sh[threadIdx.x] += sh[threadIdx.x + d]; __syncthreads(); }
    if (threadIdx.x == 0) a.partial[arr][blockIdx.x] = sh[0];
}
extern "C" __global__ void __launch_bounds__(1024) scan_partials(ScanArgs a, int64_t nparts) {
    __shared__ uint64_t sh[1024]; __shared__ uint64_t carry;
    const int arr = blockIdx.x; uint64_t *p = a.partial[arr];
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < nparts; base += 1024) {
        int64_t i = base + threadIdx.x;
        uint64_t v = i < nparts ? p[i] : 0;
        sh[threadIdx.x] = v; __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) { uint64_t t = ((int)threadIdx.x >= d) ? sh[threadIdx.x - d] : 0; __syncthreads(); sh[threadIdx.x] += t; __syncthreads(); }
        if (i < nparts) p[i] = carry + sh[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += sh[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) *a.total[arr] = carry;
}
extern "C" __global__ void __launch_bounds__(256) scan_apply(ScanArgs a) {
    __shared__ uint32_t sh[256];
    const int arr = blockIdx.y; const uint32_t *in = a.in[arr]; uint32_t *o32 = a.out32[arr]; uint64_t *o64 = a.out64[arr];
    int64_t base = (int64_t)blockIdx.x * SCAN_ITEMS + (int64_t)threadIdx.x * 16;
    const int64_t n = scan_n(a);
    uint32_t v[16]; uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) { int64_t i = base + k; v[k] = i < n ? in[i] : 0; s += v[k]; }
    sh[threadIdx.x] = s; __syncthreads();
    for (int d = 1; d < 256; d <<= 1) { uint32_t t = ((int)threadIdx.x >= d) ? sh[threadIdx.x - d] : 0; __syncthreads(); sh[threadIdx.x] += t; __syncthreads(); }
    uint64_t run = a.partial[arr][blockIdx.x] + sh[threadIdx.x] - s;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        int64_t i = base + k;
        if (i <= n) { if (o32) o32[i] = (uint32_t)run; if (o64) o64[i] = run; }     // i == n writes the total (off[n])
        run += v[k];
    }
}

// ---- string write pass (bam_tile_strings in bam_tiles_lds.hip): outputs and 16-byte converters ----
struct BamStrOut {
    const uint32_t *off_qname, *off_cigar, *off_seq, *off_qual, *off_rg;
    uint8_t *qname, *cigar, *seq, *qual, *rg;
    uint32_t *alen_qual;     // actual QUAL length (the reference assigns through a NUL-terminated API: byte 223 (+33 == 0) truncates)
    uint32_t *seq_chars;     // packed SEQ: bases per row (0: the row's SEQ is "*"); NULL otherwise
};

// SEQ: 16 bases (8 packed bytes p0,p1) -> 16 chars; "=ACMGRSVTWYHKDBN" (hts.c:260), high nibble first (sam.h:325).
// The 16-entry byte table is held in four dwords; v_perm_b32 gathers 4 table bytes per instruction.
__device__ __forceinline__ void seq16(uint32_t p0, uint32_t p1, uint32_t w[4]) {
    const uint32_t T0 = 0x4d43413du, T1 = 0x56535247u, T2 = 0x48595754u, T3 = 0x4e42444bu;   // "=ACM" "GRSV" "TWYH" "KDBN"
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t x = ((k & 2) ? p1 : p0) >> ((k & 1) * 16);                   // two packed bytes = four bases
        uint32_t sel = ((x >> 4) & 0xfu) | ((x & 0xfu) << 8) | (((x >> 12) & 0xfu) << 16) | (((x >> 8) & 0xfu) << 24);
        uint32_t lo = __builtin_amdgcn_perm(T1, T0, sel & 0x07070707u);
        uint32_t hi = __builtin_amdgcn_perm(T3, T2, sel & 0x07070707u);
        uint32_t m = ((sel >> 3) & 0x01010101u) * 0xffu;
        w[k] = (hi & m) | (lo & ~m);
    }
}
// QUAL: bytewise +33 without cross-byte carry; returns true if some byte became NUL
__device__ __forceinline__ bool qual16(uint32_t w[4]) {
    bool z = false;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t x = w[k];
        uint32_t t = ((x & 0x7f7f7f7fu) + 0x21212121u) ^ (x & 0x80808080u);
        w[k] = t;
        z |= (((t - 0x01010101u) & ~t & 0x80808080u) != 0);
    }
    return z;
}
__device__ __forceinline__ void store_n16(uint8_t *d, const uint32_t w[4], uint32_t n) {
    if (n >= 16) { uint4 v = make_uint4(w[0], w[1], w[2], w[3]); __builtin_memcpy(d, &v, 16); }
    else {
        // exact-length tail without a byte loop: 8 / 4 / 2 / 1 pieces, register moves only (no indexed access to w)
        uint32_t a0 = w[0], a1 = w[1];
        if (n & 8) { uint2 v = make_uint2(a0, a1); __builtin_memcpy(d, &v, 8); d += 8; a0 = w[2]; a1 = w[3]; }
        if (n & 4) { __builtin_memcpy(d, &a0, 4); d += 4; a0 = a1; }
        if (n & 2) { uint16_t h = (uint16_t)a0; __builtin_memcpy(d, &h, 2); d += 2; a0 >>= 16; }
        if (n & 1) *d = (uint8_t)a0;
    }
}

// ------------------------------------------------------------------------------------
// BGZF block discovery on resident bytes: signature scan + chain proof
// ------------------------------------------------------------------------------------
// One wave owns a 64 KiB span and sweeps it in 1 KiB coalesced pieces (16 B per lane); a lane tests its
// 16 byte positions for the 4-byte magic with byte-aligns, then the remaining signature bytes on a hit.
// pass 1 counts per span; pass 2 (after a prefix sum over spans) writes the candidates in file order.
#define SIG_SLOTS 16u                     /* candidates kept per 64 KiB span by the counting sweep */
__device__ __forceinline__ bool bgzf_sig_rest(const uint8_t *p) {
    // htslib bgzf.c:896-903 check_header == 0 (ID1 ID2 CM FLG.FEXTRA already matched): XLEN==6, 'B','C', SLEN==2
    return p[10] == 6 && p[11] == 0 && p[12] == 'B' && p[13] == 'C' && p[14] == 2 && p[15] == 0;
}
__device__ __forceinline__ uint32_t sig_hits16(const uint8_t *d, uint64_t n, uint64_t pos, int lane) {
    // returns a 16-bit mask of signature starts in [pos, pos+16)
    uint32_t w[5] = {0, 0, 0, 0, 0};
    if (pos + 16 <= n) { uint4 v = *(const uint4 *)(d + pos); w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; }
    else for (uint64_t k = pos; k < n && k < pos + 16; k++) w[(k - pos) >> 2] |= (uint32_t)d[k] << (8 * ((k - pos) & 3));
    uint32_t nx = __shfl_down(w[0], 1, 64);
    if (lane == 63) { nx = 0; for (uint64_t k = pos + 16; k < n && k < pos + 20; k++) nx |= (uint32_t)d[k] << (8 * (k - pos - 16)); }
    w[4] = nx;
    uint32_t m = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        uint32_t lo = w[j >> 2], hi = w[(j >> 2) + 1];
        uint32_t v = (j & 3) ? ((lo >> (8 * (j & 3))) | (hi << (32 - 8 * (j & 3)))) : lo;
        if ((v & 0x04ffffffu) == 0x04088b1fu) m |= 1u << j;           // 1f 8b 08, FLG has FEXTRA
    }
    uint32_t out = 0;
    while (m) { int j = __ffs(m) - 1; m &= m - 1; if (pos + j + 18 <= n && bgzf_sig_rest(d + pos + j)) out |= 1u << j; }
    return out;
}
extern "C" __global__ void __launch_bounds__(256)
bgzf_sig_count(const uint8_t *d, uint64_t n, uint32_t *cnt, uint16_t *hits, uint32_t *overflow) {
    // besides the count, the first SIG_SLOTS candidates of the span are kept (offsets inside the span, file order), so that the
    // usual case -- a handful of blocks per 64 KiB -- needs no second sweep over the file (bgzf_sig_gather); a span with more
    // raises `overflow` and the host runs bgzf_sig_write over everything.
    const int lane = threadIdx.x & 63;
    int64_t span = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint64_t b = (uint64_t)span * 65536;
    if (b >= n) return;
    uint32_t w = 0;
    for (uint64_t p = b; p < b + 65536 && p < n; p += 1024) {
        uint32_t m = sig_hits16(d, n, p + lane * 16, lane);
        if (__ballot(m != 0) == 0ull) continue;
        const uint32_t c = __popc(m); uint32_t inc = c;
#pragma unroll
        for (int k = 1; k < 64; k <<= 1) { uint32_t t = __shfl_up(inc, k, 64); if (lane >= k) inc += t; }
        uint32_t at = w + inc - c;
        while (m) { const int j = __ffs(m) - 1; m &= m - 1; if (at < SIG_SLOTS) hits[span * SIG_SLOTS + at] = (uint16_t)(p - b + lane * 16 + j); at++; }
        w += __shfl(inc, 63, 64);
    }
    if (lane == 0) { cnt[span] = w; if (w > SIG_SLOTS) atomicOr(overflow, 1u); }
}
extern "C" __global__ void __launch_bounds__(256)
bgzf_sig_gather(const uint32_t *cnt, const uint32_t *base, const uint16_t *hits, int64_t nspans, uint64_t *cand, uint64_t add) {
    const int64_t span = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (span >= nspans) return;
    const uint32_t c = cnt[span] < SIG_SLOTS ? cnt[span] : SIG_SLOTS, w = base[span];
    for (uint32_t k = 0; k < c; k++) cand[w + k] = add + (uint64_t)span * 65536 + hits[span * SIG_SLOTS + k];
}
extern "C" __global__ void __launch_bounds__(256)
bgzf_sig_write(const uint8_t *d, uint64_t n, const uint32_t *base, uint64_t *cand, uint64_t add) {
    const int lane = threadIdx.x & 63;
    int64_t span = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint64_t b = (uint64_t)span * 65536;
    if (b >= n) return;
    uint32_t w = base[span];
    for (uint64_t p = b; p < b + 65536 && p < n; p += 1024) {
        uint32_t m = sig_hits16(d, n, p + lane * 16, lane);
        uint32_t c = __popc(m), inc = c;
#pragma unroll
        for (int k = 1; k < 64; k <<= 1) { uint32_t t = __shfl_up(inc, k, 64); if (lane >= k) inc += t; }
        uint32_t at = w + inc - c;
        while (m) { int j = __ffs(m) - 1; m &= m - 1; cand[at++] = add + p + lane * 16 + j; }
        w += __shfl(inc, 63, 64);
    }
}
// The ISIZE trailer field as the block table keeps it.  A BGZF block inflates to at most 65,536 bytes, so a larger claim can never be
// met: it is recorded as 65,537, which (a) still fails phase B's `outlen == ISIZE` test -- the stream ends at that block like for
// any other wrong ISIZE -- and (b) keeps uoff, the prefix sum of these values, monotone and far from 32-bit wrap-around whatever
// bytes a damaged or hostile file carries there (a raw 0xFFFFxxxx used to wrap the u32 partial sums and misplace its neighbours).
__device__ __forceinline__ uint32_t isize_placed(uint32_t raw) { return raw > 65536u ? 65537u : raw; }
// Sequential restatement of the htslib chain walk (bgzf.c:1155-1236); used only when the parallel proof fails
// (corrupt or unusual files).  res[0] = number of good blocks, res[1] = status (0 clean end, -1 bad header, -2 short block).
extern "C" __global__ void bgzf_chain_walk_seq(const uint8_t *d, uint64_t n, uint64_t *coff, uint32_t *clen, uint32_t *isize,
                                               int64_t cap, int64_t *res) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    uint64_t o = 0; int64_t k = 0; int64_t status = 0;
    while (o < n && k < cap) {
        if (n - o < 18) { status = -1; break; }
        const uint8_t *p = d + o;
        if (!(p[0] == 31 && p[1] == 139 && p[2] == 8 && (p[3] & 4) && bgzf_sig_rest(p))) { status = -1; break; }
        uint32_t bl = ((uint32_t)p[16] | ((uint32_t)p[17] << 8)) + 1;
        if (bl < 18) { status = -1; break; }
        if (o + bl > n) { status = -2; break; }
        coff[k] = o; clen[k] = bl; isize[k] = bl >= 26 ? isize_placed(ldu32(p + bl - 4)) : 0;
        k++; o += bl;
    }
    res[0] = k; res[1] = status;
}
// chain proof: candidate i must start exactly where candidate i-1 ends; fills the block table.
// bad[0] counts violations (then the host falls back to a sequential chain walk).
// partial_tail: the resident bytes are a window that stops short of the end of the file, so the last block may be cut off.  Candidates
// whose block runs past the end of the buffer are then counted in bad[1] instead ("cut"), and must be the last ones.
extern "C" __global__ void __launch_bounds__(256)
bgzf_chain_check(const uint8_t *d, uint64_t n, const uint64_t *cand, int64_t ncand, uint32_t *clen, uint32_t *isize, uint32_t *bad, int partial_tail, uint64_t first_off) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ncand) return;
    uint64_t o = cand[i];
    uint32_t bl = ((uint32_t)d[o + 16] | ((uint32_t)d[o + 17] << 8)) + 1;
    bool ok = true, cut = false;
    if (i == 0 && o != first_off) ok = false;            // the chain starts at the top of the buffer (or, for an extension, behind the last known block)
    if (bl < 26) ok = false;
    if (o + bl > n) { if (partial_tail) cut = true; else ok = false; }
    if (ok && !cut) { if (i + 1 < ncand) ok = (cand[i + 1] == o + bl); else ok = partial_tail ? true : (o + bl == n); }
    if (ok && cut && i + 1 < ncand) {
        // everything behind a cut block must be cut as well (a candidate inside the cut block's span)
        const uint64_t o2 = cand[i + 1]; const uint32_t bl2 = ((uint32_t)d[o2 + 16] | ((uint32_t)d[o2 + 17] << 8)) + 1;
        if (o2 + bl2 <= n) ok = false;
    }
    clen[i] = bl;
    isize[i] = (ok && !cut) ? isize_placed(ldu32(d + o + bl - 4)) : 0;
    if (!ok) atomicAdd(bad, 1u);
    else if (cut) atomicAdd(bad + 1, 1u);
}
