// bam_tile_rows.hip -- the row pass of the BAM record stage in ONE staging of every 8 KiB tile (gfx950; round 4).
//
// Replaces the three passes bam_tile_unpack -> scan_* -> bam_tile_strings of bam_tiles_lds.hip for the plain scan (no region filter, no
// shard cut inside the batch): the tile is staged in LDS once, its records are validated and unpacked (one lane per record, the code of
// unpack_one), the five string heaps' byte counts of the tile are published and the tile's place in every heap comes from a DECOUPLED
// LOOK-BACK over the tiles in front of it (Merrill & Garland's single-pass scan: a tile publishes its aggregate, then adds up the
// aggregates of its predecessors until it meets one that already knows its inclusive prefix), and the strings are written from the same
// LDS image.  The inflated stream is read once by this pass (three times before: unpack, strings, and the per-row offset / length arrays
// that travelled through HBM between them), and the heap sizes no longer have to reach the host before the strings can be written: the
// arenas are sized from the previous batch and guarded here (a tile whose strings would not fit raises a flag and writes none; the host
// grows the arenas and repeats the pass).
//   replaces: htslib sam.c:779-855 bam_read1 (full validation), sam.c:675-730 bam_tag2cigar, sam.c:4785-4855 aux walk;
//             src/bam_reader.c:785-918 (the 13 column writers), 560-640 (seq_to_string / qual_to_string / cigar text).
//
// Look-back protocol (placement-independent; MI355X_MICROARCH.md, "Workgroup dispatch ... inter-workgroup visibility"): tiles are taken
// in TICKET order (one agent-scope atomic add per workgroup), so every tile a workgroup waits for belongs to a workgroup that is already
// running.  A tile's record is five naturally aligned 8-byte granules {state, value}, each written by ONE relaxed agent-scope store and
// read by relaxed agent-scope loads (sc1: L2-served, never L1): data and tag travel in one word, so no fence is needed.  The wait is
// bounded: a tile that has polled 2^22 times raises LOOK_TIMEOUT (the host fails the batch) instead of hanging the queue.
#pragma once

#define TR_THREADS 128
#define TR_LOOK_W 12u                    /* predecessors examined per look-back step: 5 heaps x 12 = 60 lanes, one 8-byte load each */
#define TR_ST_EMPTY 0u
#define TR_ST_AGG 1u                     /* value = the tile's own bytes */
#define TR_ST_INCL 2u                    /* value = bytes of tiles 0..t (inclusive prefix) */
#define TR_LOOK_STRIDE 8u                /* u64 words per tile (five used: one 64-byte line per tile) */
#define TR_LOOK_HDR 8u                   /* u64 words in front of the granules: word 0 = the ticket counter */

struct RowsCaps { uint64_t heap[5]; int64_t rows; };     // heap order: QNAME, CIGAR, SEQ, QUAL, READ_GROUP_ID
struct RowsStr { uint32_t *off[5]; uint8_t *heap[5]; uint32_t *alen_qual, *seq_chars; };

// everything unpack_one derives from a record, in registers
struct RowVals {
    RecInfo r; bool ok;
    uint32_t ql, cl, ls, lq, rl;         // reserved bytes in the five heaps
    uint32_t rg_rel; int32_t rgi; uint8_t rgv;
    bool star;                           // QUAL is "*" (no bases, or the first quality byte is 0xff)
};
template <class S, class D> __device__ __forceinline__ bool rows_eval(const BamStream &st, const S &s, const D &dict, uint64_t o, RowVals &v) {
    const int rcv = rec_check_t(st, s, o, v.r, true);
    v.ok = rcv == REC_OK; v.ql = v.cl = v.ls = v.lq = v.rl = 0; v.rg_rel = 0; v.rgi = -1; v.rgv = 0; v.star = true;
    if (!v.ok) { v.r.cig_off = o; v.r.n_cigar_eff = 0; return false; }
    const RecInfo &r = v.r;
    v.ql = (uint32_t)(find_nul_t(s, o + 36, o + 36 + r.l_qname) - (o + 36));
    uint32_t cl = 0;
    for (uint32_t j = 0; j < r.n_cigar_eff; j++) cl += ndigits(s.u32(r.cig_off + 4ull * j) >> 4) + 1;
    v.cl = r.n_cigar_eff ? cl : 1;
    const uint64_t seq = o + 36 + r.l_qname + 4ull * r.n_cigar;
    const uint64_t qual = seq + (((uint64_t)r.l_seq + 1) >> 1);
    v.ls = r.l_seq > 0 ? (st.seq_packed ? ((uint32_t)r.l_seq + 1u) >> 1 : (uint32_t)r.l_seq) : 1;
    v.star = !(r.l_seq > 0 && s.u8(qual) != 255);
    v.lq = v.star ? 1u : (uint32_t)r.l_seq;
    const uint64_t aux = qual + (uint64_t)r.l_seq, end = o + 4ull + r.block_len;
    bool bad = false; const uint64_t rg = st.want_rg ? aux_find_t(s, aux, end, 'R', 'G', &bad, r.cg_beg, r.cg_end) : NONE64;
    if (rg != NONE64 && (s.u8(rg) == 'Z' || s.u8(rg) == 'H')) {
        v.rgv = 1;
        const uint32_t rl = (uint32_t)(find_nul_t(s, rg + 1, end) - (rg + 1));
        for (int32_t q = 0; q < dict.n; q++) {
            const uint32_t a = dict.off[q], b = dict.off[q + 1];
            if (b - a != rl) continue;
            bool eq = true;
            for (uint32_t j = 0; j < rl; j++) if (dict.bytes[a + j] != s.u8(rg + 1 + j)) { eq = false; break; }
            if (eq) { v.rgi = q; break; }
        }
        v.rl = rl; v.rg_rel = (uint32_t)(rg + 1 - o);
    }
    return true;
}
__device__ __forceinline__ void rows_store(int64_t row, uint64_t o, const RowVals &v, uint32_t *rec_off, uint8_t *rg_flag, const BamCols &c) {
    rec_off[row] = (uint32_t)o;
    c.len_qname[row] = v.ql; c.len_cigar[row] = v.cl; c.len_seq[row] = v.ls; c.len_qual[row] = v.lq; c.len_rg[row] = v.rl;
    c.cig_rel[row] = (uint32_t)(v.r.cig_off - o); c.ncig_eff[row] = v.r.n_cigar_eff; c.rg_rel[row] = v.rg_rel; c.rg_idx[row] = v.rgi; rg_flag[row] = v.rgv;
    if (!v.ok) return;                  // (a row that fails validation ends the scan: its slots are defined, its values are never read)
    c.flag[row] = (uint16_t)v.r.flag;
    c.pos[row] = (int64_t)v.r.pos + 1;
    c.mapq[row] = (int32_t)v.r.mapq;
    c.pnext[row] = (int64_t)v.r.mpos + 1;
    c.tlen[row] = (int64_t)v.r.tlen;
    c.tid[row] = v.r.tid; c.mtid[row] = v.r.mtid;
}

__device__ __forceinline__ unsigned long long tr_load(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void tr_store(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// res (u64 words): [0] rows [1] carry [2] error [3] first error tile (bam_tile_finalize); [4] first invalid row (atomicMin);
//                  [8..12] heap totals, [13] offset of the batch's first record, [14] TR_F_* flags
// strings == 0: no string column is projected: no heap offsets, no look-back.
//
// PERSISTENT, one wave per workgroup, software-pipelined over tiles: while the wave works on tile t out of LDS, the bytes of its next tile
// (9 x 16 bytes per lane), that tile's row of the tile table and its record starts are already on their way into registers, and the ticket
// after that has been asked for.  A tile's life is a chain of latencies (stage, unpack, look-back, write); what the pipeline buys is that a
// compute unit has ~100 KB of loads in flight all the time instead of 9 KiB per tile during a sixth of its life.
// -DTR_DIAG (tools/dbg/time_rows.py with a diagnostic library): cycles per part of a tile's life, summed by lane 0
#ifdef TR_DIAG
__device__ unsigned long long g_tr_diag[16];   // 0 arrive+commit 1 record list 2 phase one 3 publish+ticket 4 look-back 5 prefetch issue 6 phase two 7 tiles 8 look-back windows 9 polls
#define TRD_T(v) const unsigned long long v = clock64()
#define TRD_ADD(i, a, b) do { trd[i] += (b) - (a); } while (0)
#define TRD_CNT(i, n) do { trd[i] += (n); } while (0)
#else
#define TRD_T(v) do { } while (0)
#define TRD_ADD(i, a, b) do { } while (0)
#define TRD_CNT(i, n) do { } while (0)
#endif
#define TR_STAGE_V ((TL_TILE + TL_HALO + 16u + 1023u) / 1024u)          /* 16-byte pieces per lane */
struct TrMeta { uint64_t first, recs_first; uint32_t count, rowbase; };
extern "C" __global__ void __launch_bounds__(64)
bam_tile_rows(BamStream st, BamDict dict, int64_t ntiles, TileOut out, const uint32_t *__restrict__ rowbase, unsigned long long *res, RowsCaps caps,
              uint32_t *__restrict__ rec_off, uint8_t *__restrict__ rg_flag, BamCols c, RowsStr s, uint32_t colmask, int strings,
              const uint16_t *__restrict__ tile_recs, const uint64_t *__restrict__ tile_recs_first, unsigned long long *look) {
    const bool w_qn = colmask & (1u << 0), w_cig = colmask & (1u << 5), w_seq = colmask & (1u << 9), w_qual = colmask & (1u << 10), w_rg = colmask & (1u << 11);
    __shared__ __attribute__((aligned(16))) uint8_t buf[TR_STAGE_V * 1024u];
    __shared__ uint32_t recs[TL_RECS];
    __shared__ uint32_t rg_lo[RGL_N + 1];
    __shared__ uint8_t rg_lb[RGL_BYTES];
    __shared__ uint32_t r_seq[64], r_lseq[64], r_oseq[64], r_oqual[64], r_c0[64], r_nul[64];
    __shared__ uint8_t cmap[TS_MAPN];
    const int lane = threadIdx.x;
    const int64_t nrows = (int64_t)res[0];
    const uint64_t e_tile = res[3];
    const bool dict_lds = st.want_rg && dict.n_rg > 0 && dict.n_rg <= RGL_N && dict.n_bytes <= RGL_BYTES;
    if (dict_lds) {
        for (int k = lane; k <= dict.n_rg; k += 64) rg_lo[k] = dict.rg_off[k];
        for (int k = lane; k < dict.n_bytes; k += 64) rg_lb[k] = dict.rg_bytes[k];
    }
    DictG dg; dg.off = dict.rg_off; dg.bytes = dict.rg_bytes; dg.n = dict.n_rg;
    DictL dl; dl.off = rg_lo; dl.bytes = rg_lb; dl.n = dict.n_rg;
    GSrc gs; gs.g = st.u;
    uint32_t *ticket = (uint32_t *)look;
    // tickets: t = the tile in LDS, t_n = the tile on its way, tk_nn = the ticket after that (lane 0, in flight)
    uint32_t tk0 = 0, tk1 = 0, tk_nn = 0;
    if (lane == 0) { tk0 = atomicAdd(ticket, 1u); tk1 = atomicAdd(ticket, 1u); tk_nn = atomicAdd(ticket, 1u); }
    int64_t t = (int64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)tk0), t_n = (int64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)tk1);
    uint4 pv[TR_STAGE_V]; TrMeta pm; uint32_t prec = 0;
    // issue the loads of tile `tt` (bytes, table row, first 64 record starts); nothing here waits
    auto prefetch = [&](int64_t tt) {
        pm.first = NONE64; pm.recs_first = 0; pm.count = 0; pm.rowbase = 0;
#pragma unroll
        for (uint32_t q = 0; q < TR_STAGE_V; q++) pv[q] = make_uint4(0, 0, 0, 0);
        if (tt >= ntiles) return;
        const uint64_t tb_ = (uint64_t)tt * TL_TILE, left = st.ulen - tb_;
        const uint32_t avail = left < (uint64_t)(TL_TILE + TL_HALO) ? (uint32_t)left : (TL_TILE + TL_HALO), pad = (avail + 15u) & ~15u;
        if ((uint64_t)tt <= e_tile) { pm.first = out.first[tt]; pm.count = out.count[tt]; pm.rowbase = rowbase[tt]; pm.recs_first = tile_recs_first[tt]; }
#pragma unroll
        for (uint32_t q = 0; q < TR_STAGE_V; q++) { const uint32_t k = q * 1024u + (uint32_t)lane * 16u; if (k < pad) pv[q] = *(const uint4 *)(st.u + tb_ + k); }
        prec = tile_recs[(size_t)tt * TL_RECS + (uint32_t)lane];
    };
    prefetch(t);
#ifdef TR_DIAG
    unsigned long long trd[16]; for (int q_ = 0; q_ < 16; q_++) trd[q_] = 0;
#endif
    while (t < ntiles) {
        TRD_T(t_a0);
        // ---- the tile's bytes, table row and record starts arrive; the next tile's are requested further down ----
        const uint64_t tb = (uint64_t)t * TL_TILE;
        uint32_t ss_len; { const uint64_t left = st.ulen - tb; ss_len = left < (uint64_t)(TL_TILE + TL_HALO) ? (uint32_t)left : (TL_TILE + TL_HALO); }
#pragma unroll
        for (uint32_t q = 0; q < TR_STAGE_V; q++) *(uint4 *)(buf + q * 1024u + (uint32_t)lane * 16u) = pv[q];
        uint32_t n = pm.count; uint64_t first = pm.first; int64_t row0 = 0;
        if (first == NONE64) n = 0;
        if (n) { row0 = pm.rowbase; if (row0 >= nrows) n = 0; else if (row0 + (int64_t)n > nrows) n = (uint32_t)(nrows - row0); }
        const bool walked = pm.recs_first == first;
        recs[lane] = (uint32_t)tb + prec;
        __syncthreads();
        TRD_T(t_a1); TRD_ADD(0, t_a0, t_a1); TRD_CNT(7, 1);
        unsigned long long *gran = look + TR_LOOK_HDR + (size_t)t * TR_LOOK_STRIDE;
        const bool rows_fit = row0 + (int64_t)n <= caps.rows;
        if (n && !rows_fit && lane == 0) atomicOr(res + 14, TR_F_ROWS);
        PSrc ls; ls.l = buf; ls.base = tb;
        if (n) {
            if (!walked) {     // the tile was re-walked by a repair round: walk again (the chain was validated by the scan / fix kernels)
                uint64_t o = first;
                for (uint32_t k = 0; k < n; k++) { if (lane == 0 && k < TL_RECS) recs[k] = (uint32_t)o; o += 4ull + (((o - tb) + 4 <= (uint64_t)ss_len) ? ls.u32(o) : gs.u32(o)); }
            } else if (n > 64) {
                for (uint32_t k = 64u + (uint32_t)lane; k < n && k < TL_RECS; k += 64) recs[k] = (uint32_t)(tb + tile_recs[(size_t)t * TL_RECS + k]);
            }
            __syncthreads();
        }
        TRD_T(t_a2); TRD_ADD(1, t_a1, t_a2);
        // ---- phase 1: one lane per record -- validate, unpack, fixed-width columns, reserved heap bytes ----
        uint32_t agg[5] = {0, 0, 0, 0, 0};
        RowVals v0; v0.ok = false; v0.ql = v0.cl = v0.ls = v0.lq = v0.rl = 0; v0.rg_rel = 0; v0.rgi = -1; v0.rgv = 0; v0.star = true;
        v0.r.cig_off = 0; v0.r.n_cigar_eff = 0; v0.r.l_seq = 0; v0.r.l_qname = 0; v0.r.n_cigar = 0;
        uint64_t o0 = 0;
        for (uint32_t g0 = 0; g0 < n; g0 += 64) {
            const uint32_t k = g0 + (uint32_t)lane;
            if (k < n) {
                const int64_t row = row0 + k;
                uint64_t o;
                if (k < TL_RECS) o = recs[k];
                else { o = first; for (uint32_t j = 0; j < k; j++) o += 4ull + gs.u32(o); }    // > 256 records per tile cannot happen (36-byte minimum)
                const uint64_t rel = o - tb;
                bool fast = false;
                if (rel + 4 <= (uint64_t)ss_len) { const uint32_t bl = ls.u32(o); fast = (bl >= 32u) && (rel + 4ull + bl <= (uint64_t)ss_len); }
                RowVals v;
                const bool good = fast ? (dict_lds ? rows_eval(st, ls, dl, o, v) : rows_eval(st, ls, dg, o, v)) : rows_eval(st, gs, dg, o, v);
                if (!good) atomicMin(res + 4, (unsigned long long)row);
                if (rows_fit) rows_store(row, o, v, rec_off, rg_flag, c);
                if (row == 0) res[13] = (unsigned long long)o;
                agg[0] += v.ql; agg[1] += v.cl; agg[2] += v.ls; agg[3] += v.lq; agg[4] += v.rl;
                if (g0 == 0) { v0 = v; o0 = o; }
            }
        }
        TRD_T(t_a3); TRD_ADD(2, t_a2, t_a3);
        uint32_t tot[5] = {0, 0, 0, 0, 0};
        if (strings) {
#pragma unroll
            for (int h = 0; h < 5; h++) { const uint32_t i_ = wave_incl_scan(agg[h], lane); tot[h] = RDLANE(i_, 63); }
        }
        const uint32_t mine = lane == 0 ? tot[0] : lane == 1 ? tot[1] : lane == 2 ? tot[2] : lane == 3 ? tot[3] : tot[4];
        if (strings && t != 0 && !(strings & 2) && lane < 5) tr_store(gran + lane, ((unsigned long long)mine << 32) | TR_ST_AGG);
        // ---- the next tile: its ticket has arrived; ask for its bytes and for the ticket after it ----
        const int64_t t_cur = t;
        {
            uint32_t nn = 0;
            if (lane == 0) { nn = tk_nn; tk_nn = atomicAdd(ticket, 1u); }
            const int64_t t_nn = (int64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)nn);
            t = t_n; t_n = t_nn;
        }
        // (the tile's own LDS image is still needed by phase 2: the next tile's bytes stay in registers until the top of the loop)
        TRD_T(t_a4); TRD_ADD(3, t_a3, t_a4);
        // ---- publish the tile's heap bytes, look back for its place ----
        uint32_t base[5] = {0, 0, 0, 0, 0};
        bool skip = !rows_fit;
        if (strings) {
            if (t_cur != 0 && !(strings & 2)) {
                // lane L < 60: heap h = L / 12, predecessor j = L % 12 of the current window
                const uint32_t h = (uint32_t)lane / TR_LOOK_W, j = (uint32_t)lane % TR_LOOK_W;
                const bool lk = lane < (int)(5u * TR_LOOK_W);
                uint32_t run = 0;                                        // (per lane: the sum of its heap, kept identical in the heap's 12 lanes)
                bool done = !lk;
                int64_t wtop = t_cur - 1;                               // the window's first (nearest) tile
                uint32_t polls = 0; bool timeout = false;
                for (;;) {
                    const int64_t tj = wtop - (int64_t)j;
                    unsigned long long g = ((unsigned long long)0 << 32) | TR_ST_INCL;       // tiles in front of tile 0: inclusive prefix 0
                    if (!done && tj >= 0) g = tr_load(look + TR_LOOK_HDR + (size_t)tj * TR_LOOK_STRIDE + h);
                    const uint32_t stt = (uint32_t)g & 3u, val = (uint32_t)(g >> 32);
                    const unsigned long long mP = __ballot(!done && stt == TR_ST_INCL), mE = __ballot(!done && stt == TR_ST_EMPTY);
                    const uint32_t bP = (uint32_t)(mP >> (TR_LOOK_W * h)) & ((1u << TR_LOOK_W) - 1u), bE = (uint32_t)(mE >> (TR_LOOK_W * h)) & ((1u << TR_LOOK_W) - 1u);
                    const uint32_t firstP = bP ? (uint32_t)__ffs((int)bP) - 1u : TR_LOOK_W;        // nearest predecessor that knows its inclusive prefix
                    const uint32_t need = firstP < TR_LOOK_W ? (2u << firstP) - 1u : (1u << TR_LOOK_W) - 1u;
                    const bool wait = !done && (bE & need) != 0u;         // a tile this heap needs has not published yet
                    if (__ballot(wait)) {
                        TRD_CNT(9, 1);
                        if (++polls > (1u << 22)) { timeout = true; break; }
                        __builtin_amdgcn_s_sleep(2);
                        continue;
                    }
                    TRD_CNT(8, 1);
                    const uint32_t contrib = (!done && j <= firstP) ? val : 0u;
                    const uint32_t inc = wave_incl_scan(contrib, lane);
                    // the heap's sum over its 12 lanes = inc[12 h + 11] - inc[12 h - 1]
                    uint32_t hs = 0;
#pragma unroll
                    for (uint32_t q = 0; q < 5u; q++) {
                        const uint32_t hi = RDLANE(inc, TR_LOOK_W * q + TR_LOOK_W - 1u), lo = q ? RDLANE(inc, TR_LOOK_W * q - 1u) : 0u;
                        if (h == q) hs = hi - lo;
                    }
                    if (!done) { run += hs; if (firstP < TR_LOOK_W) done = true; }
                    if (__ballot(!done) == 0ull) break;
                    wtop -= (int64_t)TR_LOOK_W;
                }
                if (timeout) { if (lane == 0) atomicOr(res + 14, TR_F_TIMEOUT); run = 0; }
#pragma unroll
                for (uint32_t q = 0; q < 5u; q++) base[q] = RDLANE(run, TR_LOOK_W * q);     // lane 12 h holds heap h's exclusive prefix
            }
            const uint32_t myb = lane == 0 ? base[0] : lane == 1 ? base[1] : lane == 2 ? base[2] : lane == 3 ? base[3] : base[4];
            if (lane < 5) tr_store(gran + lane, ((unsigned long long)(myb + mine) << 32) | TR_ST_INCL);
            // the batch's totals (the last tile knows them) and the arena guard
            const bool proj = lane == 0 ? w_qn : lane == 1 ? w_cig : lane == 2 ? w_seq : lane == 3 ? w_qual : w_rg;
            const unsigned long long capv = lane == 0 ? caps.heap[0] : lane == 1 ? caps.heap[1] : lane == 2 ? caps.heap[2] : lane == 3 ? caps.heap[3] : caps.heap[4];
            const unsigned long long mo = __ballot(lane < 5 && proj && (unsigned long long)myb + mine > capv);
            if (t_cur == ntiles - 1 && lane < 5) res[8 + lane] = (unsigned long long)myb + mine;
            if (mo) { skip = true; if (lane == 0) atomicOr(res + 14, TR_F_HEAP); }
        }
        TRD_T(t_a5); TRD_ADD(4, t_a4, t_a5);
        // the next tile's loads go out now: they travel while this tile's strings are written (nothing below waits for a load)
        prefetch(t);
        TRD_T(t_a6); TRD_ADD(5, t_a5, t_a6);
        if (!strings || n == 0 || skip) { __syncthreads(); continue; }
        // ---- phase 2: heap offsets of the rows, the strings ----
        TsSrc src; src.l = buf; src.g = st.u; src.base = tb; src.len = ss_len;
        uint32_t runb[5];
#pragma unroll
        for (int h = 0; h < 5; h++) runb[h] = base[h];
        for (uint32_t g0 = 0; g0 < n; g0 += 64) {
            const uint32_t k = g0 + (uint32_t)lane;
            const bool act = k < n;
            const int64_t d = row0 + k;
            RowVals v = v0; uint64_t o = o0;
            if (g0 != 0) {
                v.ok = false; v.ql = v.cl = v.ls = v.lq = v.rl = 0; v.star = true;
                if (act) {
                    if (k < TL_RECS) o = recs[k];
                    else { o = first; for (uint32_t j = 0; j < k; j++) o += 4ull + gs.u32(o); }
                    const uint64_t rel = o - tb;
                    bool fast = false;
                    if (rel + 4 <= (uint64_t)ss_len) { const uint32_t bl = ls.u32(o); fast = (bl >= 32u) && (rel + 4ull + bl <= (uint64_t)ss_len); }
                    (void)(fast ? (dict_lds ? rows_eval(st, ls, dl, o, v) : rows_eval(st, ls, dg, o, v)) : rows_eval(st, gs, dg, o, v));
                }
            }
            const bool ok = act && v.ls != 0;                            // rows that failed validation reserve nothing
            // heap offsets: exclusive sums over the group's lanes behind the tile's running place
            const uint32_t lq_ = act ? v.ql : 0u, lc_ = act ? v.cl : 0u, ls_ = act ? v.ls : 0u, lu_ = act ? v.lq : 0u, lr_ = act ? v.rl : 0u;
            const uint32_t i0 = wave_incl_scan(lq_, lane), i1 = wave_incl_scan(lc_, lane), i2 = wave_incl_scan(ls_, lane), i3 = wave_incl_scan(lu_, lane), i4 = wave_incl_scan(lr_, lane);
            const uint32_t off_qn = runb[0] + i0 - lq_, off_cig = runb[1] + i1 - lc_, off_seq = runb[2] + i2 - ls_, off_qual = runb[3] + i3 - lu_, off_rg = runb[4] + i4 - lr_;
            runb[0] += RDLANE(i0, 63); runb[1] += RDLANE(i1, 63); runb[2] += RDLANE(i2, 63); runb[3] += RDLANE(i3, 63); runb[4] += RDLANE(i4, 63);
            if (act) {
                s.off[0][d] = off_qn; s.off[1][d] = off_cig; s.off[2][d] = off_seq; s.off[3][d] = off_qual; s.off[4][d] = off_rg;
                if (d + 1 == nrows) { s.off[0][nrows] = off_qn + lq_; s.off[1][nrows] = off_cig + lc_; s.off[2][nrows] = off_seq + ls_; s.off[3][nrows] = off_qual + lu_; s.off[4][nrows] = off_rg + lr_; }
            }
            const uint32_t l_seq = (ok && v.r.l_seq > 0) ? (uint32_t)v.r.l_seq : 0u;
            const uint64_t seq = o + 36 + v.r.l_qname + 4ull * v.r.n_cigar;
            const bool star = !ok || v.star;
            const bool lng = l_seq > TS_LONG;
            const uint32_t nch = (ok && !lng && (w_seq || w_qual)) ? (l_seq + 15u) >> 4 : 0u;
            const uint32_t cinc = wave_incl_scan(nch, lane);
            const uint32_t c0 = cinc - nch;
            const uint32_t T = RDLANE(cinc, 63);
            r_seq[lane] = (uint32_t)(seq - tb); r_lseq[lane] = star ? (l_seq | 0x80000000u) : l_seq; r_oseq[lane] = off_seq; r_oqual[lane] = off_qual;
            r_c0[lane] = c0; r_nul[lane] = 0xffffffffu;
            for (uint32_t q = 0; q < nch; q++) cmap[c0 + q] = (uint8_t)lane;
            __syncthreads();
            // ---- SEQ / QUAL by chunk ----
            for (uint32_t cb = 0; cb < T; cb += 64) {
                const uint32_t ch = cb + (uint32_t)lane;
                if (ch < T) {
                    const uint32_t j = cmap[ch], q = ch - r_c0[j], lsq = r_lseq[j], lq = lsq & 0x7fffffffu;
                    const uint64_t sq = tb + r_seq[j];
                    if (w_seq) ts_seq_chunk(src, sq, lq, q, s.heap[2] + r_oseq[j], st.seq_packed);
                    if (w_qual && !(lsq >> 31)) {
                        const uint32_t fz = ts_qual_chunk(src, sq + (((uint64_t)lq + 1) >> 1), lq, q, s.heap[3] + r_oqual[j]);
                        if (fz != 0xffffffffu) atomicMin(&r_nul[j], fz);
                    }
                }
            }
            // ---- per-row pieces ----
            if (ok) {
                if (w_seq && l_seq == 0) s.heap[2][off_seq] = st.seq_packed ? 0 : '*';
                if (w_qual && star) s.heap[3][off_qual] = '*';
                for (uint32_t b = 0; w_qn && b < v.ql; b += 16) {
                    const uint64_t a = src.u64(o + 36 + b), e = src.u64(o + 36 + b + 8);
                    const uint32_t w[4] = {(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)e, (uint32_t)(e >> 32)};
                    store_n16(s.heap[0] + off_qn + b, w, v.ql - b);
                }
                for (uint32_t b = 0; w_rg && b < v.rl; b += 16) {
                    const uint64_t a = src.u64(o + v.rg_rel + b), e = src.u64(o + v.rg_rel + b + 8);
                    const uint32_t w[4] = {(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)e, (uint32_t)(e >> 32)};
                    store_n16(s.heap[4] + off_rg + b, w, v.rl - b);
                }
                uint8_t *dc = s.heap[1] + off_cig;
                const uint32_t ne = v.r.n_cigar_eff;
                if (w_cig && ne == 0) dc[0] = '*';
                uint32_t p = 0;
                for (uint32_t q = 0; w_cig && q < ne; q++) {
                    const uint32_t op = src.u32(v.r.cig_off + 4ull * q);
                    uint32_t ol = op >> 4; const uint32_t nd = ndigits(ol);
                    for (uint32_t z = 0; z < nd; z++) { dc[p + nd - 1 - z] = (uint8_t)('0' + ol % 10); ol /= 10; }
                    // sam.h:112 BAM_CIGAR_STR "MIDNSHP=XB", '?' beyond; byte gather with v_perm_b32
                    const uint32_t C0 = 0x4e44494du, C1 = 0x3d504853u, C2 = 0x3f3f4258u, C3 = 0x3f3f3f3fu;
                    const uint32_t oc = op & 0xf, selb = (oc & 7u) | 0x0c0c0c00u;
                    dc[p + nd] = (uint8_t)((oc & 8u) ? __builtin_amdgcn_perm(C3, C2, selb) : __builtin_amdgcn_perm(C1, C0, selb));
                    p += nd + 1;
                }
            }
            // ---- long fields: the wave streams one row at a time ----
            {
                uint64_t LM = __ballot(ok && lng && (w_seq || w_qual));
                while (LM) {
                    const int i = __ffsll((unsigned long long)LM) - 1; LM &= LM - 1;
                    const uint32_t lq = RDLANE(l_seq, i), st_i = RDLANE((uint32_t)star, i), os = RDLANE(off_seq, i), oq = RDLANE(off_qual, i);
                    const uint64_t sq = ((uint64_t)RDLANE((uint32_t)(seq >> 32), i) << 32) | RDLANE((uint32_t)seq, i);
                    uint32_t fzm = 0xffffffffu;
                    for (uint32_t q = lane; q < (lq + 15u) >> 4; q += 64) {
                        if (w_seq) ts_seq_chunk(src, sq, lq, q, s.heap[2] + os, st.seq_packed);
                        if (w_qual && !st_i) { const uint32_t fz = ts_qual_chunk(src, sq + (((uint64_t)lq + 1) >> 1), lq, q, s.heap[3] + oq); fzm = fz < fzm ? fz : fzm; }
                    }
                    if (fzm != 0xffffffffu) atomicMin(&r_nul[i], fzm);
                }
            }
            __syncthreads();
            if (ok && w_qual) { const uint32_t fz = r_nul[lane]; s.alen_qual[d] = star ? 1u : (fz != 0xffffffffu ? fz : l_seq); }
            if (act && w_seq && s.seq_chars) s.seq_chars[d] = ok ? l_seq : 0u;
            __syncthreads();
        }
        TRD_T(t_a7); TRD_ADD(6, t_a6, t_a7);
    }
#ifdef TR_DIAG
    if (lane == 0) for (int q_ = 0; q_ < 16; q_++) if (trd[q_]) atomicAdd(&g_tr_diag[q_], trd[q_]);
#endif
}

// validity words of READ_GROUP_ID for the rows the device counted (the host does not know the row count when this is queued)
extern "C" __global__ void __launch_bounds__(256)
bam_pack_validity_dev(const uint8_t *flag, const unsigned long long *res, int64_t row_cap, uint64_t *words) {
    int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t nrows = (int64_t)res[0]; if (nrows > row_cap) nrows = row_cap;
    bool v = row < nrows && flag[row] != 0;
    uint64_t m = __ballot(v);
    if ((threadIdx.x & 63) == 0 && row < nrows) words[row >> 6] = m;
}
