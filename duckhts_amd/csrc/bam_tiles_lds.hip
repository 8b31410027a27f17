// bam_tiles_lds.hip -- LDS-staged tile kernels for the BAM record stage (gfx950).
//
// Same exact algorithm as bam_records.hip (speculate the first record of every 8 KiB tile with bam_read1's predicates,
// walk the chain, prove continuity tile-to-tile, unpack the 13 core columns), but one WAVE owns one tile: the tile (+1 KiB
// halo) is staged into LDS with 16-byte coalesced loads, the 64 lanes test 64 candidate offsets at once, the chain walk and
// every per-record byte access (QNAME NUL scan, CIGAR ops, aux tag walk) run out of LDS instead of issuing one global load
// per field per lane.  Bytes beyond the staged window (long records) fall back to global memory through the accessor.
//   replaces: htslib sam.c:779-855 bam_read1, 675-730 bam_tag2cigar, 4124-4134 sam_read1_bam, 4785-4855 aux walk;
//             src/bam_reader.c:785-918 fixed-width writers and string-length bookkeeping.
#pragma once

#ifndef TL_TILE
#define TL_TILE 8192u
#endif
#define TL_RECS (TL_TILE / 32u)            /* >= records per tile (36-byte minimum for BAM, 32 for BCF) */
#define TL_HALO 1024u
// flags in res[14] of a batch whose rows are queued without the host in between (dhts_bam_scan.inc: rows_queue)
#define TR_F_ROWS 1ull                   /* the row arrays are too small for this batch */
#define TR_F_HEAP 2ull                   /* a string arena is too small */
#define TR_F_TIMEOUT 4ull                /* look-back wait expired (bam_tile_rows; internal error) */

// global-memory accessor (records that do not fit the staged window)
struct GSrc {
    const uint8_t *g;
    __device__ __forceinline__ uint8_t u8(uint64_t o) const { return g[o]; }
    __device__ __forceinline__ uint32_t u32(uint64_t o) const { return ldu32(g + o); }
    __device__ __forceinline__ uint64_t u64(uint64_t o) const { uint64_t v; __builtin_memcpy(&v, g + o, 8); return v; }   // buffers are padded
};
struct LSrc { const uint8_t *g; const uint8_t *l; uint64_t base; uint32_t len; };   // staged-window descriptor

// direct accessor into the staged window: every offset touched must lie inside it (checked by the caller once per record).
// Kept as (LDS pointer, 32-bit index) so the compiler emits ds_read, not flat loads.
struct PSrc {
    const uint8_t *l; uint64_t base;
    __device__ __forceinline__ uint8_t u8(uint64_t o) const { return l[(uint32_t)(o - base)]; }
    __device__ __forceinline__ uint32_t u32(uint64_t o) const { uint32_t v; __builtin_memcpy(&v, l + (uint32_t)(o - base), 4); return v; }
    __device__ __forceinline__ uint64_t u64(uint64_t o) const { uint64_t v; __builtin_memcpy(&v, l + (uint32_t)(o - base), 8); return v; }   // window is padded by 16 B
};

// first q in [p, end) with byte q == 0, or end: eight bytes per step (one LDS / global read instead of eight dependent ones).
// (v - 0x01..) & ~v & 0x80..: the LOWEST set bit marks the first zero byte exactly; bytes at or past `end` are ignored.
template <class S> __device__ __forceinline__ uint64_t find_nul_t(const S &s, uint64_t p, uint64_t end) {
    while (p < end) {
        const uint64_t v = s.u64(p), z = (v - 0x0101010101010101ull) & ~v & 0x8080808080808080ull;
        if (z) { const uint64_t q = p + (uint64_t)((__ffsll((unsigned long long)z) - 1) >> 3); return q < end ? q : end; }
        p += 8;
    }
    return end;
}

template <class S> __device__ __forceinline__ uint64_t aux_skip_t(const S &s, uint64_t p, uint64_t end) {
    if (p >= end) return end;
    uint8_t t = s.u8(p); ++p;
    if (t == 'Z' || t == 'H') {
        p = find_nul_t(s, p, end);
        return p < end ? p + 1 : end;
    }
    if (t == 'B') {
        if (end - p < 5) return NONE64;
        uint8_t sub = s.u8(p);
        int sz = aux_size(sub); if (sub == 'Z' || sub == 'H' || sub == 'B') sz = sub;
        ++p;
        uint64_t n = s.u32(p); p += 4;
        if (sz == 0 || end - p < (uint64_t)sz * n) return NONE64;
        return p + (uint64_t)sz * n;
    }
    int sz = aux_size(t);
    if (sz == 0) return NONE64;
    if (end - p < (uint64_t)sz) return NONE64;
    return p + sz;
}

// aux_skip_t for a field whose type byte (and the byte after it) have already been read
template <class S> __device__ __forceinline__ uint64_t aux_skip_known(const S &s, uint64_t p, uint64_t end, uint8_t t, uint8_t nxt) {
    if (p >= end) return end;
    ++p;
    if (t == 'Z' || t == 'H') {
        p = find_nul_t(s, p, end);
        return p < end ? p + 1 : end;
    }
    if (t == 'B') {
        if (end - p < 5) return NONE64;
        const uint8_t sub = nxt;
        int sz = aux_size(sub); if (sub == 'Z' || sub == 'H' || sub == 'B') sz = sub;
        ++p;
        uint64_t n = s.u32(p); p += 4;
        if (sz == 0 || end - p < (uint64_t)sz * n) return NONE64;
        return p + (uint64_t)sz * n;
    }
    int sz = aux_size(t);
    if (sz == 0) return NONE64;
    if (end - p < (uint64_t)sz) return NONE64;
    return p + sz;
}

// (one 4-byte read per field: tag, type and the byte after the type -- the walk is a chain of dependent LDS reads)
template <class S> __device__ __forceinline__ uint64_t aux_find_t(const S &s, uint64_t aux, uint64_t end, uint8_t t0, uint8_t t1, bool *bad, uint64_t skip_beg, uint64_t skip_end) {
    *bad = false;
    uint64_t eff_len = (end - aux) - (skip_end - skip_beg);
    if (eff_len <= 2) return NONE64;
    uint64_t p = aux;
    if (p == skip_beg) p = skip_end;
    p += 2;
    for (;;) {
        const uint32_t x = s.u32(p - 2);                      // tag[0] tag[1] type next   (reads <= 1 byte past `end`: buffers are padded)
        const uint8_t ty = (uint8_t)(x >> 16);
        const uint64_t e = aux_skip_known(s, p, end, ty, (uint8_t)(x >> 24));
        if ((uint8_t)x == t0 && (uint8_t)(x >> 8) == t1) {
            if (e == NONE64) { *bad = true; return NONE64; }
            if ((ty == 'Z' || ty == 'H') && s.u8(e - 1) != 0) { *bad = true; return NONE64; }
            return p;
        }
        uint64_t nx = e;
        if (nx == NONE64) { *bad = true; return NONE64; }
        if (nx == skip_beg) nx = skip_end;
        if (end - nx <= 2) return NONE64;
        p = nx + 2;
    }
}

// same predicates as rec_check (bam_records.hip), reading through the accessor
template <class S> __device__ __forceinline__ int rec_check_t(const BamStream &st, const S &s, uint64_t o, RecInfo &r, bool full) {
    if (st.ulen - o < 4) return REC_INCOMPLETE;
    int32_t block_len = (int32_t)s.u32(o);
    if (block_len < 32) return REC_INVALID;
    if (st.ulen - o - 4 < 32) return REC_INCOMPLETE;
    r.block_len = (uint32_t)block_len;
    r.tid = (int32_t)s.u32(o + 4); r.pos = (int32_t)s.u32(o + 8);
    uint32_t x2 = s.u32(o + 12), x3 = s.u32(o + 16);
    r.mapq = (x2 >> 8) & 0xff; r.l_qname = x2 & 0xff;
    r.flag = x3 >> 16; r.n_cigar = x3 & 0xffff;
    r.l_seq = (int32_t)s.u32(o + 20); r.mtid = (int32_t)s.u32(o + 24); r.mpos = (int32_t)s.u32(o + 28); r.tlen = (int32_t)s.u32(o + 32);
    uint64_t body = (uint64_t)r.block_len - 32;
    if (r.l_seq < 0 || r.l_qname < 1) return REC_INVALID;
    if (((uint64_t)r.n_cigar << 2) + r.l_qname + (((uint64_t)r.l_seq + 1) >> 1) + (uint64_t)r.l_seq > body) return REC_INVALID;
    if (!full) {
        // speculation filter (not part of exactness; a wrong rejection only costs a repair round).  The header-range and
        // aux-size tests come BEFORE the "body runs past the buffer" exit, so a hop that lands on garbage with a huge
        // block_len is rejected instead of being mistaken for the incomplete last record of the batch.
        if (r.tid < -1 || r.tid >= st.n_ref || r.mtid < -1 || r.mtid >= st.n_ref) return REC_INVALID;
        uint64_t core = ((uint64_t)r.n_cigar << 2) + r.l_qname + (((uint64_t)r.l_seq + 1) >> 1) + (uint64_t)r.l_seq;
        if (body - core > 8 * core + 65536) return REC_INVALID;
        if (st.ulen - o - 36 < body) return REC_INCOMPLETE;
        if (s.u8(o + 36 + r.l_qname - 1) != 0) return REC_INVALID;
        return REC_OK;
    }
    if (st.ulen - o - 36 < body) return REC_INCOMPLETE;
    r.cig_off = o + 36 + r.l_qname; r.n_cigar_eff = r.n_cigar; r.cg_beg = r.cg_end = 0;
    uint64_t end = o + 4 + r.block_len;
    uint64_t aux = r.cig_off + 4ull * r.n_cigar + (((uint64_t)r.l_seq + 1) >> 1) + (uint64_t)r.l_seq;
    if (r.n_cigar > 0 && s.u32(r.cig_off) == (4u | ((uint32_t)r.l_seq << 4)) && r.tid >= 0 && r.pos >= 0) {
        bool bad; uint64_t cg = aux_find_t(s, aux, end, 'C', 'G', &bad, 0, 0);
        if (cg == NONE64 && bad) return REC_INVALID;
        if (cg != NONE64 && s.u8(cg) == 'B' && (s.u8(cg + 1) == 'I' || s.u8(cg + 1) == 'i')) {
            uint32_t cgl = s.u32(cg + 2);
            if (cgl >= r.n_cigar && cgl < (1u << 29)) { r.cig_off = cg + 6; r.n_cigar_eff = cgl; r.cg_beg = cg - 2; r.cg_end = cg + 6 + 4ull * cgl; }
        }
    }
    if (r.n_cigar_eff > 0) {
        int64_t qlen = 0;
        for (uint32_t k = 0; k < r.n_cigar_eff; k++) { uint32_t c = s.u32(r.cig_off + 4ull * k); if (CIG_QUERY(c & 0xf)) qlen += c >> 4; }
        if (r.l_seq > 0 && !(r.flag & 4) && qlen != r.l_seq) return REC_INVALID;
    }
    if (r.tid >= st.n_ref || r.tid < -1 || r.mtid >= st.n_ref || r.mtid < -1) return REC_INVALID;
    return REC_OK;
}

// One hop of the record chain: the exact bam_read1 tests that need only the 36-byte core (sam.c:794, 820-823) plus the
// header range test (sam.c:4127-4131).  A walk that starts on garbage fails these almost immediately, which is what keeps
// a wrong speculation from propagating.  The CIGAR/qlen test and the CG swap run lane-parallel in bam_tile_unpack,
// which reports the first invalid row.
template <class S> __device__ __forceinline__ int rec_hop(const BamStream &st, const S &s, uint64_t o, uint32_t &bl) {
    if (st.ulen - o < 4) return REC_INCOMPLETE;
    const int32_t b = (int32_t)s.u32(o);
    if (b < 32) return REC_INVALID;
    if (st.ulen - o - 4 < 32) return REC_INCOMPLETE;
    const int32_t tid = (int32_t)s.u32(o + 4), mtid = (int32_t)s.u32(o + 24), l_seq = (int32_t)s.u32(o + 20);
    const uint32_t l_qname = s.u32(o + 12) & 0xff, n_cigar = s.u32(o + 16) & 0xffff;
    const uint64_t body = (uint64_t)(uint32_t)b - 32;
    if (l_seq < 0 || l_qname < 1) return REC_INVALID;
    if (((uint64_t)n_cigar << 2) + l_qname + (((uint64_t)l_seq + 1) >> 1) + (uint64_t)l_seq > body) return REC_INVALID;
    if (st.ulen - o - 36 < body) return REC_INCOMPLETE;
    if (tid >= st.n_ref || tid < -1 || mtid >= st.n_ref || mtid < -1) return REC_INVALID;
    bl = (uint32_t)b;
    return REC_OK;
}

// rec_hop for a record whose 36-byte core lies inside the staged window, evaluated wave-uniformly: ONE (unaligned) LDS read
// fetches the core as dwords across lanes 0..8, v_readlane moves the six fields into SGPRs and the tests run on the scalar unit.
// Same predicates, same order of outcomes as rec_hop.
__device__ __forceinline__ int rec_hop_uniform(const BamStream &st, const PSrc &ls, uint64_t o, int lane, uint32_t &bl) {
    if (st.ulen - o < 4) return REC_INCOMPLETE;
    const uint32_t v = ls.u32(o + 4u * (uint32_t)(lane & 15));
    const int32_t b = (int32_t)__builtin_amdgcn_readlane((int)v, 0);
    if (b < 32) return REC_INVALID;
    if (st.ulen - o - 4 < 32) return REC_INCOMPLETE;
    const int32_t tid = __builtin_amdgcn_readlane((int)v, 1), l_seq = __builtin_amdgcn_readlane((int)v, 5), mtid = __builtin_amdgcn_readlane((int)v, 6);
    const uint32_t l_qname = (uint32_t)__builtin_amdgcn_readlane((int)v, 3) & 0xff, n_cigar = (uint32_t)__builtin_amdgcn_readlane((int)v, 4) & 0xffff;
    const uint64_t body = (uint64_t)(uint32_t)b - 32;
    if (l_seq < 0 || l_qname < 1) return REC_INVALID;
    if (((uint64_t)n_cigar << 2) + l_qname + (((uint64_t)l_seq + 1) >> 1) + (uint64_t)l_seq > body) return REC_INVALID;
    if (st.ulen - o - 36 < body) return REC_INCOMPLETE;
    if (tid >= st.n_ref || tid < -1 || mtid >= st.n_ref || mtid < -1) return REC_INVALID;
    bl = (uint32_t)b;
    return REC_OK;
}

// speculation filter on a candidate inside the staged window: the core dwords are read up front (no branch between the LDS
// reads), then the same tests as rec_check_t(full = false).  Returns REC_OK / REC_INVALID / REC_INCOMPLETE and the block length.
__device__ __forceinline__ int spec_check_lds(const BamStream &st, const PSrc &ls, uint64_t o, uint32_t &bl) {
    const uint32_t w0 = ls.u32(o), w1 = ls.u32(o + 4), w3 = ls.u32(o + 12), w4 = ls.u32(o + 16), w5 = ls.u32(o + 20), w6 = ls.u32(o + 24);
    if (st.ulen - o < 4) return REC_INCOMPLETE;
    const int32_t block_len = (int32_t)w0;
    if (block_len < 32) return REC_INVALID;
    if (st.ulen - o - 4 < 32) return REC_INCOMPLETE;
    const int32_t tid = (int32_t)w1, l_seq = (int32_t)w5, mtid = (int32_t)w6;
    const uint32_t l_qname = w3 & 0xff, n_cigar = w4 & 0xffff;
    const uint64_t body = (uint64_t)(uint32_t)block_len - 32;
    if (l_seq < 0 || l_qname < 1) return REC_INVALID;
    const uint64_t core = ((uint64_t)n_cigar << 2) + l_qname + (((uint64_t)l_seq + 1) >> 1) + (uint64_t)l_seq;
    if (core > body) return REC_INVALID;
    if (tid < -1 || tid >= st.n_ref || mtid < -1 || mtid >= st.n_ref) return REC_INVALID;
    if (body - core > 8 * core + 65536) return REC_INVALID;
    if (st.ulen - o - 36 < body) return REC_INCOMPLETE;
    if (ls.u8(o + 36 + l_qname - 1) != 0) return REC_INVALID;
    bl = (uint32_t)block_len;
    return REC_OK;
}

__device__ __forceinline__ void tile_stage(const BamStream &st, uint64_t tb, uint8_t *buf, LSrc &s, int lane, uint32_t nthr = 64u) {
    uint64_t left = st.ulen - tb;
    uint32_t avail = left < (uint64_t)(TL_TILE + TL_HALO) ? (uint32_t)left : (TL_TILE + TL_HALO);
    uint32_t pad = (avail + 15u) & ~15u;                      // the inflated buffer carries >= 256 bytes of padding
    for (uint32_t k = (uint32_t)lane * 16u; k < pad; k += nthr * 16u) { uint4 v = *(const uint4 *)(st.u + tb + k); *(uint4 *)(buf + k) = v; }
    s.g = st.u; s.l = buf; s.base = tb; s.len = avail;
    __syncthreads();
}

// One wave per tile: speculate the first record start (64 candidates per step), walk the chain.
extern "C" __global__ void __launch_bounds__(64)
bam_tile_scan(BamStream st, uint64_t start0, int64_t ntiles, TileOut out, uint16_t *tile_recs, uint64_t *tile_recs_first, uint64_t spec_from) {
    __shared__ __attribute__((aligned(16))) uint8_t buf[TL_TILE + TL_HALO];
    __shared__ uint16_t rl[TL_RECS];                          // record starts found by the walk, relative to the tile
    const int lane = threadIdx.x;
    const int64_t t = blockIdx.x;
    if (t >= ntiles) return;
    const uint64_t tb = (uint64_t)t * TL_TILE;
    uint64_t te = tb + TL_TILE; if (te > st.ulen) te = st.ulen;
    LSrc s; tile_stage(st, tb, buf, s, lane);
    PSrc ls; ls.l = buf; ls.base = tb;                        // valid for offsets inside the staged window only
    GSrc gs; gs.g = st.u;
    uint64_t first = NONE64;
    if (t == 0 && start0 != NONE64) first = start0;
    else {
        const uint64_t lim = (t == 0) ? st.ulen : te;         // a shard that begins mid-stream keeps looking past its first tile
        const uint64_t from = (t == 0) ? spec_from : 0;        // (a retry after a false start resumes behind the failed candidate)
        for (uint64_t base = (t == 0) ? (from & ~63ull) : tb; base < lim; base += 64) {
            const uint64_t o = base + (uint64_t)lane;
            bool ok = false;
            if (o < lim && o >= from) {
                RecInfo r; uint32_t bl0 = 0;
                // the cheap test touches <= 36 + 255 bytes: straight LDS when that lies inside the staged window
                const bool inw = (o - tb) + 300u <= (uint64_t)s.len;
                int rc0;
                if (inw) rc0 = spec_check_lds(st, ls, o, bl0); else { rc0 = rec_check_t(st, gs, o, r, false); bl0 = r.block_len; }
                if (rc0 == REC_OK) {
                    ok = true;
                    uint64_t o2 = o + 4ull + bl0;
                    for (int k = 0; k < 2 && ok; k++) {
                        RecInfo r2; uint32_t bl2 = 0;
                        const bool inw2 = (o2 >= tb) && (o2 - tb) + 300u <= (uint64_t)s.len;
                        int rc;
                        if (inw2) rc = spec_check_lds(st, ls, o2, bl2); else { rc = rec_check_t(st, gs, o2, r2, false); bl2 = r2.block_len; }
                        if (rc == REC_INVALID) ok = false;
                        else if (rc == REC_INCOMPLETE) break;
                        else o2 += 4ull + bl2;
                    }
                }
            }
            const uint64_t m = __ballot(ok);
            if (m) { first = base + (uint64_t)(__ffsll((unsigned long long)m) - 1); break; }
        }
    }
    // chain walk: wave-uniform (every lane computes the same values; LDS reads broadcast)
    uint64_t en = NONE64; uint32_t cnt = 0; int err = 0;
    if (first != NONE64 && first < te) {
        uint64_t o = first;
        // Inside the staged window the walk runs on 32-bit offsets relative to the tile (a batch is < 2^32 bytes: ulen - tb fits; a hop
        // is 4 + block_len < 2^31 + 4): the scalar unit has no 64-bit compare, so the 64-bit form of rec_hop_uniform spent a third of its
        // ~75 instructions per hop moving scalars into vector registers for v_cmp_u64 (1,794 SALU + 1,019 VALU per tile, profiles/r03).
        // Same predicates, same order of outcomes as rec_hop; a record whose core is not inside the window takes the general path.
        {
            uint32_t rel = (uint32_t)(first - tb);
            const uint32_t lim = (uint32_t)(te - tb), win = s.len;
            const uint64_t left64 = st.ulen - tb;
            const uint32_t left = left64 > 0xffffffffull ? 0xffffffffu : (uint32_t)left64;      // bytes of the stream from the tile's first byte on
            bool general = false;
#ifdef TL_STRICT_WALK
            while (rel < lim) {
                if (rel + 64u > win) { general = true; break; }
                uint32_t v; __builtin_memcpy(&v, buf + rel + 4u * (uint32_t)(lane & 15), 4);      // the core as dwords across lanes 0..8
                const int32_t b = (int32_t)__builtin_amdgcn_readlane((int)v, 0);
                if (b < 32) { err = 1; break; }
                // (rel + 64 <= win <= left: the core is inside the stream, so neither of rec_hop's first two "incomplete" tests can hold)
                const int32_t l_seq = __builtin_amdgcn_readlane((int)v, 5);
                const uint32_t l_qname = (uint32_t)__builtin_amdgcn_readlane((int)v, 3) & 0xffu, n_cigar = (uint32_t)__builtin_amdgcn_readlane((int)v, 4) & 0xffffu;
                if (l_seq < 0 || l_qname < 1u) { err = 1; break; }
                // n_cigar * 4 + l_qname + ceil(l_seq / 2) + l_seq < 2^18 + 2^8 + 2^30 + 2^31: no 32-bit overflow
                const uint32_t core = (n_cigar << 2) + l_qname + (((uint32_t)l_seq + 1u) >> 1) + (uint32_t)l_seq, body = (uint32_t)b - 32u;
                if (core > body) { err = 1; break; }
                if (left - rel - 36u < body) { if (st.final_batch) err = 1; break; }          // incomplete (o < ulen holds: the core is inside the stream)
                const int32_t tid = __builtin_amdgcn_readlane((int)v, 1), mtid = __builtin_amdgcn_readlane((int)v, 6);
                if (tid >= st.n_ref || tid < -1 || mtid >= st.n_ref || mtid < -1) { err = 1; break; }
                if (lane == 0 && cnt < TL_RECS) rl[cnt] = (uint16_t)rel;
                cnt++; rel += 4u + (uint32_t)b;
            }
#else
            // Round 4: the hop needs block_size only.  The core tests of bam_read1 (lengths that fit the body, reference ids in range: the
            // strict form above, ~45 scalar instructions per hop) are made again for every row by the row pass (rec_check_t, full), which
            // reports the first invalid row; rows are cut there either way, and a chain that started on a wrong speculation is replaced by
            // the repair rounds whatever it walked over.  What the walk must not do is leave the stream (the incomplete test) or stand still
            // (block_size < 32).  tools/soak.py --corrupt compares the rows and the error sign of damaged files with the oracle.
            while (rel < lim) {
                if (rel + 4u > win) { general = true; break; }
                uint32_t v; __builtin_memcpy(&v, buf + rel, 4);                              // (every lane reads the same word: an LDS broadcast)
                const int32_t b = (int32_t)__builtin_amdgcn_readfirstlane((int)v);
                if (b < 32) { err = 1; break; }
                if (left - rel - 4u < (uint32_t)b) { if (st.final_batch) err = 1; break; }     // the record runs past the stream: incomplete
                if (lane == 0 && cnt < TL_RECS) rl[cnt] = (uint16_t)rel;
                cnt++; rel += 4u + (uint32_t)b;
            }
#endif
            o = tb + rel;
            if (general && err == 0) {
                while (o < te) {
                    uint32_t bl = 0;
                    const int rc = ((o - tb) + 64 <= (uint64_t)s.len) ? rec_hop_uniform(st, ls, o, lane, bl) : rec_hop(st, gs, o, bl);
                    if (rc == REC_INCOMPLETE) { if (st.final_batch && o < st.ulen) err = 1; break; }
                    if (rc == REC_INVALID) { err = 1; break; }
                    if (lane == 0 && cnt < TL_RECS) rl[cnt] = (uint16_t)(o - tb);
                    cnt++; o += 4ull + bl;
                }
            }
        }
        en = o;
    } else if (first != NONE64) en = first;
    if (lane == 0) {
        out.first[t] = (first != NONE64 && first < te) ? first : NONE64;
        out.end_next[t] = en; out.count[t] = cnt; out.err[t] = err;
        tile_recs_first[t] = (first != NONE64 && first < te) ? first : NONE64;   // the list below belongs to a walk from here
    }
    // hand the record starts to bam_tile_unpack (it falls back to its own walk when a repair round moved `first`)
    __syncthreads();
    for (uint32_t k = lane; k < cnt && k < TL_RECS; k += 64) tile_recs[(size_t)t * TL_RECS + k] = rl[k];
}

// The @RG dictionary as unpack_one reads it: in HBM (DictG), or -- when it is small, as it nearly always is -- in a copy the workgroup made
// in LDS (DictL).  Two types with the same members so that each instantiation keeps its address space (a pointer that may be either
// becomes a flat access, and flat accesses to LDS fault on this system).  The comparison loop is a chain of dependent byte loads per
// record: out of HBM that was three ~600-cycle round trips per record of a file with one three-letter read group.
struct DictG { const uint32_t *off; const uint8_t *bytes; int32_t n; };
struct DictL { const uint32_t *off; const uint8_t *bytes; int32_t n; };
#define RGL_N 32
#define RGL_BYTES 256
template <class S, class D> __device__ __forceinline__ bool unpack_one(const BamStream &st, const S &s, const D &dict, uint64_t o, int64_t row,
                                                         uint32_t *rec_off, uint8_t *rg_flag, const BamCols &c) {
    RecInfo r; const int rcv = rec_check_t(st, s, o, r, true);
    if (row < 0) return rcv == REC_OK;                      // filtered out by the region predicate: still validated (the iterator reads it)
    rec_off[row] = (uint32_t)o;
    if (rcv != REC_OK) {          // first such row ends the scan (bam_reader.c:754-766); keep the column slots defined
        c.len_qname[row] = 0; c.len_cigar[row] = 0; c.len_seq[row] = 0; c.len_qual[row] = 0; c.len_rg[row] = 0;
        c.cig_rel[row] = 0; c.ncig_eff[row] = 0; c.rg_rel[row] = 0; c.rg_idx[row] = -1; rg_flag[row] = 0;
        return false;
    }
    c.flag[row] = (uint16_t)r.flag;
    c.pos[row] = (int64_t)r.pos + 1;
    c.mapq[row] = (int32_t)r.mapq;
    c.pnext[row] = (int64_t)r.mpos + 1;
    c.tlen[row] = (int64_t)r.tlen;
    c.tid[row] = r.tid; c.mtid[row] = r.mtid;
    const uint32_t ql = (uint32_t)(find_nul_t(s, o + 36, o + 36 + r.l_qname) - (o + 36));
    c.len_qname[row] = ql;
    uint32_t cl = 0;
    for (uint32_t j = 0; j < r.n_cigar_eff; j++) cl += ndigits(s.u32(r.cig_off + 4ull * j) >> 4) + 1;
    c.len_cigar[row] = r.n_cigar_eff ? cl : 1;
    c.cig_rel[row] = (uint32_t)(r.cig_off - o); c.ncig_eff[row] = r.n_cigar_eff;
    const uint64_t seq = o + 36 + r.l_qname + 4ull * r.n_cigar;
    const uint64_t qual = seq + (((uint64_t)r.l_seq + 1) >> 1);
    c.len_seq[row] = r.l_seq > 0 ? (st.seq_packed ? ((uint32_t)r.l_seq + 1u) >> 1 : (uint32_t)r.l_seq) : 1;
    c.len_qual[row] = (r.l_seq > 0 && s.u8(qual) != 255) ? (uint32_t)r.l_seq : 1;
    const uint64_t aux = qual + (uint64_t)r.l_seq, end = o + 4ull + r.block_len;
    // (bam_read1 does not look at the auxiliary fields: the walk serves the two read-group columns only -- a chain of dependent reads per
    //  record that a projection without them does not pay)
    bool bad = false; const uint64_t rg = st.want_rg ? aux_find_t(s, aux, end, 'R', 'G', &bad, r.cg_beg, r.cg_end) : NONE64;
    uint32_t rl = 0; int32_t rgi = -1; uint8_t rgv = 0;
    if (rg != NONE64 && (s.u8(rg) == 'Z' || s.u8(rg) == 'H')) {
        rgv = 1;
        rl = (uint32_t)(find_nul_t(s, rg + 1, end) - (rg + 1));      // aux_find_t proved the value is NUL-terminated inside the record
        for (int32_t q = 0; q < dict.n; q++) {
            const uint32_t a = dict.off[q], b = dict.off[q + 1];
            if (b - a != rl) continue;
            bool eq = true;
            for (uint32_t j = 0; j < rl; j++) if (dict.bytes[a + j] != s.u8(rg + 1 + j)) { eq = false; break; }
            if (eq) { rgi = q; break; }
        }
        c.rg_rel[row] = (uint32_t)(rg + 1 - o);
    } else c.rg_rel[row] = 0;
    c.len_rg[row] = rl; c.rg_idx[row] = rgi; rg_flag[row] = rgv;
    return true;
}

// One wave per tile: rebuild the record list in LDS, then one lane per record writes the fixed-width columns, the
// string lengths and the per-row scratch of the string pass.  rg_flag is one byte per row (packed to validity words later).
extern "C" __global__ void __launch_bounds__(64)
bam_tile_unpack(BamStream st, BamDict dict, int64_t ntiles, TileOut out, const uint32_t *rowbase, const uint64_t *res,
                int64_t nrows, uint32_t *rec_off, uint8_t *rg_flag, BamCols c, unsigned long long *bad_row, const uint32_t *row_map,
                const uint16_t *tile_recs, const uint64_t *tile_recs_first) {
    __shared__ __attribute__((aligned(16))) uint8_t buf[TL_TILE + TL_HALO + 16];     // +16: 8-byte reads may run past the last record
    __shared__ uint32_t recs[TL_RECS];
    __shared__ uint32_t rg_lo[RGL_N + 1];
    __shared__ uint8_t rg_lb[RGL_BYTES];
    const int lane = threadIdx.x;
    const int64_t t = blockIdx.x;
    if (t >= ntiles || (uint64_t)t > res[3]) return;
    // nrows < 0: the host has not seen the row count yet (the batch runs without a round trip in between): it is res[0], and the row arrays
    // hold -nrows - 1 rows (sized from the previous batches); a batch that needs more raises TR_F_ROWS in res[14] and writes nothing
    if (nrows < 0) {
        const int64_t cap = -nrows - 1; nrows = (int64_t)res[0];
        if (nrows > cap) { if (t == 0 && lane == 0) atomicOr((unsigned long long *)res + 14, TR_F_ROWS); return; }
    }
    const uint64_t first = out.first[t];
    const uint32_t n = out.count[t];
    if (first == NONE64 || n == 0) return;
    const uint32_t row0 = rowbase[t];
    if ((int64_t)row0 >= nrows) return;
    const uint64_t tb = (uint64_t)t * TL_TILE;
    const bool dict_lds = st.want_rg && dict.n_rg > 0 && dict.n_rg <= RGL_N && dict.n_bytes <= RGL_BYTES;
    if (dict_lds) {                                             // (visible behind tile_stage's barrier)
        for (int k = lane; k <= dict.n_rg; k += 64) rg_lo[k] = dict.rg_off[k];
        for (int k = lane; k < dict.n_bytes; k += 64) rg_lb[k] = dict.rg_bytes[k];
    }
    DictG dg; dg.off = dict.rg_off; dg.bytes = dict.rg_bytes; dg.n = dict.n_rg;
    DictL dl; dl.off = rg_lo; dl.bytes = rg_lb; dl.n = dict.n_rg;
    LSrc s; tile_stage(st, tb, buf, s, lane);
    PSrc ls; ls.l = buf; ls.base = tb;
    GSrc gs; gs.g = st.u;
    if (tile_recs_first[t] == first) {
        // record starts left by bam_tile_scan's walk from this same `first` (one coalesced load instead of a serial chain of hops)
        for (uint32_t k = lane; k < n && k < TL_RECS; k += 64) recs[k] = (uint32_t)(tb + tile_recs[(size_t)t * TL_RECS + k]);
    } else {   // the tile was re-walked by a repair round: walk again (the chain was validated by the scan / fix kernels)
        uint64_t o = first;
        for (uint32_t k = 0; k < n; k++) { if (lane == 0 && k < TL_RECS) recs[k] = (uint32_t)o; o += 4ull + (((o - tb) + 4 <= (uint64_t)s.len) ? ls.u32(o) : gs.u32(o)); }
    }
    __syncthreads();
    for (uint32_t k = lane; k < n; k += 64) {
        const int64_t row = (int64_t)row0 + k;
        if (row >= nrows) break;
        uint64_t o;
        if (k < TL_RECS) o = recs[k];
        else { o = first; for (uint32_t j = 0; j < k; j++) o += 4ull + gs.u32(o); }    // > 256 records per tile cannot happen (36-byte minimum)
        const uint64_t rel = o - tb;
        bool fast = false;
        if (rel + 4 <= (uint64_t)s.len) { const uint32_t bl = ls.u32(o); fast = (bl >= 32u) && (rel + 4ull + bl <= (uint64_t)s.len); }
        // with a region filter: row_map[row] = compacted row if kept (map[row+1] > map[row]), rows are validated either way
        int64_t dst = row;
        if (row_map) dst = (row_map[row + 1] > row_map[row]) ? (int64_t)row_map[row] : -1;
        const bool good = fast ? (dict_lds ? unpack_one(st, ls, dl, o, dst, rec_off, rg_flag, c) : unpack_one(st, ls, dg, o, dst, rec_off, rg_flag, c))
                               : unpack_one(st, gs, dg, o, dst, rec_off, rg_flag, c);
        if (!good) atomicMin(bad_row, (unsigned long long)row);
    }
}

// ---- string write pass, tile-centric (QNAME, CIGAR, SEQ, QUAL, READ_GROUP_ID heaps) ---------------------------------------
// One wave per tile, the tile staged in LDS like the other passes.  Rows are taken 64 at a time (one lane per row loads the row's
// offsets and parses its core); SEQ and QUAL are then written by CHUNK: the 16-byte output chunks of all rows of the group are
// numbered consecutively (wave prefix sum, chunk -> row map in LDS) and dealt to the lanes 64 at a time, so every lane converts
// and stores 16 bytes per step whatever the read length, and consecutive lanes store consecutive heap bytes.  QNAME, the RG
// value and the CIGAR text are short and written by the row's own lane.  Fields longer than TS_LONG bytes are streamed by the
// whole wave from HBM afterwards.  Same bytes as seq_to_string / qual_to_string / cigar_to_kstring (src/bam_reader.c:560-640).
#define TS_LONG 512u
#define TS_MAPN (64u * (TS_LONG / 16u))
struct TsSrc {                 // bytes of the inflated stream by absolute offset: LDS inside the staged window, HBM outside
    const uint8_t *l, *g; uint64_t base; uint32_t len;
    __device__ __forceinline__ uint64_t u64(uint64_t o) const {
        uint64_t v; const uint64_t rel = o - base;
        if (rel + 8 <= (uint64_t)len) __builtin_memcpy(&v, l + (uint32_t)rel, 8); else __builtin_memcpy(&v, g + o, 8);   // HBM side is padded
        return v;
    }
    __device__ __forceinline__ uint32_t u32(uint64_t o) const {
        uint32_t v; const uint64_t rel = o - base;
        if (rel + 4 <= (uint64_t)len) __builtin_memcpy(&v, l + (uint32_t)rel, 4); else __builtin_memcpy(&v, g + o, 4);
        return v;
    }
};
__device__ __forceinline__ void ts_seq_chunk(const TsSrc &src, uint64_t seq, uint32_t l_seq, uint32_t k, uint8_t *d, int packed) {
    const uint64_t p = src.u64(seq + 8ull * k);
    if (packed) {                                           // the 8 source bytes of this chunk as they are (16 bases)
        const uint32_t nb = ((l_seq + 1u) >> 1) - 8u * k, w[4] = {(uint32_t)p, (uint32_t)(p >> 32), 0u, 0u};
        store_n16(d + 8u * k, w, nb < 8u ? nb : 8u);
        return;
    }
    uint32_t w[4]; seq16((uint32_t)p, (uint32_t)(p >> 32), w);
    store_n16(d + 16u * k, w, l_seq - 16u * k);
}
// returns the index of the first byte that became NUL (absolute in the field) or 0xffffffff
__device__ __forceinline__ uint32_t ts_qual_chunk(const TsSrc &src, uint64_t qual, uint32_t l_seq, uint32_t k, uint8_t *d) {
    const uint64_t a = src.u64(qual + 16ull * k), b = src.u64(qual + 16ull * k + 8);
    uint32_t w[4] = {(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)};
    const uint32_t n = l_seq - 16u * k < 16u ? l_seq - 16u * k : 16u;
    uint32_t fz = 0xffffffffu;
    if (qual16(w)) {
#pragma unroll
        for (int q = 3; q >= 0; q--) {
            const uint32_t z = (w[q] - 0x01010101u) & ~w[q] & 0x80808080u;      // lowest set bit marks the first zero byte exactly
            if (z) fz = 4u * q + ((uint32_t)__ffs((int)z) - 1u) / 8u;
        }
        fz = fz < n ? 16u * k + fz : 0xffffffffu;
    }
    store_n16(d + 16u * k, w, n);
    return fz;
}
// Two waves per tile: wave 0 owns the rows (offsets, the short per-row fields), both waves share the SEQ/QUAL chunks; the tile is
// latency-bound, so a second wave on the same LDS image is nearly free occupancy.
#define TS_THREADS 128
extern "C" __global__ void __launch_bounds__(TS_THREADS)
bam_tile_strings(BamStream st, int64_t ntiles, TileOut out, const uint32_t *rowbase, const uint64_t *res, int64_t nrows_all, int64_t nrows,
                 const uint32_t *rec_off, const uint32_t *row_map, BamCols c, BamStrOut s, uint32_t colmask) {
    // colmask: projection pushdown (bit = read_bam column id): heaps of columns that are not projected are not written
    const bool w_qn = colmask & (1u << 0), w_cig = colmask & (1u << 5), w_seq = colmask & (1u << 9), w_qual = colmask & (1u << 10), w_rg = colmask & (1u << 11);
    __shared__ __attribute__((aligned(16))) uint8_t buf[TL_TILE + TL_HALO];
    __shared__ uint32_t r_seq[64], r_lseq[64], r_oseq[64], r_oqual[64], r_c0[64], r_nul[64];
    __shared__ uint8_t cmap[TS_MAPN];
    __shared__ uint32_t sT;
    const int tid = threadIdx.x, lane = tid & 63;
    const bool w0 = tid < 64;
    const int64_t t = blockIdx.x;
    if (t >= ntiles || (uint64_t)t > res[3]) return;
    if (nrows_all < 0) {          // the row count is on the device (see bam_tile_unpack); rows_guard has said whether everything fits
        if (res[14] & (TR_F_ROWS | TR_F_HEAP)) return;
        nrows_all = nrows = (int64_t)res[0];
    }
    const uint32_t n = out.count[t];
    if (out.first[t] == NONE64 || n == 0) return;
    const int64_t row0 = rowbase[t];
    if (row0 >= nrows_all) return;
    int64_t row1 = row0 + n; if (row1 > nrows_all) row1 = nrows_all;
    int64_t d0 = row_map ? (int64_t)row_map[row0] : row0, d1 = row_map ? (int64_t)row_map[row1] : row1;
    if (d1 > nrows) d1 = nrows;
    if (d0 >= d1) return;
    const uint64_t tb = (uint64_t)t * TL_TILE;
    // the first row group's offsets are requested before the tile is staged: both HBM round trips overlap
    struct RowIn { uint32_t o, len_seq, len_qn, rl, rg_rel, ne, cig_rel, off_qn, off_rg, off_cig, off_seq, off_qual; };
    auto row_in = [&](int64_t r) {
        RowIn q; q.o = rec_off[r];
        q.len_seq = c.len_seq[r]; q.len_qn = c.len_qname[r]; q.rl = c.len_rg[r]; q.rg_rel = c.rg_rel[r]; q.ne = c.ncig_eff[r]; q.cig_rel = c.cig_rel[r];
        q.off_qn = s.off_qname[r]; q.off_rg = s.off_rg[r]; q.off_cig = s.off_cigar[r]; q.off_seq = s.off_seq[r]; q.off_qual = s.off_qual[r];
        return q;
    };
    RowIn ri; memset(&ri, 0, sizeof(ri));
    if (w0) ri = row_in(d0 + lane < d1 ? d0 + lane : d0);
    LSrc ss; tile_stage(st, tb, buf, ss, tid, TS_THREADS);
    TsSrc src; src.l = buf; src.g = st.u; src.base = tb; src.len = ss.len;

    for (int64_t g0 = d0; g0 < d1; g0 += 64) {
        const int64_t d = g0 + lane;
        const bool act = w0 && d < d1;
        if (w0 && g0 != d0) ri = row_in(act ? d : g0);
        // ---- wave 0, one lane per row: offsets, core fields ----
        const uint64_t o = ri.o;
        const uint32_t len_seq = ri.len_seq, len_qn = ri.len_qn, rl = ri.rl, rg_rel = ri.rg_rel, ne = ri.ne, cig_rel = ri.cig_rel;
        const uint32_t off_qn = ri.off_qn, off_rg = ri.off_rg, off_cig = ri.off_cig, off_seq = ri.off_seq, off_qual = ri.off_qual;
        const bool ok = act && len_seq != 0;                       // rows that failed validation reserve nothing
        bool star = true, lng = false; uint32_t l_seq = 0; uint64_t seq = 0;
        if (w0) {
            const uint32_t x2 = src.u32(o + 12), x3 = src.u32(o + 16);
            const int32_t l_seq_raw = (int32_t)src.u32(o + 20);
            l_seq = (ok && l_seq_raw > 0) ? (uint32_t)l_seq_raw : 0u;
            seq = o + 36 + (x2 & 0xff) + 4ull * (x3 & 0xffff);
            const uint64_t qual = seq + (((uint64_t)l_seq + 1) >> 1);
            star = !(l_seq > 0 && (uint8_t)src.u32(qual) != 255);
            lng = l_seq > TS_LONG;
            const uint32_t nch = (ok && !lng && (w_seq || w_qual)) ? (l_seq + 15u) >> 4 : 0u;
            const uint32_t cinc = wave_incl_scan(nch, lane);
            const uint32_t c0 = cinc - nch;
            if (lane == 63) sT = cinc;
            r_seq[lane] = (uint32_t)(seq - tb); r_lseq[lane] = star ? (l_seq | 0x80000000u) : l_seq; r_oseq[lane] = off_seq; r_oqual[lane] = off_qual;
            r_c0[lane] = c0; r_nul[lane] = 0xffffffffu;
            for (uint32_t k = 0; k < nch; k++) cmap[c0 + k] = (uint8_t)lane;
        }
        __syncthreads();
        const uint32_t T = sT;
        // ---- SEQ / QUAL by chunk, both waves ----
        for (uint32_t cb = 0; cb < T; cb += TS_THREADS) {
            const uint32_t ch = cb + (uint32_t)tid;
            if (ch < T) {
                const uint32_t j = cmap[ch], k = ch - r_c0[j], ls = r_lseq[j], lq = ls & 0x7fffffffu;
                const uint64_t sq = tb + r_seq[j];
                if (w_seq) ts_seq_chunk(src, sq, lq, k, s.seq + r_oseq[j], st.seq_packed);
                if (w_qual && !(ls >> 31)) {
                    const uint32_t fz = ts_qual_chunk(src, sq + (((uint64_t)lq + 1) >> 1), lq, k, s.qual + r_oqual[j]);
                    if (fz != 0xffffffffu) atomicMin(&r_nul[j], fz);
                }
            }
        }
        // ---- per-row pieces (wave 0) ----
        if (ok) {
            if (w_seq && l_seq == 0) s.seq[off_seq] = st.seq_packed ? 0 : '*';
            if (w_qual && star) s.qual[off_qual] = '*';
            for (uint32_t b = 0; w_qn && b < len_qn; b += 16) {
                const uint64_t a = src.u64(o + 36 + b), e = src.u64(o + 36 + b + 8);
                const uint32_t w[4] = {(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)e, (uint32_t)(e >> 32)};
                store_n16(s.qname + off_qn + b, w, len_qn - b);
            }
            for (uint32_t b = 0; w_rg && b < rl; b += 16) {
                const uint64_t a = src.u64(o + rg_rel + b), e = src.u64(o + rg_rel + b + 8);
                const uint32_t w[4] = {(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)e, (uint32_t)(e >> 32)};
                store_n16(s.rg + off_rg + b, w, rl - b);
            }
            uint8_t *dc = s.cigar + off_cig;
            if (w_cig && ne == 0) dc[0] = '*';
            uint32_t p = 0;
            for (uint32_t q = 0; w_cig && q < ne; q++) {
                const uint32_t op = src.u32(o + cig_rel + 4ull * q);
                uint32_t ol = op >> 4; const uint32_t nd = ndigits(ol);
                for (uint32_t z = 0; z < nd; z++) { dc[p + nd - 1 - z] = (uint8_t)('0' + ol % 10); ol /= 10; }
                // sam.h:112 BAM_CIGAR_STR "MIDNSHP=XB", '?' beyond; byte gather with v_perm_b32
                const uint32_t C0 = 0x4e44494du, C1 = 0x3d504853u, C2 = 0x3f3f4258u, C3 = 0x3f3f3f3fu;
                const uint32_t oc = op & 0xf, selb = (oc & 7u) | 0x0c0c0c00u;
                dc[p + nd] = (uint8_t)((oc & 8u) ? __builtin_amdgcn_perm(C3, C2, selb) : __builtin_amdgcn_perm(C1, C0, selb));
                p += nd + 1;
            }
        }
        // ---- long fields: wave 0 streams one row at a time ----
        if (w0) {
            uint64_t LM = __ballot(ok && lng && (w_seq || w_qual));
            while (LM) {
                const int i = __ffsll((unsigned long long)LM) - 1; LM &= LM - 1;
                const uint32_t lq = RDLANE(l_seq, i), st_i = RDLANE((uint32_t)star, i), os = RDLANE(off_seq, i), oq = RDLANE(off_qual, i);
                const uint64_t sq = ((uint64_t)RDLANE((uint32_t)(seq >> 32), i) << 32) | RDLANE((uint32_t)seq, i);
                uint32_t fzm = 0xffffffffu;
                for (uint32_t k = lane; k < (lq + 15u) >> 4; k += 64) {
                    if (w_seq) ts_seq_chunk(src, sq, lq, k, s.seq + os, st.seq_packed);
                    if (w_qual && !st_i) { const uint32_t fz = ts_qual_chunk(src, sq + (((uint64_t)lq + 1) >> 1), lq, k, s.qual + oq); fzm = fz < fzm ? fz : fzm; }
                }
                if (fzm != 0xffffffffu) atomicMin(&r_nul[i], fzm);
            }
        }
        __syncthreads();
        if (ok && w_qual) { const uint32_t fz = r_nul[lane]; s.alen_qual[d] = star ? 1u : (fz != 0xffffffffu ? fz : l_seq); }
        if (act && w_seq && s.seq_chars) s.seq_chars[d] = ok ? l_seq : 0u;
        __syncthreads();
    }
}

// validity words from the per-row flags
extern "C" __global__ void __launch_bounds__(256)
bam_pack_validity(const uint8_t *flag, int64_t nrows, uint64_t *words) {
    int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool v = row < nrows && flag[row] != 0;
    uint64_t m = __ballot(v);
    if ((threadIdx.x & 63) == 0 && row < nrows) words[row >> 6] = m;
}


// behind the string scans of a batch whose sizes the host has not seen: heap totals against the arenas' room, the batch's first record
//   res[8..12] = heap bytes (QNAME, CIGAR, SEQ, QUAL, READ_GROUP_ID), res[13] = offset of the first record, res[14] |= TR_F_HEAP
struct HeapCaps { uint64_t cap[5]; };
extern "C" __global__ void __launch_bounds__(64)
bam_rows_guard(unsigned long long *res, const uint64_t *scan_total, HeapCaps caps, const uint32_t *rec_off, int strings) {
    const int k = threadIdx.x;
    if (k == 0) res[13] = res[0] > 0 && !(res[14] & TR_F_ROWS) ? (unsigned long long)rec_off[0] : 0ull;
    if (strings && k < 5) {
        const unsigned long long t = scan_total[k];
        res[8 + k] = t;
        if (caps.cap[k] != ~0ull && t > caps.cap[k]) atomicOr(res + 14, TR_F_HEAP);
    }
}

// ---- QUAL over PCIe by the batch's own alphabet (dhts_bam_set_qual_packed; the read-back of dhts_fetch.inc) ---------------------------------
// qual_presence: which characters occur in the heap -- 128 bits for the ASCII range (a quality + 33 is at most 126, '*' is 42) and one flag
// for anything above (a quality beyond 93: the batch then travels as characters).  qual_pack: 16 characters per lane through a 256-entry code
// table in LDS into 2- / 4-bit codes, little end first; the heap is dense, so the code stream ignores row boundaries and the consumer finds
// character k at bit k * bits.
extern "C" __global__ void __launch_bounds__(256)
qual_presence(const uint8_t *__restrict__ q, uint64_t n, unsigned long long *__restrict__ mask3) {
    __shared__ unsigned long long sm[3];
    if (threadIdx.x < 3) sm[threadIdx.x] = 0ull;
    __syncthreads();
    unsigned long long lo = 0, hi = 0; uint32_t big = 0;
    for (uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16u; i < n; i += (uint64_t)gridDim.x * blockDim.x * 16u) {
        uint32_t w[4] = {0, 0, 0, 0};
        if (i + 16 <= n) { const uint4 v = *(const uint4 *)(q + i); w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; }
        else { const uint8_t first = q[i]; for (uint64_t k = i; k < i + 16; k++) w[(k - i) >> 2] |= (uint32_t)(k < n ? q[k] : first) << (8 * ((k - i) & 3)); }
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const uint32_t b = (w[k >> 2] >> (8 * (k & 3))) & 0xffu;
            const unsigned long long bit = 1ull << (b & 63u);
            lo |= (b & 64u) ? 0ull : bit; hi |= (b & 64u) ? bit : 0ull; big |= b >> 7;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { lo |= __shfl_xor(lo, d, 64); hi |= __shfl_xor(hi, d, 64); big |= __shfl_xor(big, d, 64); }
    if ((threadIdx.x & 63) == 0) { atomicOr(&sm[0], lo); atomicOr(&sm[1], hi); atomicOr(&sm[2], (unsigned long long)big); }
    __syncthreads();
    if (threadIdx.x < 3 && sm[threadIdx.x]) atomicOr(mask3 + threadIdx.x, sm[threadIdx.x]);
}
struct QualCodes { uint8_t code[128]; };            // character (< 128) -> code
template <int BITS> __global__ void __launch_bounds__(256)
qual_pack(const uint8_t *__restrict__ q, uint64_t n, QualCodes lut, uint8_t *__restrict__ dst) {
    __shared__ uint8_t code[128];
    if (threadIdx.x < 128) code[threadIdx.x] = lut.code[threadIdx.x];
    __syncthreads();
    const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16u;
    if (i >= n) return;
    uint32_t w[4] = {0, 0, 0, 0};
    if (i + 16 <= n) { const uint4 v = *(const uint4 *)(q + i); w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; }
    else for (uint64_t k = i; k < n; k++) w[(k - i) >> 2] |= (uint32_t)q[k] << (8 * ((k - i) & 3));
    unsigned long long out = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) out |= (unsigned long long)code[(w[k >> 2] >> (8 * (k & 3))) & 0x7fu] << (BITS * k);
    if (BITS == 2) { const uint32_t o32 = (uint32_t)out; __builtin_memcpy(dst + i / 4, &o32, 4); }       // (the stream is padded to whole 16-character groups)
    else __builtin_memcpy(dst + i / 2, &out, 8);
}

// ---- region predicate -----------------------------------------------------------------------------------
// keep[row] = 1 iff the record overlaps one of the merged query intervals of its reference:
//   end > iv.beg && iv.end > beg, beg = pos, end = bam_endpos (htslib sam.c:668-673: pos + reference length of the effective
//   CIGAR, 1 if that is 0 or the read is unmapped)          -- hts_itr_multi_next hts.c:4575-4592, sam_readrec sam.c:1596-1612
// Unplaced reads (tid < 0) are kept only by the "*" region.  Intervals are sorted and merged per tid (region.c:100-120).
struct RegionDev { const int64_t *beg, *end; const uint32_t *tid_first; int32_t n_ref, all, nocoor, pad; };

extern "C" __global__ void __launch_bounds__(256)
bam_region_keep(BamStream st, RegionDev rg, const uint32_t *rec_off, int64_t nrows, uint32_t *keep, uint64_t end_rel) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= nrows) return;
    GSrc gs; gs.g = st.u;
    const uint64_t o = rec_off[row];
    RecInfo r;
    uint32_t k = 0;
    // end_rel: where the current index window ends in this buffer; a record that starts there or later belongs to the next window
    if (o < end_rel && rec_check_t(st, gs, o, r, true) == REC_OK) {
        if (rg.all) k = 1;
        else if (r.tid < 0) k = rg.nocoor ? 1u : 0u;
        else if (r.tid < rg.n_ref) {
            int64_t rlen = 0;
            if (!(r.flag & 4)) for (uint32_t j = 0; j < r.n_cigar_eff; j++) { const uint32_t c = gs.u32(r.cig_off + 4ull * j); if ((0x3C1A7u >> ((c & 0xf) << 1)) & 2u) rlen += c >> 4; }
            if (rlen == 0) rlen = 1;
            const int64_t beg = r.pos, end = (int64_t)r.pos + rlen;
            uint32_t lo = rg.tid_first[r.tid], hi = rg.tid_first[r.tid + 1];
            // first interval with iv.end > beg (ends are increasing after the merge), then test its begin
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (rg.end[mid] > beg) hi = mid; else lo = mid + 1; }
            if (lo < rg.tid_first[r.tid + 1] && end > rg.beg[lo]) k = 1;
        }
    }
    keep[row] = k;
}

// ---- interval overlap join ----------------------------------------------------------------
// For every row the ids of the caller's intervals that overlap the read, with the semantics of cgranges' cr_overlap
// (ref: third_party/cgranges/cgranges.c:255-297; the test is `st < en_i && st_i < en`, half-open, same contig).  The read's
// interval is the one the region iterator uses: [pos, bam_endpos) (htslib sam.c:668-673).  Intervals arrive sorted by
// (tid, start) with a per-contig running maximum of their ends, which replaces the implicit max-end tree: candidates are
// [first index whose running max end > beg, first index whose start >= end) and each is tested for en_i > beg.
// Ids come out in (start, input order) order, not in cgranges' tree-walk order: the result is a set per row.
struct OverlapDev { const int64_t *beg, *end, *pmax, *bmax; const uint32_t *id; const uint32_t *tid_first; int32_t n_ref, pad; };   // bmax: max end of every 64 sorted intervals

template <bool WRITE> __global__ void __launch_bounds__(256)
bam_overlap_cells(BamStream st, OverlapDev ov, const uint32_t *rec_off, BamCols c, int64_t nrows, uint32_t *cnt, const uint32_t *off, uint32_t *ids) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= nrows) return;
    // the row was validated and unpacked by bam_tile_unpack: tid / POS / FLAG and the effective CIGAR location come from its columns
    const int32_t tid = c.tid[row];
    uint32_t n = 0;
    if (tid >= 0 && tid < ov.n_ref && c.len_seq[row] != 0) {
        const uint8_t *cig = st.u + rec_off[row] + c.cig_rel[row];
        const uint32_t ne = c.ncig_eff[row];
        int64_t rlen = 0;
        if (!(c.flag[row] & 4)) for (uint32_t j = 0; j < ne; j++) { const uint32_t op = ldu32(cig + 4ull * j); if ((0x3C1A7u >> ((op & 0xf) << 1)) & 2u) rlen += op >> 4; }
        if (rlen == 0) rlen = 1;
        const int64_t beg = c.pos[row] - 1, end = beg + rlen;
        const uint32_t f0 = ov.tid_first[tid], f1 = ov.tid_first[tid + 1];
        uint32_t lo = f0, hi = f1;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (ov.beg[mid] >= end) hi = mid; else lo = mid + 1; }
        const uint32_t stop = lo;                                   // intervals [f0, stop) start before the read ends
        lo = f0; hi = stop;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (ov.pmax[mid] > beg) hi = mid; else lo = mid + 1; }
        uint32_t w = WRITE ? off[row] : 0u;
        // candidates [lo, stop): 64-interval blocks whose largest end does not reach the read are skipped whole (long nested
        // intervals make the running-max bound loose; the block maxima keep the walk proportional to the blocks that can match)
        for (uint32_t i = lo; i < stop;) {
            const uint32_t be = (i | 63u) + 1u < stop ? (i | 63u) + 1u : stop;
            if (ov.bmax[i >> 6] > beg) { for (; i < be; i++) if (ov.end[i] > beg) { if (WRITE) ids[w++] = ov.id[i]; n++; } }
            i = be;
        }
    }
    if (!WRITE) cnt[row] = n;
}

// full bam_read1 validation of every row of a batch (what bam_tile_unpack checks while it writes): used to tell a false start of a
// speculated shard from a good one before any column is written
extern "C" __global__ void __launch_bounds__(256)
bam_validate_rows(BamStream st, const uint32_t *rec_off, int64_t nrows, unsigned long long *bad_row) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= nrows) return;
    GSrc gs; gs.g = st.u;
    RecInfo r;
    if (rec_check_t(st, gs, rec_off[row], r, true) != REC_OK) atomicMin(bad_row, (unsigned long long)row);
}
