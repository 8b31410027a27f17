// hts_index.hip -- BAI / CSI / TBI construction as array passes over the records of a scan (gfx950 kernels + host assembly).
//
// Replaces htslib's record-by-record index builder (hts.c: hts_idx_push 2553-2640, hts_idx_finish 2400-2690 with update_loff 2426-2455
// and compress_binning 2457-2508, idx_save_core 2754-2818 / hts_idx_save_as 2820-2900) for src/hts_index_builder.c's bam_index, bcf_index
// and tabix_index.  The output must equal htslib's byte for byte (golden range.bam.bai, vcf_file.bcf.csi, index.vcf.gz.tbi/.csi, the four
// tabix presets), but nothing here pushes one record at a time through a state machine:
//
//   rows -> (bin, "a run of equal (sequence, bin) starts here", the offset in front of the row)          one lane per row, neighbours compared
//        -> run table {sequence, bin, first offset}: a scan over the start flags compacts it            (a chunk of the index = one run: it ends
//                                                                                                         where the next run begins)
//        -> linear index: min over the rows that touch a 2^min_shift window of the offset in front of them (atomicMin; only rows that reach
//                                                                                                         a window their predecessor did not)
//        -> per sequence: mapped / unmapped counts, one-run-per-sequence and sortedness violations        (wave-reduced atomics, flags)
//
// The device does this for read_bam's batches (bam_index_rows + idx_runs_write); the text formats (VCF text, tabix presets), whose sequence
// numbering is discovered while reading, hand arrays to the host version of the same passes (IndexAcc::add_rows).  IndexAcc::finish turns
// the run table into bins: sort by (sequence, bin), move bins that span less than 64 KiB of file into their parents level by level, sort
// and coalesce chunks that touch the same BGZF block, fill empty linear-index windows from the right -- the rules of the format.
#pragma once
#include <algorithm>
#include <string.h>
#include <string>
#include <vector>

struct IdxGeom { int32_t min_shift, n_lvls; };
__host__ __device__ static inline uint32_t idx_reg2bin(int64_t beg, int64_t end, IdxGeom g) {       // hts_reg2bin (htslib/hts.h), SAM spec 5.3
    int s = g.min_shift; uint32_t t = (uint32_t)((((uint64_t)1 << ((g.n_lvls << 1) + g.n_lvls)) - 1) / 7);
    --end;
    for (int l = g.n_lvls; l > 0; --l, s += 3, t -= 1u << ((l << 1) + l)) if ((beg >> s) == (end >> s)) return t + (uint32_t)(beg >> s);
    return 0;
}

// ---- device side (read_bam batches) ------------------------------------------------------------------------------------------------------
struct IdxRun { int32_t tid; uint32_t bin; uint64_t u; };                         // one run: its sequence, its bin, the virtual offset in front of its first row
struct IdxCarry { int32_t tid; uint32_t bin; int64_t beg; int64_t e; uint64_t v; uint32_t any, pad; };      // the last row of the previous batch
#define IDX_ERR_MAXPOS 1u
#define IDX_ERR_NOCOOR 2u
#define IDX_ERR_UNSORTED 4u
#define IDX_ERR_ENDBEG 8u
#define IDX_ERR_TIDRANGE 16u
#define IDX_ERR_LINCAP 32u
struct IdxDev {
    IdxGeom g; int32_t n_ref;
    const uint64_t *lin_base;        // [n_ref + 1] first window of every sequence in `lin`
    unsigned long long *lin;         // windows of all sequences, initialised to ~0
    unsigned long long *nmap, *nunmap;   // [n_ref]
    uint32_t *tid_runs;              // [n_ref] how many separate stretches of the file a sequence has (must end up <= 1)
    uint32_t *max_win;               // [n_ref] highest window touched + 1
    unsigned long long *n_nocoor; uint32_t *err;
    const IdxCarry *carry_in; IdxCarry *carry_out;   // the last row of the previous batch / of this one (two slots: waves run in any order)
};
// bgzf_tell after the reader has consumed the inflated stream up to offset u (bgzf.c bgzf_read: a read that ends exactly at a block end
// reports the NEXT block's address with in-block offset 0)
__device__ __forceinline__ uint64_t idx_tell(const uint64_t *uoff, const uint64_t *coff, int64_t nb, uint64_t comp_len, uint64_t u) {
    int64_t lo = 0, hi = nb + 1;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (uoff[mid] < u) lo = mid + 1; else hi = mid; }
    if (lo <= nb && uoff[lo] == u) return (lo < nb ? coff[lo] : comp_len) << 16;
    return (coff[lo - 1] << 16) | (u - uoff[lo - 1]);
}
// one lane per row of the batch: the row's interval (bam_endpos: the CIGAR's reference length, 1 for none), bin, the virtual offsets around
// it; start[r] = 1 where a run begins; the linear index, the counts and the violation flags by atomics
extern "C" __global__ void __launch_bounds__(256)
bam_index_rows(BamStream st, const uint32_t *rec_off, BamCols c, const int32_t *tid_col, int64_t nrows, uint64_t out_base, uint64_t end_uoff,
               const uint64_t *uoff, const uint64_t *coff, int64_t nb, uint64_t comp_len, IdxDev d, uint32_t *start, IdxRun *row_run) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int32_t tid = -1; int64_t beg = -1, end = 0; bool mapped = false; uint64_t v_after = 0;
    const bool in = row < nrows;
    if (in) {
        const uint8_t *cig = st.u + rec_off[row] + c.cig_rel[row];
        const uint32_t ne = c.ncig_eff[row];
        int64_t rlen = 0;
        mapped = !(c.flag[row] & 4);
        if (mapped) for (uint32_t j = 0; j < ne; j++) { const uint32_t op = ldu32(cig + 4ull * j); if ((0x3C1A7u >> ((op & 0xf) << 1)) & 2u) rlen += op >> 4; }
        if (rlen == 0) rlen = 1;
        tid = tid_col[row]; beg = c.pos[row] - 1; end = beg + rlen;
        const uint64_t u_end = (row + 1 < nrows) ? out_base + rec_off[row + 1] : end_uoff;      // records are contiguous: a record ends where the next one starts
        v_after = idx_tell(uoff, coff, nb, comp_len, u_end);
    }
    if (tid < 0) { beg = -1; end = 0; }
    // the row in front of this one: the previous lane, the last lane of the previous wave (recomputed by lane 0: a second, cheap evaluation),
    // or the carry of the previous batch
    int32_t p_tid; uint32_t p_bin; int64_t p_beg, p_e; uint64_t v_before; bool has_prev = true;
    int64_t cb = beg < 0 ? 0 : beg, ce = end <= 0 ? 1 : end;                          // (hts_idx_push clamps the interval for the linear index and the bin alike)
    const uint32_t bin = in ? idx_reg2bin(tid < 0 ? beg : cb, tid < 0 ? end : ce, d.g) : 0u;
    const int64_t e_win = tid >= 0 ? (ce - 1) >> d.g.min_shift : -1;
    {
        // neighbour exchange inside the wave
        p_tid = __shfl_up(tid, 1, 64); p_bin = (uint32_t)__shfl_up((int)bin, 1, 64);
        // (hts_idx_push remembers the CLAMPED begin of a placed row -- last_coor = beg after `if (beg < 0) beg = 0`, hts.c:2591, 2620-2633 -- and
        //  compares it with the next row's unclamped one: two placed rows with POS = 0 in a row are "unsorted" there, and here)
        const int64_t kbeg = tid >= 0 ? cb : beg;
        const uint32_t pb_lo = (uint32_t)__shfl_up((int)(uint32_t)kbeg, 1, 64), pb_hi = (uint32_t)__shfl_up((int)(uint32_t)((uint64_t)kbeg >> 32), 1, 64);
        p_beg = (int64_t)(((uint64_t)pb_hi << 32) | pb_lo);
        const uint32_t pe_lo = (uint32_t)__shfl_up((int)(uint32_t)e_win, 1, 64), pe_hi = (uint32_t)__shfl_up((int)(uint32_t)((uint64_t)e_win >> 32), 1, 64);
        p_e = (int64_t)(((uint64_t)pe_hi << 32) | pe_lo);
        const uint32_t pv_lo = (uint32_t)__shfl_up((int)(uint32_t)v_after, 1, 64), pv_hi = (uint32_t)__shfl_up((int)(uint32_t)(v_after >> 32), 1, 64);
        v_before = ((uint64_t)pv_hi << 32) | pv_lo;
        if (lane == 0 && in) {
            if (row == 0) {
                const IdxCarry k = *d.carry_in;
                has_prev = k.any != 0; p_tid = k.tid; p_bin = k.bin; p_beg = k.beg; p_e = k.e; v_before = k.v;
            } else {
                const int64_t q = row - 1;
                const uint8_t *cg = st.u + rec_off[q] + c.cig_rel[q];
                int64_t rl = 0;
                if (!(c.flag[q] & 4)) for (uint32_t j = 0; j < c.ncig_eff[q]; j++) { const uint32_t op = ldu32(cg + 4ull * j); if ((0x3C1A7u >> ((op & 0xf) << 1)) & 2u) rl += op >> 4; }
                if (rl == 0) rl = 1;
                p_tid = tid_col[q]; p_beg = c.pos[q] - 1; int64_t pe = p_beg + rl;
                if (p_tid < 0) { p_beg = -1; pe = 0; }
                const int64_t qb = p_beg < 0 ? 0 : p_beg, qe = pe <= 0 ? 1 : pe;
                p_bin = idx_reg2bin(p_tid < 0 ? p_beg : qb, p_tid < 0 ? pe : qe, d.g);
                p_e = p_tid >= 0 ? (qe - 1) >> d.g.min_shift : -1;
                if (p_tid >= 0) p_beg = qb;
                v_before = idx_tell(uoff, coff, nb, comp_len, out_base + rec_off[row]);
            }
        }
    }
    if (!in) return;
    uint32_t err = 0;
    const int64_t maxpos = 1ll << (d.g.min_shift + 3 * d.g.n_lvls);
    if (tid >= 0 && !(beg <= maxpos && end <= maxpos)) err |= IDX_ERR_MAXPOS;
    if (tid >= d.n_ref) err |= IDX_ERR_TIDRANGE;
    if (end < beg) err |= IDX_ERR_ENDBEG;
    const bool new_tid = !has_prev || p_tid != tid;
    if (has_prev && !new_tid && tid >= 0 && p_beg > beg) err |= IDX_ERR_UNSORTED;
    if (has_prev && new_tid && tid >= 0 && p_tid < 0) err |= IDX_ERR_NOCOOR;              // placed reads behind unplaced ones
    const bool st_run = new_tid || p_bin != bin;
    start[row] = st_run ? 1u : 0u;
    IdxRun rr; rr.tid = tid; rr.bin = bin; rr.u = v_before; row_run[row] = rr;
    if (tid >= 0 && tid < d.n_ref) {
        if (new_tid) atomicAdd(d.tid_runs + tid, 1u);
        // windows this row reaches that the row in front of it did not (rows are sorted: everything up to p_e is already claimed by an earlier
        // offset); an out-of-order row is an error anyway
        const int64_t b_win = cb >> d.g.min_shift;
        int64_t w0 = b_win; if (!new_tid && p_e >= w0) w0 = p_e + 1;
        const uint64_t base = d.lin_base[tid], cap = d.lin_base[tid + 1] - base;
        // (a row that reaches beyond the room still says how many windows its sequence needs: the host repeats the build with that room)
        if ((uint64_t)(e_win + 1) > cap) { err |= IDX_ERR_LINCAP; atomicMax(d.max_win + tid, (uint32_t)(e_win + 1 > 0x7fffffffll ? 0x7fffffffll : e_win + 1)); }
        else {
            for (int64_t w = w0; w <= e_win; w++) atomicMin(d.lin + base + (uint64_t)w, (unsigned long long)v_before);
            if (w0 <= e_win || new_tid) atomicMax(d.max_win + tid, (uint32_t)(e_win + 1));
        }
    }
    // counts: one atomic per wave when all its rows belong to one sequence (the usual case)
    {
        const int32_t t0 = __shfl(tid, 0, 64);
        const bool uni = __all(tid == t0 || !in);
        const uint64_t mm = __ballot(mapped), act = __ballot(true);
        if (uni) {
            if (lane == (int)(__ffsll((unsigned long long)act) - 1)) {
                if (t0 >= 0 && t0 < d.n_ref) { atomicAdd(d.nmap + t0, (unsigned long long)__popcll(mm)); atomicAdd(d.nunmap + t0, (unsigned long long)__popcll(act & ~mm)); }
                else if (t0 < 0) atomicAdd(d.n_nocoor, (unsigned long long)__popcll(act));
            }
        } else {
            if (tid >= 0 && tid < d.n_ref) atomicAdd(mapped ? d.nmap + tid : d.nunmap + tid, 1ull);
            else if (tid < 0) atomicAdd(d.n_nocoor, 1ull);
        }
    }
    if (err) atomicOr(d.err, err);
    if (row == nrows - 1) { IdxCarry k; k.tid = tid; k.bin = bin; k.beg = tid >= 0 ? cb : beg; k.e = e_win; k.v = v_after; k.any = 1; k.pad = 0; *d.carry_out = k; }
}
// the runs of a batch, compacted: pos[] = exclusive scan of start[]
extern "C" __global__ void __launch_bounds__(256)
idx_runs_write(const uint32_t *start, const uint32_t *pos, const IdxRun *row_run, int64_t nrows, IdxRun *runs) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row < nrows && start[row]) runs[pos[row]] = row_run[row];
}

// ---- host side: run table -> bins -> bytes --------------------------------------------------------------------------------------------------
namespace {
struct IndexAcc {
    IdxGeom g{14, 5}; bool csi = false, tbi = false, grow = false;
    int n_ref = 0; std::vector<uint8_t> aux; std::string err;
    std::vector<IdxRun> runs;                                   // in file order
    std::vector<std::vector<uint64_t>> lin;                     // per sequence: offset of the first row that touches each window (~0: none)
    std::vector<uint64_t> nmap, nunmap; std::vector<uint32_t> tid_runs;
    uint64_t n_nocoor = 0, v0 = 0;
    IdxCarry last{};                                            // the last row seen (host passes)
    // finished form: per sequence its bins in ascending order, every bin a range of `chunks`
    struct Chunk { uint64_t u, v; };
    struct Bin { uint32_t id; uint64_t loff; uint32_t c0, nc; };
    std::vector<std::vector<Bin>> bins; std::vector<Chunk> chunks; std::vector<char> has;
    uint32_t n_bins() const { return (uint32_t)((((uint64_t)1 << (3 * g.n_lvls + 3)) - 1) / 7); }
    void set_csi(int ms, int lv) { csi = true; g.min_shift = ms; g.n_lvls = lv; }
    void begin(int n, uint64_t first_offset) { n_ref = n; v0 = first_offset; lin.assign(n, {}); nmap.assign(n, 0); nunmap.assign(n, 0); tid_runs.assign(n, 0); last = IdxCarry{}; last.v = first_offset; }
    void reset_rows() { runs.clear(); n_nocoor = 0; }              // a build that starts over (begin() follows)
    void ensure_ref(int32_t tid) { if (tid >= n_ref) { n_ref = tid + 1; lin.resize(n_ref); nmap.resize(n_ref, 0); nunmap.resize(n_ref, 0); tid_runs.resize(n_ref, 0); } }
    const char *err_text(uint32_t e) const {
        if (e & IDX_ERR_MAXPOS) return csi ? "Region cannot be stored in a csi index with these parameters. Please use a larger min_shift or depth" : "Region cannot be stored in a bai index. Try using a csi index";
        if (e & IDX_ERR_TIDRANGE) return "record refers to a reference beyond the header";
        if (e & IDX_ERR_NOCOOR) return "NO_COOR reads not in a single block at the end";
        if (e & IDX_ERR_UNSORTED) return "Unsorted positions";
        if (e & IDX_ERR_ENDBEG) return "Invalid record: end < begin";
        if (e & IDX_ERR_LINCAP) return "record reaches beyond its reference sequence";
        return "index build failed";
    }
    // the same passes as bam_index_rows over host arrays (text formats): every row [beg, end) 0-based on sequence tid (< 0: unplaced),
    // vafter = the virtual offset behind it
    // One row of the passes, the state of the row in front of it carried in `last`.  Returns the violation flags of the row (0: none); a row
    // with an interval violation (IDX_ROW_ERRS) leaves the tables untouched.
    static constexpr uint32_t IDX_ROW_ERRS = IDX_ERR_MAXPOS | IDX_ERR_ENDBEG | IDX_ERR_TIDRANGE;
    inline uint32_t push_row(int32_t tid, int64_t beg, int64_t end, uint64_t vafter, bool mapped, int64_t maxpos) {
        uint32_t e = 0;
        if (tid < 0) { beg = -1; end = 0; }
        if (tid >= 0 && !(beg <= maxpos && end <= maxpos)) e |= IDX_ERR_MAXPOS;
        if (end < beg) e |= IDX_ERR_ENDBEG;
        if (tid >= n_ref) { if (!grow) e |= IDX_ERR_TIDRANGE; else ensure_ref(tid); }
        if (e) return e;
        const int64_t cb = beg < 0 ? 0 : beg, ce = end <= 0 ? 1 : end;
        const uint32_t bin = idx_reg2bin(tid < 0 ? beg : cb, tid < 0 ? end : ce, g);
        const int64_t bw = cb >> g.min_shift, ew = tid >= 0 ? (ce - 1) >> g.min_shift : -1;
        const bool hp = last.any != 0;
        const bool new_tid = !hp || last.tid != tid;
        const uint64_t vb = last.v;
        if (hp && !new_tid && tid >= 0 && last.beg > beg) e |= IDX_ERR_UNSORTED;      // last.beg: the clamped begin of a placed row (hts_idx_push's last_coor, hts.c:2620-2633)
        if (hp && new_tid && tid >= 0 && last.tid < 0) e |= IDX_ERR_NOCOOR;
        if (new_tid || last.bin != bin) runs.push_back({tid, bin, vb});
        if (tid >= 0) {
            if (new_tid) tid_runs[(size_t)tid]++;
            std::vector<uint64_t> &l = lin[(size_t)tid];
            if ((int64_t)l.size() < ew + 1) l.resize((size_t)ew + 1, ~0ull);
            int64_t w0 = bw; if (!new_tid && last.e >= w0) w0 = last.e + 1;
            for (int64_t w = w0; w <= ew; w++) if (vb < l[(size_t)w]) l[(size_t)w] = vb;
            (mapped ? nmap : nunmap)[(size_t)tid]++;
        } else n_nocoor++;
        last.any = 1; last.tid = tid; last.bin = bin; last.beg = tid >= 0 ? cb : beg; last.e = ew; last.v = vafter;
        return e;
    }
    // A batch of rows reports like the device passes do: an interval violation anywhere in the batch comes before an order violation.
    struct BatchErr { uint32_t row = 0, order = 0; };
    inline void note(BatchErr &b, uint32_t e) { b.row |= e & IDX_ROW_ERRS; b.order |= e & ~IDX_ROW_ERRS; }
    bool batch_ok(const BatchErr &b) { if (b.row) { err = err_text(b.row); return false; } if (b.order) { err = err_text(b.order); return false; } return true; }
    int64_t maxpos() const { return 1ll << (g.min_shift + 3 * g.n_lvls); }
    // the same passes as bam_index_rows over host arrays (text formats): every row [beg, end) 0-based on sequence tid (< 0: unplaced),
    // vafter = the virtual offset behind it
    bool add_rows(const int32_t *tid, const int64_t *beg_in, const int64_t *end_in, const uint64_t *vafter, const uint8_t *mapped, int64_t n) {
        BatchErr be; const int64_t mp = maxpos();
        for (int64_t i = 0; i < n; i++) { const uint32_t e = push_row(tid[i], beg_in[i], end_in[i], vafter[i], mapped[i] != 0, mp); if (e) { note(be, e); if (e & IDX_ROW_ERRS) break; } }
        if (be.row) for (int64_t i = 0; i < n; i++) {                  // (the whole batch's interval violations decide the message)
            int64_t b = beg_in[i], e2 = end_in[i]; if (tid[i] < 0) { b = -1; e2 = 0; }
            if (tid[i] >= 0 && !(b <= mp && e2 <= mp)) be.row |= IDX_ERR_MAXPOS;
            if (e2 < b) be.row |= IDX_ERR_ENDBEG;
            if (tid[i] >= n_ref && !grow) be.row |= IDX_ERR_TIDRANGE;
        }
        return batch_ok(be);
    }
    // run table -> bins.  vfinal: where the reader stands after the last record (the address of the trailing empty block, or the file's end).
    // Array passes, linear in the number of runs.  What they rest on: positions ascend within a sequence (checked by the row passes), so on
    // every level of the binning scheme the bins a sequence's runs name ascend in file order, and so do the runs' offsets; a run [u, v) ends
    // where the next one begins.  The format's bin compaction (deepest level first, a bin whose chunks span fewer than 64 KiB of compressed
    // file joins its parent if the parent has rows of its own) then needs only each bin's extent (first u, last v): a joined bin widens its
    // parent's extent before the parent's own level is judged.  Each piece lands in the bin its chain of joins ends in, in file order, which
    // is the order "sorted by offset" asks for.
    bool finish(uint64_t vfinal) {
        for (int t = 0; t < n_ref; t++) if (tid_runs[(size_t)t] > 1) { err = "Chromosome blocks not continuous"; return false; }
        const uint32_t nb = n_bins(); const int NL = g.n_lvls + 1;
        bins.assign((size_t)n_ref, {}); has.assign((size_t)n_ref, 0); chunks.clear(); chunks.reserve(runs.size() + 2 * (size_t)n_ref + 2);
        uint32_t lvl_first[16]; for (int l = 0; l <= NL && l < 16; l++) lvl_first[l] = (uint32_t)((((uint64_t)1 << (3 * l)) - 1) / 7);
        struct Node { uint32_t id, up, slot; uint64_t lo, hi; };              // up: index of the parent it joins on the level above (~0: it stays)
        std::vector<std::vector<Node>> lv((size_t)NL);
        std::vector<uint32_t> p_node, cnt; std::vector<uint8_t> p_lvl;
        for (size_t a = 0; a < runs.size();) {
            size_t b = a; while (b < runs.size() && runs[b].tid == runs[a].tid) b++;
            const int32_t t = runs[a].tid;
            if (t < 0) { a = b; continue; }
            has[(size_t)t] = 1;
            for (auto &L : lv) L.clear();
            p_node.resize(b - a); p_lvl.resize(b - a);
            for (size_t k = a; k < b; k++) {                       // the distinct bins of every level, their extents
                const uint32_t bin = runs[k].bin; int l = 0; while (l + 1 < NL && bin >= lvl_first[l + 1]) l++;
                const uint64_t u = runs[k].u, v = k + 1 < runs.size() ? runs[k + 1].u : vfinal;
                std::vector<Node> &L = lv[(size_t)l];
                if (L.empty() || L.back().id != bin) {
                    if (bin >= nb || (!L.empty() && L.back().id > bin)) { err = "index build: bins out of order"; return false; }
                    L.push_back({bin, ~0u, 0, u, v});
                } else L.back().hi = v;
                p_node[k - a] = (uint32_t)L.size() - 1; p_lvl[k - a] = (uint8_t)l;
            }
            for (int l = g.n_lvls; l > 0; --l) {                   // joins, deepest level first
                std::vector<Node> &C = lv[(size_t)l], &P = lv[(size_t)l - 1]; size_t j = 0;
                for (Node &x : C) {
                    if ((x.hi >> 16) - (x.lo >> 16) >= 0x10000ull) continue;
                    const uint32_t par = (x.id - 1) >> 3;
                    while (j < P.size() && P[j].id < par) j++;
                    if (j == P.size() || P[j].id != par) continue;
                    x.up = (uint32_t)j; if (x.lo < P[j].lo) P[j].lo = x.lo; if (x.hi > P[j].hi) P[j].hi = x.hi;
                }
            }
            std::vector<Bin> &B = bins[(size_t)t];                 // the bins that stay, ascending; every level's nodes learn their final bin
            for (int l = 0; l < NL; l++) for (Node &x : lv[(size_t)l]) {
                if (x.up != ~0u) { x.slot = lv[(size_t)l - 1][x.up].slot; continue; }
                x.slot = (uint32_t)B.size();
                B.push_back({x.id, 0, 0, 0});
            }
            cnt.assign(B.size() + 1, 0);
            for (size_t k = 0; k < b - a; k++) cnt[lv[p_lvl[k]][p_node[k]].slot + 1]++;
            const size_t base = chunks.size();
            for (size_t i = 0; i < B.size(); i++) { B[i].c0 = (uint32_t)(base + cnt[i]); cnt[i + 1] += cnt[i]; }
            chunks.resize(base + (b - a));
            for (size_t k = a; k < b; k++) {                       // pieces in file order; one that begins in the BGZF block where the bin's previous one ends extends it
                Bin &x = B[lv[p_lvl[k - a]][p_node[k - a]].slot];
                const uint64_t u = runs[k].u, v = k + 1 < runs.size() ? runs[k + 1].u : vfinal;
                if (x.nc && (chunks[x.c0 + x.nc - 1].v >> 16) >= (u >> 16)) { Chunk &m = chunks[x.c0 + x.nc - 1]; if (m.v < v) m.v = v; }
                else chunks[x.c0 + x.nc++] = {u, v};
            }
            // the pseudo-bin: the sequence's extent in the file and its mapped / unmapped counts
            const uint64_t first_u = runs[a].u, last_v = b < runs.size() ? runs[b].u : vfinal;
            B.push_back({nb + 1, 0, (uint32_t)chunks.size(), 2});
            chunks.push_back({first_u, last_v}); chunks.push_back({nmap[(size_t)t], nunmap[(size_t)t]});
            a = b;
        }
        for (int t = 0; t < n_ref; t++) {
            std::vector<uint64_t> &l = lin[(size_t)t];
            for (int64_t k = (int64_t)l.size() - 2; k >= 0; k--) if (l[(size_t)k] == ~0ull) l[(size_t)k] = l[(size_t)k + 1];       // an empty window takes the next one's offset
            if (!has[(size_t)t] || !csi) continue;
            for (Bin &x : bins[(size_t)t]) {                       // a bin's loff: the linear-index entry of its first window
                if (x.id >= nb) continue;
                int lvl = 0; while (lvl + 1 < NL && x.id >= lvl_first[lvl + 1]) lvl++;
                const uint64_t bot = (uint64_t)(x.id - lvl_first[lvl]) << ((g.n_lvls - lvl) * 3);
                x.loff = bot < l.size() ? l[(size_t)bot] : 0;
            }
        }
        return true;
    }
    void save(std::vector<uint8_t> &o) const {
        size_t cap = 64 + aux.size();
        for (int i = 0; i < n_ref; i++) { cap += 16 + lin[(size_t)i].size() * 8; if (has[(size_t)i]) for (const Bin &x : bins[(size_t)i]) cap += 16 + (size_t)x.nc * 16; }
        o.resize(cap);
        uint8_t *w = o.data();
        auto w32 = [&](uint32_t x) { memcpy(w, &x, 4); w += 4; };
        auto w64 = [&](uint64_t x) { memcpy(w, &x, 8); w += 8; };
        auto wch = [&](const Bin &x) { w32(x.nc); memcpy(w, chunks.data() + x.c0, (size_t)x.nc * 16); w += (size_t)x.nc * 16; };
        if (csi) {                                                 // "CSI\1", min_shift, depth, l_aux, aux, n_ref, per sequence: bins with loff; n_no_coor
            memcpy(w, "CSI\1", 4); w += 4;
            w32((uint32_t)g.min_shift); w32((uint32_t)g.n_lvls); w32((uint32_t)aux.size());
            if (!aux.empty()) { memcpy(w, aux.data(), aux.size()); w += aux.size(); }
            w32((uint32_t)n_ref);
            for (int i = 0; i < n_ref; i++) {
                w32(has[(size_t)i] ? (uint32_t)bins[(size_t)i].size() : 0u);
                if (has[(size_t)i]) for (const Bin &x : bins[(size_t)i]) { w32(x.id); w64(x.loff); wch(x); }
            }
            w64(n_nocoor);
            o.resize((size_t)(w - o.data()));
            return;
        }
        memcpy(w, tbi ? "TBI\1" : "BAI\1", 4); w += 4;
        w32((uint32_t)n_ref);
        if (tbi && !aux.empty()) { memcpy(w, aux.data(), aux.size()); w += aux.size(); }
        for (int i = 0; i < n_ref; i++) {
            w32(has[(size_t)i] ? (uint32_t)bins[(size_t)i].size() : 0u);
            if (has[(size_t)i]) for (const Bin &x : bins[(size_t)i]) { w32(x.id); wch(x); }
            w32((uint32_t)lin[(size_t)i].size());
            if (!lin[(size_t)i].empty()) { memcpy(w, lin[(size_t)i].data(), lin[(size_t)i].size() * 8); w += lin[(size_t)i].size() * 8; }
        }
        w64(n_nocoor);
        o.resize((size_t)(w - o.data()));
    }
};
}  // namespace
