// dhts_api.hip -- host side of the thin C ABI (include/duckhts_amd.h): HBM residency, one HIP
// stream per context (= per GPU), kernel orchestration for BGZF index / inflate / BAM unpack.
// Single translation unit with the kernels (no -fgpu-rdc needed).
#include "../../include/duckhts_amd.h"
#include "bgzf_inflate.hip"
#include "bgzf_huff_wave.hip"
#include "bam_records.hip"
#include "bam_tiles_lds.hip"
#include "bam_tile_rows.hip"
#include "bcf_records.hip"
#include "vcf_text.hip"
#include "gzip_serial.hip"
#include "bam_tags.hip"
#include "bgzf_deflate.hip"
#include "hts_index.hip"
#include "bcf_header.h"

#include <errno.h>
#include <time.h>
#include <fcntl.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <sys/stat.h>
#include <unistd.h>
#include <sched.h>
#include <ctype.h>
#include <vector>
#include <map>
#include <algorithm>
#include <limits.h>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <chrono>

#define TILE_BYTES TL_TILE               /* record-stage tile = what bam_tiles_lds.hip stages per wave */
#define PAD_BYTES 256u

// Device memory is pooled for the life of the process, like the pinned host buffers: a scan context allocates ~100 buffers, some of
// them gigabytes (the phase-A scratch), and hipMalloc / hipFree of those costs up to hundreds of milliseconds and synchronises the
// device.  A long-lived host (DuckDB) creates one context per query, so buffers go back to a per-device free list when a context is
// destroyed and the next one picks them up.  The list is bounded (DHTS_POOL_GB, default 96 GB per process); beyond it memory is freed.
namespace {
// A pooled buffer may keep its CONTENTS: the compressed bytes of a file that was staged whole are tagged with the file's identity (path,
// device, inode, size, mtime), and the next context that opens the same unchanged file takes the buffer instead of reading and copying
// the file again -- 288 GB of HBM is the file cache of a long-lived host (DHTS_FILE_CACHE=0 turns it off).  Tagged buffers are the last
// to be reused for something else and the pool evicts its least recently used entries when it is over its limit.
struct PoolBuf { void *p; size_t cap; int dev; std::string tag; uint64_t tag_len; uint64_t stamp; };
std::mutex g_pool_mu;
std::vector<PoolBuf> g_pool;
size_t g_pool_bytes = 0;
uint64_t g_pool_clock = 0;
size_t pool_limit() { static const size_t lim = (size_t)(getenv("DHTS_POOL_GB") ? atof(getenv("DHTS_POOL_GB")) : 96.0) * (size_t)(1u << 30); return lim; }
bool file_cache_on() { static const bool on = !(getenv("DHTS_FILE_CACHE") && atoi(getenv("DHTS_FILE_CACHE")) == 0); return on; }
void *pool_take(size_t n, size_t *cap_out) {
    int dev = 0; if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    int best = -1;
    for (int pass = 0; pass < 2 && best < 0; pass++)             // untagged buffers first; a cached file only when nothing else fits (oldest first)
        for (size_t i = 0; i < g_pool.size(); i++) {
            if (g_pool[i].dev != dev || g_pool[i].cap < n || g_pool[i].cap / 4 > n + 65536 || g_pool[i].tag.empty() != (pass == 0)) continue;
            if (best < 0 || (pass == 0 ? g_pool[i].cap < g_pool[best].cap : g_pool[i].stamp < g_pool[best].stamp)) best = (int)i;
        }
    if (best < 0) return nullptr;
    void *p = g_pool[best].p; *cap_out = g_pool[best].cap; g_pool_bytes -= g_pool[best].cap;
    g_pool[best] = g_pool.back(); g_pool.pop_back();
    return p;
}
// the buffer that holds `tag`'s bytes, if it is idle in the pool
void *pool_take_tagged(const std::string &tag, size_t *cap_out, uint64_t *len_out) {
    int dev = 0; if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (size_t i = 0; i < g_pool.size(); i++) if (g_pool[i].dev == dev && !g_pool[i].tag.empty() && g_pool[i].tag == tag) {
        void *p = g_pool[i].p; *cap_out = g_pool[i].cap; *len_out = g_pool[i].tag_len; g_pool_bytes -= g_pool[i].cap;
        g_pool[i] = g_pool.back(); g_pool.pop_back();
        return p;
    }
    return nullptr;
}
void pool_give(void *p, size_t cap, const std::string &tag = std::string(), uint64_t tag_len = 0) {
    int dev = 0; hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) == hipSuccess) dev = at.device; else (void)hipGetDevice(&dev);
    std::vector<void *> drop;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        if (cap > pool_limit() || g_pool.size() >= 4096) drop.push_back(p);
        else {
            if (!tag.empty()) for (size_t i = 0; i < g_pool.size();) {      // one copy of a file per device
                if (g_pool[i].dev == dev && g_pool[i].tag == tag) { drop.push_back(g_pool[i].p); g_pool_bytes -= g_pool[i].cap; g_pool[i] = g_pool.back(); g_pool.pop_back(); } else i++;
            }
            g_pool.push_back({p, cap, dev, tag, tag_len, ++g_pool_clock}); g_pool_bytes += cap;
            while (g_pool_bytes > pool_limit() && g_pool.size() > 1) {     // over the limit: the least recently used entries go (never the one just given)
                size_t o = 0; for (size_t i = 1; i + 1 < g_pool.size(); i++) if (g_pool[i].stamp < g_pool[o].stamp) o = i;
                if (o + 1 == g_pool.size()) break;
                drop.push_back(g_pool[o].p); g_pool_bytes -= g_pool[o].cap; g_pool[o] = g_pool.back(); g_pool.pop_back();
            }
        }
    }
    for (void *q : drop) (void)hipFree(q);
}
}
static std::atomic<uint64_t> g_malloc_calls{0}, g_malloc_bytes{0}, g_malloc_ns{0};     // hipMalloc calls the pool could not serve (DHTS_TRACE reports them)
extern "C" void dhts_debug_malloc_stats(uint64_t *calls, uint64_t *bytes, double *seconds) { if (calls) *calls = g_malloc_calls; if (bytes) *bytes = g_malloc_bytes; if (seconds) *seconds = 1e-9 * (double)g_malloc_ns; }
// Owning device allocation: returned to the pool by its destructor, so deleting a context gives back every byte of HBM it held.
struct DevBuf {
    void *p = nullptr; size_t cap = 0;
    std::string tag; uint64_t tag_len = 0;      // set when the buffer holds a whole file that may serve the next context (see the pool)
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void swap(DevBuf &o) { void *tp = p; p = o.p; o.p = tp; size_t tc = cap; cap = o.cap; o.cap = tc; tag.swap(o.tag); uint64_t tl = tag_len; tag_len = o.tag_len; o.tag_len = tl; }
    // takes the pooled buffer that holds `t`'s bytes; returns their length, or -1 when it is not there
    int64_t adopt(const std::string &t) {
        size_t got = 0; uint64_t len = 0;
        void *q = pool_take_tagged(t, &got, &len);
        if (!q) return -1;
        release(); p = q; cap = got; tag = t; tag_len = len;
        return (int64_t)len;
    }
    int ensure(size_t n) {
        tag.clear(); tag_len = 0;                // whoever asks for room is about to write: the old contents are nobody's cache any more
        if (n <= cap) return 0;
        release();
        size_t want = n + n / 8 + 4096, got = 0;
        if (void *q = pool_take(want, &got)) { p = q; cap = got; return 0; }
        struct timespec t0_, t1_; clock_gettime(CLOCK_MONOTONIC, &t0_);
        struct Tick { struct timespec &a, &b; size_t w; ~Tick() { clock_gettime(CLOCK_MONOTONIC, &b); g_malloc_calls++; g_malloc_bytes += w; g_malloc_ns += (uint64_t)((b.tv_sec - a.tv_sec) * 1000000000ll + (b.tv_nsec - a.tv_nsec)); } } tick_{t0_, t1_, want};
        if (hipMalloc(&p, want) != hipSuccess) {
            // out of memory: drop what the pool holds on to and try once more
            { std::lock_guard<std::mutex> lk(g_pool_mu); for (auto &b : g_pool) (void)hipFree(b.p); g_pool.clear(); g_pool_bytes = 0; }
            if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; return -1; }
        }
        cap = want; return 0;
    }
    void release() { if (p) pool_give(p, cap, tag, tag_len); p = nullptr; cap = 0; tag.clear(); tag_len = 0; }
};

// progress of a staging run: `frontier` = contiguous bytes of the range that are in HBM (their copies have completed)
struct StageProg {
    std::mutex mu; std::condition_variable cv;
    std::vector<uint8_t> piece_done; uint64_t next_piece = 0, frontier = 0, len = 0; bool finished = false; int rc = 0;
    std::atomic<bool> cancel{false};          // the context is being closed: nobody will read the rest of the file (a LIMIT query, an error)
    void mark(uint64_t pi, uint64_t ch) {
        std::lock_guard<std::mutex> lk(mu);
        piece_done[pi] = 1;
        while (next_piece < piece_done.size() && piece_done[next_piece]) next_piece++;
        const uint64_t f = next_piece * ch; frontier = f < len ? f : len;
        cv.notify_all();
    }
};
struct dhts_ctx;
static void stop_stager(dhts_ctx *c);
struct dhts_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // phase B of the NEXT batch runs on a second stream while this batch's record stage (latency-bound, mostly idle SIMDs) runs on `stream`
    hipStream_t stream_b = nullptr; hipEvent_t pf_done = nullptr;
    // overlapped read-back (dhts_bam_batch_fetch_begin / _wait): a batch's columns are gathered into one of two snapshots on the scan
    // stream and leave for the host on a copy stream while the next batch is computed
    hipStream_t copy_stream = nullptr; hipEvent_t ev_snap[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr}; DevBuf snap[2];
    struct Prefetch { bool valid = false; int64_t b0 = 0, nb = 0; uint64_t carry = 0; int ucur = 0; } pf;
    std::string err;
    // resident compressed bytes
    DevBuf comp; uint64_t comp_len = 0; uint64_t file_off = 0, file_size = 0;   // resident bytes = file bytes [file_off, file_off + comp_len)
    // dhts_open_path_shard: resident bytes = file[0, seg_split) ++ file[seg_file_off, ...): the header blocks, then this rank's window
    uint64_t seg_split = 0, seg_file_off = 0; bool partial_tail = false; uint64_t hdr_bytes_known = 0;
    // dhts_open_path_segments: resident bytes = a concatenation of file ranges (the header blocks, then the index windows of a region
    // query), each made of whole BGZF blocks; sorted by file offset, so resident order = file order
    struct Seg { uint64_t res_off, file_off, len; };
    std::vector<Seg> segs;
    const uint8_t *last_bcf_u = nullptr;   // where the records of the last read_bcf batch live (inflated stream or, for text, v_out)
    bool gz_any = false;               // bgunzip: a plain-gzip file is inflated whatever it holds
    bool gz_plain = false, gz_error = false;   // the file is plain (non-BGZF) gzip: gz_out holds what its members inflate to and serves as the uncompressed text; gz_error: the stream ends in an error behind those bytes
    DevBuf gz_out; uint64_t gz_len = 0;
    bool plain_text = false;          // the file is not BGZF: its bytes ARE the stream (text VCF); the "block table" cuts it into 65,280-byte pieces
    bool vcf_text = false;            // read_bcf on VCF text (vcf_text.hip)
    DevBuf v_pos_hi;                   // VCF text: the high words of the batch's 0-based positions (BcfStream::pos_hi)
    DevBuf v_keep, v_endsv; uint64_t proj_gen = 1, keep_gen = 0; int32_t keep_none = 0, fmt_none = 0; bool keep_all = true, fmt_keep_all = true; DevBuf v_fkeep;   // VCF text: the INFO keys the projection reads (VcfArgs::info_keep)
    DevBuf v_cnt, v_base, v_line_off, v_rec_len, v_out, v_ctr, v_undef, v_patch, vd_ctg_off, vd_ctg_bytes, vd_ctg_id, vd_id_off, vd_id_bytes, vd_id_id, vd_id_typ, vd_id_ftyp, vd_ctg_hash, vd_id_hash, v_tok_off, v_tok_bytes, v_tok_bits;
    uint32_t v_undef_cap = 65536, v_patch_cap = 1u << 20;      // entries the device may record per batch; grown (and the pass repeated) when a batch needs more
    uint32_t vd_ctg_hmask = 0, vd_id_hmask = 0;
    bool cache_hit = false; std::string pending_tag;       // the file's bytes came out of the pool (no read, no copy); tag to put on `comp` once staging has succeeded
    // dhts_open_path_async: the file is still arriving; the block table covers the staged prefix and grows (dhts_bgzf_index_staged)
    std::thread stager; StageProg *prog = nullptr; bool growing = false; uint64_t stage_total = 0;
    uint64_t debug_full_len = 0; bool debug_prefix = false;     // dhts_debug_index_prefix (tests)
    // block table
    int64_t n_blocks = 0; int bgzf_status = 0;
    DevBuf coff, clen, isize, uoff, blk_status;
    std::vector<uint64_t> h_coff, h_uoff; std::vector<uint32_t> h_clen, h_isize;
    // inflate scratch
    DevBuf lit, tok, meta;
    DevBuf stg_lit, stg_tok, wave_ctr;             // wave kernel: staging slices of the resident workgroups, block counter
    DevBuf wg_lit[2], wg_tok[2], stg2_lit, stg2_tok;  // the workgroups' own literal / token areas ([1], stg2_*: fused launches on stream_b)
    DevBuf blk_off; bool huff_packed = false;      // packed phase-A scratch: per-block offsets into `lit` (used as the pool)
    int64_t isize_repaired = 0;                    // blocks whose table entry was replaced by their decoded length (isize_repair)
    bool pool_full_pending = false;                // the next phase-A range is the repeat of one that overflowed the packed scratch
    int64_t pool_per_block = 65536 + 4096;         // room per block in the packed scratch (a retry after DHTS_BLK_ERR_SCRATCH uses the full 152 KiB)
    DevBuf sg_cnt, sg_base, sg_cand, sg_hits;      // block discovery scratch
    // index writer
    std::vector<uint8_t> built_index; DevBuf ix_end; BamStream last_stream;   // last_stream: the inflated buffer of the latest batch
    // interval overlap join
    bool ov_active = false; int64_t ov_n = 0;
    DevBuf ov_beg, ov_end, ov_pmax, ov_bmax, ov_id, ov_first, ov_cnt, ov_off, ov_ids;
    int64_t huff_b0 = 0, huff_nb = 0;     // block range whose tokens are in the scratch
    int64_t wave_slots = 0;              // workgroups of the wave kernel the device holds at once (occupancy query, first use)
    int64_t super_blocks = 524288;       // phase A runs ahead over up to this many blocks (1,536 waves are resident at once, six per CU;
                                         // a long launch keeps every SIMD backfilled).  Scratch is 152 KiB per block: see inflate_blocks.
    // inflated stream double buffer (carry moves between them)
    DevBuf ubuf[2]; int ucur = 0; uint64_t carry_len = 0;
    // tiles
    DevBuf t_first, t_end, t_count, t_err, t_rowbase, d_res, d_nfixed, d_bstat, t_recs, t_recs_first;
    DevBuf t2_first, t2_end, t2_count, t2_err;      // second tile table: repair rounds are out of place
    DevBuf t_look; int64_t rows_slots = 0, rows_hint = 0; uint64_t heap_hint[5] = {0, 0, 0, 0, 0};   // bam_tile_rows: look-back records; sizes the last batches needed (+ 1/8)
    // rows
    DevBuf c_rgflag;
    DevBuf rec_off, c_flag, c_pos, c_mapq, c_pnext, c_tlen, c_tid, c_mtid, c_rgidx, c_rgvalid;
    DevBuf l_qname, l_cigar, l_seq, l_qual, l_rg, cig_rel, ncig_eff, rg_rel, alen_qual;
    DevBuf o_qname, o_cigar, o_seq, o_qual, o_rg, scan_partial, scan_total;
    DevBuf a_qname, a_cigar, a_seq, a_qual, a_rg;
    // header
    bool bam_open = false;
    std::vector<std::string> ref_name; std::vector<const char *> ref_name_p; std::vector<uint32_t> ref_len;
    std::string text;
    std::vector<std::string> rg_id, rg_sm; std::vector<char> rg_has_sm; std::vector<const char *> rg_id_p, rg_sm_p;
    uint64_t first_rec_uoff = 0;
    DevBuf d_rg_off, d_rg_bytes;
    // standard-tag columns (row A5)
    std::vector<int32_t> tag_sel; std::vector<dhts_col> tag_out;
    bool aux_on = false, aux_excl_std = false; dhts_aux_map aux_out;
    DevBuf x_excl, x_valid, x_le, x_lp, x_oe, x_op, x_key, x_kind, x_sub, x_payoff, x_payload;
    DevBuf t_codes, t_dir, t_lens, t_offs, t_partial, t_total, t_coldev, t_fixed, t_valid, t_var;
    // region filter (row A11)
    bool rg_active = false, rg_all = false, rg_nocoor = false;
    std::vector<int64_t> rg_beg, rg_end; std::vector<uint32_t> rg_tid_first;
    DevBuf d_rg_beg, d_rg_end, d_rg_first, c_keep, c_rowmap;
    bool rg_empty_window = false;          // the index shows no chunk for the regions: the scan yields nothing
    // the chunk list of a region query (hts_itr_multi_bam / reg2intervals, hts.c:3597-3739, 3299-3354) as disjoint scan windows in file
    // order; the scan walks them one after the other.  scan_end_uoff: rows whose record starts at or behind it belong to a later window.
    struct ScanWin { int64_t b0, b1; uint64_t first_uoff, end_uoff; };
    std::vector<ScanWin> wins; size_t win_cur = 0; uint64_t scan_end_uoff = ~0ull;
    uint64_t scan_first_uoff = 0;          // inflated offset of the first record of a non-speculative scan (header end, or an index chunk start)
    // read_bcf
    bool bcf_open = false; bool bcf_tidy_req = false;
    dhts::BcfHeader bh; dhts::BcfSchema bsch;
    std::vector<dhts_bcf_colinfo> bcf_colinfo; std::vector<const char *> bcf_ctg_p, bcf_dict_p, bcf_smp_p;
    std::vector<int32_t> bcf_proj; std::vector<dhts_bcf_col> bcf_out;
    struct Arena { const uint8_t *p = nullptr; uint64_t n = 0; } bcf_ar[4];      // device arenas of the last batch's columns: validity, fixed payloads, offsets, children / bytes
    bool bcf_rg_active = false, bcf_rg_all = false; int32_t bcf_rg_tid = -1; int64_t bcf_rg_beg = 0, bcf_rg_end = 0;
    // VCF text: a region names a sequence of the tabix index (tbx_name2id), so it is resolved when the index arrives (dhts_bcf_load_index)
    std::vector<uint8_t> idx_cache; uint64_t idx_cache_len = 0, idx_cache_n = 0; const uint8_t *idx_cache_src = nullptr; uint8_t idx_cache_key[128] = {0};   // the last BGZF index, inflated
    bool seq_packed = false; DevBuf seq_chars;
    bool qual_packed = false; DevBuf q_mask, q_pack; uint8_t q_syms[2][16];       // dhts_bam_set_qual_packed: QUAL crosses PCIe as 2- / 4-bit codes of the batch's own alphabet (dhts_fetch.inc)
    DevBuf z_in, z_slots, z_sizes, z_offs, z_out, z_tok;                           // bgzip: raw chunk, per-block slots / sizes / offsets, packed blocks
    bool bcf_rg_pending = false; std::string bcf_rg_tok; int32_t bcf_rg_itid = -1; std::vector<std::string> tbx_names;
    DevBuf b_keep, b_map, b_sel;
    DevBuf d_ctg_ok, d_id_ok, d_info_slot, d_fmt_slot, b_rec_off, b_dir, b_lens, b_offs, b_partial, b_total, b_coldev, b_fixed, b_valid, b_var;
    // scan position
    int64_t shard_b0 = 0, shard_b1 = 0;   // block range of this shard
    int shard_rank = 0, shard_world = 1;
    int64_t next_block = 0; bool stream_done = false; bool first_batch = true;
    int64_t halo_limit = 0;
    // timing
    bool timing = false;
    double k_ms[DHTS_K_COUNT] = {0}; int64_t k_n[DHTS_K_COUNT] = {0};
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> pending;
    std::vector<hipEvent_t> ev_pool;
};

static const bool g_debug = getenv("DHTS_DEBUG") != nullptr;      // read once, not per repair round

static int fail(dhts_ctx *c, const char *fmt, ...) {
    char buf[512]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    if (c) c->err = buf;
    return -1;
}
#define HIPCHK(c, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(c, "%s failed: %s", #call, hipGetErrorString(e_)); } while (0)
#define ENSURE(c, buf, n) do { if ((buf).ensure(n) != 0) return fail(c, "hipMalloc of %zu bytes failed", (size_t)(n)); } while (0)

// ---- kernel timing with HIP events on the context's stream --------------------------------
static hipEvent_t ev_get(dhts_ctx *c) {
    if (!c->ev_pool.empty()) { hipEvent_t e = c->ev_pool.back(); c->ev_pool.pop_back(); return e; }
    hipEvent_t e; (void)hipEventCreate(&e); return e;
}
struct KTimer {
    dhts_ctx *c; int id; hipEvent_t a = nullptr, b = nullptr; hipStream_t s;
    KTimer(dhts_ctx *c_, int id_, hipStream_t s_ = nullptr) : c(c_), id(id_), s(s_ ? s_ : c_->stream) { if (c->timing) { a = ev_get(c); b = ev_get(c); (void)hipEventRecord(a, s); } }
    ~KTimer() { if (c->timing) { (void)hipEventRecord(b, s); c->pending.push_back({id, {a, b}}); } }
};
static void timing_collect(dhts_ctx *c) {
    for (auto &p : c->pending) {
        float ms = 0; (void)hipEventSynchronize(p.second.second);
        if (hipEventElapsedTime(&ms, p.second.first, p.second.second) == hipSuccess) { c->k_ms[p.first] += ms; c->k_n[p.first]++; }
        c->ev_pool.push_back(p.second.first); c->ev_pool.push_back(p.second.second);
    }
    c->pending.clear();
}

static void discard_prefetch(dhts_ctx *c) {
    if (c->pf.valid) { (void)hipStreamSynchronize(c->stream_b); c->pf.valid = false; }
}

extern "C" {

int dhts_abi_version(void) { return DHTS_ABI_VERSION; }

int dhts_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }

// Streams and events of destroyed contexts are kept per device and handed to the next context: creating two streams costs a few
// milliseconds, which is most of what a query on a small file or a narrow region spends in dhts_create.
namespace {
struct StreamSet { int dev; hipStream_t s, sb; hipEvent_t ev; };
struct CopySet { int dev; hipStream_t s; hipEvent_t snap[2], done[2]; };
std::mutex g_cs_mu;
std::vector<CopySet> g_cs;
std::mutex g_ss_mu;
std::vector<StreamSet> g_ss;
}
dhts_ctx *dhts_create(int device_id) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_id < 0 || device_id >= n) return nullptr;
    if (hipSetDevice(device_id) != hipSuccess) return nullptr;
    dhts_ctx *c = new dhts_ctx();
    c->device = device_id;
    {
        std::lock_guard<std::mutex> lk(g_ss_mu);
        for (size_t i = 0; i < g_ss.size(); i++) if (g_ss[i].dev == device_id) {
            c->stream = g_ss[i].s; c->stream_b = g_ss[i].sb; c->pf_done = g_ss[i].ev;
            g_ss[i] = g_ss.back(); g_ss.pop_back();
            return c;                                           // (a pooled set comes from a context that went through everything below)
        }
    }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return nullptr; }
    // the prefetch stream has the LOWEST priority: the record-stage kernels of the current batch (short, latency-bound) should get wave
    // slots as soon as they ask, the next batch's phase B fills what is left
    int pr_lo = 0, pr_hi = 0; (void)hipDeviceGetStreamPriorityRange(&pr_lo, &pr_hi);
    static const char *env_pp = getenv("DHTS_PF_PRIO");      // tuning knob: "hi" / "normal" instead of the lowest priority
    const int pr_b = (env_pp && !strcmp(env_pp, "hi")) ? pr_hi : (env_pp && !strcmp(env_pp, "normal")) ? 0 : pr_lo;
    if (hipStreamCreateWithPriority(&c->stream_b, hipStreamNonBlocking, pr_b) != hipSuccess || hipEventCreateWithFlags(&c->pf_done, hipEventDisableTiming) != hipSuccess) {
        (void)hipStreamDestroy(c->stream); delete c; return nullptr;
    }
    // the LDS-window kernel needs more than the default dynamic LDS limit
    if (hipFuncSetAttribute((const void *)bgzf_lz_resolve, hipFuncAttributeMaxDynamicSharedMemorySize, B_LDS_BYTES_NW) != hipSuccess ||
        hipFuncSetAttribute((const void *)bgzf_huff_decode, hipFuncAttributeMaxDynamicSharedMemorySize, A_LDS_BYTES) != hipSuccess) {
        (void)hipStreamDestroy(c->stream); delete c; return nullptr;     // no gfx950 code object for this device
    }
    (void)hipFuncSetAttribute((const void *)vcf_encode<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, VCF_LDS_BYTES);
    (void)hipFuncSetAttribute((const void *)vcf_encode<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, VCF_LDS_BYTES);
    (void)hipFuncSetAttribute((const void *)vcf_encode<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, VCF_LDS_BYTES);
    (void)hipFuncSetAttribute((const void *)vcf_encode<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, VCF_LDS_BYTES);
    hipLaunchKernelGGL(crc_const_init, dim3(1), dim3(64), 0, c->stream);      // per-device CRC constants (idempotent)
    if (hipStreamSynchronize(c->stream) != hipSuccess) { (void)hipStreamDestroy(c->stream); delete c; return nullptr; }
    return c;
}

void dhts_destroy(dhts_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    stop_stager(c);
    const bool clean = hipStreamSynchronize(c->stream_b) == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess;
    timing_collect(c);
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    if (c->copy_stream) {
        const bool cclean = hipStreamSynchronize(c->copy_stream) == hipSuccess;
        std::lock_guard<std::mutex> lk(g_cs_mu);
        static bool cs_at_exit = false;
        if (!cs_at_exit) { cs_at_exit = true; std::atexit([]() { std::lock_guard<std::mutex> lk2(g_cs_mu); for (auto &x : g_cs) { if (hipSetDevice(x.dev) != hipSuccess) continue; for (int k = 0; k < 2; k++) { (void)hipEventDestroy(x.snap[k]); (void)hipEventDestroy(x.done[k]); } (void)hipStreamDestroy(x.s); } g_cs.clear(); }); }
        if (cclean && g_cs.size() < 32) g_cs.push_back({c->device, c->copy_stream, {c->ev_snap[0], c->ev_snap[1]}, {c->ev_done[0], c->ev_done[1]}});
        else { for (int k = 0; k < 2; k++) { (void)hipEventDestroy(c->ev_snap[k]); (void)hipEventDestroy(c->ev_done[k]); } (void)hipStreamDestroy(c->copy_stream); }
        c->copy_stream = nullptr;
    }
    bool kept = false;
    if (clean) {
        std::lock_guard<std::mutex> lk(g_ss_mu);
        // (pooled streams are destroyed before the HIP runtime shuts down: a process that exits with live streams can hang in the
        //  runtime's teardown when a profiler is attached)
        static bool at_exit = false;
        if (!at_exit) { at_exit = true; std::atexit([]() { std::lock_guard<std::mutex> lk2(g_ss_mu); for (auto &x : g_ss) { if (hipSetDevice(x.dev) != hipSuccess) continue; (void)hipEventDestroy(x.ev); (void)hipStreamDestroy(x.sb); (void)hipStreamDestroy(x.s); } g_ss.clear(); }); }
        if (g_ss.size() < 32) { g_ss.push_back({c->device, c->stream, c->stream_b, c->pf_done}); kept = true; }
    }
    if (!kept) { (void)hipEventDestroy(c->pf_done); (void)hipStreamDestroy(c->stream_b); (void)hipStreamDestroy(c->stream); }
    delete c;                                               // every DevBuf member frees its allocation (device `c->device` is current)
}

const char *dhts_error(const dhts_ctx *c) { return c ? c->err.c_str() : "no context (no MI355X device or code object)"; }

static void stop_stager(dhts_ctx *c) {
    if (c->prog && c->stager.joinable()) c->prog->cancel.store(true);     // whoever closes the context does not want the rest of the file: the readers stop after their current piece
    if (c->stager.joinable()) c->stager.join();
    if (c->prog && !c->pending_tag.empty() && c->prog->finished && c->prog->rc == 0 && c->prog->frontier == c->prog->len) { c->comp.tag = c->pending_tag; c->comp.tag_len = c->prog->len; }
    c->pending_tag.clear();
    delete c->prog; c->prog = nullptr; c->growing = false;
}
static void reset_file_state(dhts_ctx *c) {
    stop_stager(c);
    c->huff_b0 = c->huff_nb = 0; c->file_off = 0; c->file_size = 0; c->seg_split = 0; c->seg_file_off = 0; c->partial_tail = false; c->segs.clear(); c->cache_hit = false; c->gz_plain = c->gz_error = false; c->gz_len = 0; c->plain_text = false; c->vcf_text = false;
    c->n_blocks = 0; c->bgzf_status = 0; c->bam_open = false; c->carry_len = 0; c->next_block = 0; c->stream_done = false; c->first_batch = true;
    c->h_coff.clear(); c->h_clen.clear(); c->h_isize.clear(); c->h_uoff.clear();
}

int dhts_open_host(dhts_ctx *c, const void *bytes, uint64_t n) {
    if (c) discard_prefetch(c);
    if (!c) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    reset_file_state(c);
    ENSURE(c, c->comp, n + PAD_BYTES);
    HIPCHK(c, hipMemcpyAsync(c->comp.p, bytes, n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync((uint8_t *)c->comp.p + n, 0, PAD_BYTES, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->comp_len = n;
    return 0;
}

int dhts_open_tiled(dhts_ctx *c, const void *head, uint64_t n_head, const void *body, uint64_t n_body, int reps, const void *tail, uint64_t n_tail) {
    if (c) discard_prefetch(c);
    if (!c) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    reset_file_state(c);
    uint64_t n = n_head + n_body * (uint64_t)reps + n_tail;
    ENSURE(c, c->comp, n + PAD_BYTES);
    uint8_t *d = (uint8_t *)c->comp.p;
    if (n_head) HIPCHK(c, hipMemcpyAsync(d, head, n_head, hipMemcpyHostToDevice, c->stream));
    if (n_body && reps > 0) {
        HIPCHK(c, hipMemcpyAsync(d + n_head, body, n_body, hipMemcpyHostToDevice, c->stream));
        for (int r = 1; r < reps; r++) HIPCHK(c, hipMemcpyAsync(d + n_head + n_body * (uint64_t)r, d + n_head, n_body, hipMemcpyDeviceToDevice, c->stream));
    }
    if (n_tail) HIPCHK(c, hipMemcpyAsync(d + n_head + n_body * (uint64_t)reps, tail, n_tail, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(d + n, 0, PAD_BYTES, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->comp_len = n;
    return 0;
}

// The rest of the C ABI, in the order it builds on itself (one translation unit: the pieces share the static helpers above)
#include "dhts_staging.inc"
#include "dhts_bgzf.inc"
#include "dhts_bam_open.inc"
#include "dhts_regions.inc"
#include "dhts_bam_tags.inc"
#include "dhts_index_write.inc"
#include "dhts_tools.inc"
#include "dhts_bam_join.inc"
#include "dhts_bam_scan.inc"
#include "dhts_bcf_scan.inc"
#include "dhts_fetch.inc"

}  // extern "C"
