// dhts_api.hip -- host side of the thin C ABI (include/duckhts_amd.h): HBM residency, one HIP
// stream per context (= per GPU), kernel orchestration for BGZF index / inflate / BAM unpack.
// Single translation unit with the kernels (no -fgpu-rdc needed).
#include "../../include/duckhts_amd.h"
#include "bgzf_inflate.hip"
#include "bgzf_huff_wave.hip"
#include "bam_records.hip"
#include "bam_tiles_lds.hip"
#include "bcf_records.hip"
#include "vcf_text.hip"
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
    bool plain_text = false;          // the file is not BGZF: its bytes ARE the stream (text VCF); the "block table" cuts it into 65,280-byte pieces
    bool vcf_text = false;            // read_bcf on VCF text (vcf_text.hip)
    DevBuf v_cnt, v_base, v_line_off, v_rec_len, v_out, v_ctr, v_undef, v_patch, vd_ctg_off, vd_ctg_bytes, vd_ctg_id, vd_id_off, vd_id_bytes, vd_id_id, vd_id_typ, vd_id_ftyp, vd_ctg_hash, vd_id_hash, v_tok_off, v_tok_bytes, v_tok_bits;
    uint32_t v_undef_cap = 65536, v_patch_cap = 1u << 20;      // entries the device may record per batch; grown (and the pass repeated) when a batch needs more
    uint32_t vd_ctg_hmask = 0, vd_id_hmask = 0;
    bool cache_hit = false; std::string pending_tag;       // the file's bytes came out of the pool (no read, no copy); tag to put on `comp` once staging has succeeded
    // dhts_open_path_async: the file is still arriving; the block table covers the staged prefix and grows (dhts_bgzf_index_staged)
    std::thread stager; StageProg *prog = nullptr; bool growing = false; uint64_t stage_total = 0;
    // block table
    int64_t n_blocks = 0; int bgzf_status = 0;
    DevBuf coff, clen, isize, uoff, blk_status;
    std::vector<uint64_t> h_coff, h_uoff; std::vector<uint32_t> h_clen, h_isize;
    // inflate scratch
    DevBuf lit, tok, meta;
    DevBuf stg_lit, stg_tok, wave_ctr;             // wave kernel: staging slices of the resident workgroups, block counter
    DevBuf wg_lit[2], wg_tok[2], stg2_lit, stg2_tok;  // the workgroups' own literal / token areas ([1], stg2_*: fused launches on stream_b)
    DevBuf blk_off; bool huff_packed = false;      // packed phase-A scratch: per-block offsets into `lit` (used as the pool)
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
    (void)hipFuncSetAttribute((const void *)vcf_encode<false>, hipFuncAttributeMaxDynamicSharedMemorySize, VCF_LDS_BYTES);
    (void)hipFuncSetAttribute((const void *)vcf_encode<true>, hipFuncAttributeMaxDynamicSharedMemorySize, VCF_LDS_BYTES);
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
    c->huff_b0 = c->huff_nb = 0; c->file_off = 0; c->file_size = 0; c->seg_split = 0; c->seg_file_off = 0; c->partial_tail = false; c->segs.clear(); c->cache_hit = false; c->plain_text = false; c->vcf_text = false;
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

// ---- pinned host memory, pooled for the life of the process ---------------------------------------------------------------------
// Page-locking is slow (a few GB/s), so staging and read-back buffers are kept and handed out again: a long-lived host (DuckDB)
// pays for them once, not per query.  Portable: usable from every device.
namespace {
struct PinBuf { void *p; size_t cap; bool busy; };
std::mutex g_pin_mu;
std::vector<PinBuf> g_pin;
}
// streams of the staging threads (host -> device copies).  DHTS_STAGE_PRIO: 0 normal (default), 1 highest, 2 lowest -- which hardware queue
// class the copies share (see the copy stream of the read-back)
static hipError_t stage_stream_create(hipStream_t *st) {
    static const int prio = getenv("DHTS_STAGE_PRIO") ? atoi(getenv("DHTS_STAGE_PRIO")) : 0;
    if (prio == 0) return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
    int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    return hipStreamCreateWithPriority(st, hipStreamNonBlocking, prio == 1 ? hi : lo);
}
extern "C" void *dhts_host_alloc(uint64_t n) {
    std::lock_guard<std::mutex> lk(g_pin_mu);
    int best = -1;
    for (size_t i = 0; i < g_pin.size(); i++) if (!g_pin[i].busy && g_pin[i].cap >= n && (best < 0 || g_pin[i].cap < g_pin[best].cap)) best = (int)i;
    if (best >= 0) { g_pin[best].busy = true; return g_pin[best].p; }
    void *p = nullptr; size_t want = (size_t)n + (size_t)n / 8 + 4096;
    if (hipHostMalloc(&p, want, hipHostMallocPortable) != hipSuccess) return nullptr;
    // drop idle buffers that are too small to matter any more (keeps the pool from growing without bound)
    for (size_t i = 0; i < g_pin.size();) { if (!g_pin[i].busy && g_pin[i].cap * 2 <= want && g_pin.size() > 16) { (void)hipHostFree(g_pin[i].p); g_pin.erase(g_pin.begin() + i); } else i++; }
    g_pin.push_back({p, want, true});
    return p;
}
// gives the idle pooled buffers (device and pinned host) back to the driver; buffers in use are untouched
extern "C" void dhts_release_pools(void) {
    { std::lock_guard<std::mutex> lk(g_ss_mu); for (auto &x : g_ss) { (void)hipSetDevice(x.dev); (void)hipEventDestroy(x.ev); (void)hipStreamDestroy(x.sb); (void)hipStreamDestroy(x.s); } g_ss.clear(); }
    { std::lock_guard<std::mutex> lk(g_pool_mu); for (auto &b : g_pool) (void)hipFree(b.p); g_pool.clear(); g_pool_bytes = 0; }
    std::lock_guard<std::mutex> lk(g_pin_mu);
    for (size_t i = 0; i < g_pin.size();) { if (!g_pin[i].busy) { (void)hipHostFree(g_pin[i].p); g_pin.erase(g_pin.begin() + i); } else i++; }
}
// free / total HBM of a device as the driver reports it (what the leak tests and the pool sizing look at)
extern "C" int dhts_device_mem_info(int device, uint64_t *free_bytes, uint64_t *total_bytes) {
    size_t f = 0, t = 0;
    if (hipSetDevice(device) != hipSuccess || hipDeviceSynchronize() != hipSuccess || hipMemGetInfo(&f, &t) != hipSuccess) return -1;
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return 0;
}
extern "C" void dhts_host_free(void *p) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_pin_mu);
    for (auto &b : g_pin) if (b.p == p) { b.busy = false; return; }
}

// file bytes [off, off+len) -> dst[0, len): reader threads pread 8 MiB pieces into their own pair of pinned buffers and queue the
// H2D copies on their own streams, so the page-cache copy (one core moves ~5-10 GB/s) and the PCIe transfer overlap and scale.
static int stage_file_range(dhts_ctx *c, int fd, uint64_t off, uint64_t len, uint8_t *dst, StageProg *prog = nullptr) {
    const size_t CH = 8u << 20;
    const uint64_t npieces = (len + CH - 1) / CH;
    static const int env_thr = getenv("DHTS_READ_THREADS") ? atoi(getenv("DHTS_READ_THREADS")) : 0;
    // (measured on the MI355X host, per 0.85 GB query: 4 readers 75 ms, 8: 80, 16: 120, 32: 180)
    int nthr = env_thr > 0 ? env_thr : 4; if ((uint64_t)nthr > npieces) nthr = (int)npieces; if (nthr < 1) nthr = 1;
    std::atomic<uint64_t> next(0); std::atomic<int> rc(0);
    const int dev = c->device;
    if (prog) { std::lock_guard<std::mutex> lk(prog->mu); prog->piece_done.assign(npieces, 0); prog->next_piece = 0; prog->frontier = 0; prog->len = len; }
    auto worker = [&]() {
        if (hipSetDevice(dev) != hipSuccess) { rc = -1; return; }
        hipStream_t st = nullptr; hipEvent_t ev[2] = {nullptr, nullptr}; void *pin[2] = {dhts_host_alloc(CH), dhts_host_alloc(CH)}; bool used[2] = {false, false};
        uint64_t piece_of[2] = {0, 0};
        if (!pin[0] || !pin[1] || stage_stream_create(&st) != hipSuccess || hipEventCreateWithFlags(&ev[0], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ev[1], hipEventDisableTiming) != hipSuccess) rc = -2;
        int k = 0;
        while (rc == 0) {
            if (prog && prog->cancel.load()) { rc = -9; break; }                      // (the caller is closing the context: stop reading)
            const uint64_t pi = next.fetch_add(1);
            if (pi >= npieces) break;
            const uint64_t o = pi * CH; const size_t want = (size_t)(len - o < CH ? len - o : CH);
            if (used[k]) { (void)hipEventSynchronize(ev[k]); if (prog) prog->mark(piece_of[k], CH); used[k] = false; }
            size_t got = 0;
            while (got < want) { ssize_t r = pread(fd, (char *)pin[k] + got, want - got, (off_t)(off + o + got)); if (r <= 0) { rc = -3; break; } got += (size_t)r; }
            if (rc != 0) break;
            if (hipMemcpyAsync(dst + o, pin[k], want, hipMemcpyHostToDevice, st) != hipSuccess || hipEventRecord(ev[k], st) != hipSuccess) { rc = -4; break; }
            used[k] = true; piece_of[k] = pi; k ^= 1;
            // the other buffer's copy was queued a whole pread ago: it has usually landed, report it now rather than a piece later
            if (prog && used[k] && hipEventQuery(ev[k]) == hipSuccess) { prog->mark(piece_of[k], CH); used[k] = false; }
        }
        if (st) (void)hipStreamSynchronize(st);
        for (int q = 0; q < 2; q++) { if (used[q] && prog && rc == 0) prog->mark(piece_of[q], CH); if (ev[q]) (void)hipEventDestroy(ev[q]); dhts_host_free(pin[q]); }
        if (st) (void)hipStreamDestroy(st);
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nthr; t++) th.emplace_back(worker);
    worker();
    for (auto &t : th) t.join();
    return rc.load();
}

// identity of an unchanged file: what a cached copy of its bytes is valid for
static std::string file_tag(const struct stat &sb) {
    char b[160];
    snprintf(b, sizeof(b), "%llx:%llx:%llu:%lld.%09ld:%lld.%09ld", (unsigned long long)sb.st_dev, (unsigned long long)sb.st_ino, (unsigned long long)sb.st_size,
             (long long)sb.st_mtim.tv_sec, (long)sb.st_mtim.tv_nsec, (long long)sb.st_ctim.tv_sec, (long)sb.st_ctim.tv_nsec);
    return b;
}

// several file ranges in one go (the index windows of a region query): one set of reader threads works through all their pieces
struct StagePiece { uint64_t file_off; size_t len; uint8_t *dst; };
static int stage_file_pieces(dhts_ctx *c, int fd, const std::vector<StagePiece> &ranges) {
    const size_t CH = 8u << 20;
    std::vector<StagePiece> pc;
    for (auto &r : ranges) for (size_t o = 0; o < r.len; o += CH) pc.push_back({r.file_off + o, r.len - o < CH ? r.len - o : CH, r.dst + o});
    if (pc.empty()) return 0;
    static const int env_thr = getenv("DHTS_READ_THREADS") ? atoi(getenv("DHTS_READ_THREADS")) : 0;
    int nthr = env_thr > 0 ? env_thr : 4; if ((size_t)nthr > pc.size()) nthr = (int)pc.size();
    size_t tot = 0; for (auto &x : pc) tot += x.len;
    if (tot <= (1u << 20)) nthr = 1;                            // a sliver: thread start-up would cost more than the copy
    std::atomic<size_t> next(0); std::atomic<int> rc(0);
    const int dev = c->device;
    auto worker = [&]() {
        if (hipSetDevice(dev) != hipSuccess) { rc = -1; return; }
        hipStream_t st = nullptr; hipEvent_t ev[2] = {nullptr, nullptr}; void *pin[2] = {dhts_host_alloc(CH), dhts_host_alloc(CH)}; bool used[2] = {false, false};
        if (!pin[0] || !pin[1] || stage_stream_create(&st) != hipSuccess || hipEventCreateWithFlags(&ev[0], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ev[1], hipEventDisableTiming) != hipSuccess) rc = -2;
        int k = 0;
        while (rc == 0) {
            const size_t pi = next.fetch_add(1);
            if (pi >= pc.size()) break;
            const StagePiece &x = pc[pi];
            if (used[k]) { (void)hipEventSynchronize(ev[k]); used[k] = false; }
            size_t got = 0;
            while (got < x.len) { ssize_t r = pread(fd, (char *)pin[k] + got, x.len - got, (off_t)(x.file_off + got)); if (r <= 0) { rc = -3; break; } got += (size_t)r; }
            if (rc != 0) break;
            if (hipMemcpyAsync(x.dst, pin[k], x.len, hipMemcpyHostToDevice, st) != hipSuccess || hipEventRecord(ev[k], st) != hipSuccess) { rc = -4; break; }
            used[k] = true; k ^= 1;
        }
        if (st) (void)hipStreamSynchronize(st);
        for (int q = 0; q < 2; q++) { if (ev[q]) (void)hipEventDestroy(ev[q]); dhts_host_free(pin[q]); }
        if (st) (void)hipStreamDestroy(st);
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nthr; t++) th.emplace_back(worker);
    worker();
    for (auto &t : th) t.join();
    return rc.load();
}

// htslib hts_open + the reads underneath bgzf_read_block, for a BYTE RANGE of the file: [off, off+len) (len = 0: to the end of the
// file) becomes the context's resident bytes.  A rank of a multi-GPU scan stages only its own block range plus the halo.
int dhts_open_path_range(dhts_ctx *c, const char *path, uint64_t off, uint64_t len) {
    if (!c) return -1;
    discard_prefetch(c);
    int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(c, "cannot open %s", path);
    struct stat sb; if (fstat(fd, &sb) != 0) { close(fd); return fail(c, "cannot stat %s", path); }
    const uint64_t fsize = (uint64_t)sb.st_size;
    if (off > fsize) off = fsize;
    uint64_t n = fsize - off; if (len != 0 && len < n) n = len;
    if (hipSetDevice(c->device) != hipSuccess) { close(fd); return fail(c, "hipSetDevice failed"); }
    reset_file_state(c);
    const bool whole = off == 0 && n == fsize && n > 0 && file_cache_on();
    const std::string tag = whole ? file_tag(sb) : std::string();
    if (whole && c->comp.adopt(tag) == (int64_t)n) {            // the unchanged file is still in HBM from an earlier query
        close(fd);
        c->comp_len = n; c->file_off = 0; c->file_size = fsize; c->cache_hit = true;
        return 0;
    }
    if (c->comp.ensure(n + PAD_BYTES) != 0) { close(fd); return fail(c, "hipMalloc of %llu bytes failed", (unsigned long long)(n + PAD_BYTES)); }
    int rc = n ? stage_file_range(c, fd, off, n, (uint8_t *)c->comp.p) : 0;
    close(fd);
    if (rc) return fail(c, rc == -3 ? "read error on %s" : "staging %s failed", path);
    HIPCHK(c, hipMemsetAsync((uint8_t *)c->comp.p + n, 0, PAD_BYTES, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->comp_len = n; c->file_off = off; c->file_size = fsize;
    if (whole) { c->comp.tag = tag; c->comp.tag_len = n; }
    return 0;
}
int dhts_open_path(dhts_ctx *c, const char *path) { return dhts_open_path_range(c, path, 0, 0); }

static bool host_is_bgzf_header(const uint8_t *p);
// Only the parts of a file a region query needs (htslib seeks to each chunk, hts.c:4320-4607; here they are staged): the header blocks
// file[0, header_bytes) and the windows [beg[k], end[k] + length of the BGZF block at end[k]) from dhts_bam_region_segments, merged where
// they touch, laid out one after the other in file order.  dhts_bgzf_index, dhts_bam_open, dhts_bam_set_regions and dhts_bam_load_index
// follow as for a whole file; the block table maps resident blocks back to file offsets, so virtual offsets stay those of the file.
extern "C" int dhts_open_path_segments(dhts_ctx *c, const char *path, uint64_t header_bytes, const uint64_t *beg, const uint64_t *end, int64_t n) {
    if (!c || n < 0 || (n > 0 && (!beg || !end))) return -1;
    discard_prefetch(c);
    int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(c, "cannot open %s", path);
    struct stat sb; if (fstat(fd, &sb) != 0) { close(fd); return fail(c, "cannot stat %s", path); }
    const uint64_t fsize = (uint64_t)sb.st_size;
    if (header_bytes > fsize) header_bytes = fsize;
    std::vector<std::pair<uint64_t, uint64_t>> rg;              // [first byte, one past the last byte)
    if (header_bytes) rg.push_back({0, header_bytes});
    for (int64_t k = 0; k < n; k++) {
        uint64_t b = beg[k], e = end[k];
        if (b >= fsize) continue;
        if (e == ~0ull || e >= fsize) e = fsize;
        else {
            uint8_t h[18];
            if (e + 18 > fsize || pread(fd, h, 18, (off_t)e) != 18 || !host_is_bgzf_header(h)) { close(fd); return fail(c, "index does not match the file (no BGZF block at offset %llu)", (unsigned long long)e); }
            e += ((uint64_t)h[16] | ((uint64_t)h[17] << 8)) + 1;
            if (e > fsize) e = fsize;
        }
        if (e > b) rg.push_back({b, e});
    }
    std::sort(rg.begin(), rg.end());
    std::vector<std::pair<uint64_t, uint64_t>> mg;
    for (auto &x : rg) { if (!mg.empty() && x.first <= mg.back().second) { if (x.second > mg.back().second) mg.back().second = x.second; } else mg.push_back(x); }
    uint64_t tot = 0; for (auto &x : mg) tot += x.second - x.first;
    if (hipSetDevice(c->device) != hipSuccess) { close(fd); return fail(c, "hipSetDevice failed"); }
    reset_file_state(c);
    if (c->comp.ensure(tot + PAD_BYTES) != 0) { close(fd); return fail(c, "hipMalloc of %llu bytes failed", (unsigned long long)(tot + PAD_BYTES)); }
    uint64_t at = 0;
    std::vector<StagePiece> ranges;
    for (auto &x : mg) {
        ranges.push_back({x.first, (size_t)(x.second - x.first), (uint8_t *)c->comp.p + at});
        c->segs.push_back({at, x.first, x.second - x.first});
        at += x.second - x.first;
    }
    const int rc = stage_file_pieces(c, fd, ranges);
    close(fd);
    if (rc) { c->segs.clear(); return fail(c, rc == -3 ? "read error on %s" : "staging %s failed", path); }
    HIPCHK(c, hipMemsetAsync((uint8_t *)c->comp.p + tot, 0, PAD_BYTES, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->comp_len = tot; c->file_off = 0; c->file_size = fsize; c->hdr_bytes_known = header_bytes;
    c->partial_tail = mg.empty() || mg.back().second < fsize;
    if (c->segs.size() == 1 && c->segs[0].file_off == 0 && c->segs[0].len == fsize) c->segs.clear();      // everything is resident after all
    return 0;
}

// The same for a scan that starts before the file has arrived: staging runs on background threads, dhts_stage_wait reports how many
// contiguous bytes are resident, dhts_bgzf_index_staged (re)builds the block table over that prefix, and dhts_bam_next_batch serves the
// blocks known so far (the stream is not "at its end" while bytes are still arriving).
int dhts_open_path_async(dhts_ctx *c, const char *path) {
    if (!c) return -1;
    discard_prefetch(c);
    int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(c, "cannot open %s", path);
    struct stat sb; if (fstat(fd, &sb) != 0) { close(fd); return fail(c, "cannot stat %s", path); }
    const uint64_t n = (uint64_t)sb.st_size;
    if (hipSetDevice(c->device) != hipSuccess) { close(fd); return fail(c, "hipSetDevice failed"); }
    reset_file_state(c);
    const std::string tag = (n > 0 && file_cache_on()) ? file_tag(sb) : std::string();
    if (!tag.empty() && c->comp.adopt(tag) == (int64_t)n) {     // still in HBM from an earlier query: nothing to stage
        close(fd);
        c->prog = new StageProg(); c->prog->len = n; c->prog->frontier = n; c->prog->finished = true;
        c->stage_total = n; c->file_off = 0; c->file_size = n; c->comp_len = 0; c->growing = true; c->cache_hit = true;
        return 0;
    }
    if (c->comp.ensure(n + PAD_BYTES) != 0) { close(fd); return fail(c, "hipMalloc of %llu bytes failed", (unsigned long long)(n + PAD_BYTES)); }
    HIPCHK(c, hipMemsetAsync((uint8_t *)c->comp.p + n, 0, PAD_BYTES, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->prog = new StageProg(); c->prog->len = n; c->stage_total = n; c->file_off = 0; c->file_size = n; c->comp_len = 0; c->growing = true;
    c->pending_tag = tag;
    StageProg *pg = c->prog; uint8_t *dst = (uint8_t *)c->comp.p;
    c->stager = std::thread([c, fd, n, dst, pg]() {
        int rc = n ? stage_file_range(c, fd, 0, n, dst, pg) : 0;
        close(fd);
        std::lock_guard<std::mutex> lk(pg->mu); pg->rc = rc; pg->finished = true; if (rc == 0) pg->frontier = n; pg->cv.notify_all();
    });
    return 0;
}
int64_t dhts_stage_wait(dhts_ctx *c, uint64_t min_bytes, int *done) {
    if (!c || !c->prog) { if (done) *done = 1; return c ? (int64_t)c->comp_len : -1; }
    StageProg *pg = c->prog;
    std::unique_lock<std::mutex> lk(pg->mu);
    pg->cv.wait(lk, [&] { return pg->finished || pg->frontier >= min_bytes; });
    if (pg->finished && pg->rc != 0) { lk.unlock(); return fail(c, "staging failed (read error)"); }
    if (done) *done = pg->finished ? 1 : 0;
    return (int64_t)pg->frontier;
}
static int64_t index_impl(dhts_ctx *c, bool extend);
int64_t dhts_bgzf_index_staged(dhts_ctx *c) {
    if (!c) return -1;
    if (!c->prog) return dhts_bgzf_index(c);
    int done = 0; const int64_t f = dhts_stage_wait(c, 0, &done);
    if (f < 0) return -1;
    const bool extend = c->comp_len > 0 && c->n_blocks > 0;
    c->comp_len = (uint64_t)f; c->partial_tail = !done; c->growing = !done;
    return index_impl(c, extend);
}
int64_t dhts_blocks_ahead(const dhts_ctx *c) { return c ? c->n_blocks - c->next_block : 0; }

uint64_t dhts_resident_bytes(const dhts_ctx *c) { return c ? c->comp_len : 0; }
extern "C" int dhts_resident_from_cache(const dhts_ctx *c) { return c && c->cache_hit ? 1 : 0; }

// ---- scans --------------------------------------------------------------------------------
static int run_scan(dhts_ctx *c, int narr, const uint32_t **in, uint32_t **out32, uint64_t **out64, int64_t n, uint64_t *totals_host) {
    ScanArgs a; memset(&a, 0, sizeof(a));
    int64_t nparts = (n + 1 + SCAN_ITEMS - 1) / SCAN_ITEMS;      // +1: the apply pass also writes off[n]
    if (nparts < 1) nparts = 1;
    ENSURE(c, c->scan_partial, (size_t)narr * (size_t)nparts * 8);
    ENSURE(c, c->scan_total, 8 * 8);
    for (int k = 0; k < narr; k++) {
        a.in[k] = in[k]; a.out32[k] = out32 ? out32[k] : nullptr; a.out64[k] = out64 ? out64[k] : nullptr;
        a.partial[k] = (uint64_t *)c->scan_partial.p + (size_t)k * nparts; a.total[k] = (uint64_t *)c->scan_total.p + k;
    }
    a.n = n; a.narr = narr;
    hipLaunchKernelGGL(scan_reduce, dim3((unsigned)nparts, narr), dim3(256), 0, c->stream, a);
    hipLaunchKernelGGL(scan_partials, dim3(narr), dim3(1024), 0, c->stream, a, nparts);
    hipLaunchKernelGGL(scan_apply, dim3((unsigned)nparts, narr), dim3(256), 0, c->stream, a);
    HIPCHK(c, hipGetLastError());
    if (totals_host) {
        HIPCHK(c, hipMemcpyAsync(totals_host, c->scan_total.p, 8 * narr, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return 0;
}

// ---- BGZF index -------------------------------------------------------------------------------
int64_t dhts_bgzf_index(dhts_ctx *c) { return c ? index_impl(c, false) : -1; }
// extend: the resident bytes have grown since the last call (a file that is still being staged): the table is rebuilt over the longer
// prefix -- the blocks known before keep their numbers, offsets and scratch -- and the scan position is left alone
static int64_t index_impl(dhts_ctx *c, bool extend) {
    HIPCHK(c, hipSetDevice(c->device));
    discard_prefetch(c);
    const int64_t old_nb = c->n_blocks;
    if (!extend) c->huff_b0 = c->huff_nb = 0;
    if (c->comp_len == 0) { c->n_blocks = 0; return 0; }
    if (!extend && !c->growing && c->comp_len >= 16) {
        // not gzip at all but VCF text (hts_detect_format: "##fileformat=VCF"): the bytes are the stream; pieces of 65,280 bytes stand in for blocks
        uint8_t head[16];
        HIPCHK(c, hipMemcpy(head, c->comp.p, 16, hipMemcpyDeviceToHost));
        if (!(head[0] == 0x1f && head[1] == 0x8b) && memcmp(head, "##fileformat=VCF", 16) == 0) {
            const uint64_t P = 65280; const int64_t nb = (int64_t)((c->comp_len + P - 1) / P);
            c->h_coff.resize(nb); c->h_clen.resize(nb); c->h_isize.resize(nb); c->h_uoff.resize(nb + 1);
            for (int64_t i = 0; i < nb; i++) { c->h_coff[i] = (uint64_t)i * P; c->h_uoff[i] = (uint64_t)i * P; const uint64_t l = c->comp_len - (uint64_t)i * P < P ? c->comp_len - (uint64_t)i * P : P; c->h_clen[i] = (uint32_t)l; c->h_isize[i] = (uint32_t)l; }
            c->h_uoff[nb] = c->comp_len;
            ENSURE(c, c->coff, (size_t)nb * 8 + 64); ENSURE(c, c->clen, (size_t)nb * 4 + 64); ENSURE(c, c->isize, (size_t)nb * 4 + 64); ENSURE(c, c->uoff, (size_t)(nb + 1) * 8 + 64); ENSURE(c, c->blk_status, (size_t)(nb + 1) * 4);
            HIPCHK(c, hipMemcpy(c->coff.p, c->h_coff.data(), nb * 8, hipMemcpyHostToDevice)); HIPCHK(c, hipMemcpy(c->clen.p, c->h_clen.data(), nb * 4, hipMemcpyHostToDevice));
            HIPCHK(c, hipMemcpy(c->isize.p, c->h_isize.data(), nb * 4, hipMemcpyHostToDevice)); HIPCHK(c, hipMemcpy(c->uoff.p, c->h_uoff.data(), (nb + 1) * 8, hipMemcpyHostToDevice));
            HIPCHK(c, hipMemset(c->blk_status.p, 0, (size_t)(nb + 1) * 4));
            c->n_blocks = nb; c->bgzf_status = 0; c->plain_text = true;
            c->shard_b0 = 0; c->shard_b1 = nb; c->shard_rank = 0; c->shard_world = 1;
            return nb;
        }
    }
    const uint8_t *d = (const uint8_t *)c->comp.p; const uint64_t n = c->comp_len;
    int64_t nspans = (int64_t)((n + 65535) / 65536);
    DevBuf &cnt = c->sg_cnt, &base = c->sg_base, &cand = c->sg_cand, &hits = c->sg_hits; uint64_t total = 0; int rc = 0;   // kept across calls
    auto cleanup = [&]() {};
    if (cnt.ensure((size_t)nspans * 4 + 64) || base.ensure((size_t)(nspans + 1) * 4 + 64) || hits.ensure((size_t)nspans * SIG_SLOTS * 2 + 64) || c->d_nfixed.ensure(64)) { cleanup(); return fail(c, "hipMalloc failed"); }
    (void)hipMemsetAsync(c->d_nfixed.p, 0, 4, c->stream);
    {
        KTimer t(c, DHTS_K_SIGSCAN);
        hipLaunchKernelGGL(bgzf_sig_count, dim3((unsigned)((nspans * 64 + 255) / 256)), dim3(256), 0, c->stream, d, n, (uint32_t *)cnt.p, (uint16_t *)hits.p, (uint32_t *)c->d_nfixed.p);
    }
    const uint32_t *in[1] = {(const uint32_t *)cnt.p}; uint32_t *o32[1] = {(uint32_t *)base.p};
    rc = run_scan(c, 1, in, o32, nullptr, nspans, &total);
    if (rc) { cleanup(); return -1; }
    int64_t ncand = (int64_t)total;
    bool need_seq = (ncand == 0);
    if (!need_seq) {
        if (cand.ensure((size_t)ncand * 8) || c->coff.ensure((size_t)ncand * 8) || c->clen.ensure((size_t)ncand * 4) || c->isize.ensure((size_t)ncand * 4) ||
            c->d_nfixed.ensure(64)) { cleanup(); return fail(c, "hipMalloc failed"); }
        uint32_t ovf = 0;
        if (hipMemcpyAsync(&ovf, c->d_nfixed.p, 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) { cleanup(); return fail(c, "bgzf index sync failed"); }
        {
            KTimer t(c, DHTS_K_SIGSCAN);
            if (ovf) hipLaunchKernelGGL(bgzf_sig_write, dim3((unsigned)((nspans * 64 + 255) / 256)), dim3(256), 0, c->stream, d, n, (const uint32_t *)base.p, (uint64_t *)c->coff.p);
            else hipLaunchKernelGGL(bgzf_sig_gather, dim3((unsigned)((nspans + 255) / 256)), dim3(256), 0, c->stream, (const uint32_t *)cnt.p, (const uint32_t *)base.p, (const uint16_t *)hits.p, nspans, (uint64_t *)c->coff.p);
        }
        (void)hipMemsetAsync(c->d_nfixed.p, 0, 8, c->stream);
        hipLaunchKernelGGL(bgzf_chain_check, dim3((unsigned)((ncand + 255) / 256)), dim3(256), 0, c->stream, d, n, (const uint64_t *)c->coff.p, ncand,
                           (uint32_t *)c->clen.p, (uint32_t *)c->isize.p, (uint32_t *)c->d_nfixed.p, c->partial_tail ? 1 : 0);
        uint32_t bad[2] = {0, 0};
        if (hipMemcpyAsync(bad, c->d_nfixed.p, 8, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) { cleanup(); return fail(c, "bgzf index sync failed"); }
        need_seq = bad[0] != 0;
        c->n_blocks = ncand - (int64_t)bad[1]; c->bgzf_status = 0;       // (cut candidates are the last ones: chain proof)
    }
    if (need_seq) {
        // unusual / corrupt container: restate htslib's sequential walk on the device
        int64_t cap = (int64_t)(n / 26) + 2;
        if (c->coff.ensure((size_t)cap * 8) || c->clen.ensure((size_t)cap * 4) || c->isize.ensure((size_t)cap * 4) || c->d_res.ensure(64)) { cleanup(); return fail(c, "hipMalloc failed"); }
        hipLaunchKernelGGL(bgzf_chain_walk_seq, dim3(1), dim3(1), 0, c->stream, d, n, (uint64_t *)c->coff.p, (uint32_t *)c->clen.p, (uint32_t *)c->isize.p, cap, (int64_t *)c->d_res.p);
        int64_t res[2] = {0, 0};
        if (hipMemcpyAsync(res, c->d_res.p, 16, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) { cleanup(); return fail(c, "bgzf index sync failed"); }
        c->n_blocks = res[0]; c->bgzf_status = (int)res[1];
        if (c->partial_tail && c->bgzf_status == -2) c->bgzf_status = 0;      // the window ends inside a block: expected, the block belongs to the next window
    }
    cleanup();
    // uoff = exclusive prefix of the recorded ISIZE values (each <= 65,537: isize_placed)
    const int64_t nb = c->n_blocks;
    ENSURE(c, c->uoff, (size_t)(nb + 1) * 8 + 64);
    ENSURE(c, c->blk_status, (size_t)(nb + 1) * 4);
    if (nb > 0) {
        const uint32_t *in2[1] = {(const uint32_t *)c->isize.p}; uint64_t *o64[1] = {(uint64_t *)c->uoff.p};
        uint64_t tot2 = 0;
        if (run_scan(c, 1, in2, nullptr, o64, nb, &tot2)) return -1;
    } else { uint64_t z = 0; HIPCHK(c, hipMemcpy(c->uoff.p, &z, 8, hipMemcpyHostToDevice)); }
    c->h_coff.resize(nb); c->h_clen.resize(nb); c->h_isize.resize(nb); c->h_uoff.resize(nb + 1);
    if (nb) {
        HIPCHK(c, hipMemcpy(c->h_coff.data(), c->coff.p, nb * 8, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(c->h_clen.data(), c->clen.p, nb * 4, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(c->h_isize.data(), c->isize.p, nb * 4, hipMemcpyDeviceToHost));
    }
    HIPCHK(c, hipMemcpy(c->h_uoff.data(), c->uoff.p, (nb + 1) * 8, hipMemcpyDeviceToHost));
    if (extend) {
        if (nb < old_nb) return fail(c, "internal: the block table shrank while the file was being staged");
        if (c->shard_b1 == old_nb) c->shard_b1 = nb;            // a whole-file scan follows the table
        return nb;
    }
    c->shard_b0 = 0; c->shard_b1 = nb; c->shard_rank = 0; c->shard_world = 1;
    return nb;
}

int dhts_bgzf_table(const dhts_ctx *c, uint64_t *coff, uint32_t *clen, uint32_t *isize, int64_t cap) {
    if (!c) return -1;
    int64_t n = c->n_blocks < cap ? c->n_blocks : cap;
    if (coff) memcpy(coff, c->h_coff.data(), n * 8);
    if (clen) memcpy(clen, c->h_clen.data(), n * 4);
    if (isize) memcpy(isize, c->h_isize.data(), n * 4);
    return c->bgzf_status;
}

static BgzfTable dev_table(dhts_ctx *c) {
    BgzfTable t; t.coff = (const uint64_t *)c->coff.p; t.clen = (const uint32_t *)c->clen.p; t.isize = (const uint32_t *)c->isize.p;
    t.uoff = (const uint64_t *)c->uoff.p; t.n = c->n_blocks; return t;
}

// workgroups of the persistent inflate kernels that the device holds at once (occupancy query, first use)
static int wave_slots(dhts_ctx *c) {
    if (c->wave_slots == 0) {
        static const int64_t env_wg = getenv("DHTS_WAVE_WG_PER_CU") ? atoll(getenv("DHTS_WAVE_WG_PER_CU")) : 0;      // tuning knob
        int per_cu = 0, per_cu2 = 0; hipDeviceProp_t pr;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)bgzf_inflate_fused, 64, 0) != hipSuccess || per_cu < 1) per_cu = 8;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu2, (const void *)bgzf_huff_decode_wave, 64, 0) != hipSuccess || per_cu2 < 1) per_cu2 = 8;
        if (per_cu2 > per_cu) per_cu = per_cu2;
        if (hipGetDeviceProperties(&pr, c->device) != hipSuccess) return fail(c, "hipGetDeviceProperties failed");
        if (env_wg > 0) per_cu = (int)env_wg;
        c->wave_slots = (int64_t)per_cu * pr.multiProcessorCount;
    }
    return 0;
}
// phase A over [b0, b0+nb): tokens + literals into the scratch
static int huff_blocks(dhts_ctx *c, int64_t b0, int64_t nb, int force = -1) {      // force: -1 default choice, 0 / 1 lane kernel (all symbols in LDS / far table), 2 / 3 wave kernel into fixed slots / the packed pool
    ENSURE(c, c->meta, (size_t)nb * sizeof(InflateMeta));
    BgzfTable t = dev_table(c);
    // Two kernels fill the scratch (same format, cross-checked by the tests):
    //   wave: one WAVE per BGZF block, table-driven, persistent (bgzf_huff_wave.hip): the product path for every launch size since round 3
    //         (measured on MI355X per 65,536 blocks in launches of 131,072: 9.5 ms; the round-2 version of it took 16.3 ms);
    //   lane: one LANE per block, canonical arithmetic (bgzf_inflate.hip): 13.0 ms per 65,536 blocks in long launches and never less than one
    //         lane's serial decode of a whole block (13-15 ms) in short ones; kept as the cross-check of the wave kernel (DHTS_PHASE_A=lane).
    static const char *env_a = getenv("DHTS_PHASE_A");
    const bool env_lane = force >= 0 ? (force != 2 && force != 3) : (env_a && !strcmp(env_a, "lane"));
    if (!env_lane) {
        // persistent launch: as many workgroups (one wave each) as the device holds at once, each with its own staging slices and block
        // assembly area; the workgroups take blocks from a counter that starts behind the blocks they begin with.  `packed`: the blocks'
        // tokens and literals go to a pool, each block exactly the room it needs (force == 2, the tests' cross-check: fixed slots)
        const bool packed = force < 0 || force == 3;
        if (wave_slots(c)) return -1;
        {   // DHTS_POOL_PER_BLOCK: room per block in the packed scratch (tuning knob; the tests use a tiny value to exercise the retry)
            const int64_t env_ppb = getenv("DHTS_POOL_PER_BLOCK") ? atoll(getenv("DHTS_POOL_PER_BLOCK")) : 0;        // (read per launch: a test sets it)
            if (env_ppb > 0 && c->pool_per_block == 65536 + 4096) c->pool_per_block = env_ppb;
        }
        const int64_t grid = nb < c->wave_slots ? nb : c->wave_slots;
        ENSURE(c, c->stg_lit, (size_t)grid * HW_STAGE_LIT_BYTES + 64);
        ENSURE(c, c->stg_tok, (size_t)grid * HW_STAGE_TOK_WORDS * 4 + 64);
        ENSURE(c, c->wave_ctr, 64);
        HIPCHK(c, hipMemsetD32Async((hipDeviceptr_t)c->wave_ctr.p, (int)grid, 1, c->stream));
        uint64_t pool_cap = 0;
        if (packed) {
            pool_cap = (uint64_t)nb * (uint64_t)c->pool_per_block;
            ENSURE(c, c->lit, pool_cap + 8192);
            ENSURE(c, c->blk_off, (size_t)nb * 8 + 64);
            ENSURE(c, c->wg_lit[0], (size_t)grid * (DHTS_LIT_STRIDE + 64) + 8192);
            ENSURE(c, c->wg_tok[0], (size_t)grid * DHTS_TOK_STRIDE * 4 + 64);
            HIPCHK(c, hipMemsetAsync((uint8_t *)c->wave_ctr.p + 16, 0, 8, c->stream));
        } else {
            ENSURE(c, c->lit, (size_t)nb * DHTS_LIT_STRIDE + 8192);
            ENSURE(c, c->tok, (size_t)nb * DHTS_TOK_STRIDE * 4 + 64);
        }
        c->huff_packed = packed;
        KTimer tm(c, DHTS_K_HUFF);
        hipLaunchKernelGGL(bgzf_huff_decode_wave, dim3((unsigned)grid), dim3(64), 0, c->stream, (const uint8_t *)c->comp.p, t, b0, (int32_t)nb,
                           (uint8_t *)c->lit.p, (uint32_t *)c->tok.p, (InflateMeta *)c->meta.p, (uint8_t *)c->stg_lit.p, (uint32_t *)c->stg_tok.p, (uint32_t *)c->wave_ctr.p,
                           packed ? (uint8_t *)c->lit.p : (uint8_t *)nullptr, (unsigned long long)pool_cap, (unsigned long long *)((uint8_t *)c->wave_ctr.p + 16),
                           (unsigned long long *)c->blk_off.p, (uint8_t *)c->wg_lit[0].p, (uint32_t *)c->wg_tok[0].p);
    } else {
        ENSURE(c, c->lit, (size_t)nb * DHTS_LIT_STRIDE + 8192);
        ENSURE(c, c->tok, (size_t)nb * DHTS_TOK_STRIDE * 4 + 64);
        c->huff_packed = false;
        KTimer tm(c, DHTS_K_HUFF);
        // a launch that six waves per CU can hold at once keeps every symbol in LDS; a longer one runs eight waves per CU
        static const int64_t env_nlo = getenv("DHTS_PHASE_A_NLO") ? atoll(getenv("DHTS_PHASE_A_NLO")) : 0;     // tuning knob: 196 or 288
        const uint32_t nlo = force == 0 ? A_NLO_ALL : force == 1 ? A_NLO_FAR : env_nlo == 196 ? A_NLO_FAR : env_nlo == 288 ? A_NLO_ALL : (nb > 98304 ? A_NLO_FAR : A_NLO_ALL);
        hipLaunchKernelGGL(bgzf_huff_decode, dim3((unsigned)((nb + A_SL - 1) / A_SL)), dim3(64), A_LDS_BYTES_FOR(nlo), c->stream, (const uint8_t *)c->comp.p, t, b0, (int32_t)nb,
                           (uint8_t *)c->lit.p, (uint32_t *)c->tok.p, (InflateMeta *)c->meta.p, nlo);
    }
    HIPCHK(c, hipGetLastError());
    c->huff_b0 = b0; c->huff_nb = nb;
    return 0;
}

// inflate blocks [b0, b0+nb) into `out` (device) so that block b lands at out + (uoff[b] - out_base)
static int launch_lz(dhts_ctx *c, int64_t b0, int64_t nb, uint8_t *out, uint64_t out_base, hipStream_t s) {
    BgzfTable t = dev_table(c);
    {
        KTimer tm(c, DHTS_K_LZ, s);
        hipLaunchKernelGGL(bgzf_lz_resolve, dim3((unsigned)((nb + B_NW - 1) / B_NW)), dim3(64 * B_NW), B_LDS_BYTES_NW, s, (const uint8_t *)c->comp.p, t, b0, (int32_t)nb,
                           (const uint8_t *)c->lit.p, (const uint32_t *)c->tok.p, (const InflateMeta *)c->meta.p, c->huff_b0, out, out_base, (int32_t *)c->blk_status.p,
                           c->huff_packed ? (const unsigned long long *)c->blk_off.p : (const unsigned long long *)nullptr);
    }
    HIPCHK(c, hipGetLastError());
    return 0;
}
// the product path of the inflate stage: ONE persistent launch decodes and resolves blocks [b0, b0+nb) (bgzf_inflate_fused); `alt` selects the
// second set of workgroup areas (a launch on stream_b may run beside one on the main stream)
static int launch_fused(dhts_ctx *c, int64_t b0, int64_t nb, uint8_t *out, uint64_t out_base, hipStream_t s, int alt) {
    if (wave_slots(c)) return -1;
    const int64_t grid = nb < c->wave_slots ? nb : c->wave_slots;
    DevBuf &sl = alt ? c->stg2_lit : c->stg_lit, &stk = alt ? c->stg2_tok : c->stg_tok;
    ENSURE(c, sl, (size_t)grid * HW_STAGE_LIT_BYTES + 64);
    ENSURE(c, stk, (size_t)grid * HW_STAGE_TOK_WORDS * 4 + 64);
    ENSURE(c, c->wg_lit[alt], (size_t)grid * (DHTS_LIT_STRIDE + 64) + 8192);
    ENSURE(c, c->wg_tok[alt], (size_t)grid * DHTS_TOK_STRIDE * 4 + 64);
    ENSURE(c, c->wave_ctr, 64);
    uint32_t *ctr = (uint32_t *)c->wave_ctr.p + (alt ? 8 : 0);
    HIPCHK(c, hipMemsetD32Async((hipDeviceptr_t)ctr, (int)grid, 1, s));
    BgzfTable t = dev_table(c);
    {
        KTimer tm(c, DHTS_K_LZ, s);
        hipLaunchKernelGGL(bgzf_inflate_fused, dim3((unsigned)grid), dim3(64), 0, s, (const uint8_t *)c->comp.p, t, b0, (int32_t)nb,
                           (uint8_t *)c->wg_lit[alt].p, (uint32_t *)c->wg_tok[alt].p, (uint8_t *)sl.p, (uint32_t *)stk.p, ctr, out, out_base, (int32_t *)c->blk_status.p);
    }
    HIPCHK(c, hipGetLastError());
    return 0;
}
static bool inflate_split() {       // default: phase A over the shard into the packed scratch, phase B per batch; DHTS_INFLATE=fused: one launch per batch does both
    static const bool v = !(getenv("DHTS_INFLATE") && !strcmp(getenv("DHTS_INFLATE"), "fused"));
    return v;
}
static int inflate_blocks(dhts_ctx *c, int64_t b0, int64_t nb, uint8_t *out, uint64_t out_base, int64_t ahead_limit) {
    if (nb <= 0) return 0;
    if (c->plain_text) {                                     // nothing to inflate: the pieces are copied to their place in the stream
        HIPCHK(c, hipMemcpyAsync(out + (c->h_uoff[b0] - out_base), (const uint8_t *)c->comp.p + c->h_coff[b0], c->h_uoff[b0 + nb] - c->h_uoff[b0], hipMemcpyDeviceToDevice, c->stream));
        return 0;
    }
    discard_prefetch(c);                                   // (phase A below may reallocate the scratch a prefetched phase B reads)
    if (!inflate_split()) return launch_fused(c, b0, nb, out, out_base, c->stream, 0);
    // (a block whose ISIZE field exceeds 64 KiB is recorded and placed as 65,537 bytes -- isize_placed in bam_records.hip -- and fails
    //  phase B's outlen == ISIZE test like any other block with a wrong ISIZE: the stream ends there, rows before it are kept)
    if (!(b0 >= c->huff_b0 && b0 + nb <= c->huff_b0 + c->huff_nb)) {
        static const int64_t env_super = getenv("DHTS_SUPER_BLOCKS") ? atoll(getenv("DHTS_SUPER_BLOCKS")) : 0;   // tuning knob
        int64_t sb = env_super > 0 ? env_super : c->super_blocks;
        {
            // the token/literal scratch of a super-batch may take at most 45 % of the HBM that is free (or already ours)
            size_t fr = 0, tot = 0;
            if (hipMemGetInfo(&fr, &tot) == hipSuccess) {
                const double per_block = ((double)c->pool_per_block + 8.0 + sizeof(InflateMeta)) * 1.125;
                size_t pooled = 0; { std::lock_guard<std::mutex> lk(g_pool_mu); pooled = g_pool_bytes; }       // (idle buffers of earlier contexts can be taken back)
                const int64_t fit = (int64_t)(0.45 * ((double)fr + (double)pooled + (double)c->lit.cap + (double)c->tok.cap) / per_block);
                if (sb > fit) sb = fit;
            }
            if (sb < 16384) sb = 16384;
        }
        int64_t want = sb > nb ? sb : nb;
        if (b0 + want > ahead_limit) want = ahead_limit - b0;
        if (want < nb) want = nb;
        if (huff_blocks(c, b0, want)) return -1;
    }
    return launch_lz(c, b0, nb, out, out_base, c->stream);
}

// how many blocks phase A may decode ahead of the batch that needs them (default 524,288: a whole 10 GB file in one launch, 67 GB of token /
// literal scratch).  A caller that serves many queries from one process wants a scratch the device pool can keep: 196,608 blocks = two full
// rounds of the 1,536 resident waves, 29 GB.
extern "C" void dhts_set_super_blocks(dhts_ctx *c, int64_t n) { if (c && n >= 16384) c->super_blocks = n; }
// SEQ as the file's 4-bit codes: seq.bytes holds (l + 1) / 2 bytes per row at seq.off, seq.len the number of bases (0: "*"); the batch says so
// in seq_packed.  Saves 18 % of an all-column read-back; the consumer expands with "=ACMGRSVTWYHKDBN" (high nibble first).
extern "C" void dhts_bam_set_seq_packed(dhts_ctx *c, int on) { if (c) c->seq_packed = on != 0; }
int64_t dhts_bgzf_inflate_to_host(dhts_ctx *c, int64_t blk0, int64_t nblk, uint8_t *out, uint64_t cap, int32_t *blk_status) {
    if (!c) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    if (blk0 < 0 || nblk < 0 || blk0 + nblk > c->n_blocks) return fail(c, "block range out of bounds");
    if (nblk == 0) return 0;
    uint64_t base = c->h_uoff[blk0], total = c->h_uoff[blk0 + nblk] - base;
    if (total > cap) return fail(c, "output capacity too small (%llu needed)", (unsigned long long)total);
    ENSURE(c, c->ubuf[0], total + PAD_BYTES);
    HIPCHK(c, hipMemsetAsync(c->ubuf[0].p, 0, total + PAD_BYTES, c->stream));
    std::vector<int32_t> bs((size_t)nblk);
    for (int attempt = 0; ; attempt++) {
        if (inflate_blocks(c, blk0, nblk, (uint8_t *)c->ubuf[0].p, base, blk0 + nblk)) return -1;
        HIPCHK(c, hipMemcpyAsync(bs.data(), (int32_t *)c->blk_status.p + blk0, nblk * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        bool scratch = false;
        for (int64_t k = 0; k < nblk; k++) if (bs[(size_t)k] == DHTS_BLK_ERR_SCRATCH) { scratch = true; break; }
        if (!scratch) break;
        if (attempt > 0) return fail(c, "internal: phase-A scratch exhausted twice");
        c->pool_per_block = (int64_t)DHTS_LIT_STRIDE + (int64_t)DHTS_TOK_STRIDE * 4; c->huff_b0 = c->huff_nb = 0;      // (see batch_begin)
    }
    HIPCHK(c, hipMemcpyAsync(out, c->ubuf[0].p, total, hipMemcpyDeviceToHost, c->stream));
    if (blk_status) memcpy(blk_status, bs.data(), (size_t)nblk * 4);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    timing_collect(c);
    return (int64_t)total;
}

// ---- BAM header (host mirror of bam_hdr_read + the @RG dictionary) ---------------------------
static inline uint32_t hle32(const uint8_t *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
static bool is_alpha(char ch) { return (ch >= 'a' && ch <= 'z') || (ch >= 'A' && ch <= 'Z'); }

// htslib/header.c:995-1075 + 830-893 + 271-318: any malformed line voids the dictionary
static void parse_rg_dict(dhts_ctx *c) {
    c->rg_id.clear(); c->rg_sm.clear(); c->rg_has_sm.clear();
    const char *text = c->text.data(); size_t len = c->text.size();
    bool ok = true;
    if (len < 3) { return; }
    size_t i = 0;
    while (ok && i < len - 3 && text[i] != '\0') {
        const char *h = text + i; size_t rem = len - i;
        if (h[0] != '@' || !is_alpha(h[1]) || !is_alpha(h[2]) || rem < 3 || h[3] == '\n') { ok = false; break; }
        bool is_rg = h[1] == 'R' && h[2] == 'G', is_co = h[1] == 'C' && h[2] == 'O';
        size_t j = 3; std::string id, sm; bool has_id = false, sm_seen = false;
        if (is_co) {
            if (rem == 3 || h[3] != '\t') { ok = false; break; }
            for (j = 4; j < rem && h[j] != '\0' && h[j] != '\n'; j++) {}
        } else {
            do {
                if (j == rem || h[j] != '\t') { ok = false; break; }
                size_t k = ++j;
                while (k < rem && h[k] != '\0' && h[k] != '\n' && h[k] != '\t') k++;
                if (k - j < 3 || h[j + 2] != ':') { ok = false; break; }
                if (is_rg && h[j] == 'I' && h[j + 1] == 'D' && !has_id) { id.assign(h + j + 3, k - j - 3); has_id = true; }
                if (is_rg && h[j] == 'S' && h[j + 1] == 'M' && !sm_seen) { sm.assign(h + j + 3, k - j - 3); sm_seen = true; }
                j = k;
            } while (j < rem && h[j] != '\0' && h[j] != '\n');
            if (!ok) break;
        }
        if (is_rg) {
            if (!has_id) { ok = false; break; }
            bool dup = false; for (auto &e : c->rg_id) if (e == id) dup = true;
            if (!dup) { c->rg_id.push_back(id); c->rg_sm.push_back(sm); c->rg_has_sm.push_back(sm_seen && !sm.empty()); }
        }
        i += j + 1;
    }
    if (!ok) { c->rg_id.clear(); c->rg_sm.clear(); c->rg_has_sm.clear(); }
}

int dhts_bam_open(dhts_ctx *c) {
    if (!c) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->n_blocks <= 0) return fail(c, "Failed to read SAM/BAM/CRAM header");
    // inflate leading blocks until the header is complete
    int64_t k = c->n_blocks < 4 ? c->n_blocks : 4;
    std::vector<uint8_t> h; std::vector<int32_t> bs;
    for (;;) {
        uint64_t total = c->h_uoff[k];
        h.resize(total + 16); bs.resize(k);
        if (dhts_bgzf_inflate_to_host(c, 0, k, h.data(), total, bs.data()) < 0) return -1;
        uint64_t good = total;
        for (int64_t b = 0; b < k; b++) if (bs[b] != 0) { good = c->h_uoff[b]; break; }
        // try to parse
        bool need_more = false, bad = false; uint64_t p = 0;
        auto need = [&](uint64_t nbytes) { if (p + nbytes > good) { if (good == total && k < c->n_blocks) need_more = true; else bad = true; return false; } return true; };
        c->ref_name.clear(); c->ref_len.clear();
        do {
            if (!need(8)) break;
            if (memcmp(h.data(), "BAM\1", 4) != 0) { bad = true; break; }
            uint32_t l_text = hle32(h.data() + 4); p = 8;
            if (!need(l_text)) break;
            c->text.assign((const char *)h.data() + p, l_text); p += l_text;
            if (!need(4)) break;
            int32_t n_ref = (int32_t)hle32(h.data() + p); p += 4;
            if (n_ref < 0) { bad = true; break; }
            for (int32_t i = 0; i < n_ref; i++) {
                if (!need(4)) break;
                int32_t l_name = (int32_t)hle32(h.data() + p); p += 4;
                if (l_name <= 0) { bad = true; break; }
                if (!need((uint64_t)l_name + 4)) break;
                size_t nl = strnlen((const char *)h.data() + p, (size_t)l_name);
                c->ref_name.emplace_back((const char *)h.data() + p, nl); p += (uint64_t)l_name;
                c->ref_len.push_back(hle32(h.data() + p)); p += 4;
            }
        } while (0);
        if (bad) return fail(c, "Failed to read SAM/BAM/CRAM header");
        if (need_more) { k = (k * 4 < c->n_blocks) ? k * 4 : c->n_blocks; continue; }
        c->first_rec_uoff = p; c->scan_first_uoff = p;
        break;
    }
    c->ref_name_p.clear(); for (auto &s : c->ref_name) c->ref_name_p.push_back(s.c_str());
    parse_rg_dict(c);
    c->rg_id_p.clear(); c->rg_sm_p.clear();
    std::vector<uint32_t> off; std::string bytes; off.push_back(0);
    for (size_t i = 0; i < c->rg_id.size(); i++) {
        c->rg_id_p.push_back(c->rg_id[i].c_str()); c->rg_sm_p.push_back(c->rg_has_sm[i] ? c->rg_sm[i].c_str() : nullptr);
        bytes += c->rg_id[i]; off.push_back((uint32_t)bytes.size());
    }
    ENSURE(c, c->d_rg_off, off.size() * 4 + 16); ENSURE(c, c->d_rg_bytes, bytes.size() + 16);
    HIPCHK(c, hipMemcpy(c->d_rg_off.p, off.data(), off.size() * 4, hipMemcpyHostToDevice));
    if (!bytes.empty()) HIPCHK(c, hipMemcpy(c->d_rg_bytes.p, bytes.data(), bytes.size(), hipMemcpyHostToDevice));
    c->bam_open = true;
    return dhts_bam_rewind(c);
}

int dhts_bam_header_get(const dhts_ctx *c, dhts_bam_header *out) {
    if (!c || !c->bam_open) return -1;
    out->n_ref = (int32_t)c->ref_name.size(); out->ref_name = c->ref_name_p.data(); out->ref_len = c->ref_len.data();
    out->text = c->text.data(); out->l_text = (uint32_t)c->text.size();
    out->n_rg = (int32_t)c->rg_id.size(); out->rg_id = c->rg_id_p.data(); out->rg_sm = c->rg_sm_p.data();
    out->first_rec_uoff = c->first_rec_uoff;
    return 0;
}

// contiguous block ranges balanced by compressed bytes; pure host arithmetic, no device needed
int dhts_shard_cut(const uint64_t *coff, int64_t n_blocks, uint64_t comp_len, int rank, int world, int64_t *b0, int64_t *b1) {
    if (!coff || world < 1 || rank < 0 || rank >= world || !b0 || !b1) return -1;
    auto cut = [&](int r) -> int64_t {
        if (r <= 0) return 0;
        if (r >= world) return n_blocks;
        uint64_t target = (uint64_t)((__uint128_t)comp_len * (unsigned)r / (unsigned)world);
        int64_t lo = 0, hi = n_blocks;
        while (lo < hi) { int64_t mid = (lo + hi) / 2; if (coff[mid] < target) lo = mid + 1; else hi = mid; }
        return lo;
    };
    *b0 = cut(rank); *b1 = cut(rank + 1);
    return 0;
}

int dhts_bam_set_block_range(dhts_ctx *c, int64_t b0, int64_t b1, int speculative_start) {
    if (!c || b0 < 0 || b1 < b0 || b1 > c->n_blocks) return -1;
    c->wins.clear(); c->scan_end_uoff = ~0ull;
    c->shard_b0 = b0; c->shard_b1 = b1; c->shard_rank = speculative_start ? 1 : 0; c->shard_world = (b1 < c->n_blocks || speculative_start) ? 2 : 1;
    c->scan_first_uoff = c->first_rec_uoff;
    return dhts_bam_rewind(c);
}

// ---- one file, several GPUs: every rank stages only its own byte window --------------------------------------------------------------
// Cut points: t_0 = 0, t_r = H + (size - H) * r / world for r >= 1 (H = bytes of the header blocks), so rank 0 always owns the header
// and the first records; a block belongs to the rank whose [t_r, t_r+1) holds its first byte -- the rule of dhts_shard_cut applied to
// file offsets.  Rank r > 0 finds the first block start at or behind t_r on the host (BGZF signature + three chained hops), and its
// resident bytes become  file[0, H) ++ file[start_r, t_r+1 + halo) : the header blocks (every rank needs the dictionaries) followed
// by its window and a halo in which the last record of the window completes.
#define DHTS_SHARD_HALO (4u << 20)
static uint64_t shard_target(uint64_t fsize, uint64_t hdr, int r, int world) {
    if (r <= 0) return 0;
    if (r >= world) return fsize;
    if (hdr > fsize) hdr = fsize;
    return hdr + (uint64_t)((__uint128_t)(fsize - hdr) * (unsigned)r / (unsigned)world);
}
static bool host_is_bgzf_header(const uint8_t *p) {
    return p[0] == 31 && p[1] == 139 && p[2] == 8 && (p[3] & 4) && p[10] == 6 && p[11] == 0 && p[12] == 'B' && p[13] == 'C' && p[14] == 2 && p[15] == 0;
}
// the byte window [wbeg, wend) of rank `rank` (host only: signature probe on the open file); t1 = where the rank's ownership ends
static void shard_window_fd(int fd, uint64_t fsize, uint64_t header_bytes, int rank, int world, uint64_t *wbeg_o, uint64_t *wend_o, uint64_t *t1_o) {
    if (header_bytes > fsize) header_bytes = fsize;
    const uint64_t t0 = shard_target(fsize, header_bytes, rank, world), t1 = shard_target(fsize, header_bytes, rank + 1, world);
    uint64_t wend = t1 + DHTS_SHARD_HALO; if (wend > fsize || rank == world - 1) wend = fsize;
    uint64_t wbeg = 0;
    if (rank > 0) {
        // first block start in [t0, ...): probe on the host
        wbeg = fsize;
        const size_t PROBE = 1u << 20;
        std::vector<uint8_t> buf(PROBE + 32);
        for (uint64_t base = t0; base < fsize && wbeg == fsize; base += PROBE - 65536 - 18) {
            const size_t want = (size_t)(fsize - base < PROBE ? fsize - base : PROBE);
            size_t got = 0;
            while (got < want) { ssize_t r = pread(fd, buf.data() + got, want - got, (off_t)(base + got)); if (r <= 0) break; got += (size_t)r; }
            if (got < 18) break;
            for (size_t q = 0; q + 18 <= got && q < PROBE - 65536 - 18; q++) {
                if (!host_is_bgzf_header(buf.data() + q)) continue;
                // three hops must land on headers (or exactly on the end of the file)
                size_t o = q; int hops = 0; bool ok = true;
                while (hops < 3) {
                    const size_t bl = ((size_t)buf[o + 16] | ((size_t)buf[o + 17] << 8)) + 1;
                    if (bl < 26) { ok = false; break; }
                    o += bl;
                    if (base + o == fsize) break;
                    if (o + 18 > got) { ok = (base + o < fsize) && hops >= 1; break; }      // ran out of probe bytes: accept after at least one verified hop
                    if (!host_is_bgzf_header(buf.data() + o)) { ok = false; break; }
                    hops++;
                }
                if (ok) { wbeg = base + q; break; }
            }
            if (got < want) break;
        }
        if (wbeg >= wend) wbeg = wend = fsize > 0 ? fsize : 0;          // no block starts in this rank's range: it scans nothing
    }
    *wbeg_o = wbeg; *wend_o = wend; *t1_o = t1;
}
// host-only view of the cut (no device needed): rank r stages file[win_begin, win_end) behind the header blocks and owns the blocks
// that start in [win_begin, own_end)
extern "C" int dhts_shard_window(const char *path, int rank, int world, uint64_t header_bytes, uint64_t *win_begin, uint64_t *win_end, uint64_t *own_end) {
    if (!path || world < 1 || rank < 0 || rank >= world) return -1;
    int fd = open(path, O_RDONLY);
    if (fd < 0) return -1;
    struct stat sb; if (fstat(fd, &sb) != 0) { close(fd); return -1; }
    uint64_t a = 0, b = 0, t1 = 0;
    shard_window_fd(fd, (uint64_t)sb.st_size, header_bytes, rank, world, &a, &b, &t1);
    close(fd);
    if (win_begin) *win_begin = a;
    if (win_end) *win_end = b;
    if (own_end) *own_end = t1;
    return 0;
}
int dhts_open_path_shard(dhts_ctx *c, const char *path, int rank, int world, uint64_t header_bytes) {
    if (!c || world < 1 || rank < 0 || rank >= world) return -1;
    discard_prefetch(c);
    int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(c, "cannot open %s", path);
    struct stat sb; if (fstat(fd, &sb) != 0) { close(fd); return fail(c, "cannot stat %s", path); }
    const uint64_t fsize = (uint64_t)sb.st_size;
    if (header_bytes > fsize) header_bytes = fsize;
    uint64_t wbeg = 0, wend = 0, t1_ = 0;
    shard_window_fd(fd, fsize, header_bytes, rank, world, &wbeg, &wend, &t1_);
    if (hipSetDevice(c->device) != hipSuccess) { close(fd); return fail(c, "hipSetDevice failed"); }
    reset_file_state(c);
    const uint64_t n_hdr = rank > 0 ? header_bytes : 0, n_win = wend > wbeg ? wend - wbeg : 0, n = n_hdr + n_win;
    if (c->comp.ensure(n + PAD_BYTES) != 0) { close(fd); return fail(c, "hipMalloc of %llu bytes failed", (unsigned long long)(n + PAD_BYTES)); }
    int rc = 0;
    if (n_hdr) rc = stage_file_range(c, fd, 0, n_hdr, (uint8_t *)c->comp.p);
    if (rc == 0 && n_win) rc = stage_file_range(c, fd, wbeg, n_win, (uint8_t *)c->comp.p + n_hdr);
    close(fd);
    if (rc) return fail(c, rc == -3 ? "read error on %s" : "staging %s failed", path);
    HIPCHK(c, hipMemsetAsync((uint8_t *)c->comp.p + n, 0, PAD_BYTES, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->comp_len = n; c->file_off = 0; c->file_size = fsize;
    c->seg_split = n_hdr; c->seg_file_off = rank > 0 ? wbeg : 0; c->partial_tail = wend < fsize; c->hdr_bytes_known = header_bytes;
    return 0;
}
// file offset of resident block i (the two-segment layout of dhts_open_path_shard; identity otherwise)
static uint64_t block_file_off(const dhts_ctx *c, int64_t i) {
    const uint64_t o = i < c->n_blocks ? c->h_coff[i] : c->comp_len;
    if (!c->segs.empty()) {
        size_t lo = 0, hi = c->segs.size();                   // last segment with res_off <= o
        while (hi - lo > 1) { const size_t mid = (lo + hi) / 2; if (c->segs[mid].res_off <= o) lo = mid; else hi = mid; }
        return o - c->segs[lo].res_off + c->segs[lo].file_off;
    }
    return (c->seg_split && o >= c->seg_split) ? o - c->seg_split + c->seg_file_off : o + c->file_off;
}
int dhts_bam_set_file_shard(dhts_ctx *c, int rank, int world) {
    if (!c || !c->bam_open || world < 1 || rank < 0 || rank >= world) return -1;
    const uint64_t t1 = shard_target(c->file_size, c->hdr_bytes_known, rank + 1, world);
    int64_t b0 = 0;
    if (rank > 0) { while (b0 < c->n_blocks && c->h_coff[b0] < c->seg_split) b0++; }      // behind the header blocks
    int64_t lo = b0, hi = c->n_blocks;
    if (rank == world - 1) lo = c->n_blocks;
    else while (lo < hi) { const int64_t mid = (lo + hi) / 2; if (block_file_off(c, mid) < t1) lo = mid + 1; else hi = mid; }
    return dhts_bam_set_block_range(c, b0, lo, rank > 0);
}
// compressed bytes of the blocks that hold the header: what every rank of a multi-GPU scan stages in front of its window
uint64_t dhts_bam_header_bytes(const dhts_ctx *c) {
    if (!c || !c->bam_open || c->n_blocks <= 0) return 0;
    int64_t k = 0;
    while (k + 1 < c->n_blocks && c->h_uoff[k + 1] < c->first_rec_uoff) k++;
    return c->h_coff[k] + c->h_clen[k];
}
// BGZF virtual offset (file offset of the block << 16 | offset inside its inflated payload; htslib bgzf.h bgzf_tell) of a position of
// the inflated stream as this context numbers it: what adjacent ranks compare at a shard boundary (their uoff numbering differs)
uint64_t dhts_voffset(const dhts_ctx *c, uint64_t uoff) {
    if (!c || c->n_blocks <= 0) return 0;
    int64_t lo = 0, hi = c->n_blocks;                       // smallest i with h_uoff[i + 1] > uoff
    while (lo < hi) { const int64_t mid = (lo + hi) / 2; if (c->h_uoff[mid + 1] > uoff) hi = mid; else lo = mid + 1; }
    if (lo >= c->n_blocks) return block_file_off(c, c->n_blocks) << 16;
    return (block_file_off(c, lo) << 16) | (uoff - c->h_uoff[lo]);
}

int dhts_bam_set_shard(dhts_ctx *c, int rank, int world) {
    if (!c) return -1;
    int64_t b0, b1;
    if (dhts_shard_cut(c->h_coff.data(), c->n_blocks, c->comp_len, rank, world, &b0, &b1)) return -1;
    return dhts_bam_set_block_range(c, b0, b1, rank > 0);
}

// a scan that starts at the top of the file begins with the block that holds the first record (headers may span many blocks)
static void skip_header_blocks(dhts_ctx *c) {
    if (c->shard_rank != 0 || c->n_blocks <= 0) return;
    // the first block that ends behind scan_first_uoff; blocks that declare no bytes (ISIZE 0) at that position are not skipped: a damaged
    // ISIZE of 0 on the block that holds the first record has to be seen by the inflate stage (round-2 soak, seed 2001161)
    int64_t lo = 0, hi = c->n_blocks;
    while (lo < hi) { int64_t mid = (lo + hi) / 2; if (c->h_uoff[mid + 1] > c->scan_first_uoff) hi = mid; else lo = mid + 1; }
    while (lo > 0 && c->h_uoff[lo - 1] >= c->scan_first_uoff && c->h_isize[lo - 1] == 0) lo--;
    if (lo >= c->n_blocks) { lo = c->n_blocks - 1; while (lo > 0 && c->h_uoff[lo] >= c->scan_first_uoff && c->h_uoff[lo - 1] >= c->scan_first_uoff) lo--; }
    if (lo >= c->shard_b0 && lo < c->shard_b1) c->next_block = lo;
}

// ---- region queries: host side = region strings -> merged intervals, BAI -> scan window -------------------
// hts_parse_decimal with HTS_PARSE_THOUSANDS_SEP (htslib hts.c:3884-3940)
static long long parse_decimal_sep(const char *str, const char **strend) {
    unsigned long long n = 0; int digits = 0, decimals = 0, e = 0; char sign = '+', esign = '+';
    const char *s = str;
    while (*s == ' ' || (*s >= '\t' && *s <= '\r')) s++;
    if (*s == '+' || *s == '-') sign = *s++;
    while (*s) { if (*s >= '0' && *s <= '9') { digits++; n = n * 10 + (unsigned)(*s++ - '0'); } else if (*s == ',') s++; else break; }
    if (*s == '.') { s++; while (*s >= '0' && *s <= '9') { decimals++; digits++; n = n * 10 + (unsigned)(*s++ - '0'); } }
    switch (*s) {
    case 'e': case 'E': s++; if (*s == '+' || *s == '-') esign = *s++; while (*s >= '0' && *s <= '9') e = e * 10 + (*s++ - '0'); if (esign == '-') e = -e; break;
    case 'k': case 'K': e += 3; s++; break;
    case 'm': case 'M': e += 6; s++; break;
    case 'g': case 'G': e += 9; s++; break;
    }
    e -= decimals;
    while (e > 0) { n *= 10; e--; }
    while (e < 0) { n /= 10; e++; }
    *strend = digits > 0 ? s : str;
    return sign == '+' ? (long long)n : -(long long)n;
}

// hts_parse_region (hts.c:3995-4150) for one region token; name lookup through `names`.  Returns false when the token does not
// name a known reference / is malformed (hts_reglist_create then skips it with a warning, region.c:203-215).
static bool parse_region_token(const std::vector<std::string> &names, const std::string &tok, int &tid, int64_t &beg, int64_t &end) {
    auto getid = [&](const std::string &nm) -> int { for (size_t i = 0; i < names.size(); i++) if (names[i] == nm) return (int)i; return -1; };
    const int64_t POS_MAX = ((((int64_t)INT32_MAX) << 32) | 0xffffffffll);       // HTS_POS_MAX = INT64_MAX in htslib >= 1.10
    (void)POS_MAX;
    const int64_t PMAX = INT64_MAX;
    std::string s = tok; bool quoted = false; size_t colon = std::string::npos;
    if (!s.empty() && s[0] == '{') {
        size_t close = s.find('}');
        if (close == std::string::npos) return false;
        std::string name = s.substr(1, close - 1);
        quoted = true;
        if (close + 1 < s.size() && s[close + 1] == ':') colon = close + 1;
        if (colon == std::string::npos) { beg = 0; end = PMAX; tid = getid(name); return tid >= 0; }
        tid = getid(name);
        if (tid < 0) return false;
    } else {
        colon = s.rfind(':');
        if (colon == std::string::npos) { beg = 0; end = PMAX; tid = getid(s); return tid >= 0; }
        beg = 0; end = PMAX;
        if ((tid = getid(s)) >= 0) return getid(s.substr(0, colon)) < 0;       // whole string is a name; ambiguous if the prefix is one too
        tid = getid(s.substr(0, colon));
        if (tid < 0) return false;
    }
    (void)quoted;
    const char *c1 = s.c_str() + colon + 1, *hy = nullptr;
    beg = parse_decimal_sep(c1, &hy) - 1;
    if (beg < 0) {
        if (beg != -1 && *hy == '-' && *c1 != '\0') return false;               // "Coordinates must be > 0"
        if ((*hy >= '0' && *hy <= '9') || *hy == '\0' || *hy == ',') { end = beg == -1 ? PMAX : -(beg + 1); beg = 0; return true; }   // chr:-100 = chr:1-100
        else if (beg < -1) return false;
    }
    if (*hy == '\0') end = PMAX;
    else if (*hy == '-') { const char *h2; end = parse_decimal_sep(hy + 1, &h2); if (*h2 != '\0' && *h2 != ',') return false; }
    else return false;
    if (end == 0) end = PMAX;
    if (beg >= end) return false;
    return true;
}

// read_bam(region := ...): comma split as src/bam_reader.c:318-348 (strtok: empty tokens vanish), then hts_reglist_create
// (region.c:177-260: "." = everything, "*" = unplaced reads, unknown names skipped, intervals sorted and merged per tid).
// Returns 0, or 1 when no usable region remains (the reference reports "No reads found for region(s): ...").
int dhts_bam_set_regions(dhts_ctx *c, const char *regions) {
    if (!c || !c->bam_open) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    c->rg_active = false; c->rg_all = false; c->rg_nocoor = false; c->rg_empty_window = false; c->rg_beg.clear(); c->rg_end.clear(); c->rg_tid_first.clear();
    c->wins.clear(); c->win_cur = 0; c->scan_end_uoff = ~0ull;
    c->scan_first_uoff = c->first_rec_uoff; c->shard_b0 = 0; c->shard_b1 = c->n_blocks; c->shard_rank = 0; c->shard_world = 1;
    if (!regions || !*regions) return dhts_bam_rewind(c);
    const size_t n_ref = c->ref_name.size();
    std::vector<std::vector<std::pair<int64_t, int64_t>>> per(n_ref);
    int usable = 0, ntok = 0;
    std::string all(regions); size_t p = 0;
    while (p <= all.size()) {
        size_t q = all.find(',', p); if (q == std::string::npos) q = all.size();
        std::string tok = all.substr(p, q - p); p = q + 1;
        if (tok.empty()) continue;
        ntok++;
        if (tok == ".") { c->rg_all = true; usable++; continue; }
        if (tok == "*") { c->rg_nocoor = true; usable++; continue; }
        int tid; int64_t b, e;
        if (!parse_region_token(c->ref_name, tok, tid, b, e)) continue;
        per[tid].push_back({b, e}); usable++;
    }
    if (ntok == 0) return dhts_bam_rewind(c);               // a string of commas only: the reference's strtok split yields no region at all (bam_reader.c:319-345) => plain scan
    if (!usable) return 1;
    c->rg_tid_first.assign(n_ref + 1, 0);
    for (size_t t = 0; t < n_ref; t++) {
        auto &v = per[t];
        std::sort(v.begin(), v.end());
        c->rg_tid_first[t] = (uint32_t)c->rg_beg.size();
        for (size_t j = 0; j < v.size(); j++) {
            if (!c->rg_beg.empty() && c->rg_beg.size() > c->rg_tid_first[t] && !(c->rg_end.back() < v[j].first)) { if (c->rg_end.back() < v[j].second) c->rg_end.back() = v[j].second; }
            else { c->rg_beg.push_back(v[j].first); c->rg_end.push_back(v[j].second); }
        }
    }
    c->rg_tid_first[n_ref] = (uint32_t)c->rg_beg.size();
    ENSURE(c, c->d_rg_beg, c->rg_beg.size() * 8 + 16); ENSURE(c, c->d_rg_end, c->rg_end.size() * 8 + 16); ENSURE(c, c->d_rg_first, (n_ref + 1) * 4 + 16);
    if (!c->rg_beg.empty()) { HIPCHK(c, hipMemcpy(c->d_rg_beg.p, c->rg_beg.data(), c->rg_beg.size() * 8, hipMemcpyHostToDevice)); HIPCHK(c, hipMemcpy(c->d_rg_end.p, c->rg_end.data(), c->rg_end.size() * 8, hipMemcpyHostToDevice)); }
    HIPCHK(c, hipMemcpy(c->d_rg_first.p, c->rg_tid_first.data(), (n_ref + 1) * 4, hipMemcpyHostToDevice));
    c->rg_active = true;
    return dhts_bam_rewind(c);
}

// ---- index -> scan window ----------------------------------------------------------------------------------------------------
// BAI (SAM spec 5.2) and CSI (CSIv1; BGZF-compressed, inflated here on the GPU through a scratch context) as loaded by
// hts_idx_load (htslib hts.c:2920-3055).  For a set of query intervals the window is [smallest chunk start, largest chunk end] over
// the bins the intervals touch (reg2bins hts.c:3142-3213, generalised to min_shift / depth), pruned by the BAI linear index
// (hts.c:3556-3563).  It is a superset of the iterator's chunk list; the device predicate decides the rows, so results are exact.
struct QIv { int32_t tid; int64_t beg, end; };
struct IdxWindow { bool any = false; uint64_t vmin = ~0ull, vmax = 0, last_end = 0; std::vector<std::pair<uint64_t, uint64_t>> chunks; };   // chunks: (begin, end) virtual offsets of every bin chunk the query touches

// the index bytes as stored in memory: a BGZF file (.csi, .tbi) is inflated on the device through a scratch context
static int index_plain(dhts_ctx *c, const uint8_t *&d, uint64_t &n, std::vector<uint8_t> &inflated) {
    if (n >= 18 && d[0] == 0x1f && d[1] == 0x8b) {
        // the same index comes back for every region of a query (and twice per region on text: names, then windows): inflate it once per
        // context.  Identity = address, length and the bytes at both ends.
        uint8_t key[128]; const uint64_t kn = n < 64 ? n : 64;
        memset(key, 0, sizeof(key)); memcpy(key, d, kn); memcpy(key + 64, d + n - kn, kn);
        if (c->idx_cache_src == d && c->idx_cache_n == n && memcmp(c->idx_cache_key, key, 128) == 0 && !c->idx_cache.empty()) { d = c->idx_cache.data(); n = c->idx_cache_len; return 0; }
        const uint8_t *src0 = d; const uint64_t n0 = n;
        dhts_ctx *t = dhts_create(c->device);
        if (!t) return fail(c, "cannot create a scratch context for the index");
        int64_t nb = -1;
        if (dhts_open_host(t, d, n) == 0) nb = dhts_bgzf_index(t);
        if (nb <= 0) { dhts_destroy(t); return fail(c, "index is not readable BGZF"); }
        uint64_t tot = t->h_uoff[nb];
        inflated.resize(tot + 16);
        std::vector<int32_t> bs(nb);
        int64_t got = dhts_bgzf_inflate_to_host(t, 0, nb, inflated.data(), tot, bs.data());
        dhts_destroy(t);
        HIPCHK(c, hipSetDevice(c->device));
        if (got < 0) return fail(c, "index inflate failed");
        for (int64_t k = 0; k < nb; k++) if (bs[k] != 0) return fail(c, "index inflate failed (block %lld)", (long long)k);
        c->idx_cache.swap(inflated); c->idx_cache_len = tot; c->idx_cache_src = src0; c->idx_cache_n = n0; memcpy(c->idx_cache_key, key, 128);
        d = c->idx_cache.data(); n = tot;
    }
    return 0;
}
// the tabix header of an index (TBI: behind n_ref; CSI: the aux block), tbx.c:552-597: preset + the sequence names in index order.
// Returns 0, 1 when the index carries no tabix header, <0 on a malformed one.
static int tabix_header(dhts_ctx *c, const uint8_t *d, uint64_t n, int32_t &preset, std::vector<std::string> &names) {
    const uint8_t *m = nullptr; uint64_t lm = 0;
    if (n >= 8 && memcmp(d, "TBI\1", 4) == 0) { m = d + 8; lm = n - 8; }
    else if (n >= 16 && memcmp(d, "CSI\1", 4) == 0) { lm = hle32(d + 12); if (lm > n - 16) return fail(c, "bad CSI header"); m = d + 16; }
    else return 1;
    if (lm < 28) return 1;
    preset = (int32_t)hle32(m);
    const uint32_t l_nm = hle32(m + 24);
    if (l_nm > lm - 28) return fail(c, "Invalid index header");
    names.clear();
    for (uint32_t p = 0; p < l_nm;) { uint32_t e = p; while (e < l_nm && m[28 + e]) e++; names.emplace_back((const char *)m + 28 + p, e - p); p = e + 1; }
    return 0;
}

static int index_window(dhts_ctx *c, const uint8_t *d, uint64_t n, const std::vector<QIv> &q, bool whole, IdxWindow &w) {
    std::vector<uint8_t> inflated;
    if (index_plain(c, d, n, inflated)) return -1;
    uint64_t p = 0;
    auto need = [&](uint64_t k) { return p + k <= n; };
    auto hle64 = [&](const uint8_t *x) { return (uint64_t)hle32(x) | ((uint64_t)hle32(x + 4) << 32); };
    int min_shift = 14, depth = 5; bool csi = false; bool tbi = false;
    if (need(8) && memcmp(d, "BAI\1", 4) == 0) p = 4;
    else if (need(36) && memcmp(d, "TBI\1", 4) == 0) { p = 4; tbi = true; }         // BAI's body behind n_ref + the tabix header
    else if (need(16) && memcmp(d, "CSI\1", 4) == 0) {
        csi = true; min_shift = (int32_t)hle32(d + 4); depth = (int32_t)hle32(d + 8); const uint32_t l_aux = hle32(d + 12); p = 16;
        if (min_shift < 0 || depth < 0 || depth > 10 || !need(l_aux)) return fail(c, "bad CSI header");
        p += l_aux;
    } else return fail(c, "index is neither BAI nor CSI");
    if (!need(4)) return fail(c, "truncated index");
    const int32_t n_ref = (int32_t)hle32(d + p); p += 4;
    if (tbi) { const uint64_t l_nm = hle32(d + p + 24); if (!need(28 + l_nm)) return fail(c, "truncated index"); p += 28 + l_nm; }
    const uint32_t meta_bin = (uint32_t)(((1ull << (depth * 3 + 3)) - 1) / 7 + 1);
    const int maxs = min_shift + 3 * depth;
    for (int32_t t = 0; t < n_ref; t++) {
        if (!need(4)) return fail(c, "truncated index");
        const int32_t n_bin = (int32_t)hle32(d + p); p += 4;
        std::vector<std::pair<uint32_t, uint32_t>> binr; int64_t first_beg = -1;
        for (auto &iv : q) if (iv.tid == t) {
            int64_t b = iv.beg < 0 ? 0 : iv.beg, e = iv.end; if (e > (1ll << maxs)) e = 1ll << maxs; if (b >= e) continue; --e;
            if (first_beg < 0 || b < first_beg) first_beg = b;
            uint32_t tt = 0;
            for (int l = 0, sft = maxs; l <= depth; l++, sft -= 3) { binr.push_back({(uint32_t)(tt + (b >> sft)), (uint32_t)(tt + (e >> sft))}); tt += 1u << (l * 3); }
        }
        struct Ch { uint64_t u, v; };
        std::vector<Ch> chunks;
        for (int32_t bi = 0; bi < n_bin; bi++) {
            if (!need(csi ? 16 : 8)) return fail(c, "truncated index");
            const uint32_t bin = hle32(d + p); p += 4;
            if (csi) p += 8;                                             // loffset: not needed for a superset window
            const int32_t n_chunk = (int32_t)hle32(d + p); p += 4;
            if (n_chunk < 0 || !need((uint64_t)n_chunk * 16)) return fail(c, "truncated index");
            for (int32_t k = 0; k < n_chunk; k++) {
                const uint64_t u = hle64(d + p), v = hle64(d + p + 8); p += 16;
                if (bin == meta_bin) { if (k == 0 && v > w.last_end) w.last_end = v; continue; }   // pseudo-bin: (ref_beg, ref_end), (n_mapped, n_unmapped)
                if (v > w.last_end) w.last_end = v;
                bool hit = whole;
                for (auto &r : binr) if (bin >= r.first && bin <= r.second) { hit = true; break; }
                if (hit) chunks.push_back({u, v});
            }
        }
        uint64_t min_off = 0;
        if (!csi) {
            if (!need(4)) return fail(c, "truncated index");
            const int32_t n_intv = (int32_t)hle32(d + p); p += 4;
            if (n_intv < 0 || !need((uint64_t)n_intv * 8)) return fail(c, "truncated index");
            if (first_beg >= 0 && n_intv > 0) { int64_t wdw = first_beg >> 14; if (wdw >= n_intv) wdw = n_intv - 1; min_off = hle64(d + p + (uint64_t)wdw * 8); }
            p += (uint64_t)n_intv * 8;
        }
        for (auto &ch : chunks) {
            if (ch.v <= min_off) continue;
            w.any = true;
            if (ch.u < w.vmin) w.vmin = ch.u;
            if (ch.v > w.vmax) w.vmax = ch.v;
            w.chunks.push_back({ch.u > min_off ? ch.u : min_off, ch.v});      // (the linear index: nothing below min_off can overlap, hts.c:3540-3560)
        }
    }
    return 0;
}

// Disjoint windows of a query: the chunks sorted by their start, neighbours merged while the gap between them is cheaper to scan than a
// window is to start (a window costs a batch: ~1.5 ms of launches and hand-shakes, i.e. tens of MB at scan speed).  Rows are decided by
// the overlap predicate, so merging never changes the result; the windows only bound what is staged and inflated.  !multi: one
// covering window [vmin, vmax].
static std::vector<std::pair<uint64_t, uint64_t>> merged_windows(const IdxWindow &w, bool multi) {
    std::vector<std::pair<uint64_t, uint64_t>> mg;
    if (!w.any) return mg;
    if (!multi || w.chunks.size() <= 1) { mg.push_back({w.vmin, w.vmax}); return mg; }
    const uint64_t gap_bytes = (uint64_t)(getenv("DHTS_WINDOW_GAP_MB") ? atof(getenv("DHTS_WINDOW_GAP_MB")) : 32.0) * (1u << 20);
    std::vector<std::pair<uint64_t, uint64_t>> ch = w.chunks;
    std::sort(ch.begin(), ch.end());
    for (auto &x : ch) {
        if (!mg.empty() && (x.first >> 16) <= (mg.back().second >> 16) + gap_bytes) { if (x.second > mg.back().second) mg.back().second = x.second; }
        else mg.push_back(x);
    }
    if (mg.size() == 1) { mg[0] = {w.vmin, w.vmax}; }
    return mg;
}

// The byte ranges of the file a region query needs, from a context that holds only the header (bam_open done) with the regions set:
// beg[k] = file offset of the window's first BGZF block, end[k] = file offset of its LAST block (the opener adds that block's length),
// ~0 = up to the end of the file (the "*" region).  *count = -1: the query needs the whole file.  Same windows as dhts_bam_load_index
// builds afterwards on the context that holds the staged ranges.
extern "C" int dhts_bam_region_segments(dhts_ctx *c, const void *index_bytes, uint64_t n, uint64_t *beg, uint64_t *end, int64_t cap, int64_t *count) {
    if (!c || !c->bam_open || !count) return -1;
    std::vector<QIv> q;
    if (c->rg_active) for (size_t t = 0; t + 1 < c->rg_tid_first.size(); t++) for (uint32_t k = c->rg_tid_first[t]; k < c->rg_tid_first[t + 1]; k++) q.push_back({(int32_t)t, c->rg_beg[k], c->rg_end[k]});
    const bool whole = !c->rg_active || c->rg_all;
    if (whole) { *count = -1; return 0; }
    IdxWindow w;
    if (index_window(c, (const uint8_t *)index_bytes, n, q, whole, w)) return -1;
    std::vector<std::pair<uint64_t, uint64_t>> sg;
    if (c->rg_nocoor) { const uint64_t s0 = w.any ? w.vmin : w.last_end; sg.push_back({s0 >> 16, ~0ull}); }
    else for (auto &x : merged_windows(w, true)) sg.push_back({x.first >> 16, x.second >> 16});
    *count = (int64_t)sg.size();
    if ((int64_t)sg.size() > cap) return fail(c, "room for %lld segments, the query has %lld", (long long)cap, (long long)sg.size());
    for (size_t k = 0; k < sg.size(); k++) { beg[k] = sg[k].first; end[k] = sg[k].second; }
    return 0;
}

// turns a window into the context's scan range; nocoor = also everything after the last mapped chunk ("*")
static int apply_window(dhts_ctx *c, const IdxWindow &w, bool whole, bool nocoor, bool multi = false) {
    int64_t b0 = 0, b1 = c->n_blocks; uint64_t first_uoff = c->first_rec_uoff;
    auto block_of = [&](uint64_t coffset) -> int64_t {          // first resident block at or behind FILE offset coffset
        int64_t lo = 0, hi = c->n_blocks;
        while (lo < hi) { int64_t mid = (lo + hi) / 2; if (block_file_off(c, mid) < coffset) lo = mid + 1; else hi = mid; }
        return lo;
    };
    const bool sparse = !c->segs.empty();                       // only the windows are resident: every window is cut exactly at its end
    c->rg_empty_window = false;
    c->wins.clear(); c->win_cur = 0; c->scan_end_uoff = ~0ull;
    if (!whole && !nocoor) {
        if (!w.any) { c->rg_empty_window = true; return 0; }
        const bool windows = (multi && w.chunks.size() > 1) || sparse;
        if (!windows || merged_windows(w, multi).size() <= 1) {      // (the covering window starts at the smallest chunk start, which the linear index may have pruned from a list of windows)
            b0 = block_of(w.vmin >> 16); b1 = block_of(w.vmax >> 16) + 1; if (b1 > c->n_blocks) b1 = c->n_blocks;
            if (b0 >= c->n_blocks || block_file_off(c, b0) != (w.vmin >> 16)) return fail(c, "index does not match the file (chunk offset %llu)", (unsigned long long)(w.vmin >> 16));
            first_uoff = c->h_uoff[b0] + (w.vmin & 0xffff);
            if (first_uoff < c->first_rec_uoff) first_uoff = c->first_rec_uoff;
        }
        if (windows) {
            // Disjoint windows: the chunks sorted by their start, neighbours merged while the gap between them is cheaper to scan than a
            // window is to start (a window costs a batch: ~1.5 ms of launches and hand-shakes, i.e. tens of MB at scan speed).  Rows are
            // decided by the overlap predicate, so merging never changes the result; the windows only bound what is inflated.
            std::vector<std::pair<uint64_t, uint64_t>> mg = merged_windows(w, multi);
            if (mg.size() > 1 || sparse) {
                for (auto &x : mg) {
                    dhts_ctx::ScanWin sw;
                    sw.b0 = block_of(x.first >> 16);
                    if (sw.b0 >= c->n_blocks || block_file_off(c, sw.b0) != (x.first >> 16)) return fail(c, "index does not match the file (chunk offset %llu)", (unsigned long long)(x.first >> 16));
                    int64_t be = block_of(x.second >> 16); if (be >= c->n_blocks) be = c->n_blocks - 1;
                    sw.b1 = be + 1;
                    sw.first_uoff = c->h_uoff[sw.b0] + (x.first & 0xffff); if (sw.first_uoff < c->first_rec_uoff) sw.first_uoff = c->first_rec_uoff;
                    sw.end_uoff = (block_file_off(c, be) == (x.second >> 16)) ? c->h_uoff[be] + (x.second & 0xffff) : c->h_uoff[be + 1];
                    c->wins.push_back(sw);
                }
                b0 = c->wins[0].b0; b1 = c->wins[0].b1; first_uoff = c->wins[0].first_uoff; c->scan_end_uoff = c->wins[0].end_uoff;
            }
        }
    } else if (!whole && nocoor) {
        const uint64_t s0 = w.any ? w.vmin : w.last_end;
        b0 = block_of(s0 >> 16); if (b0 >= c->n_blocks) b0 = c->n_blocks > 0 ? c->n_blocks - 1 : 0;
        if (c->n_blocks > 0 && block_file_off(c, b0) == (s0 >> 16)) first_uoff = c->h_uoff[b0] + (s0 & 0xffff);
        else if (sparse) return fail(c, "index does not match the file (chunk offset %llu)", (unsigned long long)(s0 >> 16));
        else { b0 = 0; first_uoff = c->first_rec_uoff; }
        if (first_uoff < c->first_rec_uoff) first_uoff = c->first_rec_uoff;
    }
    c->shard_b0 = b0; c->shard_b1 = b1; c->shard_rank = 0; c->shard_world = (b1 < c->n_blocks) ? 2 : 1; c->scan_first_uoff = first_uoff;
    return 0;
}

// how much of the file the current scan range covers: windows (1 without an index) and BGZF blocks
int dhts_scan_window_stats(const dhts_ctx *c, int64_t *n_windows, int64_t *n_blocks) {
    if (!c) return -1;
    int64_t w = 1, b = c->shard_b1 - c->shard_b0;
    if (!c->wins.empty()) { w = (int64_t)c->wins.size(); b = 0; for (auto &x : c->wins) b += x.b1 - x.b0; }
    if (c->rg_empty_window) { w = 0; b = 0; }
    if (n_windows) *n_windows = w;
    if (n_blocks) *n_blocks = b;
    return 0;
}
int dhts_bam_load_index(dhts_ctx *c, const void *bytes, uint64_t n) {
    if (!c || !c->bam_open) return -1;
    std::vector<QIv> q;
    if (c->rg_active) for (size_t t = 0; t + 1 < c->rg_tid_first.size(); t++) for (uint32_t k = c->rg_tid_first[t]; k < c->rg_tid_first[t + 1]; k++) q.push_back({(int32_t)t, c->rg_beg[k], c->rg_end[k]});
    const bool whole = !c->rg_active || c->rg_all;
    IdxWindow w;
    if (index_window(c, (const uint8_t *)bytes, n, q, whole, w)) return -1;
    if (apply_window(c, w, whole, c->rg_nocoor, true)) return -1;
    return dhts_bam_rewind(c);
}

static int bcf_upload_dicts(dhts_ctx *c);
// VCF text (tbx_index_load3 + vcf_hdr_read's "add the missing contigs", vcf.c:2649-2668; tbx_itr_querys): the index names its sequences
// itself.  Returns 1 when the pending region names none of them (no iterator: the reference skips the region).
static int bcf_text_index(dhts_ctx *c, const uint8_t *d, uint64_t n) {
    std::vector<uint8_t> inflated;
    if (index_plain(c, d, n, inflated)) return -1;
    int32_t preset = 0; std::vector<std::string> names;
    const int rc = tabix_header(c, d, n, preset, names);
    if (rc < 0) return -1;
    if (rc == 1) return fail(c, "read_bcf: the index of a VCF text file has no tabix header");
    if ((preset & 0xffff) != 2) return fail(c, "read_bcf: the tabix index was not built with the VCF preset");
    bool added = false;
    for (auto &nm : names) {
        bool have = false;
        for (size_t i = 0; i < c->bh.ctg.size() && !have; i++) have = c->bh.ctg_present[i] && c->bh.ctg[i] == nm;
        if (have) continue;
        if (nm.find('\n') != std::string::npos || !dhts::bcf_header_add_line(c->bh, ("##contig=<ID=" + nm + ">").c_str())) return fail(c, "read_bcf: cannot add contig '%s' of the index to the header", nm.c_str());
        added = true;
    }
    if (added && bcf_upload_dicts(c)) return -1;
    c->tbx_names = names;
    if (c->bcf_rg_pending) {
        int tid; int64_t b, e;
        if (!parse_region_token(names, c->bcf_rg_tok, tid, b, e)) { c->bcf_rg_pending = false; c->bcf_rg_active = false; c->rg_empty_window = true; (void)dhts_bcf_rewind(c); return 1; }
        int32_t rid = -1;
        for (size_t i = 0; i < c->bh.ctg.size(); i++) if (c->bh.ctg_present[i] && c->bh.ctg[i] == names[tid]) { rid = (int32_t)i; break; }
        if (rid < 0) return fail(c, "internal: index sequence without a header id");
        c->bcf_rg_pending = false; c->bcf_rg_itid = tid; c->bcf_rg_tid = rid; c->bcf_rg_beg = b; c->bcf_rg_end = e;
    }
    return 0;
}

int dhts_bcf_load_index(dhts_ctx *c, const void *bytes, uint64_t n) {
    if (!c || !c->bcf_open) return -1;
    if (c->vcf_text) {
        if (c->plain_text) return fail(c, "read_bcf: an uncompressed VCF has no index");
        const int rc = bcf_text_index(c, (const uint8_t *)bytes, n);
        if (rc) return rc;
    }
    std::vector<QIv> q;
    const bool whole = !c->bcf_rg_active || c->bcf_rg_all;
    if (!whole) q.push_back({c->vcf_text ? c->bcf_rg_itid : c->bcf_rg_tid, c->bcf_rg_beg, c->bcf_rg_end});
    IdxWindow w;
    if (index_window(c, (const uint8_t *)bytes, n, q, whole, w)) return -1;
    if (apply_window(c, w, whole, false, true)) return -1;         // disjoint windows, each cut exactly at its end (as for read_bam)
    return dhts_bcf_rewind(c);
}

// compressed bytes of the blocks that hold the header of the open BCF / VCF: what a region query stages in front of its index windows
extern "C" uint64_t dhts_bcf_header_bytes(const dhts_ctx *c) {
    if (!c || !c->bcf_open || c->n_blocks <= 0 || c->plain_text) return 0;
    int64_t k = 0;
    while (k + 1 < c->n_blocks && c->h_uoff[k + 1] < c->first_rec_uoff) k++;
    return c->h_coff[k] + c->h_clen[k];
}
// The byte ranges of the file the regions of read_bcf(region := 'a,b,...') need, from a context that holds the header (dhts_bcf_open done;
// the bind context of the table function): the union of every region's index windows, as dhts_bam_region_segments returns them.
// *count = -1: the query needs the whole file (a "." region, or no usable index).  Regions the index does not know contribute nothing.
extern "C" int dhts_bcf_region_segments(dhts_ctx *c, const char *regions, const void *index_bytes, uint64_t n, uint64_t *beg, uint64_t *end, int64_t cap, int64_t *count) {
    if (!c || !c->bcf_open || !count || !regions) return -1;
    *count = -1;
    if (c->plain_text) return 0;
    IdxWindow all; bool whole = false;
    std::string csv(regions); size_t p = 0;
    while (p <= csv.size() && !whole) {
        size_t q = csv.find(',', p); if (q == std::string::npos) q = csv.size();
        const std::string tok = csv.substr(p, q - p); p = q + 1;
        if (tok.empty()) continue;
        if (dhts_bcf_set_region(c, tok.c_str()) != 0) continue;                 // unknown contig: skipped by the scan as well
        if (c->bcf_rg_all) { whole = true; break; }
        if (c->vcf_text) { const int rc = bcf_text_index(c, (const uint8_t *)index_bytes, n); if (rc == 1) continue; if (rc < 0) return -1; }
        std::vector<QIv> qv; qv.push_back({c->vcf_text ? c->bcf_rg_itid : c->bcf_rg_tid, c->bcf_rg_beg, c->bcf_rg_end});
        IdxWindow w;
        if (index_window(c, (const uint8_t *)index_bytes, n, qv, false, w)) return -1;
        if (!w.any) continue;
        all.any = true;
        for (auto &x : merged_windows(w, true)) all.chunks.push_back(x);         // exactly the windows dhts_bcf_load_index will ask for
    }
    (void)dhts_bcf_set_region(c, nullptr);
    if (whole) return 0;
    std::vector<std::pair<uint64_t, uint64_t>> sg;
    if (all.any) {
        // (the union of the regions' windows: they may overlap; sorted and merged like one region's chunks)
        std::sort(all.chunks.begin(), all.chunks.end());
        std::vector<std::pair<uint64_t, uint64_t>> mg;
        const uint64_t gap_bytes = (uint64_t)(getenv("DHTS_WINDOW_GAP_MB") ? atof(getenv("DHTS_WINDOW_GAP_MB")) : 32.0) * (1u << 20);
        for (auto &x : all.chunks) {
            if (!mg.empty() && (x.first >> 16) <= (mg.back().second >> 16) + gap_bytes) { if (x.second > mg.back().second) mg.back().second = x.second; }
            else mg.push_back(x);
        }
        for (auto &x : mg) sg.push_back({x.first >> 16, x.second >> 16});
    }
    *count = (int64_t)sg.size();
    if ((int64_t)sg.size() > cap) { *count = -1; return 0; }                     // too many ranges for the caller's room: the whole file
    for (size_t k = 0; k < sg.size(); k++) { beg[k] = sg[k].first; end[k] = sg[k].second; }
    return 0;
}

// the scan range becomes window k of a multi-window region query
static void enter_window(dhts_ctx *c, size_t k) {
    const dhts_ctx::ScanWin &w = c->wins[k];
    c->win_cur = k; c->shard_b0 = w.b0; c->shard_b1 = w.b1; c->shard_rank = 0; c->shard_world = (w.b1 < c->n_blocks) ? 2 : 1;
    c->scan_first_uoff = w.first_uoff; c->scan_end_uoff = w.end_uoff;
}
int dhts_bam_rewind(dhts_ctx *c) {
    if (!c) return -1;
    discard_prefetch(c);
    if (!c->wins.empty()) enter_window(c, 0);
    c->next_block = c->shard_b0; c->carry_len = 0; c->stream_done = c->rg_empty_window; c->first_batch = true; c->ucur = 0;
    c->huff_b0 = c->huff_nb = 0;            // a new pass redoes phase A (nothing is cached across scans)
    skip_header_blocks(c);
    return 0;
}

// ---- standard_tags (row A5): the reference's tag table src/bam_reader.c:54-70, in its order ---------------------------------
struct StdTag { const char *tag; char type, subtype; };
static const StdTag kStdTags[] = {
    {"AM",'i',0},{"AS",'i',0},{"BC",'Z',0},{"BQ",'Z',0},{"BZ",'Z',0},{"CB",'Z',0},{"CC",'Z',0},{"CG",'B','I'},{"CM",'i',0},{"CO",'Z',0},{"CP",'i',0},{"CQ",'Z',0},
    {"CR",'Z',0},{"CS",'Z',0},{"CT",'Z',0},{"CY",'Z',0},{"E2",'Z',0},{"FI",'i',0},{"FS",'Z',0},{"FZ",'B','S'},{"H0",'i',0},{"H1",'i',0},{"H2",'i',0},{"HI",'i',0},
    {"IH",'i',0},{"LB",'Z',0},{"MC",'Z',0},{"MD",'Z',0},{"MI",'Z',0},{"ML",'B','C'},{"MM",'Z',0},{"MN",'i',0},{"MQ",'i',0},{"NH",'i',0},{"NM",'i',0},{"OA",'Z',0},
    {"OC",'Z',0},{"OP",'i',0},{"OQ",'Z',0},{"OX",'Z',0},{"PG",'Z',0},{"PQ",'i',0},{"PT",'Z',0},{"PU",'Z',0},{"Q2",'Z',0},{"QT",'Z',0},{"QX",'Z',0},{"R2",'Z',0},
    {"RG",'Z',0},{"RX",'Z',0},{"SA",'Z',0},{"SM",'i',0},{"TC",'i',0},{"TS",'A',0},{"U2",'Z',0},{"UQ",'i',0}};
static const int kNStdTags = (int)(sizeof(kStdTags) / sizeof(kStdTags[0]));

int dhts_bam_std_tag_count(void) { return kNStdTags; }
int dhts_bam_std_tag_info(int idx, char name[3], char *type, char *subtype) {
    if (idx < 0 || idx >= kNStdTags) return -1;
    name[0] = kStdTags[idx].tag[0]; name[1] = kStdTags[idx].tag[1]; name[2] = 0;
    if (type) *type = kStdTags[idx].type;
    if (subtype) *subtype = kStdTags[idx].subtype;
    return 0;
}
int dhts_bam_set_tag_columns(dhts_ctx *c, const int32_t *ids, int32_t n) {
    if (!c) return -1;
    std::vector<int32_t> v;
    for (int32_t i = 0; i < n; i++) { if (ids[i] < 0 || ids[i] >= kNStdTags) return fail(c, "standard tag id %d out of range", ids[i]); v.push_back(ids[i]); }
    c->tag_sel = v;
    return 0;
}

int dhts_bam_set_aux_map(dhts_ctx *c, int enable, int exclude_standard_tags) {
    if (!c) return -1;
    c->aux_on = enable != 0; c->aux_excl_std = exclude_standard_tags != 0;
    return 0;
}

// AUXILIARY_TAGS: typed entries of the non-excluded tags for the (final, compacted) rows of the current batch
static int bam_aux_map(dhts_ctx *c, const BamStream &st, int64_t nrows, dhts_bam_batch *out) {
    out->aux_map = nullptr;
    if (!c->aux_on) return 0;
    memset(&c->aux_out, 0, sizeof(c->aux_out));
    out->aux_map = &c->aux_out;
    if (nrows <= 0) return 0;
    const size_t n = (size_t)nrows;
    std::vector<uint16_t> excl;
    if (c->aux_excl_std) for (int i = 0; i < kNStdTags; i++) excl.push_back((uint16_t)((uint8_t)kStdTags[i].tag[0] | ((uint8_t)kStdTags[i].tag[1] << 8)));
    ENSURE(c, c->x_excl, excl.size() * 2 + 16); ENSURE(c, c->x_valid, n + 64); ENSURE(c, c->x_le, n * 4 + 16); ENSURE(c, c->x_lp, n * 4 + 16);
    ENSURE(c, c->x_oe, (n + 1) * 4 + 16); ENSURE(c, c->x_op, (n + 1) * 4 + 16);
    if (!excl.empty()) HIPCHK(c, hipMemcpyAsync(c->x_excl.p, excl.data(), excl.size() * 2, hipMemcpyHostToDevice, c->stream));
    AuxMapDev a; memset(&a, 0, sizeof(a));
    a.excl = (const uint16_t *)c->x_excl.p; a.n_excl = (int32_t)excl.size(); a.valid = (uint8_t *)c->x_valid.p;
    a.lens_ent = (uint32_t *)c->x_le.p; a.lens_pay = (uint32_t *)c->x_lp.p; a.off_ent = (const uint32_t *)c->x_oe.p; a.off_pay = (const uint32_t *)c->x_op.p;
    { KTimer tm(c, DHTS_K_CORE); hipLaunchKernelGGL(bam_aux_list<false>, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, c->stream, st, (const uint32_t *)c->rec_off.p, nrows, a); }
    const uint32_t *in[2] = {a.lens_ent, a.lens_pay}; uint32_t *o32[2] = {(uint32_t *)c->x_oe.p, (uint32_t *)c->x_op.p}; uint64_t tot[2] = {0, 0};
    { KTimer tm(c, DHTS_K_SCAN); if (run_scan(c, 2, in, o32, nullptr, nrows, tot)) return -1; }
    if (tot[0] >= (1ull << 32) || tot[1] >= (1ull << 32)) return fail(c, "AUXILIARY_TAGS too large for one batch");
    ENSURE(c, c->x_key, tot[0] * 2 + 16); ENSURE(c, c->x_kind, tot[0] + 16); ENSURE(c, c->x_sub, tot[0] + 16); ENSURE(c, c->x_payoff, (tot[0] + 1) * 4 + 16); ENSURE(c, c->x_payload, tot[1] + 64);
    a.key = (uint16_t *)c->x_key.p; a.kind = (uint8_t *)c->x_kind.p; a.sub = (uint8_t *)c->x_sub.p; a.pay_off = (uint32_t *)c->x_payoff.p; a.payload = (uint8_t *)c->x_payload.p;
    { KTimer tm(c, DHTS_K_STRINGS); hipLaunchKernelGGL(bam_aux_list<true>, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, c->stream, st, (const uint32_t *)c->rec_off.p, nrows, a); }
    HIPCHK(c, hipGetLastError());
    c->aux_out.valid = a.valid; c->aux_out.off = a.off_ent; c->aux_out.n_ent = tot[0]; c->aux_out.key = a.key; c->aux_out.kind = a.kind; c->aux_out.sub = a.sub;
    c->aux_out.pay_off = a.pay_off; c->aux_out.payload = a.payload; c->aux_out.payload_bytes = tot[1];
    return 0;
}

// materialises the selected tag columns for the (final, compacted) rows of the current batch
// ---- BAI writer ----------------------------------------------------------------------------------------
// Restates samtools-style index building: sam_index (htslib sam.c:989-1027) = hts_idx_init + hts_idx_push per record +
// hts_idx_finish (hts.c:2400-2690: insert_to_b / insert_to_l 2315-2363, update_loff 2426-2455, compress_binning 2457-2508) and
// idx_save_core (hts.c:2754-2818).  The scan supplies (tid, pos, bam_endpos, mapped) per record; virtual offsets follow
// bgzf_tell's rule (bgzf.c bgzf_read: a read that ends exactly at a block end reports the NEXT block's address with offset 0).
// Bins are written in ascending order (the reference writes them in khash order; readers do not depend on it).
// Builds a BAI for the open BAM with one full scan (the scan state is rewound before and after).  Returns the index size.
static int64_t bam_build_index_impl(dhts_ctx *c, int min_shift);
int64_t dhts_bam_build_index(dhts_ctx *c) { return bam_build_index_impl(c, 0); }
// min_shift > 0: CSI with that min_shift, the depth from the longest reference (sam_index, htslib sam.c:989-1007: hts_adjust_csi_settings
// from n_lvls = 0); min_shift <= 0: BAI
extern "C" int64_t dhts_bam_build_index_csi(dhts_ctx *c, int min_shift) { return bam_build_index_impl(c, min_shift); }
static int64_t bam_build_index_impl(dhts_ctx *c, int min_shift) {
    if (!c) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->bam_open) return fail(c, "dhts_bam_open not called");
    if (c->rg_active || c->shard_b0 != 0 || c->shard_b1 != c->n_blocks || c->ov_active) return fail(c, "index build needs a whole-file scan (no region, shard or join)");
    if (dhts_bam_rewind(c)) return -1;
    const int64_t nb = c->n_blocks;
    auto tell = [&](uint64_t u) -> uint64_t {                                        // bgzf_tell after having read up to inflated offset u
        const uint64_t *uo = c->h_uoff.data();
        int64_t lo = 0, hi = nb + 1;
        while (lo < hi) { const int64_t mid = (lo + hi) / 2; if (uo[mid] < u) lo = mid + 1; else hi = mid; }
        if (lo <= nb && uo[lo] == u) return (lo < nb ? c->h_coff[lo] : c->comp_len) << 16;
        return (c->h_coff[lo - 1] << 16) | (u - uo[lo - 1]);
    };
    IndexAcc ib;
    if (min_shift > 0) {
        int64_t max_len = 0; for (uint32_t l : c->ref_len) if ((int64_t)l > max_len) max_len = l;
        int n_lvls = 0;                                                              // hts_adjust_csi_settings (hts.c:2367-2400) from n_lvls = 0
        const int64_t need = max_len + 256;
        if (need <= (1ll << (min_shift + 27))) { int64_t maxpos = 1ll << min_shift; while (need > maxpos) { ++n_lvls; maxpos *= 8; } }
        else { n_lvls = 9; int64_t maxpos = 1ll << (min_shift + 27); while (need > maxpos) { ++min_shift; maxpos *= 2; } }
        ib.set_csi(min_shift, n_lvls);
    }
    const int n_ref = (int)c->ref_name.size();
    ib.begin(n_ref, tell(c->first_rec_uoff));
    // device state of the passes (hts_index.hip): the windows of every sequence (its length plus 1 Mb of room for reads that hang over its end),
    // counts, violation flags, the last row of the previous batch
    std::vector<uint64_t> lin_base((size_t)n_ref + 1, 0);
    for (int t = 0; t < n_ref; t++) lin_base[(size_t)t + 1] = lin_base[(size_t)t] + ((((uint64_t)c->ref_len[(size_t)t] + (1u << 20)) >> ib.g.min_shift) + 2);
    const uint64_t n_win = lin_base[(size_t)n_ref];
    DevBuf d_lin, d_base, d_cnt, d_misc;
    ENSURE(c, d_lin, n_win * 8 + 64); ENSURE(c, d_base, ((size_t)n_ref + 1) * 8 + 64); ENSURE(c, d_cnt, (size_t)n_ref * 24 + 64); ENSURE(c, d_misc, 256);
    HIPCHK(c, hipMemsetAsync(d_lin.p, 0xff, n_win * 8 + 64, c->stream));
    HIPCHK(c, hipMemsetAsync(d_cnt.p, 0, (size_t)n_ref * 24 + 64, c->stream));
    HIPCHK(c, hipMemsetAsync(d_misc.p, 0, 256, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_base.p, lin_base.data(), ((size_t)n_ref + 1) * 8, hipMemcpyHostToDevice, c->stream));
    {   // the carry of "batch -1": no row yet, the reader stands in front of the first record
        IdxCarry k{}; k.v = ib.v0;
        HIPCHK(c, hipMemcpyAsync((uint8_t *)d_misc.p + 64, &k, sizeof(k), hipMemcpyHostToDevice, c->stream));
    }
    IdxDev dv; dv.g = ib.g; dv.n_ref = n_ref; dv.lin_base = (const uint64_t *)d_base.p; dv.lin = (unsigned long long *)d_lin.p;
    dv.nmap = (unsigned long long *)d_cnt.p; dv.nunmap = dv.nmap + n_ref; dv.tid_runs = (uint32_t *)(dv.nunmap + n_ref); dv.max_win = dv.tid_runs + n_ref;
    dv.n_nocoor = (unsigned long long *)d_misc.p; dv.err = (uint32_t *)((uint8_t *)d_misc.p + 8);
    IdxCarry *carry2 = (IdxCarry *)((uint8_t *)d_misc.p + 64);
    dhts_bam_batch b; int slot = 0; int rc = 0;
    std::vector<IdxRun> hruns;
    const bool timing = getenv("DHTS_IDX_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_scan = 0, t_pass = 0; const double t_begin = now();
    for (;;) {
        const uint32_t mask = (1u << DHTS_BAM_FLAG) | (1u << DHTS_BAM_RNAME) | (1u << DHTS_BAM_POS);
        const double t0 = now();
        if (dhts_bam_next_batch(c, 0, mask, &b)) return -1;
        const double t1 = now(); t_scan += t1 - t0;
        const int64_t n = b.n_rows;
        if (n > 0) {
            ENSURE(c, c->ix_end, (size_t)n * sizeof(IdxRun) * 2 + 64); ENSURE(c, c->c_keep, (size_t)n * 4 + 16); ENSURE(c, c->c_rowmap, ((size_t)n + 1) * 4 + 16);
            BamCols bc; memset(&bc, 0, sizeof(bc));
            bc.flag = (uint16_t *)c->c_flag.p; bc.pos = (int64_t *)c->c_pos.p; bc.cig_rel = (uint32_t *)c->cig_rel.p; bc.ncig_eff = (uint32_t *)c->ncig_eff.p;
            uint32_t r0 = 0; HIPCHK(c, hipMemcpyAsync(&r0, c->rec_off.p, 4, hipMemcpyDeviceToHost, c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream));
            const uint64_t out_base = b.first_rec_uoff - r0;
            dv.carry_in = carry2 + slot; dv.carry_out = carry2 + (slot ^ 1); slot ^= 1;
            IdxRun *row_run = (IdxRun *)c->ix_end.p, *runs_dev = row_run + n;
            hipLaunchKernelGGL(bam_index_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->last_stream, (const uint32_t *)c->rec_off.p, bc, (const int32_t *)b.tid, n,
                               out_base, b.end_uoff, (const uint64_t *)c->uoff.p, (const uint64_t *)c->coff.p, nb, c->comp_len, dv, (uint32_t *)c->c_keep.p, row_run);
            const uint32_t *kin[1] = {(const uint32_t *)c->c_keep.p}; uint32_t *kout[1] = {(uint32_t *)c->c_rowmap.p}; uint64_t nruns = 0;
            if (run_scan(c, 1, kin, kout, nullptr, n, &nruns)) return -1;
            hipLaunchKernelGGL(idx_runs_write, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (const uint32_t *)c->c_keep.p, (const uint32_t *)c->c_rowmap.p, (const IdxRun *)row_run, n, runs_dev);
            HIPCHK(c, hipGetLastError());
            const size_t at = hruns.size(); hruns.resize(at + (size_t)nruns);
            if (nruns) HIPCHK(c, hipMemcpyAsync(hruns.data() + at, runs_dev, (size_t)nruns * sizeof(IdxRun), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
        t_pass += now() - t1;
        if (b.status != 0) { if (b.status < 0) { dhts_bam_rewind(c); return fail(c, "index build: the scan ended on an error (status %d)", b.status); } break; }
    }
    const double t_loop = now();
    dhts_bam_rewind(c);
    {   // the passes' results: violation flags, counts, the linear index up to the last window touched
        uint32_t err = 0; std::vector<uint64_t> cnt((size_t)n_ref * 3 + 8, 0);
        HIPCHK(c, hipMemcpy(&err, dv.err, 4, hipMemcpyDeviceToHost));
        if (err) return fail(c, "index build: %s", ib.err_text(err));
        HIPCHK(c, hipMemcpy(&ib.n_nocoor, dv.n_nocoor, 8, hipMemcpyDeviceToHost));
        if (n_ref) HIPCHK(c, hipMemcpy(cnt.data(), d_cnt.p, (size_t)n_ref * 24, hipMemcpyDeviceToHost));
        const uint32_t *tr = (const uint32_t *)(cnt.data() + 2 * (size_t)n_ref), *mw = tr + n_ref;
        for (int t = 0; t < n_ref; t++) {
            ib.nmap[(size_t)t] = cnt[(size_t)t]; ib.nunmap[(size_t)t] = cnt[(size_t)n_ref + (size_t)t]; ib.tid_runs[(size_t)t] = tr[t];
            ib.lin[(size_t)t].resize(mw[t]);
            if (mw[t]) HIPCHK(c, hipMemcpy(ib.lin[(size_t)t].data(), (const uint64_t *)d_lin.p + lin_base[(size_t)t], (size_t)mw[t] * 8, hipMemcpyDeviceToHost));
        }
        ib.runs.swap(hruns);
    }
    (void)rc;
    // where the reader stands after the failing read at EOF: the address of the last trailing empty block, else the file size
    uint64_t fin = c->comp_len;
    if (nb > 0 && c->h_isize[nb - 1] == 0) fin = c->h_coff[nb - 1];
    const double t_dl = now();
    if (!ib.finish(fin << 16)) return fail(c, "index build: %s", ib.err.c_str());
    ib.save(c->built_index);
    if (timing) fprintf(stderr, "index build: setup %.2f ms, scan %.2f ms, passes %.2f ms, rewind+download %.2f ms, finish+save %.2f ms (%zu runs)\n", 0.0, t_scan, t_pass, t_dl - t_loop, now() - t_dl, ib.runs.size());
    (void)t_begin;
    return (int64_t)c->built_index.size();
}
int dhts_bam_index_bytes(dhts_ctx *c, uint8_t *out, uint64_t cap) {
    if (!c || !out) return -1;
    if (cap < c->built_index.size()) return fail(c, "index buffer too small");
    memcpy(out, c->built_index.data(), c->built_index.size());
    return 0;
}

// CSI for the open BCF (bcf_index, htslib vcf.c:4657-4688: hts_idx_push(rid, pos, pos + rlen, bgzf_tell) per record; min_shift 14 by
// default, the number of levels from the longest contig of the header, idx_calc_n_lvls_ids + hts_adjust_csi_settings hts.c:2367-2400).
// One scan of the file: the device delivers every record's contig id, position and rlen (the core of the BCF2 record).  The bytes are the
// UNCOMPRESSED index; dhts_bgzf_wrap makes the .csi file of them.
int64_t dhts_bcf_build_index(dhts_ctx *c, int min_shift) {
    if (!c) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->bcf_open) return fail(c, "dhts_bcf_open not called");
    if (c->vcf_text && c->plain_text) return fail(c, "index build: an uncompressed VCF cannot be indexed (tabix needs BGZF)");
    if (c->bcf_rg_active || c->shard_b0 != 0 || c->shard_b1 != c->n_blocks) return fail(c, "index build needs a whole-file scan (no region or shard)");
    const bool text = c->vcf_text, tbi = text && min_shift <= 0;
    if (min_shift <= 0) min_shift = 14;
    int64_t max_len = 0; int nids = 0;
    for (size_t i = 0; i < c->bh.ctg.size(); i++) if (c->bh.ctg_present[i]) { nids++; if (i < c->bh.ctg_len.size() && c->bh.ctg_len[i] > max_len) max_len = c->bh.ctg_len[i]; }
    int n_lvls = 0;
    auto adjust = [&](int64_t len) {                                                 // hts_adjust_csi_settings (hts.c:2367-2400)
        const int64_t need = len + 256;
        if (need <= (1ll << (min_shift + 27))) { int64_t maxpos = 1ll << (min_shift + 3 * n_lvls); while (need > maxpos) { ++n_lvls; maxpos *= 8; } }
        else { n_lvls = 9; int64_t maxpos = 1ll << (min_shift + 27); while (need > maxpos) { ++min_shift; maxpos *= 2; } }
    };
    if (!text) { if (!max_len) max_len = (1ll << 31) - 1; adjust(max_len); }        // idx_calc_n_lvls_ids
    else if (tbi) n_lvls = 5;
    else {                                                                           // tbx_index (tbx.c:451-484): TBX_MAX_SHIFT 31; the ##contig lengths, or a generous default
        n_lvls = (31 - min_shift + 2) / 3;
        if (max_len) adjust(max_len);
        else n_lvls = min_shift < 10 ? 9 : min_shift < 25 ? 9 - (min_shift - 10) / 3 : 4;
    }
    const std::vector<int32_t> saved_proj = c->bcf_proj;
    const int32_t none = 0;
    if (dhts_bcf_set_projection(c, &none, 0) || dhts_bcf_rewind(c)) return -1;
    const int64_t nb = c->n_blocks;
    auto tell = [&](uint64_t u) -> uint64_t {
        const uint64_t *uo = c->h_uoff.data();
        int64_t lo = 0, hi = nb + 1;
        while (lo < hi) { const int64_t mid = (lo + hi) / 2; if (uo[mid] < u) lo = mid + 1; else hi = mid; }
        if (lo <= nb && uo[lo] == u) return (lo < nb ? c->h_coff[lo] : c->comp_len) << 16;
        return (c->h_coff[lo - 1] << 16) | (u - uo[lo - 1]);
    };
    IndexAcc ib;
    if (tbi) ib.tbi = true; else ib.set_csi(min_shift, n_lvls);
    ib.grow = true;                                                                  // (sequences are numbered as they are met / as the header numbers them: the table grows)
    ib.begin(text ? 0 : nids, tell(c->first_rec_uoff));
    std::vector<int32_t> tid_of; std::vector<std::string> tnames;                    // text: sequence ids in the order of first appearance (get_tid, tbx.c:82-107)
    bool ok = true; int rc = 0;
    std::vector<uint32_t> ro; std::vector<uint8_t> core;
    std::vector<int32_t> a_tid; std::vector<int64_t> a_beg, a_end; std::vector<uint64_t> a_v; std::vector<uint8_t> a_map;
    struct Restore {                                                                 // every way out puts the caller's projection back and rewinds the scan
        dhts_ctx *c; const std::vector<int32_t> &proj;
        ~Restore() { const std::string keep = c->err; (void)dhts_bcf_set_projection(c, proj.data(), (int32_t)proj.size()); (void)dhts_bcf_rewind(c); if (!keep.empty()) c->err = keep; }
    } restore{c, saved_proj};
    for (;;) {
        dhts_bcf_batch b;
        if (dhts_bcf_next_batch(c, 0, &b)) { rc = -1; break; }
        const int64_t n = b.n_rows / (c->bsch.tidy && c->bsch.n_samples > 0 ? c->bsch.n_samples : 1);
        if (n > 0) {
            // rid / pos / rlen sit in the 32-byte head of every record: offsets from the batch, the heads gathered by one strided copy
            ro.resize(n); core.resize((size_t)n * 12);
            HIPCHK(c, hipMemcpyAsync(ro.data(), text ? c->v_line_off.p : c->b_rec_off.p, n * 4, hipMemcpyDeviceToHost, c->stream));   // (text: a line ends where the next one starts)
            HIPCHK(c, hipStreamSynchronize(c->stream));
            ENSURE(c, c->ix_end, (size_t)n * 12 + 64);
            hipLaunchKernelGGL(bcf_index_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->last_bcf_u, (const uint32_t *)c->b_rec_off.p, n, (uint32_t *)c->ix_end.p);
            HIPCHK(c, hipMemcpyAsync(core.data(), c->ix_end.p, (size_t)n * 12, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            const uint64_t base = b.first_rec_uoff - ro[0];
            // the batch as arrays: sequence, interval, the virtual offset behind every record (IndexAcc::add_rows makes the passes)
            a_tid.resize(n); a_beg.resize(n); a_end.resize(n); a_v.resize(n); a_map.assign(n, 1);
            for (int64_t i = 0; i < n && ok; i++) {
                int32_t rid, pos, rlen; memcpy(&rid, &core[(size_t)i * 12], 4); memcpy(&pos, &core[(size_t)i * 12 + 4], 4); memcpy(&rlen, &core[(size_t)i * 12 + 8], 4);
                const uint64_t u_end = (i + 1 < n) ? base + ro[i + 1] : b.end_uoff;
                const int64_t p64 = (uint32_t)pos == 0xffffffffu ? -1 : (int64_t)pos;
                a_v[(size_t)i] = tell(u_end);
                if (text) {
                    if (rid < 0 || rid >= (int32_t)c->bh.ctg.size()) { ok = false; ib.err = "record without a sequence name"; break; }
                    if ((int32_t)tid_of.size() < (int32_t)c->bh.ctg.size()) tid_of.resize(c->bh.ctg.size(), -1);
                    if (tid_of[rid] < 0) { tid_of[rid] = (int32_t)tnames.size(); tnames.push_back(c->bh.ctg[rid]); }
                    a_tid[(size_t)i] = tid_of[rid]; a_beg[(size_t)i] = p64 < 0 ? 0 : p64; a_end[(size_t)i] = p64 + rlen;      // the interval of tbx_parse1: rlen of a text record is its tabix END - pos
                } else { a_tid[(size_t)i] = rid; a_beg[(size_t)i] = p64; a_end[(size_t)i] = p64 + rlen; }
            }
            if (ok) ok = ib.add_rows(a_tid.data(), a_beg.data(), a_end.data(), a_v.data(), a_map.data(), n);
            if (!ok) break;
        }
        if (b.status != 0) { if (b.status < 0) { rc = fail(c, "index build: the scan ended on an error (status %d)", b.status); } break; }
    }
    if (rc) return -1;
    if (!ok) return fail(c, "index build: %s", ib.err.c_str());
    uint64_t fin = c->comp_len;
    if (nb > 0 && c->h_isize[nb - 1] == 0) fin = c->h_coff[nb - 1];
    if (!ib.finish(fin << 16)) return fail(c, "index build: %s", ib.err.c_str());
    if (text) {                                                                      // tbx_set_meta: the VCF preset {TBX_VCF, 1, 2, 0, '#', 0}, l_nm, names
        const uint32_t conf[6] = {2, 1, 2, 0, '#', 0}; uint32_t l_nm = 0;
        for (auto &nm : tnames) l_nm += (uint32_t)nm.size() + 1;
        auto w32 = [&](uint32_t x) { for (int k = 0; k < 4; k++) ib.aux.push_back((uint8_t)(x >> (8 * k))); };
        for (uint32_t x : conf) w32(x);
        w32(l_nm);
        for (auto &nm : tnames) { ib.aux.insert(ib.aux.end(), nm.begin(), nm.end()); ib.aux.push_back(0); }
    }
    ib.save(c->built_index);
    return (int64_t)c->built_index.size();
}

// ---- bgzip / bgunzip (src/bgzip.c: bgzf_write / bgzf_read loops, htslib bgzf.c) ----------------------------------------------------------
static const uint8_t BGZF_EOF_BLOCK[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
// raw bytes already in z_in (n of them, padded) -> packed BGZF blocks in z_out; *out_len = their size
static int bgzf_compress_device(dhts_ctx *c, uint64_t n, int level, uint64_t *out_len) {
    const int64_t nblk = (int64_t)((n + DFL_IN - 1) / DFL_IN);
    *out_len = 0;
    if (nblk == 0) return 0;
    ENSURE(c, c->z_slots, (size_t)nblk * DFL_SLOT + 64); ENSURE(c, c->z_sizes, (size_t)(nblk + 1) * 4 + 64); ENSURE(c, c->z_offs, (size_t)(nblk + 2) * 8 + 64);
    if (level != 0) ENSURE(c, c->z_tok, (size_t)nblk * DFL_IN * 4 + 64);                  // the parse of every block: one word per token
    hipLaunchKernelGGL(bgzf_deflate_blocks, dim3((unsigned)nblk), dim3(64), DFL_LDS_BYTES, c->stream, (const uint8_t *)c->z_in.p, n, nblk, level, (uint8_t *)c->z_slots.p, (uint32_t *)c->z_sizes.p,
                       (uint32_t *)c->z_tok.p);
    HIPCHK(c, hipGetLastError());
    const uint32_t *in1[1] = {(const uint32_t *)c->z_sizes.p}; uint64_t *o64[1] = {(uint64_t *)c->z_offs.p}; uint64_t total = 0;
    if (run_scan(c, 1, in1, nullptr, o64, nblk, &total)) return -1;
    ENSURE(c, c->z_out, total + 64);
    hipLaunchKernelGGL(bgzf_pack_blocks, dim3((unsigned)nblk), dim3(64), 0, c->stream, (const uint8_t *)c->z_slots.p, (const uint32_t *)c->z_sizes.p, (const uint64_t *)c->z_offs.p, nblk, (uint8_t *)c->z_out.p);
    HIPCHK(c, hipGetLastError());
    *out_len = total;
    return 0;
}
// GPU bgzip of a host buffer: one BGZF block per 0xff00 input bytes + the EOF block.  Returns the file size; with out == NULL or cap too small
// nothing is written and the return value is an upper bound (call twice).  level 0 stores, 1..9 (and -1) compress (one setting).
extern "C" int64_t dhts_bgzf_compress(dhts_ctx *c, const void *raw, uint64_t n, int level, void *out, uint64_t cap) {
    if (!c) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    const uint64_t nblk = (n + DFL_IN - 1) / DFL_IN, bound = n + nblk * 31 + 28;
    if (!out || cap < bound) return (int64_t)bound;
    if (n >= (1ull << 40)) return fail(c, "input too large");
    uint64_t done = 0, at = 0;
    const uint64_t CH = (uint64_t)DFL_IN * 4096;                                  // 267 MB of input per launch
    while (done < n) {
        const uint64_t m = n - done < CH ? n - done : CH;
        ENSURE(c, c->z_in, m + PAD_BYTES);
        HIPCHK(c, hipMemcpyAsync(c->z_in.p, (const uint8_t *)raw + done, m, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemsetAsync((uint8_t *)c->z_in.p + m, 0, PAD_BYTES, c->stream));
        uint64_t ol = 0;
        if (bgzf_compress_device(c, m, level, &ol)) return -1;
        if (at + ol + 28 > cap) return fail(c, "internal: compressed size above its bound");
        HIPCHK(c, hipMemcpyAsync((uint8_t *)out + at, c->z_out.p, ol, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        at += ol; done += m;
    }
    memcpy((uint8_t *)out + at, BGZF_EOF_BLOCK, 28);
    return (int64_t)(at + 28);
}
// bgzip(path): file -> BGZF file (bgzip.c:231-293).  Returns 0, -2 input cannot be opened, -3 output cannot be opened, -4 read error, -5 write error.
extern "C" int dhts_bgzip_file(dhts_ctx *c, const char *in_path, const char *out_path, int level, int64_t *bytes_in, int64_t *bytes_out) {
    if (!c || !in_path || !out_path) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    const int fd = open(in_path, O_RDONLY);
    if (fd < 0) { fail(c, "bgzip: cannot open input %s: %s", in_path, strerror(errno)); return -2; }
    FILE *fo = fopen(out_path, "wb");
    if (!fo) { close(fd); fail(c, "bgzip: cannot open output %s", out_path); return -3; }
    const uint64_t CH = (uint64_t)DFL_IN * 4096;                                  // 267 MB of input per launch
    uint8_t *pin = (uint8_t *)dhts_host_alloc(CH + CH / 1024 + (1u << 20));
    int rc = 0; int64_t nin = 0, nout = 0;
    if (!pin) { rc = -1; fail(c, "bgzip: out of pinned memory"); }
    if (level < 0 || level > 9) level = -1;
    while (rc == 0) {
        size_t got = 0;
        while (got < CH) { const ssize_t r = read(fd, pin + got, CH - got); if (r < 0) { rc = -4; fail(c, "bgzip: read error"); break; } if (r == 0) break; got += (size_t)r; }
        if (rc || got == 0) break;
        uint64_t ol = 0;
        if (c->z_in.ensure(got + PAD_BYTES)) { rc = -1; fail(c, "hipMalloc failed"); break; }
        if (hipMemcpyAsync(c->z_in.p, pin, got, hipMemcpyHostToDevice, c->stream) != hipSuccess || hipMemsetAsync((uint8_t *)c->z_in.p + got, 0, PAD_BYTES, c->stream) != hipSuccess ||
            bgzf_compress_device(c, got, level == 0 ? 0 : 6, &ol) || hipMemcpyAsync(pin, c->z_out.p, ol, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) {
            rc = -1; if (!*dhts_error(c)) fail(c, "bgzip: device error"); break;
        }
        if (fwrite(pin, 1, ol, fo) != ol) { rc = -5; fail(c, "bgzip: write error"); break; }
        nin += (int64_t)got; nout += (int64_t)ol;
        if (got < CH) break;
    }
    if (rc == 0 && fwrite(BGZF_EOF_BLOCK, 1, 28, fo) != 28) { rc = -5; fail(c, "bgzip: write error"); }
    if (rc == 0) nout += 28;
    if (pin) dhts_host_free(pin);
    close(fd);
    if (fclose(fo) != 0 && rc == 0) { rc = -5; fail(c, "bgzip: close error"); }
    if (rc) unlink(out_path);
    if (bytes_in) *bytes_in = nin;
    if (bytes_out) *bytes_out = nout;
    return rc;
}
// bgunzip(path): BGZF file -> the bytes it holds (bgzip.c:167-230: bgzf_read until 0).  Blocks are inflated on the device in batches.
// Returns 0, -2 input cannot be opened / is not BGZF, -3 output cannot be opened, -4 a block failed (the reference: "bgunzip: read error"), -5 write error.
extern "C" int dhts_bgunzip_file(dhts_ctx *c, const char *in_path, const char *out_path, int64_t *bytes_in, int64_t *bytes_out) {
    if (!c || !in_path || !out_path) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    {
        // a file that is not gzip at all is handed through as it is (bgzf_open "r" reads such a file transparently, bgzf.c:412-450, so the
        // reference's bgunzip copies it): nothing to inflate, the copy is plain file I/O
        const int fd = open(in_path, O_RDONLY);
        if (fd < 0) { fail(c, "bgunzip: cannot open input %s", in_path); return -2; }
        uint8_t magic[2] = {0, 0}; const ssize_t got = pread(fd, magic, 2, 0);
        if (got >= 0 && !(got == 2 && magic[0] == 0x1f && magic[1] == 0x8b)) {   // (a file shorter than the magic is not gzip either)
            FILE *fo = fopen(out_path, "wb");
            if (!fo) { close(fd); fail(c, "bgunzip: cannot open output %s: %s", out_path, strerror(errno)); return -3; }
            std::vector<uint8_t> buf(1u << 20); int rc = 0; int64_t n = 0;
            for (;;) { const ssize_t r = read(fd, buf.data(), buf.size()); if (r < 0) { rc = -4; fail(c, "bgunzip: read error"); break; } if (r == 0) break;
                       if (fwrite(buf.data(), 1, (size_t)r, fo) != (size_t)r) { rc = -5; fail(c, "bgunzip: write error"); break; } n += r; }
            close(fd);
            if (fclose(fo) != 0 && rc == 0) { rc = -5; fail(c, "bgunzip: write error"); }
            if (rc) unlink(out_path);                        // no partial output behind an error
            if (bytes_in) *bytes_in = n;
            if (bytes_out) *bytes_out = n;
            return rc;
        }
        close(fd);
    }
    if (dhts_open_path(c, in_path) != 0) return -2;
    const int64_t nb = dhts_bgzf_index(c);
    if (nb < 0 || (nb == 0 && c->comp_len > 0)) { if (nb == 0) fail(c, "bgunzip: %s is gzip but not BGZF (plain gzip is not read by this build)", in_path); return -2; }
    if (c->bgzf_status != 0) { fail(c, "bgunzip: read error"); return -4; }
    FILE *fo = fopen(out_path, "wb");
    if (!fo) { fail(c, "bgunzip: cannot open output %s: %s", out_path, strerror(errno)); return -3; }
    const int64_t B = 4096;
    uint64_t cap = 0;
    for (int64_t b = 0; b < nb; b += B) { const int64_t e = b + B < nb ? b + B : nb; if (c->h_uoff[e] - c->h_uoff[b] > cap) cap = c->h_uoff[e] - c->h_uoff[b]; }
    uint8_t *pin = (uint8_t *)dhts_host_alloc(cap + 64);
    std::vector<int32_t> bs(B);
    int rc = pin ? 0 : -1; int64_t nout = 0;
    for (int64_t b = 0; b < nb && rc == 0; b += B) {
        const int64_t k = b + B < nb ? B : nb - b;
        const int64_t got = dhts_bgzf_inflate_to_host(c, b, k, pin, cap, bs.data());
        if (got < 0) { rc = -1; break; }
        for (int64_t i = 0; i < k; i++) if (bs[i] != 0) { rc = -4; fail(c, "bgunzip: read error"); break; }
        if (rc) break;
        if (got > 0 && fwrite(pin, 1, (size_t)got, fo) != (size_t)got) { rc = -5; fail(c, "bgunzip: write error"); break; }
        nout += got;
    }
    if (pin) dhts_host_free(pin);
    if (fclose(fo) != 0 && rc == 0) { rc = -5; fail(c, "bgunzip: write error"); }
    if (rc) unlink(out_path);
    if (bytes_in) *bytes_in = (int64_t)c->file_size;
    if (bytes_out) *bytes_out = nout;
    return rc;
}

// raw bytes -> a valid BGZF file (what hts_idx_save writes a .csi / .tbi through): stored (uncompressed) DEFLATE blocks of up to 65,280
// bytes with CRC-32 and ISIZE, and the 28-byte EOF block.  Host only.  Returns the size (also when out is NULL or too small: call twice).
extern "C" int64_t dhts_bgzf_wrap(const void *raw, uint64_t n, void *out, uint64_t cap) {
    static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const uint8_t *src = (const uint8_t *)raw; uint8_t *dst = (uint8_t *)out; uint64_t need = 28, at = 0;
    for (uint64_t p = 0; p < n; p += 65280) need += 18 + 5 + (n - p < 65280 ? n - p : 65280) + 8;
    if (!dst || cap < need) return (int64_t)need;
    uint32_t tab[256];
    for (uint32_t i = 0; i < 256; i++) { uint32_t x = i; for (int k = 0; k < 8; k++) x = (x & 1) ? 0xEDB88320u ^ (x >> 1) : x >> 1; tab[i] = x; }
    for (uint64_t p = 0; p < n; p += 65280) {
        const uint32_t l = (uint32_t)(n - p < 65280 ? n - p : 65280), total = 18 + 5 + l + 8;
        const uint8_t hd[18] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, (uint8_t)((total - 1) & 0xff), (uint8_t)((total - 1) >> 8)};
        memcpy(dst + at, hd, 18); at += 18;
        dst[at++] = 1; dst[at++] = (uint8_t)l; dst[at++] = (uint8_t)(l >> 8); dst[at++] = (uint8_t)~l; dst[at++] = (uint8_t)((~l) >> 8);      // BFINAL = 1, BTYPE = 00, LEN, NLEN
        memcpy(dst + at, src + p, l); at += l;
        uint32_t crc = 0xffffffffu; for (uint32_t i = 0; i < l; i++) crc = tab[(crc ^ src[p + i]) & 0xff] ^ (crc >> 8);
        crc ^= 0xffffffffu;
        for (int k = 0; k < 4; k++) dst[at++] = (uint8_t)(crc >> (8 * k));
        for (int k = 0; k < 4; k++) dst[at++] = (uint8_t)(l >> (8 * k));
    }
    memcpy(dst + at, eof, 28); at += 28;
    return (int64_t)at;
}

// ---- interval overlap join ------------------------------------------------------------------------------------------------
int dhts_bam_set_overlap_intervals(dhts_ctx *c, const int32_t *tid, const int64_t *beg, const int64_t *end, int64_t n) {
    if (!c) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    c->ov_active = false; c->ov_n = 0;
    if (n <= 0) return 0;
    if (!c->bam_open) return fail(c, "dhts_bam_open not called");      // the intervals are resolved against the header's reference table
    if (!tid || !beg || !end) return fail(c, "overlap intervals: null array");
    if (n > 0xfffffff0ll) return fail(c, "overlap intervals: too many intervals");
    const int32_t n_ref = (int32_t)c->ref_name.size();
    std::vector<uint32_t> order; order.reserve((size_t)n);
    for (int64_t i = 0; i < n; i++) if (tid[i] >= 0 && tid[i] < n_ref) order.push_back((uint32_t)i);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return tid[a] != tid[b] ? tid[a] < tid[b] : beg[a] < beg[b]; });
    const size_t m = order.size();
    std::vector<int64_t> sb(m), se(m), pm(m), bm((m + 63) / 64 + 1, INT64_MIN); std::vector<uint32_t> first((size_t)n_ref + 1, 0);
    for (size_t k = 0; k < m; k++) {
        const uint32_t i = order[k];
        sb[k] = beg[i]; se[k] = end[i];
        pm[k] = (k > 0 && tid[order[k - 1]] == tid[i] && pm[k - 1] > end[i]) ? pm[k - 1] : end[i];
        if (end[i] > bm[k >> 6]) bm[k >> 6] = end[i];
        first[(size_t)tid[i] + 1]++;
    }
    for (int32_t t = 0; t < n_ref; t++) first[(size_t)t + 1] += first[t];
    ENSURE(c, c->ov_beg, m * 8 + 64); ENSURE(c, c->ov_end, m * 8 + 64); ENSURE(c, c->ov_pmax, m * 8 + 64); ENSURE(c, c->ov_bmax, bm.size() * 8 + 64); ENSURE(c, c->ov_id, m * 4 + 64); ENSURE(c, c->ov_first, ((size_t)n_ref + 1) * 4 + 64);
    if (m) {
        HIPCHK(c, hipMemcpy(c->ov_beg.p, sb.data(), m * 8, hipMemcpyHostToDevice)); HIPCHK(c, hipMemcpy(c->ov_end.p, se.data(), m * 8, hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(c->ov_pmax.p, pm.data(), m * 8, hipMemcpyHostToDevice)); HIPCHK(c, hipMemcpy(c->ov_id.p, order.data(), m * 4, hipMemcpyHostToDevice));
    }
    HIPCHK(c, hipMemcpy(c->ov_bmax.p, bm.data(), bm.size() * 8, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->ov_first.p, first.data(), ((size_t)n_ref + 1) * 4, hipMemcpyHostToDevice));
    c->ov_active = true; c->ov_n = (int64_t)m;
    return 0;
}

// read_bed rows as the join's intervals (src/interval_udf.c:330-426): the device splits the text into lines and parses chrom / start / end of
// every row (bed_intervals); the host only maps the chrom names, where they change, to the header's reference ids.  Interval id = row number of
// read_bed (meta lines do not count).  A row whose start or end is NULL, or whose chrom the BAM header does not have, never matches.  text = the
// (uncompressed) BED bytes.  Returns the number of rows, < 0 on error (a line with fewer than 3 fields: read_bed's error).
extern "C" int64_t dhts_bam_set_overlap_bed(dhts_ctx *c, const uint8_t *text, uint64_t n) {
    if (!c) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    c->ov_active = false; c->ov_n = 0;
    if (!c->bam_open) return fail(c, "dhts_bam_open not called");
    if (n && !text) return fail(c, "overlap bed: null text");
    if (n >= 0xfffffff0ull) return fail(c, "overlap bed: text of 4 GiB or more");
    if (n == 0) return 0;
    DevBuf d_text, d_cnt, d_base, d_off, d_rows;
    ENSURE(c, d_text, n + 64);
    HIPCHK(c, hipMemcpyAsync(d_text.p, text, n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync((uint8_t *)d_text.p + n, 0, 64, c->stream));
    const uint8_t *u = (const uint8_t *)d_text.p;
    const int64_t nchunks = (int64_t)((n + VCF_CHUNK - 1) / VCF_CHUNK);
    ENSURE(c, d_cnt, (size_t)nchunks * 4 + 64); ENSURE(c, d_base, (size_t)(nchunks + 1) * 4 + 64);
    hipLaunchKernelGGL(vcf_line_count, dim3((unsigned)nchunks), dim3(256), 0, c->stream, u, (uint64_t)0, n, (uint32_t *)d_cnt.p, nchunks);
    const uint32_t *kin[1] = {(const uint32_t *)d_cnt.p}; uint32_t *kout[1] = {(uint32_t *)d_base.p}; uint64_t nl = 0;
    if (run_scan(c, 1, kin, kout, nullptr, nchunks, &nl)) return -1;
    ENSURE(c, d_off, (size_t)(nl + 2) * 4 + 64);
    hipLaunchKernelGGL(vcf_line_fill, dim3((unsigned)nchunks), dim3(256), 0, c->stream, u, (uint64_t)0, n, (const uint32_t *)d_base.p, (uint32_t *)d_off.p, nchunks);
    int64_t nlines = (int64_t)nl; int last_open = 0;
    if (text[n - 1] != '\n') { nlines++; last_open = 1; }
    ENSURE(c, d_rows, (size_t)nlines * sizeof(TbxLine) + 64);
    hipLaunchKernelGGL(bed_intervals, dim3((unsigned)((nlines + 255) / 256)), dim3(256), 0, c->stream, u, (const uint32_t *)d_off.p, nlines, n, last_open, (TbxLine *)d_rows.p);
    HIPCHK(c, hipGetLastError());
    std::vector<TbxLine> rows((size_t)nlines);
    HIPCHK(c, hipMemcpyAsync(rows.data(), d_rows.p, (size_t)nlines * sizeof(TbxLine), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::map<std::string, int32_t> tid_of;
    for (size_t t = 0; t < c->ref_name.size(); t++) tid_of.emplace(c->ref_name[t], (int32_t)t);   // (a name listed twice: the first @SQ, as sam_hdr_name2tid's hash keeps)
    std::vector<int32_t> tid; std::vector<int64_t> beg, end; tid.reserve((size_t)nlines); beg.reserve((size_t)nlines); end.reserve((size_t)nlines);
    int32_t cur = -1; int64_t lineno = 0;
    for (const TbxLine &r : rows) {
        ++lineno;
        if (r.flag == 1) continue;
        if (r.flag == 2) return fail(c, "read_bed: BED line has fewer than 3 tab-delimited fields (line %lld)", (long long)lineno);
        if (!r.same || tid.empty()) { auto it = tid_of.find(std::string((const char *)text + r.name_off, r.name_len)); cur = it == tid_of.end() ? -1 : it->second; }
        tid.push_back((r.flag & 12u) ? -1 : cur); beg.push_back(r.beg); end.push_back(r.end);
    }
    const int64_t m = (int64_t)tid.size();
    if (m && dhts_bam_set_overlap_intervals(c, tid.data(), beg.data(), end.data(), m)) return -1;
    return m;
}
// the same from a file: plain text, or BGZF / gzip members (inflated by a context of its own on the same device)
extern "C" int64_t dhts_bam_set_overlap_bed_path(dhts_ctx *c, const char *path) {
    if (!c) return -1;
    if (!path) return fail(c, "overlap bed: null path");
    std::vector<uint8_t> raw;
    {
        FILE *f = fopen(path, "rb");
        if (!f) return fail(c, "overlap bed: cannot open %s", path);
        uint8_t buf[1 << 16]; size_t got;
        while ((got = fread(buf, 1, sizeof(buf), f)) > 0) raw.insert(raw.end(), buf, buf + got);
        fclose(f);
    }
    if (raw.size() >= 18 && raw[0] == 0x1f && raw[1] == 0x8b) {
        dhts_ctx *t = dhts_create(c->device);
        if (!t) return fail(c, "overlap bed: no context for %s", path);
        std::vector<uint8_t> text; std::string err; bool ok = false;
        do {
            if (dhts_open_host(t, raw.data(), raw.size())) { err = dhts_error(t); break; }
            const int64_t nb = dhts_bgzf_index(t);
            if (nb < 0) { err = dhts_error(t); break; }
            text.resize((size_t)t->h_uoff[(size_t)nb] + 64);
            std::vector<int32_t> st((size_t)nb + 1, 0);
            const int64_t got = nb ? dhts_bgzf_inflate_to_host(t, 0, nb, text.data(), text.size(), st.data()) : 0;
            if (got < 0) { err = dhts_error(t); break; }
            text.resize((size_t)got); ok = true;
        } while (0);
        dhts_destroy(t);
        HIPCHK(c, hipSetDevice(c->device));
        if (!ok) return fail(c, "overlap bed: %s: %s", path, err.c_str());
        return dhts_bam_set_overlap_bed(c, text.data(), text.size());
    }
    return dhts_bam_set_overlap_bed(c, raw.data(), raw.size());
}

static int bam_overlap_join(dhts_ctx *c, const BamStream &st, const BamCols &bc, int64_t nrows, dhts_bam_batch *out) {
    out->ov_off = nullptr; out->ov_ids = nullptr; out->n_ov = 0;
    if (!c->ov_active) return 0;
    const size_t n = (size_t)(nrows > 0 ? nrows : 0);
    ENSURE(c, c->ov_cnt, (n + 1) * 4 + 64); ENSURE(c, c->ov_off, (n + 1) * 4 + 64);
    uint64_t total = 0;
    OverlapDev ov; ov.beg = (const int64_t *)c->ov_beg.p; ov.end = (const int64_t *)c->ov_end.p; ov.pmax = (const int64_t *)c->ov_pmax.p; ov.bmax = (const int64_t *)c->ov_bmax.p; ov.id = (const uint32_t *)c->ov_id.p;
    ov.tid_first = (const uint32_t *)c->ov_first.p; ov.n_ref = (int32_t)c->ref_name.size(); ov.pad = 0;
    if (nrows > 0) {
        {
            KTimer tm(c, DHTS_K_CORE);
            hipLaunchKernelGGL(bam_overlap_cells<false>, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, c->stream, st, ov, (const uint32_t *)c->rec_off.p, bc, nrows,
                               (uint32_t *)c->ov_cnt.p, (const uint32_t *)nullptr, (uint32_t *)nullptr);
        }
        const uint32_t *in[1] = {(const uint32_t *)c->ov_cnt.p}; uint32_t *o32[1] = {(uint32_t *)c->ov_off.p};
        { KTimer tm(c, DHTS_K_SCAN); if (run_scan(c, 1, in, o32, nullptr, nrows, &total)) return -1; }
        if (total > 0xfffffff0ull) return fail(c, "overlap join: more than 2^32 pairs in one batch");
        ENSURE(c, c->ov_ids, total * 4 + 64);
        if (total) {
            KTimer tm(c, DHTS_K_CORE);
            hipLaunchKernelGGL(bam_overlap_cells<true>, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, c->stream, st, ov, (const uint32_t *)c->rec_off.p, bc, nrows,
                               (uint32_t *)nullptr, (const uint32_t *)c->ov_off.p, (uint32_t *)c->ov_ids.p);
        }
        HIPCHK(c, hipGetLastError());
    } else {
        HIPCHK(c, hipMemsetAsync(c->ov_off.p, 0, 4, c->stream));
        ENSURE(c, c->ov_ids, 64);
    }
    out->ov_off = (const uint32_t *)c->ov_off.p; out->ov_ids = (const uint32_t *)c->ov_ids.p; out->n_ov = total;
    return 0;
}

static int bam_tag_columns(dhts_ctx *c, const BamStream &st, int64_t nrows, dhts_bam_batch *out) {
    const int nt = (int)c->tag_sel.size();
    c->tag_out.assign(nt, dhts_col());
    for (int i = 0; i < nt; i++) { memset(&c->tag_out[i], 0, sizeof(dhts_col)); c->tag_out[i].col = c->tag_sel[i]; c->tag_out[i].child_width = 8; }
    out->n_tag_cols = nt; out->tag_cols = c->tag_out.data();
    if (nt == 0 || nrows <= 0) return 0;
    const size_t n = (size_t)nrows;
    const uint32_t stride = (uint32_t)((n + 63) & ~(size_t)63), ostride = (uint32_t)((n + 1 + 63) & ~(size_t)63);
    std::vector<uint16_t> codes(nt);
    std::vector<TagColDev> cd(nt);
    int nsa = 0; size_t nfixed = 0; std::vector<size_t> fixed_at(nt, 0);
    for (int i = 0; i < nt; i++) {
        const StdTag &t = kStdTags[c->tag_sel[i]];
        codes[i] = (uint16_t)((uint8_t)t.tag[0] | ((uint8_t)t.tag[1] << 8));
        memset(&cd[i], 0, sizeof(TagColDev));
        cd[i].kind = t.type == 'H' ? 'Z' : t.type; cd[i].slot = i; cd[i].sa_cnt = cd[i].sa_bytes = -1;
        if (t.type == 'B') cd[i].sa_cnt = nsa++;
        else if (t.type == 'i') { fixed_at[i] = nfixed; nfixed += (n * 8 + 63) & ~(size_t)63; }
        else cd[i].sa_bytes = nsa++;
    }
    ENSURE(c, c->t_codes, nt * 2 + 16); ENSURE(c, c->t_dir, (size_t)nt * stride * 4 + 16); ENSURE(c, c->t_valid, (size_t)nt * n + 64); ENSURE(c, c->t_fixed, nfixed + 64);
    ENSURE(c, c->t_lens, (size_t)(nsa ? nsa : 1) * ostride * 4 + 64); ENSURE(c, c->t_offs, (size_t)(nsa ? nsa : 1) * ostride * 4 + 64); ENSURE(c, c->t_coldev, sizeof(TagColDev) * nt);
    for (int i = 0; i < nt; i++) { cd[i].valid = (uint8_t *)c->t_valid.p + (size_t)i * n; if (cd[i].kind == 'i') cd[i].fixed = (int64_t *)((uint8_t *)c->t_fixed.p + fixed_at[i]); }
    HIPCHK(c, hipMemcpyAsync(c->t_codes.p, codes.data(), nt * 2, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->t_coldev.p, cd.data(), sizeof(TagColDev) * nt, hipMemcpyHostToDevice, c->stream));
    TagCellArgs ta; memset(&ta, 0, sizeof(ta));
    ta.rec_off = (const uint32_t *)c->rec_off.p; ta.dir = (const uint32_t *)c->t_dir.p; ta.stride = stride; ta.nrows = nrows;
    ta.lens = (uint32_t *)c->t_lens.p; ta.offs = (const uint32_t *)c->t_offs.p; ta.ostride = ostride; ta.cols = (const TagColDev *)c->t_coldev.p;
    {
        KTimer tm(c, DHTS_K_CORE);
        hipLaunchKernelGGL(bam_tag_dir, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, c->stream, st, (const uint32_t *)c->rec_off.p, nrows, (const uint16_t *)c->t_codes.p, nt,
                           (uint32_t *)c->t_dir.p, stride);
        hipLaunchKernelGGL(bam_tag_cells<false>, dim3((unsigned)((nrows + 255) / 256), (unsigned)nt), dim3(256), 0, c->stream, st, ta);
    }
    std::vector<uint64_t> tot(nsa ? nsa : 1, 0);
    if (nsa > 0) {
        MScanArgs ma; ma.in = (const uint32_t *)c->t_lens.p; ma.out = (uint32_t *)c->t_offs.p; ma.stride = ostride; ma.n = nrows;
        ma.nparts = (nrows + 1 + SCAN_ITEMS - 1) / SCAN_ITEMS; if (ma.nparts < 1) ma.nparts = 1;
        ENSURE(c, c->t_partial, (size_t)nsa * ma.nparts * 8 + 64); ENSURE(c, c->t_total, (size_t)nsa * 8 + 64);
        ma.partial = (uint64_t *)c->t_partial.p; ma.total = (uint64_t *)c->t_total.p;
        {
            KTimer tm(c, DHTS_K_SCAN);
            hipLaunchKernelGGL(mscan_reduce, dim3((unsigned)ma.nparts, (unsigned)nsa), dim3(256), 0, c->stream, ma);
            hipLaunchKernelGGL(mscan_partials, dim3((unsigned)nsa), dim3(1024), 0, c->stream, ma);
            hipLaunchKernelGGL(mscan_apply, dim3((unsigned)ma.nparts, (unsigned)nsa), dim3(256), 0, c->stream, ma);
        }
        HIPCHK(c, hipMemcpyAsync(tot.data(), c->t_total.p, (size_t)nsa * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        size_t var = 0; std::vector<size_t> at(nt, 0);
        for (int i = 0; i < nt; i++) {
            if (cd[i].sa_cnt >= 0) { if (tot[cd[i].sa_cnt] >= (1ull << 32)) return fail(c, "tag column too large for one batch"); at[i] = var; var += (tot[cd[i].sa_cnt] * 8 + 63) & ~(size_t)63; }
            if (cd[i].sa_bytes >= 0) { if (tot[cd[i].sa_bytes] >= (1ull << 32)) return fail(c, "tag column too large for one batch"); at[i] = var; var += (tot[cd[i].sa_bytes] + 63) & ~(size_t)63; }
        }
        ENSURE(c, c->t_var, var + 64);
        for (int i = 0; i < nt; i++) {
            if (cd[i].sa_cnt >= 0) cd[i].child = (uint64_t *)((uint8_t *)c->t_var.p + at[i]);
            if (cd[i].sa_bytes >= 0) cd[i].bytes = (uint8_t *)c->t_var.p + at[i];
        }
        HIPCHK(c, hipMemcpyAsync(c->t_coldev.p, cd.data(), sizeof(TagColDev) * nt, hipMemcpyHostToDevice, c->stream));
        {
            KTimer tm(c, DHTS_K_STRINGS);
            hipLaunchKernelGGL(bam_tag_cells<true>, dim3((unsigned)((nrows + 255) / 256), (unsigned)nt), dim3(256), 0, c->stream, st, ta);
        }
    }
    HIPCHK(c, hipGetLastError());
    for (int i = 0; i < nt; i++) {
        dhts_col &o = c->tag_out[i];
        o.valid = cd[i].valid; o.fixed = cd[i].fixed;
        if (cd[i].sa_cnt >= 0) { o.off = (const uint32_t *)c->t_offs.p + (size_t)cd[i].sa_cnt * ostride; o.child_n = tot[cd[i].sa_cnt]; o.child_fixed = (const uint32_t *)cd[i].child; }
        if (cd[i].sa_bytes >= 0) { o.off = (const uint32_t *)c->t_offs.p + (size_t)cd[i].sa_bytes * ostride; o.bytes = cd[i].bytes; o.nbytes = tot[cd[i].sa_bytes]; }
    }
    return 0;
}

// ---- one batch of inflated bytes: carry + blocks [b0, b0+nb), shared by the read_bam and read_bcf drivers ---------------
struct Batch {
    int64_t b0 = 0, nb = 0; bool in_halo = false, last_of_stream = false, sharded_tail = false, final_batch = false;
    uint64_t carry = 0, ulen = 0, out_base = 0; uint8_t *u = nullptr; int blk_err = 0;
    bool status_pending = false;      // the blocks' status words have not been looked at yet (batch_status_resolve)
};
static const uint32_t NONE32 = 0xffffffffu;
// `defer`: do not wait for the inflate here.  The first damaged block is found on the device (bgzf_first_bad_block -> c->d_bstat) and the
// caller reads those four words together with its own first results, then calls batch_status_resolve: the record stage is queued behind
// the inflate without a host round trip, and is queued again on the shortened stream in the rare case that a block was bad.
static int batch_begin(dhts_ctx *c, int64_t max_blocks, Batch &B, bool defer = false) {
    if (max_blocks <= 0) max_blocks = 16384;
    if (max_blocks > 24576) max_blocks = 24576;               // keep every in-batch offset below 2^32
    B.sharded_tail = (c->shard_b1 < c->n_blocks);              // later shards exist: our last record may need halo blocks
    int64_t b0 = c->next_block;
    int64_t limit = c->shard_b1;
    bool in_halo = b0 >= c->shard_b1;
    if (in_halo) limit = c->n_blocks;
    int64_t nb = limit - b0; if (nb > max_blocks) nb = max_blocks;
    if (in_halo && nb > 4) nb = 4;
    if (nb < 0) nb = 0;
    if (c->first_batch && c->shard_rank != 0 && !in_halo) {
        // a shard that starts mid-stream finds its first record by speculation (candidates validated three records deep): give the
        // search at least 1 MiB of inflated bytes to validate against, whatever batch size the caller asked for
        while (b0 + nb < limit && c->h_uoff[b0 + nb] - c->h_uoff[b0] < (1u << 20) && nb < 24576) nb++;
    }
    B.b0 = b0; B.nb = nb; B.in_halo = in_halo;
    B.last_of_stream = (b0 + nb >= c->n_blocks) && !c->growing;        // (more blocks may still arrive: dhts_open_path_async)
    const uint64_t carry = c->carry_len;
    const uint64_t inflated = c->h_uoff[b0 + nb] - c->h_uoff[b0];
    uint64_t ulen = carry + inflated;
    if (ulen + PAD_BYTES >= (1ull << 32)) return fail(c, "batch too large");
    DevBuf &ub = c->ubuf[c->ucur];
    if (ub.cap < ulen + PAD_BYTES) {
        discard_prefetch(c);
        // grow while preserving the carry bytes at the front
        DevBuf nbuf; if (nbuf.ensure(ulen + PAD_BYTES)) return fail(c, "hipMalloc failed");
        if (carry) HIPCHK(c, hipMemcpyAsync(nbuf.p, ub.p, carry, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        ub.swap(nbuf);                                          // nbuf's destructor frees the old buffer
    }
    uint8_t *u = (uint8_t *)ub.p;
    const uint64_t out_base = c->h_uoff[b0] - carry;          // absolute stream offset of u[0]
    if (c->pf.valid && c->pf.b0 == b0 && c->pf.nb == nb && c->pf.carry == carry && c->pf.ucur == c->ucur) {
        // this batch's phase B was started during the previous batch's record stage: just order the streams
        c->pf.valid = false;
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->pf_done, 0));
    } else {
        // phase A may run ahead of this batch, but not beyond what the scan can reach: the shard (or index window) plus its halo
        const int64_t ahead = c->shard_b1 + 8 < c->n_blocks ? c->shard_b1 + 8 : c->n_blocks;
        if (inflate_blocks(c, b0, nb, u, out_base, ahead)) return -1;
        HIPCHK(c, hipMemsetAsync(u + ulen, 0, PAD_BYTES, c->stream));
    }
    // first bad block (if any) ends the byte stream there (bgzf.c:1241-1291: the read fails)
    int blk_err = 0;
    if (nb > 0 && defer) {
        ENSURE(c, c->d_bstat, 64);
        hipLaunchKernelGGL(bgzf_first_bad_block, dim3(1), dim3(1024), 0, c->stream, (const int32_t *)c->blk_status.p, b0, (int32_t)nb, (uint32_t *)c->d_bstat.p);
        B.status_pending = true;
    } else if (nb > 0) {
        std::vector<int32_t> bs(nb);
        for (int attempt = 0; ; attempt++) {
            HIPCHK(c, hipMemcpyAsync(bs.data(), (int32_t *)c->blk_status.p + b0, nb * 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            bool scratch = false;
            for (int64_t k = 0; k < nb; k++) if (bs[k] == DHTS_BLK_ERR_SCRATCH) { scratch = true; break; }
            if (!scratch) break;
            // the packed phase-A scratch was too small for some block (data that expands into far more tokens than a BAM or BCF does):
            // decode this range again with the full 152 KiB per block
            if (attempt > 0) return fail(c, "internal: phase-A scratch exhausted twice");
            discard_prefetch(c);
            c->pool_per_block = (int64_t)DHTS_LIT_STRIDE + (int64_t)DHTS_TOK_STRIDE * 4; c->huff_b0 = c->huff_nb = 0;
            const int64_t ahead = c->shard_b1 + 8 < c->n_blocks ? c->shard_b1 + 8 : c->n_blocks;
            if (inflate_blocks(c, b0, nb, u, out_base, ahead)) return -1;
            HIPCHK(c, hipMemsetAsync(u + ulen, 0, PAD_BYTES, c->stream));
        }
        for (int64_t k = 0; k < nb; k++) if (bs[k] != 0) { blk_err = bs[k]; ulen = carry + (c->h_uoff[b0 + k] - c->h_uoff[b0]); break; }
    }
    // (where only index windows are resident, the last resident block is not the end of the FILE: the record that runs out of it is cut by
    // the staging, not truncated)
    B.final_batch = (B.last_of_stream && !(c->partial_tail && !c->segs.empty())) || blk_err != 0;
    B.carry = carry; B.ulen = ulen; B.out_base = out_base; B.u = u; B.blk_err = blk_err;
    return 0;
}
// The four words of bgzf_first_bad_block have reached the host (`bst`).  Returns 0 when every block of the batch is good, 1 when the batch
// changed and the caller has to queue its record stage again (a damaged block cut the stream: B.ulen / B.final_batch / B.blk_err are
// updated; or the packed phase-A scratch was too small: the range has been inflated again with full-size room and the status is pending
// again), -1 on failure.
static int batch_status_resolve(dhts_ctx *c, Batch &B, const uint32_t bst[4], int attempt) {
    B.status_pending = false;
    if (bst[2] != NONE32) {
        if (attempt > 0) return fail(c, "internal: phase-A scratch exhausted twice");
        discard_prefetch(c);
        c->pool_per_block = (int64_t)DHTS_LIT_STRIDE + (int64_t)DHTS_TOK_STRIDE * 4; c->huff_b0 = c->huff_nb = 0;
        const int64_t ahead = c->shard_b1 + 8 < c->n_blocks ? c->shard_b1 + 8 : c->n_blocks;
        if (inflate_blocks(c, B.b0, B.nb, B.u, B.out_base, ahead)) return -1;
        HIPCHK(c, hipMemsetAsync(B.u + B.ulen, 0, PAD_BYTES, c->stream));
        hipLaunchKernelGGL(bgzf_first_bad_block, dim3(1), dim3(1024), 0, c->stream, (const int32_t *)c->blk_status.p, B.b0, (int32_t)B.nb, (uint32_t *)c->d_bstat.p);
        B.status_pending = true;
        return 1;
    }
    if (bst[0] != NONE32) {
        B.blk_err = (int)(int32_t)bst[1];
        B.ulen = B.carry + (c->h_uoff[B.b0 + bst[0]] - c->h_uoff[B.b0]);
        B.final_batch = true;
        return 1;
    }
    return 0;
}
// advance the scan position; *status as documented for dhts_bam_batch.status
static int batch_end(dhts_ctx *c, const Batch &B, uint64_t carry_start, bool rec_err, bool shard_finished, int32_t *status) {
    c->first_batch = false;
    c->next_block = B.b0 + B.nb;
    if (rec_err || B.blk_err || shard_finished || B.last_of_stream) discard_prefetch(c);
    if (rec_err || B.blk_err) { c->stream_done = true; *status = B.blk_err ? B.blk_err * 100 : -4; }
    else if (shard_finished) { c->stream_done = true; *status = 1; }
    else if (B.last_of_stream) { c->stream_done = true; *status = (c->bgzf_status != 0) ? c->bgzf_status : 1; if (carry_start < B.ulen && *status == 1 && !(c->partial_tail && !c->segs.empty())) *status = -4; }
    else {
        // move the incomplete tail to the front of the other buffer
        uint64_t tail = B.ulen - carry_start;
        DevBuf &nx = c->ubuf[c->ucur ^ 1];
        if (c->pf.valid && !(c->pf.carry == tail && c->pf.b0 == B.b0 + B.nb && c->pf.ucur == (c->ucur ^ 1))) discard_prefetch(c);
        if (!c->pf.valid) {                                   // (a live prefetch has already copied the tail there)
            ENSURE(c, nx, tail + PAD_BYTES);
            if (tail) HIPCHK(c, hipMemcpyAsync(nx.p, B.u + carry_start, tail, hipMemcpyDeviceToDevice, c->stream));
        }
        c->carry_len = tail; c->ucur ^= 1;
        if (B.in_halo && tail == 0) { c->stream_done = true; *status = 1; }
        if (B.b0 + B.nb >= c->shard_b1 && B.sharded_tail && tail == 0) { c->stream_done = true; *status = 1; }
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    timing_collect(c);
    return 0;
}

// ---- tabix index of a bgzipped line format other than VCF (tbx_index_build3 / tbx_index, tbx.c:437-541, with tbx_conf_bed / _gff / _sam or
// custom columns): the device finds the lines and their intervals (tabix_intervals), the host numbers the sequence names in order of first
// appearance and feeds hts_idx_push.  min_shift <= 0: TBI.  The context needs dhts_open_path + dhts_bgzf_index only.
extern "C" int64_t dhts_tabix_build_index(dhts_ctx *c, int preset, int sc, int bc, int ec, int meta_char, int line_skip, int min_shift) {
    if (!c) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->n_blocks <= 0 || c->plain_text) return fail(c, "tabix_index: the file is not BGZF");
    if ((preset & 0xffff) == 2) return fail(c, "tabix_index: the vcf preset goes through dhts_bcf_build_index");
    if ((preset & 0xffff) > 2) return fail(c, "tabix_index: preset not supported");
    discard_prefetch(c);
    c->shard_b0 = 0; c->shard_b1 = c->n_blocks; c->shard_rank = 0; c->shard_world = 1; c->wins.clear(); c->win_cur = 0; c->scan_end_uoff = ~0ull; c->rg_empty_window = false;
    c->next_block = 0; c->carry_len = 0; c->stream_done = false; c->first_batch = true; c->ucur = 0; c->huff_b0 = c->huff_nb = 0; c->scan_first_uoff = 0;
    const bool tbi = min_shift <= 0;
    if (tbi) min_shift = 14;
    const int64_t nb = c->n_blocks;
    auto tell = [&](uint64_t u) -> uint64_t {
        const uint64_t *uo = c->h_uoff.data();
        int64_t lo = 0, hi = nb + 1;
        while (lo < hi) { const int64_t mid = (lo + hi) / 2; if (uo[mid] < u) lo = mid + 1; else hi = mid; }
        if (lo <= nb && uo[lo] == u) return (lo < nb ? c->h_coff[lo] : c->comp_len) << 16;
        return (c->h_coff[lo - 1] << 16) | (u - uo[lo - 1]);
    };
    IndexAcc ib; ib.grow = true; bool inited = false; uint64_t last_off = tell(0); int64_t lineno = 0, max_ref_len = 0;
    std::vector<int32_t> a_tid; std::vector<int64_t> a_beg, a_end; std::vector<uint64_t> a_v; std::vector<uint8_t> a_map;     // the data lines of a batch as arrays
    auto init_index = [&]() {
        if (tbi) ib.tbi = true;
        else {
            int n_lvls = (31 - min_shift + 2) / 3;
            if (max_ref_len) {
                const int64_t need = max_ref_len + 256;
                if (need <= (1ll << (min_shift + 27))) { int64_t maxpos = 1ll << (min_shift + 3 * n_lvls); while (need > maxpos) { ++n_lvls; maxpos *= 8; } }
                else { n_lvls = 9; int64_t maxpos = 1ll << (min_shift + 27); while (need > maxpos) { ++min_shift; maxpos *= 2; } }
            } else n_lvls = min_shift < 10 ? 9 : min_shift < 25 ? 9 - (min_shift - 10) / 3 : 4;
            ib.set_csi(min_shift, n_lvls);
        }
        ib.begin(0, last_off); inited = true;
    };
    std::vector<std::string> names; std::map<std::string, int32_t> tid_of; int32_t last_tid = -1;
    std::vector<TbxLine> rows; std::vector<uint32_t> lo; std::string nm;
    TbxConf cf = {preset, sc, bc, ec, meta_char, line_skip};
    int rc = 0; std::string err;
    for (;;) {
        Batch B;
        if (batch_begin(c, 0, B)) return -1;
        const uint8_t *u = B.u; const uint64_t ulen = B.ulen, out_base = B.out_base;
        uint64_t carry_start = ulen; int64_t nlines = 0; int last_open = 0;
        if (ulen > 0) {
            const int64_t nchunks = (int64_t)((ulen + VCF_CHUNK - 1) / VCF_CHUNK);
            ENSURE(c, c->v_cnt, (size_t)nchunks * 4 + 64); ENSURE(c, c->v_base, (size_t)(nchunks + 1) * 4 + 64);
            hipLaunchKernelGGL(vcf_line_count, dim3((unsigned)nchunks), dim3(256), 0, c->stream, u, (uint64_t)0, ulen, (uint32_t *)c->v_cnt.p, nchunks);
            const uint32_t *kin[1] = {(const uint32_t *)c->v_cnt.p}; uint32_t *kout[1] = {(uint32_t *)c->v_base.p}; uint64_t nl = 0;
            if (run_scan(c, 1, kin, kout, nullptr, nchunks, &nl)) return -1;
            if (nl + 2 >= (1ull << 32)) return fail(c, "batch too large");
            ENSURE(c, c->v_line_off, (size_t)(nl + 2) * 4 + 64);
            hipLaunchKernelGGL(vcf_line_fill, dim3((unsigned)nchunks), dim3(256), 0, c->stream, u, (uint64_t)0, ulen, (const uint32_t *)c->v_base.p, (uint32_t *)c->v_line_off.p, nchunks);
            lo.resize((size_t)nl + 2);
            HIPCHK(c, hipMemcpyAsync(lo.data(), c->v_line_off.p, (size_t)(nl + 1) * 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            nlines = (int64_t)nl; carry_start = lo[nl];
            if (B.final_batch && lo[nl] < ulen) { nlines++; last_open = 1; carry_start = ulen; lo[nl + 1] = (uint32_t)ulen; }
        }
        if (nlines > 0) {
            ENSURE(c, c->v_undef, (size_t)nlines * sizeof(TbxLine) + 64);
            hipLaunchKernelGGL(tabix_intervals, dim3((unsigned)((nlines + 255) / 256)), dim3(256), 0, c->stream, u, (const uint32_t *)c->v_line_off.p, nlines, ulen, last_open, cf, (TbxLine *)c->v_undef.p);
            rows.resize((size_t)nlines);
            HIPCHK(c, hipMemcpyAsync(rows.data(), c->v_undef.p, (size_t)nlines * sizeof(TbxLine), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            a_tid.clear(); a_beg.clear(); a_end.clear(); a_v.clear();
            for (int64_t i = 0; i < nlines && rc == 0; i++) {
                const TbxLine &r = rows[(size_t)i];
                ++lineno;
                const uint64_t after = tell(out_base + ((last_open && i + 1 == nlines) ? ulen : (uint64_t)lo[(size_t)i + 1]));
                if (lineno <= line_skip || r.flag == 1) {
                    if (r.flag == 1 && !tbi && (preset & 0xffff) == 1) {                   // adjust_max_ref_len_sam (tbx.c:425-435): @SQ ... LN:
                        const uint32_t l0 = lo[(size_t)i], l1 = (last_open && i + 1 == nlines) ? (uint32_t)ulen : lo[(size_t)i + 1] - 1u;
                        std::string line(l1 - l0, '\0');
                        if (l1 > l0) HIPCHK(c, hipMemcpy(&line[0], u + l0, l1 - l0, hipMemcpyDeviceToHost));
                        if (line.compare(0, 3, "@SQ") == 0) { const size_t q = line.find("\tLN:", 3); if (q != std::string::npos) { const long long len = strtoll(line.c_str() + q + 4, nullptr, 10); if (len > max_ref_len) max_ref_len = len; } }
                    }
                    last_off = after; continue;
                }
                if (!inited) init_index();
                if (r.flag == 2) { rc = -1; err = "Failed to parse the line (was the wrong preset used?)"; break; }
                int32_t tid = last_tid;
                if (!r.same || last_tid < 0) {
                    nm.assign(r.name_len, '\0');
                    if (r.name_len) HIPCHK(c, hipMemcpy(&nm[0], u + r.name_off, r.name_len, hipMemcpyDeviceToHost));
                    auto it = tid_of.find(nm);
                    if (it == tid_of.end()) { tid = (int32_t)names.size(); tid_of[nm] = tid; names.push_back(nm); } else tid = it->second;
                    last_tid = tid;
                }
                a_tid.push_back(tid); a_beg.push_back(r.beg); a_end.push_back(r.end); a_v.push_back(after);
            }
            if (rc == 0 && !a_tid.empty()) {
                a_map.assign(a_tid.size(), 1);
                if (!ib.add_rows(a_tid.data(), a_beg.data(), a_end.data(), a_v.data(), a_map.data(), (int64_t)a_tid.size())) { rc = -1; err = ib.err; }
            }
        }
        if (rc) break;
        int32_t status = 0;
        if (batch_end(c, B, carry_start, false, false, &status)) return -1;
        if (status == 1) break;
        if (status < 0) { rc = -1; err = "the BGZF stream ended on an error"; break; }
    }
    discard_prefetch(c);
    c->next_block = 0; c->carry_len = 0; c->stream_done = false; c->first_batch = true; c->ucur = 0;
    if (rc) return fail(c, "tabix_index: %s", err.c_str());
    if (!inited) init_index();
    uint64_t fin = c->comp_len;
    if (nb > 0 && c->h_isize[nb - 1] == 0) fin = c->h_coff[nb - 1];
    if (!ib.finish(fin << 16)) return fail(c, "tabix_index: %s", ib.err.c_str());
    {
        const uint32_t conf[6] = {(uint32_t)preset, (uint32_t)sc, (uint32_t)bc, (uint32_t)ec, (uint32_t)meta_char, (uint32_t)line_skip}; uint32_t l_nm = 0;
        for (auto &x : names) l_nm += (uint32_t)x.size() + 1;
        auto w32 = [&](uint32_t x) { for (int k = 0; k < 4; k++) ib.aux.push_back((uint8_t)(x >> (8 * k))); };
        for (uint32_t x : conf) w32(x);
        w32(l_nm);
        for (auto &x : names) { ib.aux.insert(ib.aux.end(), x.begin(), x.end()); ib.aux.push_back(0); }
    }
    ib.save(c->built_index);
    return (int64_t)c->built_index.size();
}

static int bam_next_batch_one(dhts_ctx *c, int64_t max_blocks, uint32_t colmask, dhts_bam_batch *out);
int dhts_bam_next_batch(dhts_ctx *c, int64_t max_blocks, uint32_t colmask, dhts_bam_batch *out) {
    if (!c || !out) return -1;
    for (;;) {
        if (bam_next_batch_one(c, max_blocks, colmask, out)) return -1;
        // a region query with several index windows: the end of one window is the start of the next, not the end of the scan
        if (out->status == 1 && c->win_cur + 1 < c->wins.size()) {
            enter_window(c, c->win_cur + 1);
            discard_prefetch(c);
            c->next_block = c->shard_b0; c->carry_len = 0; c->stream_done = false; c->first_batch = true; c->ucur = 0;
            out->status = 0;
            if (out->n_rows == 0) continue;
        }
        return 0;
    }
}
static int bam_next_batch_one(dhts_ctx *c, int64_t max_blocks, uint32_t colmask, dhts_bam_batch *out) {
    if (!c || !out) return -1;
    memset(out, 0, sizeof(*out));
    if (!c->bam_open) return fail(c, "dhts_bam_open not called");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->stream_done) { out->status = 1; return 0; }
    // Host round trips of one batch (the plain case: no region filter, no shard cut inside the batch): (1) behind the tile pass -- rows,
    // carry position, the repair rounds' verdict and the inflate's block status in ONE copy; (2) behind the string scan -- heap sizes,
    // first invalid row, first record offset; (3) the end of the batch.  The inflate, the tile scan, two repair rounds and the finalize
    // pass are queued back to back; so are unpack, validity packing and the scans.
    Batch B;
    if (batch_begin(c, max_blocks, B, true)) return -1;
    const bool sharded_tail = B.sharded_tail;
    bool final_batch = B.final_batch;
    uint8_t *u = B.u; uint64_t ulen = B.ulen; const uint64_t out_base = B.out_base;
    BamStream st; st.u = u; st.ulen = ulen; st.n_ref = (int32_t)c->ref_name.size(); st.final_batch = final_batch ? 1 : 0; st.seq_packed = c->seq_packed ? 1 : 0;
    st.want_rg = (colmask & ((1u << DHTS_BAM_READ_GROUP_ID) | (1u << DHTS_BAM_SAMPLE_ID))) ? 1 : 0;
    c->last_stream = st;

    // ---- tiles ----
    int64_t ntiles = (int64_t)((ulen + TILE_BYTES - 1) / TILE_BYTES); if (ntiles < 1) ntiles = 1;
    ENSURE(c, c->t_first, ntiles * 8); ENSURE(c, c->t_end, ntiles * 8); ENSURE(c, c->t_count, ntiles * 4); ENSURE(c, c->t_err, ntiles * 4);
    ENSURE(c, c->t_rowbase, ntiles * 4 + 16); ENSURE(c, c->d_res, 64); ENSURE(c, c->d_nfixed, 64);
    ENSURE(c, c->t_recs, (size_t)ntiles * TL_RECS * 2 + 64); ENSURE(c, c->t_recs_first, ntiles * 8 + 64);
    ENSURE(c, c->t2_first, ntiles * 8); ENSURE(c, c->t2_end, ntiles * 8); ENSURE(c, c->t2_count, ntiles * 4); ENSURE(c, c->t2_err, ntiles * 4);
    TileOut to; to.first = (uint64_t *)c->t_first.p; to.end_next = (uint64_t *)c->t_end.p; to.count = (uint32_t *)c->t_count.p; to.err = (int32_t *)c->t_err.p;
    TileOut to2; to2.first = (uint64_t *)c->t2_first.p; to2.end_next = (uint64_t *)c->t2_end.p; to2.count = (uint32_t *)c->t2_count.p; to2.err = (int32_t *)c->t2_err.p;
    uint64_t start0;
    if (c->first_batch) start0 = (c->shard_rank == 0) ? c->scan_first_uoff - out_base : NONE64;   // later shards speculate their first record
    else start0 = 0;                                           // the carry begins on a record boundary
    if (c->first_batch && c->shard_rank == 0 && c->scan_first_uoff < out_base) return fail(c, "internal: header beyond first batch");
    uint64_t res[4] = {0, 0, 0, 0};
    uint64_t first0 = NONE64;
    // A shard that starts mid-stream speculates its first record.  If the chain that grows from the candidate breaks inside this
    // batch, the candidate was a false start (or the file is damaged): resume the search behind it.  The true first record always
    // survives; when every retry fails as well the damage is real and the first attempt's result stands.
    const bool speculative = (start0 == NONE64);
    uint64_t spec_from = 0; int spec_tries = 0; bool restoring = false;
    int status_attempt = 0;
    for (;;) {
        to.first = (uint64_t *)c->t_first.p; to.end_next = (uint64_t *)c->t_end.p; to.count = (uint32_t *)c->t_count.p; to.err = (int32_t *)c->t_err.p;
        to2.first = (uint64_t *)c->t2_first.p; to2.end_next = (uint64_t *)c->t2_end.p; to2.count = (uint32_t *)c->t2_count.p; to2.err = (int32_t *)c->t2_err.p;
        uint32_t nfh[2] = {0, 0}, bst[4] = {NONE32, 0, NONE32, 0};
        {
            KTimer tm(c, DHTS_K_TILES);
            hipLaunchKernelGGL(bam_tile_scan, dim3((unsigned)ntiles), dim3(64), 0, c->stream, st, start0, ntiles, to, (uint16_t *)c->t_recs.p, (uint64_t *)c->t_recs_first.p, spec_from);
            // two repair rounds are queued unconditionally (a 30x BAM converges in two; a round that finds nothing to repair copies the table)
            uint32_t *nf = (uint32_t *)c->d_nfixed.p;
            (void)hipMemsetAsync(nf, 0, 8, c->stream);
            for (int r = 0; r < 2; r++) {
                hipLaunchKernelGGL(bam_tile_fix, dim3((unsigned)((ntiles + 255) / 256)), dim3(256), 0, c->stream, st, TILE_BYTES, ntiles, to, to2, nf + r);
                TileOut tmp = to; to = to2; to2 = tmp;             // the round's output is the current table
            }
            hipLaunchKernelGGL(bam_tile_finalize, dim3(1), dim3(1024), 0, c->stream, ntiles, to, (uint32_t *)c->t_rowbase.p, (uint64_t *)c->d_res.p);
        }
        HIPCHK(c, hipMemcpyAsync(res, c->d_res.p, 32, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(&first0, to.first, 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(nfh, c->d_nfixed.p, 8, hipMemcpyDeviceToHost, c->stream));
        if (B.status_pending) HIPCHK(c, hipMemcpyAsync(bst, c->d_bstat.p, 16, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (B.status_pending) {
            const int rs = batch_status_resolve(c, B, bst, status_attempt++);
            if (rs < 0) return -1;
            if (rs > 0) {           // the stream is shorter than assumed (or was inflated again): the tile pass starts over
                ulen = B.ulen; final_batch = B.final_batch;
                st.ulen = ulen; st.final_batch = final_batch ? 1 : 0; c->last_stream = st;
                ntiles = (int64_t)((ulen + TILE_BYTES - 1) / TILE_BYTES); if (ntiles < 1) ntiles = 1;
                spec_from = 0; spec_tries = 0; restoring = false;
                continue;
            }
        }
        if (g_debug) fprintf(stderr, "[dhts] tiles=%lld nfixed=%u,%u\n", (long long)ntiles, nfh[0], nfh[1]);
        if (nfh[1] != 0) {
            // the second round still repaired tiles: go on round by round, then finalize again
            KTimer tm(c, DHTS_K_TILES);
            int rounds = 2;
            for (;;) {
                (void)hipMemsetAsync(c->d_nfixed.p, 0, 4, c->stream);
                hipLaunchKernelGGL(bam_tile_fix, dim3((unsigned)((ntiles + 255) / 256)), dim3(256), 0, c->stream, st, TILE_BYTES, ntiles, to, to2, (uint32_t *)c->d_nfixed.p);
                { TileOut tmp = to; to = to2; to2 = tmp; }
                uint32_t nfixed = 0;
                HIPCHK(c, hipMemcpyAsync(&nfixed, c->d_nfixed.p, 4, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(c, hipStreamSynchronize(c->stream));
                if (g_debug) fprintf(stderr, "[dhts] tiles=%lld round=%d nfixed=%u\n", (long long)ntiles, rounds, nfixed);
                if (nfixed == 0) break;
                if (++rounds > 256) { hipLaunchKernelGGL(bam_tile_fix_seq, dim3(1), dim3(1), 0, c->stream, st, TILE_BYTES, ntiles, to); break; }
            }
            hipLaunchKernelGGL(bam_tile_finalize, dim3(1), dim3(1024), 0, c->stream, ntiles, to, (uint32_t *)c->t_rowbase.p, (uint64_t *)c->d_res.p);
            HIPCHK(c, hipMemcpyAsync(res, c->d_res.p, 32, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipMemcpyAsync(&first0, to.first, 8, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
        if (!speculative || restoring) break;
        bool false_start = res[2] != 0 && first0 != NONE64;
        const bool exhausted = spec_tries > 0 && first0 == NONE64;
        if (!false_start && !exhausted && first0 != NONE64 && (int64_t)res[0] > 0) {
            // the chain holds: do its "records" also pass bam_read1's full validation?  (only speculated shard starts pay for this pass)
            const int64_t nr = (int64_t)res[0];
            unsigned long long bad0 = ~0ull;
            ENSURE(c, c->rec_off, (size_t)nr * 4 + 16);
            HIPCHK(c, hipMemsetAsync((uint64_t *)c->d_res.p + 4, 0xff, 8, c->stream));
            hipLaunchKernelGGL(bam_tile_offsets, dim3((unsigned)((ntiles + 255) / 256)), dim3(256), 0, c->stream, st, TILE_BYTES, ntiles, to, (const uint32_t *)c->t_rowbase.p,
                               (const uint64_t *)c->d_res.p, (uint32_t *)c->rec_off.p);
            hipLaunchKernelGGL(bam_validate_rows, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, c->stream, st, (const uint32_t *)c->rec_off.p, nr, (unsigned long long *)((uint64_t *)c->d_res.p + 4));
            HIPCHK(c, hipMemcpyAsync(&bad0, (uint64_t *)c->d_res.p + 4, 8, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            false_start = bad0 < (unsigned long long)nr;
        }
        if (!false_start && !exhausted) break;
        if (exhausted || spec_tries == 16) { spec_from = 0; restoring = true; continue; }
        spec_from = first0 + 1; spec_tries++;
    }
    int64_t nrows = (int64_t)res[0]; uint64_t carry_start = res[1]; bool rec_err = res[2] != 0;
    if (carry_start == NONE64) carry_start = ulen;             // nothing recognisable in this batch

    // sharding: rows belong to this shard iff their record STARTS before the shard's end in the inflated stream
    uint64_t shard_end_u = sharded_tail ? c->h_uoff[c->shard_b1] : ~0ull;
    // ---- phase B of the NEXT batch, concurrent with the rest of this one (DHTS_PREFETCH=1; off by default since round 3) ----
    // The carry (where this batch's last complete record ends) is known now, so the next batch's destination is too.  Only the
    // plain case is prefetched: the stream goes on inside this shard, phase A already covers the blocks, nothing can cut this
    // batch short from here on except a bad row (then batch_end discards the prefetch).
    // Measured on the 92 M-record bench file: with the prefetch 173.0-174.5 ms per step, without 176.2 ms -- phase B and the record stage
    // are both bound by instruction issue, so running them side by side only stretches both (bgzf_lz_resolve 4.8 ms per launch beside
    // the record stage, 2.8 ms alone; bam_tile_unpack 1.35 / 0.88 ms), and the 1 % it gains is not worth a second stream, a second set of
    // buffers in flight and per-kernel times that no longer say what a kernel costs.
    {
        const bool no_pf = !(getenv("DHTS_PREFETCH") && atoi(getenv("DHTS_PREFETCH")) != 0) || getenv("DHTS_NO_PREFETCH") != nullptr;
        int64_t mb = max_blocks <= 0 ? 16384 : (max_blocks > 24576 ? 24576 : max_blocks);
        const int64_t nb0 = B.b0 + B.nb;
        int64_t nbn = c->shard_b1 - nb0; if (nbn > mb) nbn = mb;
        const bool plain = !no_pf && !rec_err && !B.blk_err && !B.last_of_stream && !B.in_halo && nbn > 0 && !(sharded_tail && out_base + ulen > shard_end_u) &&
                           (!inflate_split() || (nb0 >= c->huff_b0 && nb0 + nbn <= c->huff_b0 + c->huff_nb)) && !c->pf.valid;
        if (plain) {
            const uint64_t tail = ulen - carry_start, ulen_n = tail + (c->h_uoff[nb0 + nbn] - c->h_uoff[nb0]);
            bool ok = ulen_n + PAD_BYTES < (1ull << 32);
            DevBuf &nx = c->ubuf[c->ucur ^ 1];
            if (ok && nx.cap < ulen_n + PAD_BYTES && nx.ensure(ulen_n + PAD_BYTES)) ok = false;
            if (ok) {
                uint8_t *un = (uint8_t *)nx.p;
                if (tail) HIPCHK(c, hipMemcpyAsync(un, u + carry_start, tail, hipMemcpyDeviceToDevice, c->stream_b));
                if (inflate_split() ? launch_lz(c, nb0, nbn, un, c->h_uoff[nb0] - tail, c->stream_b) : launch_fused(c, nb0, nbn, un, c->h_uoff[nb0] - tail, c->stream_b, 1)) return -1;
                HIPCHK(c, hipMemsetAsync(un + ulen_n, 0, PAD_BYTES, c->stream_b));
                HIPCHK(c, hipEventRecord(c->pf_done, c->stream_b));
                c->pf.valid = true; c->pf.b0 = nb0; c->pf.nb = nbn; c->pf.carry = tail; c->pf.ucur = c->ucur ^ 1;
            }
        }
    }
    bool shard_finished = false;

    // ---- rows ----
    BamCols bc; memset(&bc, 0, sizeof(bc));
    const int64_t nrows_scan = nrows;                          // rows the tile pass counted (before the region filter / a bad row / the shard cut)
    const uint32_t *row_map_s = nullptr;
    uint32_t first_rec_rel = 0; bool have_first_rec = false;   // offset of the batch's first row (read with the heap sizes when there are any)
    if (nrows > 0) {
        size_t n = (size_t)nrows;
        ENSURE(c, c->rec_off, n * 4 + 16); ENSURE(c, c->c_rgflag, n + 64);
        ENSURE(c, c->c_flag, n * 2 + 16); ENSURE(c, c->c_pos, n * 8); ENSURE(c, c->c_mapq, n * 4); ENSURE(c, c->c_pnext, n * 8); ENSURE(c, c->c_tlen, n * 8);
        ENSURE(c, c->c_tid, n * 4); ENSURE(c, c->c_mtid, n * 4); ENSURE(c, c->c_rgidx, n * 4); ENSURE(c, c->c_rgvalid, (n / 64 + 2) * 8);
        ENSURE(c, c->l_qname, n * 4 + 16); ENSURE(c, c->l_cigar, n * 4 + 16); ENSURE(c, c->l_seq, n * 4 + 16); ENSURE(c, c->l_qual, n * 4 + 16); ENSURE(c, c->l_rg, n * 4 + 16);
        ENSURE(c, c->cig_rel, n * 4); ENSURE(c, c->ncig_eff, n * 4); ENSURE(c, c->rg_rel, n * 4); ENSURE(c, c->alen_qual, n * 4);
        ENSURE(c, c->o_qname, (n + 1) * 4 + 16); ENSURE(c, c->o_cigar, (n + 1) * 4 + 16); ENSURE(c, c->o_seq, (n + 1) * 4 + 16); ENSURE(c, c->o_qual, (n + 1) * 4 + 16); ENSURE(c, c->o_rg, (n + 1) * 4 + 16);
        BamDict dict; dict.n_rg = (int32_t)c->rg_id.size(); dict.rg_off = (const uint32_t *)c->d_rg_off.p; dict.rg_bytes = (const uint8_t *)c->d_rg_bytes.p;
        bc.flag = (uint16_t *)c->c_flag.p; bc.pos = (int64_t *)c->c_pos.p; bc.mapq = (int32_t *)c->c_mapq.p; bc.pnext = (int64_t *)c->c_pnext.p;
        bc.tlen = (int64_t *)c->c_tlen.p; bc.tid = (int32_t *)c->c_tid.p; bc.mtid = (int32_t *)c->c_mtid.p; bc.rg_idx = (int32_t *)c->c_rgidx.p;
        bc.rg_valid = (uint64_t *)c->c_rgvalid.p; bc.len_qname = (uint32_t *)c->l_qname.p; bc.len_cigar = (uint32_t *)c->l_cigar.p; bc.len_seq = (uint32_t *)c->l_seq.p;
        bc.len_qual = (uint32_t *)c->l_qual.p; bc.len_rg = (uint32_t *)c->l_rg.p; bc.cig_rel = (uint32_t *)c->cig_rel.p; bc.ncig_eff = (uint32_t *)c->ncig_eff.p; bc.rg_rel = (uint32_t *)c->rg_rel.p;
        HIPCHK(c, hipMemsetAsync((uint64_t *)c->d_res.p + 4, 0xff, 8, c->stream));     // first invalid row (none)
        const bool filtered = c->rg_active && !c->rg_all;
        const uint32_t *row_map = nullptr;
        uint64_t kept_total = (uint64_t)nrows;
        if (filtered) {
            // region predicate per record, then a scan turns the keep flags into compacted row ids
            ENSURE(c, c->c_keep, n * 4 + 16); ENSURE(c, c->c_rowmap, (n + 1) * 4 + 16);
            RegionDev rg; rg.beg = (const int64_t *)c->d_rg_beg.p; rg.end = (const int64_t *)c->d_rg_end.p; rg.tid_first = (const uint32_t *)c->d_rg_first.p;
            rg.n_ref = (int32_t)c->ref_name.size(); rg.all = 0; rg.nocoor = c->rg_nocoor ? 1 : 0; rg.pad = 0;
            {
                KTimer tm(c, DHTS_K_CORE);
                hipLaunchKernelGGL(bam_tile_offsets, dim3((unsigned)((ntiles + 255) / 256)), dim3(256), 0, c->stream, st, TILE_BYTES, ntiles, to, (const uint32_t *)c->t_rowbase.p,
                                   (const uint64_t *)c->d_res.p, (uint32_t *)c->rec_off.p);
                const uint64_t end_rel = c->scan_end_uoff == ~0ull ? ~0ull : (c->scan_end_uoff > out_base ? c->scan_end_uoff - out_base : 0ull);
                hipLaunchKernelGGL(bam_region_keep, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, c->stream, st, rg, (const uint32_t *)c->rec_off.p, nrows, (uint32_t *)c->c_keep.p, end_rel);
            }
            const uint32_t *kin[1] = {(const uint32_t *)c->c_keep.p}; uint32_t *kout[1] = {(uint32_t *)c->c_rowmap.p};
            { KTimer tm(c, DHTS_K_SCAN); if (run_scan(c, 1, kin, kout, nullptr, nrows, &kept_total)) return -1; }
            row_map = (const uint32_t *)c->c_rowmap.p; row_map_s = row_map;
        }
        {
            KTimer tm(c, DHTS_K_CORE);
            hipLaunchKernelGGL(bam_tile_unpack, dim3((unsigned)ntiles), dim3(64), 0, c->stream, st, dict, ntiles, to, (const uint32_t *)c->t_rowbase.p,
                               (const uint64_t *)c->d_res.p, nrows, (uint32_t *)c->rec_off.p, (uint8_t *)c->c_rgflag.p, bc, (unsigned long long *)((uint64_t *)c->d_res.p + 4), row_map,
                               (const uint16_t *)c->t_recs.p, (const uint64_t *)c->t_recs_first.p);
        }
        // the first row that fails bam_read1's validation ends the scan there (rows before it are kept).  In the plain case the verdict is
        // read together with the heap sizes below; a region filter or a shard cut inside the batch need it now.
        bool bad_pending = true;
        unsigned long long bad = ~0ull;
        const bool shard_cut = !filtered && sharded_tail && out_base + ulen > shard_end_u;
        auto apply_bad = [&]() -> int {
            if (bad < (unsigned long long)nrows) {
                rec_err = true;
                if (filtered) { uint32_t kb = 0; HIPCHK(c, hipMemcpy(&kb, (const uint32_t *)c->c_rowmap.p + bad, 4, hipMemcpyDeviceToHost)); kept_total = kb; }
                nrows = (int64_t)bad;
            }
            return 0;
        };
        if (filtered || shard_cut) {
            HIPCHK(c, hipMemcpyAsync(&bad, (uint64_t *)c->d_res.p + 4, 8, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            bad_pending = false;
            if (apply_bad()) return -1;
        }
        if (filtered) {
            if (sharded_tail && out_base + ulen > shard_end_u && carry_start >= shard_end_u - out_base) shard_finished = true;   // rows past the index window never match
            // the window's exact end lies inside this batch: the window is done, whatever follows it (with only the windows resident the
            // last one ends in the last resident block, and the records behind the cut are not a truncated tail)
            if (c->scan_end_uoff != ~0ull && out_base + ulen >= c->scan_end_uoff && !rec_err) shard_finished = true;
            nrows = (int64_t)kept_total;
        } else if (nrows > 0 && shard_cut) {
            // drop rows whose record starts at/after the shard end (they belong to the next shard): binary search on rec_off
            std::vector<uint32_t> ro(nrows);
            HIPCHK(c, hipMemcpyAsync(ro.data(), c->rec_off.p, nrows * 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            uint64_t lim = shard_end_u - out_base;
            int64_t lo = 0, hi = nrows;
            while (lo < hi) { int64_t mid = (lo + hi) / 2; if (ro[mid] < lim) lo = mid + 1; else hi = mid; }
            if (lo < nrows) { carry_start = ro[lo]; nrows = lo; shard_finished = true; }
            else if (carry_start >= lim) shard_finished = true;
        }
        if (nrows > 0) {
            {
                KTimer tm(c, DHTS_K_CORE);
                hipLaunchKernelGGL(bam_pack_validity, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, c->stream, (const uint8_t *)c->c_rgflag.p, nrows, (uint64_t *)c->c_rgvalid.p);
            }
            uint32_t *o32[5] = {(uint32_t *)c->o_qname.p, (uint32_t *)c->o_cigar.p, (uint32_t *)c->o_seq.p, (uint32_t *)c->o_qual.p, (uint32_t *)c->o_rg.p};
            uint64_t tot[5] = {0, 0, 0, 0, 0};
            BamStrOut so; memset(&so, 0, sizeof(so));
            so.off_qname = o32[0]; so.off_cigar = o32[1]; so.off_seq = o32[2]; so.off_qual = o32[3]; so.off_rg = o32[4]; so.alen_qual = (uint32_t *)c->alen_qual.p;
            if (c->seq_packed && (colmask & (1u << DHTS_BAM_SEQ))) { ENSURE(c, c->seq_chars, (size_t)nrows * 4 + 64); so.seq_chars = (uint32_t *)c->seq_chars.p; }
            const uint32_t str_cols = (1u << DHTS_BAM_QNAME) | (1u << DHTS_BAM_CIGAR) | (1u << DHTS_BAM_SEQ) | (1u << DHTS_BAM_QUAL) | (1u << DHTS_BAM_READ_GROUP_ID);
            const bool strings = (colmask & str_cols) != 0;     // projection pushdown: with no string column projected the whole string pass is skipped
            if (strings) {
                const uint32_t *in[5] = {bc.len_qname, bc.len_cigar, bc.len_seq, bc.len_qual, bc.len_rg};
                KTimer tm(c, DHTS_K_SCAN);
                if (run_scan(c, 5, in, o32, nullptr, nrows, nullptr)) return -1;
                HIPCHK(c, hipMemcpyAsync(tot, c->scan_total.p, 40, hipMemcpyDeviceToHost, c->stream));
            }
            if (strings || bad_pending) {
                // ONE round trip: heap sizes, the first invalid row and the first record's offset
                if (bad_pending) HIPCHK(c, hipMemcpyAsync(&bad, (uint64_t *)c->d_res.p + 4, 8, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(c, hipMemcpyAsync(&first_rec_rel, c->rec_off.p, 4, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(c, hipStreamSynchronize(c->stream));
                have_first_rec = true;
                if (bad_pending) {
                    bad_pending = false;
                    const int64_t before = nrows;
                    if (apply_bad()) return -1;
                    if (nrows != before && nrows > 0) {
                        // rare: a row failed validation.  The offsets of the rows in front of it stand (exclusive sums); the heap sizes and the
                        // validity words are made again for the shorter table
                        hipLaunchKernelGGL(bam_pack_validity, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, c->stream, (const uint8_t *)c->c_rgflag.p, nrows, (uint64_t *)c->c_rgvalid.p);
                        if (strings) {
                            uint32_t t32[5] = {0, 0, 0, 0, 0};
                            for (int k = 0; k < 5; k++) HIPCHK(c, hipMemcpyAsync(&t32[k], o32[k] + nrows, 4, hipMemcpyDeviceToHost, c->stream));
                            HIPCHK(c, hipStreamSynchronize(c->stream));
                            for (int k = 0; k < 5; k++) tot[k] = t32[k];
                        }
                    }
                }
            }
            if (strings && nrows > 0) {
                ENSURE(c, c->a_qname, tot[0] + PAD_BYTES); ENSURE(c, c->a_cigar, tot[1] + PAD_BYTES); ENSURE(c, c->a_seq, tot[2] + PAD_BYTES);
                ENSURE(c, c->a_qual, tot[3] + PAD_BYTES); ENSURE(c, c->a_rg, tot[4] + PAD_BYTES);
                so.qname = (uint8_t *)c->a_qname.p; so.cigar = (uint8_t *)c->a_cigar.p; so.seq = (uint8_t *)c->a_seq.p; so.qual = (uint8_t *)c->a_qual.p; so.rg = (uint8_t *)c->a_rg.p;
                KTimer tm(c, DHTS_K_STRINGS);
                hipLaunchKernelGGL(bam_tile_strings, dim3((unsigned)ntiles), dim3(TS_THREADS), 0, c->stream, st, ntiles, to, (const uint32_t *)c->t_rowbase.p, (const uint64_t *)c->d_res.p,
                                   nrows_scan, nrows, (const uint32_t *)c->rec_off.p, row_map_s, bc, so, colmask);
            }
            HIPCHK(c, hipGetLastError());
            if (nrows > 0) {
                out->flag = bc.flag; out->pos = bc.pos; out->mapq = bc.mapq; out->pnext = bc.pnext; out->tlen = bc.tlen; out->tid = bc.tid; out->mtid = bc.mtid;
                out->rg_idx = bc.rg_idx; out->rg_valid = bc.rg_valid;
                out->qname = {o32[0], bc.len_qname, so.qname, tot[0]};
                out->cigar = {o32[1], bc.len_cigar, so.cigar, tot[1]};
                out->seq = {o32[2], so.seq_chars ? so.seq_chars : bc.len_seq, so.seq, tot[2]};
                out->seq_packed = so.seq_chars ? 1 : 0;
                out->qual = {o32[3], so.alen_qual, so.qual, tot[3]};
                out->rg = {o32[4], bc.len_rg, so.rg, tot[4]};
            }
        }
    }
    if (bam_tag_columns(c, st, nrows, out)) return -1;
    if (bam_aux_map(c, st, nrows, out)) return -1;
    if (bam_overlap_join(c, st, bc, nrows, out)) return -1;
    out->n_rows = nrows;
    out->end_uoff = out_base + carry_start;
    {
        uint64_t f = NONE64;
        if (nrows > 0 && !have_first_rec) { HIPCHK(c, hipMemcpyAsync(&first_rec_rel, c->rec_off.p, 4, hipMemcpyDeviceToHost, c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream)); }
        if (nrows > 0) f = out_base + first_rec_rel;
        out->first_rec_uoff = f;
    }
    // ---- advance ----
    return batch_end(c, B, carry_start, rec_err, shard_finished, &out->status);
}

// ---- read_bcf ---------------------------------------------------------------------------------------
// (re)builds everything derived from the header dictionaries: the name arrays of dhts_bcf_info, the validity / slot tables of bcf_rec_check
// and, for VCF text, the sorted name tables the text encoder looks names up in.  Called again when a text scan has added names.
static int bcf_upload_dicts(dhts_ctx *c) {
    const size_t nc = c->bh.ctg.size(), ni = c->bh.ids.size();
    if (c->vcf_text && (nc > c->bcf_ctg_p.capacity() || ni > c->bcf_dict_p.capacity())) return fail(c, "read_bcf: too many names without a header definition");
    c->bcf_ctg_p.clear(); c->bcf_dict_p.clear();
    for (size_t i = 0; i < nc; i++) c->bcf_ctg_p.push_back(c->bh.ctg_present[i] ? c->bh.ctg[i].c_str() : nullptr);
    for (auto &e : c->bh.ids) c->bcf_dict_p.push_back(e.present ? e.key.c_str() : nullptr);
    std::vector<uint8_t> ctg_ok(nc + 1, 0), id_ok(ni + 1, 0); std::vector<int16_t> islot(ni + 1, -1), fslot(ni + 1, -1);
    for (size_t i = 0; i < nc; i++) ctg_ok[i] = c->bh.ctg_present[i] ? 1 : 0;
    for (size_t i = 0; i < ni; i++) id_ok[i] = c->bh.ids[i].present ? 1 : 0;
    for (size_t f = 0; f < c->bsch.info_fields.size(); f++) islot[c->bsch.info_fields[f].id] = (int16_t)f;
    for (size_t f = 0; f < c->bsch.format_fields.size(); f++) if (c->bsch.format_fields[f].id >= 0) fslot[c->bsch.format_fields[f].id] = (int16_t)f;
    ENSURE(c, c->d_ctg_ok, nc + 16); ENSURE(c, c->d_id_ok, ni + 16); ENSURE(c, c->d_info_slot, ni * 2 + 16); ENSURE(c, c->d_fmt_slot, ni * 2 + 16);
    HIPCHK(c, hipMemcpy(c->d_ctg_ok.p, ctg_ok.data(), nc + 1, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_id_ok.p, id_ok.data(), ni + 1, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_info_slot.p, islot.data(), (ni + 1) * 2, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_fmt_slot.p, fslot.data(), (ni + 1) * 2, hipMemcpyHostToDevice));
    if (!c->vcf_text) return 0;
    auto upload = [&](std::vector<std::pair<std::string, int32_t>> &names, const std::vector<uint8_t> *typ_of_id, DevBuf &d_off, DevBuf &d_bytes, DevBuf &d_id, DevBuf *d_typ, DevBuf &d_hash, uint32_t &hmask, const std::vector<uint8_t> *ftyp_of_id = nullptr, DevBuf *d_ftyp = nullptr) -> int {
        std::sort(names.begin(), names.end(), [](const std::pair<std::string, int32_t> &a, const std::pair<std::string, int32_t> &b) {
            const size_t n = a.first.size() < b.first.size() ? a.first.size() : b.first.size();
            const int cmp = memcmp(a.first.data(), b.first.data(), n);
            return cmp != 0 ? cmp < 0 : a.first.size() < b.first.size(); });
        std::vector<uint32_t> off(names.size() + 1, 0); std::string bytes; std::vector<int32_t> ids(names.size() + 1, 0); std::vector<uint8_t> typ(names.size() + 1, 15);
        for (size_t i = 0; i < names.size(); i++) { off[i] = (uint32_t)bytes.size(); bytes += names[i].first; ids[i] = names[i].second; if (typ_of_id) typ[i] = (*typ_of_id)[names[i].second]; }
        off[names.size()] = (uint32_t)bytes.size();
        if (d_off.ensure(off.size() * 4 + 64) || d_bytes.ensure(bytes.size() + 64) || d_id.ensure(ids.size() * 4 + 64) || (d_typ && d_typ->ensure(typ.size() + 64))) return fail(c, "hipMalloc failed");
        HIPCHK(c, hipMemcpy(d_off.p, off.data(), off.size() * 4, hipMemcpyHostToDevice));
        if (!bytes.empty()) HIPCHK(c, hipMemcpy(d_bytes.p, bytes.data(), bytes.size(), hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(d_id.p, ids.data(), ids.size() * 4, hipMemcpyHostToDevice));
        if (d_typ) HIPCHK(c, hipMemcpy(d_typ->p, typ.data(), typ.size(), hipMemcpyHostToDevice));
        {   // the lookup table of vcf_dict_find: at most half full, linear probing
            uint32_t cap = 16; while (cap < names.size() * 2 + 2) cap <<= 1;
            std::vector<uint32_t> tab(cap, 0); hmask = cap - 1;
            for (size_t i = 0; i < names.size(); i++) {
                uint32_t h = vcf_name_hash((const uint8_t *)names[i].first.data(), (uint32_t)names[i].first.size()) & hmask;
                while (tab[h]) h = (h + 1) & hmask;
                tab[h] = (uint32_t)i + 1;
            }
            if (d_hash.ensure((size_t)cap * 4 + 64)) return fail(c, "hipMalloc failed");
            HIPCHK(c, hipMemcpy(d_hash.p, tab.data(), (size_t)cap * 4, hipMemcpyHostToDevice));
        }
        if (d_ftyp) {
            std::vector<uint8_t> ft(names.size() + 1, 15);
            for (size_t i = 0; i < names.size(); i++) ft[i] = (*ftyp_of_id)[names[i].second];
            if (d_ftyp->ensure(ft.size() + 64)) return fail(c, "hipMalloc failed");
            HIPCHK(c, hipMemcpy(d_ftyp->p, ft.data(), ft.size(), hipMemcpyHostToDevice));
        }
        return 0;
    };
    std::vector<std::pair<std::string, int32_t>> cn, in; std::vector<uint8_t> ityp(ni + 1, 15), ftyp(ni + 1, 15);
    for (size_t i = 0; i < nc; i++) if (c->bh.ctg_present[i]) cn.push_back({c->bh.ctg[i], (int32_t)i});
    for (size_t i = 0; i < ni; i++) if (c->bh.ids[i].present) {
        in.push_back({c->bh.ids[i].key, (int32_t)i});
        if (c->bh.ids[i].has[dhts::BCF_HL_INFO]) ityp[i] = (uint8_t)c->bh.ids[i].type[dhts::BCF_HL_INFO];
        if (c->bh.ids[i].has[dhts::BCF_HL_FMT]) ftyp[i] = (uint8_t)c->bh.ids[i].type[dhts::BCF_HL_FMT];
    }
    if (upload(cn, nullptr, c->vd_ctg_off, c->vd_ctg_bytes, c->vd_ctg_id, nullptr, c->vd_ctg_hash, c->vd_ctg_hmask) || upload(in, &ityp, c->vd_id_off, c->vd_id_bytes, c->vd_id_id, &c->vd_id_typ, c->vd_id_hash, c->vd_id_hmask, &ftyp, &c->vd_id_ftyp)) return -1;
    return 0;
}

int dhts_bcf_open(dhts_ctx *c, int tidy_format) {
    if (!c) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    c->bcf_open = false;
    if (c->n_blocks <= 0) return fail(c, "Failed to read BCF/VCF header");
    int64_t k = c->n_blocks < 4 ? c->n_blocks : 4;
    std::vector<uint8_t> h; std::vector<int32_t> bs;
    uint64_t text_end = 0; bool is_text = false;
    for (;;) {
        uint64_t total = c->h_uoff[k];
        h.assign(total + 16, 0); bs.resize(k);
        if (dhts_bgzf_inflate_to_host(c, 0, k, h.data(), total, bs.data()) < 0) return -1;
        uint64_t good = total;
        for (int64_t b = 0; b < k; b++) if (bs[b] != 0) { good = c->h_uoff[b]; break; }
        const bool more = (good == total && k < c->n_blocks);
        if (good < 9) { if (more) { k = (k * 4 < c->n_blocks) ? k * 4 : c->n_blocks; continue; } return fail(c, "Failed to read BCF/VCF header"); }
        if (good >= 16 && memcmp(h.data(), "##fileformat=VCF", 16) == 0) {
            // VCF text (vcf_hdr_read vcf.c:2594-2680): lines up to and including the first one that does not start with "##"; empty lines are
            // skipped, a line that does not start with '#' before that is "No sample line"
            std::string text; uint64_t p = 0; bool found = false, broken = false;
            while (p < good) {
                const uint8_t *nl = (const uint8_t *)memchr(h.data() + p, '\n', good - p);
                if (!nl && (more || good < total)) break;                       // the line may continue in blocks not read yet
                const uint64_t ls = p, e = nl ? (uint64_t)(nl - h.data()) : good;
                uint64_t l = e - ls;
                if (l && h[e - 1] == '\r') l--;
                { uint64_t z = 0; while (z < l && h[ls + z]) z++; l = z; }    // (a C string to the parser)
                p = nl ? e + 1 : good;
                if (l == 0) continue;
                if (h[ls] != '#') { broken = true; break; }
                text.append((const char *)h.data() + ls, l); text.push_back('\n');
                if (l < 2 || h[ls + 1] != '#') { found = true; break; }
            }
            if (broken) return fail(c, "Failed to read BCF/VCF header");
            if (!found) { if (more) { k = (k * 4 < c->n_blocks) ? k * 4 : c->n_blocks; continue; } return fail(c, "Failed to read BCF/VCF header"); }
            std::string perr;
            if (!dhts::bcf_parse_header(text.c_str(), c->bh, &perr)) return fail(c, "Failed to read BCF/VCF header");
            text_end = p; is_text = true;
            break;
        }
        if (memcmp(h.data(), "BCF\2\2", 5) != 0) return fail(c, "Failed to read BCF/VCF header");       // vcf.c:1733-1740
        const uint64_t l_text = hle32(h.data() + 5);
        if (9 + l_text > good) { if (more) { k = (k * 4 < c->n_blocks) ? k * 4 : c->n_blocks; continue; } return fail(c, "Failed to read BCF/VCF header"); }
        text_end = 9 + l_text;
        std::string text((const char *)h.data() + 9, l_text);      // bcf_hdr_parse works on the NUL-terminated text (vcf.c:1752-1753)
        std::string perr;
        if (!dhts::bcf_parse_header(text.c_str(), c->bh, &perr)) return fail(c, "Failed to read BCF/VCF header");
        break;
    }
    dhts::bcf_build_schema(c->bh, tidy_format != 0, c->bsch);
    c->bcf_tidy_req = tidy_format != 0;
    c->first_rec_uoff = text_end; c->scan_first_uoff = text_end;
    c->vcf_text = is_text;
    if (is_text) {
        // names a record uses without a definition are added to the dictionaries while scanning (vcf_parse): room is reserved now so that
        // the name arrays handed out by dhts_bcf_info_get never move
        c->bh.ctg.reserve(c->bh.ctg.size() + 65536); c->bh.ctg_present.reserve(c->bh.ctg.size() + 65536); c->bh.ids.reserve(c->bh.ids.size() + 65536);
        c->bcf_ctg_p.clear(); c->bcf_dict_p.clear();
        c->bcf_ctg_p.reserve(c->bh.ctg.size() + 65536); c->bcf_dict_p.reserve(c->bh.ids.size() + 65536);
    }
    // host-visible dictionaries
    c->bcf_colinfo.clear(); c->bcf_smp_p.clear();
    for (auto &col : c->bsch.cols) {
        dhts_bcf_colinfo ci; ci.name = col.name.c_str(); ci.type = col.duck_type; ci.is_list = col.is_list ? 1 : 0; ci.reserved = 0;
        ci.encoding = col.kind == dhts::BK_CHROM ? DHTS_ENC_CONTIG : col.kind == dhts::BK_FILTER ? DHTS_ENC_DICT : col.kind == dhts::BK_SAMPLE_ID ? DHTS_ENC_SAMPLE :
                      (col.kind == dhts::BK_VEP && col.duck_type == dhts::DT_FLOAT) ? DHTS_ENC_FLOAT_TEXT : DHTS_ENC_PLAIN;
        c->bcf_colinfo.push_back(ci);
    }
    for (auto &sm : c->bh.samples) c->bcf_smp_p.push_back(sm.c_str());
    if (c->bsch.info_fields.size() > 30000 || c->bsch.format_fields.size() > 30000) return fail(c, "read_bcf: too many INFO/FORMAT fields");
    if (bcf_upload_dicts(c)) return -1;
    c->bcf_proj.clear();
    for (size_t i = 0; i < c->bsch.cols.size(); i++) c->bcf_proj.push_back((int32_t)i);
    c->shard_b0 = 0; c->shard_b1 = c->n_blocks; c->shard_rank = 0; c->shard_world = 1;
    c->bcf_open = true;
    return dhts_bcf_rewind(c);
}

extern "C" int dhts_bcf_is_text(const dhts_ctx *c) { return (!c || !c->bcf_open || !c->vcf_text) ? 0 : c->plain_text ? 2 : 1; }
int dhts_bcf_info_get(const dhts_ctx *c, dhts_bcf_info *out) {
    if (!c || !c->bcf_open || !out) return -1;
    out->n_cols = (int32_t)c->bcf_colinfo.size(); out->cols = c->bcf_colinfo.data();
    out->n_contigs = (int32_t)c->bcf_ctg_p.size(); out->contig_name = c->bcf_ctg_p.data();
    out->n_dict = (int32_t)c->bcf_dict_p.size(); out->dict_name = c->bcf_dict_p.data();
    out->n_samples = (int32_t)c->bcf_smp_p.size(); out->sample_name = c->bcf_smp_p.data();
    out->tidy = c->bsch.tidy ? 1 : 0; out->first_rec_uoff = c->first_rec_uoff;
    return 0;
}

int dhts_bcf_set_projection(dhts_ctx *c, const int32_t *col_ids, int32_t n) {
    if (!c || !c->bcf_open) return -1;
    std::vector<int32_t> pr;
    for (int32_t i = 0; i < n; i++) { if (col_ids[i] < 0 || col_ids[i] >= (int32_t)c->bsch.cols.size()) return fail(c, "projection column %d out of range", col_ids[i]); pr.push_back(col_ids[i]); }
    c->bcf_proj = pr;
    return 0;
}

// one region of read_bcf(region := ...): bcf_itr_querys (htslib vcf.h:1391 -> hts_itr_querys hts.c:4179-4200, "." = everything).
// Returns 0, 1 when the region yields no iterator (unknown contig / malformed: the reference skips it, bcf_reader.c:935-953), <0 on error.
int dhts_bcf_set_region(dhts_ctx *c, const char *region) {
    if (!c || !c->bcf_open) return -1;
    c->bcf_rg_active = false; c->bcf_rg_all = false; c->rg_empty_window = false; c->bcf_rg_pending = false;
    c->wins.clear(); c->win_cur = 0; c->scan_end_uoff = ~0ull;
    c->shard_b0 = 0; c->shard_b1 = c->n_blocks; c->shard_rank = 0; c->shard_world = 1; c->scan_first_uoff = c->first_rec_uoff;
    if (!region || !*region) return dhts_bcf_rewind(c);
    std::string tok(region);
    if (tok == ".") { c->bcf_rg_active = true; c->bcf_rg_all = true; return dhts_bcf_rewind(c); }
    if (c->vcf_text) {                                                          // names are the index's (tbx_itr_querys): resolved by dhts_bcf_load_index
        c->bcf_rg_active = true; c->bcf_rg_pending = true; c->bcf_rg_tok = tok; c->bcf_rg_tid = c->bcf_rg_itid = -1;
        return dhts_bcf_rewind(c);
    }
    std::vector<std::string> names;
    for (size_t i = 0; i < c->bh.ctg.size(); i++) names.push_back(c->bh.ctg_present[i] ? c->bh.ctg[i] : std::string("\x01"));
    int tid; int64_t b, e;
    if (!parse_region_token(names, tok, tid, b, e)) return 1;
    c->bcf_rg_active = true; c->bcf_rg_tid = tid; c->bcf_rg_beg = b; c->bcf_rg_end = e;
    return dhts_bcf_rewind(c);
}

int dhts_bcf_set_block_range(dhts_ctx *c, int64_t b0, int64_t b1, int speculative_start) {
    if (!c || !c->bcf_open) return -1;
    if (b0 < 0 || b1 > c->n_blocks || b0 > b1) return fail(c, "bad block range");
    c->shard_b0 = b0; c->shard_b1 = b1; c->shard_rank = speculative_start ? 1 : 0; c->shard_world = 2;
    return dhts_bcf_rewind(c);
}

int dhts_bcf_rewind(dhts_ctx *c) {
    if (!c) return -1;
    discard_prefetch(c);
    if (!c->wins.empty()) enter_window(c, 0);
    c->next_block = c->shard_b0; c->carry_len = 0; c->stream_done = c->rg_empty_window; c->first_batch = true; c->ucur = 0;
    c->huff_b0 = c->huff_nb = 0;
    skip_header_blocks(c);
    if (c->vcf_text && c->shard_rank != 0 && c->shard_b0 > 0 && c->wins.empty()) c->next_block = c->shard_b0 - 1;   // (the byte in front of the shard: see vcf_text_records)
    return 0;
}

static size_t fixed_width(const dhts::BcfColumn &col) {
    if (col.is_list) return 0;
    switch (col.kind) {
    case dhts::BK_CHROM: case dhts::BK_SAMPLE_ID: return 4;
    case dhts::BK_POS: case dhts::BK_QUAL: return 8;
    case dhts::BK_ID: case dhts::BK_REF: return 0;
    default: break;
    }
    return col.duck_type == dhts::DT_BOOLEAN ? 1 : col.duck_type == dhts::DT_VARCHAR ? 0 : 4;
}

// debugging aid (not part of the public header): the BCF2 records the last VCF text batch was turned into
extern "C" int64_t dhts_debug_vcf_records(dhts_ctx *c, uint8_t *dst, uint64_t cap, uint32_t *rec_off, int64_t nrec) {
    if (!c || !c->v_out.p) return -1;
    const uint64_t n = cap < c->v_out.cap ? cap : c->v_out.cap;
    HIPCHK(c, hipMemcpy(dst, c->v_out.p, n, hipMemcpyDeviceToHost));
    if (rec_off && nrec > 0) HIPCHK(c, hipMemcpy(rec_off, c->b_rec_off.p, (size_t)nrec * 4, hipMemcpyDeviceToHost));
    return (int64_t)n;
}

// ---- VCF text batches (vcf_text.hip): the lines of the batch become BCF2 records in v_out; rec_off / dir as for binary input ----------
// out: nrec, carry_start (start of the incomplete last line), rec_err (a line failed: the scan ends before it), rec0_text (text offset of
// the first line), st re-pointed at the records.  Names without a definition are added to the header and the batch is measured again.
// lim: lines that start at or behind this text offset belong to the next shard / lie behind the index window (finished = one was met).
static int vcf_text_records(dhts_ctx *c, const Batch &B, BcfStream &st, int64_t &nrec, uint64_t &carry_start, bool &rec_err, uint32_t &rec0_text, uint32_t &stride, unsigned long long &bad_rec,
                            uint64_t lim, bool &finished) {
    const uint8_t *u = B.u; const uint64_t ulen = B.ulen, out_base = B.out_base;
    uint64_t t0 = 0;
    if (c->first_batch && c->shard_rank == 0) { if (c->scan_first_uoff < out_base) return fail(c, "internal: header beyond first batch"); t0 = c->scan_first_uoff - out_base; }
    else if (c->first_batch) {
        // a block-range shard that starts inside the file owns the lines that START at or behind its first block: lines synchronise on the
        // newline, so the start is the byte behind the first newline at or after (shard start - 1); the batch begins one block early for that byte
        const uint64_t U0 = c->h_uoff[c->shard_b0];
        uint64_t q = U0 > out_base ? U0 - 1 - out_base : 0; bool found = U0 <= out_base;
        std::vector<uint8_t> piece(1u << 16);
        while (!found && q < ulen) {
            const uint64_t k = ulen - q < piece.size() ? ulen - q : piece.size();
            HIPCHK(c, hipMemcpy(piece.data(), u + q, k, hipMemcpyDeviceToHost));
            const void *nl = memchr(piece.data(), '\n', k);
            if (nl) { q += (uint64_t)((const uint8_t *)nl - piece.data()) + 1; found = true; } else q += k;
        }
        t0 = found ? q : ulen;
        if (c->first_rec_uoff > out_base + t0) t0 = c->first_rec_uoff - out_base;     // (a shard that starts inside the header)
    }
    nrec = 0; carry_start = ulen; rec_err = false; rec0_text = (uint32_t)t0; bad_rec = ~0ull;
    if (c->first_batch) {
        // a scan range that ends before its first line starts (a shard inside the header, a shard swallowed by one long line) owns no line
        uint64_t end_abs = c->shard_b1 < c->n_blocks ? c->h_uoff[c->shard_b1] : ~0ull;
        if (c->scan_end_uoff < end_abs) end_abs = c->scan_end_uoff;
        if (out_base + t0 >= end_abs) { finished = true; carry_start = t0 < ulen ? t0 : ulen; return 0; }
    }
    if (t0 >= ulen) { carry_start = ulen; return 0; }
    const int64_t nchunks = (int64_t)((ulen - (t0 & ~(uint64_t)15) + VCF_CHUNK - 1) / VCF_CHUNK);
    ENSURE(c, c->v_cnt, (size_t)nchunks * 4 + 64); ENSURE(c, c->v_base, (size_t)(nchunks + 1) * 4 + 64);
    hipLaunchKernelGGL(vcf_line_count, dim3((unsigned)nchunks), dim3(256), 0, c->stream, u, t0, ulen, (uint32_t *)c->v_cnt.p, nchunks);
    const uint32_t *kin[1] = {(const uint32_t *)c->v_cnt.p}; uint32_t *kout[1] = {(uint32_t *)c->v_base.p}; uint64_t nl = 0;
    if (run_scan(c, 1, kin, kout, nullptr, nchunks, &nl)) return -1;
    if (nl + 2 >= (1ull << 32)) return fail(c, "batch too large");
    ENSURE(c, c->v_line_off, (size_t)(nl + 2) * 4 + 64);
    hipLaunchKernelGGL(vcf_line_fill, dim3((unsigned)nchunks), dim3(256), 0, c->stream, u, t0, ulen, (const uint32_t *)c->v_base.p, (uint32_t *)c->v_line_off.p, nchunks);
    uint32_t last_start = 0;
    HIPCHK(c, hipMemcpyAsync(&last_start, (const uint32_t *)c->v_line_off.p + nl, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    int64_t nlines = (int64_t)nl; int last_open = 0;
    carry_start = last_start;
    if (B.final_batch && last_start < ulen) { nlines++; last_open = 1; carry_start = ulen; }     // the last line of the file need not end in a newline
    if (lim < ulen) {
        std::vector<uint32_t> lo_(nlines + 1);
        HIPCHK(c, hipMemcpy(lo_.data(), c->v_line_off.p, (size_t)(nl + 1) * 4, hipMemcpyDeviceToHost));
        int64_t lo = 0, hi = nlines;                                                            // first line that starts at or behind lim
        while (lo < hi) { const int64_t mid = (lo + hi) / 2; if (lo_[mid] < lim) lo = mid + 1; else hi = mid; }
        if (lo < nlines) { finished = true; carry_start = lo_[lo]; nlines = lo; last_open = 0; }
        else if (carry_start >= lim) finished = true;
    }
    if (nlines == 0) return 0;
    ENSURE(c, c->v_rec_len, (size_t)(nlines + 1) * 4 + 64); ENSURE(c, c->b_rec_off, (size_t)(nlines + 1) * 4 + 64); ENSURE(c, c->v_ctr, 64);
    if (const char *e = getenv("DHTS_VCF_CAPS")) { const uint32_t v = (uint32_t)atoi(e); if (v > 0 && c->v_undef_cap == 65536 && c->v_patch_cap == (1u << 20)) c->v_undef_cap = c->v_patch_cap = v; }   // (tests: small caps make the growth path cheap to reach)
    ENSURE(c, c->v_undef, (size_t)c->v_undef_cap * sizeof(VcfUndef)); ENSURE(c, c->v_patch, (size_t)c->v_patch_cap * sizeof(VcfPatch));
    // the tokens of `n` recorded entries (VcfUndef / VcfPatch: pos and len in words pos_w / len_w), fetched with one gather and one copy
    std::vector<uint32_t> tk_off; std::vector<char> tk_bytes;
    auto fetch_tokens = [&](const void *ent_dev, const uint32_t *ent_host, int pos_w, int len_w, uint32_t n) -> int {
        tk_off.assign((size_t)n + 1, 0);
        for (uint32_t i = 0; i < n; i++) tk_off[(size_t)i + 1] = tk_off[i] + ent_host[4u * i + (uint32_t)len_w];
        tk_bytes.assign((size_t)tk_off[n] + 1, 0);
        if (tk_off[n] == 0) return 0;
        ENSURE(c, c->v_tok_off, (size_t)n * 4 + 64); ENSURE(c, c->v_tok_bytes, (size_t)tk_off[n] + 64);
        HIPCHK(c, hipMemcpyAsync(c->v_tok_off.p, tk_off.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(vcf_gather_tokens, dim3((n + 255) / 256), dim3(256), 0, c->stream, u, (const uint32_t *)ent_dev, pos_w, len_w, (const uint32_t *)c->v_tok_off.p, n, (uint8_t *)c->v_tok_bytes.p);
        HIPCHK(c, hipMemcpyAsync(tk_bytes.data(), c->v_tok_bytes.p, tk_off[n], hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return 0;
    };
    VcfArgs a; memset(&a, 0, sizeof(a));
    a.u = u; a.line_off = (const uint32_t *)c->v_line_off.p; a.nlines = nlines; a.text_end = ulen; a.last_open = last_open;
    a.n_smp = (int32_t)c->bh.samples.size(); a.v44 = c->bh.version >= 4004000 ? 1 : 0;
    { static const bool no_stage = getenv("DHTS_VCF_STAGE") && atoi(getenv("DHTS_VCF_STAGE")) == 0; a.lds_budget = no_stage ? 0u : VCF_LDS_BYTES; if (getenv("DHTS_VCF_STAGE") && atoi(getenv("DHTS_VCF_STAGE")) == 2) a.lds_budget = VCF_LDS_BYTES - 15u; }
    a.rec_len = (uint32_t *)c->v_rec_len.p; a.rec_off = (const uint32_t *)c->b_rec_off.p; a.first_bad = (unsigned long long *)((uint64_t *)c->v_ctr.p + 2);
    a.counters = (uint32_t *)c->v_ctr.p; a.undef = (VcfUndef *)c->v_undef.p; a.undef_cap = c->v_undef_cap; a.patch = (VcfPatch *)c->v_patch.p; a.patch_cap = c->v_patch_cap;
    // Lines so long that 64 of them overflow a workgroup's LDS staging get a wave each (vcf_encode_wave): INFO fields parsed one per lane.
    // DHTS_VCF_WAVE=0 / 1 forces the choice (tests run both ways).
    bool wave_lines = nlines > 0 && ((ulen - t0) / (uint64_t)nlines) * VCF_ENC_THREADS > VCF_LDS_BYTES;
    if (const char *e = getenv("DHTS_VCF_WAVE")) wave_lines = atoi(e) != 0;
    unsigned long long first_bad = ~0ull;
    for (int round = 0;; round++) {
        if (round > 1000) return fail(c, "read_bcf: too many names without a header definition");
        a.ctg = {(const uint32_t *)c->vd_ctg_off.p, (const uint8_t *)c->vd_ctg_bytes.p, (const int32_t *)c->vd_ctg_id.p, nullptr, nullptr, 0, (const uint32_t *)c->vd_ctg_hash.p, c->vd_ctg_hmask};
        a.ids = {(const uint32_t *)c->vd_id_off.p, (const uint8_t *)c->vd_id_bytes.p, (const int32_t *)c->vd_id_id.p, (const uint8_t *)c->vd_id_typ.p, (const uint8_t *)c->vd_id_ftyp.p, 0, (const uint32_t *)c->vd_id_hash.p, c->vd_id_hmask};
        { int32_t n1 = 0, n2 = 0; for (size_t i = 0; i < c->bh.ctg.size(); i++) n1 += c->bh.ctg_present[i] ? 1 : 0; for (auto &e : c->bh.ids) n2 += e.present ? 1 : 0; a.ctg.n = n1; a.ids.n = n2; }
        HIPCHK(c, hipMemsetAsync(c->v_ctr.p, 0, 16, c->stream)); HIPCHK(c, hipMemsetAsync((uint64_t *)c->v_ctr.p + 2, 0xff, 8, c->stream));
        if (wave_lines) hipLaunchKernelGGL(vcf_encode_wave<false>, dim3((unsigned)nlines), dim3(64), 0, c->stream, a);
        else hipLaunchKernelGGL(vcf_encode<false>, dim3((unsigned)((nlines + VCF_ENC_THREADS - 1) / VCF_ENC_THREADS)), dim3(VCF_ENC_THREADS), VCF_LDS_BYTES, c->stream, a);
        uint64_t ctr[3] = {0, 0, 0};
        HIPCHK(c, hipMemcpyAsync(ctr, c->v_ctr.p, 24, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        first_bad = ctr[2];
        const uint32_t n_undef = (uint32_t)(ctr[0] & 0xffffffffu);
        if (n_undef == 0) break;
        // names used without a definition: htslib adds dummy definitions as it meets them, so ids follow the order of first appearance;
        // FORMAT Floats in strtod's forms are checked here (the number has to end where the token ends)
        if (n_undef > c->v_undef_cap) {                                          // more than there was room to record: make room, measure again
            c->v_undef_cap = n_undef + n_undef / 4 + 1024;
            ENSURE(c, c->v_undef, (size_t)c->v_undef_cap * sizeof(VcfUndef));
            a.undef = (VcfUndef *)c->v_undef.p; a.undef_cap = c->v_undef_cap;
            continue;
        }
        const uint32_t got = n_undef;
        std::vector<VcfUndef> ud(got);
        HIPCHK(c, hipMemcpy(ud.data(), c->v_undef.p, (size_t)got * sizeof(VcfUndef), hipMemcpyDeviceToHost));
        if (fetch_tokens(c->v_undef.p, (const uint32_t *)ud.data(), 1, 2, got)) return -1;
        std::vector<uint32_t> order(got); for (uint32_t i = 0; i < got; i++) order[i] = i;
        std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return ud[x].line != ud[y].line ? ud[x].line < ud[y].line : ud[x].pos < ud[y].pos; });
        bool added = false; unsigned long long cut = first_bad;
        for (const uint32_t oi : order) {
            const VcfUndef &x = ud[oi];
            if ((unsigned long long)x.line >= cut) break;                        // lines behind the first bad one are never parsed
            std::string name(tk_bytes.data() + tk_off[oi], x.len);
            if (x.cls == 4) {
                char *end = nullptr; (void)strtod(name.c_str(), &end);
                if (strlen(name.c_str()) != name.size() || end != name.c_str() + name.size()) { cut = x.line; break; }      // "Invalid character": the record is an error
                continue;
            }
            const int hl = x.cls == 3 ? dhts::BCF_HL_FMT : dhts::BCF_HL_INFO;
            auto defined = [&]() {
                if (x.cls == 0) { for (size_t i = 0; i < c->bh.ctg.size(); i++) if (c->bh.ctg_present[i] && c->bh.ctg[i] == name) return true; return false; }
                const int id = c->bh.find_id(name); return id >= 0 && (x.cls == 1 || c->bh.ids[id].has[hl]);
            };
            if (defined()) continue;
            const std::string line = x.cls == 0 ? "##contig=<ID=" + name + ">" : x.cls == 1 ? "##FILTER=<ID=" + name + ",Description=\"Dummy\">"
                                   : x.cls == 2 ? "##INFO=<ID=" + name + ",Number=1,Type=String,Description=\"Dummy\">" : "##FORMAT=<ID=" + name + ",Number=1,Type=String,Description=\"Dummy\">";
            if (name.find('\n') != std::string::npos || !dhts::bcf_header_add_line(c->bh, line.c_str()) || !defined()) { cut = x.line; break; }   // "Could not add dummy header": the record is an error
            added = true;
        }
        if (added && bcf_upload_dicts(c)) return -1;
        const bool shortened = cut < (unsigned long long)nlines;
        if (shortened) {
            first_bad = cut; a.nlines = nlines = (int64_t)cut; rec_err = true; a.last_open = 0;
            if (nlines == 0) { nrec = 0; return 0; }
        }
        if (!added && !shortened) break;                                         // only float checks are left, and they passed
        st.n_ctg = (int32_t)c->bh.ctg.size(); st.n_ids = (int32_t)c->bh.ids.size();
        st.ctg_ok = (const uint8_t *)c->d_ctg_ok.p; st.id_ok = (const uint8_t *)c->d_id_ok.p; st.info_slot = (const int16_t *)c->d_info_slot.p; st.fmt_slot = (const int16_t *)c->d_fmt_slot.p;
    }
    nrec = nlines;
    if (first_bad < (unsigned long long)nlines) { nrec = (int64_t)first_bad; rec_err = true; }
    if (nrec == 0) return 0;
    const uint32_t *lin[1] = {(const uint32_t *)c->v_rec_len.p}; uint32_t *lout[1] = {(uint32_t *)c->b_rec_off.p}; uint64_t total = 0;
    if (run_scan(c, 1, lin, lout, nullptr, nrec, &total)) return -1;
    if (total + PAD_BYTES >= (1ull << 32)) return fail(c, "batch too large");
    ENSURE(c, c->v_out, total + PAD_BYTES + 64);
    a.nlines = nrec; a.out = (uint8_t *)c->v_out.p;
    if (nrec < nlines) { a.last_open = 0; }
    HIPCHK(c, hipMemsetAsync(c->v_ctr.p, 0, 16, c->stream));
    if (wave_lines) hipLaunchKernelGGL(vcf_encode_wave<true>, dim3((unsigned)nrec), dim3(64), 0, c->stream, a);
    else hipLaunchKernelGGL(vcf_encode<true>, dim3((unsigned)((nrec + VCF_ENC_THREADS - 1) / VCF_ENC_THREADS)), dim3(VCF_ENC_THREADS), VCF_LDS_BYTES, c->stream, a);
    HIPCHK(c, hipMemsetAsync((uint8_t *)c->v_out.p + total, 0, PAD_BYTES, c->stream));
    uint32_t ctr2[2] = {0, 0};
    HIPCHK(c, hipMemcpyAsync(ctr2, c->v_ctr.p, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (ctr2[1] > c->v_patch_cap) {                          // more numbers for the host than there was room to record: make room, write again
        c->v_patch_cap = ctr2[1] + ctr2[1] / 4 + 1024;
        ENSURE(c, c->v_patch, (size_t)c->v_patch_cap * sizeof(VcfPatch));
        a.patch = (VcfPatch *)c->v_patch.p; a.patch_cap = c->v_patch_cap;
        HIPCHK(c, hipMemsetAsync(c->v_ctr.p, 0, 16, c->stream));
        if (wave_lines) hipLaunchKernelGGL(vcf_encode_wave<true>, dim3((unsigned)nrec), dim3(64), 0, c->stream, a);
        else hipLaunchKernelGGL(vcf_encode<true>, dim3((unsigned)((nrec + VCF_ENC_THREADS - 1) / VCF_ENC_THREADS)), dim3(VCF_ENC_THREADS), VCF_LDS_BYTES, c->stream, a);
        HIPCHK(c, hipMemcpyAsync(ctr2, c->v_ctr.p, 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (ctr2[1] > c->v_patch_cap) return fail(c, "read_bcf: the write pass recorded %u numbers for the host, room for %u", ctr2[1], c->v_patch_cap);
    }
    if (ctr2[1]) {
        // numbers hts_str2dbl hands to strtod (exponents, > 14 digits, inf / nan / hex) and QUAL values outside that form: converted here.
        // One gather + one copy brings the tokens, one copy + one scatter takes the words back.
        const uint32_t np = ctr2[1];
        std::vector<VcfPatch> pt(np); std::vector<uint32_t> words(np);
        HIPCHK(c, hipMemcpy(pt.data(), c->v_patch.p, (size_t)np * sizeof(VcfPatch), hipMemcpyDeviceToHost));
        if (fetch_tokens(c->v_patch.p, (const uint32_t *)pt.data(), 0, 1, np)) return -1;
        std::string tok;
        for (uint32_t i = 0; i < np; i++) {
            const VcfPatch &x = pt[i];
            tok.assign(tk_bytes.data() + tk_off[i], x.len);
            uint32_t bits;
            if (x.kind == 0) { const float f = (float)atof(tok.c_str()); memcpy(&bits, &f, 4); }
            else {
                char *end = nullptr; const double d = strtod(tok.c_str(), &end);
                if (end == tok.c_str() && x.kind == 1) bits = 0x7F800001u;                     // INFO: a failed conversion is a missing value
                else { const float f = (end == tok.c_str()) ? 0.0f : (float)d; memcpy(&bits, &f, 4); }   // (FORMAT stores what strtod returned: 0.0)
            }
            words[i] = bits;
        }
        ENSURE(c, c->v_tok_bits, (size_t)np * 4 + 64);
        HIPCHK(c, hipMemcpyAsync(c->v_tok_bits.p, words.data(), (size_t)np * 4, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(vcf_scatter_words, dim3((np + 255) / 256), dim3(256), 0, c->stream, (uint8_t *)c->v_out.p, (const VcfPatch *)c->v_patch.p, (const uint32_t *)c->v_tok_bits.p, np);
        HIPCHK(c, hipStreamSynchronize(c->stream));          // (`words` is pageable host memory: the copy must have left it before it goes out of scope)
    }
    HIPCHK(c, hipMemcpyAsync(&rec0_text, c->v_line_off.p, 4, hipMemcpyDeviceToHost, c->stream));
    st.u = (const uint8_t *)c->v_out.p; st.ulen = total;
    st.n_ctg = (int32_t)c->bh.ctg.size(); st.n_ids = (int32_t)c->bh.ids.size();
    st.ctg_ok = (const uint8_t *)c->d_ctg_ok.p; st.id_ok = (const uint8_t *)c->d_id_ok.p; st.info_slot = (const int16_t *)c->d_info_slot.p; st.fmt_slot = (const int16_t *)c->d_fmt_slot.p;
    const size_t n = (size_t)nrec; const int D = 2 + st.n_info_f + st.n_fmt_f;
    stride = (uint32_t)((n + 63) & ~(size_t)63);
    ENSURE(c, c->b_dir, (size_t)D * stride * 4 + 16); ENSURE(c, c->d_res, 64);
    HIPCHK(c, hipMemsetAsync((uint64_t *)c->d_res.p + 4, 0xff, 8, c->stream));
    {
        KTimer tm(c, DHTS_K_BCF_CHECK);
        hipLaunchKernelGGL(bcf_rec_check, dim3((unsigned)((nrec + 255) / 256)), dim3(256), 0, c->stream, st, (const uint32_t *)c->b_rec_off.p, nrec, (uint32_t *)c->b_dir.p, stride,
                           (unsigned long long *)((uint64_t *)c->d_res.p + 4));
    }
    HIPCHK(c, hipMemcpyAsync(&bad_rec, (uint64_t *)c->d_res.p + 4, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

static int bcf_next_batch_one(dhts_ctx *c, int64_t max_blocks, dhts_bcf_batch *out);
int dhts_bcf_next_batch(dhts_ctx *c, int64_t max_blocks, dhts_bcf_batch *out) {
    if (!c || !out) return -1;
    for (;;) {
        if (bcf_next_batch_one(c, max_blocks, out)) return -1;
        // a region query with several index windows: the end of one window is the start of the next, not the end of the scan
        if (out->status == 1 && c->win_cur + 1 < c->wins.size()) {
            enter_window(c, c->win_cur + 1);
            discard_prefetch(c);
            c->next_block = c->shard_b0; c->carry_len = 0; c->stream_done = false; c->first_batch = true; c->ucur = 0;
            out->status = 0;
            if (out->n_rows == 0) continue;
        }
        return 0;
    }
}
static int bcf_next_batch_one(dhts_ctx *c, int64_t max_blocks, dhts_bcf_batch *out) {
    if (!c || !out) return -1;
    memset(out, 0, sizeof(*out));
    for (auto &a : c->bcf_ar) { a.p = nullptr; a.n = 0; }
    if (!c->bcf_open) return fail(c, "dhts_bcf_open not called");
    HIPCHK(c, hipSetDevice(c->device));
    const int ncols = (int)c->bcf_proj.size();
    c->bcf_out.assign(ncols, dhts_bcf_col());
    for (int i = 0; i < ncols; i++) { memset(&c->bcf_out[i], 0, sizeof(dhts_bcf_col)); c->bcf_out[i].col = c->bcf_proj[i]; }
    out->n_cols = ncols; out->cols = c->bcf_out.data();
    if (c->stream_done) { out->status = 1; return 0; }
    Batch B;
    if (batch_begin(c, max_blocks, B)) return -1;
    uint8_t *u = B.u; uint64_t ulen = B.ulen; const uint64_t out_base = B.out_base;
    BcfStream st; memset(&st, 0, sizeof(st));
    st.u = u; st.ulen = ulen; st.n_ctg = (int32_t)c->bh.ctg.size(); st.n_ids = (int32_t)c->bh.ids.size(); st.n_smp = c->bsch.n_samples;
    st.final_batch = B.final_batch ? 1 : 0; st.n_info_f = (int32_t)c->bsch.info_fields.size(); st.n_fmt_f = (int32_t)c->bsch.format_fields.size();
    st.ctg_ok = (const uint8_t *)c->d_ctg_ok.p; st.id_ok = (const uint8_t *)c->d_id_ok.p;
    st.info_slot = (const int16_t *)c->d_info_slot.p; st.fmt_slot = (const int16_t *)c->d_fmt_slot.p;

    // ---- tiles: record chain ----
    int64_t ntiles = (int64_t)((ulen + TILE_BYTES - 1) / TILE_BYTES); if (ntiles < 1) ntiles = 1;
    ENSURE(c, c->t_first, ntiles * 8); ENSURE(c, c->t_end, ntiles * 8); ENSURE(c, c->t_count, ntiles * 4); ENSURE(c, c->t_err, ntiles * 4);
    ENSURE(c, c->t_rowbase, ntiles * 4 + 16); ENSURE(c, c->d_res, 64); ENSURE(c, c->d_nfixed, 64);
    ENSURE(c, c->t2_first, ntiles * 8); ENSURE(c, c->t2_end, ntiles * 8); ENSURE(c, c->t2_count, ntiles * 4); ENSURE(c, c->t2_err, ntiles * 4);
    TileOut to; to.first = (uint64_t *)c->t_first.p; to.end_next = (uint64_t *)c->t_end.p; to.count = (uint32_t *)c->t_count.p; to.err = (int32_t *)c->t_err.p;
    TileOut to2; to2.first = (uint64_t *)c->t2_first.p; to2.end_next = (uint64_t *)c->t2_end.p; to2.count = (uint32_t *)c->t2_count.p; to2.err = (int32_t *)c->t2_err.p;
    uint64_t start0;
    if (c->first_batch) start0 = (c->shard_rank == 0) ? c->scan_first_uoff - out_base : NONE64;
    else start0 = 0;
    if (c->first_batch && c->shard_rank == 0 && c->scan_first_uoff < out_base) return fail(c, "internal: header beyond first batch");
    uint64_t res[4] = {0, 0, 0, 0};
    // (same false-start retry as in dhts_bam_next_batch: a speculated shard start whose chain breaks inside the batch, or whose
    //  "records" fail bcf_record_check, is replaced by the next candidate; if every retry fails too, the first attempt stands)
    const bool speculative = (start0 == NONE64);
    uint64_t first0 = NONE64, spec_from = 0; int spec_tries = 0; bool restoring = false;
    int64_t nrec = 0; uint64_t carry_start = 0; bool rec_err = false;
    // the scan range ends with its last block (later shards / the rest of the file exist) or, for one of several index windows, exactly
    // at the window's end: windows are disjoint, so no record is delivered twice
    uint64_t shard_end_u = B.sharded_tail ? c->h_uoff[c->shard_b1] : ~0ull;
    if (c->scan_end_uoff < shard_end_u) shard_end_u = c->scan_end_uoff;
    const bool cut_tail = B.sharded_tail || c->scan_end_uoff != ~0ull;
    bool shard_finished = false;
    const int reps = c->bsch.tidy ? c->bsch.n_samples : 1;
    const int D = 2 + st.n_info_f + st.n_fmt_f;
    uint32_t rec0_off = 0;
    unsigned long long bad = ~0ull;
    uint32_t stride = 64;
    if (c->vcf_text) {
        if (c->bcf_rg_pending) return fail(c, "read_bcf: a region query on VCF text needs the tabix index (dhts_bcf_load_index) before the scan");
        uint32_t rec0_text = 0;
        const uint64_t lim = (cut_tail && out_base + ulen > shard_end_u) ? shard_end_u - out_base : ~0ull;
        if (vcf_text_records(c, B, st, nrec, carry_start, rec_err, rec0_text, stride, bad, lim, shard_finished)) return -1;
        rec0_off = rec0_text;
    } else
    for (;;) {
        to.first = (uint64_t *)c->t_first.p; to.end_next = (uint64_t *)c->t_end.p; to.count = (uint32_t *)c->t_count.p; to.err = (int32_t *)c->t_err.p;
        to2.first = (uint64_t *)c->t2_first.p; to2.end_next = (uint64_t *)c->t2_end.p; to2.count = (uint32_t *)c->t2_count.p; to2.err = (int32_t *)c->t2_err.p;
        {
            KTimer tm(c, DHTS_K_TILES);
            hipLaunchKernelGGL(bcf_tile_scan, dim3((unsigned)ntiles), dim3(64), 0, c->stream, st, start0, ntiles, to, spec_from);
            int rounds = 0;
            for (;;) {
                (void)hipMemsetAsync(c->d_nfixed.p, 0, 4, c->stream);
                hipLaunchKernelGGL(bcf_tile_fix, dim3((unsigned)((ntiles + 255) / 256)), dim3(256), 0, c->stream, st, TILE_BYTES, ntiles, to, to2, (uint32_t *)c->d_nfixed.p);
                { TileOut tmp = to; to = to2; to2 = tmp; }
                uint32_t nfixed = 0;
                HIPCHK(c, hipMemcpyAsync(&nfixed, c->d_nfixed.p, 4, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(c, hipStreamSynchronize(c->stream));
                if (g_debug) fprintf(stderr, "[dhts] bcf tiles=%lld round=%d nfixed=%u\n", (long long)ntiles, rounds, nfixed);
                if (nfixed == 0) break;
                if (++rounds > 256) { hipLaunchKernelGGL(bcf_tile_fix_seq, dim3(1), dim3(1), 0, c->stream, st, TILE_BYTES, ntiles, to); break; }
            }
            hipLaunchKernelGGL(bam_tile_finalize, dim3(1), dim3(1024), 0, c->stream, ntiles, to, (uint32_t *)c->t_rowbase.p, (uint64_t *)c->d_res.p);
        }
        HIPCHK(c, hipMemcpyAsync(res, c->d_res.p, 32, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(&first0, to.first, 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        nrec = (int64_t)res[0]; carry_start = res[1]; rec_err = res[2] != 0;
        if (carry_start == NONE64) carry_start = ulen;
        bad = ~0ull; rec0_off = 0;
        if (nrec > 0) {
            const size_t n = (size_t)nrec;
            stride = (uint32_t)((n + 63) & ~(size_t)63);
            ENSURE(c, c->b_rec_off, n * 4 + 16); ENSURE(c, c->b_dir, (size_t)D * stride * 4 + 16);
            hipLaunchKernelGGL(bcf_tile_offsets, dim3((unsigned)((ntiles + 255) / 256)), dim3(256), 0, c->stream, st, TILE_BYTES, ntiles, to, (const uint32_t *)c->t_rowbase.p,
                               (const uint64_t *)c->d_res.p, nrec, (uint32_t *)c->b_rec_off.p);
            HIPCHK(c, hipMemsetAsync((uint64_t *)c->d_res.p + 4, 0xff, 8, c->stream));
            {
                KTimer tm(c, DHTS_K_BCF_CHECK);
                hipLaunchKernelGGL(bcf_rec_check, dim3((unsigned)((nrec + 255) / 256)), dim3(256), 0, c->stream, st, (const uint32_t *)c->b_rec_off.p, nrec, (uint32_t *)c->b_dir.p, stride,
                                   (unsigned long long *)((uint64_t *)c->d_res.p + 4));
            }
            HIPCHK(c, hipMemcpyAsync(&bad, (uint64_t *)c->d_res.p + 4, 8, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipMemcpyAsync(&rec0_off, c->b_rec_off.p, 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
        if (!speculative || restoring) break;
        const bool false_start = first0 != NONE64 && (res[2] != 0 || bad < (unsigned long long)nrec), exhausted = spec_tries > 0 && first0 == NONE64;
        if (!false_start && !exhausted) break;
        if (exhausted || spec_tries == 16) { spec_from = 0; restoring = true; continue; }
        spec_from = first0 + 1; spec_tries++;
    }
    if (nrec > 0) {
        if (bad < (unsigned long long)nrec) { nrec = (int64_t)bad; rec_err = true; }    // the first bad record ends the scan (bcf_reader.c:1319-1349)
        if (nrec > 0 && !c->vcf_text && cut_tail && out_base + ulen > shard_end_u) {
            std::vector<uint32_t> ro(nrec);
            HIPCHK(c, hipMemcpyAsync(ro.data(), c->b_rec_off.p, nrec * 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            const uint64_t lim = shard_end_u - out_base;
            int64_t lo = 0, hi = nrec;
            while (lo < hi) { int64_t mid = (lo + hi) / 2; if (ro[mid] < lim) lo = mid + 1; else hi = mid; }
            if (lo < nrec) { carry_start = ro[lo]; nrec = lo; shard_finished = true; }
            else if (carry_start >= lim) shard_finished = true;
        }
        const uint32_t *sel = nullptr;
        if (nrec > 0 && c->bcf_rg_active && !c->bcf_rg_all) {
            // region predicate per record -> compacted list of kept record ids (validation above covered every record read)
            const size_t nn = (size_t)nrec;
            ENSURE(c, c->b_keep, nn * 4 + 16); ENSURE(c, c->b_map, (nn + 1) * 4 + 16); ENSURE(c, c->b_sel, nn * 4 + 16);
            hipLaunchKernelGGL(bcf_region_keep, dim3((unsigned)((nrec + 255) / 256)), dim3(256), 0, c->stream, st, (const uint32_t *)c->b_rec_off.p, (const uint32_t *)c->b_dir.p, nrec,
                               c->bcf_rg_tid, c->bcf_rg_beg, c->bcf_rg_end, 0, (uint32_t *)c->b_keep.p);
            const uint32_t *kin[1] = {(const uint32_t *)c->b_keep.p}; uint32_t *kout[1] = {(uint32_t *)c->b_map.p}; uint64_t kept = 0;
            if (run_scan(c, 1, kin, kout, nullptr, nrec, &kept)) return -1;
            hipLaunchKernelGGL(bcf_select, dim3((unsigned)((nrec + 255) / 256)), dim3(256), 0, c->stream, (const uint32_t *)c->b_map.p, nrec, (uint32_t *)c->b_sel.p);
            sel = (const uint32_t *)c->b_sel.p;
            nrec = (int64_t)kept;
        }
        if (nrec > 0 && ncols > 0) {
            const int64_t nrows = nrec * reps;
            const uint32_t ostride = (uint32_t)(((size_t)nrows + 1 + 63) & ~(size_t)63);
            // column program for this projection
            std::vector<BcfColDev> cd(ncols);
            int nsa = 0; size_t fixed_bytes = 0;
            std::vector<size_t> fixed_at(ncols, 0);
            for (int i = 0; i < ncols; i++) {
                const dhts::BcfColumn &col = c->bsch.cols[c->bcf_proj[i]];
                BcfColDev &d = cd[i]; memset(&d, 0, sizeof(d));
                d.kind = col.kind; d.is_list = col.is_list ? 1 : 0; d.sample = col.sample; d.slot = col.field < 0 ? 0 : col.field; d.flags = 0; d.sa_cnt = d.sa_bytes = -1;
                if (col.kind == dhts::BK_INFO) d.htype = c->bsch.info_fields[col.field].htype;
                if (col.kind == dhts::BK_VEP) {                                                      // one field of every transcript of the annotation tag
                    d.slot = c->bsch.vep_info_field; d.vep_field = col.field; d.htype = c->bsch.vep_fields[col.field].htype;
                    if (c->bsch.info_fields[c->bsch.vep_info_field].htype != dhts::BCF_HT_STR) d.flags |= BF_NULL_ALWAYS;   // bcf_get_info_string's type check (vcf.c:6056-6066)
                }
                if (col.kind == dhts::BK_FORMAT) {
                    const dhts::BcfField &f = c->bsch.format_fields[col.field];
                    d.htype = f.htype;
                    if (f.htype == dhts::BCF_HT_FLAG || f.id < 0) d.flags |= BF_NULL_ALWAYS;          // no getter yields a value for a FORMAT Flag; default GT column (bcf_reader.c:683-692)
                    if (f.htype == dhts::BCF_HT_STR && f.name == "GT") { d.flags |= BF_GT; if (!c->bsch.gt_string_ok) d.flags |= BF_NULL_ALWAYS; if (c->bh.version < 4004000 && f.id == c->bsch.gt_id) d.flags |= BF_GT_FIX; }
                    else if (f.htype == dhts::BCF_HT_STR && f.is_list) d.flags |= BF_NULL_ALWAYS;       // LIST(VARCHAR) FORMAT strings: undefined in the reference, NULL here
                    if (f.name == "GT" && f.htype != dhts::BCF_HT_STR) d.flags |= BF_NULL_ALWAYS;      // getter type check vcf.c:6183-6187
                }
                const bool varchar = (col.duck_type == dhts::DT_VARCHAR && col.kind != dhts::BK_CHROM && col.kind != dhts::BK_SAMPLE_ID && col.kind != dhts::BK_FILTER) ||
                                     (col.kind == dhts::BK_VEP && col.duck_type == dhts::DT_FLOAT);      // Float fields of a transcript travel as text (DHTS_ENC_FLOAT_TEXT)
                if (col.is_list) { d.sa_cnt = nsa++; if (varchar) d.sa_bytes = nsa++; }
                else if (varchar) d.sa_bytes = nsa++;
                const size_t w = fixed_width(col);
                fixed_at[i] = fixed_bytes; fixed_bytes += (w * (size_t)nrows + 63) & ~(size_t)63;
            }
            ENSURE(c, c->b_valid, (size_t)ncols * nrows + 64); ENSURE(c, c->b_fixed, fixed_bytes + 64);
            ENSURE(c, c->b_lens, (size_t)(nsa ? nsa : 1) * ostride * 4 + 64); ENSURE(c, c->b_offs, (size_t)(nsa ? nsa : 1) * ostride * 4 + 64);
            ENSURE(c, c->b_coldev, sizeof(BcfColDev) * ncols);
            for (int i = 0; i < ncols; i++) {
                cd[i].valid = (uint8_t *)c->b_valid.p + (size_t)i * nrows;
                cd[i].fixed = fixed_width(c->bsch.cols[c->bcf_proj[i]]) ? (void *)((uint8_t *)c->b_fixed.p + fixed_at[i]) : nullptr;
            }
            HIPCHK(c, hipMemcpyAsync(c->b_coldev.p, cd.data(), sizeof(BcfColDev) * ncols, hipMemcpyHostToDevice, c->stream));
            BcfCellArgs ca; memset(&ca, 0, sizeof(ca));
            ca.rec_off = (const uint32_t *)c->b_rec_off.p; ca.dir = (const uint32_t *)c->b_dir.p; ca.stride = stride; ca.nrows = nrows; ca.tidy = c->bsch.tidy ? 1 : 0;
            ca.n_smp = c->bsch.n_samples > 0 ? c->bsch.n_samples : 1; ca.lens = (uint32_t *)c->b_lens.p; ca.offs = (const uint32_t *)c->b_offs.p; ca.ostride = ostride;
            ca.cols = (const BcfColDev *)c->b_coldev.p; ca.sel = sel; ca.ncols = (uint32_t)ncols;
            {
                KTimer tm(c, DHTS_K_BCF_MEASURE);
                hipLaunchKernelGGL(bcf_cells<false>, dim3((unsigned)(((nrows + 255) / 256) * ncols)), dim3(256), 0, c->stream, st, ca);
            }
            std::vector<uint64_t> tot(nsa ? nsa : 1, 0);
            uint64_t var_total = 0;
            if (nsa > 0) {
                MScanArgs ma; ma.in = (const uint32_t *)c->b_lens.p; ma.out = (uint32_t *)c->b_offs.p; ma.stride = ostride; ma.n = nrows;
                ma.nparts = (nrows + 1 + SCAN_ITEMS - 1) / SCAN_ITEMS; if (ma.nparts < 1) ma.nparts = 1;
                ENSURE(c, c->b_partial, (size_t)nsa * ma.nparts * 8 + 64); ENSURE(c, c->b_total, (size_t)nsa * 8 + 64);
                ma.partial = (uint64_t *)c->b_partial.p; ma.total = (uint64_t *)c->b_total.p;
                {
                    KTimer tm(c, DHTS_K_SCAN);
                    hipLaunchKernelGGL(mscan_reduce, dim3((unsigned)ma.nparts, (unsigned)nsa), dim3(256), 0, c->stream, ma);
                    hipLaunchKernelGGL(mscan_partials, dim3((unsigned)nsa), dim3(1024), 0, c->stream, ma);
                    hipLaunchKernelGGL(mscan_apply, dim3((unsigned)ma.nparts, (unsigned)nsa), dim3(256), 0, c->stream, ma);
                }
                HIPCHK(c, hipMemcpyAsync(tot.data(), c->b_total.p, (size_t)nsa * 8, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(c, hipStreamSynchronize(c->stream));
                for (int k = 0; k < nsa; k++) if (tot[k] >= (1ull << 32)) return fail(c, "read_bcf: a column exceeds 4 GiB in one batch; use a smaller max_blocks");
                // arena for children / bytes
                size_t var_bytes = 0; std::vector<size_t> at_child(ncols, 0), at_coff(ncols, 0), at_bytes(ncols, 0), at_cvalid(ncols, 0);
                for (int i = 0; i < ncols; i++) {
                    if (cd[i].kind == dhts::BK_VEP) { at_cvalid[i] = var_bytes; var_bytes += (tot[cd[i].sa_cnt] + 63) & ~(size_t)63; }
                    if (cd[i].sa_cnt >= 0 && cd[i].sa_bytes < 0) { at_child[i] = var_bytes; var_bytes += (tot[cd[i].sa_cnt] * 4 + 63) & ~(size_t)63; }
                    if (cd[i].sa_cnt >= 0 && cd[i].sa_bytes >= 0) { at_coff[i] = var_bytes; var_bytes += ((tot[cd[i].sa_cnt] + 1) * 4 + 63) & ~(size_t)63; }
                    if (cd[i].sa_bytes >= 0) { at_bytes[i] = var_bytes; var_bytes += (tot[cd[i].sa_bytes] + 63) & ~(size_t)63; }
                }
                ENSURE(c, c->b_var, var_bytes + 64); var_total = var_bytes;
                for (int i = 0; i < ncols; i++) {
                    uint8_t *base = (uint8_t *)c->b_var.p;
                    if (cd[i].sa_cnt >= 0 && cd[i].sa_bytes < 0) cd[i].child_fixed = (uint32_t *)(base + at_child[i]);
                    if (cd[i].sa_cnt >= 0 && cd[i].sa_bytes >= 0) cd[i].child_off = (uint32_t *)(base + at_coff[i]);
                    if (cd[i].sa_bytes >= 0) cd[i].bytes = base + at_bytes[i];
                    if (cd[i].kind == dhts::BK_VEP) cd[i].child_valid = base + at_cvalid[i];
                }
                HIPCHK(c, hipMemcpyAsync(c->b_coldev.p, cd.data(), sizeof(BcfColDev) * ncols, hipMemcpyHostToDevice, c->stream));
                {
                    KTimer tm(c, DHTS_K_BCF_WRITE);
                    hipLaunchKernelGGL(bcf_cells<true>, dim3((unsigned)(((nrows + 255) / 256) * ncols)), dim3(256), 0, c->stream, st, ca);
                }
            }
            HIPCHK(c, hipGetLastError());
            c->bcf_ar[0].p = (const uint8_t *)c->b_valid.p; c->bcf_ar[0].n = (uint64_t)ncols * (uint64_t)nrows;
            c->bcf_ar[1].p = (const uint8_t *)c->b_fixed.p; c->bcf_ar[1].n = fixed_bytes;
            c->bcf_ar[2].p = (const uint8_t *)c->b_offs.p; c->bcf_ar[2].n = (uint64_t)nsa * ostride * 4;
            c->bcf_ar[3].p = (const uint8_t *)c->b_var.p; c->bcf_ar[3].n = var_total;
            for (int i = 0; i < ncols; i++) {
                dhts_bcf_col &o = c->bcf_out[i];
                o.valid = cd[i].valid; o.fixed = cd[i].fixed;
                if (cd[i].sa_cnt >= 0) { o.off = (const uint32_t *)c->b_offs.p + (size_t)cd[i].sa_cnt * ostride; o.child_n = tot[cd[i].sa_cnt]; o.child_fixed = cd[i].child_fixed; o.child_off = cd[i].child_off; o.child_valid = cd[i].child_valid; }
                else if (cd[i].sa_bytes >= 0) o.off = (const uint32_t *)c->b_offs.p + (size_t)cd[i].sa_bytes * ostride;
                if (cd[i].sa_bytes >= 0) { o.bytes = cd[i].bytes; o.nbytes = tot[cd[i].sa_bytes]; }
            }
        }
    }
    out->n_rows = nrec * reps; c->last_bcf_u = st.u;
    out->end_uoff = out_base + carry_start;
    out->first_rec_uoff = nrec > 0 ? out_base + rec0_off : NONE64;
    return batch_end(c, B, carry_start, rec_err, shard_finished, &out->status);
}

#ifdef DHTS_DIAG
int dhts_debug_diag(dhts_ctx *c, unsigned long long *out8) {
    if (!c) return -1;
    HIPCHK(c, hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_diag), 64));
    HIPCHK(c, hipMemcpyFromSymbol(out8 + 8, HIP_SYMBOL(g_diagt), 64));
    HIPCHK(c, hipMemcpyFromSymbol(out8 + 16, HIP_SYMBOL(g_diagA), 64));
    return 0;
}
#endif
#ifdef HW_DIAG
extern "C" int dhts_debug_hw_diag(dhts_ctx *c, unsigned long long *out16, int reset) {
    if (!c) return -1;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_hw_diag), 128));
    if (reset) { unsigned long long z[16] = {0}; HIPCHK(c, hipMemcpyToSymbol(HIP_SYMBOL(g_hw_diag), z, 128)); }
    return 0;
}
#endif
// debugging aid (not part of the public header): phase-A metadata of scratch slot s
// kernel experiments (tools/dbg): phase A alone over blocks [b0, b0+nb), `reps` launches; returns ms per launch
extern "C" double dhts_debug_time_huff(dhts_ctx *c, int64_t b0, int64_t nb, int reps) {
    if (!c || hipSetDevice(c->device) != hipSuccess) return -1;
    if (huff_blocks(c, b0, nb)) return -1;
    (void)hipStreamSynchronize(c->stream);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, c->stream);
    for (int r = 0; r < reps; r++) if (huff_blocks(c, b0, nb)) return -1;
    (void)hipEventRecord(e1, c->stream); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return ms / reps;
}
// cross-check aid for the tests (not part of the public header): run ONE phase-A kernel over blocks [b0, b0+nb) (kernel: 0 / 1 = lane
// per block with all symbols in LDS / with the far table, 2 = wave per block), then read a scratch slot back
extern "C" int dhts_debug_huff_run(dhts_ctx *c, int64_t b0, int64_t nb, int kernel) {
    if (!c || b0 < 0 || nb <= 0 || b0 + nb > c->n_blocks || kernel < 0 || kernel > 3) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    discard_prefetch(c);
    if (huff_blocks(c, b0, nb, kernel)) return -1;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}
extern "C" int dhts_debug_scratch_get(dhts_ctx *c, int64_t s, uint32_t *meta4, uint8_t *lit, uint32_t *tok) {
    if (!c || s < 0 || s >= c->huff_nb) return -1;
    HIPCHK(c, hipMemcpy(meta4, (InflateMeta *)c->meta.p + s, 16, hipMemcpyDeviceToHost));
    if (meta4[3] != 0u) return 0;                                     // failed block: the scratch content is unspecified
    if (meta4[1] > DHTS_LIT_STRIDE || meta4[0] > DHTS_TOK_STRIDE) return fail(c, "scratch counts out of range");
    if (c->huff_packed) {
        unsigned long long off = 0;
        HIPCHK(c, hipMemcpy(&off, (unsigned long long *)c->blk_off.p + s, 8, hipMemcpyDeviceToHost));
        const uint8_t *lp = (const uint8_t *)c->lit.p + off;
        if (lit && meta4[1]) HIPCHK(c, hipMemcpy(lit, lp, meta4[1], hipMemcpyDeviceToHost));
        if (tok && meta4[0]) HIPCHK(c, hipMemcpy(tok, lp + ((meta4[1] + 15u) & ~15u), (size_t)meta4[0] * 4, hipMemcpyDeviceToHost));
        return 0;
    }
    if (lit && meta4[1]) HIPCHK(c, hipMemcpy(lit, (uint8_t *)c->lit.p + (size_t)s * DHTS_LIT_STRIDE, meta4[1], hipMemcpyDeviceToHost));
    if (tok && meta4[0]) HIPCHK(c, hipMemcpy(tok, (uint32_t *)c->tok.p + (size_t)s * DHTS_TOK_STRIDE, (size_t)meta4[0] * 4, hipMemcpyDeviceToHost));
    return 0;
}
int dhts_debug_meta(dhts_ctx *c, int64_t s, uint32_t *out4) {
    if (!c || s < 0 || s >= c->huff_nb) return -1;
    HIPCHK(c, hipMemcpy(out4, (InflateMeta *)c->meta.p + s, 16, hipMemcpyDeviceToHost));
    return 0;
}

// ---- one batch -> one host arena --------------------------------------------------------------------------------------------------
// The projected core columns of a batch are laid out back to back (64-byte aligned pieces) in caller memory -- pinned memory from
// dhts_host_alloc makes the copies true DMA -- with every copy queued before the single wait, instead of one synchronous copy per
// column array.  `out` is `b` with HOST pointers (tag columns, the auxiliary map and the overlap lists stay device pointers).
static inline uint64_t al64(uint64_t v) { return (v + 63u) & ~(uint64_t)63; }
uint64_t dhts_bam_batch_host_bytes(const dhts_bam_batch *b, uint32_t m) {
    if (!b || b->n_rows <= 0) return 0;
    const uint64_t n = (uint64_t)b->n_rows; uint64_t t = 0;
    if (m & (1u << DHTS_BAM_FLAG)) t += al64(n * 2);
    if (m & (1u << DHTS_BAM_POS)) t += al64(n * 8);
    if (m & (1u << DHTS_BAM_MAPQ)) t += al64(n * 4);
    if (m & (1u << DHTS_BAM_PNEXT)) t += al64(n * 8);
    if (m & (1u << DHTS_BAM_TLEN)) t += al64(n * 8);
    if (m & (1u << DHTS_BAM_RNAME)) t += al64(n * 4);
    if (m & (1u << DHTS_BAM_RNEXT)) t += al64(n * 4);
    if (m & (1u << DHTS_BAM_SAMPLE_ID)) t += al64(n * 4);
    if (m & ((1u << DHTS_BAM_READ_GROUP_ID) | (1u << DHTS_BAM_SAMPLE_ID))) t += al64(((n + 63) / 64) * 8);
    const dhts_strcol *sc[5] = {&b->qname, &b->cigar, &b->seq, &b->qual, &b->rg};
    const int bit[5] = {DHTS_BAM_QNAME, DHTS_BAM_CIGAR, DHTS_BAM_SEQ, DHTS_BAM_QUAL, DHTS_BAM_READ_GROUP_ID};
    for (int k = 0; k < 5; k++) if (m & (1u << bit[k])) t += al64((n + 1) * 4) + al64(n * 4) + al64(sc[k]->nbytes + 1);
    return t;
}
int dhts_bam_batch_fetch(dhts_ctx *c, const dhts_bam_batch *b, uint32_t m, void *dst, uint64_t cap, dhts_bam_batch *out) {
    if (!c || !b || !out) return -1;
    *out = *b;
    if (b->n_rows <= 0) return 0;
    const uint64_t need = dhts_bam_batch_host_bytes(b, m);
    if (need > cap || (need && !dst)) return fail(c, "host arena too small for the batch");
    HIPCHK(c, hipSetDevice(c->device));
    const uint64_t n = (uint64_t)b->n_rows; uint8_t *h = (uint8_t *)dst; uint64_t at = 0;
    auto put = [&](const void *src, uint64_t bytes) -> const void * {
        void *d = h + at; at += al64(bytes);
        if (bytes && hipMemcpyAsync(d, src, bytes, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return nullptr;
        return d;
    };
#define DHTS_FETCH(field, type, bitno, bytes) do { if (m & (1u << (bitno))) { out->field = (const type *)put(b->field, (bytes)); if (!out->field) return fail(c, "hipMemcpyAsync failed"); } else out->field = nullptr; } while (0)
    DHTS_FETCH(flag, uint16_t, DHTS_BAM_FLAG, n * 2); DHTS_FETCH(pos, int64_t, DHTS_BAM_POS, n * 8); DHTS_FETCH(mapq, int32_t, DHTS_BAM_MAPQ, n * 4);
    DHTS_FETCH(pnext, int64_t, DHTS_BAM_PNEXT, n * 8); DHTS_FETCH(tlen, int64_t, DHTS_BAM_TLEN, n * 8); DHTS_FETCH(tid, int32_t, DHTS_BAM_RNAME, n * 4);
    DHTS_FETCH(mtid, int32_t, DHTS_BAM_RNEXT, n * 4); DHTS_FETCH(rg_idx, int32_t, DHTS_BAM_SAMPLE_ID, n * 4);
#undef DHTS_FETCH
    if (m & ((1u << DHTS_BAM_READ_GROUP_ID) | (1u << DHTS_BAM_SAMPLE_ID))) { out->rg_valid = (const uint64_t *)put(b->rg_valid, ((n + 63) / 64) * 8); if (!out->rg_valid) return fail(c, "hipMemcpyAsync failed"); }
    else out->rg_valid = nullptr;
    const dhts_strcol *sc[5] = {&b->qname, &b->cigar, &b->seq, &b->qual, &b->rg};
    dhts_strcol *oc[5] = {&out->qname, &out->cigar, &out->seq, &out->qual, &out->rg};
    const int bit[5] = {DHTS_BAM_QNAME, DHTS_BAM_CIGAR, DHTS_BAM_SEQ, DHTS_BAM_QUAL, DHTS_BAM_READ_GROUP_ID};
    for (int k = 0; k < 5; k++) {
        if (!(m & (1u << bit[k]))) { oc[k]->off = oc[k]->len = nullptr; oc[k]->bytes = nullptr; oc[k]->nbytes = 0; continue; }
        oc[k]->off = (const uint32_t *)put(sc[k]->off, (n + 1) * 4); oc[k]->len = (const uint32_t *)put(sc[k]->len, n * 4);
        const uint64_t at0 = at;
        oc[k]->bytes = (const uint8_t *)put(sc[k]->bytes, sc[k]->nbytes);
        at = at0 + al64(sc[k]->nbytes + 1);                             // (one readable byte behind the heap, as in the size formula)
        if (!oc[k]->off || !oc[k]->len || !oc[k]->bytes) return fail(c, "hipMemcpyAsync failed");
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// The same read-back, overlapped with the next batch: the columns are gathered into snapshot `slot` (0 / 1) on the scan stream -- device to
// device, in the layout of the host arena -- and ONE copy takes the snapshot to the host on a copy stream; the call returns at once, `out`
// already points into dst.  dhts_bam_batch_fetch_wait(slot) returns when the bytes have landed.  The scan may go on to the next batch in
// between: its kernels are ordered behind the gather on the scan stream and never touch the snapshot.
static int copy_set_acquire(dhts_ctx *c) {
    if (c->copy_stream) return 0;
    {
        std::lock_guard<std::mutex> lk(g_cs_mu);
        for (size_t i = 0; i < g_cs.size(); i++) if (g_cs[i].dev == c->device) {
            c->copy_stream = g_cs[i].s; for (int k = 0; k < 2; k++) { c->ev_snap[k] = g_cs[i].snap[k]; c->ev_done[k] = g_cs[i].done[k]; }
            g_cs[i] = g_cs.back(); g_cs.pop_back();
            return 0;
        }
    }
    // (highest priority: the runtime maps streams onto a handful of hardware queues, and a copy stream that lands on the scan stream's queue
    //  serialises behind its kernels -- every other query of a process did, when the two pooled stream sets swapped roles; the high-priority
    //  streams have queues of their own)
    int pr_lo = 0, pr_hi = 0; (void)hipDeviceGetStreamPriorityRange(&pr_lo, &pr_hi);
    if (hipStreamCreateWithPriority(&c->copy_stream, hipStreamNonBlocking, pr_hi) != hipSuccess) { c->copy_stream = nullptr; return fail(c, "hipStreamCreate failed"); }
    for (int k = 0; k < 2; k++)
        if (hipEventCreateWithFlags(&c->ev_snap[k], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_done[k], hipEventDisableTiming) != hipSuccess) return fail(c, "hipEventCreate failed");
    return 0;
}
extern "C" int dhts_bam_batch_fetch_begin(dhts_ctx *c, const dhts_bam_batch *b, uint32_t m, void *dst, uint64_t cap, dhts_bam_batch *out, int slot) {
    if (!c || !b || !out || slot < 0 || slot > 1) return -1;
    *out = *b;
    if (b->n_rows <= 0) return 0;
    const uint64_t need = dhts_bam_batch_host_bytes(b, m);
    if (need > cap || (need && !dst)) return fail(c, "host arena too small for the batch");
    HIPCHK(c, hipSetDevice(c->device));
    if (copy_set_acquire(c)) return -1;
    // (the snapshot may still be on its way to the host from two batches ago)
    HIPCHK(c, hipEventSynchronize(c->ev_done[slot]));
    if (c->snap[slot].ensure(need + 64)) return fail(c, "hipMalloc failed");
    const uint64_t n = (uint64_t)b->n_rows; uint8_t *h = (uint8_t *)dst, *sp = (uint8_t *)c->snap[slot].p; uint64_t at = 0;
    auto put = [&](const void *src, uint64_t bytes) -> const void * {
        void *d = h + at;
        if (bytes && hipMemcpyAsync(sp + at, src, bytes, hipMemcpyDeviceToDevice, c->stream) != hipSuccess) return nullptr;
        at += al64(bytes);
        return d;
    };
#define DHTS_FETCH(field, type, bitno, bytes) do { if (m & (1u << (bitno))) { out->field = (const type *)put(b->field, (bytes)); if (!out->field) return fail(c, "hipMemcpyAsync failed"); } else out->field = nullptr; } while (0)
    DHTS_FETCH(flag, uint16_t, DHTS_BAM_FLAG, n * 2); DHTS_FETCH(pos, int64_t, DHTS_BAM_POS, n * 8); DHTS_FETCH(mapq, int32_t, DHTS_BAM_MAPQ, n * 4);
    DHTS_FETCH(pnext, int64_t, DHTS_BAM_PNEXT, n * 8); DHTS_FETCH(tlen, int64_t, DHTS_BAM_TLEN, n * 8); DHTS_FETCH(tid, int32_t, DHTS_BAM_RNAME, n * 4);
    DHTS_FETCH(mtid, int32_t, DHTS_BAM_RNEXT, n * 4); DHTS_FETCH(rg_idx, int32_t, DHTS_BAM_SAMPLE_ID, n * 4);
#undef DHTS_FETCH
    if (m & ((1u << DHTS_BAM_READ_GROUP_ID) | (1u << DHTS_BAM_SAMPLE_ID))) { out->rg_valid = (const uint64_t *)put(b->rg_valid, ((n + 63) / 64) * 8); if (!out->rg_valid) return fail(c, "hipMemcpyAsync failed"); }
    else out->rg_valid = nullptr;
    const dhts_strcol *sc[5] = {&b->qname, &b->cigar, &b->seq, &b->qual, &b->rg};
    dhts_strcol *oc[5] = {&out->qname, &out->cigar, &out->seq, &out->qual, &out->rg};
    const int bit[5] = {DHTS_BAM_QNAME, DHTS_BAM_CIGAR, DHTS_BAM_SEQ, DHTS_BAM_QUAL, DHTS_BAM_READ_GROUP_ID};
    for (int k = 0; k < 5; k++) {
        if (!(m & (1u << bit[k]))) { oc[k]->off = oc[k]->len = nullptr; oc[k]->bytes = nullptr; oc[k]->nbytes = 0; continue; }
        oc[k]->off = (const uint32_t *)put(sc[k]->off, (n + 1) * 4); oc[k]->len = (const uint32_t *)put(sc[k]->len, n * 4);
        const uint64_t at0 = at;
        oc[k]->bytes = (const uint8_t *)put(sc[k]->bytes, sc[k]->nbytes);
        at = at0 + al64(sc[k]->nbytes + 1);                             // (one readable byte behind the heap, as in the size formula)
        if (!oc[k]->off || !oc[k]->len || !oc[k]->bytes) return fail(c, "hipMemcpyAsync failed");
    }
    HIPCHK(c, hipEventRecord(c->ev_snap[slot], c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->copy_stream, c->ev_snap[slot], 0));
    HIPCHK(c, hipMemcpyAsync(dst, sp, at, hipMemcpyDeviceToHost, c->copy_stream));
    HIPCHK(c, hipEventRecord(c->ev_done[slot], c->copy_stream));
    return 0;
}
extern "C" int dhts_bam_batch_fetch_wait(dhts_ctx *c, int slot) {
    if (!c || slot < 0 || slot > 1) return -1;
    if (!c->copy_stream) return 0;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipEventSynchronize(c->ev_done[slot]));
    return 0;
}

// read_bcf batches in host memory: the columns of a batch live in four device arenas (validity, fixed payloads, the offset matrix,
// children / bytes), so the read-back is four queued copies and one wait, and every column pointer is re-based onto the host copy
uint64_t dhts_bcf_batch_host_bytes(const dhts_ctx *c) {
    if (!c) return 0;
    uint64_t t = 0; for (auto &a : c->bcf_ar) t += al64(a.n);
    return t;
}
int dhts_bcf_batch_fetch(dhts_ctx *c, const dhts_bcf_batch *b, void *dst, uint64_t cap, dhts_bcf_col *out_cols) {
    if (!c || !b || (b->n_cols > 0 && !out_cols)) return -1;
    for (int i = 0; i < b->n_cols; i++) out_cols[i] = b->cols[i];
    if (b->n_rows <= 0 || b->n_cols <= 0) return 0;
    const uint64_t need = dhts_bcf_batch_host_bytes(c);
    if (need > cap || (need && !dst)) return fail(c, "host arena too small for the batch");
    HIPCHK(c, hipSetDevice(c->device));
    uint8_t *h = (uint8_t *)dst; uint64_t at = 0; const uint8_t *hb[4];
    for (int k = 0; k < 4; k++) {
        hb[k] = h + at;
        if (c->bcf_ar[k].n) HIPCHK(c, hipMemcpyAsync(h + at, c->bcf_ar[k].p, c->bcf_ar[k].n, hipMemcpyDeviceToHost, c->stream));
        at += al64(c->bcf_ar[k].n);
    }
    auto rebase = [&](const void *p) -> const void * {
        if (!p) return nullptr;
        const uint8_t *q = (const uint8_t *)p;
        for (int k = 0; k < 4; k++) if (c->bcf_ar[k].p && q >= c->bcf_ar[k].p && q <= c->bcf_ar[k].p + c->bcf_ar[k].n) return hb[k] + (q - c->bcf_ar[k].p);
        return nullptr;
    };
    for (int i = 0; i < b->n_cols; i++) {
        dhts_bcf_col &o = out_cols[i]; const dhts_bcf_col &d = b->cols[i];
        o.valid = (const uint8_t *)rebase(d.valid); o.fixed = rebase(d.fixed); o.off = (const uint32_t *)rebase(d.off); o.bytes = (const uint8_t *)rebase(d.bytes);
        o.child_fixed = (const uint32_t *)rebase(d.child_fixed); o.child_off = (const uint32_t *)rebase(d.child_off); o.child_valid = (const uint8_t *)rebase(d.child_valid);
        if ((d.valid && !o.valid) || (d.fixed && !o.fixed) || (d.off && !o.off) || (d.bytes && !o.bytes) || (d.child_fixed && !o.child_fixed) || (d.child_off && !o.child_off) || (d.child_valid && !o.child_valid))
            return fail(c, "batch column outside the batch arenas");
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// read_bcf's read-back overlapped with the next batch (see dhts_bam_batch_fetch_begin): the four arenas are gathered into snapshot `slot`
// on the scan stream and leave as one copy on the copy stream; out_cols point into dst at once, the bytes are there after _wait(slot).
extern "C" int dhts_bcf_batch_fetch_begin(dhts_ctx *c, const dhts_bcf_batch *b, void *dst, uint64_t cap, dhts_bcf_col *out_cols, int slot) {
    if (!c || !b || (b->n_cols > 0 && !out_cols) || slot < 0 || slot > 1) return -1;
    for (int i = 0; i < b->n_cols; i++) out_cols[i] = b->cols[i];
    if (b->n_rows <= 0 || b->n_cols <= 0) return 0;
    const uint64_t need = dhts_bcf_batch_host_bytes(c);
    if (need > cap || (need && !dst)) return fail(c, "host arena too small for the batch");
    HIPCHK(c, hipSetDevice(c->device));
    if (copy_set_acquire(c)) return -1;
    HIPCHK(c, hipEventSynchronize(c->ev_done[slot]));
    if (c->snap[slot].ensure(need + 64)) return fail(c, "hipMalloc failed");
    uint8_t *h = (uint8_t *)dst, *sp = (uint8_t *)c->snap[slot].p; uint64_t at = 0; const uint8_t *hb[4];
    for (int k = 0; k < 4; k++) {
        hb[k] = h + at;
        if (c->bcf_ar[k].n) HIPCHK(c, hipMemcpyAsync(sp + at, c->bcf_ar[k].p, c->bcf_ar[k].n, hipMemcpyDeviceToDevice, c->stream));
        at += al64(c->bcf_ar[k].n);
    }
    auto rebase = [&](const void *p) -> const void * {
        if (!p) return nullptr;
        const uint8_t *q = (const uint8_t *)p;
        for (int k = 0; k < 4; k++) if (c->bcf_ar[k].p && q >= c->bcf_ar[k].p && q <= c->bcf_ar[k].p + c->bcf_ar[k].n) return hb[k] + (q - c->bcf_ar[k].p);
        return nullptr;
    };
    for (int i = 0; i < b->n_cols; i++) {
        dhts_bcf_col &o = out_cols[i]; const dhts_bcf_col &d = b->cols[i];
        o.valid = (const uint8_t *)rebase(d.valid); o.fixed = rebase(d.fixed); o.off = (const uint32_t *)rebase(d.off); o.bytes = (const uint8_t *)rebase(d.bytes);
        o.child_fixed = (const uint32_t *)rebase(d.child_fixed); o.child_off = (const uint32_t *)rebase(d.child_off); o.child_valid = (const uint8_t *)rebase(d.child_valid);
        if ((d.valid && !o.valid) || (d.fixed && !o.fixed) || (d.off && !o.off) || (d.bytes && !o.bytes) || (d.child_fixed && !o.child_fixed) || (d.child_off && !o.child_off) || (d.child_valid && !o.child_valid))
            return fail(c, "batch column outside the batch arenas");
    }
    HIPCHK(c, hipEventRecord(c->ev_snap[slot], c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->copy_stream, c->ev_snap[slot], 0));
    if (at) HIPCHK(c, hipMemcpyAsync(dst, sp, at, hipMemcpyDeviceToHost, c->copy_stream));
    HIPCHK(c, hipEventRecord(c->ev_done[slot], c->copy_stream));
    return 0;
}
extern "C" int dhts_bcf_batch_fetch_wait(dhts_ctx *c, int slot) { return dhts_bam_batch_fetch_wait(c, slot); }

int dhts_memcpy_d2h(dhts_ctx *c, void *dst, const void *src_dev, uint64_t n) {
    if (!c) return -1;
    if (n == 0) return 0;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(dst, src_dev, n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int dhts_sync(dhts_ctx *c) { if (!c) return -1; HIPCHK(c, hipSetDevice(c->device)); HIPCHK(c, hipStreamSynchronize(c->stream)); timing_collect(c); return 0; }

double dhts_kernel_time_ms(const dhts_ctx *c, int id, int64_t *launches) {
    if (!c || id < 0 || id >= DHTS_K_COUNT) return 0.0;
    if (launches) *launches = c->k_n[id];
    return c->k_ms[id];
}
void dhts_kernel_time_reset(dhts_ctx *c) { if (!c) return; for (int i = 0; i < DHTS_K_COUNT; i++) { c->k_ms[i] = 0; c->k_n[i] = 0; } }
void dhts_set_timing(dhts_ctx *c, int enabled) { if (c) c->timing = enabled != 0; }

}  // extern "C"
