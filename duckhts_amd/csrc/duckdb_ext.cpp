// duckdb_ext.cpp -- DuckDB C-API extension surface (outer drop-in boundary) over the MI355X scan path.
//
// Mirrors, callback for callback, the reference's read_bam table function:
//   register_read_bam_function   src/bam_reader.c:1044-1068
//   bam_read_bind                src/bam_reader.c:410-557   (parameters, schema, error strings)
//   bam_read_global_init         src/bam_reader.c:563-588
//   bam_read_local_init          src/bam_reader.c:594-682   (projection ids)
//   bam_read_function            src/bam_reader.c:722-1038  (<= vector_size rows per call, size 0 = done)
// The htslib calls underneath are replaced by include/duckhts_amd.h (HIP kernels); DuckDB is reached only
// through the function-pointer table returned by access->get_api(info, "v1.2.0").
//
// Scan mode: the reference's sequential mode (i): all records in file order including
// unplaced reads, full 2048-row chunks except the last.  region, standard_tags and auxiliary_tags are served by
// the GPU path; CRAM / SAM text input fails at bind with the reference's header error.
//
// Linkage (src/duckhts.c:13-16,54-55): register_read_bam_function / register_read_bcf_function have external
// linkage and read the DuckDB API through the global `duckdb_ext_api`, exactly like the reference's readers
// (DUCKDB_EXTENSION_EXTERN, duckdb_extension.h:1161), so a reference-built src/duckhts.c links against them
// unchanged.  This file also carries a weak definition of that global and a weak duckhts_init_c_api, which
// are what a stand-alone libduckhts_amd.so uses; strong definitions from src/duckhts.c win at link time.
#include "../../include/duckhts_amd.h"
#include "../../include/duckhts_extension.h"

#include <stdio.h>
#include <time.h>
#include <stdlib.h>
#include <string.h>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <cmath>
#include <string>
#include <thread>
#include <vector>

// duckdb_ext_api_v1 viewed as an array of function pointers (slot numbers: include/duckdb_abi_slots.h).  Weak: the definition
// DUCKDB_EXTENSION_GLOBAL emits in a reference-built src/duckhts.c (duckdb_extension.h:1151) replaces it.
extern "C" __attribute__((visibility("default"), weak)) const void *duckdb_ext_api[DUCKDB_ABI_V120_NSLOTS] = {nullptr};

#define API(ret, name, ...) ((ret(*)(__VA_ARGS__))duckdb_ext_api[SLOT_##name])

// what DUCKDB_EXTENSION_API_INIT does (`duckdb_ext_api = *res`, duckdb_extension.h:1153-1158), for hosts that hold the table
extern "C" __attribute__((visibility("default"))) void dhts_set_duckdb_api(const void *api_table) {
    if (api_table) memcpy((void *)duckdb_ext_api, api_table, sizeof(void *) * DUCKDB_ABI_V120_NSLOTS);
}

extern "C" void dhts_debug_malloc_stats(uint64_t *calls, uint64_t *bytes, double *seconds);
static double now_s() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

static inline void set_null(duckdb_vector vec, idx_t row) {           // src/bam_reader.c:38-42
    API(void, duckdb_vector_ensure_validity_writable, duckdb_vector)(vec);
    uint64_t *v = API(uint64_t *, duckdb_vector_get_validity, duckdb_vector)(vec);
    v[row / 64] &= ~((uint64_t)1 << (row % 64));
}

struct BamBind {
    std::string path, region, index_file;
    dhts_ctx *ctx = nullptr;          // bind-time context: holds only the head of the file (header + dictionaries)
    dhts_bam_header hdr;
    uint64_t header_bytes = 0;        // compressed bytes [0, header_bytes) cover the header blocks
    int has_index = 0;
    int standard_tags = 0, auxiliary_tags = 0;
    idx_t aux_col_idx = (idx_t)-1;
    std::vector<duckdb_string_t> ref_inl; std::vector<char> ref_is_inl;      // RNAME / RNEXT values of <= 12 bytes, ready to store
    duckdb_string_t star_inl;
};

// ---- scan pipeline: GPU producer thread(s) -> pinned host batches -> scan callbacks ------------------------------------------------
// The reference's callback reads one record at a time from its htsFile (src/bam_reader.c:747-1035).  Here a producer thread per GPU
// owns a scan context and turns the file into batches: stage (reader threads -> pinned -> HBM), inflate + unpack on the device, then
// ONE queued read-back of the projected columns into a pinned arena (dhts_bam_batch_fetch).  The scan callbacks only copy from
// those arenas into DataChunk vectors, so the device works on batch k+1 while the engine's threads fill chunks from batch k.
//   DHTS_THREADS = 1 (default): the reference's sequential mode (i) -- one worker, rows in file order, full 2048-row chunks.
//   DHTS_THREADS = k > 1: k workers claim 2048-row slices of the ready batches, the row ORDER across workers is unspecified,
//                  exactly like the reference's own parallel mode (contig-parallel, src/bam_reader.c:577-585, 689-716).
//   DHTS_DEVICES = 0,1,...: one producer per listed GPU, each staging and scanning its own BGZF block range of the file.
struct HostTag { std::vector<uint8_t> valid, bytes; std::vector<int64_t> fixed; std::vector<uint32_t> off; std::vector<int64_t> child; };
struct HostBatch {
    void *arena = nullptr; uint64_t cap = 0;
    dhts_bam_batch b;                 // HOST pointers for the core columns
    int64_t n = 0; int status = 0;
    std::vector<HostTag> tags;
    std::vector<uint8_t> aux_valid; std::vector<uint32_t> aux_off; std::vector<std::string> aux_key, aux_val;
    std::vector<uint32_t> qual_lut;   // QUAL as 2- / 4-bit codes (dhts_bam_batch.qual_bits): code byte -> its 4 / 2 characters, built when the batch's bytes have landed
    // parallel mode
    int64_t next = 0; int readers = 0; bool retired = false;
};
struct Producer {
    int device = 0, rank = 0, world = 1;
    std::thread th;
    std::deque<HostBatch *> ready; std::vector<HostBatch *> free_slots; std::vector<HostBatch *> all;
    bool done = false;
    // where this rank's rows begin and end in the file, as BGZF virtual offsets: adjacent ranks must meet exactly (SURVEY 8(e) hand-off)
    bool has_rows = false, clean_end = false; uint64_t first_v = 0, end_v = 0;
};
struct BamScan {
    BamBind *bind = nullptr;
    std::vector<idx_t> column_ids; uint32_t colmask = 0;
    std::vector<int32_t> tag_ids; std::vector<int> tag_slot; bool want_aux = false;
    int n_workers = 1;
    std::mutex mu; std::condition_variable cv_ready, cv_free;
    std::vector<Producer *> prod; size_t cur_prod = 0;
    std::string error; bool cancel = false, handoff_checked = false;
    std::vector<uint8_t> index_bytes;
    std::vector<uint64_t> seg_beg, seg_end; int64_t seg_count = -1;      // region query: the file byte ranges to stage (-1: the whole file)
    ~BamScan() {
        { std::lock_guard<std::mutex> lk(mu); cancel = true; }
        cv_free.notify_all(); cv_ready.notify_all();
        for (auto p : prod) { if (p->th.joinable()) p->th.join(); for (auto hb : p->all) { dhts_host_free(hb->arena); delete hb; } delete p; }
    }
};
struct BamLocal {
    std::vector<char> seq_tmp;         // packed SEQ expands here before it is assigned
    std::vector<char> qual_tmp;        // packed QUAL: a row that starts inside a code byte is expanded here first
    bool done = false;
    HostBatch *cur = nullptr; Producer *cur_owner = nullptr; int64_t pos = 0, end = 0;     // rows [pos, end) of `cur` are this worker's
};

static void destroy_bind(void *p) { BamBind *b = (BamBind *)p; if (!b) return; if (b->ctx) dhts_destroy(b->ctx); delete b; }
static void destroy_local(void *p) { delete (BamLocal *)p; }
static void destroy_global(void *p) { delete (BamScan *)p; }

static char *get_named_varchar(duckdb_bind_info info, const char *name) {
    duckdb_value v = API(duckdb_value, duckdb_bind_get_named_parameter, duckdb_bind_info, const char *)(info, name);
    char *s = nullptr;
    if (v && !API(bool, duckdb_is_null_value, duckdb_value)(v)) s = API(char *, duckdb_get_varchar, duckdb_value)(v);
    if (v) API(void, duckdb_destroy_value, duckdb_value *)(&v);
    return s;
}
static int get_named_bool(duckdb_bind_info info, const char *name) {
    duckdb_value v = API(duckdb_value, duckdb_bind_get_named_parameter, duckdb_bind_info, const char *)(info, name);
    int r = 0;
    if (v && !API(bool, duckdb_is_null_value, duckdb_value)(v)) r = API(bool, duckdb_get_bool, duckdb_value)(v) ? 1 : 0;
    if (v) API(void, duckdb_destroy_value, duckdb_value *)(&v);
    return r;
}
static bool file_exists(const std::string &p) { FILE *f = fopen(p.c_str(), "rb"); if (!f) return false; fclose(f); return true; }
static std::vector<int> device_list() {
    std::vector<int> d;
    if (const char *e = getenv("DHTS_DEVICES")) { for (const char *q = e; *q;) { char *end; long v = strtol(q, &end, 10); if (end == q) break; d.push_back((int)v); q = *end ? end + 1 : end; } }
    if (d.empty()) d.push_back(getenv("DHTS_DEVICE") ? atoi(getenv("DHTS_DEVICE")) : 0);
    return d;
}
// a string of <= 12 bytes is stored inside duckdb_string_t itself (duckdb.h:377-391: length, then the bytes, zero padded): no heap, no call
static inline bool inl_string(duckdb_string_t *d, const char *s, size_t len) {
    if (len > 12) return false;
    memset(d, 0, sizeof(*d)); d->value.inlined.length = (uint32_t)len; memcpy(d->value.inlined.inlined, s, len);
    return true;
}
// 4-bit base codes -> text, high nibble first ("=ACMGRSVTWYHKDBN", htslib hts.c:260, sam.h:325): 16 bases per step through pshufb
// (the table is the shuffle's own 16-byte lookup), a 512-byte pair table for the tail.  out must have room for n + 16 bytes.
#include <immintrin.h>
static const char kSeqNt16[] = "=ACMGRSVTWYHKDBN";
static uint16_t g_seq_pair[256];
static const bool g_seq_pair_init = [] { for (int b = 0; b < 256; b++) { const uint8_t p[2] = {(uint8_t)kSeqNt16[b >> 4], (uint8_t)kSeqNt16[b & 15]}; uint16_t v; memcpy(&v, p, 2); g_seq_pair[b] = v; } return true; }();
__attribute__((target("ssse3"))) static inline void expand_seq(const uint8_t *src, uint32_t n, char *out) {
    const __m128i tab = _mm_loadu_si128((const __m128i *)kSeqNt16), lo_mask = _mm_set1_epi8(0x0f);
    uint32_t i = 0;
    for (; i + 16 <= n; i += 16) {
        const __m128i v = _mm_loadl_epi64((const __m128i *)(src + i / 2));             // 8 bytes = 16 bases
        const __m128i hi = _mm_and_si128(_mm_srli_epi16(v, 4), lo_mask), lo = _mm_and_si128(v, lo_mask);
        _mm_storeu_si128((__m128i *)(out + i), _mm_shuffle_epi8(tab, _mm_unpacklo_epi8(hi, lo)));
    }
    for (; i < n; i += 2) { const uint16_t v = g_seq_pair[src[i / 2]]; memcpy(out + i, &v, 2); }    // (may write one byte past an odd n: room is there)
    (void)g_seq_pair_init;
}

static void bam_read_bind(duckdb_bind_info info) {
    auto set_error = API(void, duckdb_bind_set_error, duckdb_bind_info, const char *);
    auto dfree = API(void, duckdb_free, void *);
    duckdb_value pv = API(duckdb_value, duckdb_bind_get_parameter, duckdb_bind_info, idx_t)(info, 0);
    char *file_path = API(char *, duckdb_get_varchar, duckdb_value)(pv);
    API(void, duckdb_destroy_value, duckdb_value *)(&pv);
    if (!file_path || strlen(file_path) == 0) {
        set_error(info, "read_bam requires a file path");                         // bam_reader.c:416
        if (file_path) dfree(file_path);
        return;
    }
    char *region = get_named_varchar(info, "region");
    char *index_path = get_named_varchar(info, "index_path");
    char *reference = get_named_varchar(info, "reference");
    int standard_tags = get_named_bool(info, "standard_tags"), auxiliary_tags = get_named_bool(info, "auxiliary_tags");
    BamBind *b = new BamBind();
    b->path = file_path;
    std::string idx = index_path ? index_path : "";
    // parse_regions (bam_reader.c:319-345) splits with strtok: a string without a non-empty token ('' or ',,') is no region at all
    bool has_region = false;
    for (const char *q = region; q && *q; q++) if (*q != ',') { has_region = true; break; }
    std::string region_copy = has_region ? region : "";
    dfree(file_path); if (region) dfree(region); if (index_path) dfree(index_path); if (reference) dfree(reference);

    char err[768];
    if (!file_exists(b->path)) {
        snprintf(err, sizeof(err), "Failed to open SAM/BAM/CRAM file: %s", b->path.c_str());   // bam_reader.c:446
        set_error(info, err); delete b; return;
    }
    static const bool trace_bind = getenv("DHTS_TRACE") != nullptr;
    const double tb0 = now_s();
    b->ctx = dhts_create(device_list()[0]);
    const double tb1 = now_s();
    if (!b->ctx) { set_error(info, "read_bam: no MI355X (gfx950) device available; this build has no CPU fallback"); destroy_bind(b); return; }
    // like the reference, bind reads the header only (sam_open + sam_hdr_read, bam_reader.c:441-461): the head of the file is staged,
    // four times more whenever the header turns out to be longer.  The scan stages the file itself (bam_read_global_init).
    bool hdr_ok = false;
    for (uint64_t head = 1u << 20;; head *= 4) {
        if (dhts_open_path_range(b->ctx, b->path.c_str(), 0, head) != 0) {
            snprintf(err, sizeof(err), "Failed to open SAM/BAM/CRAM file: %s", b->path.c_str());
            set_error(info, err); destroy_bind(b); return;
        }
        const bool whole = dhts_resident_bytes(b->ctx) < head;
        if (dhts_bgzf_index(b->ctx) > 0 && dhts_bam_open(b->ctx) == 0 && dhts_bam_header_get(b->ctx, &b->hdr) == 0) { hdr_ok = true; break; }
        if (whole || head >= (1ull << 34)) break;
    }
    if (!hdr_ok) {
        set_error(info, "Failed to read SAM/BAM/CRAM header");                    // bam_reader.c:461 (also what SAM/CRAM input gets here)
        destroy_bind(b); return;
    }
    b->header_bytes = dhts_bam_header_bytes(b->ctx);
    if (trace_bind) fprintf(stderr, "[dhts] bind: context %.4f s, head of the file + block table + header %.4f s\n", tb1 - tb0, now_s() - tb1);
    for (int32_t i = 0; i < b->hdr.n_ref; i++) { duckdb_string_t t; b->ref_is_inl.push_back(inl_string(&t, b->hdr.ref_name[i], strlen(b->hdr.ref_name[i])) ? 1 : 0); b->ref_inl.push_back(t); }
    inl_string(&b->star_inl, "*", 1);
    // index lookup order of sam_index_load3 (hts.c:4720-4790): explicit path, <file>.csi, <file>.bai, <file minus .bam>.bai/.csi
    {
        std::string stem = b->path.size() > 4 && b->path.compare(b->path.size() - 4, 4, ".bam") == 0 ? b->path.substr(0, b->path.size() - 4) : std::string();
        std::vector<std::string> cand;
        if (!idx.empty()) cand.push_back(idx);
        else { cand.push_back(b->path + ".csi"); cand.push_back(b->path + ".bai"); if (!stem.empty()) { cand.push_back(stem + ".csi"); cand.push_back(stem + ".bai"); } }
        for (auto &f : cand) if (file_exists(f)) { b->index_file = f; break; }
    }
    b->has_index = !b->index_file.empty();                                       // bam_reader.c:499-503
    if (has_region) b->region = region_copy;
    b->standard_tags = standard_tags; b->auxiliary_tags = auxiliary_tags;

    auto mk = API(duckdb_logical_type, duckdb_create_logical_type, int);
    auto add = API(void, duckdb_bind_add_result_column, duckdb_bind_info, const char *, duckdb_logical_type);
    auto rm = API(void, duckdb_destroy_logical_type, duckdb_logical_type *);
    duckdb_logical_type t_varchar = mk(DUCKDB_TYPE_VARCHAR), t_int = mk(DUCKDB_TYPE_INTEGER), t_big = mk(DUCKDB_TYPE_BIGINT), t_us = mk(DUCKDB_TYPE_USMALLINT);
    add(info, "QNAME", t_varchar); add(info, "FLAG", t_us); add(info, "RNAME", t_varchar); add(info, "POS", t_big); add(info, "MAPQ", t_int);   // bam_reader.c:514-526
    add(info, "CIGAR", t_varchar); add(info, "RNEXT", t_varchar); add(info, "PNEXT", t_big); add(info, "TLEN", t_big); add(info, "SEQ", t_varchar);
    add(info, "QUAL", t_varchar); add(info, "READ_GROUP_ID", t_varchar); add(info, "SAMPLE_ID", t_varchar);
    if (b->standard_tags) {                                                                     // bam_reader.c:527-537 + bam_std_tag_type 88-104
        auto mklist = API(duckdb_logical_type, duckdb_create_list_type, duckdb_logical_type);
        duckdb_logical_type t_list = mklist(t_big);
        for (int i = 0; i < dhts_bam_std_tag_count(); i++) {
            char nm[3], ty, sub; dhts_bam_std_tag_info(i, nm, &ty, &sub);
            add(info, nm, ty == 'i' ? t_big : ty == 'B' ? t_list : t_varchar);
        }
        rm(&t_list);
    }
    if (b->auxiliary_tags) {                                                                    // bam_reader.c:539-548
        duckdb_logical_type t_map = API(duckdb_logical_type, duckdb_create_map_type, duckdb_logical_type, duckdb_logical_type)(t_varchar, t_varchar);
        b->aux_col_idx = DHTS_BAM_CORE_COUNT + (b->standard_tags ? (idx_t)dhts_bam_std_tag_count() : 0);
        add(info, "AUXILIARY_TAGS", t_map);
        rm(&t_map);
    }
    rm(&t_varchar); rm(&t_int); rm(&t_big); rm(&t_us);
    API(void, duckdb_bind_set_bind_data, duckdb_bind_info, void *, duckdb_delete_callback_t)(info, b, destroy_bind);
}

// copies the standard-tag columns and the auxiliary map of a device batch to pageable host memory (optional columns, off by default)
static int fetch_optional(dhts_ctx *c, BamScan *g, const dhts_bam_batch &b, HostBatch *hb) {
    const int64_t n = b.n_rows;
    if (g->want_aux && b.aux_map) {
        // typed entries -> value text, bam_aux_to_string (bam_reader.c:140-183); assigned through the NUL-terminated API
        const dhts_aux_map &am = *b.aux_map; const size_t ne = (size_t)am.n_ent;
        std::vector<uint16_t> key(ne + 1); std::vector<uint8_t> kind(ne + 1), sub(ne + 1), pay(am.payload_bytes + 8); std::vector<uint32_t> po(ne + 2);
        hb->aux_valid.resize(n); hb->aux_off.resize(n + 1);
        if (dhts_memcpy_d2h(c, hb->aux_valid.data(), am.valid, n) || dhts_memcpy_d2h(c, hb->aux_off.data(), am.off, (n + 1) * 4)) return -1;
        if (ne && (dhts_memcpy_d2h(c, key.data(), am.key, ne * 2) || dhts_memcpy_d2h(c, kind.data(), am.kind, ne) || dhts_memcpy_d2h(c, sub.data(), am.sub, ne))) return -1;
        if (dhts_memcpy_d2h(c, po.data(), am.pay_off, (ne + 1) * 4)) return -1;
        if (am.payload_bytes && dhts_memcpy_d2h(c, pay.data(), am.payload, am.payload_bytes)) return -1;
        hb->aux_key.assign(ne, std::string()); hb->aux_val.assign(ne, std::string());
        char tmp[64];
        for (size_t i = 0; i < ne; i++) {
            char kb[3] = {(char)(key[i] & 0xff), (char)(key[i] >> 8), 0};
            hb->aux_key[i] = kb;
            const uint8_t *p = pay.data() + po[i]; const size_t pl = po[i + 1] - po[i];
            std::string v;
            int64_t iv; double dv;
            switch (kind[i]) {
            case 0: memcpy(&iv, p, 8); snprintf(tmp, sizeof tmp, "%lld", (long long)iv); v = tmp; break;
            case 1: memcpy(&dv, p, 8); snprintf(tmp, sizeof tmp, "%g", dv); v = tmp; break;
            case 2: case 3: v.assign((const char *)p, pl); break;
            case 4: v.push_back((char)sub[i]); for (size_t q = 0; q < pl / 8; q++) { memcpy(&iv, p + 8 * q, 8); snprintf(tmp, sizeof tmp, ",%lld", (long long)iv); v += tmp; } break;
            case 5: v.push_back((char)sub[i]); for (size_t q = 0; q < pl / 8; q++) { memcpy(&dv, p + 8 * q, 8); snprintf(tmp, sizeof tmp, ",%g", dv); v += tmp; } break;
            default: break;
            }
            hb->aux_val[i] = v.c_str();                              // C-string semantics: cut at the first NUL
        }
    }
    hb->tags.resize(b.n_tag_cols);
    for (int i = 0; i < b.n_tag_cols; i++) {
        const dhts_col &d = b.tag_cols[i]; HostTag &h = hb->tags[i];
        h.valid.resize(n); if (dhts_memcpy_d2h(c, h.valid.data(), d.valid, n)) return -1;
        if (d.fixed) { h.fixed.resize(n); if (dhts_memcpy_d2h(c, h.fixed.data(), d.fixed, n * 8)) return -1; }
        if (d.off) { h.off.resize(n + 1); if (dhts_memcpy_d2h(c, h.off.data(), d.off, (n + 1) * 4)) return -1; }
        if (d.bytes || d.nbytes == 0) { h.bytes.resize(d.nbytes + 1); if (d.nbytes && dhts_memcpy_d2h(c, h.bytes.data(), d.bytes, d.nbytes)) return -1; }
        if (d.child_fixed) { h.child.resize(d.child_n + 1); if (d.child_n && dhts_memcpy_d2h(c, h.child.data(), d.child_fixed, d.child_n * 8)) return -1; }
    }
    return 0;
}

// producer thread: one GPU, one scan context, one block range of the file
// QUAL as codes of the batch's own alphabet (dhts_bam_batch.qual_bits = 2 / 4): qual.bytes = 16-byte symbol table + code stream, character k
// of the heap at bit k * bits.  The table of a batch: code byte -> its 4 (2-bit) or 2 (4-bit) characters.
static void build_qual_lut(HostBatch *hb) {
    hb->qual_lut.clear();
    const dhts_bam_batch &b = hb->b;
    if (!b.qual_bits || !b.qual.bytes) return;
    const uint8_t *sym = b.qual.bytes;
    hb->qual_lut.resize(256);
    for (uint32_t v = 0; v < 256; v++) {
        if (b.qual_bits == 2) hb->qual_lut[v] = (uint32_t)sym[v & 3] | ((uint32_t)sym[(v >> 2) & 3] << 8) | ((uint32_t)sym[(v >> 4) & 3] << 16) | ((uint32_t)sym[v >> 6] << 24);
        else hb->qual_lut[v] = (uint32_t)sym[v & 15] | ((uint32_t)sym[v >> 4] << 8);
    }
}
// characters [off, off + n) of the heap -> out[0, n) (out has room for n + 8)
static inline void expand_qual(const uint8_t *stream, int bits, const uint32_t *lut, uint32_t off, uint32_t n, char *out, std::vector<char> &tmp) {
    const uint32_t per = bits == 2 ? 4u : 2u, first = off / per, skip = off % per, nb = (skip + n + per - 1) / per;
    char *w = out;
    if (skip) { if (tmp.size() < (size_t)nb * per + 8) tmp.resize((size_t)nb * per + 8 + n / 2); w = tmp.data(); }
    if (bits == 2) for (uint32_t k = 0; k < nb; k++) { const uint32_t v = lut[stream[first + k]]; memcpy(w + 4 * k, &v, 4); }
    else for (uint32_t k = 0; k < nb; k++) { const uint16_t v = (uint16_t)lut[stream[first + k]]; memcpy(w + 2 * k, &v, 2); }
    if (skip) memcpy(out, w + skip, n);
}
static void producer_main(BamScan *g, Producer *p) {
    BamBind *bind = g->bind;
    static const bool trace = getenv("DHTS_TRACE") != nullptr;       // stage timings of every producer on stderr
    const double t_start = now_s(); double t_open = 0, t_gpu = 0, t_fetch = 0, t_slot = 0, t_wait = 0, t_index = 0; int64_t n_batches = 0, n_rows = 0, n_index = 0;
    auto fail_with = [&](const std::string &msg) {
        std::lock_guard<std::mutex> lk(g->mu);
        if (g->error.empty()) g->error = msg;
        p->done = true; g->cv_ready.notify_all();
    };
    // the producer, the staging readers it starts and the pinned arenas it allocates live on the NUMA node of its GPU
    const int numa_rc = dhts_bind_thread_near_device(p->device);
    if (trace) fprintf(stderr, "[dhts] producer %d: device %d on NUMA node %d (%s)\n", p->rank, p->device, dhts_device_numa_node(p->device), numa_rc == 0 ? "bound" : numa_rc == 1 ? "not bound" : "bind failed");
    dhts_ctx *c = dhts_create(p->device);
    if (!c) { fail_with("read_bam: no MI355X (gfx950) device available; this build has no CPU fallback"); return; }
    dhts_set_super_blocks(c, 196608);                    // a scratch the device pool keeps from query to query (29 GB instead of 67 GB for a 10 GB file)
    { static const bool env_qraw = getenv("DHTS_QUAL_PACKED") && atoi(getenv("DHTS_QUAL_PACKED")) == 0; dhts_bam_set_qual_packed(c, env_qraw ? 0 : 1); }         // QUAL crosses PCIe as 2- / 4-bit codes when the batch holds at most 4 / 16 different characters
    { static const bool env_unpacked = getenv("DHTS_SEQ_PACKED") && atoi(getenv("DHTS_SEQ_PACKED")) == 0; dhts_bam_set_seq_packed(c, env_unpacked ? 0 : 1); }   // SEQ crosses PCIe as 4-bit codes, the fill threads expand it
    const double t_created = now_s() - t_start; double t_staged = 0;
    int rc;
    // a plain whole-file scan on one device starts decoding while the file is still being staged: the block table is built over the
    // resident prefix and extended as more bytes arrive (DHTS_STREAM=0 stages the whole file first)
    static const bool env_nostream = getenv("DHTS_STREAM") && atoi(getenv("DHTS_STREAM")) == 0;
    const bool streaming = p->world == 1 && bind->region.empty() && !env_nostream;
    int staged_all = 1;
    if (g->seg_count >= 0) rc = dhts_open_path_segments(c, bind->path.c_str(), bind->header_bytes, g->seg_beg.data(), g->seg_end.data(), g->seg_count);
    else if (p->world > 1) rc = dhts_open_path_shard(c, bind->path.c_str(), p->rank, p->world, bind->header_bytes);
    else if (streaming) rc = dhts_open_path_async(c, bind->path.c_str());
    else rc = dhts_open_path(c, bind->path.c_str());
    t_staged = now_s() - t_start;
    const bool from_cache = rc == 0 && dhts_resident_from_cache(c) != 0;
    double t_idx = 0, t_hdr = 0;
    if (rc == 0 && !streaming) {
        const double q0 = now_s();
        if (dhts_bgzf_index(c) <= 0) rc = -1;
        const double q1 = now_s(); t_idx = q1 - q0;
        if (rc == 0 && dhts_bam_open(c) != 0) rc = -1;
        t_hdr = now_s() - q1;
    }
    if (rc == 0 && streaming) {
        // the header needs the first blocks only: start with what the bind saw, four times more whenever that is not enough
        uint64_t want = bind->header_bytes + (32u << 20);
        for (;;) {
            const int64_t f = dhts_stage_wait(c, want, &staged_all);
            if (f < 0) { rc = -1; break; }
            if (dhts_bgzf_index_staged(c) > 0 && dhts_bam_open(c) == 0) break;
            if (staged_all) { rc = -1; break; }
            want *= 4;
        }
    }
    if (rc != 0) { std::string m = std::string("Failed to open SAM/BAM/CRAM file: ") + bind->path; dhts_destroy(c); fail_with(m); return; }
    t_open = now_s() - t_start;
    dhts_bam_set_tag_columns(c, g->tag_ids.data(), (int32_t)g->tag_ids.size());
    dhts_bam_set_aux_map(c, g->want_aux ? 1 : 0, bind->standard_tags);
    if (!bind->region.empty()) {
        rc = dhts_bam_set_regions(c, bind->region.c_str());
        if (rc == 0 && !g->index_bytes.empty()) rc = dhts_bam_load_index(c, g->index_bytes.data(), g->index_bytes.size());
    } else rc = dhts_bam_set_regions(c, nullptr);
    if (rc == 0 && p->world > 1) rc = dhts_bam_set_file_shard(c, p->rank, p->world);
    else if (rc == 0 && bind->region.empty()) rc = dhts_bam_rewind(c);
    if (rc != 0) { std::string m = dhts_error(c); dhts_destroy(c); fail_with(m); return; }
    static const int64_t env_mb = getenv("DHTS_BATCH_BLOCKS") ? atoll(getenv("DHTS_BATCH_BLOCKS")) : 0;
    const int64_t max_blocks = env_mb > 0 ? env_mb : 4096;       // ~270 MB of inflated stream per batch: the engine gets its first chunk early and the stages overlap
    HostBatch *pending = nullptr; int pending_slot = 0, slot_no = 0;
    int64_t n_qual[3] = {0, 0, 0};                               // batches whose QUAL crossed PCIe as 2-bit codes / 4-bit codes / characters
    auto publish = [&](HostBatch *hb, int sl) -> bool {
        if (dhts_bam_batch_fetch_wait(c, sl) != 0) return false;
        build_qual_lut(hb);
        { std::lock_guard<std::mutex> lk(g->mu); p->ready.push_back(hb); }
        g->cv_ready.notify_all();
        return true;
    };
    for (;;) {
        dhts_bam_batch b;
        if (streaming && !staged_all && dhts_blocks_ahead(c) < max_blocks) {
            // not enough known blocks for a full batch: wait for (at least) another 128 MiB of the file, then extend the block table
            const double tw0 = now_s();
            int64_t f = dhts_stage_wait(c, 0, &staged_all);
            if (f >= 0 && !staged_all) f = dhts_stage_wait(c, (uint64_t)f + (128u << 20), &staged_all);
            const double tw1 = now_s(); t_wait += tw1 - tw0;
            if (f < 0 || dhts_bgzf_index_staged(c) < 0) { std::string m = dhts_error(c); dhts_destroy(c); fail_with(m); return; }
            t_index += now_s() - tw1; n_index++;
        }
        const double tb0 = now_s();
        if (dhts_bam_next_batch(c, max_blocks, g->colmask, &b) != 0) { std::string m = dhts_error(c); dhts_destroy(c); fail_with(m); return; }
        const double tb1 = now_s(); t_gpu += tb1 - tb0; n_batches++; n_rows += b.n_rows;
        if (b.n_rows > 0) {
            HostBatch *hb = nullptr;
            {
                std::unique_lock<std::mutex> lk(g->mu);
                g->cv_free.wait(lk, [&] { return g->cancel || !p->free_slots.empty(); });
                if (g->cancel) break;
                hb = p->free_slots.back(); p->free_slots.pop_back();
            }
            const double tb2 = now_s(); t_slot += tb2 - tb1;
            const uint64_t need = dhts_bam_batch_host_bytes(&b, g->colmask);
            if (need > hb->cap) { dhts_host_free(hb->arena); hb->arena = dhts_host_alloc(need); hb->cap = hb->arena ? need : 0; }
            // the read-back of this batch runs on a copy stream while the next batch is scanned: the batch is handed to the fill threads one
            // turn later, when its bytes have had a whole scan's time to cross PCIe (DHTS_OVERLAP_READBACK=0: copy, wait, hand over)
            static const bool env_serial = getenv("DHTS_OVERLAP_READBACK") && atoi(getenv("DHTS_OVERLAP_READBACK")) == 0;
            const int frc = env_serial ? dhts_bam_batch_fetch(c, &b, g->colmask, hb->arena, hb->cap, &hb->b) : dhts_bam_batch_fetch_begin(c, &b, g->colmask, hb->arena, hb->cap, &hb->b, slot_no);
            if ((need && !hb->arena) || frc != 0 || fetch_optional(c, g, b, hb) != 0) {
                std::string m = hb->arena || !need ? dhts_error(c) : "read_bam: out of pinned host memory"; dhts_destroy(c); fail_with(m); return;
            }
            hb->n = b.n_rows; hb->status = b.status; hb->next = 0; hb->readers = 0; hb->retired = false;
            if (g->colmask & (1u << DHTS_BAM_QUAL)) n_qual[hb->b.qual_bits == 2 ? 0 : hb->b.qual_bits == 4 ? 1 : 2]++;
            if (!p->has_rows) { p->has_rows = true; p->first_v = dhts_voffset(c, b.first_rec_uoff); }
            p->end_v = dhts_voffset(c, b.end_uoff);
            if (pending && !publish(pending, pending_slot)) { std::string m = dhts_error(c); dhts_destroy(c); fail_with(m); return; }
            pending = nullptr;
            if (env_serial) { build_qual_lut(hb); std::lock_guard<std::mutex> lk(g->mu); p->ready.push_back(hb); }
            else { pending = hb; pending_slot = slot_no; slot_no ^= 1; }
            if (env_serial) g->cv_ready.notify_all();
            t_fetch += now_s() - tb2;
        }
        if (b.status != 0) { p->clean_end = b.status == 1; break; }       // end of the stream, or the silent stop at the first bad block / record (bam_reader.c:754-766)
        { std::lock_guard<std::mutex> lk(g->mu); if (g->cancel) break; }
    }
    if (pending && !publish(pending, pending_slot)) { std::string m = dhts_error(c); dhts_destroy(c); fail_with(m); return; }
    dhts_destroy(c);
    if (trace) { uint64_t mc = 0, mb = 0; double ms = 0; dhts_debug_malloc_stats(&mc, &mb, &ms); fprintf(stderr, "[dhts] hipMalloc calls the pool could not serve so far in this process: %llu, %.2f GB, %.3f s\n", (unsigned long long)mc, 1e-9 * (double)mb, ms); }
    if (trace && (n_qual[0] + n_qual[1] + n_qual[2])) fprintf(stderr, "[dhts] producer %d QUAL over PCIe: %lld batches as 2-bit codes, %lld as 4-bit codes, %lld as characters (the batch's own alphabet: <= 4 / <= 16 / more distinct characters)\n",
                       p->rank, (long long)n_qual[0], (long long)n_qual[1], (long long)n_qual[2]);
    if (trace) fprintf(stderr, "[dhts] producer %d/%d dev %d: context %.4f s, staged at %.4f s%s, block table %.4f s, header %.4f s, open+index+header %.4f s, %lld batches %lld rows: device %.3f s, waiting for a free host slot %.3f s, read-back %.3f s, waiting for staged bytes %.3f s, %lld table extensions %.3f s, total %.3f s\n",
                       p->rank, p->world, p->device, t_created, t_staged, from_cache ? " (file still resident in HBM)" : "", t_idx, t_hdr, t_open, (long long)n_batches, (long long)n_rows, t_gpu, t_slot, t_fetch, t_wait, (long long)n_index, t_index, now_s() - t_start);
    { std::lock_guard<std::mutex> lk(g->mu); p->done = true; }
    g->cv_ready.notify_all();
}

static void bam_read_global_init(duckdb_init_info info) {
    BamBind *bind = (BamBind *)API(void *, duckdb_init_get_bind_data, duckdb_init_info)(info);
    auto init_error = API(void, duckdb_init_set_error, duckdb_init_info, const char *);
    BamScan *g = new BamScan();
    g->bind = bind;
    idx_t n = API(idx_t, duckdb_init_get_column_count, duckdb_init_info)(info);               // bam_reader.c:676-679
    for (idx_t i = 0; i < n; i++) {
        idx_t id = API(idx_t, duckdb_init_get_column_index, duckdb_init_info, idx_t)(info, i);
        g->column_ids.push_back(id);
        if (id < DHTS_BAM_CORE_COUNT) g->colmask |= 1u << id;
        int sl = -1;
        if (bind->standard_tags && id >= DHTS_BAM_CORE_COUNT && id < (idx_t)(DHTS_BAM_CORE_COUNT + dhts_bam_std_tag_count())) {
            const int32_t tid_ = (int32_t)(id - DHTS_BAM_CORE_COUNT);
            for (size_t k = 0; k < g->tag_ids.size(); k++) if (g->tag_ids[k] == tid_) sl = (int)k;
            if (sl < 0) { sl = (int)g->tag_ids.size(); g->tag_ids.push_back(tid_); }
        }
        g->tag_slot.push_back(sl);
        if (bind->auxiliary_tags && id == bind->aux_col_idx) g->want_aux = true;
    }
    if (!bind->region.empty()) {
        // bam_reader.c:639-668: a region needs an index; sam_itr_regarray failing reports "No reads found"
        if (!bind->has_index) { init_error(info, "Region query requires an index (.bai/.csi/.crai)"); delete g; return; }
        int rc = dhts_bam_set_regions(bind->ctx, bind->region.c_str());          // (validated on the bind context: it holds the header)
        if (rc != 0) {
            char err[640]; snprintf(err, sizeof(err), "No reads found for region(s): %s", bind->region.c_str());
            init_error(info, rc == 1 ? err : dhts_error(bind->ctx)); delete g; return;
        }
        // the index (BAI or CSI) narrows the scan window; the device predicate decides the rows
        FILE *f = fopen(bind->index_file.c_str(), "rb");
        if (f) {
            std::vector<uint8_t> ib; uint8_t tmp[65536]; size_t k;
            while ((k = fread(tmp, 1, sizeof(tmp), f)) > 0) ib.insert(ib.end(), tmp, tmp + k);
            fclose(f);
            const bool known = ib.size() >= 4 && (memcmp(ib.data(), "BAI\1", 4) == 0 || memcmp(ib.data(), "CSI\1", 4) == 0 || (ib[0] == 0x1f && ib[1] == 0x8b));
            if (known) g->index_bytes.swap(ib);
        }
        // only the index windows are staged (the reference seeks to them): byte ranges from the bind context, which holds the header
        static const bool env_nosparse = getenv("DHTS_SPARSE") && atoi(getenv("DHTS_SPARSE")) == 0;
        if (!g->index_bytes.empty() && !env_nosparse) {
            g->seg_beg.resize(4096); g->seg_end.resize(4096);
            if (dhts_bam_region_segments(bind->ctx, g->index_bytes.data(), g->index_bytes.size(), g->seg_beg.data(), g->seg_end.data(), 4096, &g->seg_count) != 0) g->seg_count = -1;   // fall back to the whole file
        }
    }
    // sequential mode unless the user asks for parallel fill (bam_reader.c:577-585: the reference goes parallel only with an index)
    int thr = getenv("DHTS_THREADS") ? atoi(getenv("DHTS_THREADS")) : 1; if (thr < 1) thr = 1; if (thr > 64) thr = 64;
    g->n_workers = thr;
    std::vector<int> devs = device_list();
    if (!bind->region.empty()) devs.resize(1);          // an index window is one short scan: a single device serves it
    for (size_t k = 0; k < devs.size(); k++) {
        Producer *p = new Producer(); p->device = devs[k]; p->rank = (int)k; p->world = (int)devs.size();
        for (int q = 0; q < 3; q++) { HostBatch *hb = new HostBatch(); p->free_slots.push_back(hb); p->all.push_back(hb); }
        g->prod.push_back(p);
    }
    for (auto p : g->prod) p->th = std::thread(producer_main, g, p);
    API(void, duckdb_init_set_max_threads, duckdb_init_info, idx_t)(info, (idx_t)thr);
    API(void, duckdb_init_set_init_data, duckdb_init_info, void *, duckdb_delete_callback_t)(info, g, destroy_global);
}

static void bam_read_local_init(duckdb_init_info info) {
    API(void, duckdb_init_set_init_data, duckdb_init_info, void *, duckdb_delete_callback_t)(info, new BamLocal(), destroy_local);
}

// hands the calling worker its next run of rows: the rest of the current batch (sequential mode) or a 2048-row slice of a ready batch.
// Returns false at the end of the scan (or on a producer error: g->error).
static bool next_rows(BamScan *g, BamLocal *l, idx_t want) {
    std::unique_lock<std::mutex> lk(g->mu);
    // give back what the worker holds
    if (l->cur) {
        HostBatch *hb = l->cur; Producer *own = l->cur_owner;
        hb->readers--;
        const bool finished = g->n_workers == 1 ? true : (hb->retired && hb->readers == 0);
        if (finished) { own->free_slots.push_back(hb); g->cv_free.notify_all(); }
        l->cur = nullptr;
    }
    for (;;) {
        if (!g->error.empty()) return false;
        // ordered mode drains the producers one after the other (file order); parallel mode takes whatever is ready
        for (size_t k = 0; k < g->prod.size(); k++) {
            Producer *p = g->prod[g->n_workers == 1 ? g->cur_prod : (g->cur_prod + k) % g->prod.size()];
            while (!p->ready.empty()) {
                HostBatch *hb = p->ready.front();
                if (g->n_workers == 1) {
                    p->ready.pop_front(); hb->readers = 1;
                    l->cur = hb; l->cur_owner = p; l->pos = 0; l->end = hb->n;
                    return true;
                }
                if (hb->next >= hb->n) {            // every row is claimed: the last reader returns the slot
                    p->ready.pop_front(); hb->retired = true;
                    if (hb->readers == 0) { p->free_slots.push_back(hb); g->cv_free.notify_all(); }
                    continue;
                }
                l->cur = hb; l->cur_owner = p; l->pos = hb->next; l->end = hb->next + (int64_t)want < hb->n ? hb->next + (int64_t)want : hb->n;
                hb->next = l->end; hb->readers++;
                return true;
            }
            if (g->n_workers == 1) {
                if (p->done && p->ready.empty()) {
                    if (!p->clean_end) return false;      // the stream ended on an error inside this rank: the scan ends here, silently (bam_reader.c:754-766)
                    if (g->cur_prod + 1 < g->prod.size()) { g->cur_prod++; k = (size_t)-1; continue; }
                }
                break;
            }
        }
        bool all_done = true;
        for (auto p : g->prod) if (!p->done || !p->ready.empty()) all_done = false;
        if (all_done) {
            // several GPUs on one file: every rank's last record must end exactly where the next rank's first record begins
            if (g->prod.size() > 1 && !g->handoff_checked) {
                g->handoff_checked = true;
                const Producer *prev = nullptr;
                if (g->n_workers > 1) for (auto p : g->prod) if (!p->clean_end && g->error.empty())
                    g->error = "read_bam: the stream ended on an error inside one GPU's block range; rerun with DHTS_THREADS=1 for the reference's rows-before-the-error result";
                for (auto p : g->prod) {
                    if (prev && p->has_rows && prev->clean_end && prev->end_v != p->first_v && g->error.empty()) {
                        char m[256]; snprintf(m, sizeof(m), "read_bam: GPU shard hand-off mismatch between ranks %d and %d (%llx vs %llx)", prev->rank, p->rank, (unsigned long long)prev->end_v, (unsigned long long)p->first_v);
                        g->error = m;
                    }
                    if (p->has_rows) prev = p;
                    if (!p->clean_end) break;                 // the stream ended on an error inside this rank: later ranks' rows are not reachable sequentially
                }
                if (!g->error.empty()) return false;
            }
            return false;
        }
        g->cv_ready.wait(lk);
    }
}

static void bam_read_function(duckdb_function_info info, duckdb_data_chunk output) {
    BamBind *bind = (BamBind *)API(void *, duckdb_function_get_bind_data, duckdb_function_info)(info);
    BamScan *g = (BamScan *)API(void *, duckdb_function_get_init_data, duckdb_function_info)(info);
    BamLocal *l = (BamLocal *)API(void *, duckdb_function_get_local_init_data, duckdb_function_info)(info);
    auto set_size = API(void, duckdb_data_chunk_set_size, duckdb_data_chunk, idx_t);
    if (!l || !g || l->done) { set_size(output, 0); return; }                                // bam_reader.c:730-733
    const idx_t vector_size = API(idx_t, duckdb_vector_size, void)();
    auto get_vec = API(duckdb_vector, duckdb_data_chunk_get_vector, duckdb_data_chunk, idx_t);
    auto get_data = API(void *, duckdb_vector_get_data, duckdb_vector);
    auto assign_len = API(void, duckdb_vector_assign_string_element_len, duckdb_vector, idx_t, const char *, idx_t);
    idx_t row_count = 0;
    while (row_count < vector_size) {
        if (!l->cur || l->pos >= l->end) {
            if (g->n_workers > 1 && row_count > 0) break;           // parallel mode: one slice per chunk
            if (!next_rows(g, l, vector_size)) {
                l->done = true;
                if (!g->error.empty()) { API(void, duckdb_function_set_error, duckdb_function_info, const char *)(info, g->error.c_str()); set_size(output, 0); return; }
                break;
            }
        }
        const HostBatch *hb = l->cur; const dhts_bam_batch &b = hb->b;
        idx_t take = (idx_t)(l->end - l->pos); if (take > vector_size - row_count) take = vector_size - row_count;
        const int64_t s = l->pos;
        for (size_t ci = 0; ci < g->column_ids.size(); ci++) {
            duckdb_vector vec = get_vec(output, ci);
            // strings of <= 12 bytes are written in place (no call, no heap); longer ones are copied into the vector's heap by the engine
            auto put_str = [&](const dhts_strcol &h) {
                duckdb_string_t *d = (duckdb_string_t *)get_data(vec) + row_count;
                for (idx_t r = 0; r < take; r++) { const char *p = (const char *)h.bytes + h.off[s + r]; const uint32_t n = h.len[s + r]; if (!inl_string(d + r, p, n)) assign_len(vec, row_count + r, p, n); } };
            auto put_name = [&](const int32_t *ids) {
                duckdb_string_t *d = (duckdb_string_t *)get_data(vec) + row_count;
                for (idx_t r = 0; r < take; r++) {
                    const int32_t t = ids[s + r];
                    if (t < 0) d[r] = bind->star_inl; else if (bind->ref_is_inl[t]) d[r] = bind->ref_inl[t];
                    else { const char *nm = bind->hdr.ref_name[t]; assign_len(vec, row_count + r, nm, strlen(nm)); }
                } };
            switch (g->column_ids[ci]) {
            case DHTS_BAM_QNAME: put_str(b.qname); break;
            case DHTS_BAM_FLAG: memcpy((uint16_t *)get_data(vec) + row_count, b.flag + s, take * 2); break;
            case DHTS_BAM_RNAME: put_name(b.tid); break;
            case DHTS_BAM_POS: memcpy((int64_t *)get_data(vec) + row_count, b.pos + s, take * 8); break;
            case DHTS_BAM_MAPQ: memcpy((int32_t *)get_data(vec) + row_count, b.mapq + s, take * 4); break;
            case DHTS_BAM_CIGAR: put_str(b.cigar); break;
            case DHTS_BAM_RNEXT: put_name(b.mtid); break;
            case DHTS_BAM_PNEXT: memcpy((int64_t *)get_data(vec) + row_count, b.pnext + s, take * 8); break;
            case DHTS_BAM_TLEN: memcpy((int64_t *)get_data(vec) + row_count, b.tlen + s, take * 8); break;
            case DHTS_BAM_SEQ:
                if (!b.seq_packed) { put_str(b.seq); break; }
                {   // the batch carries the file's 4-bit codes: expand here (seq_to_string, bam_reader.c:560-575; "*" for an empty SEQ)
                    duckdb_string_t *d = (duckdb_string_t *)get_data(vec) + row_count;
                    for (idx_t r = 0; r < take; r++) {
                        const uint32_t n = b.seq.len[s + r];
                        if (n == 0) { inl_string(d + r, "*", 1); continue; }
                        if (l->seq_tmp.size() < (size_t)n + 32) l->seq_tmp.resize((size_t)n + 32 + n / 2);
                        expand_seq(b.seq.bytes + b.seq.off[s + r], n, l->seq_tmp.data());
                        if (!inl_string(d + r, l->seq_tmp.data(), n)) assign_len(vec, row_count + r, l->seq_tmp.data(), n);
                    }
                }
                break;
            case DHTS_BAM_QUAL:
                if (!b.qual_bits) { put_str(b.qual); break; }
                {   // the batch carries codes of its own alphabet: expand here (qual_to_string's characters, bam_reader.c:577-600, were made on the device)
                    duckdb_string_t *d = (duckdb_string_t *)get_data(vec) + row_count;
                    const uint8_t *stream = b.qual.bytes + 16; const uint32_t *lut = hb->qual_lut.data();
                    for (idx_t r = 0; r < take; r++) {
                        const uint32_t n = b.qual.len[s + r];
                        if (l->seq_tmp.size() < (size_t)n + 40) l->seq_tmp.resize((size_t)n + 40 + n / 2);
                        expand_qual(stream, b.qual_bits, lut, b.qual.off[s + r], n, l->seq_tmp.data(), l->qual_tmp);
                        if (!inl_string(d + r, l->seq_tmp.data(), n)) assign_len(vec, row_count + r, l->seq_tmp.data(), n);
                    }
                }
                break;
            case DHTS_BAM_READ_GROUP_ID: {
                duckdb_string_t *d = (duckdb_string_t *)get_data(vec) + row_count;
                for (idx_t r = 0; r < take; r++) {
                    int64_t q = s + (int64_t)r;
                    if ((b.rg_valid[q >> 6] >> (q & 63)) & 1) { const char *p = (const char *)b.rg.bytes + b.rg.off[q]; const uint32_t n = b.rg.len[q]; if (!inl_string(d + r, p, n)) assign_len(vec, row_count + r, p, n); }
                    else set_null(vec, row_count + r);
                }
                break;
            }
            case DHTS_BAM_SAMPLE_ID:
                for (idx_t r = 0; r < take; r++) {
                    int64_t q = s + (int64_t)r; int32_t k = b.rg_idx[q];
                    const char *sm = (((b.rg_valid[q >> 6] >> (q & 63)) & 1) && k >= 0) ? bind->hdr.rg_sm[k] : nullptr;
                    if (sm) assign_len(vec, row_count + r, sm, strlen(sm)); else set_null(vec, row_count + r);
                }
                break;
            default: {
                if (g->want_aux && g->column_ids[ci] == bind->aux_col_idx) {               // bam_reader.c:967-1027
                    auto list_size = API(idx_t, duckdb_list_vector_get_size, duckdb_vector);
                    duckdb_list_entry *le = (duckdb_list_entry *)get_data(vec);
                    idx_t base = list_size(vec);
                    const uint32_t c0 = hb->aux_off[s], c1 = hb->aux_off[s + take];
                    if (c1 > c0) { API(duckdb_state, duckdb_list_vector_reserve, duckdb_vector, idx_t)(vec, base + (c1 - c0)); API(duckdb_state, duckdb_list_vector_set_size, duckdb_vector, idx_t)(vec, base + (c1 - c0)); }
                    duckdb_vector child = API(duckdb_vector, duckdb_list_vector_get_child, duckdb_vector)(vec);
                    duckdb_vector kvec = API(duckdb_vector, duckdb_struct_vector_get_child, duckdb_vector, idx_t)(child, 0);
                    duckdb_vector vvec = API(duckdb_vector, duckdb_struct_vector_get_child, duckdb_vector, idx_t)(child, 1);
                    for (idx_t r = 0; r < take; r++) {
                        le[row_count + r].offset = base + (hb->aux_off[s + r] - c0); le[row_count + r].length = hb->aux_off[s + r + 1] - hb->aux_off[s + r];
                        if (!hb->aux_valid[s + r]) set_null(vec, row_count + r);            // no tags: NULL, entry {size, 0}
                    }
                    for (uint32_t k = c0; k < c1; k++) {
                        assign_len(kvec, base + (k - c0), hb->aux_key[k].data(), hb->aux_key[k].size());
                        assign_len(vvec, base + (k - c0), hb->aux_val[k].data(), hb->aux_val[k].size());
                    }
                    break;
                }
                const int sl = g->tag_slot[ci];
                if (sl < 0) break;                                 // unknown ids (e.g. a row-id pseudo column) write nothing, like the reference's default arm
                const HostTag &h = hb->tags[sl];
                char nm[3], ty, sub; dhts_bam_std_tag_info(g->tag_ids[sl], nm, &ty, &sub);
                if (ty == 'i') {                                    // bam_reader.c:946-950
                    memcpy((int64_t *)get_data(vec) + row_count, h.fixed.data() + s, take * 8);
                    for (idx_t r = 0; r < take; r++) if (!h.valid[s + r]) set_null(vec, row_count + r);
                } else if (ty == 'B') {                             // bam_assign_list_int / _double bam_reader.c:106-138
                    auto list_size = API(idx_t, duckdb_list_vector_get_size, duckdb_vector);
                    duckdb_list_entry *le = (duckdb_list_entry *)get_data(vec);
                    idx_t base = list_size(vec);
                    const uint32_t c0 = h.off[s], c1 = h.off[s + take];
                    if (c1 > c0) { API(duckdb_state, duckdb_list_vector_reserve, duckdb_vector, idx_t)(vec, base + (c1 - c0)); API(duckdb_state, duckdb_list_vector_set_size, duckdb_vector, idx_t)(vec, base + (c1 - c0)); }
                    duckdb_vector child = API(duckdb_vector, duckdb_list_vector_get_child, duckdb_vector)(vec);
                    for (idx_t r = 0; r < take; r++) {
                        if (h.valid[s + r]) { le[row_count + r].offset = base + (h.off[s + r] - c0); le[row_count + r].length = h.off[s + r + 1] - h.off[s + r]; }
                        else set_null(vec, row_count + r);          // absent tag: set_null only, the entry is left untouched (bam_reader.c:927-930)
                    }
                    if (c1 > c0) memcpy((int64_t *)get_data(child) + base, h.child.data() + c0, (size_t)(c1 - c0) * 8);
                } else {
                    for (idx_t r = 0; r < take; r++) {
                        if (h.valid[s + r]) assign_len(vec, row_count + r, (const char *)h.bytes.data() + h.off[s + r], h.off[s + r + 1] - h.off[s + r]);
                        else set_null(vec, row_count + r);
                    }
                }
                break;
            }
            }
        }
        row_count += take; l->pos += (int64_t)take;
    }
    set_size(output, row_count);
}

extern "C" __attribute__((visibility("default"))) void register_read_bam_function(duckdb_connection connection) {                      // bam_reader.c:1044-1068
    duckdb_table_function tf = API(duckdb_table_function, duckdb_create_table_function, void)();
    API(void, duckdb_table_function_set_name, duckdb_table_function, const char *)(tf, "read_bam");
    auto mk = API(duckdb_logical_type, duckdb_create_logical_type, int);
    auto rm = API(void, duckdb_destroy_logical_type, duckdb_logical_type *);
    auto named = API(void, duckdb_table_function_add_named_parameter, duckdb_table_function, const char *, duckdb_logical_type);
    duckdb_logical_type t_varchar = mk(DUCKDB_TYPE_VARCHAR);
    API(void, duckdb_table_function_add_parameter, duckdb_table_function, duckdb_logical_type)(tf, t_varchar);
    named(tf, "region", t_varchar); named(tf, "index_path", t_varchar); named(tf, "reference", t_varchar);
    rm(&t_varchar);
    duckdb_logical_type t_bool = mk(DUCKDB_TYPE_BOOLEAN);
    named(tf, "standard_tags", t_bool); named(tf, "auxiliary_tags", t_bool);
    rm(&t_bool);
    API(void, duckdb_table_function_set_bind, duckdb_table_function, duckdb_table_function_bind_t)(tf, bam_read_bind);
    API(void, duckdb_table_function_set_init, duckdb_table_function, duckdb_table_function_init_t)(tf, bam_read_global_init);
    API(void, duckdb_table_function_set_local_init, duckdb_table_function, duckdb_table_function_init_t)(tf, bam_read_local_init);
    API(void, duckdb_table_function_set_function, duckdb_table_function, duckdb_table_function_t)(tf, bam_read_function);
    API(void, duckdb_table_function_supports_projection_pushdown, duckdb_table_function, bool)(tf, true);
    API(duckdb_state, duckdb_register_table_function, duckdb_connection, duckdb_table_function)(connection, tf);
    API(void, duckdb_destroy_table_function, duckdb_table_function *)(&tf);
}


// =====================================================================================================================
// read_bcf -- mirrors register_read_bcf_function src/bcf_reader.c:2055-2080, bcf_read_bind 452-880 (schema 540-760),
// global/local init 886-1150 (projection ids, region error), bcf_read_function 1155-2049 (<= vector_size rows per call).
// Sequential mode; tidy_format and region supported; BCF, and VCF text (plain or bgzipped) through the device text encoder.
// =====================================================================================================================
struct BcfBind {
    std::string path, region;
    std::vector<std::string> regions;    // comma split, empty tokens dropped (parse_regions_duckdb, bcf_reader.c:423-446)
    std::string index_file; std::vector<uint8_t> index_bytes;
    uint64_t header_bytes = 0; std::vector<uint64_t> seg_beg, seg_end; int64_t seg_count = -1;     // region query: header blocks + index windows are all that is staged
    dhts_ctx *ctx = nullptr;             // bind-time context: holds only the head of the file (header, dictionaries, schema)
    dhts_bcf_info inf;                   // schema of the bind context (column names / types live there)
    int has_index = 0, tidy = 0, device = 0;
};
// ---- scan pipeline, as for read_bam: a producer thread drives the device and reads every batch back into a pinned arena (four queued
// copies, dhts_bcf_batch_fetch); the scan callbacks fill DataChunks from those arenas while the device works on the next batch.
// DHTS_THREADS = 1 (default): one worker, rows in file order, full 2048-row chunks (the reference's only mode for read_bcf without
// an index); DHTS_THREADS = k: k workers claim 2048-row slices, row order across workers unspecified.
struct BcfHostBatch {
    void *arena = nullptr; uint64_t cap = 0;
    std::vector<dhts_bcf_col> cols;                 // HOST pointers, projection order (deduplicated)
    std::vector<std::vector<uint32_t>> conv;        // per column: DHTS_ENC_FLOAT_TEXT children converted to float bits
    int64_t n = 0; int status = 0; int ncols_fetched = 0;
    int64_t next = 0; int readers = 0; bool retired = false;
};
struct BcfScan {
    BcfBind *bind = nullptr;
    std::vector<idx_t> column_ids;       // schema ids per output vector
    std::vector<int> slot;               // output vector -> index into the batch's columns (or -1 for unknown ids)
    std::vector<int32_t> proj;           // projected (deduplicated) schema columns
    int n_workers = 1;
    std::mutex mu; std::condition_variable cv_ready, cv_free;
    std::deque<BcfHostBatch *> ready; std::vector<BcfHostBatch *> free_slots, all;
    std::thread th; bool done = false, cancel = false; std::string error;
    dhts_ctx *ctx = nullptr;             // the scan's own context (the producer stages the file into it); lives until the chunks are filled:
    dhts_bcf_info inf;                   // its name tables -- a text scan adds the names records use without a header definition
    ~BcfScan() {
        { std::lock_guard<std::mutex> lk(mu); cancel = true; }
        cv_free.notify_all(); cv_ready.notify_all();
        if (th.joinable()) th.join();
        if (ctx) dhts_destroy(ctx);                 // first: it waits for the copy stream, whose D2H may still be writing into an arena (error paths leave one in flight)
        for (auto hb : all) { dhts_host_free(hb->arena); delete hb; }
    }
};
struct BcfLocal {
    bool done = false;
    BcfHostBatch *cur = nullptr; int64_t pos = 0, end = 0;      // rows [pos, end) of `cur` are this worker's
};
static void destroy_bcf_bind(void *p) { BcfBind *b = (BcfBind *)p; if (!b) return; if (b->ctx) dhts_destroy(b->ctx); delete b; }
static void destroy_bcf_local(void *p) { delete (BcfLocal *)p; }
static void destroy_bcf_global(void *p) { delete (BcfScan *)p; }

static void bcf_read_bind(duckdb_bind_info info) {
    auto set_error = API(void, duckdb_bind_set_error, duckdb_bind_info, const char *);
    auto dfree = API(void, duckdb_free, void *);
    duckdb_value pv = API(duckdb_value, duckdb_bind_get_parameter, duckdb_bind_info, idx_t)(info, 0);
    char *file_path = API(char *, duckdb_get_varchar, duckdb_value)(pv);
    API(void, duckdb_destroy_value, duckdb_value *)(&pv);
    if (!file_path || strlen(file_path) == 0) {
        set_error(info, "read_bcf requires a file path");                          // bcf_reader.c:461
        if (file_path) dfree(file_path);
        return;
    }
    char *region = get_named_varchar(info, "region");
    char *index_path = get_named_varchar(info, "index_path");
    const int tidy = get_named_bool(info, "tidy_format");
    BcfBind *b = new BcfBind();
    b->path = file_path; if (region) b->region = region;
    for (size_t p0 = 0; p0 <= b->region.size() && !b->region.empty();) {
        size_t q = b->region.find(',', p0); if (q == std::string::npos) q = b->region.size();
        if (q > p0) b->regions.push_back(b->region.substr(p0, q - p0));
        p0 = q + 1;
    }
    std::string idx = index_path ? index_path : "";
    dfree(file_path); if (region) dfree(region); if (index_path) dfree(index_path);
    char err[768];
    if (!file_exists(b->path)) {
        snprintf(err, sizeof(err), "Failed to open BCF/VCF file: %s", b->path.c_str());       // bcf_reader.c:494
        set_error(info, err); delete b; return;
    }
    int dev = getenv("DHTS_DEVICE") ? atoi(getenv("DHTS_DEVICE")) : 0;
    static const bool trace_bcf_bind = getenv("DHTS_TRACE") != nullptr;
    const double tb0 = now_s();
    b->ctx = dhts_create(dev); b->tidy = tidy; b->device = dev;
    const double tb1 = now_s();
    if (!b->ctx) { set_error(info, "read_bcf: no MI355X (gfx950) device available; this build has no CPU fallback"); destroy_bcf_bind(b); return; }
    // like the reference, bind reads the header only (bcf_open + bcf_hdr_read, bcf_reader.c:480-505): the head of the file is staged, four
    // times more whenever the header turns out to be longer; every scan stages the file in its own context (bcf_read_global_init)
    bool hdr_ok = false;
    for (uint64_t head = 1u << 20;; head *= 4) {
        if (dhts_open_path_range(b->ctx, b->path.c_str(), 0, head) != 0) {
            snprintf(err, sizeof(err), "Failed to open BCF/VCF file: %s", b->path.c_str());
            set_error(info, err); destroy_bcf_bind(b); return;
        }
        const bool whole = dhts_resident_bytes(b->ctx) < head;
        if (dhts_bgzf_index(b->ctx) > 0 && dhts_bcf_open(b->ctx, tidy) == 0 && dhts_bcf_info_get(b->ctx, &b->inf) == 0) { hdr_ok = true; break; }
        if (whole || head >= (1ull << 34)) break;
        const char *m0 = dhts_error(b->ctx);
        if (m0 && strncmp(m0, "read_bcf:", 9) == 0) break;                   // a refusal, not a header that is merely longer than the head
    }
    if (!hdr_ok) {
        const char *m = dhts_error(b->ctx);
        set_error(info, (m && strncmp(m, "read_bcf:", 9) == 0) ? m : "Failed to read BCF/VCF header");       // bcf_reader.c:505 (or what this build does not read yet)
        destroy_bcf_bind(b); return;
    }
    const double tb2 = now_s();
    for (const std::string &f : {idx, b->path + ".csi", b->path + ".tbi"}) if (!f.empty() && file_exists(f)) { b->index_file = f; break; }
    b->has_index = !b->index_file.empty();
    if (b->has_index && !b->regions.empty()) {
        FILE *f = fopen(b->index_file.c_str(), "rb");
        if (f) { uint8_t tmp[65536]; size_t k; while ((k = fread(tmp, 1, sizeof(tmp), f)) > 0) b->index_bytes.insert(b->index_bytes.end(), tmp, tmp + k); fclose(f); }
    }
    {
        // only the header blocks and the index windows of the regions are staged (the reference seeks to the chunks): byte ranges from this
        // context, which holds the header.  DHTS_SPARSE=0 stages the whole file.
        static const bool env_nosparse = getenv("DHTS_SPARSE") && atoi(getenv("DHTS_SPARSE")) == 0;
        if (!b->index_bytes.empty() && !b->regions.empty() && !env_nosparse) {
            b->header_bytes = dhts_bcf_header_bytes(b->ctx);
            b->seg_beg.resize(4096); b->seg_end.resize(4096);
            if (b->header_bytes == 0 || dhts_bcf_region_segments(b->ctx, b->region.c_str(), b->index_bytes.data(), b->index_bytes.size(), b->seg_beg.data(), b->seg_end.data(), 4096, &b->seg_count) != 0) b->seg_count = -1;
        }
    }
    if (trace_bcf_bind) fprintf(stderr, "[dhts] read_bcf bind: context %.4f s, head of the file + block table + header %.4f s, index file + windows %.4f s (%lld byte ranges)\n", tb1 - tb0, tb2 - tb1, now_s() - tb2, (long long)b->seg_count);
    auto mk = API(duckdb_logical_type, duckdb_create_logical_type, int);
    auto mklist = API(duckdb_logical_type, duckdb_create_list_type, duckdb_logical_type);
    auto add = API(void, duckdb_bind_add_result_column, duckdb_bind_info, const char *, duckdb_logical_type);
    auto rm = API(void, duckdb_destroy_logical_type, duckdb_logical_type *);
    for (int i = 0; i < b->inf.n_cols; i++) {                                               // create_bcf_field_type bcf_reader.c:388-418
        const dhts_bcf_colinfo &ci = b->inf.cols[i];
        duckdb_logical_type el = mk(ci.type);
        if (ci.is_list) { duckdb_logical_type lt = mklist(el); add(info, ci.name, lt); rm(&lt); }
        else add(info, ci.name, el);
        rm(&el);
    }
    API(void, duckdb_bind_set_bind_data, duckdb_bind_info, void *, duckdb_delete_callback_t)(info, b, destroy_bcf_bind);
}

// chained single-region iterators (bcf_reader.c:1327-1345): the next region that yields an iterator; false when none is left
static bool bcf_next_region(BcfBind *bind, dhts_ctx *c, size_t *next_region) {
    while (*next_region < bind->regions.size()) {
        const std::string &rg = bind->regions[(*next_region)++];
        if (dhts_bcf_set_region(c, rg.c_str()) == 0) {                          // unknown contig / malformed: skipped (bcf_reader.c:944-953)
            // BCF: the index only narrows the window, a failure keeps the full scan.  VCF text: the region names a sequence of the tabix
            // index (tbx_itr_querys), 1 = the index does not know it; a failure surfaces with the first batch
            if (!bind->index_bytes.empty() && dhts_bcf_load_index(c, bind->index_bytes.data(), bind->index_bytes.size()) == 1) continue;
            return true;
        }
    }
    return false;
}

static void bcf_producer_main(BcfScan *g) {
    BcfBind *bind = g->bind;
    auto finish = [&](const std::string &err) {
        std::lock_guard<std::mutex> lk(g->mu);
        if (!err.empty() && g->error.empty()) g->error = err;
        g->done = true; g->cv_ready.notify_all();
    };
    static const bool trace = getenv("DHTS_TRACE") != nullptr;       // stage timings on stderr
    const double t_start = now_s();
    (void)dhts_bind_thread_near_device(bind->device);          // (as read_bam's producers: thread, staging readers and pinned arenas on the GPU's NUMA node)
    dhts_ctx *c = g->ctx = dhts_create(bind->device);
    if (!c) { finish("read_bcf: no MI355X (gfx950) device available; this build has no CPU fallback"); return; }
    dhts_set_super_blocks(c, 196608);
    const double t_ctx = now_s() - t_start;
    // a plain whole-file scan starts decoding while the file is still being staged (as read_bam does): the block table is built over the
    // resident prefix and extended as more bytes arrive.  DHTS_STREAM=0, region queries and uncompressed text stage first.
    static const bool env_nostream = getenv("DHTS_STREAM") && atoi(getenv("DHTS_STREAM")) == 0;
    const bool streaming = bind->regions.empty() && !env_nostream && bind->seg_count < 0 && dhts_bcf_is_text(bind->ctx) != 2;
    int staged_all = 1;
    int orc = bind->seg_count >= 0 ? dhts_open_path_segments(c, bind->path.c_str(), bind->header_bytes, bind->seg_beg.data(), bind->seg_end.data(), bind->seg_count)
              : streaming ? dhts_open_path_async(c, bind->path.c_str()) : dhts_open_path(c, bind->path.c_str());
    if (orc == 0 && !streaming && (dhts_bgzf_index(c) <= 0 || dhts_bcf_open(c, bind->tidy) != 0)) orc = -1;
    if (orc == 0 && streaming) {
        // the header needs the first blocks only: 32 MiB to start with, four times more whenever that is not enough
        uint64_t want = 32u << 20;
        for (;;) {
            const int64_t f = dhts_stage_wait(c, want, &staged_all);
            if (f < 0) { orc = -1; break; }
            if (dhts_bgzf_index_staged(c) > 0 && dhts_bcf_open(c, bind->tidy) == 0) break;
            if (staged_all) { orc = -1; break; }
            want *= 4;
        }
    }
    if (orc != 0 || dhts_bcf_info_get(c, &g->inf) != 0) {
        finish(std::string("Failed to open BCF/VCF file: ") + bind->path); return;
    }
    const double t_open = now_s() - t_start;
    if (dhts_bcf_set_projection(c, g->proj.data(), (int32_t)g->proj.size()) != 0 || dhts_bcf_set_region(c, nullptr) != 0) { finish("Failed to open BCF/VCF file"); return; }
    size_t next_region = 0;
    if (!bind->regions.empty() && !bcf_next_region(bind, c, &next_region)) { finish(""); return; }     // no region produced an iterator: zero rows (bcf_reader.c:955-959)
    const double t_region = now_s() - t_start;
    if (trace) fprintf(stderr, "[dhts] read_bcf producer dev %d: context %.4f s, staged + block table + header at %.4f s (%s, %llu bytes resident), first region set at %.4f s\n", bind->device, t_ctx, t_open,
                       bind->seg_count >= 0 ? "header + index windows" : streaming ? "streaming" : "whole file", (unsigned long long)dhts_resident_bytes(c), t_region);
    static const int64_t env_mb = getenv("DHTS_BATCH_BLOCKS") ? atoll(getenv("DHTS_BATCH_BLOCKS")) : 0;
    const int64_t max_blocks = env_mb > 0 ? env_mb : 4096;
    BcfHostBatch *pending = nullptr; int pending_slot = 0, slot_no = 0;
    auto publish = [&](BcfHostBatch *hb, int sl) -> bool {          // sl < 0: the bytes are already there
        if (sl >= 0 && dhts_bcf_batch_fetch_wait(c, sl) != 0) return false;
        hb->conv.assign((size_t)hb->ncols_fetched, std::vector<uint32_t>());
        for (int i = 0; i < hb->ncols_fetched; i++) {
            const dhts_bcf_col &h = hb->cols[i];
            if (bind->inf.cols[h.col].encoding != DHTS_ENC_FLOAT_TEXT) continue;
            // Float fields of a transcript arrive as text: (float)strtod, NaN unless the whole token converts (vep_parse_float, src/vep_parser.c:222-235)
            std::vector<uint32_t> &cv = hb->conv[i]; cv.assign(h.child_n + 1, 0);
            std::string tok;
            for (uint64_t k = 0; k < h.child_n; k++) {
                if (h.child_valid && !h.child_valid[k]) continue;
                tok.assign((const char *)h.bytes + h.child_off[k], h.child_off[k + 1] - h.child_off[k]);
                char *end = nullptr; const double v = strtod(tok.c_str(), &end);
                const float f = (end == tok.c_str() || *end) ? NAN : (float)v;
                memcpy(&cv[k], &f, 4);
            }
        }
        { std::lock_guard<std::mutex> lk(g->mu); g->ready.push_back(hb); }
        g->cv_ready.notify_all();
        return true;
    };
    for (;;) {
        dhts_bcf_batch b;
        if (streaming && !staged_all && dhts_blocks_ahead(c) < max_blocks) {
            // not enough known blocks for a full batch: wait for (at least) another 128 MiB of the file, then extend the block table
            int64_t f = dhts_stage_wait(c, 0, &staged_all);
            if (f >= 0 && !staged_all) f = dhts_stage_wait(c, (uint64_t)f + (128u << 20), &staged_all);
            if (f < 0 || dhts_bgzf_index_staged(c) < 0) { finish(dhts_error(c)); return; }
        }
        if (dhts_bcf_next_batch(c, max_blocks, &b) != 0) { finish(dhts_error(c)); return; }
        if (b.n_rows > 0) {
            BcfHostBatch *hb = nullptr;
            {
                std::unique_lock<std::mutex> lk(g->mu);
                g->cv_free.wait(lk, [&] { return g->cancel || !g->free_slots.empty(); });
                if (g->cancel) break;
                hb = g->free_slots.back(); g->free_slots.pop_back();
            }
            const uint64_t need = dhts_bcf_batch_host_bytes(c);
            if (need > hb->cap) { dhts_host_free(hb->arena); hb->arena = dhts_host_alloc(need); hb->cap = hb->arena ? need : 0; }
            hb->cols.assign((size_t)b.n_cols, dhts_bcf_col());
            static const bool env_serial = getenv("DHTS_OVERLAP_READBACK") && atoi(getenv("DHTS_OVERLAP_READBACK")) == 0;
            const int frc = env_serial ? dhts_bcf_batch_fetch(c, &b, hb->arena, hb->cap, hb->cols.data()) : dhts_bcf_batch_fetch_begin(c, &b, hb->arena, hb->cap, hb->cols.data(), slot_no);
            if ((need && !hb->arena) || frc != 0) { finish(hb->arena || !need ? dhts_error(c) : "read_bcf: out of pinned host memory"); return; }
            hb->n = b.n_rows; hb->status = b.status; hb->next = 0; hb->readers = 0; hb->retired = false; hb->ncols_fetched = b.n_cols;
            // the previous batch has had this batch's scan to cross PCIe: finish it (text floats) and hand it to the fill threads
            if (pending && !publish(pending, pending_slot)) { finish(dhts_error(c)); return; }
            pending = nullptr;
            if (env_serial) { if (!publish(hb, -1)) { finish(dhts_error(c)); return; } }
            else { pending = hb; pending_slot = slot_no; slot_no ^= 1; }
        }
        if (b.status != 0) {                                     // EOF, or the silent stop at the first bad record (bcf_reader.c:1319-1349)
            if (!bind->regions.empty() && bcf_next_region(bind, c, &next_region)) continue;
            break;
        }
        { std::lock_guard<std::mutex> lk(g->mu); if (g->cancel) break; }
    }
    if (pending && !publish(pending, pending_slot)) { finish(dhts_error(c)); return; }
    finish("");
}
static void bcf_read_global_init(duckdb_init_info info) {
    BcfBind *bind = (BcfBind *)API(void *, duckdb_init_get_bind_data, duckdb_init_info)(info);
    if (!bind->regions.empty() && !bind->has_index) {
        char err[900];
        snprintf(err, sizeof(err), "Region query requires an index file (.tbi or .csi). Region: %s", bind->region.c_str());   // bcf_reader.c:922-923
        API(void, duckdb_init_set_error, duckdb_init_info, const char *)(info, err);
        return;
    }
    BcfScan *g = new BcfScan();
    g->bind = bind;
    idx_t n = API(idx_t, duckdb_init_get_column_count, duckdb_init_info)(info);
    for (idx_t i = 0; i < n; i++) {
        idx_t id = API(idx_t, duckdb_init_get_column_index, duckdb_init_info, idx_t)(info, i);
        g->column_ids.push_back(id);
        int sl = -1;
        if (id < (idx_t)bind->inf.n_cols) {
            for (size_t k = 0; k < g->proj.size(); k++) if (g->proj[k] == (int32_t)id) sl = (int)k;
            if (sl < 0) { sl = (int)g->proj.size(); g->proj.push_back((int32_t)id); }
        }
        g->slot.push_back(sl);
    }
    int thr = getenv("DHTS_THREADS") ? atoi(getenv("DHTS_THREADS")) : 1; if (thr < 1) thr = 1; if (thr > 64) thr = 64;
    g->n_workers = thr;
    for (int q = 0; q < 3; q++) { BcfHostBatch *hb = new BcfHostBatch(); g->free_slots.push_back(hb); g->all.push_back(hb); }
    g->th = std::thread(bcf_producer_main, g);
    API(void, duckdb_init_set_max_threads, duckdb_init_info, idx_t)(info, (idx_t)thr);
    API(void, duckdb_init_set_init_data, duckdb_init_info, void *, duckdb_delete_callback_t)(info, g, destroy_bcf_global);
}

static void bcf_read_local_init(duckdb_init_info info) {
    API(void, duckdb_init_set_init_data, duckdb_init_info, void *, duckdb_delete_callback_t)(info, new BcfLocal(), destroy_bcf_local);
}

// rows of the next ready batch for this worker (ordered mode: the whole batch; parallel mode: a slice); false = the scan is over
static bool bcf_next_rows(BcfScan *g, BcfLocal *l, idx_t want) {
    std::unique_lock<std::mutex> lk(g->mu);
    if (l->cur) {
        BcfHostBatch *hb = l->cur;
        hb->readers--;
        if (g->n_workers == 1 || (hb->retired && hb->readers == 0)) { g->free_slots.push_back(hb); g->cv_free.notify_all(); }
        l->cur = nullptr;
    }
    for (;;) {
        if (!g->error.empty()) return false;
        while (!g->ready.empty()) {
            BcfHostBatch *hb = g->ready.front();
            if (g->n_workers == 1) { g->ready.pop_front(); hb->readers = 1; l->cur = hb; l->pos = 0; l->end = hb->n; return true; }
            if (hb->next >= hb->n) {
                g->ready.pop_front(); hb->retired = true;
                if (hb->readers == 0) { g->free_slots.push_back(hb); g->cv_free.notify_all(); }
                continue;
            }
            l->cur = hb; l->pos = hb->next; l->end = hb->next + (int64_t)want < hb->n ? hb->next + (int64_t)want : hb->n;
            hb->next = l->end; hb->readers++;
            return true;
        }
        if (g->done) return false;
        g->cv_ready.wait(lk);
    }
}

static size_t bcf_fixed_width(const dhts_bcf_colinfo &ci) {
    if (ci.is_list) return 0;
    if (ci.encoding != DHTS_ENC_PLAIN) return 4;
    switch (ci.type) { case DHTS_T_BOOLEAN: return 1; case DHTS_T_INTEGER: case DHTS_T_FLOAT: return 4; case DHTS_T_BIGINT: case DHTS_T_DOUBLE: return 8; default: return 0; }
}

// rows [s, s + take) of a host batch -> rows [row_count, row_count + take) of the output chunk
static void bcf_fill(const BcfBind *bind, const BcfScan *g, const BcfHostBatch *hb, int64_t s, idx_t take, duckdb_data_chunk output, idx_t row_count) {
    auto get_vec = API(duckdb_vector, duckdb_data_chunk_get_vector, duckdb_data_chunk, idx_t);
    auto get_data = API(void *, duckdb_vector_get_data, duckdb_vector);
    auto assign_len = API(void, duckdb_vector_assign_string_element_len, duckdb_vector, idx_t, const char *, idx_t);
    auto list_size = API(idx_t, duckdb_list_vector_get_size, duckdb_vector);
    auto list_reserve = API(duckdb_state, duckdb_list_vector_reserve, duckdb_vector, idx_t);
    auto list_set_size = API(duckdb_state, duckdb_list_vector_set_size, duckdb_vector, idx_t);
    auto list_child = API(duckdb_vector, duckdb_list_vector_get_child, duckdb_vector);
    for (size_t ci = 0; ci < g->column_ids.size(); ci++) {
        if (g->slot[ci] < 0) continue;                          // ids outside the schema write nothing
        const dhts_bcf_col &h = hb->cols[g->slot[ci]];
        const dhts_bcf_colinfo &inf = bind->inf.cols[h.col];
        duckdb_vector vec = get_vec(output, ci);
        const char *const *names = inf.encoding == DHTS_ENC_CONTIG ? g->inf.contig_name : inf.encoding == DHTS_ENC_DICT ? g->inf.dict_name :
                                   inf.encoding == DHTS_ENC_SAMPLE ? g->inf.sample_name : nullptr;      // (the SCAN's tables: a text scan may have added names)
        auto name_of = [&](int32_t id) -> const char * { if (id < 0) return "PASS"; const char *nm = names[id]; return nm ? nm : "."; };
        if (!inf.is_list) {
            const size_t w = bcf_fixed_width(inf);
            if (names) {
                for (idx_t r = 0; r < take; r++) { const char *nm = name_of(((const int32_t *)h.fixed)[s + r]); assign_len(vec, row_count + r, nm, strlen(nm)); }
            } else if (w) {
                memcpy((uint8_t *)get_data(vec) + row_count * w, (const uint8_t *)h.fixed + (size_t)s * w, take * w);
                for (idx_t r = 0; r < take; r++) if (!h.valid[s + r]) set_null(vec, row_count + r);
            } else {
                for (idx_t r = 0; r < take; r++) {
                    if (h.valid[s + r]) assign_len(vec, row_count + r, (const char *)h.bytes + h.off[s + r], h.off[s + r + 1] - h.off[s + r]);
                    else set_null(vec, row_count + r);
                }
            }
            continue;
        }
        // LIST: entries {offset = current child size, length}; children appended in row order (bcf_reader.c:1403-1424, 1436-1461, 1584-1610)
        duckdb_list_entry *le = (duckdb_list_entry *)get_data(vec);
        idx_t base = list_size(vec);
        const uint32_t c0 = h.off[s], c1 = h.off[s + take];
        if (c1 > c0) { list_reserve(vec, base + (c1 - c0)); list_set_size(vec, base + (c1 - c0)); }
        duckdb_vector child = list_child(vec);
        for (idx_t r = 0; r < take; r++) {
            le[row_count + r].offset = base + (h.off[s + r] - c0); le[row_count + r].length = h.off[s + r + 1] - h.off[s + r];
            if (!h.valid[s + r]) set_null(vec, row_count + r);
        }
        if (c1 > c0) {
            const std::vector<uint32_t> &cv32 = hb->conv[g->slot[ci]];
            if (names) for (uint32_t k = c0; k < c1; k++) { const char *nm = name_of((int32_t)h.child_fixed[k]); assign_len(child, base + (k - c0), nm, strlen(nm)); }
            else if (inf.type == DHTS_T_VARCHAR) {
                for (uint32_t k = c0; k < c1; k++)
                    if (!h.child_valid || h.child_valid[k]) assign_len(child, base + (k - c0), (const char *)h.bytes + h.child_off[k], h.child_off[k + 1] - h.child_off[k]);
            } else memcpy((uint32_t *)get_data(child) + base, (inf.encoding == DHTS_ENC_FLOAT_TEXT ? cv32.data() : h.child_fixed) + c0, (size_t)(c1 - c0) * 4);
            if (h.child_valid) {                                // NULL elements: a field the transcript does not have (bcf_reader.c:1485-1530)
                API(void, duckdb_vector_ensure_validity_writable, duckdb_vector)(child);
                uint64_t *cv = API(uint64_t *, duckdb_vector_get_validity, duckdb_vector)(child);
                for (uint32_t k = c0; k < c1; k++) {
                    const idx_t at = base + (k - c0);
                    if (h.child_valid[k]) cv[at / 64] |= (uint64_t)1 << (at % 64); else cv[at / 64] &= ~((uint64_t)1 << (at % 64));
                }
            }
        }
    }
}

static void bcf_read_function(duckdb_function_info info, duckdb_data_chunk output) {
    BcfBind *bind = (BcfBind *)API(void *, duckdb_function_get_bind_data, duckdb_function_info)(info);
    BcfScan *g = (BcfScan *)API(void *, duckdb_function_get_init_data, duckdb_function_info)(info);
    BcfLocal *l = (BcfLocal *)API(void *, duckdb_function_get_local_init_data, duckdb_function_info)(info);
    auto set_size = API(void, duckdb_data_chunk_set_size, duckdb_data_chunk, idx_t);
    if (!l || !g || l->done) { set_size(output, 0); return; }                                 // bcf_reader.c:1166-1169
    const idx_t vector_size = API(idx_t, duckdb_vector_size, void)();
    idx_t row_count = 0;
    while (row_count < vector_size) {
        if (!l->cur || l->pos >= l->end) {
            if (g->n_workers > 1 && row_count > 0) break;       // parallel mode: one slice per chunk
            if (!bcf_next_rows(g, l, vector_size)) {
                l->done = true;
                std::string err; { std::lock_guard<std::mutex> lk(g->mu); err = g->error; }
                if (!err.empty()) { API(void, duckdb_function_set_error, duckdb_function_info, const char *)(info, err.c_str()); set_size(output, 0); return; }
                break;
            }
        }
        idx_t take = (idx_t)(l->end - l->pos); if (take > vector_size - row_count) take = vector_size - row_count;
        bcf_fill(bind, g, l->cur, l->pos, take, output, row_count);
        row_count += take; l->pos += (int64_t)take;
    }
    if (l->done && l->cur) { std::lock_guard<std::mutex> lk(g->mu); l->cur->readers--; if (g->n_workers == 1 || (l->cur->retired && l->cur->readers == 0)) { g->free_slots.push_back(l->cur); g->cv_free.notify_all(); } l->cur = nullptr; }
    set_size(output, row_count);
}

extern "C" __attribute__((visibility("default"))) void register_read_bcf_function(duckdb_connection connection) {                       // bcf_reader.c:2055-2080
    duckdb_table_function tf = API(duckdb_table_function, duckdb_create_table_function, void)();
    API(void, duckdb_table_function_set_name, duckdb_table_function, const char *)(tf, "read_bcf");
    auto mk = API(duckdb_logical_type, duckdb_create_logical_type, int);
    auto rm = API(void, duckdb_destroy_logical_type, duckdb_logical_type *);
    auto named = API(void, duckdb_table_function_add_named_parameter, duckdb_table_function, const char *, duckdb_logical_type);
    duckdb_logical_type t_varchar = mk(DUCKDB_TYPE_VARCHAR);
    API(void, duckdb_table_function_add_parameter, duckdb_table_function, duckdb_logical_type)(tf, t_varchar);
    named(tf, "region", t_varchar); named(tf, "index_path", t_varchar);
    rm(&t_varchar);
    duckdb_logical_type t_bool = mk(DUCKDB_TYPE_BOOLEAN);
    named(tf, "tidy_format", t_bool);
    rm(&t_bool);
    API(void, duckdb_table_function_set_bind, duckdb_table_function, duckdb_table_function_bind_t)(tf, bcf_read_bind);
    API(void, duckdb_table_function_set_init, duckdb_table_function, duckdb_table_function_init_t)(tf, bcf_read_global_init);
    API(void, duckdb_table_function_set_local_init, duckdb_table_function, duckdb_table_function_init_t)(tf, bcf_read_local_init);
    API(void, duckdb_table_function_set_function, duckdb_table_function, duckdb_table_function_t)(tf, bcf_read_function);
    API(void, duckdb_table_function_supports_projection_pushdown, duckdb_table_function, bool)(tf, true);
    API(duckdb_state, duckdb_register_table_function, duckdb_connection, duckdb_table_function)(connection, tf);
    API(void, duckdb_destroy_table_function, duckdb_table_function *)(&tf);
}

extern "C" __attribute__((visibility("default"), weak)) bool duckhts_init_c_api(duckdb_extension_info info, struct duckdb_extension_access *access) {
    // duckdb_extension.h:1151-1158,1182-1194: fetch the API table, connect, register, disconnect
    const void *api = access->get_api(info, DUCKHTS_API_VERSION);
    if (!api) return false;
    dhts_set_duckdb_api(api);
    duckdb_database *db = access->get_database(info);
    duckdb_connection conn = nullptr;
    if (API(duckdb_state, duckdb_connect, duckdb_database, duckdb_connection *)(*db, &conn) == DuckDBError) {
        access->set_error(info, "Failed to open connection to database");
        return false;
    }
    register_read_bcf_function(conn);                     // registration order of src/duckhts.c:54-71
    register_read_bam_function(conn);
    register_bgzip_function(conn); register_bgunzip_function(conn);           // (the readers between them in src/duckhts.c are not on this path)
    register_bam_index_function(conn); register_bcf_index_function(conn); register_tabix_index_function(conn);
    API(void, duckdb_disconnect, duckdb_connection *)(&conn);
    return true;
}
