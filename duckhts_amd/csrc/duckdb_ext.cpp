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
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

// duckdb_ext_api_v1 viewed as an array of function pointers (slot numbers: include/duckdb_abi_slots.h).  Weak: the definition
// DUCKDB_EXTENSION_GLOBAL emits in a reference-built src/duckhts.c (duckdb_extension.h:1151) replaces it.
extern "C" __attribute__((visibility("default"), weak)) const void *duckdb_ext_api[DUCKDB_ABI_V120_NSLOTS] = {nullptr};

#define API(ret, name, ...) ((ret(*)(__VA_ARGS__))duckdb_ext_api[SLOT_##name])

// what DUCKDB_EXTENSION_API_INIT does (`duckdb_ext_api = *res`, duckdb_extension.h:1153-1158), for hosts that hold the table
extern "C" __attribute__((visibility("default"))) void dhts_set_duckdb_api(const void *api_table) {
    if (api_table) memcpy((void *)duckdb_ext_api, api_table, sizeof(void *) * DUCKDB_ABI_V120_NSLOTS);
}

static inline void set_null(duckdb_vector vec, idx_t row) {           // src/bam_reader.c:38-42
    API(void, duckdb_vector_ensure_validity_writable, duckdb_vector)(vec);
    uint64_t *v = API(uint64_t *, duckdb_vector_get_validity, duckdb_vector)(vec);
    v[row / 64] &= ~((uint64_t)1 << (row % 64));
}

struct BamBind {
    std::string path, region, index_file;
    dhts_ctx *ctx = nullptr;          // resident file + header, reused by the scan (one GPU)
    dhts_bam_header hdr;
    int has_index = 0;
    int standard_tags = 0, auxiliary_tags = 0;
    idx_t aux_col_idx = (idx_t)-1;
};

struct HostStr { std::vector<uint32_t> off, len; std::vector<uint8_t> bytes; };
struct BamLocal {
    std::vector<idx_t> column_ids;
    uint32_t colmask = 0;
    bool done = false;
    // current batch, host side
    int64_t n = 0, cur = 0;
    int status = 0;
    std::vector<uint16_t> flag; std::vector<int64_t> pos, pnext, tlen; std::vector<int32_t> mapq, tid, mtid, rgidx; std::vector<uint64_t> rgvalid;
    HostStr qname, cigar, seq, qual, rg;
    std::vector<int32_t> tag_ids;        // standard-tag ids requested by the projection
    std::vector<int> tag_slot;           // output vector -> index into tags (-1 = not a tag column)
    struct HostTag { std::vector<uint8_t> valid, bytes; std::vector<int64_t> fixed; std::vector<uint32_t> off; std::vector<int64_t> child; };
    std::vector<HostTag> tags;
    // AUXILIARY_TAGS of the current batch: list offsets + rendered key / value strings
    bool want_aux = false;
    std::vector<uint8_t> aux_valid; std::vector<uint32_t> aux_off; std::vector<std::string> aux_key, aux_val;
};

static void destroy_bind(void *p) { BamBind *b = (BamBind *)p; if (!b) return; if (b->ctx) dhts_destroy(b->ctx); delete b; }
static void destroy_local(void *p) { delete (BamLocal *)p; }
static void destroy_global(void *p) { free(p); }

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
    int dev = getenv("DHTS_DEVICE") ? atoi(getenv("DHTS_DEVICE")) : 0;
    b->ctx = dhts_create(dev);
    if (!b->ctx) { set_error(info, "read_bam: no MI355X (gfx950) device available; this build has no CPU fallback"); destroy_bind(b); return; }
    if (dhts_open_path(b->ctx, b->path.c_str()) != 0) {
        snprintf(err, sizeof(err), "Failed to open SAM/BAM/CRAM file: %s", b->path.c_str());
        set_error(info, err); destroy_bind(b); return;
    }
    if (dhts_bgzf_index(b->ctx) <= 0 || dhts_bam_open(b->ctx) != 0 || dhts_bam_header_get(b->ctx, &b->hdr) != 0) {
        set_error(info, "Failed to read SAM/BAM/CRAM header");                    // bam_reader.c:461 (also what SAM/CRAM input gets here)
        destroy_bind(b); return;
    }
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

static void bam_read_global_init(duckdb_init_info info) {
    // sequential mode: one scan thread (bam_reader.c:582-585).  The GPU supplies the parallelism.
    API(void, duckdb_init_set_max_threads, duckdb_init_info, idx_t)(info, 1);
    API(void, duckdb_init_set_init_data, duckdb_init_info, void *, duckdb_delete_callback_t)(info, calloc(1, 16), destroy_global);
}

static void bam_read_local_init(duckdb_init_info info) {
    BamBind *bind = (BamBind *)API(void *, duckdb_init_get_bind_data, duckdb_init_info)(info);
    BamLocal *l = new BamLocal();
    idx_t n = API(idx_t, duckdb_init_get_column_count, duckdb_init_info)(info);               // bam_reader.c:676-679
    for (idx_t i = 0; i < n; i++) {
        idx_t id = API(idx_t, duckdb_init_get_column_index, duckdb_init_info, idx_t)(info, i);
        l->column_ids.push_back(id);
        if (id < DHTS_BAM_CORE_COUNT) l->colmask |= 1u << id;
        int sl = -1;
        if (bind->standard_tags && id >= DHTS_BAM_CORE_COUNT && id < (idx_t)(DHTS_BAM_CORE_COUNT + dhts_bam_std_tag_count())) {
            const int32_t tid_ = (int32_t)(id - DHTS_BAM_CORE_COUNT);
            for (size_t k = 0; k < l->tag_ids.size(); k++) if (l->tag_ids[k] == tid_) sl = (int)k;
            if (sl < 0) { sl = (int)l->tag_ids.size(); l->tag_ids.push_back(tid_); }
        }
        l->tag_slot.push_back(sl);
    }
    l->tags.resize(l->tag_ids.size());
    dhts_bam_set_tag_columns(bind->ctx, l->tag_ids.data(), (int32_t)l->tag_ids.size());
    for (idx_t id : l->column_ids) if (bind->auxiliary_tags && id == bind->aux_col_idx) l->want_aux = true;
    dhts_bam_set_aux_map(bind->ctx, l->want_aux ? 1 : 0, bind->standard_tags);
    auto init_error = API(void, duckdb_init_set_error, duckdb_init_info, const char *);
    if (!bind->region.empty()) {
        // bam_reader.c:639-668: a region needs an index; sam_itr_regarray failing reports "No reads found"
        if (!bind->has_index) { init_error(info, "Region query requires an index (.bai/.csi/.crai)"); delete l; return; }
        int rc = dhts_bam_set_regions(bind->ctx, bind->region.c_str());
        if (rc != 0) {
            char err[640]; snprintf(err, sizeof(err), "No reads found for region(s): %s", bind->region.c_str());
            init_error(info, rc == 1 ? err : dhts_error(bind->ctx)); delete l; return;
        }
        // the index (BAI or CSI) narrows the scan window; the device predicate decides the rows
        FILE *f = fopen(bind->index_file.c_str(), "rb");
        if (f) {
            std::vector<uint8_t> ib; uint8_t tmp[65536]; size_t k;
            while ((k = fread(tmp, 1, sizeof(tmp), f)) > 0) ib.insert(ib.end(), tmp, tmp + k);
            fclose(f);
            const bool known = ib.size() >= 4 && (memcmp(ib.data(), "BAI\1", 4) == 0 || memcmp(ib.data(), "CSI\1", 4) == 0 || (ib[0] == 0x1f && ib[1] == 0x8b));
            if (known && dhts_bam_load_index(bind->ctx, ib.data(), ib.size()) != 0) { init_error(info, dhts_error(bind->ctx)); delete l; return; }
        }
    } else if (dhts_bam_set_regions(bind->ctx, nullptr) != 0) { init_error(info, "Failed to open SAM/BAM/CRAM file"); delete l; return; }
    if (bind->region.empty() && dhts_bam_rewind(bind->ctx) != 0) { init_error(info, "Failed to open SAM/BAM/CRAM file"); delete l; return; }
    API(void, duckdb_init_set_init_data, duckdb_init_info, void *, duckdb_delete_callback_t)(info, l, destroy_local);
}

static int fetch_str(dhts_ctx *c, const dhts_strcol &d, int64_t n, HostStr &h, bool want) {
    if (!want) return 0;
    h.off.resize(n + 1); h.len.resize(n); h.bytes.resize(d.nbytes + 1);
    if (dhts_memcpy_d2h(c, h.off.data(), d.off, (n + 1) * 4) || dhts_memcpy_d2h(c, h.len.data(), d.len, n * 4) || dhts_memcpy_d2h(c, h.bytes.data(), d.bytes, d.nbytes)) return -1;
    return 0;
}

// pulls the next GPU batch and copies only the projected columns to the host
static int next_host_batch(BamBind *bind, BamLocal *l) {
    dhts_bam_batch b;
    for (;;) {
        if (dhts_bam_next_batch(bind->ctx, 0, l->colmask, &b) != 0) return -1;
        l->status = b.status;
        if (b.n_rows > 0 || b.status != 0) break;
    }
    int64_t n = b.n_rows; l->n = n; l->cur = 0;
    if (n == 0) return 0;
    dhts_ctx *c = bind->ctx; uint32_t m = l->colmask;
#define FETCH(vec, ptr, bit) do { if (m & (1u << (bit))) { (vec).resize(n); if (dhts_memcpy_d2h(c, (vec).data(), ptr, n * sizeof((vec)[0]))) return -1; } } while (0)
    FETCH(l->flag, b.flag, DHTS_BAM_FLAG); FETCH(l->pos, b.pos, DHTS_BAM_POS); FETCH(l->mapq, b.mapq, DHTS_BAM_MAPQ);
    FETCH(l->pnext, b.pnext, DHTS_BAM_PNEXT); FETCH(l->tlen, b.tlen, DHTS_BAM_TLEN); FETCH(l->tid, b.tid, DHTS_BAM_RNAME); FETCH(l->mtid, b.mtid, DHTS_BAM_RNEXT);
    FETCH(l->rgidx, b.rg_idx, DHTS_BAM_SAMPLE_ID);
    if (m & ((1u << DHTS_BAM_READ_GROUP_ID) | (1u << DHTS_BAM_SAMPLE_ID))) { l->rgvalid.resize((n + 63) / 64); if (dhts_memcpy_d2h(c, l->rgvalid.data(), b.rg_valid, l->rgvalid.size() * 8)) return -1; }
    if (fetch_str(c, b.qname, n, l->qname, m & (1u << DHTS_BAM_QNAME)) || fetch_str(c, b.cigar, n, l->cigar, m & (1u << DHTS_BAM_CIGAR)) ||
        fetch_str(c, b.seq, n, l->seq, m & (1u << DHTS_BAM_SEQ)) || fetch_str(c, b.qual, n, l->qual, m & (1u << DHTS_BAM_QUAL)) ||
        fetch_str(c, b.rg, n, l->rg, m & (1u << DHTS_BAM_READ_GROUP_ID))) return -1;
    if (l->want_aux && b.aux_map) {
        // typed entries -> value text, bam_aux_to_string (bam_reader.c:140-183); assigned through the NUL-terminated API
        const dhts_aux_map &am = *b.aux_map; const size_t ne = (size_t)am.n_ent;
        std::vector<uint16_t> key(ne + 1); std::vector<uint8_t> kind(ne + 1), sub(ne + 1), pay(am.payload_bytes + 8); std::vector<uint32_t> po(ne + 2);
        l->aux_valid.resize(n); l->aux_off.resize(n + 1);
        if (dhts_memcpy_d2h(c, l->aux_valid.data(), am.valid, n) || dhts_memcpy_d2h(c, l->aux_off.data(), am.off, (n + 1) * 4)) return -1;
        if (ne && (dhts_memcpy_d2h(c, key.data(), am.key, ne * 2) || dhts_memcpy_d2h(c, kind.data(), am.kind, ne) || dhts_memcpy_d2h(c, sub.data(), am.sub, ne))) return -1;
        if (dhts_memcpy_d2h(c, po.data(), am.pay_off, (ne + 1) * 4)) return -1;
        if (am.payload_bytes && dhts_memcpy_d2h(c, pay.data(), am.payload, am.payload_bytes)) return -1;
        l->aux_key.assign(ne, std::string()); l->aux_val.assign(ne, std::string());
        char tmp[64];
        for (size_t i = 0; i < ne; i++) {
            char kb[3] = {(char)(key[i] & 0xff), (char)(key[i] >> 8), 0};
            l->aux_key[i] = kb;
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
            l->aux_val[i] = v.c_str();                              // C-string semantics: cut at the first NUL
        }
    }
    for (int i = 0; i < b.n_tag_cols; i++) {
        const dhts_col &d = b.tag_cols[i]; BamLocal::HostTag &h = l->tags[i];
        h.valid.resize(n); if (dhts_memcpy_d2h(c, h.valid.data(), d.valid, n)) return -1;
        if (d.fixed) { h.fixed.resize(n); if (dhts_memcpy_d2h(c, h.fixed.data(), d.fixed, n * 8)) return -1; }
        if (d.off) { h.off.resize(n + 1); if (dhts_memcpy_d2h(c, h.off.data(), d.off, (n + 1) * 4)) return -1; }
        if (d.bytes || d.nbytes == 0) { h.bytes.resize(d.nbytes + 1); if (d.nbytes && dhts_memcpy_d2h(c, h.bytes.data(), d.bytes, d.nbytes)) return -1; }
        if (d.child_fixed) { h.child.resize(d.child_n + 1); if (d.child_n && dhts_memcpy_d2h(c, h.child.data(), d.child_fixed, d.child_n * 8)) return -1; }
    }
    return 0;
}

static void bam_read_function(duckdb_function_info info, duckdb_data_chunk output) {
    BamBind *bind = (BamBind *)API(void *, duckdb_function_get_bind_data, duckdb_function_info)(info);
    BamLocal *l = (BamLocal *)API(void *, duckdb_function_get_local_init_data, duckdb_function_info)(info);
    auto set_size = API(void, duckdb_data_chunk_set_size, duckdb_data_chunk, idx_t);
    if (!l || l->done) { set_size(output, 0); return; }                                      // bam_reader.c:730-733
    const idx_t vector_size = API(idx_t, duckdb_vector_size, void)();
    auto get_vec = API(duckdb_vector, duckdb_data_chunk_get_vector, duckdb_data_chunk, idx_t);
    auto get_data = API(void *, duckdb_vector_get_data, duckdb_vector);
    auto assign_len = API(void, duckdb_vector_assign_string_element_len, duckdb_vector, idx_t, const char *, idx_t);
    idx_t row_count = 0;
    while (row_count < vector_size) {
        if (l->cur >= l->n) {
            if (l->status != 0) { l->done = true; break; }          // end of stream or silent stop on error (bam_reader.c:754-766)
            if (next_host_batch(bind, l) != 0) {
                API(void, duckdb_function_set_error, duckdb_function_info, const char *)(info, dhts_error(bind->ctx));
                l->done = true; set_size(output, 0); return;
            }
            if (l->n == 0) { l->done = true; break; }
        }
        idx_t take = (idx_t)(l->n - l->cur); if (take > vector_size - row_count) take = vector_size - row_count;
        const int64_t s = l->cur;
        for (size_t ci = 0; ci < l->column_ids.size(); ci++) {
            duckdb_vector vec = get_vec(output, ci);
            auto put_str = [&](const HostStr &h) { for (idx_t r = 0; r < take; r++) assign_len(vec, row_count + r, (const char *)h.bytes.data() + h.off[s + r], h.len[s + r]); };
            auto put_name = [&](const std::vector<int32_t> &ids) {
                for (idx_t r = 0; r < take; r++) { int32_t t = ids[s + r]; const char *nm = t >= 0 ? bind->hdr.ref_name[t] : "*"; assign_len(vec, row_count + r, nm, strlen(nm)); } };
            switch (l->column_ids[ci]) {
            case DHTS_BAM_QNAME: put_str(l->qname); break;
            case DHTS_BAM_FLAG: memcpy((uint16_t *)get_data(vec) + row_count, l->flag.data() + s, take * 2); break;
            case DHTS_BAM_RNAME: put_name(l->tid); break;
            case DHTS_BAM_POS: memcpy((int64_t *)get_data(vec) + row_count, l->pos.data() + s, take * 8); break;
            case DHTS_BAM_MAPQ: memcpy((int32_t *)get_data(vec) + row_count, l->mapq.data() + s, take * 4); break;
            case DHTS_BAM_CIGAR: put_str(l->cigar); break;
            case DHTS_BAM_RNEXT: put_name(l->mtid); break;
            case DHTS_BAM_PNEXT: memcpy((int64_t *)get_data(vec) + row_count, l->pnext.data() + s, take * 8); break;
            case DHTS_BAM_TLEN: memcpy((int64_t *)get_data(vec) + row_count, l->tlen.data() + s, take * 8); break;
            case DHTS_BAM_SEQ: put_str(l->seq); break;
            case DHTS_BAM_QUAL: put_str(l->qual); break;
            case DHTS_BAM_READ_GROUP_ID:
                for (idx_t r = 0; r < take; r++) {
                    int64_t g = s + (int64_t)r;
                    if ((l->rgvalid[g >> 6] >> (g & 63)) & 1) assign_len(vec, row_count + r, (const char *)l->rg.bytes.data() + l->rg.off[g], l->rg.len[g]);
                    else set_null(vec, row_count + r);
                }
                break;
            case DHTS_BAM_SAMPLE_ID:
                for (idx_t r = 0; r < take; r++) {
                    int64_t g = s + (int64_t)r; int32_t k = l->rgidx[g];
                    const char *sm = (((l->rgvalid[g >> 6] >> (g & 63)) & 1) && k >= 0) ? bind->hdr.rg_sm[k] : nullptr;
                    if (sm) assign_len(vec, row_count + r, sm, strlen(sm)); else set_null(vec, row_count + r);
                }
                break;
            default: {
                if (l->want_aux && l->column_ids[ci] == bind->aux_col_idx) {               // bam_reader.c:967-1027
                    auto list_size = API(idx_t, duckdb_list_vector_get_size, duckdb_vector);
                    duckdb_list_entry *le = (duckdb_list_entry *)get_data(vec);
                    idx_t base = list_size(vec);
                    const uint32_t c0 = l->aux_off[s], c1 = l->aux_off[s + take];
                    if (c1 > c0) { API(duckdb_state, duckdb_list_vector_reserve, duckdb_vector, idx_t)(vec, base + (c1 - c0)); API(duckdb_state, duckdb_list_vector_set_size, duckdb_vector, idx_t)(vec, base + (c1 - c0)); }
                    duckdb_vector child = API(duckdb_vector, duckdb_list_vector_get_child, duckdb_vector)(vec);
                    duckdb_vector kvec = API(duckdb_vector, duckdb_struct_vector_get_child, duckdb_vector, idx_t)(child, 0);
                    duckdb_vector vvec = API(duckdb_vector, duckdb_struct_vector_get_child, duckdb_vector, idx_t)(child, 1);
                    for (idx_t r = 0; r < take; r++) {
                        le[row_count + r].offset = base + (l->aux_off[s + r] - c0); le[row_count + r].length = l->aux_off[s + r + 1] - l->aux_off[s + r];
                        if (!l->aux_valid[s + r]) set_null(vec, row_count + r);            // no tags: NULL, entry {size, 0}
                    }
                    for (uint32_t k = c0; k < c1; k++) {
                        assign_len(kvec, base + (k - c0), l->aux_key[k].data(), l->aux_key[k].size());
                        assign_len(vvec, base + (k - c0), l->aux_val[k].data(), l->aux_val[k].size());
                    }
                    break;
                }
                const int sl = l->tag_slot[ci];
                if (sl < 0) break;                                 // unknown ids (e.g. a row-id pseudo column) write nothing, like the reference's default arm
                const BamLocal::HostTag &h = l->tags[sl];
                char nm[3], ty, sub; dhts_bam_std_tag_info(l->tag_ids[sl], nm, &ty, &sub);
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
        row_count += take; l->cur += (int64_t)take;
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
// Sequential mode; tidy_format and region supported; VCF text input is rejected with the reference's header error.
// =====================================================================================================================
struct BcfBind {
    std::string path, region;
    std::vector<std::string> regions;    // comma split, empty tokens dropped (parse_regions_duckdb, bcf_reader.c:423-446)
    std::string index_file; std::vector<uint8_t> index_bytes;
    dhts_ctx *ctx = nullptr;
    dhts_bcf_info inf;
    int has_index = 0;
};
struct HostCol {
    int col = 0;
    std::vector<uint8_t> valid, fixed, bytes;
    std::vector<uint32_t> off, child_off, child_fixed;
    uint64_t child_n = 0;
};
struct BcfLocal {
    std::vector<idx_t> column_ids;       // schema ids per output vector
    std::vector<int> slot;               // output vector -> index into `cols` (or -1 for unknown ids)
    std::vector<HostCol> cols;           // projected (deduplicated) columns of the current batch
    bool done = false; int64_t n = 0, cur = 0; int status = 0;
    size_t next_region = 0;              // chained single-region iterators (bcf_reader.c:1327-1345)
};
// advance to the next region that yields an iterator; false when none is left
static bool bcf_next_region(BcfBind *bind, BcfLocal *l) {
    while (l->next_region < bind->regions.size()) {
        const std::string &rg = bind->regions[l->next_region++];
        if (dhts_bcf_set_region(bind->ctx, rg.c_str()) == 0) {                  // unknown contig / malformed: skipped (bcf_reader.c:944-953)
            if (!bind->index_bytes.empty()) (void)dhts_bcf_load_index(bind->ctx, bind->index_bytes.data(), bind->index_bytes.size());   // window only: a failure keeps the full scan
            return true;
        }
    }
    return false;
}
static void destroy_bcf_bind(void *p) { BcfBind *b = (BcfBind *)p; if (!b) return; if (b->ctx) dhts_destroy(b->ctx); delete b; }
static void destroy_bcf_local(void *p) { delete (BcfLocal *)p; }

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
    b->ctx = dhts_create(dev);
    if (!b->ctx) { set_error(info, "read_bcf: no MI355X (gfx950) device available; this build has no CPU fallback"); destroy_bcf_bind(b); return; }
    if (dhts_open_path(b->ctx, b->path.c_str()) != 0) {
        snprintf(err, sizeof(err), "Failed to open BCF/VCF file: %s", b->path.c_str());
        set_error(info, err); destroy_bcf_bind(b); return;
    }
    if (dhts_bgzf_index(b->ctx) <= 0 || dhts_bcf_open(b->ctx, tidy) != 0 || dhts_bcf_info_get(b->ctx, &b->inf) != 0) {
        const char *m = dhts_error(b->ctx);
        set_error(info, (m && strstr(m, "VEP")) ? m : "Failed to read BCF/VCF header");       // bcf_reader.c:505
        destroy_bcf_bind(b); return;
    }
    for (const std::string &f : {idx, b->path + ".csi", b->path + ".tbi"}) if (!f.empty() && file_exists(f)) { b->index_file = f; break; }
    b->has_index = !b->index_file.empty();
    if (b->has_index && !b->regions.empty() && (b->index_file.size() < 4 || b->index_file.compare(b->index_file.size() - 4, 4, ".tbi") != 0)) {
        FILE *f = fopen(b->index_file.c_str(), "rb");
        if (f) { uint8_t tmp[65536]; size_t k; while ((k = fread(tmp, 1, sizeof(tmp), f)) > 0) b->index_bytes.insert(b->index_bytes.end(), tmp, tmp + k); fclose(f); }
    }
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

static void bcf_read_global_init(duckdb_init_info info) {
    BcfBind *bind = (BcfBind *)API(void *, duckdb_init_get_bind_data, duckdb_init_info)(info);
    if (!bind->regions.empty() && !bind->has_index) {
        char err[900];
        snprintf(err, sizeof(err), "Region query requires an index file (.tbi or .csi). Region: %s", bind->region.c_str());   // bcf_reader.c:922-923
        API(void, duckdb_init_set_error, duckdb_init_info, const char *)(info, err);
        return;
    }
    API(void, duckdb_init_set_max_threads, duckdb_init_info, idx_t)(info, 1);
    API(void, duckdb_init_set_init_data, duckdb_init_info, void *, duckdb_delete_callback_t)(info, calloc(1, 16), destroy_global);
}

static void bcf_read_local_init(duckdb_init_info info) {
    BcfBind *bind = (BcfBind *)API(void *, duckdb_init_get_bind_data, duckdb_init_info)(info);
    BcfLocal *l = new BcfLocal();
    idx_t n = API(idx_t, duckdb_init_get_column_count, duckdb_init_info)(info);
    std::vector<int32_t> proj;
    for (idx_t i = 0; i < n; i++) {
        idx_t id = API(idx_t, duckdb_init_get_column_index, duckdb_init_info, idx_t)(info, i);
        l->column_ids.push_back(id);
        int sl = -1;
        if (id < (idx_t)bind->inf.n_cols) {
            for (size_t k = 0; k < proj.size(); k++) if (proj[k] == (int32_t)id) sl = (int)k;
            if (sl < 0) { sl = (int)proj.size(); proj.push_back((int32_t)id); }
        }
        l->slot.push_back(sl);
    }
    l->cols.resize(proj.size());
    if (dhts_bcf_set_projection(bind->ctx, proj.data(), (int32_t)proj.size()) != 0 || dhts_bcf_set_region(bind->ctx, nullptr) != 0) {
        API(void, duckdb_init_set_error, duckdb_init_info, const char *)(info, "Failed to open BCF/VCF file"); delete l; return;
    }
    if (!bind->regions.empty() && !bcf_next_region(bind, l)) l->done = true;      // no region produced an iterator: zero rows (bcf_reader.c:955-959)
    API(void, duckdb_init_set_init_data, duckdb_init_info, void *, duckdb_delete_callback_t)(info, l, destroy_bcf_local);
}

static size_t bcf_fixed_width(const dhts_bcf_colinfo &ci) {
    if (ci.is_list) return 0;
    if (ci.encoding != DHTS_ENC_PLAIN) return 4;
    switch (ci.type) { case DHTS_T_BOOLEAN: return 1; case DHTS_T_INTEGER: case DHTS_T_FLOAT: return 4; case DHTS_T_BIGINT: case DHTS_T_DOUBLE: return 8; default: return 0; }
}

static int bcf_next_host_batch(BcfBind *bind, BcfLocal *l) {
    dhts_bcf_batch b;
    for (;;) {
        if (dhts_bcf_next_batch(bind->ctx, 0, &b) != 0) return -1;
        l->status = b.status;
        if (b.n_rows > 0 || b.status != 0) break;
    }
    const int64_t n = b.n_rows; l->n = n; l->cur = 0;
    if (n == 0) return 0;
    dhts_ctx *c = bind->ctx;
    for (int i = 0; i < b.n_cols; i++) {
        const dhts_bcf_col &d = b.cols[i]; HostCol &h = l->cols[i]; h.col = d.col;
        const dhts_bcf_colinfo &ci = bind->inf.cols[d.col];
        h.valid.resize(n); if (dhts_memcpy_d2h(c, h.valid.data(), d.valid, n)) return -1;
        const size_t w = bcf_fixed_width(ci);
        if (w) { h.fixed.resize(w * n); if (dhts_memcpy_d2h(c, h.fixed.data(), d.fixed, w * n)) return -1; }
        if (d.off) { h.off.resize(n + 1); if (dhts_memcpy_d2h(c, h.off.data(), d.off, (n + 1) * 4)) return -1; }
        if (d.bytes) { h.bytes.resize(d.nbytes + 1); if (dhts_memcpy_d2h(c, h.bytes.data(), d.bytes, d.nbytes)) return -1; }
        h.child_n = d.child_n;
        if (d.child_fixed) { h.child_fixed.resize(d.child_n + 1); if (dhts_memcpy_d2h(c, h.child_fixed.data(), d.child_fixed, d.child_n * 4)) return -1; }
        if (d.child_off) { h.child_off.resize(d.child_n + 1); if (dhts_memcpy_d2h(c, h.child_off.data(), d.child_off, (d.child_n + 1) * 4)) return -1; }
    }
    return 0;
}

static void bcf_read_function(duckdb_function_info info, duckdb_data_chunk output) {
    BcfBind *bind = (BcfBind *)API(void *, duckdb_function_get_bind_data, duckdb_function_info)(info);
    BcfLocal *l = (BcfLocal *)API(void *, duckdb_function_get_local_init_data, duckdb_function_info)(info);
    auto set_size = API(void, duckdb_data_chunk_set_size, duckdb_data_chunk, idx_t);
    if (!l || l->done) { set_size(output, 0); return; }                                       // bcf_reader.c:1166-1169
    const idx_t vector_size = API(idx_t, duckdb_vector_size, void)();
    auto get_vec = API(duckdb_vector, duckdb_data_chunk_get_vector, duckdb_data_chunk, idx_t);
    auto get_data = API(void *, duckdb_vector_get_data, duckdb_vector);
    auto assign_len = API(void, duckdb_vector_assign_string_element_len, duckdb_vector, idx_t, const char *, idx_t);
    auto list_size = API(idx_t, duckdb_list_vector_get_size, duckdb_vector);
    auto list_reserve = API(duckdb_state, duckdb_list_vector_reserve, duckdb_vector, idx_t);
    auto list_set_size = API(duckdb_state, duckdb_list_vector_set_size, duckdb_vector, idx_t);
    auto list_child = API(duckdb_vector, duckdb_list_vector_get_child, duckdb_vector);
    idx_t row_count = 0;
    while (row_count < vector_size) {
        if (l->cur >= l->n) {
            if (l->status != 0) {                                    // EOF, or silent stop at the first bad record (bcf_reader.c:1319-1349)
                if (!bind->regions.empty() && bcf_next_region(bind, l)) { l->status = 0; l->n = l->cur = 0; }
                else { l->done = true; break; }
            }
            if (bcf_next_host_batch(bind, l) != 0) {
                API(void, duckdb_function_set_error, duckdb_function_info, const char *)(info, dhts_error(bind->ctx));
                l->done = true; set_size(output, 0); return;
            }
            if (l->n == 0) { if (l->status != 0) continue; l->done = true; break; }
        }
        idx_t take = (idx_t)(l->n - l->cur); if (take > vector_size - row_count) take = vector_size - row_count;
        const int64_t s = l->cur;
        for (size_t ci = 0; ci < l->column_ids.size(); ci++) {
            if (l->slot[ci] < 0) continue;                          // ids outside the schema write nothing
            const HostCol &h = l->cols[l->slot[ci]];
            const dhts_bcf_colinfo &inf = bind->inf.cols[h.col];
            duckdb_vector vec = get_vec(output, ci);
            const char *const *names = inf.encoding == DHTS_ENC_CONTIG ? bind->inf.contig_name : inf.encoding == DHTS_ENC_DICT ? bind->inf.dict_name :
                                       inf.encoding == DHTS_ENC_SAMPLE ? bind->inf.sample_name : nullptr;
            auto name_of = [&](int32_t id) -> const char * { if (id < 0) return "PASS"; const char *nm = names[id]; return nm ? nm : "."; };
            if (!inf.is_list) {
                const size_t w = bcf_fixed_width(inf);
                if (names) {
                    for (idx_t r = 0; r < take; r++) { const char *nm = name_of(((const int32_t *)h.fixed.data())[s + r]); assign_len(vec, row_count + r, nm, strlen(nm)); }
                } else if (w) {
                    memcpy((uint8_t *)get_data(vec) + row_count * w, h.fixed.data() + (size_t)s * w, take * w);
                    for (idx_t r = 0; r < take; r++) if (!h.valid[s + r]) set_null(vec, row_count + r);
                } else {
                    for (idx_t r = 0; r < take; r++) {
                        if (h.valid[s + r]) assign_len(vec, row_count + r, (const char *)h.bytes.data() + h.off[s + r], h.off[s + r + 1] - h.off[s + r]);
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
                if (names) for (uint32_t k = c0; k < c1; k++) { const char *nm = name_of((int32_t)h.child_fixed[k]); assign_len(child, base + (k - c0), nm, strlen(nm)); }
                else if (inf.type == DHTS_T_VARCHAR) for (uint32_t k = c0; k < c1; k++) assign_len(child, base + (k - c0), (const char *)h.bytes.data() + h.child_off[k], h.child_off[k + 1] - h.child_off[k]);
                else memcpy((uint32_t *)get_data(child) + base, h.child_fixed.data() + c0, (size_t)(c1 - c0) * 4);
            }
        }
        row_count += take; l->cur += (int64_t)take;
    }
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
    API(void, duckdb_disconnect, duckdb_connection *)(&conn);
    return true;
}
