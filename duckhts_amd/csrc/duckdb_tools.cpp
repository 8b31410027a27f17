// duckdb_tools.cpp -- the file-writing table functions next to the readers, same DuckDB C-API surface as the reference:
//   register_bgzip_function / register_bgunzip_function                               src/bgzip.c:336-380  (bind does the work: :88-293)
//   register_bam_index_function / register_bcf_index_function / register_tabix_index_function   src/hts_index_builder.c:327-390 (binds :111-325)
// Each function does its work at bind time and returns one row, like the reference.  htslib's bgzf_write / bgzf_read / sam_index_build3 /
// bcf_index_build3 / tbx_index_build3 underneath are replaced by include/duckhts_amd.h: DEFLATE compression and inflation, the record scans
// that feed the index and the tabix interval rule all run on the GPU; the binning index itself (hts_idx_push / hts_idx_finish) is host work.
#include "../../include/duckhts_amd.h"
#include "../../include/duckhts_extension.h"

#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>
#include <string>
#include <vector>

#define API(ret, name, ...) ((ret(*)(__VA_ARGS__))duckdb_ext_api[SLOT_##name])

namespace {
struct OneRow { bool emitted = false; std::string path, format; int64_t bytes_in = 0, bytes_out = 0; bool four = false; };
void destroy_row(void *p) { delete (OneRow *)p; }

char *named_varchar(duckdb_bind_info info, const char *name) {
    duckdb_value v = API(duckdb_value, duckdb_bind_get_named_parameter, duckdb_bind_info, const char *)(info, name);
    char *s = nullptr;
    if (v && !API(bool, duckdb_is_null_value, duckdb_value)(v)) s = API(char *, duckdb_get_varchar, duckdb_value)(v);
    if (v) API(void, duckdb_destroy_value, duckdb_value *)(&v);
    return s;
}
bool named_int(duckdb_bind_info info, const char *name, int64_t *out) {
    duckdb_value v = API(duckdb_value, duckdb_bind_get_named_parameter, duckdb_bind_info, const char *)(info, name);
    bool set = false;
    if (v && !API(bool, duckdb_is_null_value, duckdb_value)(v)) { *out = API(int64_t, duckdb_get_int64, duckdb_value)(v); set = true; }
    if (v) API(void, duckdb_destroy_value, duckdb_value *)(&v);
    return set;
}
bool named_bool(duckdb_bind_info info, const char *name, bool *out) {
    duckdb_value v = API(duckdb_value, duckdb_bind_get_named_parameter, duckdb_bind_info, const char *)(info, name);
    bool set = false;
    if (v && !API(bool, duckdb_is_null_value, duckdb_value)(v)) { *out = API(bool, duckdb_get_bool, duckdb_value)(v); set = true; }
    if (v) API(void, duckdb_destroy_value, duckdb_value *)(&v);
    return set;
}
std::string take(char *s) { std::string r = s ? s : ""; if (s) API(void, duckdb_free, void *)(s); return r; }
std::string positional_path(duckdb_bind_info info) {
    duckdb_value v = API(duckdb_value, duckdb_bind_get_parameter, duckdb_bind_info, idx_t)(info, 0);
    char *s = API(char *, duckdb_get_varchar, duckdb_value)(v);
    API(void, duckdb_destroy_value, duckdb_value *)(&v);
    return take(s);
}
void bind_error(duckdb_bind_info info, const std::string &m) { API(void, duckdb_bind_set_error, duckdb_bind_info, const char *)(info, m.c_str()); }
bool ends_with(const std::string &s, const char *suf) { const size_t n = strlen(suf); return s.size() >= n && s.compare(s.size() - n, n, suf) == 0; }
int device_id() { return getenv("DHTS_DEVICE") ? atoi(getenv("DHTS_DEVICE")) : 0; }

void add_columns(duckdb_bind_info info, bool four) {
    auto mk = API(duckdb_logical_type, duckdb_create_logical_type, int);
    auto rm = API(void, duckdb_destroy_logical_type, duckdb_logical_type *);
    auto add = API(void, duckdb_bind_add_result_column, duckdb_bind_info, const char *, duckdb_logical_type);
    duckdb_logical_type tb = mk(DUCKDB_TYPE_BOOLEAN), tv = mk(DUCKDB_TYPE_VARCHAR), ti = mk(DUCKDB_TYPE_BIGINT);
    add(info, "success", tb);
    if (four) { add(info, "output_path", tv); add(info, "bytes_in", ti); add(info, "bytes_out", ti); }        // bgzip.c:74-86
    else { add(info, "index_path", tv); add(info, "index_format", tv); }                                       // hts_index_builder.c:70-79
    rm(&tb); rm(&tv); rm(&ti);
}
void row_init(duckdb_init_info info) { ((OneRow *)API(void *, duckdb_init_get_bind_data, duckdb_init_info)(info))->emitted = false; }
void row_scan(duckdb_function_info info, duckdb_data_chunk output) {
    OneRow *r = (OneRow *)API(void *, duckdb_function_get_bind_data, duckdb_function_info)(info);
    auto setn = API(void, duckdb_data_chunk_set_size, duckdb_data_chunk, idx_t);
    if (r->emitted) { setn(output, 0); return; }
    auto vec = API(duckdb_vector, duckdb_data_chunk_get_vector, duckdb_data_chunk, idx_t);
    auto data = API(void *, duckdb_vector_get_data, duckdb_vector);
    auto str = API(void, duckdb_vector_assign_string_element, duckdb_vector, idx_t, const char *);
    ((bool *)data(vec(output, 0)))[0] = true;
    str(vec(output, 1), 0, r->path.c_str());
    if (r->four) { ((int64_t *)data(vec(output, 2)))[0] = r->bytes_in; ((int64_t *)data(vec(output, 3)))[0] = r->bytes_out; }
    else str(vec(output, 2), 0, r->format.c_str());
    r->emitted = true;
    setn(output, 1);
}
void finish_bind(duckdb_bind_info info, OneRow *r) {
    add_columns(info, r->four);
    API(void, duckdb_bind_set_bind_data, duckdb_bind_info, void *, duckdb_delete_callback_t)(info, r, destroy_row);
}

// ---- bgzip / bgunzip (src/bgzip.c:88-293) ----
void bgzip_bind_common(duckdb_bind_info info, bool decompress) {
    const char *fn = decompress ? "bgunzip" : "bgzip";
    const std::string input = positional_path(info);
    if (input.empty()) { bind_error(info, std::string(fn) + " requires a file path"); return; }
    std::string output = take(named_varchar(info, "output_path"));
    int64_t threads = 4, level = -1; bool keep = true, overwrite = false;
    (void)named_int(info, "threads", &threads);                                   // (accepted; the blocks are processed by the GPU, not by bgzf_mt threads)
    if (!decompress) (void)named_int(info, "level", &level);
    (void)named_bool(info, "keep", &keep); (void)named_bool(info, "overwrite", &overwrite);
    if (output.empty()) {
        if (!decompress) output = input + ".gz";
        else output = (ends_with(input, ".gz") && input.size() > 3) ? input.substr(0, input.size() - 3) : input + ".out";
    }
    struct stat st;
    if (!overwrite && stat(output.c_str(), &st) == 0) { bind_error(info, std::string(fn) + ": output '" + output + "' already exists (use overwrite := TRUE to replace)"); return; }
    dhts_ctx *c = dhts_create(device_id());
    if (!c) { bind_error(info, std::string(fn) + ": no MI355X (gfx950) device available; this build has no CPU fallback"); return; }
    int64_t nin = 0, nout = 0;
    const int rc = decompress ? dhts_bgunzip_file(c, input.c_str(), output.c_str(), &nin, &nout) : dhts_bgzip_file(c, input.c_str(), output.c_str(), (int)level, &nin, &nout);
    if (rc != 0) {
        std::string m = dhts_error(c);
        if (m.compare(0, strlen(fn), fn) != 0) m = std::string(fn) + ": " + m;
        dhts_destroy(c);
        bind_error(info, m); return;
    }
    dhts_destroy(c);
    if (!keep) unlink(input.c_str());
    OneRow *r = new OneRow(); r->four = true; r->path = output; r->bytes_in = nin; r->bytes_out = nout;
    finish_bind(info, r);
}
void bgzip_bind(duckdb_bind_info info) { bgzip_bind_common(info, false); }
void bgunzip_bind(duckdb_bind_info info) { bgzip_bind_common(info, true); }

// ---- index builders (src/hts_index_builder.c:111-325) ----
// kind 0: sam_index_build3 (sam.c:1029-1069), 1: bcf_index_build3 (vcf.c:4700-4742), 2: tbx_index_build3 with the VCF preset (tbx.c:526-541).
// Returns htslib's code: 0, -1 indexing failed, -2 cannot open, -3 format not indexable, -4 the index could not be saved.
struct TbxConfHost { int preset, sc, bc, ec, meta, skip; };
int build_index_file(int kind, const std::string &path, const std::string &index_path, int min_shift, bool *wrote_csi, const TbxConfHost *conf = nullptr) {
    dhts_ctx *c = dhts_create(device_id());
    if (!c) return -1;
    int rc = 0; int64_t n = -1; bool csi = false, compressed = false;
    if (dhts_open_path(c, path.c_str()) != 0) rc = kind == 2 ? -1 : -2;
    else if (dhts_bgzf_index(c) <= 0) rc = kind == 0 ? -3 : kind == 1 ? -3 : -2;       // not BGZF
    else if (kind == 2 && conf && (conf->preset & 0xffff) != 2) {                           // bed / gff / sam / custom columns: lines, not VCF records
        n = dhts_tabix_build_index(c, conf->preset, conf->sc, conf->bc, conf->ec, conf->meta, conf->skip, min_shift);
        csi = min_shift > 0; compressed = true; if (n < 0) rc = -1;
    }
    else if (kind == 0) {
        if (dhts_bam_open(c) != 0) rc = -3;                                                // SAM / CRAM: not read by this build
        else { n = dhts_bam_build_index_csi(c, min_shift); csi = min_shift > 0; compressed = csi; if (n < 0) rc = -1; }
    } else {
        if (dhts_bcf_open(c, 0) != 0) rc = -3;
        else {
            const int tk = dhts_bcf_is_text(c);                                              // 0 BCF, 1 bgzipped VCF text, 2 plain VCF text
            const bool text = tk == 1;
            if (tk == 2) rc = kind == 2 ? -2 : -3;                                          // not BGZF (tbx.c:533, vcf.c:4709)
            else if (kind == 2 && !text) rc = -1;                                                // tabix parses lines: a binary BCF is not one
            else if (!text && min_shift <= 0) rc = -1;                                      // "TBI indices for BCF files are not supported"
            else { n = dhts_bcf_build_index(c, min_shift); csi = min_shift > 0; compressed = true; if (n < 0) rc = -1; }
        }
    }
    if (rc == 0) {
        std::vector<uint8_t> raw((size_t)n);
        if (dhts_bam_index_bytes(c, raw.data(), (uint64_t)n) != 0) rc = -1;
        std::vector<uint8_t> file;
        if (rc == 0 && compressed) {                                                        // hts_idx_save_as: .csi / .tbi are BGZF files
            const int64_t bound = dhts_bgzf_compress(c, raw.data(), raw.size(), -1, nullptr, 0);
            file.resize((size_t)bound);
            const int64_t z = dhts_bgzf_compress(c, raw.data(), raw.size(), -1, file.data(), file.size());
            if (z < 0) rc = -4; else file.resize((size_t)z);
        } else file.swap(raw);
        if (rc == 0) {
            FILE *f = fopen(index_path.c_str(), "wb");
            if (!f || fwrite(file.data(), 1, file.size(), f) != file.size()) rc = -4;
            if (f && fclose(f) != 0) rc = -4;
        }
    }
    if (wrote_csi) *wrote_csi = csi;
    dhts_destroy(c);
    return rc;
}
void index_bind_common(duckdb_bind_info info, int kind) {
    const char *fn = kind == 0 ? "bam_index" : kind == 1 ? "bcf_index" : "tabix_index";
    const std::string path = positional_path(info);
    if (path.empty()) { bind_error(info, std::string(fn) + " requires a file path"); return; }
    int64_t min_shift = kind == 1 && ends_with(path, ".bcf") ? 14 : 0, threads = 4, v = 0;
    std::string index_path = take(named_varchar(info, "index_path"));
    TbxConfHost conf = {2, 1, 2, 0, '#', 0};                                           // tbx_conf_vcf (tbx.c:55)
    if (kind == 2) {
        const std::string preset = take(named_varchar(info, "preset"));
        if (preset.empty() || preset == "vcf") {}
        else if (preset == "bed") conf = {0x10000, 1, 2, 3, '#', 0};                   // tbx_conf_bed / _gff / _sam (tbx.c:43-52)
        else if (preset == "gff") conf = {0, 1, 4, 5, '#', 0};
        else if (preset == "sam") conf = {1, 3, 4, 0, '@', 0};
        else { bind_error(info, "tabix_index: preset must be one of vcf, bed, gff, sam"); return; }
        if (named_int(info, "seq_col", &v)) conf.sc = (int)v;                           // hts_index_builder.c:264-287
        if (named_int(info, "start_col", &v)) conf.bc = (int)v;
        if (named_int(info, "end_col", &v)) conf.ec = (int)v;
        const std::string cc = take(named_varchar(info, "comment_char"));
        if (!cc.empty()) conf.meta = (unsigned char)cc[0];
        if (named_int(info, "skip_lines", &v)) conf.skip = (int)v;
        if ((conf.preset & 0xffff) == 2 && (conf.sc != 1 || conf.bc != 2 || conf.ec != 0 || conf.meta != '#' || conf.skip != 0)) {
            bind_error(info, "tabix_index: the vcf preset with other columns / comment character / skipped lines is not supported by this build"); return;
        }
    }
    if (named_int(info, "min_shift", &v)) min_shift = v;
    (void)named_int(info, "threads", &threads);
    const bool cram = kind == 0 && ends_with(path, ".cram");
    const std::string target = !index_path.empty() ? index_path : path + (cram ? ".crai" : min_shift > 0 ? ".csi" : kind == 0 ? ".bai" : ".tbi");
    const int rc = build_index_file(kind, path, target, (int)min_shift, nullptr, kind == 2 ? &conf : nullptr);
    if (rc != 0) {
        char err[1024]; snprintf(err, sizeof(err), "%s: failed to build index for %s (error %d)", fn, path.c_str(), rc);
        bind_error(info, err); return;
    }
    OneRow *r = new OneRow(); r->path = target;
    r->format = cram ? "CRAI" : min_shift > 0 ? "CSI" : kind == 0 ? "BAI" : "TBI";
    finish_bind(info, r);
}
void bam_index_bind(duckdb_bind_info info) { index_bind_common(info, 0); }
void bcf_index_bind(duckdb_bind_info info) { index_bind_common(info, 1); }
void tabix_index_bind(duckdb_bind_info info) { index_bind_common(info, 2); }

struct Param { const char *name; int type; };
void register_one(duckdb_connection connection, const char *name, duckdb_table_function_bind_t bind, const std::vector<Param> &params) {
    duckdb_table_function tf = API(duckdb_table_function, duckdb_create_table_function, void)();
    API(void, duckdb_table_function_set_name, duckdb_table_function, const char *)(tf, name);
    auto mk = API(duckdb_logical_type, duckdb_create_logical_type, int);
    auto rm = API(void, duckdb_destroy_logical_type, duckdb_logical_type *);
    duckdb_logical_type tv = mk(DUCKDB_TYPE_VARCHAR);
    API(void, duckdb_table_function_add_parameter, duckdb_table_function, duckdb_logical_type)(tf, tv);
    rm(&tv);
    for (const Param &p : params) {
        duckdb_logical_type t = mk(p.type);
        API(void, duckdb_table_function_add_named_parameter, duckdb_table_function, const char *, duckdb_logical_type)(tf, p.name, t);
        rm(&t);
    }
    API(void, duckdb_table_function_set_bind, duckdb_table_function, duckdb_table_function_bind_t)(tf, bind);
    API(void, duckdb_table_function_set_init, duckdb_table_function, duckdb_table_function_init_t)(tf, row_init);
    API(void, duckdb_table_function_set_function, duckdb_table_function, duckdb_table_function_t)(tf, row_scan);
    API(duckdb_state, duckdb_register_table_function, duckdb_connection, duckdb_table_function)(connection, tf);
    API(void, duckdb_destroy_table_function, duckdb_table_function *)(&tf);
}
}  // namespace

extern "C" {
__attribute__((visibility("default"))) void register_bgzip_function(duckdb_connection connection) {                                 // bgzip.c:336-357
    register_one(connection, "bgzip", bgzip_bind, {{"output_path", DUCKDB_TYPE_VARCHAR}, {"threads", DUCKDB_TYPE_INTEGER}, {"level", DUCKDB_TYPE_INTEGER}, {"keep", DUCKDB_TYPE_BOOLEAN}, {"overwrite", DUCKDB_TYPE_BOOLEAN}});
}
__attribute__((visibility("default"))) void register_bgunzip_function(duckdb_connection connection) {                               // bgzip.c:359-380
    register_one(connection, "bgunzip", bgunzip_bind, {{"output_path", DUCKDB_TYPE_VARCHAR}, {"threads", DUCKDB_TYPE_INTEGER}, {"keep", DUCKDB_TYPE_BOOLEAN}, {"overwrite", DUCKDB_TYPE_BOOLEAN}});
}
__attribute__((visibility("default"))) void register_bam_index_function(duckdb_connection connection) {                             // hts_index_builder.c:326-344
    register_one(connection, "bam_index", bam_index_bind, {{"index_path", DUCKDB_TYPE_VARCHAR}, {"min_shift", DUCKDB_TYPE_INTEGER}, {"threads", DUCKDB_TYPE_INTEGER}});
}
__attribute__((visibility("default"))) void register_bcf_index_function(duckdb_connection connection) {                             // hts_index_builder.c:346-364
    register_one(connection, "bcf_index", bcf_index_bind, {{"index_path", DUCKDB_TYPE_VARCHAR}, {"min_shift", DUCKDB_TYPE_INTEGER}, {"threads", DUCKDB_TYPE_INTEGER}});
}
__attribute__((visibility("default"))) void register_tabix_index_function(duckdb_connection connection) {                           // hts_index_builder.c:366-390
    register_one(connection, "tabix_index", tabix_index_bind, {{"preset", DUCKDB_TYPE_VARCHAR}, {"index_path", DUCKDB_TYPE_VARCHAR}, {"min_shift", DUCKDB_TYPE_INTEGER}, {"threads", DUCKDB_TYPE_INTEGER},
                 {"seq_col", DUCKDB_TYPE_INTEGER}, {"start_col", DUCKDB_TYPE_INTEGER}, {"end_col", DUCKDB_TYPE_INTEGER}, {"comment_char", DUCKDB_TYPE_VARCHAR}, {"skip_lines", DUCKDB_TYPE_INTEGER}});
}
}
