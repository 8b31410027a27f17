/*
 * minihost.c -- a minimal in-process stand-in for the DuckDB engine side of the C extension API (test tooling).
 *
 * It hands the extension a duckdb_ext_api_v1-shaped table (slot numbers from include/duckdb_abi_slots.h), lets
 * duckhts_init_c_api register its table functions, then drives bind -> init -> local_init -> scan (until a 0-row
 * chunk) for one call and serialises every DataChunk canonically (SURVEY.md 8(c) "Parity definition"):
 *   per chunk: u64 n_rows; per projected column: u32 type id, validity words (NULL pointer = all ones, tail bits
 *   masked), then the payload: fixed width = n x width bytes (invalid rows zeroed); VARCHAR = per row u32 len
 *   (0xFFFFFFFF for NULL) + bytes; LIST = n x (u64 offset, u64 length) exactly as written, u64 child size, u32 child type,
 *   then the child payload in the scalar encoding and the child's validity words (NULL elements: the VEP_* columns of read_bcf).
 * usage: minihost [--direct] <ext.so> <function> <path> [-n name=value]... [-p 0,3,5] [-o out.bin]
 * --direct plays a reference-built src/duckhts.c (src/duckhts.c:13-16,48-55): it fills the extension's `duckdb_ext_api`
 * global the way DUCKDB_EXTENSION_API_INIT does and calls register_read_bcf_function / register_read_bam_function
 * itself, never touching duckhts_init_c_api.
 */
#include <dlfcn.h>
#include <pthread.h>
#include <time.h>
#include <stdbool.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/duckhts_extension.h"

#define VSIZE 2048

typedef struct LType { int id; struct LType *child; } LType;
typedef struct Value { int is_null; int is_bool; int b; char *s; } Value;
typedef struct Vec { int type; void *data; uint64_t *validity; char **heap; size_t nheap, heap_used, heap_cap; int child_type; struct Vec *child; idx_t list_size, list_cap; struct Vec *kids[2]; idx_t vcap; /* rows the validity mask covers: a list child grows with the list, 0 = VSIZE */ } Vec;
typedef struct Chunk { Vec *vecs; size_t ncol; idx_t size; } Chunk;
typedef struct TF {
    char name[64]; duckdb_table_function_bind_t bind; duckdb_table_function_init_t init, local_init; duckdb_table_function_t func;
    char named[16][32]; int named_type[16]; int n_named; int pushdown;
} TF;
typedef struct Bind { TF *tf; const char *path; char names[16][32]; char vals[16][512]; int n_named; char colname[1024][256]; int coltype[1024]; int colchild[1024]; int ncol;
                      void *bind_data; duckdb_delete_callback_t bind_del; char err[1024]; int has_err; } Bind;
typedef struct Init { Bind *b; idx_t proj[1024]; idx_t nproj; void *data; duckdb_delete_callback_t del; idx_t max_threads; char err[1024]; int has_err; } Init;
typedef struct Func { Bind *b; Init *g, *l; char err[1024]; int has_err; } Func;

static TF g_tfs[16]; static int g_ntf = 0;
static void *g_api[DUCKDB_ABI_V120_NSLOTS];

static void *h_malloc(size_t n) { return malloc(n); }
static void h_free(void *p) { free(p); }
static idx_t h_vector_size(void) { return VSIZE; }
static duckdb_state h_connect(duckdb_database db, duckdb_connection *out) { (void)db; *out = (void *)0x1; return DuckDBSuccess; }
static void h_disconnect(duckdb_connection *c) { *c = NULL; }
static duckdb_logical_type h_create_logical_type(int id) { LType *t = calloc(1, sizeof(LType)); t->id = id; return t; }
static duckdb_logical_type h_create_list_type(duckdb_logical_type c) { LType *t = calloc(1, sizeof(LType)); t->id = DUCKDB_TYPE_LIST; t->child = calloc(1, sizeof(LType)); t->child->id = ((LType *)c)->id; return t; }
static duckdb_logical_type h_create_map_type(duckdb_logical_type k, duckdb_logical_type v) { (void)k; (void)v; LType *t = calloc(1, sizeof(LType)); t->id = DUCKDB_TYPE_MAP; return t; }
static void h_destroy_logical_type(duckdb_logical_type *t) { if (t && *t) { free(((LType *)*t)->child); free(*t); *t = NULL; } }
static char *h_get_varchar(duckdb_value v) { Value *x = v; if (!x || !x->s) return NULL; char *r = malloc(strlen(x->s) + 1); strcpy(r, x->s); return r; }
static bool h_get_bool(duckdb_value v) { Value *x = v; return x && x->b; }
static int64_t h_get_int64(duckdb_value v) { Value *x = v; return (x && x->s) ? strtoll(x->s, NULL, 10) : 0; }
static bool h_is_null_value(duckdb_value v) { Value *x = v; return !x || x->is_null; }
static void h_destroy_value(duckdb_value *v) { if (v && *v) { Value *x = *v; free(x->s); free(x); *v = NULL; } }

static duckdb_table_function h_create_table_function(void) { TF *t = &g_tfs[g_ntf++]; memset(t, 0, sizeof(*t)); return t; }
static void h_destroy_table_function(duckdb_table_function *t) { (void)t; }
static void h_tf_set_name(duckdb_table_function t, const char *n) { snprintf(((TF *)t)->name, 64, "%s", n); }
static void h_tf_add_parameter(duckdb_table_function t, duckdb_logical_type ty) { (void)t; (void)ty; }
static void h_tf_add_named(duckdb_table_function t, const char *n, duckdb_logical_type ty) { TF *f = t; snprintf(f->named[f->n_named], 32, "%s", n); f->named_type[f->n_named++] = ((LType *)ty)->id; }
static void h_tf_set_bind(duckdb_table_function t, duckdb_table_function_bind_t f) { ((TF *)t)->bind = f; }
static void h_tf_set_init(duckdb_table_function t, duckdb_table_function_init_t f) { ((TF *)t)->init = f; }
static void h_tf_set_local_init(duckdb_table_function t, duckdb_table_function_init_t f) { ((TF *)t)->local_init = f; }
static void h_tf_set_function(duckdb_table_function t, duckdb_table_function_t f) { ((TF *)t)->func = f; }
static void h_tf_pushdown(duckdb_table_function t, bool b) { ((TF *)t)->pushdown = b; }
static duckdb_state h_register_tf(duckdb_connection c, duckdb_table_function t) { (void)c; (void)t; return DuckDBSuccess; }

static duckdb_value h_bind_get_parameter(duckdb_bind_info i, idx_t k) { Bind *b = i; (void)k; Value *v = calloc(1, sizeof(Value)); v->s = strdup(b->path); return v; }
static idx_t h_bind_get_parameter_count(duckdb_bind_info i) { (void)i; return 1; }
static duckdb_value h_bind_get_named(duckdb_bind_info i, const char *name) {
    Bind *b = i;
    for (int k = 0; k < b->n_named; k++) if (!strcmp(b->names[k], name)) {
        Value *v = calloc(1, sizeof(Value)); v->s = strdup(b->vals[k]); v->b = (!strcmp(b->vals[k], "true") || !strcmp(b->vals[k], "1")); return v; }
    return NULL;                                   /* unset named parameter */
}
static void h_bind_add_result_column(duckdb_bind_info i, const char *n, duckdb_logical_type t) {
    Bind *b = i; LType *lt = t;
    if (b->ncol >= 1024) { if (!b->has_err) { snprintf(b->err, sizeof(b->err), "mini host: more than 1024 result columns"); b->has_err = 1; } return; }   /* (a limit of this test harness, not of the extension) */
    snprintf(b->colname[b->ncol], 256, "%s", n); b->colchild[b->ncol] = lt->child ? lt->child->id : 0; b->coltype[b->ncol++] = lt->id; }
static void h_bind_set_bind_data(duckdb_bind_info i, void *d, duckdb_delete_callback_t del) { Bind *b = i; b->bind_data = d; b->bind_del = del; }
static void h_bind_set_error(duckdb_bind_info i, const char *e) { Bind *b = i; snprintf(b->err, sizeof(b->err), "%s", e); b->has_err = 1; }
static void *h_init_get_bind_data(duckdb_init_info i) { return ((Init *)i)->b->bind_data; }
static void h_init_set_init_data(duckdb_init_info i, void *d, duckdb_delete_callback_t del) { Init *x = i; x->data = d; x->del = del; }
static idx_t h_init_get_column_count(duckdb_init_info i) { return ((Init *)i)->nproj; }
static idx_t h_init_get_column_index(duckdb_init_info i, idx_t k) { return ((Init *)i)->proj[k]; }
static void h_init_set_max_threads(duckdb_init_info i, idx_t n) { ((Init *)i)->max_threads = n; }
static void h_init_set_error(duckdb_init_info i, const char *e) { Init *x = i; snprintf(x->err, sizeof(x->err), "%s", e); x->has_err = 1; }
static void *h_func_get_bind_data(duckdb_function_info i) { return ((Func *)i)->b->bind_data; }
static void *h_func_get_init_data(duckdb_function_info i) { return ((Func *)i)->g->data; }
static void *h_func_get_local_init_data(duckdb_function_info i) { return ((Func *)i)->l->data; }
static void h_func_set_error(duckdb_function_info i, const char *e) { Func *x = i; snprintf(x->err, sizeof(x->err), "%s", e); x->has_err = 1; }

static duckdb_vector h_chunk_get_vector(duckdb_data_chunk c, idx_t k) { return &((Chunk *)c)->vecs[k]; }
static void h_chunk_set_size(duckdb_data_chunk c, idx_t n) { ((Chunk *)c)->size = n; }
static idx_t h_chunk_get_size(duckdb_data_chunk c) { return ((Chunk *)c)->size; }
static void *h_vector_get_data(duckdb_vector v) { return ((Vec *)v)->data; }
static uint64_t *h_vector_get_validity(duckdb_vector v) { return ((Vec *)v)->validity; }
static void h_vector_ensure_validity_writable(duckdb_vector v) { Vec *x = v; if (!x->validity) { size_t nb = (size_t)(x->vcap ? x->vcap : VSIZE) / 8; x->validity = malloc(nb); memset(x->validity, 0xff, nb); } }
static void h_validity_set_row_invalid(uint64_t *val, idx_t row) { val[row / 64] &= ~((uint64_t)1 << (row % 64)); }
static void h_assign_len(duckdb_vector v, idx_t row, const char *s, idx_t len) {
    Vec *x = v; duckdb_string_t *d = (duckdb_string_t *)x->data + row;
    memset(d, 0, sizeof(*d));
    d->value.inlined.length = (uint32_t)len;
    if (len <= 12) memcpy(d->value.inlined.inlined, s, len);
    else {
        /* the vector's string heap is an arena (slabs of 256 KiB), like the engine's StringHeap: one bump allocation + one copy per string */
        if (x->nheap == 0 || x->heap_used + len > x->heap_cap) {
            size_t cap = len > (256u << 10) ? len : (256u << 10);
            x->heap = realloc(x->heap, (x->nheap + 1) * sizeof(char *)); x->heap[x->nheap++] = malloc(cap); x->heap_used = 0; x->heap_cap = cap;
        }
        char *h = x->heap[x->nheap - 1] + x->heap_used; x->heap_used += len;
        memcpy(h, s, len); memcpy(d->value.pointer.prefix, s, 4); d->value.pointer.ptr = h;
    }
}
static void h_assign(duckdb_vector v, idx_t row, const char *s) { h_assign_len(v, row, s, strlen(s)); }

static int type_width(int t) {
    switch (t) { case DUCKDB_TYPE_BOOLEAN: return 1; case DUCKDB_TYPE_USMALLINT: return 2; case DUCKDB_TYPE_INTEGER: case DUCKDB_TYPE_FLOAT: return 4;
    case DUCKDB_TYPE_BIGINT: case DUCKDB_TYPE_DOUBLE: return 8; case DUCKDB_TYPE_VARCHAR: return 16; default: return 16; }
}

/* LIST vectors: the child grows on reserve / set_size (the readers call set_size without reserve on one path, bcf_reader.c:1446) */
static void list_grow(Vec *x, idx_t need) {
    if (need <= x->list_cap) return;
    idx_t nc = x->list_cap ? x->list_cap : VSIZE; while (nc < need) nc *= 2;
    if (x->child->type == DUCKDB_TYPE_STRUCT) {                 /* MAP child = STRUCT{key VARCHAR, value VARCHAR} */
        for (int k = 0; k < 2; k++) { Vec *c = x->child->kids[k]; c->data = realloc(c->data, nc * 16); memset((char *)c->data + x->list_cap * 16, 0, (nc - x->list_cap) * 16); }
    } else {
        size_t w = (size_t)type_width(x->child->type);
        x->child->data = realloc(x->child->data, nc * w); memset((char *)x->child->data + x->list_cap * w, 0, (nc - x->list_cap) * w);
        /* the child's validity mask (NULL elements: VEP_* columns) covers the child's capacity */
        if (x->child->validity) { x->child->validity = realloc(x->child->validity, nc / 8); memset((char *)x->child->validity + x->list_cap / 8, 0xff, (nc - x->list_cap) / 8); }
        x->child->vcap = nc;
    }
    x->list_cap = nc;
}
static duckdb_vector h_struct_get_child(duckdb_vector v, idx_t k) { return ((Vec *)v)->kids[k]; }
static idx_t h_list_get_size(duckdb_vector v) { return ((Vec *)v)->list_size; }
static duckdb_state h_list_reserve(duckdb_vector v, idx_t cap) { list_grow(v, cap); return DuckDBSuccess; }
static duckdb_state h_list_set_size(duckdb_vector v, idx_t n) { list_grow(v, n); ((Vec *)v)->list_size = n; return DuckDBSuccess; }
static duckdb_vector h_list_get_child(duckdb_vector v) { return ((Vec *)v)->child; }

static const void *get_api(duckdb_extension_info info, const char *version) { (void)info; return strcmp(version, "v1.2.0") == 0 ? g_api : NULL; }
static duckdb_database g_db = (void *)0x2;
static duckdb_database *get_database(duckdb_extension_info info) { (void)info; return &g_db; }
static void set_error(duckdb_extension_info info, const char *e) { (void)info; fprintf(stderr, "extension error: %s\n", e); }

/* ---- scan workers: what DuckDB's task scheduler does with a parallel table function: every worker runs local_init once, then calls
 * the scan function until it returns a 0-row chunk.  Chunks are serialised under a lock in arrival order. ---- */
typedef struct Worker { Bind *b; Init *g; TF *tf; FILE *fo; Init l; uint64_t rows, chunks; int failed; char err[1024]; } Worker;
static pthread_mutex_t g_out_mu = PTHREAD_MUTEX_INITIALIZER;

static void write_chunk(FILE *fo, Chunk *cp, uint64_t n) {
    Chunk c = *cp;
    fwrite(&n, 8, 1, fo);
    for (size_t k = 0; k < c.ncol; k++) {
        Vec *v = &c.vecs[k]; uint32_t t = (uint32_t)v->type; fwrite(&t, 4, 1, fo);
        uint64_t words = (n + 63) / 64;
        for (uint64_t w = 0; w < words; w++) { uint64_t m = v->validity ? v->validity[w] : ~0ull; if (w == words - 1 && (n % 64)) m &= (1ull << (n % 64)) - 1; fwrite(&m, 8, 1, fo); }
        if (v->type == DUCKDB_TYPE_MAP) {                 /* entries, child size, then keys and values in the VARCHAR child encoding */
            fwrite(v->data, 16, n, fo);
            uint64_t cn = v->list_size; fwrite(&cn, 8, 1, fo);
            for (int q = 0; q < 2; q++) for (uint64_t r = 0; r < cn; r++) {
                duckdb_string_t *d = (duckdb_string_t *)v->child->kids[q]->data + r; uint32_t len = d->value.inlined.length; fwrite(&len, 4, 1, fo);
                fwrite(len <= 12 ? d->value.inlined.inlined : d->value.pointer.ptr, 1, len, fo);
            }
            continue;
        }
        if (v->type == DUCKDB_TYPE_LIST) {
            fwrite(v->data, 16, n, fo);
            uint64_t cn = v->list_size; uint32_t ct = (uint32_t)v->child->type; fwrite(&cn, 8, 1, fo); fwrite(&ct, 4, 1, fo);
            for (uint64_t r = 0; r < cn; r++) {
                if (ct == DUCKDB_TYPE_VARCHAR) { duckdb_string_t *d = (duckdb_string_t *)v->child->data + r; uint32_t len = d->value.inlined.length; fwrite(&len, 4, 1, fo); fwrite(len <= 12 ? d->value.inlined.inlined : d->value.pointer.ptr, 1, len, fo); }
                else fwrite((char *)v->child->data + r * type_width((int)ct), 1, (size_t)type_width((int)ct), fo);
            }
            /* child validity words (all ones when the reader never touched the mask) */
            for (uint64_t w = 0, cw = (cn + 63) / 64; w < cw; w++) { uint64_t m = v->child->validity ? v->child->validity[w] : ~0ull; if (w == cw - 1 && (cn % 64)) m &= (1ull << (cn % 64)) - 1; fwrite(&m, 8, 1, fo); }
            continue;
        }
        for (uint64_t r = 0; r < n; r++) {
            int valid = v->validity ? (int)((v->validity[r / 64] >> (r % 64)) & 1) : 1;
            if (v->type == DUCKDB_TYPE_VARCHAR) {
                duckdb_string_t *d = (duckdb_string_t *)v->data + r; uint32_t len = valid ? d->value.inlined.length : 0xFFFFFFFFu; fwrite(&len, 4, 1, fo);
                if (valid) fwrite(len <= 12 ? d->value.inlined.inlined : d->value.pointer.ptr, 1, len, fo);
            } else { int w = type_width(v->type); static const char zero[16] = {0}; fwrite(valid ? (char *)v->data + r * w : zero, 1, (size_t)w, fo); }
        }
    }
}

static void vec_release(Vec *v) { for (size_t h = 0; h < v->nheap; h++) free(v->heap[h]); free(v->heap); v->heap = NULL; v->nheap = 0; v->heap_used = v->heap_cap = 0; }

static uint64_t g_limit = 0;          /* -l N: stop after N rows (what a LIMIT does to a table function) */
static void *worker_main(void *arg) {
    Worker *w = arg; Bind *b = w->b; Init *g = w->g; TF *tf = w->tf;
    memset(&w->l, 0, sizeof(w->l)); w->l.b = b; memcpy(w->l.proj, g->proj, sizeof(g->proj)); w->l.nproj = g->nproj;
    if (tf->local_init) tf->local_init(&w->l);
    if (w->l.has_err) { w->failed = 1; snprintf(w->err, sizeof(w->err), "%s", w->l.err); return NULL; }
    Func fi; memset(&fi, 0, sizeof(fi)); fi.b = b; fi.g = g; fi.l = &w->l;
    /* the DataChunk is allocated once and reset between calls, as the engine does */
    Chunk c; c.ncol = g->nproj; c.size = 0; c.vecs = calloc(c.ncol, sizeof(Vec));
    for (size_t k = 0; k < c.ncol; k++) { int t = g->proj[k] < (idx_t)b->ncol ? b->coltype[g->proj[k]] : DUCKDB_TYPE_BIGINT; c.vecs[k].type = t; c.vecs[k].data = calloc(VSIZE, (size_t)type_width(t));
        if (t == DUCKDB_TYPE_LIST) { Vec *ch = calloc(1, sizeof(Vec)); ch->type = b->colchild[g->proj[k]]; c.vecs[k].child = ch; c.vecs[k].child_type = ch->type; list_grow(&c.vecs[k], VSIZE); }
        if (t == DUCKDB_TYPE_MAP) {
            Vec *ch = calloc(1, sizeof(Vec)); ch->type = DUCKDB_TYPE_STRUCT;
            for (int q = 0; q < 2; q++) { ch->kids[q] = calloc(1, sizeof(Vec)); ch->kids[q]->type = DUCKDB_TYPE_VARCHAR; }
            c.vecs[k].child = ch; c.vecs[k].child_type = DUCKDB_TYPE_STRUCT; list_grow(&c.vecs[k], VSIZE);
        } }
    for (;;) {
        c.size = 0;
        tf->func(&fi, &c);
        if (fi.has_err) { w->failed = 2; snprintf(w->err, sizeof(w->err), "%s", fi.err); break; }
        uint64_t n = c.size;
        if (w->fo && n) { pthread_mutex_lock(&g_out_mu); write_chunk(w->fo, &c, n); pthread_mutex_unlock(&g_out_mu); }
        /* reset: validity masks dropped, string heaps released, lists emptied */
        for (size_t k = 0; k < c.ncol; k++) {
            Vec *v = &c.vecs[k];
            free(v->validity); v->validity = NULL; vec_release(v);
            if (v->type == DUCKDB_TYPE_LIST || v->type == DUCKDB_TYPE_MAP) memset(v->data, 0, (size_t)VSIZE * 16);    /* entries of NULL rows are left untouched by the readers: keep the dump deterministic */
            if (v->child) { v->list_size = 0; if (v->child->type == DUCKDB_TYPE_STRUCT) { for (int q = 0; q < 2; q++) vec_release(v->child->kids[q]); } else { vec_release(v->child); free(v->child->validity); v->child->validity = NULL; } }
        }
        if (n == 0) break;
        if (g_limit && (w->rows + n) >= g_limit) { w->rows += n; w->chunks++; break; }      /* LIMIT: the engine stops pulling and tears the scan down */
        w->rows += n; w->chunks++;
    }
    for (size_t k = 0; k < c.ncol; k++) {
        Vec *ch = c.vecs[k].child;
        if (ch && ch->type == DUCKDB_TYPE_STRUCT) { for (int q = 0; q < 2; q++) { Vec *kv = ch->kids[q]; vec_release(kv); free(kv->data); free(kv); } free(ch); ch = NULL; }
        if (ch) { vec_release(ch); free(ch->data); free(ch->validity); free(ch); }
        vec_release(&c.vecs[k]); free(c.vecs[k].data); free(c.vecs[k].validity); }
    free(c.vecs);
    return NULL;
}

int main(int argc, char **argv) {
    int direct = 0;
    if (argc > 1 && !strcmp(argv[1], "--direct")) { direct = 1; argv++; argc--; }
    if (argc < 4) { fprintf(stderr, "usage: minihost ext.so function path [-n k=v] [-p cols] [-o out]\n"); return 2; }
#define SET(name, fn) g_api[SLOT_##name] = (void *)(fn)
    SET(duckdb_malloc, h_malloc); SET(duckdb_free, h_free); SET(duckdb_vector_size, h_vector_size); SET(duckdb_connect, h_connect); SET(duckdb_disconnect, h_disconnect);
    SET(duckdb_create_logical_type, h_create_logical_type); SET(duckdb_create_list_type, h_create_list_type); SET(duckdb_create_map_type, h_create_map_type);
    SET(duckdb_destroy_logical_type, h_destroy_logical_type); SET(duckdb_get_varchar, h_get_varchar); SET(duckdb_get_bool, h_get_bool); SET(duckdb_get_int64, h_get_int64); SET(duckdb_is_null_value, h_is_null_value);
    SET(duckdb_destroy_value, h_destroy_value); SET(duckdb_create_table_function, h_create_table_function); SET(duckdb_destroy_table_function, h_destroy_table_function);
    SET(duckdb_table_function_set_name, h_tf_set_name); SET(duckdb_table_function_add_parameter, h_tf_add_parameter); SET(duckdb_table_function_add_named_parameter, h_tf_add_named);
    SET(duckdb_table_function_set_bind, h_tf_set_bind); SET(duckdb_table_function_set_init, h_tf_set_init); SET(duckdb_table_function_set_local_init, h_tf_set_local_init);
    SET(duckdb_table_function_set_function, h_tf_set_function); SET(duckdb_table_function_supports_projection_pushdown, h_tf_pushdown); SET(duckdb_register_table_function, h_register_tf);
    SET(duckdb_bind_get_parameter, h_bind_get_parameter); SET(duckdb_bind_get_parameter_count, h_bind_get_parameter_count); SET(duckdb_bind_get_named_parameter, h_bind_get_named);
    SET(duckdb_bind_add_result_column, h_bind_add_result_column); SET(duckdb_bind_set_bind_data, h_bind_set_bind_data); SET(duckdb_bind_set_error, h_bind_set_error);
    SET(duckdb_init_get_bind_data, h_init_get_bind_data); SET(duckdb_init_set_init_data, h_init_set_init_data); SET(duckdb_init_get_column_count, h_init_get_column_count);
    SET(duckdb_init_get_column_index, h_init_get_column_index); SET(duckdb_init_set_max_threads, h_init_set_max_threads); SET(duckdb_init_set_error, h_init_set_error);
    SET(duckdb_function_get_bind_data, h_func_get_bind_data); SET(duckdb_function_get_init_data, h_func_get_init_data); SET(duckdb_function_get_local_init_data, h_func_get_local_init_data);
    SET(duckdb_function_set_error, h_func_set_error); SET(duckdb_data_chunk_get_vector, h_chunk_get_vector); SET(duckdb_data_chunk_set_size, h_chunk_set_size);
    SET(duckdb_data_chunk_get_size, h_chunk_get_size); SET(duckdb_vector_get_data, h_vector_get_data); SET(duckdb_vector_get_validity, h_vector_get_validity);
    SET(duckdb_vector_ensure_validity_writable, h_vector_ensure_validity_writable); SET(duckdb_validity_set_row_invalid, h_validity_set_row_invalid);
    SET(duckdb_vector_assign_string_element, h_assign); SET(duckdb_vector_assign_string_element_len, h_assign_len);
    SET(duckdb_list_vector_get_size, h_list_get_size); SET(duckdb_list_vector_reserve, h_list_reserve); SET(duckdb_list_vector_set_size, h_list_set_size);
    SET(duckdb_list_vector_get_child, h_list_get_child); SET(duckdb_struct_vector_get_child, h_struct_get_child);

    void *so = dlopen(argv[1], RTLD_NOW);
    if (!so) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    struct duckdb_extension_access acc = { set_error, get_database, get_api };
    if (direct) {
        void **tab = dlsym(so, "duckdb_ext_api");
        void (*reg_bcf)(duckdb_connection) = dlsym(so, "register_read_bcf_function");
        void (*reg_bam)(duckdb_connection) = dlsym(so, "register_read_bam_function");
        if (!tab || !reg_bcf || !reg_bam) { fprintf(stderr, "missing duckdb_ext_api / register_read_*_function\n"); return 2; }
        memcpy(tab, g_api, sizeof(g_api));                 /* duckdb_ext_api = *res */
        duckdb_connection conn = NULL; h_connect(g_db, &conn);
        reg_bcf(conn); reg_bam(conn);                      /* src/duckhts.c:54-55 */
        static const char *more[] = {"register_bgzip_function", "register_bgunzip_function", "register_bam_index_function", "register_bcf_index_function", "register_tabix_index_function"};
        for (int k = 0; k < 5; k++) { void (*reg)(duckdb_connection) = dlsym(so, more[k]); if (reg) reg(conn); }     /* src/duckhts.c:61-65 */
        h_disconnect(&conn);
    } else {
        bool (*entry)(duckdb_extension_info, struct duckdb_extension_access *) = dlsym(so, "duckhts_init_c_api");
        if (!entry) { fprintf(stderr, "no duckhts_init_c_api\n"); return 2; }
        if (!entry((void *)0x3, &acc)) { fprintf(stderr, "entrypoint returned false\n"); return 2; }
    }
    if (!strcmp(argv[2], "--catalog")) {                   /* list what got registered, no scan */
        for (int i = 0; i < g_ntf; i++) {
            printf("TF %s pushdown=%d bind=%d init=%d local_init=%d func=%d named=", g_tfs[i].name, g_tfs[i].pushdown, g_tfs[i].bind != NULL, g_tfs[i].init != NULL, g_tfs[i].local_init != NULL, g_tfs[i].func != NULL);
            for (int k = 0; k < g_tfs[i].n_named; k++) printf("%s%s:%d", k ? "," : "", g_tfs[i].named[k], g_tfs[i].named_type[k]);
            printf("\n");
        }
        return 0;
    }
    TF *tf = NULL;
    for (int i = 0; i < g_ntf; i++) if (!strcmp(g_tfs[i].name, argv[2])) tf = &g_tfs[i];
    if (!tf) { printf("ERROR catalog: table function %s not registered\n", argv[2]); return 3; }

    const char *proj = NULL, *out = NULL;
    int threads = 0, repeat = 1;
    static char names[16][32], vals[16][512]; int n_named = 0;
    for (int i = 4; i < argc; i++) {
        if (!strcmp(argv[i], "-n") && i + 1 < argc) {
            char *eq = strchr(argv[++i], '='); if (!eq) continue;
            int known = 0; *eq = 0;
            for (int k = 0; k < tf->n_named; k++) if (!strcmp(tf->named[k], argv[i])) known = 1;
            if (!known) { printf("ERROR binder: unknown named parameter %s\n", argv[i]); return 3; }
            snprintf(names[n_named], 32, "%s", argv[i]); snprintf(vals[n_named++], 512, "%s", eq + 1);
        } else if (!strcmp(argv[i], "-p") && i + 1 < argc) proj = argv[++i];
        else if (!strcmp(argv[i], "-o") && i + 1 < argc) out = argv[++i];
        else if (!strcmp(argv[i], "-t") && i + 1 < argc) threads = atoi(argv[++i]);       /* worker threads offered to the scan (<= max_threads it asks for) */
        else if (!strcmp(argv[i], "-l") && i + 1 < argc) g_limit = strtoull(argv[++i], NULL, 10);
        else if (!strcmp(argv[i], "-r") && i + 1 < argc) repeat = atoi(argv[++i]);        /* run the query this many times in one process (warm runs) */
    }
    for (int rep = 0; rep < repeat; rep++) {
        struct timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
        static Bind b; memset(&b, 0, sizeof(b)); b.tf = tf; b.path = argv[3];
        memcpy(b.names, names, sizeof(names)); memcpy(b.vals, vals, sizeof(vals)); b.n_named = n_named;
        tf->bind(&b);
        if (b.has_err) { printf("ERROR bind: %s\n", b.err); return 3; }
        static Init g; memset(&g, 0, sizeof(g)); g.b = &b;
        if (proj) { char *dup = strdup(proj); for (char *t = strtok(dup, ","); t; t = strtok(NULL, ",")) g.proj[g.nproj++] = (idx_t)strtoull(t, NULL, 10); free(dup); }
        else for (int i = 0; i < b.ncol; i++) g.proj[g.nproj++] = (idx_t)i;
        if (tf->init) tf->init(&g);
        if (g.has_err) { printf("ERROR init: %s\n", g.err); if (b.bind_del) b.bind_del(b.bind_data); return 3; }
        int nthr = threads > 0 ? threads : 1; if ((idx_t)nthr > g.max_threads && g.max_threads > 0) nthr = (int)g.max_threads;
        FILE *fo = (out && rep == 0) ? fopen(out, "wb") : NULL;
        /* schema record */
        if (fo) { uint32_t nc = (uint32_t)b.ncol; fwrite(&nc, 4, 1, fo); for (int i = 0; i < b.ncol; i++) { uint32_t t = (uint32_t)b.coltype[i]; fwrite(&t, 4, 1, fo); uint32_t ct = (uint32_t)b.colchild[i]; fwrite(&ct, 4, 1, fo); fwrite(b.colname[i], 1, 256, fo); } uint32_t np = (uint32_t)g.nproj; fwrite(&np, 4, 1, fo); }
        static Worker w[64]; memset(w, 0, sizeof(w));
        pthread_t th[64];
        for (int k = 0; k < nthr; k++) { w[k].b = &b; w[k].g = &g; w[k].tf = tf; w[k].fo = fo; }
        for (int k = 1; k < nthr; k++) pthread_create(&th[k], NULL, worker_main, &w[k]);
        worker_main(&w[0]);
        for (int k = 1; k < nthr; k++) pthread_join(th[k], NULL);
        uint64_t total = 0, chunks = 0; int failed = 0;
        for (int k = 0; k < nthr; k++) { total += w[k].rows; chunks += w[k].chunks; if (w[k].failed) { if (!failed) printf("ERROR %s: %s\n", w[k].failed == 1 ? "init" : "scan", w[k].err); failed = 1; } }
        if (fo) fclose(fo);
        for (int k = 0; k < nthr; k++) if (w[k].l.del) w[k].l.del(w[k].l.data);
        if (g.del) g.del(g.data);
        if (b.bind_del) b.bind_del(b.bind_data);
        if (failed) return 3;
        clock_gettime(CLOCK_MONOTONIC, &t1);
        const double sec = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
        if (repeat > 1) printf("RUN %d seconds=%.6f rows=%llu threads=%d\n", rep, sec, (unsigned long long)total, nthr);
        if (rep == repeat - 1) printf("OK rows=%llu chunks=%llu columns=%d max_threads=%llu\n", (unsigned long long)total, (unsigned long long)chunks, b.ncol, (unsigned long long)g.max_threads);
    }
    return 0;
}
