/* orc_cols.h -- column containers + canonical serialisation shared by the oracle's table-producing functions
 * (TEST INFRASTRUCTURE ONLY, see dhts_oracle.h). */
#ifndef ORC_COLS_H
#define ORC_COLS_H
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ small buffers ------------------------------ */
typedef struct { uint8_t *p; size_t n, cap; } buf_t;
static inline void buf_push(buf_t *b, const void *src, size_t n)
{
    if (b->n + n > b->cap) {
        size_t nc = b->cap ? b->cap * 2 : 256;
        while (nc < b->n + n) nc *= 2;
        b->p = (uint8_t *)realloc(b->p, nc);
        b->cap = nc;
    }
    if (n) memcpy(b->p + b->n, src, n);
    b->n += n;
}
static inline void buf_u64(buf_t *b, uint64_t v) { buf_push(b, &v, 8); }
static inline void buf_u8(buf_t *b, uint8_t v) { buf_push(b, &v, 1); }

enum { T_VARCHAR = 1, T_BIGINT, T_DOUBLE, T_BOOLEAN, T_INTEGER, T_FLOAT };

typedef struct {
    char name[640];
    int type, is_list;
    int64_t n;
    buf_t valid, fixed, soff, sbytes, lent;
    uint64_t child_n;
    buf_t cfixed, csoff, csbytes;
    int has_cvalid; buf_t cvalid;            /* lists with NULL elements (VEP_* columns): one byte per child, serialised with is_list = 2 */
} col_t;

static inline void col_init(col_t *c, const char *name, int type, int is_list)
{
    memset(c, 0, sizeof *c);
    snprintf(c->name, sizeof c->name, "%s", name);
    c->type = type; c->is_list = is_list;
    if (!is_list && type == T_VARCHAR) buf_u64(&c->soff, 0);
    if (is_list && type == T_VARCHAR) buf_u64(&c->csoff, 0);
}
static inline void col_free(col_t *c)
{
    free(c->valid.p); free(c->fixed.p); free(c->soff.p); free(c->sbytes.p); free(c->lent.p);
    free(c->cfixed.p); free(c->csoff.p); free(c->csbytes.p); free(c->cvalid.p);
}
static inline void col_null(col_t *c)
{
    buf_u8(&c->valid, 0); c->n++;
    if (c->is_list) { buf_u64(&c->lent, c->child_n); buf_u64(&c->lent, 0); }     /* {current size, 0}: bcf_reader.c:1626-1630 */
    else if (c->type == T_VARCHAR) buf_u64(&c->soff, c->sbytes.n);
    else buf_u64(&c->fixed, 0);
}
static inline void col_fixed(col_t *c, uint64_t bits, int valid)
{
    buf_u8(&c->valid, (uint8_t)valid); c->n++; buf_u64(&c->fixed, bits);
}
static inline void col_str(col_t *c, const void *s, size_t n)
{
    buf_u8(&c->valid, 1); c->n++; buf_push(&c->sbytes, s, n); buf_u64(&c->soff, c->sbytes.n);
}
static inline void col_cstr(col_t *c, const char *s) { col_str(c, s, strlen(s)); }
static uint64_t g_list_start;
static inline void list_begin(col_t *c) { g_list_start = c->child_n; }
static inline void list_fixed(col_t *c, uint64_t bits) { buf_u64(&c->cfixed, bits); c->child_n++; }
static inline void list_str(col_t *c, const void *s, size_t n) { buf_push(&c->csbytes, s, n); buf_u64(&c->csoff, c->csbytes.n); c->child_n++; }
static inline void list_cvalid(col_t *c, int v) { buf_u8(&c->cvalid, (uint8_t)v); }       /* after list_fixed / list_str of a has_cvalid column */
static inline void list_end(col_t *c)
{
    buf_u8(&c->valid, 1); c->n++; buf_u64(&c->lent, g_list_start); buf_u64(&c->lent, c->child_n - g_list_start);
}

static inline void ser_col(buf_t *o, const col_t *c, int64_t n)
{
    uint16_t nl = (uint16_t)strlen(c->name);
    buf_push(o, &nl, 2); buf_push(o, c->name, nl);
    buf_u8(o, (uint8_t)c->type); buf_u8(o, (uint8_t)(c->is_list && c->has_cvalid ? 2 : c->is_list));
    buf_push(o, c->valid.p, (size_t)n);
    if (!c->is_list) {
        if (c->type == T_VARCHAR) { buf_push(o, c->soff.p, c->soff.n); buf_push(o, c->sbytes.p, c->sbytes.n); }
        else buf_push(o, c->fixed.p, c->fixed.n);
    } else {
        buf_push(o, c->lent.p, c->lent.n);
        buf_u64(o, c->child_n);
        if (c->type == T_VARCHAR) { buf_push(o, c->csoff.p, c->csoff.n); buf_push(o, c->csbytes.p, c->csbytes.n); }
        else buf_push(o, c->cfixed.p, c->cfixed.n);
        if (c->has_cvalid) buf_push(o, c->cvalid.p, c->cvalid.n);
    }
}


#endif
