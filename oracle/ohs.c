/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see orx.h).
 *
 * The six-symbol libhs subset the reference shim links against
 * (/root/reference/hypergrep/lib/c/hyperscanner.c:136 hs_compile_multi, :140 hs_free_compile_error,
 * :165/:324 hs_free_database, :217 hs_scan, :301 hs_alloc_scratch, :323 hs_free_scratch), backed by
 * the orx CPU restatement.  Built twice by oracle/Makefile:
 *   - into liboracle.so (used by tests / bench cpu_baseline through oshim.c), and
 *   - as oracle/_build/libhs.so.5 with SONAME libhs.so.5 so that, IN THE BUILD CONTAINER ONLY, the
 *     reference's unmodified prebuilt shim + Python can run on top of it
 *     (tests/golden/make_golden.py) to pin the oracle against the reference's own test tables.
 *
 * Multi-pattern report rules (Hyperscan 5.4 documented behaviour, restated):
 *   - every expression reports each distinct match end offset once ("to"; "from" is 0 without SOM);
 *   - HS_FLAG_SINGLEMATCH: only the first report for that expression's report id;
 *     expressions that share an id and are all SINGLEMATCH produce ONE report per scan
 *     (pinned by test_hypergrep.py:668-687 "Multiple redundant patterns");
 *   - identical (id, to) reports are delivered once;
 *   - reports are delivered in ascending end offset; ties by ascending id (unpinned).
 */
#define _POSIX_C_SOURCE 200809L
#include <stdlib.h>
#include <string.h>

#include "orx.h"

#define HS_SUCCESS 0
#define HS_INVALID (-1)
#define HS_NOMEM (-2)
#define HS_SCAN_TERMINATED (-3)
#define HS_COMPILER_ERROR (-4)
#define HS_MODE_BLOCK 1

typedef struct hs_compile_error {
    char *message;
    int expression;
} hs_compile_error_t;

typedef struct hs_database {
    unsigned n;
    orx_prog **progs;
    unsigned *ids;
    unsigned *flags;
} hs_database_t;

typedef struct hs_scratch {
    int unused;
} hs_scratch_t;

typedef int (*match_event_handler)(unsigned int id, unsigned long long from, unsigned long long to, unsigned int flags,
                                   void *context);

int hs_free_database(hs_database_t *db) {
    if (!db) return HS_SUCCESS;
    for (unsigned i = 0; i < db->n; i++) orx_free(db->progs[i]);
    free(db->progs);
    free(db->ids);
    free(db->flags);
    free(db);
    return HS_SUCCESS;
}

int hs_free_compile_error(hs_compile_error_t *err) {
    if (!err) return HS_SUCCESS;
    free(err->message);
    free(err);
    return HS_SUCCESS;
}

int hs_compile_multi(const char *const *expressions, const unsigned int *flags, const unsigned int *ids,
                     unsigned int elements, unsigned int mode, const void *platform, hs_database_t **db,
                     hs_compile_error_t **error) {
    (void)platform;
    char msg[256];
    msg[0] = 0;
    int bad = -1;
    if (error) *error = NULL;
    if (!db || !expressions || elements == 0 || mode != HS_MODE_BLOCK) {
        strcpy(msg, "invalid arguments (block mode, at least one expression required)");
        goto fail;
    }
    *db = NULL;
    hs_database_t *d = (hs_database_t *)calloc(1, sizeof *d);
    d->progs = (orx_prog **)calloc(elements, sizeof(orx_prog *));
    d->ids = (unsigned *)calloc(elements, sizeof(unsigned));
    d->flags = (unsigned *)calloc(elements, sizeof(unsigned));
    d->n = elements;
    for (unsigned i = 0; i < elements; i++) {
        unsigned f = flags ? flags[i] : 0;
        d->ids[i] = ids ? ids[i] : 0;
        d->flags[i] = f;
        d->progs[i] = orx_compile(expressions[i], f, msg, sizeof msg);
        if (!d->progs[i]) {
            bad = (int)i;
            hs_free_database(d);
            goto fail;
        }
    }
    *db = d;
    return HS_SUCCESS;
fail:
    if (error) {
        hs_compile_error_t *e = (hs_compile_error_t *)calloc(1, sizeof *e);
        e->message = strdup(msg);
        e->expression = bad;
        *error = e;
    }
    return HS_COMPILER_ERROR;
}

int hs_alloc_scratch(const hs_database_t *db, hs_scratch_t **scratch) {
    if (!db || !scratch) return HS_INVALID;
    if (!*scratch) *scratch = (hs_scratch_t *)calloc(1, sizeof(hs_scratch_t));
    return *scratch ? HS_SUCCESS : HS_NOMEM;
}

int hs_free_scratch(hs_scratch_t *scratch) {
    free(scratch);
    return HS_SUCCESS;
}

typedef struct {
    unsigned long long to;
    unsigned id;
} event_t;

typedef struct {
    event_t *ev;
    size_t n, cap;
    unsigned id;
} collect_t;

static void push_event(collect_t *c, unsigned long long to, unsigned id) {
    if (c->n == c->cap) {
        c->cap = c->cap ? c->cap * 2 : 16;
        c->ev = (event_t *)realloc(c->ev, sizeof(event_t) * c->cap);
    }
    c->ev[c->n].to = to;
    c->ev[c->n].id = id;
    c->n++;
}

static int collect_cb(size_t to, void *ctx) {
    collect_t *c = (collect_t *)ctx;
    push_event(c, (unsigned long long)to, c->id);
    return 0;
}

static int cmp_event(const void *a, const void *b) {
    const event_t *x = (const event_t *)a, *y = (const event_t *)b;
    if (x->to != y->to) return x->to < y->to ? -1 : 1;
    if (x->id != y->id) return x->id < y->id ? -1 : 1;
    return 0;
}

int hs_scan(const hs_database_t *db, const char *data, unsigned int length, unsigned int flags, hs_scratch_t *scratch,
            match_event_handler on_event, void *context) {
    (void)flags;
    if (!db || !scratch || (!data && length)) return HS_INVALID;
    collect_t all;
    memset(&all, 0, sizeof all);
    /* SINGLEMATCH expressions: first report per id (min over the expressions sharing the id). */
    collect_t singles;
    memset(&singles, 0, sizeof singles);
    for (unsigned i = 0; i < db->n; i++) {
        int single = (db->flags[i] & ORX_FLAG_SINGLEMATCH) != 0;
        if (single) {
            collect_t one;
            memset(&one, 0, sizeof one);
            one.id = db->ids[i];
            orx_scan(db->progs[i], (const unsigned char *)data, length, 1, collect_cb, &one);
            if (one.n) {
                size_t k;
                for (k = 0; k < singles.n; k++)
                    if (singles.ev[k].id == one.id) break;
                if (k == singles.n) push_event(&singles, one.ev[0].to, one.id);
                else if (one.ev[0].to < singles.ev[k].to) singles.ev[k].to = one.ev[0].to;
            }
            free(one.ev);
        } else {
            all.id = db->ids[i];
            orx_scan(db->progs[i], (const unsigned char *)data, length, 0, collect_cb, &all);
        }
    }
    for (size_t k = 0; k < singles.n; k++) push_event(&all, singles.ev[k].to, singles.ev[k].id);
    free(singles.ev);
    qsort(all.ev, all.n, sizeof(event_t), cmp_event);
    int rc = HS_SUCCESS;
    for (size_t k = 0; k < all.n; k++) {
        if (k && cmp_event(&all.ev[k - 1], &all.ev[k]) == 0) continue;
        if (on_event && on_event(all.ev[k].id, 0, all.ev[k].to, 0, context)) {
            rc = HS_SCAN_TERMINATED;
            break;
        }
    }
    free(all.ev);
    return rc;
}
