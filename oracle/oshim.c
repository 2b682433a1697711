/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see orx.h).
 *
 * CPU restatement of the reference's per-file scan driver, written from its behaviour:
 *   /root/reference/hypergrep/lib/c/hyperscanner.c
 *     :248-326  hyperscan()      -> oracle_hyperscan()      (ring of buffer_count results, rc 0..7)
 *     :179-231  hyperscan_gz()   -> scan_pieces()           (gzgets pieces, NUL rules, early exit)
 *     :83-102   hs_callback()    -> deliver()               (copy id/line_number/line, fire full batch)
 *     :154-167  check_patterns() -> oracle_check_patterns() (rc 4 on compile failure)
 *   gzgets contract: /usr/include/zlib.h "gzgets" (reads at most len-1 bytes, stops after '\n').
 *
 * Line pieces come from the system zlib's own gzgets for files (the reference's dependency for
 * plain and gzip input) and from a restated splitter for memory buffers (oracle_scan_buffer,
 * used for GPU parity); tests check that both agree.  zstd input is decoded with the system
 * libzstd (dlopen) and then split from memory.
 *
 * One documented deviation: a piece consisting only of NUL bytes makes the reference's
 * leading-NUL skip (hyperscanner.c:207-214) walk into stale / uninitialised buffer bytes
 * (undefined behaviour).  The oracle — and the product — scan such a piece as empty.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include "oracle.h"

/* libhs-subset face (ohs.c) */
typedef struct hs_database hs_database_t;
typedef struct hs_scratch hs_scratch_t;
typedef struct hs_compile_error hs_compile_error_t;
typedef int (*match_event_handler)(unsigned int, unsigned long long, unsigned long long, unsigned int, void *);
int hs_compile_multi(const char *const *, const unsigned int *, const unsigned int *, unsigned int, unsigned int,
                     const void *, hs_database_t **, hs_compile_error_t **);
int hs_free_compile_error(hs_compile_error_t *);
int hs_alloc_scratch(const hs_database_t *, hs_scratch_t **);
int hs_scan(const hs_database_t *, const char *, unsigned int, unsigned int, hs_scratch_t *, match_event_handler,
            void *);
int hs_free_scratch(hs_scratch_t *);
int hs_free_database(hs_database_t *);

enum { RC_COMPILE_MEM = 1, RC_COMPILE = 2, RC_SCRATCH = 3, RC_DB = 4, RC_STATE_MEM = 5, RC_GZ_OPEN = 6, RC_SCAN = 7 };

typedef struct {
    /* batching ring (hyperscanner.c:64-72) */
    oracle_result_t *ring;
    int ring_size, ring_fill;
    oracle_event_fn on_event;
    unsigned long long match_count;
    /* current piece */
    unsigned long long line_number;
    const unsigned char *line;
    size_t line_len;
    uint64_t line_off; /* offset of `line` in the decoded stream (buffer API only) */
    /* buffer API sink */
    oracle_hit_t *hits;
    size_t nhits, caphits;
    int collect;
} scan_state;

static int deliver(unsigned int id, unsigned long long from, unsigned long long to, unsigned int flags, void *ctx) {
    (void)from;
    (void)flags;
    scan_state *st = (scan_state *)ctx;
    st->match_count++;
    if (st->collect) {
        if (st->nhits == st->caphits) {
            st->caphits = st->caphits ? st->caphits * 2 : 64;
            st->hits = (oracle_hit_t *)realloc(st->hits, sizeof(oracle_hit_t) * st->caphits);
        }
        oracle_hit_t *h = &st->hits[st->nhits++];
        h->line_number = st->line_number;
        h->id = id;
        h->to = (uint32_t)to;
        h->line_off = st->line_off;
        h->line_len = (uint32_t)st->line_len;
        h->pad = 0;
        return 0;
    }
    oracle_result_t *r = &st->ring[st->ring_fill++];
    r->id = id;
    r->line_number = st->line_number;
    memcpy(r->line, st->line, st->line_len);
    r->line[st->line_len] = 0;
    if (st->ring_fill == st->ring_size) {
        st->on_event(st->ring, st->ring_fill);
        st->ring_fill = 0;
    }
    return 0;
}

/* One gzgets piece -> the bytes hs_scan sees: leading NULs skipped, cut at the first NUL. */
static void trim_piece(const unsigned char *p, size_t n, const unsigned char **out, size_t *outlen) {
    size_t a = 0;
    while (a < n && p[a] == 0) a++;
    size_t z = a;
    while (z < n && p[z] != 0) z++;
    *out = p + a;
    *outlen = z - a;
}

/* Returns 0, or RC_SCAN.  *stop set when max_match_count reached. */
static int scan_piece(scan_state *st, hs_database_t *db, hs_scratch_t *scratch, const unsigned char *piece, size_t n,
                      uint64_t piece_off, unsigned long long max_match_count, int *stop) {
    const unsigned char *line;
    size_t len;
    trim_piece(piece, n, &line, &len);
    st->line = line;
    st->line_len = len;
    st->line_off = piece_off + (uint64_t)(line - piece);
    if (hs_scan(db, (const char *)line, (unsigned)len, 0, scratch, deliver, st) != 0) {
        fprintf(stderr, "ERROR: Unable to scan buffer. Exiting.\n");
        return RC_SCAN;
    }
    if (max_match_count > 0 && st->match_count >= max_match_count) {
        *stop = 1;
        return 0;
    }
    st->line_number++;
    return 0;
}

/* Restated gzgets splitter over memory: pieces of at most buffer_size-1 bytes, ending after '\n'. */
static int scan_memory(scan_state *st, hs_database_t *db, hs_scratch_t *scratch, const unsigned char *data, size_t len,
                       int buffer_size, unsigned long long max_match_count) {
    size_t maxp = buffer_size > 1 ? (size_t)buffer_size - 1 : 0;
    size_t pos = 0;
    int stop = 0;
    if (maxp == 0) return 0; /* gzgets with len<=1 returns NULL */
    while (pos < len && !stop) {
        size_t n = len - pos < maxp ? len - pos : maxp;
        const unsigned char *nl = (const unsigned char *)memchr(data + pos, '\n', n);
        if (nl) n = (size_t)(nl - (data + pos)) + 1;
        int rc = scan_piece(st, db, scratch, data + pos, n, pos, max_match_count, &stop);
        if (rc) return rc;
        pos += n;
    }
    return 0;
}

/* ---- zstd via dlopen (no headers in the image) ---- */
typedef struct { const void *src; size_t size, pos; } zin_t;
typedef struct { void *dst; size_t size, pos; } zout_t;

static int zstd_decode_file(const char *path, unsigned char **out, size_t *outlen) {
    static void *lib;
    static void *(*create)(void);
    static size_t (*freeds)(void *);
    static size_t (*dstream)(void *, zout_t *, zin_t *);
    static unsigned (*is_error)(size_t);
    if (!lib) {
        lib = dlopen("libzstd.so.1", RTLD_NOW);
        if (!lib) return -1;
        create = (void *(*)(void))dlsym(lib, "ZSTD_createDStream");
        freeds = (size_t(*)(void *))dlsym(lib, "ZSTD_freeDStream");
        dstream = (size_t(*)(void *, zout_t *, zin_t *))dlsym(lib, "ZSTD_decompressStream");
        is_error = (unsigned (*)(size_t))dlsym(lib, "ZSTD_isError");
        if (!create || !freeds || !dstream || !is_error) return -1;
    }
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    void *ds = create();
    size_t cap = 1 << 16, n = 0;
    unsigned char *buf = (unsigned char *)malloc(cap);
    unsigned char inbuf[1 << 15];
    size_t got;
    int rc = 0;
    while ((got = fread(inbuf, 1, sizeof inbuf, f)) > 0) {
        zin_t in = {inbuf, got, 0};
        while (in.pos < in.size) {
            if (cap - n < (1 << 15)) {
                cap *= 2;
                buf = (unsigned char *)realloc(buf, cap);
            }
            zout_t o = {buf + n, cap - n, 0};
            size_t r = dstream(ds, &o, &in);
            if (is_error(r)) { rc = -1; goto done; }
            n += o.pos;
        }
    }
done:
    freeds(ds);
    fclose(f);
    if (rc) { free(buf); return rc; }
    *out = buf;
    *outlen = n;
    return 0;
}

static int is_zstd_file(const char *path) {
    FILE *f = fopen(path, "rb");
    if (!f) return 0;
    unsigned char m[4];
    size_t g = fread(m, 1, 4, f);
    fclose(f);
    return g == 4 && m[0] == 0x28 && m[1] == 0xB5 && m[2] == 0x2F && m[3] == 0xFD;
}

static int scan_file(scan_state *st, hs_database_t *db, hs_scratch_t *scratch, const char *path, int buffer_size,
                     unsigned long long max_match_count) {
    if (is_zstd_file(path)) {
        unsigned char *mem = NULL;
        size_t n = 0;
        if (zstd_decode_file(path, &mem, &n)) return RC_GZ_OPEN;
        int rc = scan_memory(st, db, scratch, mem, n, buffer_size, max_match_count);
        free(mem);
        return rc;
    }
    gzFile f = gzopen(path, "rb");
    if (!f) return RC_GZ_OPEN;
    char *buf = (char *)calloc((size_t)(buffer_size > 0 ? buffer_size : 1), 1);
    int rc = 0, stop = 0;
    while (!stop) {
        z_off_t before = gztell(f);
        if (!gzgets(f, buf, buffer_size)) break;
        z_off_t after = gztell(f);
        size_t n = (size_t)(after - before);
        rc = scan_piece(st, db, scratch, (const unsigned char *)buf, n, (uint64_t)before, max_match_count, &stop);
        if (rc) break;
    }
    gzclose(f);
    free(buf);
    return rc;
}

static int build_db(const char *const *patterns, const unsigned *flags, const unsigned *ids, unsigned n,
                    hs_database_t **db) {
    hs_compile_error_t *err = NULL;
    int rc = hs_compile_multi(patterns, flags, ids, n, 1 /* HS_MODE_BLOCK */, NULL, db, &err);
    hs_free_compile_error(err);
    return rc == 0 ? 0 : RC_COMPILE;
}

int oracle_check_patterns(const char *const *patterns, const unsigned int *pattern_flags,
                          const unsigned int *pattern_ids, const unsigned int elements) {
    hs_database_t *db = NULL;
    int rc = build_db(patterns, pattern_flags, pattern_ids, elements, &db) ? RC_DB : 0;
    hs_free_database(db);
    return rc;
}

int oracle_hyperscan(char *file_name, const char *const *patterns, const unsigned int *pattern_flags,
                     const unsigned int *pattern_ids, const unsigned int elements, oracle_event_fn on_event,
                     const int buffer_size, int buffer_count, unsigned long long max_match_count) {
    if (max_match_count > 0 && max_match_count < (unsigned long long)buffer_count) buffer_count = (int)max_match_count;
    scan_state st;
    memset(&st, 0, sizeof st);
    st.on_event = on_event;
    st.ring_size = buffer_count;
    st.ring = (oracle_result_t *)calloc((size_t)(buffer_count > 0 ? buffer_count : 1), sizeof(oracle_result_t));
    if (!st.ring) return RC_COMPILE_MEM;
    for (int i = 0; i < buffer_count; i++) {
        st.ring[i].line = (char *)malloc((size_t)(buffer_size > 0 ? buffer_size : 1));
        if (!st.ring[i].line) return RC_COMPILE_MEM;
    }
    int rc = 0;
    hs_database_t *db = NULL;
    hs_scratch_t *scratch = NULL;
    if (build_db(patterns, pattern_flags, pattern_ids, elements, &db)) {
        fprintf(stderr, "ERROR: Unable to create database. Exiting.\n");
        rc = RC_DB;
    } else if (hs_alloc_scratch(db, &scratch) != 0) {
        fprintf(stderr, "ERROR: Unable to allocate scratch space. Exiting.\n");
        rc = RC_SCRATCH;
    } else {
        rc = scan_file(&st, db, scratch, file_name, buffer_size, max_match_count);
        if (st.ring_fill) st.on_event(st.ring, st.ring_fill);
    }
    for (int i = 0; i < buffer_count; i++) free(st.ring[i].line);
    free(st.ring);
    hs_free_scratch(scratch);
    hs_free_database(db);
    return rc;
}

int oracle_scan_buffer(const unsigned char *data, size_t len, const char *const *patterns,
                       const unsigned int *pattern_flags, const unsigned int *pattern_ids, unsigned int elements,
                       int buffer_size, unsigned long long max_match_count, oracle_hit_t **out_hits, size_t *out_n,
                       uint64_t *out_lines) {
    scan_state st;
    memset(&st, 0, sizeof st);
    st.collect = 1;
    hs_database_t *db = NULL;
    hs_scratch_t *scratch = NULL;
    int rc = 0;
    if (build_db(patterns, pattern_flags, pattern_ids, elements, &db)) rc = RC_DB;
    else if (hs_alloc_scratch(db, &scratch) != 0) rc = RC_SCRATCH;
    else rc = scan_memory(&st, db, scratch, data, len, buffer_size, max_match_count);
    hs_free_scratch(scratch);
    hs_free_database(db);
    if (out_hits) *out_hits = st.hits; else free(st.hits);
    if (out_n) *out_n = st.nhits;
    if (out_lines) *out_lines = st.line_number;
    return rc;
}

void oracle_free(void *p) { free(p); }
