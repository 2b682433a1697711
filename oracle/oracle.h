/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see orx.h).  Public face of liboracle.so.
 * Mirrors the reference shim's ABI (/root/reference/hypergrep/lib/c/hyperscanner.c:42-54,154-159,248-258)
 * under oracle_* names, plus a memory-buffer entry used by the GPU parity tests.
 */
#ifndef ORACLE_H
#define ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* hyperscanner_result_t (hyperscanner.c:42-46): offsets 0 / 8 / 16, sizeof 24. */
typedef struct oracle_result {
    unsigned int id;
    unsigned long long line_number;
    char *line;
} oracle_result_t;

/* hs_event (hyperscanner.c:54). */
typedef void (*oracle_event_fn)(oracle_result_t *results, int result_count);

/* One (line piece, report) pair from the buffer API.  `to` is the match end offset inside the
 * scanned bytes; [line_off, line_off+line_len) are the bytes Result.line would hold. */
typedef struct oracle_hit {
    uint64_t line_number;
    uint32_t id;
    uint32_t to;
    uint64_t line_off;
    uint32_t line_len;
    uint32_t pad;
} oracle_hit_t;

int oracle_hyperscan(char *file_name, const char *const *patterns, const unsigned int *pattern_flags,
                     const unsigned int *pattern_ids, const unsigned int elements, oracle_event_fn on_event,
                     const int buffer_size, int buffer_count, unsigned long long max_match_count);

int oracle_check_patterns(const char *const *patterns, const unsigned int *pattern_flags,
                          const unsigned int *pattern_ids, const unsigned int elements);

/* Scan an in-memory byte stream with the same piece / NUL / numbering rules.  Hits are returned in
 * delivery order (ascending line_number; within a line ascending `to`, then id). Caller frees *out_hits
 * with oracle_free.  *out_lines = number of pieces fully processed. Returns 0 or the shim's rc. */
int oracle_scan_buffer(const unsigned char *data, size_t len, const char *const *patterns,
                       const unsigned int *pattern_flags, const unsigned int *pattern_ids, unsigned int elements,
                       int buffer_size, unsigned long long max_match_count, oracle_hit_t **out_hits, size_t *out_n,
                       uint64_t *out_lines);
void oracle_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
