/*
 * hypergrep_amd — C ABI of the MI355X (gfx950) multi-pattern line-scan engine.
 *
 * This one shared object replaces the native bundle the reference builds with
 * utils/build_hyperscanner.sh (libhs + libhyperscanner): every entry point is plain C, no torch or
 * C++ types cross the boundary.  Three faces:
 *
 *   Face B  hyperscan(), check_patterns()    what the reference's Python calls through ctypes
 *                                            (hypergrep/utils.py:116-121, :339-349) — same names,
 *                                            argument order, return codes and callback contract as
 *                                            hypergrep/lib/c/hyperscanner.c:154-159 and :248-258.
 *   Face A  hs_compile_multi() ... hs_scan()  the six libhs symbols the reference shim links against
 *                                            (hyperscanner.c:136,140,165,217,301,323,324), block mode.
 *   hg_*    buffer-level API for text that is already resident in HBM (bench, multi-GPU shards,
 *           framework integrations): compile once, scan device buffers, read hit records.
 *
 * The scan itself always runs on the GPU.  Without a usable HIP device the entry points fail
 * (hyperscan() returns 3 and prints the reason; hg_* return HG_ERR_HIP): there is no CPU fallback.
 */
#ifndef HYPERGREP_AMD_H
#define HYPERGREP_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ Face B: shim ABI ---------- */

/* hyperscanner_result_t, hypergrep/lib/c/hyperscanner.c:42-46 == hypergrep/utils.py:25-40 (Result).
 * Offsets 0 / 8 / 16, sizeof 24.  `line` is NUL terminated and includes the trailing '\n' if the
 * line had one.  The array and every `line` buffer are owned by the library and reused for the next
 * batch: valid only during the callback. */
typedef struct hyperscanner_result {
    unsigned int id;
    unsigned long long line_number;
    char *line;
} hyperscanner_result_t;

/* hs_event, hyperscanner.c:54 == utils.CALLBACK_TYPE (utils.py:45-51). */
typedef void (*hs_event)(hyperscanner_result_t *results, int result_count);

/* Return codes, hyperscanner.c:25-33. */
enum {
    HYPERSCANNER_COMPILE_MEM = 1,
    HYPERSCANNER_COMPILE = 2,
    HYPERSCANNER_SCRATCH = 3, /* also: no usable GPU / HIP failure while setting up the scanner */
    HYPERSCANNER_DB = 4,
    HYPERSCANNER_STATE_MEM = 5,
    HYPERSCANNER_GZ_OPEN = 6,
    HYPERSCANNER_SCAN = 7
};

/* Replaces hyperscan(), hyperscanner.c:248-326.  Reads `file_name` (plain, gzip or zstd), scans it line
 * piece by line piece on the GPU and calls `on_event` with batches of `buffer_count` results in ascending
 * line order (last batch may be short).  buffer_size: a line longer than buffer_size-1 bytes is split
 * into pieces that are scanned and numbered separately (gzgets contract).  max_match_count: stop after
 * the line on which the running number of reports reaches it (0 = no limit). */
int hyperscan(char *file_name, const char *const *patterns, const unsigned int *pattern_flags,
              const unsigned int *pattern_ids, const unsigned int elements, hs_event on_event,
              const int buffer_size, int buffer_count, unsigned long long max_match_count);

/* Replaces check_patterns(), hyperscanner.c:154-167: compile only; 0 or HYPERSCANNER_DB (4). Needs no GPU. */
int check_patterns(const char *const *patterns, const unsigned int *pattern_flags,
                   const unsigned int *pattern_ids, const unsigned int elements);

/* ------------------------------------------------------------------ Face A: libhs subset ------ */

typedef struct hs_database hs_database_t;
typedef struct hs_scratch hs_scratch_t;
typedef struct hs_compile_error {
    char *message;
    int expression;
} hs_compile_error_t;
typedef struct hs_platform_info hs_platform_info_t;
typedef int (*match_event_handler)(unsigned int id, unsigned long long from, unsigned long long to,
                                   unsigned int flags, void *context);

#define HS_SUCCESS 0
#define HS_INVALID (-1)
#define HS_NOMEM (-2)
#define HS_SCAN_TERMINATED (-3)
#define HS_COMPILER_ERROR (-4)
#define HS_MODE_BLOCK 1
#define HS_FLAG_CASELESS 1
#define HS_FLAG_DOTALL 2
#define HS_FLAG_MULTILINE 4
#define HS_FLAG_SINGLEMATCH 8

/* call site hyperscanner.c:136 */
int hs_compile_multi(const char *const *expressions, const unsigned int *flags, const unsigned int *ids,
                     unsigned int elements, unsigned int mode, const hs_platform_info_t *platform,
                     hs_database_t **db, hs_compile_error_t **error);
/* call site hyperscanner.c:140 (called with NULL when compilation succeeded) */
int hs_free_compile_error(hs_compile_error_t *error);
/* call site hyperscanner.c:301 */
int hs_alloc_scratch(const hs_database_t *db, hs_scratch_t **scratch);
/* call site hyperscanner.c:217: block-mode scan of data[0,length) as ONE unit (no line splitting).
 * The block is copied to HBM and scanned by the same kernels; `from` is always 0 (no SOM). */
int hs_scan(const hs_database_t *db, const char *data, unsigned int length, unsigned int flags,
            hs_scratch_t *scratch, match_event_handler on_event, void *context);
/* call sites hyperscanner.c:323, :165/:324 (both may receive NULL) */
int hs_free_scratch(hs_scratch_t *scratch);
int hs_free_database(hs_database_t *db);

/* ------------------------------------------------------------------ hg_*: device buffers ------ */

enum {
    HG_OK = 0,
    HG_ERR_ARG = -1,
    HG_ERR_NOMEM = -2,
    HG_ERR_COMPILE = -4,
    HG_ERR_HIP = -10,
    HG_ERR_SMALL_BUFFER = -11
};

typedef struct hg_database hg_database_t; /* compiled expressions (host memory) */
typedef struct hg_scanner hg_scanner_t;   /* database + workspace resident on one GPU; one scan at a time */

/* One report: (line piece, id).  16 bytes — the algorithmic write traffic per hit. */
typedef struct hg_hit {
    uint64_t line_number; /* 0-based piece index == hyperscanner_result_t.line_number */
    uint32_t id;          /* == hyperscanner_result_t.id */
    uint32_t to;          /* match end offset inside the scanned bytes (Hyperscan's `to`) */
} hg_hit_t;

/* Where Result.line lives in the scanned buffer. */
typedef struct hg_hit_aux {
    uint64_t start; /* byte offset of the first scanned byte of the piece */
    uint32_t len;   /* scanned length (what strlen(Result.line) would be) */
    uint32_t pattern; /* index of the expression that produced the report */
} hg_hit_aux_t;

typedef struct hg_scan_result {
    uint64_t n_hits;       /* ordered by (line_number, id, to), SINGLEMATCH / duplicate rules applied */
    uint64_t n_lines;      /* line pieces in the buffer */
    uint64_t n_candidates; /* required-literal occurrences that went to the confirm stage */
    uint64_t n_raw_hits;   /* reports before de-duplication */
    const hg_hit_t *d_hits;    /* DEVICE pointers, valid until the next scan on this scanner */
    const hg_hit_aux_t *d_aux;
    float ms_stream; /* duration of the streaming kernel (summed over its launches), HIP events on the launch stream */
    float ms_total;  /* whole launch sequence */
    uint32_t reruns; /* passes repeated because the workspace had to grow */
    uint32_t stream_launches; /* launches of the streaming kernel in this scan (one per pipeline chunk) */
    uint64_t joiner_tiles;    /* 16 KiB tiles streamed by the joiner launches (hg_stream_join_kernel), not by those */
    uint32_t joiner_launches; /* launches of the joiner kernel in this scan */
    uint32_t reserved;
} hg_scan_result_t;

typedef struct hg_db_info {
    uint32_t n_patterns;
    uint32_t n_literal_anchored; /* patterns filtered by the streaming window prefilter */
    uint32_t n_always_on;        /* patterns run on every line */
    uint32_t n_factors;
    uint32_t n_windows;
    uint32_t fold_mask;
    uint32_t max_state_words;
    uint32_t table_bytes;
    uint32_t byte_windows;       /* 1: the prefilter probes a window at every byte offset (sets with 3..6-byte required literals) */
} hg_db_info_t;

/* Compile `n` expressions (same inputs as hs_compile_multi).  On failure returns HG_ERR_COMPILE and
 * writes "<expression index>: <reason>" to err. */
int hg_db_compile(const char *const *expressions, const unsigned int *flags, const unsigned int *ids,
                  unsigned int n, hg_database_t **db, char *err, size_t errlen);
/* Optional: re-select the literal windows of the prefilter using byte statistics of a host-side text sample (any
 * part of what will be scanned) and rebuild the filter tables.  Never changes results, only how often the slower
 * stages run.  The tuned tables are built aside and swapped in on success: scanners created BEFORE the call keep the
 * tables they were created with (still valid), scanners created after it use the tuned ones; on failure nothing changes. */
int hg_db_tune(hg_database_t *db, const void *sample, size_t nbytes);
void hg_db_release(hg_database_t *db);
int hg_db_info(const hg_database_t *db, hg_db_info_t *info);

int hg_scanner_create(const hg_database_t *db, int device, hg_scanner_t **scanner, char *err, size_t errlen);
void hg_scanner_destroy(hg_scanner_t *scanner);
const char *hg_scanner_error(const hg_scanner_t *scanner);

/* Scan `nbytes` of text resident in HBM at d_text (16-byte aligned; must be readable up to nbytes
 * rounded up to 16).  Lines are numbered from line_base.  `stream` is a hipStream_t (NULL = default).
 * Blocks until the results are ready.  No size limit besides HBM: a buffer with more than 2^28 reports (or more
 * pipeline chunks than one pass has) is scanned in segments whose ordered hits are put one after the other. */
int hg_scan_device(hg_scanner_t *scanner, const void *d_text, uint64_t nbytes, int buffer_size,
                   uint64_t line_base, void *stream, hg_scan_result_t *result);

/* Copy the last scan's first `max` hits (and aux records, if aux != NULL) to host memory. */
int hg_copy_hits(hg_scanner_t *scanner, hg_hit_t *hits, hg_hit_aux_t *aux, uint64_t max);

/* Copy the last scan's first `max` hit records (16 B each) into another DEVICE buffer, asynchronously on
 * `stream` (e.g. a tensor that is then sent over RCCL). */
int hg_copy_hits_device(hg_scanner_t *scanner, void *d_dst, uint64_t max, void *stream);

/* Deterministic synthetic log used by bench.py and the parity tests: writes nbytes at d_text (device)
 * or text (host) from the same counter-based generator; see hypergrep_amd/csrc/hg_synth.h. */
typedef struct hg_synth_spec {
    uint64_t seed;
    uint64_t first_block;    /* index of the first 64 KiB block (shard offset) */
    uint32_t hit_per_million; /* probability that a line carries a needle, in 1e-6 units */
    uint32_t n_needles;
    const uint8_t *needles;       /* n_needles strings, concatenated */
    const uint32_t *needle_off;   /* n_needles + 1 offsets into needles */
} hg_synth_spec_t;
int hg_synth_device(void *d_text, uint64_t nbytes, const hg_synth_spec_t *spec, int device, void *stream);
int hg_synth_host(uint8_t *text, uint64_t nbytes, const hg_synth_spec_t *spec);

/* ---- diagnostics (tests, tools) ------------------------------------------------------------------------------------
 * A device buffer of `nbytes` (rounded up to 16) whose end is followed by reserved, unmapped address space: a read or
 * write past it is a GPU memory fault instead of a silent pass.  The parity tests run on such buffers so that an
 * over-read in any kernel fails where it happens.  hg_debug_upload / hg_debug_download: blocking copies to / from it. */
int hg_debug_alloc_guarded(uint64_t nbytes, int device, void **d_ptr, void **guard_handle);
void hg_debug_free_guarded(void *guard_handle);
int hg_debug_upload(void *d_dst, const void *src, uint64_t nbytes);
int hg_debug_download(void *dst, const void *d_src, uint64_t nbytes);
/* Face B bookkeeping, for tests and HYPERGREP_TRACE: the device the next scan context would be created on for a node of
 * `ndev` GPUs (HYPERGREP_DEVICE pins one, otherwise files round-robin; advances the round-robin), and the cache counters
 * {database cache hits, misses, entries, window tunings, contexts created, reused, re-bound to another pattern set, alive}. */
int hg_faceb_next_device(int ndev);
void hg_faceb_stats(uint64_t out[8]);

#ifdef __cplusplus
}
#endif
#endif
