/* TEST TOOL: drives a libhs-ABI shared object (Face A) with the exact call sequence of the reference shim
 * (hypergrep/lib/c/hyperscanner.c): hs_compile_multi (:136) -> hs_free_compile_error, also with NULL (:140) ->
 * hs_alloc_scratch (:301) -> per line piece: leading-NUL skip (:207-214), hs_scan(line, strlen(line)) (:217) with a
 * callback that always returns 0 like hs_callback (:101) -> hs_free_scratch, hs_free_database (:323-324), which the
 * shim also calls with NULL on its error paths (:296-306), so both are called with NULL here first.
 *
 * It is written from the shim's BEHAVIOUR (its own file reading via fgets stands in for gzgets on plain files), loads the
 * library under test with dlopen — either this repository's hypergrep_amd/lib/libhs.so.5 or the oracle's
 * oracle/_build/libhs.so.5 — and prints one line per report: "<line_number> <id> <to>".  tests/test_gpu_parity.py compares
 * the two outputs; with --time it also prints the per-call latency of hs_scan to stderr.
 *
 *   hs_call_order <libhs.so.5> <file> <buffer_size> [--time] -- <flags> <id> <expr> [<flags> <id> <expr> ...]
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct hs_database hs_database_t;
typedef struct hs_scratch hs_scratch_t;
typedef struct hs_compile_error {
  char *message;
  int expression;
} hs_compile_error_t;
typedef int (*match_event_handler)(unsigned int id, unsigned long long from, unsigned long long to, unsigned int flags, void *ctx);

typedef int (*compile_multi_fn)(const char *const *, const unsigned int *, const unsigned int *, unsigned int, unsigned int, const void *,
                                hs_database_t **, hs_compile_error_t **);
typedef int (*free_compile_error_fn)(hs_compile_error_t *);
typedef int (*alloc_scratch_fn)(const hs_database_t *, hs_scratch_t **);
typedef int (*scan_fn)(const hs_database_t *, const char *, unsigned int, unsigned int, hs_scratch_t *, match_event_handler, void *);
typedef int (*free_scratch_fn)(hs_scratch_t *);
typedef int (*free_database_fn)(hs_database_t *);

struct state {
  unsigned long long line_number;
  unsigned long long reports;
};

static int on_match(unsigned int id, unsigned long long from, unsigned long long to, unsigned int flags, void *ctx) {
  struct state *st = (struct state *)ctx;
  (void)from;
  (void)flags;
  printf("%llu %u %llu\n", st->line_number, id, to);
  st->reports++;
  return 0; /* keep scanning, as hs_callback does (hyperscanner.c:101) */
}

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int main(int argc, char **argv) {
  if (argc < 8) {
    fprintf(stderr, "usage: %s <libhs> <file> <buffer_size> [--time] -- <flags> <id> <expr> ...\n", argv[0]);
    return 64;
  }
  const char *lib_path = argv[1], *file_name = argv[2];
  const int buffer_size = atoi(argv[3]);
  int timing = 0, at = 4;
  if (strcmp(argv[at], "--time") == 0) {
    timing = 1;
    at++;
  }
  if (strcmp(argv[at], "--") != 0 || (argc - at - 1) % 3 != 0 || buffer_size < 2) {
    fprintf(stderr, "bad arguments\n");
    return 64;
  }
  at++;
  const unsigned n = (unsigned)((argc - at) / 3);
  const char **exprs = calloc(n, sizeof(*exprs));
  unsigned *flags = calloc(n, sizeof(*flags)), *ids = calloc(n, sizeof(*ids));
  for (unsigned i = 0; i < n; i++) {
    flags[i] = (unsigned)strtoul(argv[at + 3 * i], NULL, 0);
    ids[i] = (unsigned)strtoul(argv[at + 3 * i + 1], NULL, 0);
    exprs[i] = argv[at + 3 * i + 2];
  }

  void *lib = dlopen(lib_path, RTLD_NOW | RTLD_LOCAL);
  if (!lib) {
    fprintf(stderr, "dlopen: %s\n", dlerror());
    return 65;
  }
  compile_multi_fn hs_compile_multi = (compile_multi_fn)dlsym(lib, "hs_compile_multi");
  free_compile_error_fn hs_free_compile_error = (free_compile_error_fn)dlsym(lib, "hs_free_compile_error");
  alloc_scratch_fn hs_alloc_scratch = (alloc_scratch_fn)dlsym(lib, "hs_alloc_scratch");
  scan_fn hs_scan = (scan_fn)dlsym(lib, "hs_scan");
  free_scratch_fn hs_free_scratch = (free_scratch_fn)dlsym(lib, "hs_free_scratch");
  free_database_fn hs_free_database = (free_database_fn)dlsym(lib, "hs_free_database");
  if (!hs_compile_multi || !hs_free_compile_error || !hs_alloc_scratch || !hs_scan || !hs_free_scratch || !hs_free_database) {
    fprintf(stderr, "missing libhs symbol\n");
    return 66;
  }

  /* the shim's error paths free what was never allocated (hyperscanner.c:296-306 -> :323-324, and :140 after a success) */
  if (hs_free_compile_error(NULL) != 0 || hs_free_scratch(NULL) != 0 || hs_free_database(NULL) != 0) {
    fprintf(stderr, "a free function did not accept NULL\n");
    return 67;
  }

  int ret = 0;
  hs_database_t *db = NULL;
  hs_scratch_t *scratch = NULL;
  hs_compile_error_t *err = NULL;
  const int rc_compile = hs_compile_multi(exprs, flags, ids, n, 1 /* HS_MODE_BLOCK */, NULL, &db, &err);
  if (rc_compile != 0) {
    fprintf(stderr, "compile failed: rc %d expression %d: %s\n", rc_compile, err ? err->expression : -1, err && err->message ? err->message : "?");
    ret = 2; /* HYPERSCANNER_COMPILE */
  }
  hs_free_compile_error(err);
  if (ret == 0 && hs_alloc_scratch(db, &scratch) != 0) ret = 3; /* HYPERSCANNER_SCRATCH */

  if (ret == 0) {
    FILE *in = fopen(file_name, "rb");
    if (!in) {
      ret = 6; /* HYPERSCANNER_GZ_OPEN */
    } else {
      char *buf = malloc((size_t)buffer_size);
      struct state st = {0, 0};
      double t_scan = 0;
      unsigned long long calls = 0, bytes = 0;
      for (;;) {
        /* gzgets semantics on a plain file: at most buffer_size - 1 bytes, stops after '\n'; the shim then relies on
         * strlen, so embedded NULs cut the line (fgets keeps reading past a NUL exactly like gzgets does) */
        memset(buf, 0, (size_t)buffer_size); /* (defined behaviour for all-NUL pieces, where the shim reads stale bytes) */
        char *line = fgets(buf, buffer_size, in);
        if (!line) break;
        if (buf[0] == 0) {
          for (int s = 1; s < buffer_size; s++)
            if (buf[s] != 0) {
              line = buf + s;
              break;
            }
        }
        const size_t len = strlen(line);
        const double t0 = timing ? now_s() : 0;
        if (hs_scan(db, line, (unsigned)len, 0, scratch, on_match, &st) != 0) {
          fprintf(stderr, "ERROR: Unable to scan buffer. Exiting.\n");
          ret = 7; /* HYPERSCANNER_SCAN */
          break;
        }
        if (timing) t_scan += now_s() - t0;
        calls++;
        bytes += len;
        st.line_number++;
      }
      free(buf);
      fclose(in);
      if (timing && calls)
        fprintf(stderr, "hs_scan: %llu calls, %llu bytes, %llu reports, %.1f us per call\n", calls, bytes, st.reports, 1e6 * t_scan / (double)calls);
    }
  }
  hs_free_scratch(scratch);
  hs_free_database(db);
  free(exprs);
  free(flags);
  free(ids);
  return ret;
}
