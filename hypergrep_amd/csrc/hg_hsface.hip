// Face A: the six libhs symbols the reference shim links against (hypergrep/lib/c/hyperscanner.c:136,140,165,217,
// 301,323,324), block mode only.  hs_scan copies the block to HBM and runs the same stream / filter kernels in
// block mode (the buffer is one scan unit, no line splitting), then delivers reports in ascending end offset.
// Per-call cost is a few launches and two synchronisations, so this face is for compatibility (per-line callers such
// as the reference shim); bulk scanning goes through hyperscan() / hg_scan_device().
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/hypergrep_amd.h"
#include "hg_compile.h"
#include "hg_engine.h"
#include "hg_mem.h"

struct hs_database {
  std::shared_ptr<HgDb> db;
};
struct hs_scratch {
  std::shared_ptr<HgDb> db;
  HgScanner *sc = nullptr;
  hipStream_t stream = nullptr;
  uint8_t *d_text = nullptr;
  size_t d_cap = 0;
  std::vector<HgHit> hits;
};

extern "C" {

int hs_compile_multi(const char *const *expressions, const unsigned int *flags, const unsigned int *ids, unsigned int elements,
                     unsigned int mode, const hs_platform_info_t *platform, hs_database_t **db, hs_compile_error_t **error) {
  (void)platform;
  if (error) *error = nullptr;
  std::string msg;
  int bad = -1;
  HgDb *raw = nullptr;
  if (!db || !expressions || elements == 0 || mode != HS_MODE_BLOCK) msg = "invalid arguments (block mode, at least one expression)";
  else if (hgc_compile(expressions, flags, ids, elements, &raw, &msg, &bad) == 0) {
    *db = new hs_database{std::shared_ptr<HgDb>(raw, [](HgDb *d) { hgc_free(d); })};
    return HS_SUCCESS;
  }
  if (db) *db = nullptr;
  if (error) {
    hs_compile_error_t *e = static_cast<hs_compile_error_t *>(std::malloc(sizeof(hs_compile_error_t)));
    e->message = strdup(msg.c_str());
    e->expression = bad;
    *error = e;
  }
  return HS_COMPILER_ERROR;
}

int hs_free_compile_error(hs_compile_error_t *error) {
  if (!error) return HS_SUCCESS;
  std::free(error->message);
  std::free(error);
  return HS_SUCCESS;
}

int hs_free_database(hs_database_t *db) {
  delete db;
  return HS_SUCCESS;
}

int hs_alloc_scratch(const hs_database_t *db, hs_scratch_t **scratch) {
  if (!db || !scratch) return HS_INVALID;
  if (*scratch && (*scratch)->db == db->db) return HS_SUCCESS;
  if (*scratch) hs_free_scratch(*scratch);
  *scratch = nullptr;
  auto s = std::make_unique<hs_scratch>();
  s->db = db->db;
  std::string err;
  int device = 0;
  if (const char *env = std::getenv("HYPERGREP_DEVICE")) device = std::atoi(env);
  if (HgScanner::create(s->db, device, &s->sc, &err) != HG_OK) {
    std::fprintf(stderr, "hypergrep_amd: hs_alloc_scratch: %s\n", err.c_str());
    return HS_NOMEM;
  }
  if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) return HS_NOMEM;
  *scratch = s.release();
  return HS_SUCCESS;
}

int hs_free_scratch(hs_scratch_t *scratch) {
  if (!scratch) return HS_SUCCESS;
  if (scratch->sc) (void)hipSetDevice(scratch->sc->device());
  delete scratch->sc;
  hgmem::dev_free(scratch->d_text, "hs d_text");
  if (scratch->stream) (void)hipStreamDestroy(scratch->stream);
  delete scratch;
  return HS_SUCCESS;
}

int hs_scan(const hs_database_t *db, const char *data, unsigned int length, unsigned int flags, hs_scratch_t *scratch,
            match_event_handler on_event, void *context) {
  (void)flags;
  if (!db || !scratch || !scratch->sc || scratch->db != db->db || (!data && length)) return HS_INVALID;
  if (length == 0) return HS_SUCCESS;  // no expression can match the empty buffer (such expressions are rejected at compile time)
  if (hipSetDevice(scratch->sc->device()) != hipSuccess) return HS_INVALID;
  if (scratch->d_cap < length) {
    hgmem::dev_free(scratch->d_text, "hs d_text");
    scratch->d_text = nullptr;
    size_t cap = std::max<size_t>(length, 4096) * 2;
    if (hgmem::dev_alloc(&scratch->d_text, cap + 16, "hs d_text") != hipSuccess) return HS_NOMEM;
    scratch->d_cap = cap;
  }
  if (hipMemcpyAsync(scratch->d_text, data, length, hipMemcpyHostToDevice, scratch->stream) != hipSuccess) return HS_INVALID;
  HgScanOutput out{};
  if (scratch->sc->scan_block(scratch->d_text, length, scratch->stream, &out) != HG_OK) {
    std::fprintf(stderr, "hypergrep_amd: hs_scan: %s\n", scratch->sc->last_error().c_str());
    return HS_INVALID;
  }
  scratch->hits.resize(out.n_hits);
  if (out.n_hits) {
    if (hipMemcpyAsync(scratch->hits.data(), out.d_hits, out.n_hits * sizeof(HgHit), hipMemcpyDeviceToHost, scratch->stream) != hipSuccess ||
        hipStreamSynchronize(scratch->stream) != hipSuccess)
      return HS_INVALID;
  }
  // device order is (id, to); Hyperscan delivers by ascending end offset (ties by id here)
  std::sort(scratch->hits.begin(), scratch->hits.end(), [](const HgHit &a, const HgHit &b) { return a.to != b.to ? a.to < b.to : a.id < b.id; });
  for (const HgHit &h : scratch->hits)
    if (on_event && on_event(h.id, 0, h.to, 0, context)) return HS_SCAN_TERMINATED;
  return HS_SUCCESS;
}

}  // extern "C"
