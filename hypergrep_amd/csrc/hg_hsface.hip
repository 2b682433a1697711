// Face A: the six libhs symbols the reference shim links against (hypergrep/lib/c/hyperscanner.c:136,140,165,217,
// 301,323,324), block mode only.  hs_scan copies the block to HBM and runs the same stream / filter kernels in
// block mode (the buffer is one scan unit, no line splitting), then delivers reports in ascending end offset.
// Per-call cost is a few launches and two synchronisations, so this face is for compatibility (per-line callers such
// as the reference shim); bulk scanning goes through hyperscan() / hg_scan_device().
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
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
  // short blocks (HgScanner::launch_block_small): pinned copies of the block and of the raw reports
  uint8_t *h_text = nullptr;
  HgHit *h_out = nullptr;
  uint32_t *h_counts = nullptr;  // [0, 64) reports per segment, [64] completion flag
  uint32_t seq = 0;
};

namespace {
inline void cpu_relax() {  // a polite spin-wait hint, whatever the host is
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__)
  __asm__ __volatile__("yield");
#else
  std::this_thread::yield();
#endif
}
}  // namespace

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
  // (a half-built scratch is handed to hs_free_scratch on every failure path: scanner, stream and pinned buffers are released)
  struct Guard {
    hs_scratch_t *p;
    ~Guard() { if (p) hs_free_scratch(p); }
  } s{new hs_scratch()};
  s.p->db = db->db;
  std::string err;
  int device = 0;
  if (const char *env = std::getenv("HYPERGREP_DEVICE")) device = std::atoi(env);
  if (HgScanner::create(s.p->db, device, &s.p->sc, &err) != HG_OK) {
    std::fprintf(stderr, "hypergrep_amd: hs_alloc_scratch: %s\n", err.c_str());
    return HS_NOMEM;
  }
  if (hipStreamCreateWithFlags(&s.p->stream, hipStreamNonBlocking) != hipSuccess) return HS_NOMEM;
  if (hgmem::host_alloc(&s.p->h_text, HG_BLOCK_SMALL_MAX + 16, "hs h_text") != hipSuccess ||
      hgmem::host_alloc(&s.p->h_out, 64 * HG_BLOCK_SMALL_SEG * sizeof(HgHit), "hs h_out") != hipSuccess ||
      hgmem::host_alloc(&s.p->h_counts, 80 * sizeof(uint32_t), "hs h_counts") != hipSuccess)
    return HS_NOMEM;
  s.p->h_counts[64] = 0;
  *scratch = s.p;
  s.p = nullptr;
  return HS_SUCCESS;
}

int hs_free_scratch(hs_scratch_t *scratch) {
  if (!scratch) return HS_SUCCESS;
  if (scratch->sc) (void)hipSetDevice(scratch->sc->device());
  delete scratch->sc;
  hgmem::dev_free(scratch->d_text, "hs d_text");
  hgmem::host_free(scratch->h_text, "hs h_text");
  hgmem::host_free(scratch->h_out, "hs h_out");
  hgmem::host_free(scratch->h_counts, "hs h_counts");
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
  // Short blocks (the reference shim scans line by line, hyperscanner.c:217): one launch on a pinned copy of the block,
  // raw reports straight into pinned memory, the report rules on the host.
  static const bool small_path = !std::getenv("HG_NO_BLOCK_SMALL");
  if (length <= HG_BLOCK_SMALL_MAX && small_path) {
    std::memcpy(scratch->h_text, data, length);
    std::memset(scratch->h_text + length, 0, (16 - (length & 15)) & 15);
    const uint32_t seq = ++scratch->seq ? scratch->seq : ++scratch->seq;  // (never 0)
    const uint32_t segs = scratch->sc->launch_block_small(scratch->h_text, length, scratch->stream, scratch->h_out, scratch->h_counts, scratch->h_counts + 64, seq);
    if (segs) {
      // wait for the kernel's completion word in pinned memory (a few microseconds of polling; a stream synchronisation
      // sleeps until an interrupt); after ~2 ms of polling fall back to the synchronisation, which also reports errors
      volatile uint32_t *flag = scratch->h_counts + 64;
      bool done = false;
      for (uint32_t spin = 0; spin < 400000 && !(done = *flag == seq); spin++) cpu_relax();
      if (!done && hipStreamSynchronize(scratch->stream) != hipSuccess) return HS_INVALID;
      std::atomic_thread_fence(std::memory_order_acquire);
      bool fits = true;
      scratch->hits.clear();
      for (uint32_t g = 0; g < segs && fits; g++) {
        const uint32_t n = scratch->h_counts[g];
        fits = n <= HG_BLOCK_SMALL_SEG;
        if (fits) scratch->hits.insert(scratch->hits.end(), scratch->h_out + static_cast<size_t>(g) * HG_BLOCK_SMALL_SEG, scratch->h_out + static_cast<size_t>(g) * HG_BLOCK_SMALL_SEG + n);
      }
      if (fits) {
        // report rules per id (hg_post.h, restated on the raw records): SINGLEMATCH expressions sharing an id give one
        // report (the smallest end offset), the others every distinct end offset, identical (id, to) once
        auto &h = scratch->hits;
        auto to_of = [](const HgHit &x) { return x.to & ~HG_HIT_SINGLE_BIT; };
        auto single_of = [](const HgHit &x) { return (x.to & HG_HIT_SINGLE_BIT) != 0; };
        std::sort(h.begin(), h.end(), [&](const HgHit &a, const HgHit &b) {
          if (a.id != b.id) return a.id < b.id;
          if (to_of(a) != to_of(b)) return to_of(a) < to_of(b);
          return single_of(a) < single_of(b);
        });
        size_t kept = 0;
        bool seen_single = false;
        for (size_t i = 0; i < h.size(); i++) {
          if (i == 0 || h[i].id != h[i - 1].id) seen_single = false;
          const bool dup = i > 0 && h[i].id == h[i - 1].id && to_of(h[i]) == to_of(h[i - 1]);
          const bool single = single_of(h[i]);
          const bool keep = !dup && !(single && seen_single);
          if (single) seen_single = true;
          if (keep) {
            HgHit out = h[i];
            out.to = to_of(h[i]);
            h[kept++] = out;
          }
        }
        h.resize(kept);
        std::sort(h.begin(), h.end(), [](const HgHit &a, const HgHit &b) { return a.to != b.to ? a.to < b.to : a.id < b.id; });
        for (const HgHit &x : h)
          if (on_event && on_event(x.id, 0, x.to, 0, context)) return HS_SCAN_TERMINATED;
        return HS_SUCCESS;
      }
    }
  }
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
