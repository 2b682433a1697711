// hg_* entry points of include/hypergrep_amd.h: thin C wrappers over HgDb / HgScanner, plus the
// synthetic-log generator kernel.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>

#include "../../include/hypergrep_amd.h"
#include "hg_compile.h"
#include "hg_engine.h"
#include "hg_synth.h"

struct hg_database {
  std::shared_ptr<const HgDb> db;  // scanners share it; hg_db_tune swaps in a tuned copy, scanners made before keep theirs
};
struct hg_scanner {
  HgScanner *sc;
  hg_scan_result_t last;
};

static_assert(sizeof(hg_hit_t) == sizeof(HgHit) && sizeof(hg_hit_aux_t) == sizeof(HgHitAux), "ABI records mirror the device records");

static void put_err(char *err, size_t errlen, const std::string &msg) {
  if (err && errlen) std::snprintf(err, errlen, "%s", msg.c_str());
}

extern "C" {

int hg_db_compile(const char *const *expressions, const unsigned int *flags, const unsigned int *ids, unsigned int n,
                  hg_database_t **db, char *err, size_t errlen) {
  if (!db) return HG_ERR_ARG;
  *db = nullptr;
  HgDb *d = nullptr;
  std::string msg;
  int bad = -1;
  int rc = hgc_compile(expressions, flags, ids, n, &d, &msg, &bad);
  if (rc != 0) {
    put_err(err, errlen, std::to_string(bad) + ": " + msg);
    return rc == -2 ? HG_ERR_NOMEM : (rc == -1 ? HG_ERR_ARG : HG_ERR_COMPILE);
  }
  *db = new hg_database{std::shared_ptr<const HgDb>(d, [](const HgDb *x) { hgc_free(const_cast<HgDb *>(x)); })};
  return HG_OK;
}

int hg_db_tune(hg_database_t *db, const void *sample, size_t nbytes) {
  if (!db || (!sample && nbytes)) return HG_ERR_ARG;
  std::string err;
  HgDb *tuned = nullptr;
  if (hgc_tune(db->db.get(), static_cast<const uint8_t *>(sample), nbytes, &tuned, &err) != 0) return HG_ERR_COMPILE;  // (the database is unchanged)
  db->db = std::shared_ptr<const HgDb>(tuned, [](const HgDb *x) { hgc_free(const_cast<HgDb *>(x)); });
  return HG_OK;
}

void hg_db_release(hg_database_t *db) {
  delete db;  // (scanners created from it keep the compiled tables alive)
}

int hg_db_info(const hg_database_t *db, hg_db_info_t *info) {
  if (!db || !info) return HG_ERR_ARG;
  const HgDb &d = *db->db;
  info->n_patterns = static_cast<uint32_t>(d.patterns.size());
  info->n_always_on = static_cast<uint32_t>(d.slow.size());
  info->n_literal_anchored = info->n_patterns - info->n_always_on;
  info->n_factors = d.nreal_factors;
  info->n_windows = d.nreal_factors ? static_cast<uint32_t>(d.windows.size()) : 0;
  info->fold_mask = d.fold_mask;
  info->max_state_words = d.max_nw;
  info->table_bytes = static_cast<uint32_t>(d.pool.size() * 4);
  info->byte_windows = d.dense;
  return HG_OK;
}

int hg_scanner_create(const hg_database_t *db, int device, hg_scanner_t **scanner, char *err, size_t errlen) {
  if (!db || !scanner) return HG_ERR_ARG;
  *scanner = nullptr;
  HgScanner *sc = nullptr;
  std::string msg;
  int rc = HgScanner::create(db->db, device, &sc, &msg);
  if (rc != HG_OK) {
    put_err(err, errlen, msg);
    return rc;
  }
  *scanner = new hg_scanner{sc, {}};
  return HG_OK;
}

void hg_scanner_destroy(hg_scanner_t *scanner) {
  if (!scanner) return;
  delete scanner->sc;
  delete scanner;
}

const char *hg_scanner_error(const hg_scanner_t *scanner) { return scanner ? scanner->sc->last_error().c_str() : "null scanner"; }

int hg_scan_device(hg_scanner_t *scanner, const void *d_text, uint64_t nbytes, int buffer_size, uint64_t line_base, void *stream,
                   hg_scan_result_t *result) {
  if (!scanner || !result) return HG_ERR_ARG;
  HgScanOutput o{};
  int rc = scanner->sc->scan(d_text, nbytes, buffer_size, line_base, static_cast<hipStream_t>(stream), &o);
  if (rc != HG_OK) return rc;
  result->n_hits = o.n_hits;
  result->n_lines = o.n_pieces;
  result->n_candidates = o.n_cands;
  result->n_raw_hits = o.n_raw_hits;
  result->d_hits = reinterpret_cast<const hg_hit_t *>(o.d_hits);
  result->d_aux = reinterpret_cast<const hg_hit_aux_t *>(o.d_aux);
  result->ms_stream = o.ms_stream;
  result->ms_total = o.ms_total;
  result->reruns = o.reruns;
  result->stream_launches = o.stream_launches;
  result->joiner_tiles = o.joiner_tiles;
  result->joiner_launches = o.joiner_launches;
  result->reserved = 0;
  scanner->last = *result;
  return HG_OK;
}

int hg_copy_hits(hg_scanner_t *scanner, hg_hit_t *hits, hg_hit_aux_t *aux, uint64_t max) {
  if (!scanner || !hits) return HG_ERR_ARG;
  uint64_t n = scanner->last.n_hits < max ? scanner->last.n_hits : max;
  if (!n) return HG_OK;
  if (hipSetDevice(scanner->sc->device()) != hipSuccess) return HG_ERR_HIP;
  if (hipMemcpy(hits, scanner->last.d_hits, n * sizeof(hg_hit_t), hipMemcpyDeviceToHost) != hipSuccess) return HG_ERR_HIP;
  if (aux && hipMemcpy(aux, scanner->last.d_aux, n * sizeof(hg_hit_aux_t), hipMemcpyDeviceToHost) != hipSuccess) return HG_ERR_HIP;
  return HG_OK;
}

int hg_copy_hits_device(hg_scanner_t *scanner, void *d_dst, uint64_t max, void *stream) {
  if (!scanner || !d_dst) return HG_ERR_ARG;
  uint64_t n = scanner->last.n_hits < max ? scanner->last.n_hits : max;
  if (!n) return HG_OK;
  if (hipSetDevice(scanner->sc->device()) != hipSuccess) return HG_ERR_HIP;
  if (hipMemcpyAsync(d_dst, scanner->last.d_hits, n * sizeof(hg_hit_t), hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)) != hipSuccess)
    return HG_ERR_HIP;
  return HG_OK;
}

}  // extern "C"

// ---------------------------------------------------------------- guarded buffers (diagnostics / tests)
// A device buffer whose END is followed by reserved but UNMAPPED address space: any read or write past the buffer's size
// rounded up to 16 bytes is a GPU memory fault, wherever ordinary allocations would let it pass silently because the next
// bytes happen to be mapped.  (Round 1's one GPU fault was such a read: it took ~50 live contexts before a text buffer
// ended at the edge of a mapping.)  Built on the HIP virtual-memory API: reserve mapped + guard, map only the first part.
struct hg_guarded {
  void *base;
  size_t mapped, reserved;
  hipMemGenericAllocationHandle_t handle;
};
extern "C" {

int hg_debug_alloc_guarded(uint64_t nbytes, int device, void **d_ptr, void **guard_handle) {
  if (!d_ptr || !guard_handle) return HG_ERR_ARG;
  *d_ptr = *guard_handle = nullptr;
  if (hipSetDevice(device) != hipSuccess) return HG_ERR_HIP;
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = device;
  size_t gran = 0;
  if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum) != hipSuccess || !gran) return HG_ERR_HIP;
  const size_t usable = (static_cast<size_t>(nbytes) + 15) & ~static_cast<size_t>(15);
  auto g = std::make_unique<hg_guarded>();
  g->mapped = std::max<size_t>((usable + gran - 1) / gran * gran, gran);
  g->reserved = g->mapped + gran;
  if (hipMemAddressReserve(&g->base, g->reserved, gran, nullptr, 0) != hipSuccess) return HG_ERR_HIP;
  if (hipMemCreate(&g->handle, g->mapped, &prop, 0) != hipSuccess) {
    (void)hipMemAddressFree(g->base, g->reserved);
    return HG_ERR_HIP;
  }
  hipMemAccessDesc access{};
  access.location = prop.location;
  access.flags = hipMemAccessFlagsProtReadWrite;
  if (hipMemMap(g->base, g->mapped, 0, g->handle, 0) != hipSuccess || hipMemSetAccess(g->base, g->mapped, &access, 1) != hipSuccess) {
    (void)hipMemRelease(g->handle);
    (void)hipMemAddressFree(g->base, g->reserved);
    return HG_ERR_HIP;
  }
  *d_ptr = static_cast<char *>(g->base) + (g->mapped - usable);  // 16-byte aligned; [d_ptr, d_ptr + usable) ends at the guard
  *guard_handle = g.release();
  return HG_OK;
}

void hg_debug_free_guarded(void *guard_handle) {
  hg_guarded *g = static_cast<hg_guarded *>(guard_handle);
  if (!g) return;
  (void)hipDeviceSynchronize();
  (void)hipMemUnmap(g->base, g->mapped);
  (void)hipMemRelease(g->handle);
  (void)hipMemAddressFree(g->base, g->reserved);
  delete g;
}

int hg_debug_upload(void *d_dst, const void *src, uint64_t nbytes) {
  if (!nbytes) return HG_OK;
  if (!d_dst || !src) return HG_ERR_ARG;
  return hipMemcpy(d_dst, src, nbytes, hipMemcpyHostToDevice) == hipSuccess ? HG_OK : HG_ERR_HIP;
}

int hg_debug_download(void *dst, const void *d_src, uint64_t nbytes) {
  if (!nbytes) return HG_OK;
  if (!dst || !d_src) return HG_ERR_ARG;
  return hipMemcpy(dst, d_src, nbytes, hipMemcpyDeviceToHost) == hipSuccess ? HG_OK : HG_ERR_HIP;
}

}  // extern "C"

// ---------------------------------------------------------------- synthetic log
__global__ void hg_synth_kernel(uint8_t *text, uint64_t nbytes, HgSynthSpec sp) {
  uint64_t nblocks = (nbytes + HG_SYNTH_BLOCK - 1) / HG_SYNTH_BLOCK;
  for (uint64_t b = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; b < nblocks;
       b += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
    uint64_t off = b * HG_SYNTH_BLOCK;
    uint32_t len = static_cast<uint32_t>(nbytes - off < HG_SYNTH_BLOCK ? nbytes - off : HG_SYNTH_BLOCK);
    hg_synth_block(sp, sp.first_block + b, text + off, len);
  }
}

extern "C" {

int hg_synth_device(void *d_text, uint64_t nbytes, const hg_synth_spec_t *spec, int device, void *stream) {
  if (!d_text || !spec) return HG_ERR_ARG;
  if (hipSetDevice(device) != hipSuccess) return HG_ERR_HIP;
  uint32_t total = spec->n_needles ? spec->needle_off[spec->n_needles] : 0;
  uint8_t *d_needles = nullptr;
  uint32_t *d_off = nullptr;
  if (hipMalloc(reinterpret_cast<void **>(&d_needles), total + 16) != hipSuccess) return HG_ERR_HIP;
  if (hipMalloc(reinterpret_cast<void **>(&d_off), (spec->n_needles + 1) * 4 + 16) != hipSuccess) return HG_ERR_HIP;
  if (total) (void)hipMemcpy(d_needles, spec->needles, total, hipMemcpyHostToDevice);
  if (spec->n_needles) (void)hipMemcpy(d_off, spec->needle_off, (spec->n_needles + 1) * 4, hipMemcpyHostToDevice);
  HgSynthSpec sp{spec->seed, spec->first_block, spec->hit_per_million, spec->n_needles, d_needles, d_off};
  uint64_t nblocks = (nbytes + HG_SYNTH_BLOCK - 1) / HG_SYNTH_BLOCK;
  uint32_t grid = static_cast<uint32_t>(std::min<uint64_t>((nblocks + 63) / 64, 65536));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (nblocks) hipLaunchKernelGGL(hg_synth_kernel, dim3(grid), dim3(64), 0, st, static_cast<uint8_t *>(d_text), nbytes, sp);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(d_needles);
  (void)hipFree(d_off);
  return e == hipSuccess ? HG_OK : HG_ERR_HIP;
}

int hg_synth_host(uint8_t *text, uint64_t nbytes, const hg_synth_spec_t *spec) {
  if (!text || !spec) return HG_ERR_ARG;
  HgSynthSpec sp{spec->seed, spec->first_block, spec->hit_per_million, spec->n_needles, spec->needles, spec->needle_off};
  uint64_t nblocks = (nbytes + HG_SYNTH_BLOCK - 1) / HG_SYNTH_BLOCK;
  for (uint64_t b = 0; b < nblocks; b++) {
    uint64_t off = b * HG_SYNTH_BLOCK;
    uint32_t len = static_cast<uint32_t>(nbytes - off < HG_SYNTH_BLOCK ? nbytes - off : HG_SYNTH_BLOCK);
    hg_synth_block(sp, sp.first_block + b, text + off, len);
  }
  return HG_OK;
}

}  // extern "C"
