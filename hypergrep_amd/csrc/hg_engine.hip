// Host side of the device scanner: uploads the compiled database, owns the workspace in HBM and issues
// the launch sequence (stream pass -> tile scan -> confirm / always-on -> order + de-duplicate).
// There is no CPU scan path here: any HIP failure is reported as HG_ERR_HIP.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>

#include "hg_engine.h"
#include "hg_mem.h"

#include <rocprim/rocprim.hpp>

// kernels (hg_kernels.hip)
bool hg_launch_stream(const HgStreamArgs &a, uint32_t grid, hipStream_t stream);
bool hg_launch_stream_join(const HgStreamArgs &a, uint32_t grid, hipStream_t stream);
int hg_stream_blocks_per_cu(uint32_t filter_log2, uint32_t filter_wide, uint32_t dense);
__global__ void hg_tile_reduce_kernel(const HgTileSum *sums, uint64_t tile_begin, uint64_t tile_end, uint64_t bs1, HgTileElem *agg);
__global__ void hg_tile_spine_kernel(const HgTileElem *agg, uint32_t nblocks, uint64_t bs1, HgTileBase *block_base, HgTileBase *state);
__global__ void hg_tile_apply_kernel(const HgTileSum *sums, uint64_t tile_begin, uint64_t tile_end, uint64_t bs1, const HgTileBase *block_base,
                                     HgTileBase *bases);
__global__ void hg_tile_inner_kernel(const uint8_t *text, HgTileSum *sums, uint64_t tile_begin, uint64_t tile_end, uint64_t bs1);
__global__ void hg_verify_kernel(HgConfirmArgs a);
__global__ void hg_confirm_fast_kernel(HgConfirmArgs a, uint32_t blocks_per_mode);
__global__ void hg_confirm_generic_kernel(HgConfirmArgs a);
__global__ void hg_always_on_kernel(HgConfirmArgs a, uint32_t first, uint32_t last);
__global__ void hg_always_on_fast_kernel(HgConfirmArgs a);
__global__ void hg_always_on_finish_kernel(HgConfirmArgs a);
__global__ void hg_block_mark_kernel(HgConfirmArgs a, uint32_t *pattern_flags);
__global__ void hg_block_scan_kernel(HgConfirmArgs a, const uint32_t *pattern_flags);
__global__ void hg_block_small_kernel(HgDbView db, const uint8_t *h_text, uint32_t length, HgHit *h_out, uint32_t seg_cap, uint32_t *h_counts, uint32_t *d_done,
                                      uint32_t *h_flag, uint32_t seq, uint32_t ppw);
__global__ void hg_reset_kernel(uint32_t *state, uint32_t state_words, HgTileBase *final_state, uint64_t carry_start, uint64_t first_piece, uint32_t *fill, uint32_t nb, uint32_t *defer_count,
                                uint32_t ndefer);
__global__ void hg_fin_sort_small_kernel(const HgHit *hits, uint32_t *idx, const uint32_t *fill, uint32_t b_lo, uint32_t b_hi, uint32_t cap, uint32_t id_bits,
                                         uint32_t to_bits, uint32_t *kept_count, uint32_t *big_list, uint32_t *big_count, uint32_t big_stride);
template <uint32_t LCAP, bool IN_LDS>
__global__ void hg_fin_sort_big_kernel(const HgHit *hits, uint32_t *idx, const uint32_t *fill, const uint32_t *big_list, const uint32_t *big_count, uint32_t cap,
                                       uint32_t id_bits, uint32_t to_bits, uint32_t *kept_count, uint32_t *overflow, uint64_t *scratch_key, uint32_t *scratch_idx);
template <uint32_t THREADS>
__global__ void hg_fin_scan_kernel(uint32_t *kept_count, uint32_t b_lo, uint32_t b_hi, uint32_t *total, const uint32_t *fill, uint32_t cap, uint32_t *part, uint32_t epoch);
__global__ void hg_confirm_literal_kernel(HgConfirmArgs a);
__global__ void hg_verify_lean_kernel(HgConfirmArgs a);
__global__ void hg_fin_gather_kernel(const HgHit *hits, const HgHitAux *aux, const uint32_t *idx, const uint32_t *kept_base, const uint32_t *total, uint32_t b_lo,
                                     uint32_t b_hi, uint32_t cap, HgHit *oh, HgHitAux *oa);
__global__ void hg_key_kernel(const HgHit *hits, const HgHitAux *aux, const HgPattern *patterns, uint32_t n, uint64_t *key, uint32_t *idx);
__global__ void hg_line_key_kernel(const HgHit *hits, const uint32_t *perm, uint32_t n, uint64_t *key);
__global__ void hg_offset_kernel(uint32_t *idx, uint32_t n, uint32_t add);
__global__ void hg_key_packed_kernel(const HgHit *hits, const HgHitAux *aux, const HgPattern *patterns, uint32_t n, uint32_t id_bits, uint32_t to_bits,
                                     uint64_t *key, uint32_t *idx);
__global__ void hg_keep_kernel(const HgHit *hits, const HgHitAux *aux, const uint32_t *perm, const HgPattern *patterns, uint32_t n, uint8_t *keep);
__global__ void hg_scatter_kernel(const HgHit *hits, const HgHitAux *aux, const uint32_t *perm, const uint8_t *keep, const uint32_t *pos, uint32_t n,
                                  HgHit *oh, HgHitAux *oa, uint32_t *count);

// huge automata (hg_huge.hip)
size_t hg_huge_lds_bytes(uint32_t nw_max, uint32_t stage_cap);
bool hg_launch_confirm_huge(const HgConfirmArgs &a, uint32_t grid, uint32_t nw_max, uint32_t stage_cap, void *claim, uint32_t claim_mask, hipStream_t stream);
bool hg_launch_always_on_huge(const HgConfirmArgs &a, uint32_t grid, uint32_t nw_max, uint32_t stage_cap, uint32_t first, uint32_t last, hipStream_t stream);
bool hg_launch_block_huge(const HgConfirmArgs &a, uint32_t grid, uint32_t nw_max, uint32_t stage_cap, const uint32_t *pattern_flags, hipStream_t stream);

namespace {
constexpr int TS_BLOCK_TILES = 1024;  // must match hg_kernels.hip (256 threads x 4 tiles)
constexpr int STREAM_WG_WAVES = HG_STREAM_WG_WAVES;

template <typename T>
hipError_t upload(void **dst, const std::vector<T> &src, const char *name) {
  size_t bytes = std::max<size_t>(src.size() * sizeof(T), 16);
  hipError_t e = hgmem::dev_alloc(dst, bytes, name);
  if (e != hipSuccess) return e;
  if (!src.empty()) e = hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice);
  return e;
}
uint32_t bits_for(uint64_t v) {
  uint32_t b = 1;
  while (b < 64 && (v >> b)) b++;
  return b;
}
}  // namespace

HgEngineKnobs HgEngineKnobs::from_env() {
  HgEngineKnobs k;
  auto num = [](const char *name, uint64_t dflt) -> uint64_t {
    const char *env = std::getenv(name);
    return env ? std::strtoull(env, nullptr, 10) : dflt;
  };
  k.fin_target = std::max<uint64_t>(4, num("HG_FIN_TARGET", 48));
  k.no_bucket_finalize = std::getenv("HG_NO_BUCKET_FINALIZE") != nullptr;
  k.chunk_tiles = num("HG_CHUNK_TILES", 0);
  k.max_chunks = static_cast<uint32_t>(std::min<uint64_t>(num("HG_MAX_CHUNKS", 0), 1u << 20));
  if (const char *env = std::getenv("HG_CHUNK_WEIGHTS")) k.chunk_weights = env;
  k.stream_wgs_per_cu = static_cast<long>(num("HG_STREAM_WGS_PER_CU", 0));
  if (const char *env = std::getenv("HG_JOINER")) k.joiner = std::max(0l, std::min(2l, std::strtol(env, nullptr, 10)));
  k.no_early_finalize = std::getenv("HG_NO_EARLY_FINALIZE") != nullptr;
#ifdef HG_PROFILE_CONFIRM
  if (const char *env = std::getenv("HG_DEBUG_CONFIRM_MODES")) k.confirm_mode_mask = static_cast<uint32_t>(std::strtoul(env, nullptr, 0));
#endif
  k.confirm_blocks_per_cu = static_cast<long>(std::min<uint64_t>(num("HG_CONFIRM_BLOCKS_PER_CU", 0), 16));
  k.hit_limit = num("HG_HIT_LIMIT", 0);
  k.cand_limit = num("HG_CAND_LIMIT", 0);
  k.verbose = std::getenv("HG_VERBOSE") != nullptr;

  return k;
}

bool HgScanner::fail(hipError_t e, const char *what) {
  if (e == hipSuccess) return false;
  err_ = std::string(what) + ": " + hipGetErrorString(e);
  return true;
}

int HgScanner::create(std::shared_ptr<const HgDb> db, int device, HgScanner **out, std::string *err) {
  *out = nullptr;
  if (!db) {
    if (err) *err = "no database";
    return HG_ERR_ARG;
  }
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0) {
    if (err) *err = std::string("no HIP device available (") + hipGetErrorString(e) + "); the scan path has no CPU fallback";
    return HG_ERR_HIP;
  }
  if (device < 0 || device >= count) {
    if (err) *err = "device index out of range";
    return HG_ERR_ARG;
  }
  std::unique_ptr<HgScanner> s(new HgScanner());
  s->device_ = device;
  s->db_ = db;
  s->knobs_ = HgEngineKnobs::from_env();
#define HG_TRY(call, what)                         \
  if (s->fail((call), what)) {                     \
    if (err) *err = s->err_;                       \
    return HG_ERR_HIP;                             \
  }
  HG_TRY(hipSetDevice(device), "hipSetDevice");
  hipDeviceProp_t prop;
  HG_TRY(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties");
  s->num_cus_ = prop.multiProcessorCount;
  HG_TRY(upload(&s->d_patterns_, db->patterns, "d_patterns_"), "upload patterns");
  HG_TRY(upload(&s->d_pool_, db->pool, "d_pool_"), "upload tables");
  HG_TRY(upload(&s->d_factors_, db->factors, "d_factors_"), "upload factors");
  HG_TRY(upload(&s->d_windows_, db->windows, "d_windows_"), "upload windows");
  HG_TRY(upload(&s->d_bucket_, db->bucket_off, "d_bucket_"), "upload buckets");
  HG_TRY(upload(&s->d_disc_, db->disc, "d_disc_"), "upload discriminators");
  HG_TRY(upload(&s->d_bucket2_, db->bucket_off2, "d_bucket2_"), "upload buckets");
  HG_TRY(upload(&s->d_windows2_, db->windows2, "d_windows2_"), "upload windows");
  HG_TRY(upload(&s->d_wtab_, db->wtab, "d_wtab_"), "upload window table");
  HG_TRY(upload(&s->d_filter_, db->filter, "d_filter_"), "upload filter");
  HG_TRY(upload(&s->d_ext_, db->ext, "d_ext_"), "upload filter conditions");
  HG_TRY(upload(&s->d_slow_, db->slow, "d_slow_"), "upload always-on list");
  HG_TRY(upload(&s->d_groups_, db->groups, "d_groups_"), "upload always-on groups");
  s->view_.patterns = static_cast<const HgPattern *>(s->d_patterns_);
  s->view_.pool = static_cast<const uint32_t *>(s->d_pool_);
  s->view_.factors = static_cast<const HgFactor *>(s->d_factors_);
  s->view_.windows = static_cast<const HgWindow *>(s->d_windows_);
  s->view_.bucket_off = static_cast<const uint32_t *>(s->d_bucket_);
  s->view_.disc = static_cast<const uint16_t *>(s->d_disc_);
  s->view_.bucket_off2 = static_cast<const uint32_t *>(s->d_bucket2_);
  s->view_.windows2 = static_cast<const HgWindow *>(s->d_windows2_);
  s->view_.wtab = static_cast<const HgWinBucket *>(s->d_wtab_);
  s->view_.wtab_mask = db->wtab_mask;
  s->view_.wtab_first = db->wtab_first;
  s->view_.slow = static_cast<const uint32_t *>(s->d_slow_);
  s->view_.npatterns = static_cast<uint32_t>(db->patterns.size());
  s->view_.nslow = static_cast<uint32_t>(db->slow.size());
  s->view_.nslow_fast = db->nslow_fast;
  s->view_.nslow_grouped = db->nslow_grouped;
  s->view_.nslow_huge = db->nslow_huge;
  s->view_.ngroups = static_cast<uint32_t>(db->groups.size());
  s->view_.groups = static_cast<const HgSlowGroup *>(s->d_groups_);
  s->view_.fold_mask = db->fold_mask;
  s->view_.window_mask = db->window_mask;
  static_assert(HG_CNT_CURSORS == kMaxChunks, "one tile cursor per pipeline chunk");
  HG_TRY(hgmem::dev_alloc(&s->d_counters_, HG_ST_ALLOC_WORDS * 4, "d_state_"), "alloc state");  // the state block (hg_engine.h, HG_ST_*)
  HG_TRY(hipMemset(s->d_counters_, 0, HG_ST_ALLOC_WORDS * 4), "clear state");
  s->d_fin_total_ = s->d_counters_ + HG_ST_FIN_TOTAL;
  s->d_selected_ = s->d_counters_ + HG_ST_SELECTED;
  s->d_final_ = reinterpret_cast<HgTileBase *>(s->d_counters_ + HG_ST_FINAL);
  HG_TRY(hgmem::dev_alloc(&s->d_pflags_, db->patterns.size() * 4 + 16, "d_pflags_"), "alloc pattern flags");
  HG_TRY(hgmem::host_alloc(&s->h_counters_, (HG_CNT_WORDS + 4 + HG_DEFER_SHARDS) * 4, "h_counters_"), "alloc pinned");
  s->h_final_ = reinterpret_cast<HgTileBase *>(s->h_counters_ + HG_ST_FINAL);  // (the host copy of the state block)
  for (auto &ev : s->ev_) HG_TRY(hipEventCreate(&ev), "hipEventCreate");
  HG_TRY(hipStreamCreateWithFlags(&s->side_stream_, hipStreamNonBlocking), "hipStreamCreate");
  HG_TRY(hipEventCreateWithFlags(&s->ev_fin_early_, hipEventDisableTiming), "hipEventCreate");
  HG_TRY(hipEventCreateWithFlags(&s->ev_tile_done_, hipEventDisableTiming), "hipEventCreate");

  // (the bucket arrays of the finalize — 16 MiB — are allocated by the first pass that orders hits in buckets: a scratch that only
  // ever sees short hs_scan blocks never needs them)
  // (the per-chunk events of the two-stream pipeline are created by the first scan that is large enough to use it: a
  // process that keeps dozens of scanners for small files would otherwise hold thousands of events for nothing)
#undef HG_TRY
  *out = s.release();
  return HG_OK;
}

HgScanner::~HgScanner() {
  (void)hipSetDevice(device_);
  void *ptrs[] = {d_patterns_, d_pool_, d_factors_, d_windows_, d_bucket_, d_filter_, d_ext_, d_slow_, d_sums_, d_bases_, d_block_base_,
                  d_agg_, d_cands_, d_hits_raw_, d_hits_out_, d_aux_raw_, d_aux_out_,
                  d_key_a_, d_key_b_, d_perm_a_, d_perm_b_, d_keep_, d_counters_, d_temp_, d_seg_count_, d_pflags_, d_deferred_, d_defer_count_, d_seg_count2_, d_cands2_, d_disc_, d_bucket2_, d_windows2_, d_groups_, d_acc_hits_, d_acc_aux_, d_huge_claim_, d_wtab_};
  for (void *p : ptrs) hgmem::dev_free(p, "scanner");
  hgmem::host_free(h_counters_, "h_counters_");
  for (auto &ev : ev_)
    if (ev) (void)hipEventDestroy(ev);
  for (int i = 0; i < kMaxChunks; i++) {
    if (ev_k1_begin_[i]) (void)hipEventDestroy(ev_k1_begin_[i]);
    if (ev_k1_end_[i]) (void)hipEventDestroy(ev_k1_end_[i]);
    if (ev_side_done_[i]) (void)hipEventDestroy(ev_side_done_[i]);
  }
  if (side_stream_) (void)hipStreamDestroy(side_stream_);
  if (ev_fin_early_) (void)hipEventDestroy(ev_fin_early_);
  if (ev_tile_done_) (void)hipEventDestroy(ev_tile_done_);

  hgmem::dev_free(d_fin_fill_, "d_fin_fill_");
  hgmem::dev_free(d_fin_kept_, "d_fin_kept_");
  hgmem::dev_free(d_fin_big_, "d_fin_big_");
}

int HgScanner::alloc_cands(uint64_t n) {
  n = std::min<uint64_t>(n, 0x7FFFFFF0u);
  hgmem::dev_free(d_cands_, "d_cands_");
  d_cands_ = nullptr;
  if (fail(hgmem::dev_alloc(&d_cands_, n * sizeof(HgCand), "d_cands_"), "workspace alloc (candidates)")) return HG_ERR_HIP;
  hgmem::dev_free(d_cands2_, "d_cands2_");
  d_cands2_ = nullptr;
  if (fail(hgmem::dev_alloc(&d_cands2_, n * sizeof(HgCand), "d_cands2_"), "workspace alloc (candidates)")) return HG_ERR_HIP;
  hgmem::dev_free(d_deferred_, "d_deferred_");
  d_deferred_ = nullptr;
  // verified occurrences: one set of sharded lists per confirm mode the database uses
  uint32_t modes = 0;
  for (uint32_t m = 0; m < HG_CONFIRM_MODES; m++) modes += db_->n_confirm_mode[m] ? 1 : 0;
  if (fail(hgmem::dev_alloc(&d_deferred_, std::max<uint64_t>(modes, 1) * n * sizeof(HgDeferred), "d_deferred_"), "workspace alloc (deferred)")) return HG_ERR_HIP;
  // huge tier-0 expressions: the (piece, expression) claim table of hg_confirm_huge_kernel, two slots per occurrence a pass can hold
  hgmem::dev_free(d_huge_claim_, "d_huge_claim_");
  d_huge_claim_ = nullptr;
  huge_claim_slots_ = 0;
  if (db_->n_confirm_mode[4]) {
    uint64_t slots = 1u << 12;
    while (slots < 2 * n) slots <<= 1;
    if (fail(hgmem::dev_alloc(&d_huge_claim_, slots * 8, "d_huge_claim_"), "workspace alloc (claim table)")) return HG_ERR_HIP;
    huge_claim_slots_ = slots;
  }
  cand_cap_ = static_cast<uint32_t>(n);
  return HG_OK;
}

int HgScanner::alloc_hits(uint64_t n64) {
  const uint32_t n = static_cast<uint32_t>(std::min<uint64_t>(n64, 0x7FFFFFF0u));
  auto re = [&](auto *&ptr, size_t count, const char *name) -> bool {
    hgmem::dev_free(ptr, name);
    ptr = nullptr;
    return fail(hgmem::dev_alloc(&ptr, std::max<size_t>(count * sizeof(*ptr), 16), name), "workspace alloc (hits)");
  };
  if (re(d_hits_raw_, n, "d_hits_raw_") || re(d_hits_out_, n, "d_hits_out_") || re(d_aux_raw_, n, "d_aux_raw_") || re(d_aux_out_, n, "d_aux_out_") ||
      re(d_key_a_, n, "d_key_a_") || re(d_key_b_, n, "d_key_b_") || re(d_perm_a_, n, "d_perm_a_") || re(d_perm_b_, n, "d_perm_b_") || re(d_keep_, n, "d_keep_"))
    return HG_ERR_HIP;
  hit_cap_ = n;
  size_t t1 = 0, t2 = 0;
  (void)rocprim::radix_sort_pairs(nullptr, t1, d_key_a_, d_key_b_, d_perm_a_, d_perm_b_, n, 0, 64, hipStream_t(nullptr));
  (void)rocprim::exclusive_scan(nullptr, t2, d_keep_, d_perm_a_, 0u, n, rocprim::plus<uint32_t>(), hipStream_t(nullptr));
  size_t t3 = 0;
  (void)rocprim::merge(nullptr, t3, d_key_a_, d_key_a_, d_key_b_, d_perm_a_, d_perm_a_, d_perm_b_, n, n, rocprim::less<uint64_t>(), hipStream_t(nullptr));
  size_t need = std::max(std::max(t1, t2), t3) + 256;
  if (need > temp_bytes_) {
    hgmem::dev_free(d_temp_, "d_temp_");
    d_temp_ = nullptr;
    if (fail(hgmem::dev_alloc(&d_temp_, need, "d_temp_"), "workspace alloc (sort)")) return HG_ERR_HIP;
    temp_bytes_ = need;
  }
  return HG_OK;
}

int HgScanner::ensure(uint64_t nbytes) {
  uint64_t ntiles = std::max<uint64_t>((nbytes + HG_TILE_BYTES - 1) / HG_TILE_BYTES, 1);
  auto re = [&](auto *&ptr, size_t count, const char *name) -> bool {
    hgmem::dev_free(ptr, name);
    ptr = nullptr;
    return fail(hgmem::dev_alloc(&ptr, std::max<size_t>(count * sizeof(*ptr), 16), name), "workspace alloc");
  };
  if (ntiles > cap_tiles_) {
    uint64_t nblocks = (ntiles + TS_BLOCK_TILES - 1) / TS_BLOCK_TILES;
    if (re(d_sums_, ntiles, "d_sums_") || re(d_bases_, ntiles + 1, "d_bases_") || re(d_agg_, nblocks, "d_agg_") || re(d_block_base_, nblocks, "d_block_base_")) return HG_ERR_HIP;
    cap_tiles_ = ntiles;
  }
  if (!d_seg_count_) {
    max_segs_ = static_cast<uint32_t>(num_cus_) * 16;
    if (re(d_seg_count_, max_segs_, "d_seg_count_") || re(d_seg_count2_, max_segs_, "d_seg_count2_") || re(d_defer_count_, HG_CONFIRM_MODES * HG_DEFER_SHARDS, "d_defer_count_")) return HG_ERR_HIP;
  }
  // one candidate / hit per KiB of text to start with; grows (and the pass repeats) on overflow
  uint64_t want = std::max<uint64_t>(nbytes / 1024, 1u << 16);
  if (cand_cap_ < want && alloc_cands(want)) return HG_ERR_HIP;
  if (hit_cap_ < want && alloc_hits(want)) return HG_ERR_HIP;
  return HG_OK;
}

// (internal) the pass would need more hit records or pipeline chunks than one pass may have: the caller scans in segments
constexpr int HG_SPLIT = 1000;

int HgScanner::run_once(const uint8_t *text, uint64_t nbytes, uint64_t bs1, uint64_t line_base, const PassRange &range, bool block_mode, hipStream_t stream,
                        HgScanOutput *out, bool *overflow) {
  *overflow = false;
  const uint64_t tile_lo = range.tile_lo, tile_hi = range.tile_hi;
  const uint64_t ntiles = tile_hi - tile_lo;  // tiles of this pass
  // the bytes whose pieces this pass reports (buckets of the finalize count from own_lo)
  const uint64_t own_end = std::min<uint64_t>(range.own_hi, nbytes), own_len = own_end > range.own_lo ? own_end - range.own_lo : 0;
#define HG_TRY(call, what) \
  if (fail((call), what)) return HG_ERR_HIP;
  HG_TRY(hipEventRecord(ev_[0], stream), "event");
  // Bucketed emission + finalize (hg_fin_*): buckets of 2^fin_shift text bytes (1 KiB at least) by line start, at most
  // HG_FIN_MAX_BUCKETS of them, each a region of fin_cap records of the hit arrays.
  const uint32_t id_bits = bits_for(static_cast<uint64_t>(db_->max_id) + 1), to_bits = bits_for(bs1 + 1);
  // As many buckets as give ~30-50 records each (one wave orders up to 64 in registers; larger buckets go through LDS): from
  // the last pass's hits, else one hit per 8 KiB of text as a first guess.
  uint32_t fin_shift = 10, fin_nb = 1;  // (1 KiB buckets at least: a text with a report on every line has ~10 per KiB)
  if (own_len) {
    const uint64_t expect = std::max<uint64_t>(fin_expect_hits_, own_len >> 13);
    uint64_t want_nb = 1;
    const uint64_t per_bucket = knobs_.fin_target;
    while (want_nb * per_bucket < expect && want_nb < HG_FIN_MAX_BUCKETS) want_nb <<= 1;
    while (((own_len - 1) >> fin_shift) >= want_nb) fin_shift++;
    fin_nb = static_cast<uint32_t>((own_len - 1) >> fin_shift) + 1;
  }
  const uint32_t fin_cap = hit_cap_ / fin_nb;
  // (sort key of a bucket: line start inside the bucket | id | to | single; the raw records carry that start in the top
  // 24 bits of the line number, so line numbers must stay below 2^40)
  if (ntiles && !fin_fallback_ && fin_nb > fin_alloc_) {  // (64 K buckets at least, then by powers of two up to HG_FIN_MAX_BUCKETS)
    uint32_t want = 1u << 16;
    while (want < fin_nb) want <<= 1;
    hgmem::dev_free(d_fin_fill_, "d_fin_fill_");
    hgmem::dev_free(d_fin_kept_, "d_fin_kept_");
    hgmem::dev_free(d_fin_big_, "d_fin_big_");
    d_fin_fill_ = d_fin_kept_ = d_fin_big_ = nullptr;
    fin_alloc_ = 0;
    HG_TRY(hgmem::dev_alloc(&d_fin_fill_, static_cast<size_t>(want) * 4, "d_fin_fill_"), "alloc finalize buckets");
    HG_TRY(hgmem::dev_alloc(&d_fin_kept_, static_cast<size_t>(want) * 4, "d_fin_kept_"), "alloc finalize buckets");
    HG_TRY(hgmem::dev_alloc(&d_fin_big_, (2 * static_cast<size_t>(want) + 3 * 128) * 4, "d_fin_big_"), "alloc finalize buckets");  // (two work lists + the scan's partial sums)
    HG_TRY(hipMemsetAsync(d_fin_big_ + 2 * static_cast<size_t>(want), 0, 3 * 128 * 4, stream), "clear scan flags");
    fin_alloc_ = want;
  }
  const bool bucketed = ntiles && d_fin_fill_ && fin_cap && fin_shift <= 64 - HG_HIT_REL_SHIFT && fin_shift + id_bits + to_bits + 1 <= 64 &&
                        bits_for(line_base + nbytes + 1) <= HG_HIT_REL_SHIFT && !fin_fallback_ && !knobs_.no_bucket_finalize;
  uint32_t fin_done = 0;  // buckets finalized so far
  // one launch puts the device state in place (counters, cursors, finalize totals, tile-scan state, bucket fill levels, the
  // first chunk's verified-occurrence counts)
  hipLaunchKernelGGL(hg_reset_kernel, dim3(std::max<uint32_t>(1, std::min<uint32_t>((fin_nb + 255) / 256, 256))), dim3(256), 0, stream, d_counters_, static_cast<uint32_t>(HG_ST_ZERO_WORDS), d_final_,
                     range.cs0, range.piece0, d_fin_fill_, bucketed ? fin_nb : 0u, d_defer_count_, static_cast<uint32_t>(HG_CONFIRM_MODES * HG_DEFER_SHARDS));
  HG_TRY(hipGetLastError(), "reset launch");

  // Chunked pipeline (line mode, large buffers): the text is cut into tile-aligned chunks; the stream pass of chunk c+1
  // runs on the caller's stream while tile scan + verify + confirm of chunk c run on a side stream.  The stream pass
  // then leaves a quarter of the wave slots free so that the latency-bound side work is co-resident.
  const bool has_anchored = db_->patterns.size() > db_->slow.size();
  uint32_t nchunks = 1;
  // measured on MI355X (32 GiB, 256 patterns): 8 GiB chunks 15.2 ms, 2 GiB chunks 20.8 ms, no chunking 17.6 ms per pass —
  // the latency-bound side kernels need a few hundred thousand candidates per launch to fill the chip
  constexpr uint64_t kChunkTiles = 8ull << (30 - HG_TILE_SHIFT);
  if (!block_mode && ntiles >= 2 * kChunkTiles) nchunks = static_cast<uint32_t>(std::min<uint64_t>(kMaxChunks, ntiles / kChunkTiles));
  uint64_t chunk_tiles = ((ntiles + nchunks - 1) / nchunks + TS_BLOCK_TILES - 1) / TS_BLOCK_TILES * TS_BLOCK_TILES;
  if (knobs_.chunk_tiles) {  // tests: force the chunked pipeline on small buffers
    const uint64_t v = knobs_.chunk_tiles;
    if (!block_mode && v >= TS_BLOCK_TILES) chunk_tiles = std::max<uint64_t>(v / TS_BLOCK_TILES * TS_BLOCK_TILES, (ntiles + kMaxChunks - 1) / kMaxChunks / TS_BLOCK_TILES * TS_BLOCK_TILES + TS_BLOCK_TILES);
  }
  if (!block_mode && chunk_limit_tiles_ && chunk_tiles > chunk_limit_tiles_) chunk_tiles = chunk_limit_tiles_;  // (a chunk's candidates did not fit before)
  nchunks = ntiles ? static_cast<uint32_t>((ntiles + chunk_tiles - 1) / chunk_tiles) : 1;
  uint32_t max_chunks = static_cast<uint32_t>(kMaxChunks);
  if (knobs_.max_chunks) max_chunks = std::max(1u, std::min<uint32_t>(kMaxChunks, knobs_.max_chunks));  // (tests)
  if (nchunks > max_chunks) return HG_SPLIT;  // (chunks that shrank for a dense text: fewer tiles per pass then)
  std::vector<uint64_t> cut(nchunks + 1);  // chunk c = tiles [cut[c], cut[c + 1])
  for (uint32_t c = 0; c <= nchunks; c++) cut[c] = std::min<uint64_t>(tile_lo + static_cast<uint64_t>(c) * chunk_tiles, tile_hi);
  if (const char *env = knobs_.chunk_weights.empty() ? nullptr : knobs_.chunk_weights.c_str()) {  // experiment: relative chunk sizes, e.g. "10,10,8,4"
    std::vector<double> w;
    for (const char *q = env; *q;) {
      char *e = nullptr;
      const double v = std::strtod(q, &e);
      if (e == q) break;
      if (v > 0) w.push_back(v);
      q = *e ? e + 1 : e;
    }
    if (!block_mode && w.size() >= 2 && w.size() <= static_cast<size_t>(kMaxChunks) && ntiles >= w.size() * TS_BLOCK_TILES * 2) {
      double total = 0, run = 0;
      for (double v : w) total += v;
      nchunks = static_cast<uint32_t>(w.size());
      cut.assign(nchunks + 1, tile_lo);
      chunk_tiles = 0;
      for (uint32_t c = 0; c < nchunks; c++) {
        run += w[c];
        uint64_t end = c + 1 == nchunks ? tile_hi : tile_lo + static_cast<uint64_t>(static_cast<double>(ntiles) * run / total) / TS_BLOCK_TILES * TS_BLOCK_TILES;
        end = std::min<uint64_t>(std::max<uint64_t>(end, cut[c] + TS_BLOCK_TILES), tile_hi);
        cut[c + 1] = end;
        chunk_tiles = std::max<uint64_t>(chunk_tiles, end - cut[c]);
      }
    }
  }
  const bool piped = nchunks > 1;

  uint32_t wgs = 1, confirm_blocks = 1, always_blocks = 1;
  out->ms_stream = 0;
  out->stream_launches = ntiles ? nchunks : 0;
  out->joiner_launches = 0;
  if (ntiles) {
    if (stream_wgs_per_cu_ == 0) stream_wgs_per_cu_ = hg_stream_blocks_per_cu(db_->filter_log2, db_->filter_wide, db_->dense);
    uint32_t per_cu = piped ? std::max(1, stream_wgs_per_cu_ - 1) : stream_wgs_per_cu_;
    if (knobs_.stream_wgs_per_cu) {  // tuning knob: resident stream workgroups per CU
      const long v = knobs_.stream_wgs_per_cu;
      if (v >= 1 && v <= stream_wgs_per_cu_) per_cu = static_cast<uint32_t>(v);
    }
    auto grid_for = [&](uint32_t wgs_per_cu) {
      return static_cast<uint32_t>(std::min<uint64_t>((std::min<uint64_t>(chunk_tiles, ntiles) + STREAM_WG_WAVES - 1) / STREAM_WG_WAVES,
                                                      std::min<uint64_t>(static_cast<uint64_t>(num_cus_) * wgs_per_cu, max_segs_)));
    };
    const uint32_t wgs_shared = grid_for(per_cu);  // next to the side passes of the previous chunk
    // the first chunk streams alone: every workgroup slot (unless the grid was fixed by hand)
    const uint32_t wgs_alone = knobs_.stream_wgs_per_cu ? wgs_shared : grid_for(static_cast<uint32_t>(stream_wgs_per_cu_));
    // Joiners: the side passes of chunk c - 1 take about half as long as the stream pass of chunk c; behind them, on the side
    // stream, a second launch of the stream kernel (one more workgroup per CU, its own candidate segments) joins chunk c and
    // draws tiles from the same cursor until the chunk is used up.
    // (Only where a CU has room for it: the kernel's LDS allows three workgroups per CU and the shared launches use two;
    // and not when the last pass found the side passes to be the slower half (side_bound_: config 5).
    // Measured: 6.99 -> 6.84 ms per 32 GiB on config 3; HG_JOINER=0 turns it off.)
    // Not for texts so dense in candidates that the chunks had to shrink: their side passes are the slower half anyway, and
    // the joiner's segments would take workspace from the others.
    uint32_t joiner_wgs = (piped && stream_wgs_per_cu_ >= 3 && per_cu < static_cast<uint32_t>(stream_wgs_per_cu_) && chunk_limit_tiles_ == 0 && !side_bound_) ? static_cast<uint32_t>(num_cus_) : 0u;
    if (knobs_.joiner >= 0) joiner_wgs = piped ? static_cast<uint32_t>(knobs_.joiner) * static_cast<uint32_t>(num_cus_) : 0u;
    if (wgs_shared + joiner_wgs > max_segs_ || db_->filter_wide || db_->filter_log2 > 13) joiner_wgs = 0;  // (hg_launch_stream_join's instantiations)
    wgs = std::max(wgs_shared + joiner_wgs, wgs_alone);  // sizes the regrowth of the candidate segments
    hipStream_t side = piped ? side_stream_ : stream;
    if (piped && !ev_side_done_[0]) {
      for (int i = 0; i < kMaxChunks; i++) {
        HG_TRY(hipEventCreate(&ev_k1_begin_[i]), "hipEventCreate");
        HG_TRY(hipEventCreate(&ev_k1_end_[i]), "hipEventCreate");
        HG_TRY(hipEventCreate(&ev_side_done_[i]), "hipEventCreate");  // (timed: which of stream pass / side passes ends later, below)
      }
    }
    if (piped) {  // side stream starts after the counters / state are in place
      HG_TRY(hipEventRecord(ev_side_done_[kMaxChunks - 1], stream), "event");
      HG_TRY(hipStreamWaitEvent(side, ev_side_done_[kMaxChunks - 1], 0), "stream wait");
    }
    // finalize of the buckets [lo, hi) on stream `s`: order each bucket, report rules, positions, gather (hg_fin_*)
    auto launch_fin = [&](hipStream_t s, uint32_t lo, uint32_t hi, bool beside_stream = false) -> int {
      if (hi <= lo) return HG_OK;
      const uint32_t nbk = hi - lo, cu = static_cast<uint32_t>(num_cus_);
      HG_TRY(hipMemsetAsync(d_fin_total_ + 2, 0, 8, s), "memset work list");  // (larger buckets of this range: two size classes)
      hipLaunchKernelGGL(hg_fin_sort_small_kernel, dim3(std::min<uint32_t>((nbk + 3) / 4, cu * 8)), dim3(256), 0, s, d_hits_raw_, d_perm_a_, d_fin_fill_, lo, hi, fin_cap, id_bits, to_bits,
                         d_fin_kept_, d_fin_big_, d_fin_total_ + 2, fin_alloc_);
      hipLaunchKernelGGL((hg_fin_sort_big_kernel<HG_FIN_MEDIUM_CAP, true>), dim3(std::min<uint32_t>(nbk, cu * 4)), dim3(256), 0, s, d_hits_raw_, d_perm_a_, d_fin_fill_, d_fin_big_, d_fin_total_ + 2,
                         fin_cap, id_bits, to_bits, d_fin_kept_, d_selected_ + 1, static_cast<uint64_t *>(nullptr), static_cast<uint32_t *>(nullptr));
      // (scratch of the large class: the key / permutation arrays of the library sort, idle while the scanner emits into buckets)
      const uint32_t big_blocks = std::min<uint32_t>(std::min<uint32_t>(nbk, 64u), hit_cap_ / HG_FIN_BUCKET_CAP);
      if (big_blocks)
        hipLaunchKernelGGL((hg_fin_sort_big_kernel<HG_FIN_BUCKET_CAP, false>), dim3(big_blocks), dim3(256), 0, s, d_hits_raw_, d_perm_a_, d_fin_fill_, d_fin_big_ + fin_alloc_,
                           d_fin_total_ + 3, fin_cap, id_bits, to_bits, d_fin_kept_, d_selected_ + 1, d_key_a_, d_perm_b_);
      // (a block per 8192 buckets, 128 at most: their partial sums live behind the two work lists)
      const uint32_t scan_blocks = std::max<uint32_t>(1, std::min<uint32_t>(128, (nbk + 8191) / 8192));
      uint32_t *part = d_fin_big_ + 2 * static_cast<size_t>(fin_alloc_);
      const uint32_t epoch = ++fin_epoch_ ? fin_epoch_ : ++fin_epoch_;  // (never 0: the flags start out zeroed)
      if (beside_stream) hipLaunchKernelGGL(hg_fin_scan_kernel<512u>, dim3(scan_blocks), dim3(512), 0, s, d_fin_kept_, lo, hi, d_fin_total_, d_fin_fill_, fin_cap, part, epoch);
      else hipLaunchKernelGGL(hg_fin_scan_kernel<1024u>, dim3(scan_blocks), dim3(1024), 0, s, d_fin_kept_, lo, hi, d_fin_total_, d_fin_fill_, fin_cap, part, epoch);
      hipLaunchKernelGGL(hg_fin_gather_kernel, dim3(std::min<uint32_t>((nbk + 3) / 4, cu * 8)), dim3(256), 0, s, d_hits_raw_, d_aux_raw_, d_perm_a_, d_fin_kept_, d_fin_total_, lo, hi, fin_cap,
                         d_hits_out_, d_aux_out_);
      HG_TRY(hipGetLastError(), "finalize launch");
      return HG_OK;
    };
    bool fin_early = false;  // the buckets of all chunks but the last were finalized beside the last chunk's side passes
    for (uint32_t c = 0; c < nchunks; c++) {
      const uint64_t t0 = cut[c], t1 = cut[c + 1];
      const uint32_t wgs_c = c == 0 ? wgs_alone : wgs_shared;
      const uint32_t set = piped ? (c & 1u) : 0u;
      HgCand *cands = set ? d_cands2_ : d_cands_;
      uint32_t *seg_count = set ? d_seg_count2_ : d_seg_count_;
      if (piped && c >= 2) HG_TRY(hipStreamWaitEvent(stream, ev_side_done_[c - 2], 0), "stream wait");  // buffer set is free again

      HgStreamArgs sa{};
      sa.text = text;
      sa.nbytes = nbytes;
      sa.tile_begin = t0;
      sa.tile_end = t1;
      sa.db = view_;
      sa.filter = static_cast<const uint32_t *>(d_filter_);
      sa.filter_log2 = db_->filter_log2;
      sa.weights_a = db_->weights_a;
      sa.weights_b = db_->weights_b;
      sa.filter_wide = db_->filter_wide;
      sa.dense = db_->dense;
      sa.weights_c = db_->weights_c;
      // the chunk's workgroups draw runs of consecutive tiles from a cursor (hg_stream_kernel): two tiles per wave and draw
      sa.cursor_slot = HG_CNT_CURSOR0 + c;  // (zeroed by hg_reset_kernel with the rest of the state block)
      sa.ext = static_cast<const HgSlotInfo *>(d_ext_);
      sa.sums = d_sums_;
      sa.cands = cands;
      sa.seg_count = seg_count;
      const uint32_t joiners_c = c >= 1 ? joiner_wgs : 0u;
      const uint32_t segs_c = wgs_c + joiners_c;  // candidate segments of the chunk: one per stream workgroup
      // (the joiner's segments get a quarter of a stream workgroup's: it streams a few per cent of a chunk, and equal shares took a
      // third of the candidate workspace from the launch that fills it)
      sa.cand_seg_cap = static_cast<uint32_t>(static_cast<uint64_t>(cand_cap_) * 4 / (4ull * wgs_c + joiners_c));
      const uint32_t join_seg_cap = sa.cand_seg_cap / 4;
      sa.alone = (c == 0 && wgs_c == wgs_alone && !knobs_.stream_wgs_per_cu) ? 1u : 0u;
      sa.counters = d_counters_;
      HG_TRY(hipEventRecord(piped ? ev_k1_begin_[c] : ev_[1], stream), "event");
      if (!hg_launch_stream(sa, wgs_c, stream)) {
        err_ = "no stream kernel for this database's filter size / mode";
        return HG_ERR_ARG;
      }
      HG_TRY(hipGetLastError(), "hg_stream_kernel launch");
      HG_TRY(hipEventRecord(piped ? ev_k1_end_[c] : ev_[2], stream), "event");
      if (piped && bucketed && joiners_c && c + 1 == nchunks && !knobs_.no_early_finalize) {
        // The last chunk: the side stream is idle from the end of chunk c - 1's side passes to the end of this stream launch.
        // The buckets the earlier chunks have completed are finalized there, in front of the joiner (they used to be
        // finalized beside the last chunk's verify / confirm passes, competing with them for the chip: 430 us for what takes
        // 150 alone, and last to finish).
        const uint64_t prev_end = std::min<uint64_t>(t0 << HG_TILE_SHIFT, nbytes);
        const uint64_t settled = prev_end > bs1 ? prev_end - bs1 : 0;
        const uint32_t lim = static_cast<uint32_t>(std::min<uint64_t>((settled > range.own_lo ? settled - range.own_lo : 0) >> fin_shift, fin_nb));
        if (lim > fin_done) {
          if (int rc = launch_fin(side, fin_done, lim, true)) return rc;
          fin_done = lim;
        }
      }
      if (joiners_c) {  // (the side stream: behind the side passes of chunk c - 1, in front of those of chunk c)
        HgStreamArgs ja = sa;
        ja.cands = cands + static_cast<uint64_t>(wgs_c) * sa.cand_seg_cap;
        ja.cand_seg_cap = join_seg_cap;
        ja.seg_count = seg_count + wgs_c;
        ja.alone = 0;
        if (!hg_launch_stream_join(ja, joiners_c, side)) return HG_ERR_ARG;
        out->joiner_launches++;
        HG_TRY(hipGetLastError(), "hg_stream_kernel launch (joiner)");
      }
      if (piped) HG_TRY(hipStreamWaitEvent(side, ev_k1_end_[c], 0), "stream wait");

      HgConfirmArgs ca{};
      ca.text = text;
      ca.nbytes = nbytes;
      ca.tile_begin = t0;
      ca.tile_end = t1;
      ca.bs1 = bs1;
      ca.db = view_;
      ca.sums = d_sums_;
      ca.bases = d_bases_;
      ca.cands = cands;
      ca.seg_count = seg_count;
      ca.hits = d_hits_raw_;
      ca.aux = d_aux_raw_;
      ca.tmp_hits = d_hits_out_;  // free until the final select
      ca.tmp_aux = d_aux_out_;
      ca.cand_seg_cap = sa.cand_seg_cap;
      ca.join_seg0 = wgs_c;
      ca.join_seg_cap = join_seg_cap;
      ca.hit_cap = hit_cap_;
      ca.hit_direct = hit_direct_ ? 1u : 0u;
      ca.bucket_cap = bucketed ? fin_cap : 0u;
      ca.bucket_shift = fin_shift;
      ca.bucket_fill = d_fin_fill_;
      ca.counters = d_counters_;
      ca.own_lo = range.own_lo;
      ca.own_hi = range.own_hi;
      if (block_mode) {
        HG_TRY(hipMemsetAsync(d_pflags_, 0, db_->patterns.size() * 4, side), "memset pattern flags");
        if (has_anchored) hipLaunchKernelGGL(hg_block_mark_kernel, dim3(segs_c), dim3(256), 0, side, ca, d_pflags_);
        always_blocks = std::max<uint32_t>(static_cast<uint32_t>((db_->patterns.size() + 255) / 256), std::min<uint32_t>(db_->nhuge, 1024u));
        ca.hit_seg_cap = hit_cap_ / always_blocks;
        hipLaunchKernelGGL(hg_block_scan_kernel, dim3(static_cast<uint32_t>((db_->patterns.size() + 255) / 256)), dim3(256), 0, side, ca, d_pflags_);
        if (db_->nhuge && !hg_launch_block_huge(ca, std::min<uint32_t>(db_->nhuge, 1024u), db_->huge_max_nw, db_->huge_stage_words, d_pflags_, side)) {
          err_ = "the huge-automaton kernel cannot have its LDS";
          return HG_ERR_HIP;
        }
        HG_TRY(hipGetLastError(), "block-mode launch");
      } else {
        const uint64_t span = t1 - t0;
        const uint32_t nblocks = static_cast<uint32_t>((span + TS_BLOCK_TILES - 1) / TS_BLOCK_TILES);
        if (bs1 < HG_TILE_BYTES) {  // small-buffer mode: lines inside a tile can split, re-price the tile summaries
          uint32_t blocks = static_cast<uint32_t>(std::min<uint64_t>((span + 255) / 256, 4096));
          hipLaunchKernelGGL(hg_tile_inner_kernel, dim3(blocks), dim3(256), 0, side, text, d_sums_, t0, t1, bs1);
        }
        hipLaunchKernelGGL(hg_tile_reduce_kernel, dim3(nblocks), dim3(256), 0, side, d_sums_, t0, t1, bs1, d_agg_);
        hipLaunchKernelGGL(hg_tile_spine_kernel, dim3(1), dim3(256), 0, side, d_agg_, nblocks, bs1, d_block_base_, d_final_);
        hipLaunchKernelGGL(hg_tile_apply_kernel, dim3(nblocks), dim3(256), 0, side, d_sums_, t0, t1, bs1, d_block_base_, d_bases_);
        HG_TRY(hipGetLastError(), "tile scan launch");
        if (piped && c + 1 == nchunks) HG_TRY(hipEventRecord(ev_tile_done_, side), "event");  // (the early finalize starts behind the tile scan)
        if (has_anchored) {
          const uint32_t verify_blocks = segs_c * HG_CONFIRM_SPLIT;  // HG_CONFIRM_SPLIT blocks share candidate segment b
          uint32_t fast_modes = 0;
          const uint32_t mode_mask = knobs_.confirm_mode_mask;  // (all modes, except in profiling builds: HG_DEBUG_CONFIRM_MODES)
          for (uint32_t m = 0; m < 3; m++) fast_modes += (db_->n_confirm_mode[m] && ((mode_mask >> m) & 1u)) ? 1 : 0;
          // few, long-lived blocks per confirm routine (in units of 256 lanes per CU; 3 measured best next to the stream pass);
          // the last chunk's side passes have the chip to themselves
          uint32_t per_cu = c + 1 == nchunks ? 6 : 3;
          if (knobs_.confirm_blocks_per_cu) per_cu = static_cast<uint32_t>(knobs_.confirm_blocks_per_cu);
          const uint32_t mode_blocks = std::max<uint32_t>(HG_DEFER_SHARDS, static_cast<uint32_t>(num_cus_) * per_cu * (256 / HG_CONFIRM_THREADS));  // per_cu counts 256 lanes
          // huge automata (confirm mode 4): one-wave workgroups, as many per CU as their LDS allows (8 at most)
          const uint32_t huge_blocks = db_->n_confirm_mode[4] ? std::max<uint32_t>(HG_DEFER_SHARDS, static_cast<uint32_t>(num_cus_) * static_cast<uint32_t>(std::max<size_t>(1, std::min<size_t>(8, (160u << 10) / std::max<size_t>(hg_huge_lds_bytes(db_->huge_max_nw, db_->huge_stage_words), 1))))) : 0u;
          confirm_blocks = std::max(std::max(mode_blocks * std::max(fast_modes, 1u), verify_blocks), huge_blocks);  // the largest grid that stages hits
          ca.hit_seg_cap = hit_cap_ / confirm_blocks;
          ca.deferred = d_deferred_;
          ca.defer_count = d_defer_count_;
          ca.defer_shard_cap = cand_cap_ / HG_DEFER_SHARDS;
          for (uint32_t m = 0, next = 0; m < HG_CONFIRM_MODES; m++) {
            ca.mode_present[m] = (db_->n_confirm_mode[m] && ((mode_mask >> m) & 1u)) ? 1 : 0;
            ca.list_of_mode[m] = db_->n_confirm_mode[m] ? next++ : 0;
            ca.list_spread[m] = std::min<uint32_t>(HG_DEFER_SHARDS, std::max<uint32_t>(1, HG_DEFER_SHARDS / std::max<uint32_t>(1, db_->n_confirm_mode[m])) * defer_spread_boost_);
          }
          if (c > 0) HG_TRY(hipMemsetAsync(d_defer_count_, 0, HG_CONFIRM_MODES * HG_DEFER_SHARDS * 4, side), "memset deferred counts");  // (chunk 0: hg_reset_kernel)
          // (sets without automaton modes 1 / 2: the verify variant without the LDS staging area)
          if (ca.mode_present[1] || ca.mode_present[2]) hipLaunchKernelGGL(hg_verify_kernel, dim3(verify_blocks), dim3(256), 0, side, ca);
          else hipLaunchKernelGGL(hg_verify_lean_kernel, dim3(verify_blocks), dim3(256), 0, side, ca);
          const bool literal_only_set = ca.mode_present[0] && !ca.mode_present[1] && !ca.mode_present[2];
          if (literal_only_set) {
            hipLaunchKernelGGL(hg_confirm_literal_kernel, dim3(mode_blocks), dim3(256), 0, side, ca);  // (blocks of 256: as many lanes per CU as before)
          } else if (fast_modes) {
            hipLaunchKernelGGL(hg_confirm_fast_kernel, dim3(mode_blocks * fast_modes), dim3(HG_CONFIRM_THREADS), 0, side, ca, mode_blocks);
          }
          if (db_->n_confirm_mode[3]) hipLaunchKernelGGL(hg_confirm_generic_kernel, dim3(mode_blocks), dim3(256), 0, side, ca);
          if (huge_blocks && ((mode_mask >> 4) & 1u)) {
            HG_TRY(hipMemsetAsync(d_huge_claim_, 0, huge_claim_slots_ * 8, side), "memset claim table");
            if (!hg_launch_confirm_huge(ca, huge_blocks, db_->huge_max_nw, db_->huge_stage_words, d_huge_claim_, static_cast<uint32_t>(huge_claim_slots_ - 1), side)) {
              err_ = "the huge-automaton kernel cannot have its LDS";
              return HG_ERR_HIP;
            }
          }
          HG_TRY(hipGetLastError(), "confirm launch");
        }
        if (!db_->slow.empty()) {
          always_blocks = static_cast<uint32_t>(std::min<uint64_t>((span + 3) / 4, static_cast<uint64_t>(num_cus_) * 8));
          // (huge automata: one-wave workgroups, four per SIMD — the routine is a chain of LDS reads and ballots per byte)
          const uint32_t huge_always_blocks = db_->nslow_huge ? static_cast<uint32_t>(std::min<uint64_t>(span, static_cast<uint64_t>(num_cus_) * 16)) : 0u;
          ca.hit_seg_cap = hit_cap_ / std::max(always_blocks, huge_always_blocks);
          const uint32_t nfast = db_->nslow_fast, nhuge = db_->nslow_huge, nall = static_cast<uint32_t>(db_->slow.size()) - nhuge;  // [fast | scalar | huge]
          if (nfast) {
            // the match list lives in the (by now idle) verified-occurrence lists: cand_cap_ entries at least, a segment per block
            ca.deferred = d_deferred_;
            ca.always_count = seg_count;  // this chunk's candidate segment counts were consumed by the verify pass
            ca.always_list_cap = cand_cap_ / always_blocks;
            hipLaunchKernelGGL(hg_always_on_fast_kernel, dim3(always_blocks), dim3(256), 0, side, ca);
            hipLaunchKernelGGL(hg_always_on_finish_kernel, dim3(always_blocks), dim3(256), 0, side, ca);
          }
          if (nall > nfast) hipLaunchKernelGGL(hg_always_on_kernel, dim3(always_blocks), dim3(256), 0, side, ca, nfast, nall);
          if (nhuge && !hg_launch_always_on_huge(ca, huge_always_blocks, db_->huge_max_nw, db_->huge_stage_words, nall, nall + nhuge, side)) {
            err_ = "the huge-automaton kernel cannot have its LDS";
            return HG_ERR_HIP;
          }
          always_blocks = std::max(always_blocks, huge_always_blocks);  // (sizes the regrowth of the staging segments)
          HG_TRY(hipGetLastError(), "hg_always_on_kernel launch");
        }
      }
      if (piped && bucketed && c + 1 == nchunks && c >= 1 && !knobs_.no_early_finalize) {
        // The last stream launch is queued.  Behind it, on this stream, the buckets that the earlier chunks have completed
        // are finalized WHILE the side stream works through the last chunk's verify / confirm passes (both have the chip to
        // themselves by then); only the last chunk's buckets remain for after those.  (Finalizing a chunk's buckets beside
        // the NEXT chunk's stream pass was tried: it slowed the stream pass by more than it saved.)
        const uint64_t prev_end = std::min<uint64_t>(t0 << HG_TILE_SHIFT, nbytes);  // a later hit's line starts less than bs1 bytes before it
        const uint64_t settled = prev_end > bs1 ? prev_end - bs1 : 0;           // ... pieces that start below this have all their hits
        const uint32_t lim = static_cast<uint32_t>(std::min<uint64_t>((settled > range.own_lo ? settled - range.own_lo : 0) >> fin_shift, fin_nb));
        if (lim > fin_done) {
          HG_TRY(hipStreamWaitEvent(stream, ev_side_done_[c - 1], 0), "stream wait");  // the earlier chunks' hits are all in their buckets
          HG_TRY(hipStreamWaitEvent(stream, ev_tile_done_, 0), "stream wait");        // ... and the last chunk's tile scan (a one-block latency chain) is through
          if (int rc = launch_fin(stream, fin_done, lim)) return rc;
          HG_TRY(hipEventRecord(ev_fin_early_, stream), "event");
          fin_done = lim;
          fin_early = true;
        }
      }
      if (bucketed && c + 1 == nchunks) {  // the side passes of the last chunk are queued: order what is left
        if (fin_early) HG_TRY(hipStreamWaitEvent(side, ev_fin_early_, 0), "stream wait");  // (shared totals / work list: one range at a time)
        if (int rc = launch_fin(side, fin_done, fin_nb)) return rc;
        fin_done = fin_nb;
      }
      if (piped) HG_TRY(hipEventRecord(ev_side_done_[c], side), "event");
    }
    if (piped) HG_TRY(hipStreamWaitEvent(stream, ev_side_done_[nchunks - 1], 0), "stream wait");
  } else {
    HG_TRY(hipEventRecord(ev_[1], stream), "event");
    HG_TRY(hipEventRecord(ev_[2], stream), "event");
  }
  HG_TRY(hipMemcpyAsync(h_counters_, d_counters_, HG_ST_WORDS * 4, hipMemcpyDeviceToHost, stream), "copy state");  // the whole state block in one copy
  if (bucketed) HG_TRY(hipEventRecord(ev_[3], stream), "event");
  HG_TRY(hipStreamSynchronize(stream), "stream sync (scan kernels)");
  if (piped) {
    for (uint32_t c = 0; c < nchunks; c++) {
      float ms = 0;
      (void)hipEventElapsedTime(&ms, ev_k1_begin_[c], ev_k1_end_[c]);
      out->ms_stream += ms;
    }
    // Which half of the pipeline set the pace: did the side passes of chunk c - 1 end after the stream launch of chunk c?
    // (config 5: yes — 42 M candidates per pass.)  Then the side stream has no idle window for the next pass to use.
    uint32_t late = 0;
    for (uint32_t c = 1; c < nchunks; c++) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, ev_k1_end_[c], ev_side_done_[c - 1]) == hipSuccess && ms > 0) late++;
    }
    side_bound_ = late * 2 > nchunks - 1;
  }

  out->joiner_tiles = h_counters_[HG_CNT_JOIN_TILES];
  const uint64_t n_cands = h_counters_[HG_CNT_CANDS];
  const uint64_t n_raw = bucketed ? h_counters_[HG_ST_FIN_TOTAL + 1] : h_counters_[HG_CNT_HITS];
  const uint64_t cand_need = h_counters_[HG_CNT_CAND_NEED], hit_need = h_counters_[HG_CNT_HIT_NEED];
  const uint64_t defer_need = h_counters_[HG_CNT_DEFER_NEED];
  const bool fin_overflow = bucketed && h_counters_[HG_ST_SELECTED + 1] != 0;  // a bucket beyond what one block sorts
  // Hit records one pass may hold (2^28: 8 GiB each of raw and ordered records); a buffer with more is scanned in segments
  // whose ordered hits are put one after the other (scan_segments).
  uint64_t kHitLimit = 1ull << 28;
  if (knobs_.hit_limit) kHitLimit = std::max<uint64_t>(1u << 10, knobs_.hit_limit);  // (tests)
  if (!block_mode && (n_raw > kHitLimit || h_counters_[HG_CNT_HITS_WRAPPED])) return HG_SPLIT;
  if (cand_need || defer_need || hit_need || fin_overflow || (!bucketed && n_raw > hit_cap_)) {
    // a private segment, a bucket or the compact hit array was too small: grow and let the caller repeat the pass
    if (cand_need || defer_need) {
      // Candidate segments fill evenly (the stream workgroups draw their tiles on demand) and so do the position-sharded
      // lists; a pattern-keyed list (automaton modes) overflows when ONE expression owns most occurrences: its occurrences
      // are then spread over more lists instead of sizing every list for it.
      if (defer_need && defer_spread_boost_ < HG_DEFER_SHARDS) defer_spread_boost_ *= 4;
      uint64_t want = std::max<uint64_t>((cand_need + cand_need / 4 + 64) * wgs, (defer_need + defer_need / 4 + 64) * HG_DEFER_SHARDS);
      want = std::max<uint64_t>(want, static_cast<uint64_t>(cand_cap_) * 2);
      // The workspace holds ONE chunk's candidates (two buffer sets): when that would pass 2^30 records (16 GiB a set) the
      // chunks get smaller instead — a text that fits in HBM always scans, a very dense one in more, smaller chunks.
      uint64_t kCandLimit = 1ull << 30;
      if (knobs_.cand_limit) kCandLimit = std::max<uint64_t>(1u << 16, knobs_.cand_limit);  // (tests)
      if (want > kCandLimit) {
        const uint64_t cur = (std::min<uint64_t>(chunk_tiles, ntiles) + TS_BLOCK_TILES - 1) / TS_BLOCK_TILES * TS_BLOCK_TILES;
        if (block_mode || cur <= TS_BLOCK_TILES) {
          err_ = "the candidate limit is below what one 16 MiB chunk of this text produces";  // (only with HG_CAND_LIMIT lowered: 16 MiB hold 2^24 positions)
          return HG_ERR_ARG;
        }
        chunk_limit_tiles_ = std::max<uint64_t>(TS_BLOCK_TILES, cur / 2 / TS_BLOCK_TILES * TS_BLOCK_TILES);
        want = std::min<uint64_t>(want / 2, kCandLimit);
      }
      if (want > cand_cap_) {
        int rc = alloc_cands(want);
        if (rc) return rc;
      }
    }
    if (bucketed && (hit_need || fin_overflow)) {
      // hit_need = the fullest bucket's demand.  Equal bucket regions are fine while the hits are spread; when one bucket
      // holds thousands of them (every match end of an all-matches expression on one long line) the scanner leaves bucketed
      // emission for good: compact array + library sort.
      const uint64_t want = (hit_need + hit_need / 4 + 16) * fin_nb;
      fin_expect_hits_ = std::max<uint64_t>(fin_expect_hits_ * 2, n_raw);  // (more, smaller buckets next time)
      if (fin_overflow || hit_need > HG_FIN_BUCKET_CAP || want > (128ull << 20)) {
        fin_fallback_ = true;
      } else {
        int rc = alloc_hits(std::max<uint64_t>(want, static_cast<uint64_t>(hit_cap_) * 2));
        if (rc) return rc;
      }
    } else if (hit_need || n_raw > hit_cap_) {
      uint64_t want = std::max<uint64_t>((hit_need + hit_need / 4 + 64) * std::max(confirm_blocks, always_blocks), n_raw + n_raw / 4);
      // equal segments sized for the fullest block: fine while the hits are spread, absurd when one block holds most of them
      // (every match end of an all-matches expression on one very long line).  Past 64 M records (4 GiB of workspace) or 16
      // times the hits actually seen, the segments stop growing and full blocks append to the compact array directly.
      constexpr uint64_t kSegmentLimit = 64ull << 20;
      const uint64_t by_total = n_raw + n_raw / 4 + 4096;
      if (want > kSegmentLimit || want > 16 * by_total) {
        hit_direct_ = true;
        want = std::min<uint64_t>(want, std::max<uint64_t>(by_total, std::min<uint64_t>(16 * by_total, kSegmentLimit)));
      }
      const uint64_t most = kHitLimit + kHitLimit / 4 + 4096;
      want = std::max<uint64_t>(want, std::min<uint64_t>(static_cast<uint64_t>(hit_cap_) * 2, most));
      if (want > most) {
        if (!block_mode) return HG_SPLIT;
        err_ = "more than 2^28 reports for one block";  // (block mode: one scan unit of at most 2 GiB, nothing to cut at)
        return HG_ERR_ARG;
      }
      int rc = alloc_hits(want);
      if (rc) return rc;
    }
    *overflow = true;
    return HG_OK;
  }

  fin_expect_hits_ = n_raw;
  // (a pass that stops short of the text's end leaves no piece count: scan_segments takes it from its last pass)
  uint64_t n_pieces = block_mode ? 1 : !range.last ? 0 : h_final_->L - line_base + (nbytes > h_final_->cs ? hg_pieces(nbytes - h_final_->cs, bs1) : 0);
  uint32_t n = static_cast<uint32_t>(n_raw);
  uint32_t kept = 0;
  if (n && !bucketed) {  // compact array + library sort (scanners that left bucketed emission, keys wider than 64 bits)
    const HgPattern *pats = static_cast<const HgPattern *>(d_patterns_);
    uint32_t blocks = (n + 255) / 256;
    // order by (line, id, to, single-after-multi): one radix sort over exactly the bits in use when they fit in 64, else two
    const uint32_t line_bits = bits_for(line_base + (range.last ? n_pieces : nbytes) + 1);
    const uint32_t *perm = nullptr;
    uint32_t *pos = nullptr;
    size_t tb = temp_bytes_;
    if (line_bits + id_bits + to_bits + 1 <= 64) {
      hipLaunchKernelGGL(hg_key_packed_kernel, dim3(blocks), dim3(256), 0, stream, d_hits_raw_, d_aux_raw_, pats, n, id_bits, to_bits, d_key_a_, d_perm_a_);
      HG_TRY(rocprim::radix_sort_pairs(d_temp_, tb, d_key_a_, d_key_b_, d_perm_a_, d_perm_b_, n, 0, line_bits + id_bits + to_bits + 1, stream), "radix sort");
      perm = d_perm_b_;
      pos = d_perm_a_;
    } else {
      hipLaunchKernelGGL(hg_key_kernel, dim3(blocks), dim3(256), 0, stream, d_hits_raw_, d_aux_raw_, pats, n, d_key_a_, d_perm_a_);
      HG_TRY(rocprim::radix_sort_pairs(d_temp_, tb, d_key_a_, d_key_b_, d_perm_a_, d_perm_b_, n, 0, 64, stream), "radix sort (id, to)");
      hipLaunchKernelGGL(hg_line_key_kernel, dim3(blocks), dim3(256), 0, stream, d_hits_raw_, d_perm_b_, n, d_key_a_);
      tb = temp_bytes_;
      HG_TRY(rocprim::radix_sort_pairs(d_temp_, tb, d_key_a_, d_key_b_, d_perm_b_, d_perm_a_, n, 0, std::min<uint32_t>(64, line_bits), stream), "radix sort (line)");
      perm = d_perm_a_;
      pos = d_perm_b_;
    }
    hipLaunchKernelGGL(hg_keep_kernel, dim3(blocks), dim3(256), 0, stream, d_hits_raw_, d_aux_raw_, perm, pats, n, d_keep_);
    tb = temp_bytes_;
    HG_TRY(rocprim::exclusive_scan(d_temp_, tb, d_keep_, pos, 0u, n, rocprim::plus<uint32_t>(), stream), "scan");
    hipLaunchKernelGGL(hg_scatter_kernel, dim3(blocks), dim3(256), 0, stream, d_hits_raw_, d_aux_raw_, perm, d_keep_, pos, n, d_hits_out_, d_aux_out_, d_selected_);
    HG_TRY(hipGetLastError(), "finalize launch");
    HG_TRY(hipMemcpyAsync(h_counters_ + HG_ST_SELECTED, d_selected_, 4, hipMemcpyDeviceToHost, stream), "copy count");
  }
  if (!bucketed) {
    HG_TRY(hipEventRecord(ev_[3], stream), "event");
    HG_TRY(hipStreamSynchronize(stream), "stream sync (finalize)");
  }
  if (n) kept = bucketed ? h_counters_[HG_ST_FIN_TOTAL] : h_counters_[HG_ST_SELECTED];
  out->n_hits = kept;
  out->n_pieces = n_pieces;
  out->n_cands = n_cands;
  out->n_raw_hits = n_raw;
  out->d_hits = d_hits_out_;
  out->d_aux = d_aux_out_;
  out->ms_total = 0;
  if (out->ms_stream == 0) (void)hipEventElapsedTime(&out->ms_stream, ev_[1], ev_[2]);
  (void)hipEventElapsedTime(&out->ms_total, ev_[0], ev_[3]);
#undef HG_TRY
  return HG_OK;
}

uint32_t HgScanner::launch_block_small(const uint8_t *h_text, uint32_t nbytes, hipStream_t stream, HgHit *h_out, uint32_t *h_counts, uint32_t *h_flag, uint32_t seq) {
  // 32 expressions per workgroup while 64 segments hold the set (their tables then usually fit in LDS), else 256
  const uint32_t ppw = view_.npatterns <= 32u * 64u ? 32u : 256u;
  const uint32_t segs = (view_.npatterns + ppw - 1) / ppw;
  if (nbytes == 0 || nbytes > HG_BLOCK_SMALL_MAX || segs == 0 || segs > 64 || db_->nhuge) return 0;  // (huge automata: the general path)
  if (hipSetDevice(device_) != hipSuccess) return 0;
  hipLaunchKernelGGL(hg_block_small_kernel, dim3(segs), dim3(256), 0, stream, view_, h_text, nbytes, h_out, static_cast<uint32_t>(HG_BLOCK_SMALL_SEG), h_counts,
                     d_counters_ + HG_ST_BLOCK_DONE, h_flag, seq, ppw);
  return hipGetLastError() == hipSuccess ? segs : 0;
}

int HgScanner::scan_block(const void *d_text, uint64_t nbytes, hipStream_t stream, HgScanOutput *out) {
  return scan_impl(d_text, nbytes, 0x7FFFFFFF, 0, true, stream, out);
}

int HgScanner::scan(const void *d_text, uint64_t nbytes, int buffer_size, uint64_t line_base, hipStream_t stream, HgScanOutput *out) {
  return scan_impl(d_text, nbytes, buffer_size, line_base, false, stream, out);
}

int HgScanner::scan_impl(const void *d_text, uint64_t nbytes, int buffer_size, uint64_t line_base, bool block_mode, hipStream_t stream,
                         HgScanOutput *out) {
  if (!out || (!d_text && nbytes) || buffer_size < 2) {
    err_ = "invalid arguments";
    return HG_ERR_ARG;
  }
  if ((reinterpret_cast<uintptr_t>(d_text) & 15u) != 0) {
    err_ = "text pointer must be 16-byte aligned";
    return HG_ERR_ARG;
  }
  const uint64_t bs1 = static_cast<uint64_t>(buffer_size) - 1;
  if (fail(hipSetDevice(device_), "hipSetDevice")) return HG_ERR_HIP;
  std::memset(out, 0, sizeof(*out));
  if (hgmem::log_file()) hgmem::note("scan  %p text %p .. %p  %llu  bs %d block %d\n", static_cast<void *>(this), d_text, static_cast<const void *>(static_cast<const char *>(d_text) + nbytes), static_cast<unsigned long long>(nbytes), buffer_size, block_mode ? 1 : 0);
  int rc = ensure(nbytes);
  if (rc) return rc;
  const uint8_t *text = static_cast<const uint8_t *>(d_text);
  const uint64_t ntiles = (nbytes + HG_TILE_BYTES - 1) / HG_TILE_BYTES;
  const PassRange whole{0, ntiles, 0, line_base, 0, ~0ull, true};
  uint32_t reruns = 0;
  for (;;) {
    bool overflow = false;
    rc = run_once(text, nbytes, bs1, line_base, whole, block_mode, stream, out, &overflow);
    if (rc == HG_SPLIT) break;
    if (rc) return rc;
    if (!overflow) break;
    if (++reruns > 16) {
      err_ = "workspace kept overflowing";
      return HG_ERR_NOMEM;
    }
    if (knobs_.verbose) std::fprintf(stderr, "hypergrep_amd: workspace grown (cands %u, hits %u), repeating the pass\n", cand_cap_, hit_cap_);
  }
  if (rc == HG_SPLIT) {
    // More reports (or pipeline chunks) than one pass may have: the buffer is scanned in 2, 4, 8 ... segments.
    for (uint32_t nsegments = 2;; nsegments *= 2) {
      bool too_many = false;
      rc = scan_segments(text, nbytes, bs1, line_base, stream, out, nsegments, &too_many);
      if (rc) return rc;
      if (!too_many) break;
      if (nsegments >= (1u << 16)) {
        err_ = "a single stretch of the text holds more reports than one pass may have";
        return HG_ERR_NOMEM;
      }
    }
  }
  out->reruns = reruns;
  return HG_OK;
}

// The buffer in `nsegments` passes.  Segment s reports the pieces whose first scanned byte lies in its stretch of the text
// (whole tiles) and scans on past the stretch's end for as long as such a piece can reach (bs1 bytes);
// hits of pieces that belong to a neighbour are dropped where they are emitted (HitSink::push), so every piece is ordered
// and filtered (SINGLEMATCH / duplicate rules) in exactly one pass.  The tile-scan state (carry-in line start, piece index)
// at a segment's first tile is read from the previous pass, which has scanned past it.  The passes' ordered hits are put one
// after the other: segments are in text order, so is their concatenation.  *too_many: some segment still overflowed a pass.
int HgScanner::scan_segments(const uint8_t *text, uint64_t nbytes, uint64_t bs1, uint64_t line_base, hipStream_t stream, HgScanOutput *out, uint32_t nsegments,
                             bool *too_many) {
  *too_many = false;
  const uint64_t ntiles = (nbytes + HG_TILE_BYTES - 1) / HG_TILE_BYTES;
  const uint64_t reach = (bs1 >> HG_TILE_SHIFT) + 2;  // tiles a piece that starts inside a stretch can extend past its end
  constexpr uint64_t kAlign = 16;  // (tiles; real segments are gigabytes)
  const uint64_t seg_tiles = ((ntiles + nsegments - 1) / nsegments + kAlign - 1) / kAlign * kAlign;
  if (seg_tiles <= reach) {  // (scan buffers of gigabytes on a text that needs many segments)
    err_ = "more reports than one pass may have, and the scan buffer size leaves no room for segments";
    return HG_ERR_NOMEM;
  }
  fin_fallback_ = false;  // (a smaller stretch: bucketed emission gets another chance)
  fin_expect_hits_ /= nsegments;
  uint64_t acc = 0, cands = 0, raw = 0, cs0 = 0, piece0 = line_base;
  uint32_t reruns = 0, launches = 0, join_launches = 0;
  uint64_t join_tiles = 0;
  float ms_stream = 0, ms_total = 0;
  HgScanOutput part{};
  for (uint64_t lo = 0; lo < ntiles; lo += seg_tiles) {
    const bool last = lo + seg_tiles >= ntiles;
    const PassRange range{lo, last ? ntiles : std::min<uint64_t>(ntiles, lo + seg_tiles + reach), cs0, piece0, lo << HG_TILE_SHIFT, last ? ~0ull : (lo + seg_tiles) << HG_TILE_SHIFT, last};
    for (uint32_t tries = 0;; tries++) {
      bool overflow = false;
      std::memset(&part, 0, sizeof part);
      const int rc = run_once(text, nbytes, bs1, line_base, range, false, stream, &part, &overflow);
      if (rc == HG_SPLIT) {
        *too_many = true;
        return HG_OK;
      }
      if (rc) return rc;
      if (!overflow) break;
      reruns++;
      if (tries > 16) {
        err_ = "workspace kept overflowing";
        return HG_ERR_NOMEM;
      }
    }
    // this pass's ordered hits behind those of the earlier segments
    if (acc + part.n_hits > acc_cap_) {
      // sized from the hits so far, the share of the text they came from, and a quarter on top
      const uint64_t done = std::max<uint64_t>(std::min<uint64_t>(ntiles, lo + seg_tiles), 1);
      const uint64_t guess = static_cast<uint64_t>(static_cast<double>(acc + part.n_hits) * static_cast<double>(ntiles) / static_cast<double>(done) * 1.25) + 4096;
      const uint64_t cap = std::max<uint64_t>(acc + part.n_hits, guess);
      HgHit *nh = nullptr;
      HgHitAux *na = nullptr;
      if (fail(hgmem::dev_alloc(&nh, cap * sizeof(HgHit), "d_acc_hits_"), "alloc (segment hits)") || fail(hgmem::dev_alloc(&na, cap * sizeof(HgHitAux), "d_acc_aux_"), "alloc (segment hits)")) {
        hgmem::dev_free(nh, "d_acc_hits_");
        return HG_ERR_HIP;
      }
      if (acc) {
        if (fail(hipMemcpyAsync(nh, d_acc_hits_, acc * sizeof(HgHit), hipMemcpyDeviceToDevice, stream), "copy") ||
            fail(hipMemcpyAsync(na, d_acc_aux_, acc * sizeof(HgHitAux), hipMemcpyDeviceToDevice, stream), "copy") || fail(hipStreamSynchronize(stream), "sync"))
          return HG_ERR_HIP;
      }
      hgmem::dev_free(d_acc_hits_, "d_acc_hits_");
      hgmem::dev_free(d_acc_aux_, "d_acc_aux_");
      d_acc_hits_ = nh;
      d_acc_aux_ = na;
      acc_cap_ = cap;
    }
    if (part.n_hits) {
      if (fail(hipMemcpyAsync(d_acc_hits_ + acc, part.d_hits, part.n_hits * sizeof(HgHit), hipMemcpyDeviceToDevice, stream), "copy") ||
          fail(hipMemcpyAsync(d_acc_aux_ + acc, part.d_aux, part.n_hits * sizeof(HgHitAux), hipMemcpyDeviceToDevice, stream), "copy"))
        return HG_ERR_HIP;
    }
    acc += part.n_hits;
    cands += part.n_cands;
    raw += part.n_raw_hits;
    ms_stream += part.ms_stream;
    ms_total += part.ms_total;
    launches += part.stream_launches;
    join_launches += part.joiner_launches;
    join_tiles += part.joiner_tiles;
    if (!last) {  // the tile-scan state at the next segment's first tile (this pass has scanned past it)
      HgTileBase next{};
      if (fail(hipMemcpyAsync(&next, d_bases_ + (lo + seg_tiles), sizeof next, hipMemcpyDeviceToHost, stream), "copy") || fail(hipStreamSynchronize(stream), "sync")) return HG_ERR_HIP;
      cs0 = next.cs;
      piece0 = next.L;
    }
  }
  if (fail(hipStreamSynchronize(stream), "sync")) return HG_ERR_HIP;
  out->n_hits = acc;
  out->n_pieces = part.n_pieces;
  out->n_cands = cands;
  out->n_raw_hits = raw;
  out->d_hits = d_acc_hits_;
  out->d_aux = d_acc_aux_;
  out->ms_stream = ms_stream;
  out->ms_total = ms_total;
  out->reruns = reruns;
  out->stream_launches = launches;
  out->joiner_launches = join_launches;
  out->joiner_tiles = join_tiles;
  return HG_OK;
}
