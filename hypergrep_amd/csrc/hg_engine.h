// Device-side scanner object: uploaded database + workspace + the launch sequence of the scan path.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/hypergrep_amd.h"
#include "hg_compile.h"
#include "hg_core.h"
#include "hg_post.h"

enum { HG_CNT_CANDS = 0, HG_CNT_HITS = 1, HG_CNT_WORDS = 8 };

struct HgStreamArgs {
  const uint8_t *text;
  uint64_t nbytes, ntiles;
  HgDbView db;
  const uint32_t *bitmap;
  HgTileSum *sums;
  HgCand *cands;
  uint32_t cand_cap, pad;
  uint32_t *counters;
};

struct HgConfirmArgs {
  const uint8_t *text;
  uint64_t nbytes, ntiles, bs1;
  HgDbView db;
  const HgTileSum *sums;
  const HgTileBase *bases;
  const HgCand *cands;
  HgHit *hits;
  HgHitAux *aux;
  uint32_t cand_cap, hit_cap;
  uint32_t *counters;
};

struct HgScanOutput {
  uint64_t n_hits;       // final (ordered, de-duplicated) hits
  uint64_t n_pieces;     // line pieces in the buffer (the reference's final line_number)
  uint64_t n_cands;      // verified required-literal occurrences
  uint64_t n_raw_hits;   // reports before de-duplication
  const HgHit *d_hits;   // device arrays, valid until the next scan on this scanner
  const HgHitAux *d_aux;
  float ms_stream;       // hg_stream_kernel alone (HIP events on the launch stream)
  float ms_total;        // whole launch sequence
  uint32_t reruns;       // workspace grew and the pass was repeated this many times
};

class HgScanner {
 public:
  static int create(const HgDb *db, int device, HgScanner **out, std::string *err);
  ~HgScanner();
  // d_text: device pointer, 16-byte aligned, readable up to nbytes rounded up to 16.
  int scan(const void *d_text, uint64_t nbytes, int buffer_size, uint64_t line_base, hipStream_t stream, HgScanOutput *out);
  const std::string &last_error() const { return err_; }
  int device() const { return device_; }

 private:
  HgScanner() = default;
  int ensure(uint64_t nbytes);
  int run_once(const uint8_t *text, uint64_t nbytes, uint64_t bs1, uint64_t line_base, hipStream_t stream, HgScanOutput *out, bool *overflow);
  bool fail(hipError_t e, const char *what);

  int device_ = 0;
  int num_cus_ = 256;
  std::string err_;
  // database on device
  HgDbView view_{};
  const HgDb *db_ = nullptr;
  void *d_patterns_ = nullptr, *d_pool_ = nullptr, *d_factors_ = nullptr, *d_windows_ = nullptr, *d_bucket_ = nullptr,
       *d_bitmap_ = nullptr, *d_slow_ = nullptr;
  // workspace
  uint64_t cap_tiles_ = 0;
  uint32_t cand_cap_ = 0, hit_cap_ = 0;
  HgTileSum *d_sums_ = nullptr;
  HgTileBase *d_bases_ = nullptr, *d_block_base_ = nullptr, *d_final_ = nullptr;
  HgTileElem *d_agg_ = nullptr;
  HgCand *d_cands_ = nullptr;
  HgHit *d_hits_raw_ = nullptr, *d_hits_sorted_ = nullptr, *d_hits_out_ = nullptr;
  HgHitAux *d_aux_raw_ = nullptr, *d_aux_sorted_ = nullptr, *d_aux_out_ = nullptr;
  uint64_t *d_key_a_ = nullptr, *d_key_b_ = nullptr;
  uint32_t *d_perm_a_ = nullptr, *d_perm_b_ = nullptr;
  uint8_t *d_keep_ = nullptr;
  uint32_t *d_counters_ = nullptr, *d_selected_ = nullptr;
  void *d_temp_ = nullptr;
  size_t temp_bytes_ = 0;
  uint32_t *h_counters_ = nullptr;  // pinned
  HgTileBase *h_final_ = nullptr;   // pinned
  hipEvent_t ev_[4] = {nullptr, nullptr, nullptr, nullptr};
};
