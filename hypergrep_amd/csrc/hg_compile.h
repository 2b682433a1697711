// Host-side pattern compiler: the product's counterpart of hs_compile_multi
// (reference call site hypergrep/lib/c/hyperscanner.c:136, mode HS_MODE_BLOCK, platform NULL).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "hg_core.h"
#include "hg_db.h"

struct HgDb {
  std::vector<HgPattern> patterns;
  std::vector<uint32_t> pool;        // all automaton tables
  std::vector<HgFactor> factors;     // required literals of tier-0 patterns
  uint32_t nreal_factors = 0;
  std::vector<HgWindow> windows;     // grouped by bucket
  std::vector<uint32_t> bucket_off;  // (1 << HG_HASH_BITS) + 1 offsets into windows
  std::vector<uint16_t> disc;         // per hash-C group: discriminator dword (hg_db.h)
  std::vector<uint32_t> bucket_off2;  // (1 << HG_HASH_BITS) + 1 offsets into windows2, indexed by hg_disc_bucket
  std::vector<HgWindow> windows2;     // the same windows ordered by discriminated bucket (GPU verify pass)
  std::vector<HgWinBucket> wtab;      // direct window table (hg_db.h): wtab_mask + 1 buckets
  uint32_t wtab_mask = 0;
  uint32_t shared_windows = 0;        // window values that several literals share (they take the discriminated buckets)
  uint32_t wtab_first = 0;            // at most one window value in twenty is shared: the verify pass asks the table first (config 5:
                                      // 190 of 16 384; 14.9 against 16.4 ms per 32 GiB).  Else it starts at the discriminated buckets, as
                                      // it did before the table existed: for a shared window the table is one more fetch in front of that
                                      // chain (config 3, 150 of 1400 values, the class expressions' stems: 315 against 225 us per 8 GiB)
  std::vector<uint32_t> filter;      // 1 << filter_log2 slots holding hash C of the owning window (staged in LDS by the stream kernel)
  uint32_t filter_log2 = HG_FILTER_MIN_LOG2;
  uint32_t filter_wide = 0;          // 1: two 16-bit fingerprints per slot, no neighbour conditions (large pattern sets)
  uint32_t window_bytes = HG_WINDOW_BYTES;  // bytes of a window: 4, or 3 with byte-aligned probing when 3-byte literals are too many to enumerate
  uint32_t window_mask = HG_WINDOW_MASK;
  uint32_t weights_c = HG_HASH_WEIGHTS;     // hash C weights (the top weight is zero for 3-byte windows: that byte is not part of the window)
  uint32_t dense = 0;                // byte-aligned probing (sets with short literals): the stream pass probes a window every `dense` bytes
                                     // (1: one window per literal; 2: literals of >= 5 bytes, a window on both residues mod 2); 0: dword-aligned
  uint32_t weights_a = HG_SLOT_WEIGHT_CHOICES[0][0], weights_b = HG_SLOT_WEIGHT_CHOICES[0][1];
  std::vector<HgSlotInfo> ext;       // per filter slot: the window values in it and their neighbour-dword conditions (second-level check)
  std::vector<uint32_t> slow;        // indices of tier-1 (always-on) patterns, the nslow_fast bounded ones with <= 2 state words first
  uint32_t nslow_fast = 0;
  std::vector<HgSlowGroup> groups;   // always-on expressions packed into shared state words (hg_db.h); their members are the
  uint32_t nslow_grouped = 0;        // first nslow_grouped entries of `slow`, group by group
  uint32_t fold_mask = 0;            // 0x20202020 when any tier-0 pattern is case-insensitive
  uint32_t max_nw = 1;               // state words of the largest automaton with dense tables (<= HG_MAX_W)
  uint32_t nhuge = 0;                // expressions with sparse tables (more than HG_MAX_NODES nodes: hg_db.h HgHugeHeader)
  uint32_t huge_max_nw = 0;          // ... and the state words of the largest of them (sizes the LDS of the huge routines)
  uint32_t huge_stage_words = 0;     // LDS words the huge routines get for a staged copy of ONE expression's per-byte tables: the largest
                                     // need among the huge expressions that fits HG_HUGE_STAGE_MAX (an expression that needs more reads L2)
  uint32_t nslow_huge = 0;           // huge always-on expressions: the LAST nslow_huge entries of `slow`
  uint32_t max_id = 0;               // largest report id (sizes the sort key)
  uint32_t n_confirm_mode[HG_CONFIRM_MODES] = {};  // tier-0 patterns by confirm routine (hg_confirm_mode)
  std::vector<std::string> exprs;
  bool tuned = false;
};

// Re-select the literal windows using byte statistics of a text sample (any part of the text to be scanned) and rebuild the
// filter tables; never changes results, only how often the slow stages run.  The database itself is immutable once
// compiled (scanners read it live): the tuned tables are built into a COPY, returned in *out only on success (0).
int hgc_tune(const HgDb *db, const uint8_t *sample, size_t nbytes, HgDb **out, std::string *err);

// Returns 0 on success.  On failure returns non-zero, sets *err and *bad_index (expression index or -1).

int hgc_compile(const char *const *exprs, const unsigned *flags, const unsigned *ids, unsigned n, HgDb **out,
               std::string *err, int *bad_index);
void hgc_free(HgDb *db);
