// Device-side scanner object: uploaded database + workspace + the launch sequence of the scan path.
#pragma once
#include <hip/hip_runtime.h>

#include <memory>
#include <string>

#include "../../include/hypergrep_amd.h"
#include "hg_compile.h"
#include "hg_core.h"
#include "hg_post.h"

// Waves per stream workgroup (one 16 KiB tile per wave at a time); shared by the kernel and the grid sizing.
#ifndef HG_STREAM_WG_WAVES
#define HG_STREAM_WG_WAVES 8
#endif
constexpr uint32_t HG_STREAM_WG_WAVES_DEFAULT = HG_STREAM_WG_WAVES;

// device counters: totals, and the largest per-segment demand seen when a private segment overflowed
constexpr uint32_t HG_CONFIRM_SPLIT = 4;   // verify blocks per candidate segment
constexpr uint32_t HG_DEFER_SHARDS = 64;   // append-only lists of verified candidates that need an automaton run

// A verified literal occurrence whose expression still needs its automaton: pattern may match in the line holding pos.
struct HgDeferred {
  uint64_t pos;       // the window's text offset (locates the tile and the line)
  uint32_t pattern;   // pattern index | offset of the window inside the literal << 24
  uint32_t rank;      // newlines between the tile start and pos
};
// The scanner's small device state is ONE block of words (reset by one launch, read back by one copy):
//   [0, 8) counters, [24, 28) finalize totals {kept, raw, large buckets, -}, [28, 30) {count, overflow flag}
//   of the finalize, [32, 36) tile-scan state (HgTileBase), [64, 128) tile cursors: one per pipeline chunk (a cursor of its own
//   for each of the 64 chunks a pass may have, all zeroed by hg_reset_kernel: no chunk ever waits for, or races with, the
//   reset of a cursor an earlier chunk has used)
enum { HG_ST_FIN_TOTAL = 24, HG_ST_SELECTED = 28, HG_ST_FINAL = 32, HG_ST_WORDS = 36, HG_ST_ZERO_WORDS = 32,
       HG_ST_BLOCK_DONE = 40 };  // (outside the words a pass resets: workgroups of hg_block_small_kernel that have finished)
enum { HG_CNT_CANDS = 0, HG_CNT_HITS = 1, HG_CNT_CAND_NEED = 2, HG_CNT_HIT_NEED = 3, HG_CNT_DEFER_NEED = 4,
       HG_CNT_HITS_WRAPPED = 5,  // the 32-bit hit counter went round (direct appends): the buffer is scanned in segments instead
       HG_CNT_JOIN_TILES = 6,  // tiles taken by the joiner launches of the pass (hg_stream_join_kernel)
       HG_CNT_WORDS = 8,
       HG_CNT_CURSOR0 = 64,     // one tile cursor per pipeline chunk (HgScanner::kMaxChunks = 64 of them)
       HG_CNT_CURSORS = 64,
       HG_ST_ALLOC_WORDS = 128 };
constexpr uint32_t HG_STREAM_GRAB = 2 * HG_STREAM_WG_WAVES_DEFAULT;  // tiles per draw of a stream workgroup: two per wave

// Bucketed finalize (hg_fin_*): buckets of the final ordering and the largest bucket one wave sorts in LDS.
constexpr uint32_t HG_FIN_MAX_BUCKETS = 1u << 22;  // (a pass holds at most 2^28 reports: 64 per bucket then, what one wave orders)
constexpr uint32_t HG_FIN_BUCKET_CAP = 4096;
constexpr uint32_t HG_FIN_MEDIUM_CAP = 512;  // size class of hg_fin_sort_big_kernel that needs little LDS

// Threads of a confirm block: its LDS (the per-lane follow tables of the one-word automata, 128 B per lane) decides how many
// fit on a CU next to the stream pass.
#ifndef HG_CONFIRM_THREADS
#define HG_CONFIRM_THREADS 128
#endif



struct HgStreamArgs {
  const uint8_t *text;
  uint64_t nbytes, tile_begin, tile_end;  // the launch covers tiles [tile_begin, tile_end) of the text
  HgDbView db;
  const uint32_t *filter;  // 1 << filter_log2 window-hash slots
  const HgSlotInfo *ext;   // per slot: window values + neighbour-dword conditions
  HgTileSum *sums;
  HgCand *cands;         // nsegs x cand_seg_cap: one private segment per stream workgroup
  uint32_t *seg_count;   // candidates in each segment
  uint32_t cand_seg_cap, filter_log2;
  uint32_t weights_a, weights_b;
  uint32_t filter_wide;
  uint32_t dense;  // byte-aligned probing (HgDb::dense)
  uint32_t weights_c;  // hash C weights (HgDb::weights_c)
  uint32_t alone;  // no other kernel runs next to this launch (every workgroup slot is its own)
  uint32_t cursor_slot;  // counters[cursor_slot] = tiles of the chunk handed out so far (hg_stream_kernel draws runs of HG_STREAM_GRAB tiles)
  uint32_t *counters;
};

struct HgConfirmArgs {
  const uint8_t *text;
  uint64_t nbytes, tile_begin, tile_end, bs1;
  HgDbView db;
  const HgTileSum *sums;
  const HgTileBase *bases;
  const HgCand *cands;
  const uint32_t *seg_count;
  HgHit *hits;  // compact outputs
  HgHitAux *aux;
  HgHit *tmp_hits;  // gridDim x hit_seg_cap block-private staging
  HgHitAux *tmp_aux;
  HgDeferred *deferred;   // one list per (confirm mode present in the database, shard): list_of_mode[m] * HG_DEFER_SHARDS + shard, each defer_shard_cap entries
  uint32_t *defer_count;  // entries appended to each list, indexed mode * HG_DEFER_SHARDS + shard
  uint32_t list_of_mode[HG_CONFIRM_MODES];
  uint32_t mode_present[HG_CONFIRM_MODES];
  uint32_t *always_count;                 // matches noted per block (reuses the chunk's candidate segment counts)
  uint32_t always_list_cap;               // entries per block of the always-on match list (it reuses `deferred`)
  uint32_t list_spread[HG_CONFIRM_MODES];  // automaton modes: lists per pattern (few patterns: each is spread over several lists)
  uint32_t cand_seg_cap, hit_cap, hit_seg_cap, defer_shard_cap;
  // Candidate segments [0, join_seg0) belong to the chunk's stream launch and hold cand_seg_cap records each; the segments from
  // join_seg0 on belong to the joiner launch and hold join_seg_cap (a quarter: the joiner streams a few per cent of a chunk)
  uint32_t join_seg0, join_seg_cap;
  uint32_t hit_direct;  // 1: a block whose staging segment is full appends to the compact array itself (HitSink)
  // Bucketed emission (the default, bucket_cap != 0): a hit goes straight into the region of the bucket its line starts in
  // (bucket = aux.start >> bucket_shift, bucket_cap records each, bucket_fill = records so far); the finalize kernels
  // (hg_fin_*) then sort every bucket on its own.  bucket_cap == 0: the compact array + library sort (HitSink, flush_hits).
  uint32_t bucket_shift, bucket_cap;
  uint32_t *bucket_fill;
  uint32_t *counters;
  // A pass reports the pieces whose first scanned byte lies in [own_lo, own_hi): the whole buffer normally; one segment of it
  // when the buffer is scanned in several passes (HgScanner::scan_segments).  Buckets count from own_lo.
  uint64_t own_lo, own_hi;
};
constexpr uint32_t HG_BLOCK_SMALL_POOL = 10240;  // ... words of automaton tables a workgroup of that path stages in LDS (40 KiB)
constexpr uint32_t HG_BLOCK_SLICED_MAX = 2047;   // ... blocks up to this many bytes are split over the lanes by start position
constexpr uint32_t HG_BLOCK_SLICED_WORDS = 64;   // ... (bitmap of emitted ends per expression: ends 0 .. HG_BLOCK_SLICED_MAX)
constexpr uint32_t HG_BLOCK_SMALL_MAX = 8192;   // Face A: blocks up to this many bytes take the one-launch path (hg_block_small_kernel)
constexpr uint32_t HG_BLOCK_SMALL_SEG = 1024;   // ... reports per workgroup (32 or 256 expressions) it can hold
constexpr uint32_t HG_HIT_REL_SHIFT = 40;  // raw bucketed records: line_no (< 2^40) | line start inside the bucket (< 2^24) << 40
constexpr uint32_t HG_HIT_SINGLE_BIT = 0x80000000u;  // raw bucketed records: bit 31 of `to` = the expression has HS_FLAG_SINGLEMATCH (`to` < 2^31)

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
  uint32_t stream_launches;  // hg_stream_kernel launches of the (last) pass: one per pipeline chunk
  uint32_t joiner_launches;  // ... and hg_stream_join_kernel launches
  uint64_t joiner_tiles;     // tiles (of 16 KiB) the joiner launches took: bytes the hg_stream_kernel launches did NOT stream
};

// Test / experiment knobs of the engine, read from the environment ONCE, when a scanner is created (never during a scan:
// getenv is not safe against a concurrent setenv, and a scan must not change behaviour half-way).  None is needed in normal
// use.  The limit-lowering ones exist so that tests reach segmented scans / chunk halving on small texts.
struct HgEngineKnobs {
  uint64_t fin_target = 48;        // HG_FIN_TARGET: reports per finalize bucket
  bool no_bucket_finalize = false; // HG_NO_BUCKET_FINALIZE
  uint64_t chunk_tiles = 0;        // HG_CHUNK_TILES: pipeline chunk size (0: default)
  uint32_t max_chunks = 0;         // HG_MAX_CHUNKS (0: default)
  std::string chunk_weights;       // HG_CHUNK_WEIGHTS: relative chunk sizes
  long stream_wgs_per_cu = 0;      // HG_STREAM_WGS_PER_CU (0: default)
  long joiner = -1;                // HG_JOINER (-1: default)
  bool no_early_finalize = false;  // HG_NO_EARLY_FINALIZE
  uint32_t confirm_mode_mask = 0x1F;  // HG_DEBUG_CONFIRM_MODES (profiling builds only: results are incomplete)
  long confirm_blocks_per_cu = 0;  // HG_CONFIRM_BLOCKS_PER_CU (0: default)
  uint64_t hit_limit = 0;          // HG_HIT_LIMIT (0: default 2^28)
  uint64_t cand_limit = 0;         // HG_CAND_LIMIT (0: default 2^30)
  bool verbose = false;            // HG_VERBOSE
  static HgEngineKnobs from_env();
};

class HgScanner {
 public:
  // The scanner shares ownership of the (immutable) database: launch parameters are read from it on every scan.
  static int create(std::shared_ptr<const HgDb> db, int device, HgScanner **out, std::string *err);
  const std::shared_ptr<const HgDb> &database() const { return db_; }
  ~HgScanner();
  // d_text: device pointer, 16-byte aligned, readable up to nbytes rounded up to 16.
  int scan(const void *d_text, uint64_t nbytes, int buffer_size, uint64_t line_base, hipStream_t stream, HgScanOutput *out);
  // Block mode (hs_scan): the whole buffer is one scan unit; hits carry line_no 0 and `to` relative to the buffer start.
  int scan_block(const void *d_text, uint64_t nbytes, hipStream_t stream, HgScanOutput *out);
  // Block mode for short blocks held in PINNED host memory (readable up to nbytes rounded up to 16): one launch, raw
  // reports {0, id, to | HG_HIT_SINGLE_BIT} in h_out (a segment of HG_BLOCK_SMALL_SEG records per workgroup), their
  // number per segment in h_counts; the caller synchronises the stream and applies the report rules.  Returns the number
  // of segments, 0 if the block or the pattern set is too large for this path.
  // *h_flag (pinned) receives `seq` when every segment is written: the caller may poll it instead of synchronising.
  uint32_t launch_block_small(const uint8_t *h_text, uint32_t nbytes, hipStream_t stream, HgHit *h_out, uint32_t *h_counts, uint32_t *h_flag, uint32_t seq);
  const std::string &last_error() const { return err_; }
  int device() const { return device_; }

 private:
  HgScanner() = default;
  int ensure(uint64_t nbytes);
  int alloc_cands(uint64_t n);
  int alloc_hits(uint64_t n);
  int scan_impl(const void *d_text, uint64_t nbytes, int buffer_size, uint64_t line_base, bool block_mode, hipStream_t stream, HgScanOutput *out);
  // One pass over the tiles [tile_lo, tile_hi) of the text: the whole buffer, or a segment of it (scan_segments).  The tile
  // scan starts from (cs0, piece0): the start of the line that contains the range's first byte and that line's piece index;
  // only pieces whose first scanned byte lies in [own_lo, own_hi) are reported.
  struct PassRange {
    uint64_t tile_lo, tile_hi, cs0, piece0, own_lo, own_hi;
    bool last;  // the range ends with the text: the pass leaves the final piece count
  };
  int scan_segments(const uint8_t *text, uint64_t nbytes, uint64_t bs1, uint64_t line_base, hipStream_t stream, HgScanOutput *out, uint32_t nsegments, bool *too_many);
  int run_once(const uint8_t *text, uint64_t nbytes, uint64_t bs1, uint64_t line_base, const PassRange &range, bool block_mode, hipStream_t stream, HgScanOutput *out,
               bool *overflow);
  bool fail(hipError_t e, const char *what);

  HgEngineKnobs knobs_;
  int device_ = 0;
  int num_cus_ = 256;
  int stream_wgs_per_cu_ = 0;
  std::string err_;
  // database on device
  HgDbView view_{};
  std::shared_ptr<const HgDb> db_;
  void *d_disc_ = nullptr, *d_bucket2_ = nullptr, *d_windows2_ = nullptr, *d_groups_ = nullptr, *d_wtab_ = nullptr;
  void *d_patterns_ = nullptr, *d_pool_ = nullptr, *d_factors_ = nullptr, *d_windows_ = nullptr, *d_bucket_ = nullptr,
       *d_filter_ = nullptr, *d_ext_ = nullptr, *d_slow_ = nullptr;
  // workspace
  uint64_t cap_tiles_ = 0;
  uint32_t cand_cap_ = 0, hit_cap_ = 0;
  uint64_t chunk_limit_tiles_ = 0;   // pipeline chunks no larger than this (0: the default 8 GiB): halved when a chunk's candidates would not fit
  uint32_t defer_spread_boost_ = 1;  // automaton confirm modes: lists per expression, raised when one expression's occurrences overflow its list
  bool hit_direct_ = false;  // the hits are too unevenly spread for equal per-block segments (HitSink::direct)
  HgTileSum *d_sums_ = nullptr;
  HgTileBase *d_bases_ = nullptr, *d_block_base_ = nullptr, *d_final_ = nullptr;
  HgTileElem *d_agg_ = nullptr;
  HgCand *d_cands_ = nullptr;
  HgDeferred *d_deferred_ = nullptr;
  uint32_t *d_defer_count_ = nullptr;
  HgHit *d_hits_raw_ = nullptr, *d_hits_out_ = nullptr;
  HgHitAux *d_aux_raw_ = nullptr, *d_aux_out_ = nullptr;
  HgHit *d_acc_hits_ = nullptr;  // segmented scans: the segments' ordered hits, one after the other
  HgHitAux *d_acc_aux_ = nullptr;
  uint64_t acc_cap_ = 0;
  bool side_bound_ = false;  // the last piped pass: the side passes, not the stream launches, set the pace (no joiner, no finalize on the side stream)
  uint64_t *d_key_a_ = nullptr, *d_key_b_ = nullptr;
  uint32_t *d_perm_a_ = nullptr, *d_perm_b_ = nullptr;
  uint8_t *d_keep_ = nullptr;
  uint32_t *d_counters_ = nullptr, *d_selected_ = nullptr, *d_seg_count_ = nullptr;
  uint32_t max_segs_ = 0;
  uint32_t *d_pflags_ = nullptr;
  void *d_temp_ = nullptr;
  size_t temp_bytes_ = 0;
  uint32_t *h_counters_ = nullptr;  // pinned
  HgTileBase *h_final_ = nullptr;   // pinned
  hipEvent_t ev_[4] = {nullptr, nullptr, nullptr, nullptr};
  // chunked pipeline: the stream pass of chunk c+1 overlaps the verify / confirm passes of chunk c
  static constexpr int kMaxChunks = 64;
  hipStream_t side_stream_ = nullptr;
  hipEvent_t ev_tile_done_ = nullptr;  // the last chunk's tile scan is done (the early finalize starts behind it)
  hipEvent_t ev_fin_early_ = nullptr;  // the finalize of the earlier chunks' buckets (beside the last chunk's side passes) is done
  hipEvent_t ev_k1_begin_[kMaxChunks] = {}, ev_k1_end_[kMaxChunks] = {}, ev_side_done_[kMaxChunks] = {};
  // bucketed emission + finalize (hg_fin_*): records per bucket / kept counts -> output positions / {kept, raw} totals;
  // fin_fallback_: a bucket outgrew what one block sorts, the compact array + library sort is used from then on
  uint32_t *d_fin_fill_ = nullptr, *d_fin_kept_ = nullptr, *d_fin_total_ = nullptr, *d_fin_big_ = nullptr;
  bool fin_fallback_ = false;
  uint32_t fin_alloc_ = 0;  // buckets the arrays above hold (they grow with the bucket count a pass asks for)
  uint32_t fin_epoch_ = 0;  // hg_fin_scan_kernel: a fresh value per launch marks the blocks' partial sums as this launch's
  uint64_t fin_expect_hits_ = 0;  // raw hits of the last pass: the next one picks its bucket count for ~24 records a bucket
  void *d_huge_claim_ = nullptr;      // huge automata: (piece start, expression) pairs already run (hg_confirm_huge_kernel), 8-byte slots
  uint64_t huge_claim_slots_ = 0;
  uint32_t *d_seg_count2_ = nullptr;  // second set for double buffering
  HgCand *d_cands2_ = nullptr;
};
