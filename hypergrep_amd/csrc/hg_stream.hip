// gfx950 stream pass of the line-scan path (split from hg_kernels.hip in round 3: thirty-odd instantiations of one kernel that
// every experiment on the side passes used to rebuild).  Replaces the reference's per-line hot loop
// (hypergrep/lib/c/hyperscanner.c:198-226: gzgets -> strlen -> hs_scan per line) for the part that touches every byte:
//
//   hg_stream_kernel      one pass over the text in HBM: 16 B per lane coalesced loads, per-dword window
//                         fingerprint (v_dot4_u32_u8) probed in the single-probe LDS filter, exact newline counts
//                         per 16 KiB wave tile; chunks with a fingerprint match are queued in LDS and examined
//                         64 at a time (neighbour conditions), survivors go to the workgroup's candidate segment
//   hg_stream_join_kernel the same code under its own name: the launch that joins a chunk behind the previous chunk's side passes
//
// Byte/integer work, HBM-bound: no MFMA anywhere.  Wave64 only.
#include <hip/hip_runtime.h>
#include <cstdlib>

#include "hg_core.h"
#include "hg_engine.h"

namespace {

constexpr int WG_WAVES = HG_STREAM_WG_WAVES;
constexpr int WG_THREADS = WG_WAVES * 64;
constexpr int ITERS = HG_TILE_BYTES / 1024;  // 1 KiB per wave-iteration

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, uint32_t lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t u = __shfl_up(v, o, 64);
    if (lane >= static_cast<uint32_t>(o)) v += u;
  }
  return v;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// Stream pass.  One wave owns one 16 KiB tile at a time; every workgroup streams its own consecutive range of tiles.
//
// Per 16 bytes of text a lane spends: 1 coalesced 16 B load (non-temporal), 4 x (4 ops: exact newline count) and
// 4 x (fold, 2 x v_dot4_u32_u8, 1 LDS read, 2 SDWA ops) for the window filter — 16 x with byte-aligned probing.  A chunk
// whose filter matched is queued in LDS; ranks, second level and the append to the candidate segment run out of line
// for 64 queued chunks at a time, so the steady state is pure streaming.
namespace {

// "not a newline" bits: bit 7 of each byte is CLEAR iff that byte is '\n' (exact, no carries between bytes)
__device__ __forceinline__ uint32_t not_newline_bits(uint32_t w) {
  uint32_t b = ((w & 0x7f7f7f7fu) ^ 0x0a0a0a0au) + 0x7f7f7f7fu;  // bit 7 set iff the low 7 bits differ from 0x0a
  return b | w | 0x7f7f7f7fu;                                     // ... or the byte's own bit 7 is set
}

// LDS is addressed through explicit address-space-3 pointers so that the out-of-line drain routine also gets ds_* instructions.
using lds_u32 = __attribute__((address_space(3))) uint32_t;

// entries per wave: an iteration adds at most 64, and the queue is drained as soon as it holds a full batch of 64
constexpr uint32_t queue_cap(int) { return 128u; }
// dwords per entry: {chunk | rank << 10, tile} and, while LDS has room (filters up to 16 KiB, three workgroups per CU), also
// {left, right, the chunk's four dwords} so that the drain does not read the text again (measured: re-reading costs ~15 % extra
// HBM fetches on the round-1 workload, the streamed tiles have left the L2 by then)
// (byte-aligned probing re-reads the text: its drain wants up to 28 bytes around the chunk)
constexpr bool queue_stash(int log2, bool dense) { return log2 <= 12 && !dense; }
constexpr uint32_t queue_entry_dw(int log2, bool dense) { return queue_stash(log2, dense) ? 8u : 2u; }

template <int LOG2, bool WIDE>
struct Probe {
  static constexpr uint32_t BYTE_MASK = ((1u << LOG2) - 1u) << 2;
  __device__ __forceinline__ static uint32_t at(const lds_u32 *filter, uint32_t byte_off) {
    return *reinterpret_cast<const lds_u32 *>(reinterpret_cast<const __attribute__((address_space(3))) uint8_t *>(filter) + byte_off);
  }
  // First level for the lane's four dwords: a slot matches if it holds the window's hash C.
  // ANY_ONLY: non-zero iff any of the four windows matched (hot path); else per-window, per-slot match bits.
  // FOLD: the set folds the text before hashing it (case-insensitive literals stored folded); false: nothing is folded (no such
  // literal, or their windows are stored in every case variant): one instruction per dword less in the hot loop
  template <bool ANY_ONLY, bool FOLD = true>
  __device__ __forceinline__ static uint32_t probe4(const lds_u32 *filter, uint32_t fold, uint32_t wa, uint32_t wb, uint4 v) {
    const uint32_t f0 = FOLD ? v.x | fold : v.x, f1 = FOLD ? v.y | fold : v.y, f2 = FOLD ? v.z | fold : v.z, f3 = FOLD ? v.w | fold : v.w;
    // the hashes of the four windows first (independent v_dot4), then the eight LDS reads
    const uint32_t a0 = hg_dot4(f0, wa), a1 = hg_dot4(f1, wa), a2 = hg_dot4(f2, wa), a3 = hg_dot4(f3, wa);
    const uint32_t b0 = hg_dot4(f0, wb), b1 = hg_dot4(f1, wb), b2 = hg_dot4(f2, wb), b3 = hg_dot4(f3, wb);
    const uint32_t c0 = hg_dot4(f0, HG_HASH_WEIGHTS), c1 = hg_dot4(f1, HG_HASH_WEIGHTS);
    const uint32_t c2 = hg_dot4(f2, HG_HASH_WEIGHTS), c3 = hg_dot4(f3, HG_HASH_WEIGHTS);
    if (WIDE) {  // slots mix both sums (hg_slot_wide); two 16-bit fingerprints per slot; no second level in wide mode
      auto wide = [&](uint32_t x, uint32_t y) { return at(filter, hg_slot_wide(x, y, BYTE_MASK)); };
      const uint32_t ta0 = wide(a0, b0), tb0 = wide(b0, a0), ta1 = wide(a1, b1), tb1 = wide(b1, a1);
      const uint32_t ta2 = wide(a2, b2), tb2 = wide(b2, a2), ta3 = wide(a3, b3), tb3 = wide(b3, a3);
      auto m = [](uint32_t t, uint32_t c) { return static_cast<uint16_t>(t) == static_cast<uint16_t>(c) || static_cast<uint16_t>(t >> 16) == static_cast<uint16_t>(c); };
      const bool m0 = m(ta0, c0) || m(tb0, c0), m1 = m(ta1, c1) || m(tb1, c1), m2 = m(ta2, c2) || m(tb2, c2), m3 = m(ta3, c3) || m(tb3, c3);
      if (ANY_ONLY) return (m0 || m1 || m2 || m3) ? 1u : 0u;
      return (m0 ? 1u : 0u) | (m1 ? 2u : 0u) | (m2 ? 4u : 0u) | (m3 ? 8u : 0u);
    }
    // single probe: the window's one slot must agree with hash C on every fingerprint bit the slot cares about
    const uint32_t t0 = at(filter, a0 & BYTE_MASK), t1 = at(filter, a1 & BYTE_MASK), t2 = at(filter, a2 & BYTE_MASK), t3 = at(filter, a3 & BYTE_MASK);
    const bool m0 = hg_slot_match(t0, c0), m1 = hg_slot_match(t1, c1), m2 = hg_slot_match(t2, c2), m3 = hg_slot_match(t3, c3);
    if (ANY_ONLY) return (m0 || m1 || m2 || m3) ? 1u : 0u;
    return (m0 ? 1u : 0u) | (m1 ? 2u : 0u) | (m2 ? 4u : 0u) | (m3 ? 8u : 0u);  // bit k: window k matched
  }
};

// Byte-aligned probing (pattern sets with required literals shorter than HG_FAST_MIN_FACTOR, db.dense): every byte of the
// lane's 16 is the start of a window, `nxt` = the dword after the chunk.  One window per literal instead of one per
// residue; four times the probes of the dword-aligned filter.
template <int LOG2, int STEP>  // STEP: a window starts at every byte (1) or every second byte (2: sets whose literals all have >= 5 bytes)
struct ProbeBytes {
  static constexpr uint32_t BYTE_MASK = ((1u << LOG2) - 1u) << 2;
  // ANY_ONLY: non-zero iff any of the 16 windows matched; else bit k = the window that starts at byte k matched
  template <bool ANY_ONLY>
  __device__ __forceinline__ static uint32_t probe16(const lds_u32 *filter, uint32_t fold, uint32_t wa, uint32_t wc, uint4 v, uint32_t nxt) {
    const uint32_t d[5] = {v.x | fold, v.y | fold, v.z | fold, v.w | fold, nxt | fold};
    uint32_t bits = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      uint32_t w[4], t[4];
      w[0] = d[j];
#pragma unroll
      for (int k = STEP; k < 4; k += STEP) w[k] = __builtin_amdgcn_alignbyte(d[j + 1], d[j], k);
#pragma unroll
      for (int k = 0; k < 4; k += STEP) t[k] = Probe<LOG2, false>::at(filter, hg_dot4(w[k], wa) & BYTE_MASK);
#pragma unroll
      for (int k = 0; k < 4; k += STEP) {
        const bool m = hg_slot_match(t[k], hg_dot4(w[k], wc));  // (3-byte windows: both weight vectors end in zero)
        if (ANY_ONLY) bits |= m ? 1u : 0u;
        else bits |= m ? (1u << (j * 4 + k)) : 0u;
      }
    }
    return bits;
  }
};

// Dword at byte offset `at` (a multiple of 4) of the text, bytes past the end of the text zeroed.
__device__ __forceinline__ uint32_t load_dword_checked(const uint4 *__restrict__ text16, uint64_t nbytes, uint64_t at) {
  if (at >= nbytes) return 0u;
  uint32_t v = reinterpret_cast<const uint32_t *>(text16)[at >> 2];
  const uint64_t rest = nbytes - at;
  if (rest < 4) v &= (1u << (rest * 8)) - 1u;
  return v;
}

// Everything the out-of-line drain routine needs besides the per-tile state (wave-uniform; lives in SGPRs).
struct StreamCtx {
  const uint4 *text16;
  uint64_t nbytes;
  const lds_u32 *filter;
  const HgSlotInfo *ext;       // the slots' window values and neighbour conditions (HBM, L2-resident)
  lds_u32 *queue;              // this wave's queue, one entry per 16-byte chunk whose first level matched: {chunk inside the tile | newlines of the
                               // tile before it << 10, tile, dword left of the chunk, dword right of it, the chunk} — the drain never re-reads the text
  lds_u32 *cand_count;         // the workgroup's candidate counter
  HgCand *seg;                 // the workgroup's private candidate segment
  uint32_t seg_cap, fold, wa, wb;  // wb: slot weights B in wide mode; with byte-aligned probing the hash C weights (their top byte is zero
                                   // for 3-byte windows).  The struct travels by value to drain_batch: any larger and it goes through scratch
};

#if defined(__HIP_DEVICE_COMPILE__)  // (LDS pointers are 4 bytes on the device)
static_assert(sizeof(StreamCtx) <= 64, "StreamCtx is passed by value to the out-of-line drain routine: beyond 16 dwords it travels through scratch (measured: 1.1 GB of scratch writes per 8 GiB launch, stream pass 20 % slower)");
#endif

// Chunk `g` of the text with the bytes past the end of the text zeroed.
__device__ __forceinline__ uint4 load_chunk_checked(const uint4 *__restrict__ text16, uint64_t nbytes, uint64_t g) {
  uint4 v = make_uint4(0, 0, 0, 0);
  if ((g << 4) < nbytes) {
    v = text16[g];
    const uint64_t byte0 = g << 4;
    if (byte0 + 16 > nbytes) {
      const uint32_t valid = static_cast<uint32_t>(nbytes - byte0);
      uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t lo = k * 4u;
        if (valid <= lo) w[k] = 0;
        else if (valid < lo + 4) w[k] &= (1u << ((valid - lo) * 8)) - 1u;
      }
      v = make_uint4(w[0], w[1], w[2], w[3]);
    }
  }
  return v;
}

typedef uint32_t __attribute__((aligned(1))) hg_u32_unaligned;
// The window that starts at byte k (0..15) of the chunk `cur`, `nxt` = the dword after the chunk.
__device__ __forceinline__ uint32_t dense_window(uint4 cur, uint32_t nxt, uint32_t k) {
  const uint32_t j = k >> 2;
  const uint32_t lo = j == 0 ? cur.x : (j == 1 ? cur.y : (j == 2 ? cur.z : cur.w));
  const uint32_t hi = j == 0 ? cur.y : (j == 1 ? cur.z : (j == 2 ? cur.w : nxt));
  return static_cast<uint32_t>(((static_cast<uint64_t>(hi) << 32) | lo) >> (8u * (k & 3u)));
}

// Drain of one batch of the wave's queue (the rare half of the stream pass, out of line).  The hot loop only records
// WHICH 16-byte chunks had a first-level match (tile, chunk inside the tile, newlines of the tile before the chunk); here
// one lane takes one such chunk, re-reads it and its two neighbouring dwords (L2-resident: the tile was streamed moments
// ago), repeats the first level per window, applies the second level (the slot's neighbour conditions) and appends the
// survivors to the workgroup's candidate segment.  The queue outlives tiles, so batches are full (64 entries) except the
// last one of the kernel.
template <int LOG2, bool WIDE, int DENSE>  // DENSE: 0 dword-aligned windows, else the byte step of byte-aligned probing
__device__ __noinline__ void drain_batch(const StreamCtx cx, uint32_t first, uint32_t n, uint32_t lane) {
  const bool active = lane < n;
  uint32_t hits = 0, rank = 0;
  uint64_t g = 0;
  uint4 cur = make_uint4(0, 0, 0, 0);
  if constexpr (DENSE) {
    // byte-aligned windows: the chunk and the dword after it are read again (L2), the 16 windows probed one by one, the
    // neighbour conditions taken from the text itself
    uint32_t nxt = 0;
    if (active) {
      const lds_u32 *e = cx.queue + (first + lane) * queue_entry_dw(LOG2, DENSE);
      const uint32_t e_lo = e[0], e_hi = e[1];
      g = static_cast<uint64_t>(e_hi) * (HG_TILE_BYTES / 16) + (e_lo & 1023u);
      rank = e_lo >> 10;
      cur = load_chunk_checked(cx.text16, cx.nbytes, g);
      nxt = load_dword_checked(cx.text16, cx.nbytes, (g + 1) << 4);
      uint32_t l1 = ProbeBytes<LOG2, DENSE ? DENSE : 1>::template probe16<false>(cx.filter, cx.fold, cx.wa, cx.wb, cur, nxt);
      // only windows that START inside the text (the bytes after a short literal at the very end read as zeros; a window
      // past the end is nothing, and the neighbour reads below rely on pos < nbytes)
      if ((g << 4) + 16 > cx.nbytes) l1 &= (g << 4) < cx.nbytes ? (1u << static_cast<uint32_t>(cx.nbytes - (g << 4))) - 1u : 0u;
      const uint32_t wbytes = (cx.wb >> 24) ? 4u : 3u, wmask = (cx.wb >> 24) ? 0xFFFFFFFFu : 0x00FFFFFFu;
      constexpr uint32_t BYTE_MASK = ((1u << LOG2) - 1u) << 2;
      const uint8_t *text = reinterpret_cast<const uint8_t *>(cx.text16);
      for (uint32_t todo = l1; todo; todo &= todo - 1) {
        const uint32_t k = __ffs(todo) - 1;
        const uint64_t pos = (g << 4) + k;
        const uint32_t wk = dense_window(cur, nxt, k);
        // the four bytes before / after the window, zero outside the text
        uint32_t prev = 0, next = 0;
        if (pos >= 4 && pos + 8 <= cx.nbytes) {
          prev = *reinterpret_cast<const hg_u32_unaligned *>(text + pos - 4);
          next = *reinterpret_cast<const hg_u32_unaligned *>(text + pos + wbytes);
        } else {
          for (uint32_t b = 0; b < 4; b++) {  // (pos < nbytes, so pos + b - 4 is inside the text)
            if (pos + b >= 4) prev |= static_cast<uint32_t>(text[pos + b - 4]) << (8 * b);
            if (pos + wbytes + b < cx.nbytes) next |= static_cast<uint32_t>(text[pos + wbytes + b]) << (8 * b);
          }
        }
        const uint32_t f = (wk | cx.fold) & wmask;
        const HgSlotInfo info = cx.ext[(hg_dot4(f, cx.wa) & BYTE_MASK) >> 2];
        if (hg_slot_pass(info, f, prev | cx.fold, next | cx.fold, 0xFFFFFFFFu, 0xFFFFFFFFu)) hits |= 1u << k;
      }
    }
    const uint32_t cnt = __popc(hits);
    if (!__builtin_amdgcn_ballot_w64(cnt != 0)) return;
    const uint32_t incl = wave_inclusive_scan(cnt, lane);
    const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
    uint32_t base = 0;
    if (lane == 0) base = __hip_atomic_fetch_add(cx.cand_count, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    base = __builtin_amdgcn_readfirstlane(base);
    uint32_t s = base + incl - cnt;
    // newline bytes of the chunk: bit b set iff byte b is '\n'
    uint32_t nl = 0;
    {
      const uint32_t words[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint32_t m = ~not_newline_bits(words[j]);  // bit 7 of each newline byte
        nl |= (((m >> 7) & 1u) | ((m >> 14) & 2u) | ((m >> 21) & 4u) | ((m >> 28) & 8u)) << (4 * j);
      }
    }
    for (uint32_t todo = hits; todo; todo &= todo - 1, s++) {
      const uint32_t k = __ffs(todo) - 1;
      if (s < cx.seg_cap) cx.seg[s] = HgCand{(g << 4) + k, dense_window(cur, nxt, k), rank + __popc(nl & ((1u << k) - 1u))};
    }
    return;
  }
  if (active) {
    const lds_u32 *e = cx.queue + (first + lane) * queue_entry_dw(LOG2, DENSE);
    const uint32_t e_lo = e[0], e_hi = e[1];
    g = static_cast<uint64_t>(e_hi) * (HG_TILE_BYTES / 16) + (e_lo & 1023u);
    rank = e_lo >> 10;
    uint32_t left = 0, right = 0;
    bool have_left = (e_lo & 63u) != 0, have_right = (e_lo & 63u) != 63u;  // stashed neighbours come from the adjacent lanes of the 1 KiB row
    if (queue_stash(LOG2, DENSE)) {
      left = e[2];
      right = e[3];
      cur = make_uint4(e[4], e[5], e[6], e[7]);
    } else {
      cur = load_chunk_checked(cx.text16, cx.nbytes, g);
      if (!WIDE) {  // the real neighbours (zero outside the text: a literal cannot extend past either end)
        const uint32_t *text32 = reinterpret_cast<const uint32_t *>(cx.text16);
        left = g ? text32[g * 4 - 1] : 0u;
        if (((g + 1) << 4) < cx.nbytes) {
          right = text32[g * 4 + 4];
          const uint64_t rest = cx.nbytes - ((g + 1) << 4);
          if (rest < 4) right &= (1u << (rest * 8)) - 1u;
        }
        have_left = have_right = true;
      }
    }
    uint32_t l1 = Probe<LOG2, WIDE>::template probe4<false>(cx.filter, cx.fold, cx.wa, cx.wb, cur);
    // only windows that start inside the text (see stream_tile: chunks past the end are not queued at all)
    if ((g << 4) + 16 > cx.nbytes) l1 &= (g << 4) < cx.nbytes ? (1u << static_cast<uint32_t>((cx.nbytes - (g << 4) + 3) >> 2)) - 1u : 0u;
    if (WIDE) {
      hits = l1;
    } else if (l1) {
      // a missing neighbour (first / last lane of the row) skips the condition on that side
      constexpr uint32_t BYTE_MASK = ((1u << LOG2) - 1u) << 2;
      // one window per lane and trip (a chunk rarely has two first-level matches): the lanes' matches sit at different k,
      // and a loop over k would pay one round trip for the conditions per k
      for (uint32_t todo = l1 & 15u; todo; todo &= todo - 1) {
        const uint32_t k = __ffs(todo) - 1;
        const uint32_t wk = k == 0 ? cur.x : (k == 1 ? cur.y : (k == 2 ? cur.z : cur.w));
        const uint32_t wp = k == 0 ? left : (k == 1 ? cur.x : (k == 2 ? cur.y : cur.z));
        const uint32_t wn = k == 0 ? cur.y : (k == 1 ? cur.z : (k == 2 ? cur.w : right));
        const uint32_t f = wk | cx.fold;
        // read from HBM / L2: rare, and keeping the table out of LDS leaves room for more resident waves
        const HgSlotInfo info = cx.ext[(hg_dot4(f, cx.wa) & BYTE_MASK) >> 2];
        const uint32_t pm_keep = (k == 0 && !have_left) ? 0u : 0xFFFFFFFFu;
        const uint32_t nm_keep = (k == 3 && !have_right) ? (HG_WINDOW_BYTES == 4 ? 0u : 0xFFu) : 0xFFFFFFFFu;
        const uint32_t prev = wp | cx.fold;
        const uint32_t next = (HG_WINDOW_BYTES == 4 ? wn : ((wk >> 24) | (wn << 8))) | cx.fold;
        if (hg_slot_pass(info, f & HG_WINDOW_MASK, prev, next, pm_keep, nm_keep)) hits |= 1u << k;
      }
    }
  }
  if (!__builtin_amdgcn_ballot_w64(hits != 0)) return;
  // append: slots from ballots (no wave scan), one LDS atomic per batch
  const uint32_t words[4] = {cur.x, cur.y, cur.z, cur.w};
  uint32_t total = 0, slot[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint64_t mk = __builtin_amdgcn_ballot_w64(((hits >> k) & 1u) != 0);
    slot[k] = total + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mk >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mk), 0u));
    total += __popcll(mk);
  }
  uint32_t base = 0;
  if (lane == 0) base = __hip_atomic_fetch_add(cx.cand_count, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  base = __builtin_amdgcn_readfirstlane(base);
  const uint64_t pos = g << 4;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if ((hits >> k) & 1u) {
      const uint32_t s = base + slot[k];
      if (s < cx.seg_cap) cx.seg[s] = HgCand{pos + 4u * k, words[k], rank};
    }
    rank += __popc(~not_newline_bits(words[k]));
  }
}

// One tile.  FULL: the tile lies entirely inside the text (no bounds checks on the hot path).
// qn: entries in the wave's queue (wave-uniform, carried from tile to tile).
template <int LOG2, bool WIDE, int DENSE, bool FULL, int DEPTH, bool FOLD>
__device__ __forceinline__ void stream_tile(const StreamCtx &cx, uint64_t tile, HgTileSum *__restrict__ sums, uint32_t lane, uint32_t &qn) {
  const uint4 *__restrict__ text16 = cx.text16;
  const uint64_t nbytes = cx.nbytes;
  const uint64_t chunk0 = tile * (HG_TILE_BYTES / 16) + lane;

  auto load_chunk = [&](int it) -> uint4 {
    const uint64_t g = chunk0 + static_cast<uint64_t>(it) * 64u;
    if (FULL) {
      // streaming (non-temporal) cache policy: the text is read once and must not push the filter's second level, the
      // tile summaries and the side passes' working set out of the L2 (measured: 5.0 -> 5.4 TB/s in the pipeline).
      // Not for wide filters: their drain reads every queued chunk AGAIN a few microseconds later (no room in LDS to carry
      // it in the queue), and with the default policy most of those reads hit the L2 (config 5: 14.98 -> 14.65 ms per 32 GiB)
      if (WIDE) return text16[g];
#if defined(HG_NO_NT_LOADS)
      return text16[g];
#else
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
      const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(text16) + g);
      return make_uint4(v.x, v.y, v.z, v.w);
#endif
    }
    return load_chunk_checked(text16, nbytes, g);
  };

  uint32_t seen = 0;                           // wave-uniform: newlines of the tile in the iterations done so far
  uint32_t first_it = HG_NONE32, last_it = 0;  // wave-uniform: iterations holding the first / last newline
  uint32_t first_lane = 0, last_lane = 0;

  // byte-aligned probing: the dword after the lane's chunk is the next lane's first; the row's last lane reads it
  auto load_after = [&](int it) -> uint32_t {
    if (!DENSE || lane != 63u) return 0u;
    return load_dword_checked(text16, nbytes, (chunk0 + static_cast<uint64_t>(it) * 64u + 1u) << 4);
  };

  auto body = [&](int it, uint4 cur, uint32_t after) {
    // exact newline count of this lane's 16 bytes: 128 - popcount of the "not a newline" bits
    uint32_t notnl = __popc(not_newline_bits(cur.x));
    notnl += __popc(not_newline_bits(cur.y));
    notnl += __popc(not_newline_bits(cur.z));
    notnl += __popc(not_newline_bits(cur.w));
#if defined(HG_ABLATE) && HG_ABLATE == 2  // profiling aid: no newline counting (results are wrong)
    const uint32_t c = cur.x == 0x0a0a0a0au ? 1u : 0u;
#else
    const uint32_t c = 128u - notnl;
#endif

#if defined(HG_ABLATE) && HG_ABLATE == 1  // profiling aid: no window filter (results are wrong)
    bool any = (cur.x ^ cur.y ^ cur.z ^ cur.w) == 0x12345678u;
#else
    bool any;
    if constexpr (DENSE) {
      const uint32_t nxt = __builtin_amdgcn_update_dpp(after, cur.x, 0x130 /* wave_shl:1 */, 0xF, 0xF, false);  // lane 63 keeps `after`
      any = ProbeBytes<LOG2, DENSE ? DENSE : 1>::template probe16<true>(cx.filter, cx.fold, cx.wa, cx.wb, cur, nxt) != 0;
    } else {
      any = Probe<LOG2, WIDE>::template probe4<true, FOLD>(cx.filter, cx.fold, cx.wa, cx.wb, cur) != 0;
    }
#endif

    // The last, partial tile: lanes past the end of the text hold zeros, and a zero window can pass the filter (the case
    // mask folds NUL onto ' ': a required literal of spaces, or a window value of zero).  Such a chunk holds no occurrence
    // and must not reach the drain, whose neighbour reads assume a position inside the text.  (Round 1 shipped without
    // this: the drain read up to a tile past the buffer, a GPU memory fault whenever that memory was not mapped.)
    if constexpr (!FULL) any = any && ((chunk0 + static_cast<uint64_t>(it) * 64u) << 4) < nbytes;

    const uint64_t nlm = __builtin_amdgcn_ballot_w64(c != 0);
    const uint64_t multi = __builtin_amdgcn_ballot_w64(c > 1);  // a 16-byte chunk with several newlines: rare in logs
    if (nlm) {
      if (first_it == HG_NONE32) {
        first_it = it;
        first_lane = __builtin_ctzll(nlm);
      }
      last_it = it;
      last_lane = 63u - __builtin_clzll(nlm);
    }
    // newlines of the iteration: a popcount of the ballot while every chunk has at most one (the per-lane prefix `before`,
    // newlines of the tile before this lane's chunk, is only needed by lanes that queue their chunk)
    uint32_t total, incl = 0;
    if (__builtin_expect(multi == 0, 1)) {
      total = __popcll(nlm);
    } else {
      incl = wave_inclusive_scan(c, lane);
      total = __builtin_amdgcn_readlane(incl, 63);
    }
    const uint64_t am = __builtin_amdgcn_ballot_w64(any);
    if (am) {  // remember the chunks; their windows are examined in batches of 64 (drain_batch)
      const uint32_t before = multi == 0 ? seen + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(nlm >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(nlm), 0u))
                                         : seen + incl - c;
      // the dwords next to the chunk, from the adjacent lanes (DPP wave shifts; the row's edge lanes get 0 and skip that condition)
      const uint32_t left = __builtin_amdgcn_update_dpp(0u, cur.w, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
      const uint32_t right = __builtin_amdgcn_update_dpp(0u, cur.x, 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
      if (any) {
        const uint32_t idx = qn + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(am >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(am), 0u));
        lds_u32 *e = cx.queue + idx * queue_entry_dw(LOG2, DENSE);
        e[0] = (static_cast<uint32_t>(it) * 64u + lane) | (before << 10);
        e[1] = static_cast<uint32_t>(tile);
        if (queue_stash(LOG2, DENSE)) {
          e[2] = left;
          e[3] = right;
          e[4] = cur.x;
          e[5] = cur.y;
          e[6] = cur.z;
          e[7] = cur.w;
        }
      }
      qn += __popcll(am);
      if (qn >= queue_cap(LOG2) - 64u) {
        const uint32_t n = qn < 64u ? qn : 64u;
        qn -= n;
        drain_batch<LOG2, WIDE, DENSE>(cx, qn, n, lane);
      }
    }
    seen += total;
  };

  // DEPTH: 16-byte loads in flight per lane
  if constexpr (FULL) {
    uint4 buf[DEPTH];
    uint32_t abuf[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; d++) { buf[d] = load_chunk(d); abuf[d] = load_after(d); }
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
      const uint4 cur = buf[it % DEPTH];
      const uint32_t after = abuf[it % DEPTH];
      if (it + DEPTH < ITERS) { buf[it % DEPTH] = load_chunk(it + DEPTH); abuf[it % DEPTH] = load_after(it + DEPTH); }
      body(it, cur, after);
    }
  } else {
#pragma unroll 1
    for (int it = 0; it < ITERS; it++) body(it, load_chunk(it), load_after(it));
  }

  // tile summary: exact offsets of the first / last newline (re-read two 16-byte chunks, L2-resident)
  const uint32_t nl_count = seen;
  uint32_t first_nl = HG_NONE32, last_nl = HG_NONE32;
  if (nl_count) {
    auto chunk_masks = [&](uint32_t it_, uint32_t lane_) -> uint32_t {  // bit b set: byte b of the chunk is '\n'
      const uint64_t g = tile * (HG_TILE_BYTES / 16) + it_ * 64u + lane_;
      const uint4 v = text16[g];
      const uint64_t byte0 = g << 4;
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
      uint32_t bitsm = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t m = hg_newline_mask(w[k]);
#pragma unroll
        for (int b = 0; b < 4; b++)
          if ((m >> (8 * b + 7)) & 1u) bitsm |= 1u << (k * 4 + b);
      }
      if (!FULL && byte0 + 16 > nbytes) bitsm &= (1u << static_cast<uint32_t>(nbytes - byte0)) - 1u;
      return bitsm;
    };
    const uint32_t fm = chunk_masks(first_it, first_lane), lm = chunk_masks(last_it, last_lane);
    first_nl = first_it * 1024u + first_lane * 16u + (__ffs(fm) - 1);
    last_nl = last_it * 1024u + last_lane * 16u + (31 - __clz(lm));
  }
  if (lane == 0) sums[tile] = HgTileSum{nl_count, first_nl, last_nl, nl_count ? nl_count - 1 : 0};
}

}  // namespace

// Register budget: HG_STREAM_WAVES resident waves per SIMD (the hot loop wants ~110 VGPRs with three 16-byte loads in flight).
#ifndef HG_STREAM_WAVES
#define HG_STREAM_WAVES 6
#endif
#ifndef HG_DEPTH_ALONE
#define HG_DEPTH_ALONE 3
#endif
#ifndef HG_DEPTH_SHARED
#define HG_DEPTH_SHARED 3
#endif
// DEPTH: 16-byte loads in flight per lane.  With non-temporal loads three is best for both kinds of launch (before, a launch
// that had three workgroups per CU to itself did better with one: deeper prefetch thrashed the L2).
// JOIN: the launch that joins a chunk behind the previous chunk's side passes (hg_stream_join_kernel, same code under its
// own name so that profiles keep the two kinds of launch apart); it also counts the tiles it took (HG_CNT_JOIN_TILES).
template <int LOG2, bool WIDE, int DENSE, int DEPTH, bool JOIN, bool FOLD>
__device__ __forceinline__ void stream_body(const uint4 *__restrict__ text16, uint64_t nbytes, uint64_t tile_begin, uint64_t tile_end, const uint4 *__restrict__ filter16,
                                            const uint4 *__restrict__ ext16, uint32_t fold, uint32_t wa, uint32_t wb, HgTileSum *__restrict__ sums, HgCand *__restrict__ cands,
                                            uint32_t seg_cap, uint32_t *__restrict__ seg_count, uint32_t *__restrict__ counters, uint32_t cursor_slot) {
  // LDS, one block so that the filter starts at offset 0 (its byte offsets then fold into the ds_read instructions):
  //   window hash slots (4 B each) | per-wave chunk queues | candidate counter
  constexpr uint32_t FILTER_U4 = (4u << LOG2) / 16, QUEUE_U4 = WG_WAVES * queue_cap(LOG2) * queue_entry_dw(LOG2, DENSE) * 4 / 16;
  __shared__ uint4 s_mem[FILTER_U4 + QUEUE_U4 + 1];
  {
    for (uint32_t i = threadIdx.x; i < FILTER_U4; i += WG_THREADS) s_mem[i] = filter16[i];
    if (threadIdx.x == 0) s_mem[FILTER_U4 + QUEUE_U4] = make_uint4(0, 0, 0, 0);
  }
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  StreamCtx cx;
  cx.text16 = text16;
  cx.nbytes = nbytes;
  cx.filter = (const lds_u32 *)(&s_mem[0]);
  cx.queue = (lds_u32 *)(&s_mem[FILTER_U4]) + wave * queue_cap(LOG2) * queue_entry_dw(LOG2, DENSE);
  cx.cand_count = (lds_u32 *)(&s_mem[FILTER_U4 + QUEUE_U4]);
  cx.ext = reinterpret_cast<const HgSlotInfo *>(ext16);
  cx.seg = cands + static_cast<uint64_t>(blockIdx.x) * seg_cap;  // this workgroup's private output segment
  cx.seg_cap = seg_cap;
  cx.fold = fold;
  cx.wa = wa;
  cx.wb = wb;
  const uint64_t full_tiles = nbytes >> HG_TILE_SHIFT;
  // Tile order: the workgroups of a chunk's launches draw runs of HG_STREAM_GRAB consecutive tiles from one cursor
  // (counters[cursor_slot], one atomic per run).  A run is consecutive in the text, so a segment's candidates cluster (the verify / confirm passes touch
  // neighbouring lines from neighbouring lanes).  Dealing the tiles out on demand instead of giving every workgroup a fixed
  // range took the pass from 73 % to 82 % of the HBM peak on an 8 GiB launch: no workgroup waits at the end for the slowest.
  // (A second launch joins a chunk late — a third workgroup per CU once the previous chunk's side passes have left: JOIN.
  // It lost 2-3 % while those passes took most of a stream launch, and pays since the confirm pass runs on windows.)
  lds_u32 *s_run = cx.cand_count + 1;  // (the dword next to the candidate counter: one LDS block, the filter stays at offset 0)
  const uint32_t ntile = static_cast<uint32_t>(tile_end - tile_begin);  // (tile numbers relative to the chunk: 32-bit scalars)
  const uint32_t nfull = full_tiles > tile_begin ? static_cast<uint32_t>(full_tiles - tile_begin < ntile ? full_tiles - tile_begin : ntile) : 0u;
  uint32_t qn = 0, joined = 0;
  for (;;) {
    if (threadIdx.x == 0) *s_run = atomicAdd(&counters[cursor_slot], HG_STREAM_GRAB);
    __syncthreads();
    const uint32_t r0 = __builtin_amdgcn_readfirstlane(*s_run);  // (block-uniform)
    __syncthreads();                                              // read by every wave before the next draw overwrites it
    if (r0 >= ntile) break;
    const uint32_t r1 = r0 + HG_STREAM_GRAB < ntile ? r0 + HG_STREAM_GRAB : ntile;
    if (JOIN) joined += r1 - r0;
    for (uint32_t r = r0 + wave; r < r1; r += WG_WAVES) {
      if (r < nfull) stream_tile<LOG2, WIDE, DENSE, true, DEPTH, FOLD>(cx, tile_begin + r, sums, lane, qn);
      else stream_tile<LOG2, WIDE, DENSE, false, DEPTH, FOLD>(cx, tile_begin + r, sums, lane, qn);
    }
  }
  if (qn) drain_batch<LOG2, WIDE, DENSE>(cx, 0u, qn, lane);
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t n = *cx.cand_count;
    seg_count[blockIdx.x] = n < seg_cap ? n : seg_cap;
    atomicAdd(&counters[HG_CNT_CANDS], n < seg_cap ? n : seg_cap);
    if (n > seg_cap) atomicMax(&counters[HG_CNT_CAND_NEED], n);
    if (JOIN && joined) atomicAdd(&counters[HG_CNT_JOIN_TILES], joined);
  }
}
template <int LOG2, bool WIDE, int DENSE, int DEPTH, bool FOLD = true>
__global__ __launch_bounds__(WG_THREADS) __attribute__((amdgpu_waves_per_eu(HG_STREAM_WAVES, 8))) void hg_stream_kernel(
    const uint4 *__restrict__ text16, uint64_t nbytes, uint64_t tile_begin, uint64_t tile_end, const uint4 *__restrict__ filter16, const uint4 *__restrict__ ext16, uint32_t fold,
    uint32_t wa, uint32_t wb, HgTileSum *__restrict__ sums, HgCand *__restrict__ cands, uint32_t seg_cap, uint32_t *__restrict__ seg_count, uint32_t *__restrict__ counters,
    uint32_t cursor_slot) {
#ifdef HG_STREAM_PRIO  // (experiment builds: a higher wave priority next to the side kernels changed nothing, profiles/r03_experiments.txt)
  __builtin_amdgcn_s_setprio(HG_STREAM_PRIO);
#endif
  stream_body<LOG2, WIDE, DENSE, DEPTH, false, FOLD>(text16, nbytes, tile_begin, tile_end, filter16, ext16, fold, wa, wb, sums, cands, seg_cap, seg_count, counters, cursor_slot);
}
template <int LOG2, int DENSE, bool FOLD = true>  // (only where three workgroups fit on a CU: filters of up to 32 KiB, single-probe mode)
__global__ __launch_bounds__(WG_THREADS) __attribute__((amdgpu_waves_per_eu(HG_STREAM_WAVES, 8))) void hg_stream_join_kernel(
    const uint4 *__restrict__ text16, uint64_t nbytes, uint64_t tile_begin, uint64_t tile_end, const uint4 *__restrict__ filter16, const uint4 *__restrict__ ext16, uint32_t fold,
    uint32_t wa, uint32_t wb, HgTileSum *__restrict__ sums, HgCand *__restrict__ cands, uint32_t seg_cap, uint32_t *__restrict__ seg_count, uint32_t *__restrict__ counters,
    uint32_t cursor_slot) {
#ifdef HG_STREAM_PRIO
  __builtin_amdgcn_s_setprio(HG_STREAM_PRIO);
#endif
  stream_body<LOG2, false, DENSE, HG_DEPTH_SHARED, true, FOLD>(text16, nbytes, tile_begin, tile_end, filter16, ext16, fold, wa, wb, sums, cands, seg_cap, seg_count, counters, cursor_slot);
}

// Host-side launcher: picks the instantiation for the database's filter size / mode.
namespace {
template <int L, bool W, int B, int D>
void launch_depth(const HgStreamArgs &a, uint32_t grid, hipStream_t stream) {
  const uint4 *t = reinterpret_cast<const uint4 *>(a.text);
  const uint4 *f = reinterpret_cast<const uint4 *>(a.filter);
  const uint4 *x = reinterpret_cast<const uint4 *>(a.ext);
  // (dword-aligned single-probe filters of sets that fold nothing: the variant without the fold instruction)
  if constexpr (!W && !B) {
    if (a.db.fold_mask == 0) {
      hipLaunchKernelGGL((hg_stream_kernel<L, W, B, D, false>), dim3(grid), dim3(WG_THREADS), 0, stream, t, a.nbytes, a.tile_begin, a.tile_end, f, x, a.db.fold_mask,
                         a.weights_a, a.dense ? a.weights_c : a.weights_b, a.sums, a.cands, a.cand_seg_cap, a.seg_count, a.counters, a.cursor_slot);
      return;
    }
  }
  hipLaunchKernelGGL((hg_stream_kernel<L, W, B, D, true>), dim3(grid), dim3(WG_THREADS), 0, stream, t, a.nbytes, a.tile_begin, a.tile_end, f, x, a.db.fold_mask,
                     a.weights_a, a.dense ? a.weights_c : a.weights_b, a.sums, a.cands, a.cand_seg_cap, a.seg_count, a.counters, a.cursor_slot);
}
template <int L, bool W, int B>
void launch_one(const HgStreamArgs &a, uint32_t grid, hipStream_t stream) {
  // filters up to 16 KiB leave room for three workgroups per CU: a launch that has the chip to itself prefetches one chunk ahead
  if (!W && !B && L <= 12 && a.alone) launch_depth<L, W, B, (!W && !B && L <= 12) ? HG_DEPTH_ALONE : HG_DEPTH_SHARED>(a, grid, stream);
  else launch_depth<L, W, B, HG_DEPTH_SHARED>(a, grid, stream);
}
template <int L, bool W, int B>
int blocks_one() {
  int n = 0;
  // (the launch that has the chip to itself decides how many workgroups a CU holds)
  (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (hg_stream_kernel<L, W, B, (!W && !B && L <= 12) ? HG_DEPTH_ALONE : HG_DEPTH_SHARED>), WG_THREADS, 0);
  return n > 0 ? n : 1;
}
}  // namespace
namespace {
template <int L, int B>
void launch_join(const HgStreamArgs &a, uint32_t grid, hipStream_t stream) {
  if constexpr (B == 0) {
    if (a.db.fold_mask == 0) {
      hipLaunchKernelGGL((hg_stream_join_kernel<L, B, false>), dim3(grid), dim3(WG_THREADS), 0, stream, reinterpret_cast<const uint4 *>(a.text), a.nbytes, a.tile_begin, a.tile_end,
                         reinterpret_cast<const uint4 *>(a.filter), reinterpret_cast<const uint4 *>(a.ext), a.db.fold_mask, a.weights_a, a.dense ? a.weights_c : a.weights_b, a.sums,
                         a.cands, a.cand_seg_cap, a.seg_count, a.counters, a.cursor_slot);
      return;
    }
  }
  hipLaunchKernelGGL((hg_stream_join_kernel<L, B, true>), dim3(grid), dim3(WG_THREADS), 0, stream, reinterpret_cast<const uint4 *>(a.text), a.nbytes, a.tile_begin, a.tile_end,
                     reinterpret_cast<const uint4 *>(a.filter), reinterpret_cast<const uint4 *>(a.ext), a.db.fold_mask, a.weights_a, a.dense ? a.weights_c : a.weights_b, a.sums,
                     a.cands, a.cand_seg_cap, a.seg_count, a.counters, a.cursor_slot);
}
}  // namespace
// The joiner launch (hg_engine.hip).  false: no such instantiation (wide filters, filters beyond 32 KiB: no room for a third
// workgroup on a CU anyway).
bool hg_launch_stream_join(const HgStreamArgs &a, uint32_t grid, hipStream_t stream) {
  if (a.filter_wide || a.filter_log2 < 11 || a.filter_log2 > 13 || a.dense > 2) return false;
  switch (a.filter_log2 * 4 + a.dense) {
    case 11 * 4 + 0: launch_join<11, 0>(a, grid, stream); break;
    case 11 * 4 + 1: launch_join<11, 1>(a, grid, stream); break;
    case 11 * 4 + 2: launch_join<11, 2>(a, grid, stream); break;
    case 12 * 4 + 0: launch_join<12, 0>(a, grid, stream); break;
    case 12 * 4 + 1: launch_join<12, 1>(a, grid, stream); break;
    case 12 * 4 + 2: launch_join<12, 2>(a, grid, stream); break;
    case 13 * 4 + 0: launch_join<13, 0>(a, grid, stream); break;
    case 13 * 4 + 1: launch_join<13, 1>(a, grid, stream); break;
    case 13 * 4 + 2: launch_join<13, 2>(a, grid, stream); break;
    default: return false;
  }
  return true;
}
// Returns false when no instantiation exists for the database's (filter size, mode): nothing was launched.
bool hg_launch_stream(const HgStreamArgs &a, uint32_t grid, hipStream_t stream) {
  if (a.filter_wide) {
    switch (a.filter_log2) {
      case 13: launch_one<13, true, 0>(a, grid, stream); break;
      case 14: launch_one<14, true, 0>(a, grid, stream); break;
      case 15: launch_one<15, true, 0>(a, grid, stream); break;
      default: return false;
    }
    return true;
  }
  if (a.dense == 1) {  // byte-aligned probing, a window at every byte
    switch (a.filter_log2) {
      case 11: launch_one<11, false, 1>(a, grid, stream); break;
      case 12: launch_one<12, false, 1>(a, grid, stream); break;
      case 13: launch_one<13, false, 1>(a, grid, stream); break;
      case 14: launch_one<14, false, 1>(a, grid, stream); break;
      case 15: launch_one<15, false, 1>(a, grid, stream); break;
      default: return false;
    }
    return true;
  }
  if (a.dense == 2) {  // ... at every second byte
    switch (a.filter_log2) {
      case 11: launch_one<11, false, 2>(a, grid, stream); break;
      case 12: launch_one<12, false, 2>(a, grid, stream); break;
      case 13: launch_one<13, false, 2>(a, grid, stream); break;
      case 14: launch_one<14, false, 2>(a, grid, stream); break;
      case 15: launch_one<15, false, 2>(a, grid, stream); break;
      default: return false;
    }
    return true;
  }
  switch (a.filter_log2) {
    case 11: launch_one<11, false, 0>(a, grid, stream); break;
    case 12: launch_one<12, false, 0>(a, grid, stream); break;
    case 13: launch_one<13, false, 0>(a, grid, stream); break;
    case 14: launch_one<14, false, 0>(a, grid, stream); break;
    case 15: launch_one<15, false, 0>(a, grid, stream); break;
    default: return false;
  }
  return true;
}
int hg_stream_blocks_per_cu(uint32_t filter_log2, uint32_t filter_wide, uint32_t dense) {
  if (filter_wide) return filter_log2 == 13 ? blocks_one<13, true, 0>() : (filter_log2 == 14 ? blocks_one<14, true, 0>() : blocks_one<15, true, 0>());
  if (dense == 1) {
    switch (filter_log2) {
      case 11: return blocks_one<11, false, 1>();
      case 12: return blocks_one<12, false, 1>();
      case 13: return blocks_one<13, false, 1>();
      case 14: return blocks_one<14, false, 1>();
      default: return blocks_one<15, false, 1>();
    }
  }
  if (dense == 2) {
    switch (filter_log2) {
      case 11: return blocks_one<11, false, 2>();
      case 12: return blocks_one<12, false, 2>();
      case 13: return blocks_one<13, false, 2>();
      case 14: return blocks_one<14, false, 2>();
      default: return blocks_one<15, false, 2>();
    }
  }
  switch (filter_log2) {
    case 11: return blocks_one<11, false, 0>();
    case 12: return blocks_one<12, false, 0>();
    case 13: return blocks_one<13, false, 0>();
    case 14: return blocks_one<14, false, 0>();
    default: return blocks_one<15, false, 0>();
  }
}

