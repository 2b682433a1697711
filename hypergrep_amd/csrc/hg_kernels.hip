// gfx950 kernels of the line-scan path behind the stream pass (hg_stream.hip; huge automata: hg_huge.hip).  Replaces the
// reference's hot loop (hypergrep/lib/c/hyperscanner.c:198-226: gzgets -> strlen -> hs_scan -> hs_callback per line) with
//
//   hg_tile_*             3-launch scan of the tile newline summaries -> global piece numbers
//   hg_verify_kernel      (candidate, literal) pairs flattened over the wave: one straight-line literal compare per lane
//   hg_confirm_*          verified occurrences by automaton shape: locate the line piece, run the automaton, emit hits
//   (hg_always_on_*       patterns without a usable required literal: hg_always_on.hip)
//   hg_key/keep/scatter   ordering + SINGLEMATCH / duplicate rules (sort itself: rocPRIM radix sort)
//
// Byte/integer work, HBM-bound: no MFMA anywhere.  Wave64 only.
#include <hip/hip_runtime.h>
#include <cstdlib>

#include "hg_confirm_dev.h"
#include "hg_core.h"
#include "hg_engine.h"
#include "hg_post.h"
#include "hg_sink_dev.h"
#include "hg_tables_dev.h"

namespace {

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
// Tile scan (reduce / spine / apply).  256 threads x 4 tiles per block.
namespace {
constexpr int TS_THREADS = 256;
constexpr int TS_PER_THREAD = 4;

__device__ HgTileElem identity_elem() {
  HgTileElem e;
  e.k1 = e.d = e.new_cs = 0;
  e.has_nl = 0;
  e.pad = 0;
  return e;
}
// Inclusive block scan of monoid elements (Hillis-Steele in LDS).
__device__ HgTileElem block_scan(HgTileElem v, HgTileElem *sh, uint64_t bs1) {
  const int t = threadIdx.x;
  sh[t] = v;
  __syncthreads();
  for (int o = 1; o < TS_THREADS; o <<= 1) {
    HgTileElem left = identity_elem();
    if (t >= o) left = sh[t - o];
    __syncthreads();
    if (t >= o) {
      v = hg_tile_combine(left, v, bs1);
      sh[t] = v;
    }
    __syncthreads();
  }
  return v;
}
__device__ HgTileElem thread_elems(const HgTileSum *sums, uint64_t tile_end, uint64_t first, uint64_t bs1, HgTileElem *each) {
  HgTileElem acc = identity_elem();
  for (int k = 0; k < TS_PER_THREAD; k++) {
    uint64_t t = first + k;
    HgTileElem e = identity_elem();
    if (t < tile_end) e = hg_tile_elem(sums[t], t << HG_TILE_SHIFT);
    if (each) each[k] = e;
    acc = hg_tile_combine(acc, e, bs1);
  }
  return acc;
}
}  // namespace

// The three kernels work on the tile range [tile_begin, tile_end) (one chunk of the text).
__global__ __launch_bounds__(TS_THREADS) void hg_tile_reduce_kernel(const HgTileSum *sums, uint64_t tile_begin, uint64_t tile_end, uint64_t bs1,
                                                                    HgTileElem *agg) {
  __shared__ HgTileElem sh[TS_THREADS];
  uint64_t first = tile_begin + (static_cast<uint64_t>(blockIdx.x) * TS_THREADS + threadIdx.x) * TS_PER_THREAD;
  HgTileElem v = thread_elems(sums, tile_end, first, bs1, nullptr);
  v = block_scan(v, sh, bs1);
  if (threadIdx.x == TS_THREADS - 1) agg[blockIdx.x] = v;
}

// One block: exclusive scan of the block aggregates -> state at the start of each block.  *state holds the state at
// the start of the range on entry and the state at its end on exit (chunks chain through it on the device).
__global__ __launch_bounds__(TS_THREADS) void hg_tile_spine_kernel(const HgTileElem *agg, uint32_t nblocks, uint64_t bs1, HgTileBase *block_base,
                                                                   HgTileBase *state) {
  __shared__ HgTileElem sh[TS_THREADS];
  __shared__ HgTileBase carry;
  if (threadIdx.x == 0) carry = *state;
  __syncthreads();
  for (uint32_t base = 0; base < nblocks; base += TS_THREADS) {
    uint32_t i = base + threadIdx.x;
    HgTileElem v = identity_elem();
    if (i < nblocks) v = agg[i];
    HgTileElem inc = block_scan(v, sh, bs1);
    HgTileBase st = carry;
    // exclusive: state before block i = apply(carry, inclusive(i-1))
    HgTileElem excl = identity_elem();
    if (threadIdx.x > 0) excl = sh[threadIdx.x - 1];
    if (i < nblocks) block_base[i] = hg_tile_apply(st, excl, bs1);
    __syncthreads();
    if (threadIdx.x == TS_THREADS - 1) carry = hg_tile_apply(st, inc, bs1);
    __syncthreads();
  }
  if (threadIdx.x == 0) *state = carry;
}

__global__ __launch_bounds__(TS_THREADS) void hg_tile_apply_kernel(const HgTileSum *sums, uint64_t tile_begin, uint64_t tile_end, uint64_t bs1,
                                                                   const HgTileBase *block_base, HgTileBase *bases) {
  __shared__ HgTileElem sh[TS_THREADS];
  uint64_t first = tile_begin + (static_cast<uint64_t>(blockIdx.x) * TS_THREADS + threadIdx.x) * TS_PER_THREAD;
  HgTileElem each[TS_PER_THREAD];
  HgTileElem v = thread_elems(sums, tile_end, first, bs1, each);
  block_scan(v, sh, bs1);
  HgTileElem excl = identity_elem();
  if (threadIdx.x > 0) excl = sh[threadIdx.x - 1];
  HgTileBase st = hg_tile_apply(block_base[blockIdx.x], excl, bs1);
  for (int k = 0; k < TS_PER_THREAD; k++) {
    uint64_t t = first + k;
    if (t < tile_end) bases[t] = st;
    st = hg_tile_apply(st, each[k], bs1);
  }
}

// ------------------------------------------------------------------------------------------------
// Small-buffer mode only (buffer_size - 1 < tile): lines inside a tile may split into several pieces, so the
// per-tile count of inner pieces is recomputed by walking the tile (one thread per tile; slow path).
__global__ __launch_bounds__(256) void hg_tile_inner_kernel(const uint8_t *text, HgTileSum *sums, uint64_t tile_begin, uint64_t tile_end, uint64_t bs1) {
  for (uint64_t t = tile_begin + static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < tile_end; t += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
    HgTileSum s = sums[t];
    if (s.nl_count < 2) continue;
    uint64_t base = t << HG_TILE_SHIFT;
    sums[t].inner = static_cast<uint32_t>(hg_inner_pieces(text, base + s.first_nl + 1, base + s.last_nl + 1, bs1));
  }
}

// One lane per window hit of the stream pass.  Candidate segment s (written by stream workgroup s) is consumed by
// the CONFIRM_SPLIT blocks s*CONFIRM_SPLIT .. +CONFIRM_SPLIT-1, so that even a few hundred thousand candidates
// keep every CU busy: the work is all memory latency.
// Verify pass.  A window hit of the stream pass names a bucket of (literal, offset) pairs that contain that window; the
// wave flattens (hit, pair) over its 64 hits (prefix sum of the bucket sizes + binary search by shuffles) so that every
// lane compares exactly ONE literal against the text per round, with straight-line code: one window load, the factor
// record and the text bytes as 16-byte loads (the text unaligned), no per-lane loops.  The pass is bound by the number of
// divergent memory instructions a wave issues (each costs >= 64 address cycles), not by bytes: the per-lane bucket loop it
// replaces issued ~700 per wave on the round-1 workload, this issues ~30.
// Verified occurrences are appended (one atomic per wave, mode and round) to per-mode, sharded lists; every confirm
// routine, including the literal-only one, then runs over its own list with all lanes doing the same work.
typedef uint32_t hg_u32x4_unaligned __attribute__((ext_vector_type(4), aligned(1)));

// Automaton modes (1, 2): the confirm pass wants its lists keyed by pattern (confirm_tables_body), i.e. up to 128 lists per
// block.  One global atomic per occurrence and list costs the pass 2 ms per 32 GiB, so the block stages these occurrences in
// LDS and files them bin by bin: one global atomic per non-empty (mode, list) bin and flush.
constexpr uint32_t VERIFY_STAGE_CAP = 768;  // 12 KiB
struct VerifyStage {
  HgDeferred item[VERIFY_STAGE_CAP];
  uint8_t bin[VERIFY_STAGE_CAP];  // (mode - 1) * HG_DEFER_SHARDS + list
  uint32_t n;
  uint32_t bin_count[2 * HG_DEFER_SHARDS], bin_base[2 * HG_DEFER_SHARDS];
};
__device__ __forceinline__ void verify_stage_flush(const HgConfirmArgs &a, VerifyStage &st) {  // whole block
  __syncthreads();
  const uint32_t n = st.n < VERIFY_STAGE_CAP ? st.n : VERIFY_STAGE_CAP;
  if (threadIdx.x < 2 * HG_DEFER_SHARDS) st.bin_count[threadIdx.x] = 0;
  __syncthreads();
  constexpr uint32_t PER_THREAD = (VERIFY_STAGE_CAP + 255) / 256;
  uint32_t rank[PER_THREAD];
#pragma unroll
  for (uint32_t q = 0; q < PER_THREAD; q++) {
    const uint32_t i = threadIdx.x + q * 256u;
    rank[q] = i < n ? atomicAdd(&st.bin_count[st.bin[i]], 1u) : 0u;
  }
  __syncthreads();
  if (threadIdx.x < 2 * HG_DEFER_SHARDS && st.bin_count[threadIdx.x])
    st.bin_base[threadIdx.x] = atomicAdd(&a.defer_count[HG_DEFER_SHARDS + threadIdx.x], st.bin_count[threadIdx.x]);  // modes 1, 2 follow mode 0's counters
  __syncthreads();
#pragma unroll
  for (uint32_t q = 0; q < PER_THREAD; q++) {
    const uint32_t i = threadIdx.x + q * 256u;
    if (i < n) {
      const uint32_t bin = st.bin[i], m = 1u + bin / HG_DEFER_SHARDS, list = bin % HG_DEFER_SHARDS;
      const uint32_t slot = st.bin_base[bin] + rank[q];
      if (slot < a.defer_shard_cap) a.deferred[(static_cast<uint64_t>(a.list_of_mode[m]) * HG_DEFER_SHARDS + list) * a.defer_shard_cap + slot] = st.item[i];
      else atomicMax(&a.counters[HG_CNT_DEFER_NEED], slot + 1);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) st.n = 0;
  __syncthreads();
}

// Direct window table lookup (hg_db.h HgWinBucket): the bucket's four values in one 16-byte fetch, then the payload of the one
// that matched (the same 32-byte sector).  Returns the owner's factor_off, HG_WTAB_SHARED or HG_WTAB_EMPTY.
__device__ __forceinline__ uint32_t wtab_lookup(const HgWinBucket *tab, uint32_t bucket_mask, uint32_t folded) {
  for (uint32_t b = hg_wtab_bucket(folded, bucket_mask);; b = (b + 1u) & bucket_mask) {
    const uint4 *e = reinterpret_cast<const uint4 *>(tab + b);
    uint4 v = e[0], p = e[1];  // the four values and their payloads: one 32-byte sector, both loads in flight together
    // (all eight words are wanted HERE: left to itself the compiler loads value[0], then value[1..2] behind a branch, then the
    // payload behind another — three dependent round trips where one was meant; config 3's verify pass took 510 us per 8 GiB
    // with that against 223 without the table)
    asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w), "+v"(p.x), "+v"(p.y), "+v"(p.z), "+v"(p.w));
    // (an empty slot holds value 0 and payload EMPTY: a match on an empty slot's value returns EMPTY, which is right)
    if (v.x == folded) return p.x;
    if (v.y == folded) return p.y;
    if (v.z == folded) return p.z;
    if (v.w == folded) return p.w;
    if (p.w == HG_WTAB_EMPTY) return HG_WTAB_EMPTY;  // the bucket has room left: the value would be here (buckets fill front to back)
  }
}
// Does the literal of `f` (len bytes, HgFactor) occur at text[fs, fs + len)?  The record is one cache line: the literal as
// 16-byte pieces, the compare masks from its case bits; the text unaligned.  fs + len <= nbytes.
__device__ __forceinline__ bool literal_occurs(const HgFactor *f, uint32_t len, const uint8_t *text, uint64_t fs, uint64_t readable) {
  const uint8_t *tp = text + fs;
  const uint32_t cb = f->casebits;  // (zero for an exact literal; a set may hold case-insensitive literals without folding its windows)
  uint32_t diff = 0;
  if (fs + (len > 16 ? 32u : 16u) <= readable) {
    const uint4 l0 = *reinterpret_cast<const uint4 *>(f->lit);
    const hg_u32x4_unaligned x0 = *reinterpret_cast<const hg_u32x4_unaligned *>(tp);
    diff = ((x0[0] ^ l0.x) & hg_factor_mask_dword(cb, len, 0)) | ((x0[1] ^ l0.y) & hg_factor_mask_dword(cb, len, 1)) |
           ((x0[2] ^ l0.z) & hg_factor_mask_dword(cb, len, 2)) | ((x0[3] ^ l0.w) & hg_factor_mask_dword(cb, len, 3));
    if (len > 16) {
      const uint4 l1 = *reinterpret_cast<const uint4 *>(f->lit + 16);
      const hg_u32x4_unaligned x1 = *reinterpret_cast<const hg_u32x4_unaligned *>(tp + 16);
      diff |= ((x1[0] ^ l1.x) & hg_factor_mask_dword(cb, len, 4)) | ((x1[1] ^ l1.y) & hg_factor_mask_dword(cb, len, 5)) |
              ((x1[2] ^ l1.z) & hg_factor_mask_dword(cb, len, 6)) | ((x1[3] ^ l1.w) & hg_factor_mask_dword(cb, len, 7));
    }
  } else {  // the last bytes of the buffer: byte compares
    for (uint32_t b = 0; b < len; b++) diff |= (tp[b] ^ f->lit[b]) & (((cb >> b) & 1u) ? 0xDFu : 0xFFu);
  }
  return diff == 0;
}

// STAGED: the set has automaton expressions of confirm modes 1 / 2, whose occurrences are staged in LDS and filed by pattern.
// A set without them (literals, config 5) runs the variant without the 14 KiB staging area: beside two stream workgroups
// with 64 KiB filters a CU has 16 KiB of LDS left, and the staging area held the pass to ONE block per CU.
template <bool STAGED>
__device__ __forceinline__ void verify_body(const HgConfirmArgs &a, VerifyStage *stage) {
  VerifyStage &s_stage = *stage;
  if (STAGED) {
    if (threadIdx.x == 0) s_stage.n = 0;
    __syncthreads();
  }
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#ifdef HG_PROFILE_CONFIRM
  const uint64_t pf_t0 = wall_clock64();
  uint32_t pf_rounds = 0, pf_pairs = 0;
#endif
  const uint32_t seg = blockIdx.x / HG_CONFIRM_SPLIT, sub = blockIdx.x % HG_CONFIRM_SPLIT;
  const bool joined = seg >= a.join_seg0;  // a segment of the joiner launch (smaller)
  const uint32_t seg_cap = joined ? a.join_seg_cap : a.cand_seg_cap;
  const HgCand *cseg = joined ? a.cands + static_cast<uint64_t>(a.join_seg0) * a.cand_seg_cap + static_cast<uint64_t>(seg - a.join_seg0) * a.join_seg_cap
                              : a.cands + static_cast<uint64_t>(seg) * a.cand_seg_cap;
  uint32_t n = a.seg_count[seg];
  if (n > seg_cap) n = seg_cap;  // (never more than a segment holds, whatever the counter says)
  const uint32_t shard = blockIdx.x % HG_DEFER_SHARDS;
  const uint32_t fold = a.db.fold_mask;
  const uint64_t readable = (a.nbytes + 15) & ~15ull;  // the text buffer can be read up to here
  for (uint32_t base0 = sub * 256u; base0 < n; base0 += HG_CONFIRM_SPLIT * 256u) {  // block-uniform: the staging flush is a block affair
    if (STAGED) {
      __syncthreads();  // every wave has finished the previous round: the fill level is final ...
      const uint32_t staged = s_stage.n;
      __syncthreads();  // ... and has been read by all before anyone stages again
      if (staged >= VERIFY_STAGE_CAP / 2) verify_stage_flush(a, s_stage);
    }
    const uint32_t base = base0 + wave * 64u;
    const uint32_t i = base + lane;
    HgCand c{0, 0, 0};
    uint32_t j0 = 0, cnt = 0, folded = 0, direct_fo = 0, disc_h = 0, disc_d = 0;
    bool direct = false, shared = false;
    if (i < n) {
      c = cseg[i];
      folded = (c.word | fold) & a.db.window_mask;
      // the direct table: a value it does not hold belongs to no literal (the filter's false positives end here, one fetch
      // each); a value with one owner names it; only windows that several literals share go through the discriminated buckets
      // (the shared windows' discriminator word is fetched beside the table bucket, not behind it: a shared window's chain of
      // fetches is then no longer than it was before the table existed — config 3's class expressions share their windows)
      disc_h = hg_hash_window(folded);
      disc_d = a.db.disc[disc_h];
      if (a.db.wtab_first) {  // (uniform: a property of the pattern set, HgDb::wtab_first)
        direct_fo = wtab_lookup(a.db.wtab, a.db.wtab_mask, folded);
        shared = direct_fo == HG_WTAB_SHARED;
        direct = !shared && direct_fo != HG_WTAB_EMPTY;
      } else {
        shared = true;
      }
      if (direct) cnt = 1;
    }
    if (shared) {
      // the group's discriminator dword of the text selects the bucket (hg_disc_range, with one aligned dword load)
      const uint32_t h = disc_h;
      const uint32_t d = disc_d;
      const uint32_t sel = d >> 8;
      uint32_t key = 0;
      bool inside = true;
      if (sel) {
        const int64_t at = static_cast<int64_t>(c.pos) + static_cast<int8_t>(d & 0xFFu);
        const uint32_t lo = hg_ctz(sel), hi = 31u - hg_clz32(sel);
        inside = at + static_cast<int64_t>(lo) >= 0 && static_cast<uint64_t>(at + hi) < a.nbytes;
        if (inside) {
          // dword-aligned windows: pos and delta are multiples of 4, one aligned load (it ends before the buffer's size
          // rounded up to 4).  Byte-aligned windows (and at < 0, where only later bytes are selected): the selected bytes
          // one by one, each of them inside the text.
          uint32_t v = 0;
          if (at >= 0 && (at & 3) == 0) {
            v = *reinterpret_cast<const uint32_t *>(a.text + at);
          } else {
            for (uint32_t b = lo; b <= hi; b++) v |= static_cast<uint32_t>(a.text[at + static_cast<int64_t>(b)]) << (8u * b);
          }
          key = (v | fold) & hg_disc_bytes(sel);
        }
      }
      if (inside) {
        const uint32_t h2 = hg_disc_bucket(h, key);
        j0 = a.db.bucket_off2[h2];
        cnt = a.db.bucket_off2[h2 + 1] - j0;
      }
    }
    const uint32_t incl = wave_inclusive_scan(cnt, lane), start = incl - cnt;
    const uint32_t total = __shfl(incl, 63, 64);
#ifdef HG_PROFILE_CONFIRM
    pf_rounds++;
    pf_pairs += total;
#endif
    const uint32_t pos_lo = static_cast<uint32_t>(c.pos), pos_hi = static_cast<uint32_t>(c.pos >> 32);
    for (uint32_t t0 = 0; t0 < total; t0 += 64) {  // wave-uniform
      const uint32_t t = t0 + lane;
      // owner of item t: the last lane whose exclusive prefix is <= t
      uint32_t owner = 0;
#pragma unroll
      for (uint32_t step = 32; step; step >>= 1) {
        const uint32_t probe = owner + step;
        const uint32_t sp = __shfl(start, probe & 63u, 64);
        if (probe < 64u && sp <= t) owner = probe;
      }
      const uint32_t o_start = __shfl(start, owner, 64), o_j0 = __shfl(j0, owner, 64), o_folded = __shfl(folded, owner, 64);
      const uint32_t o_direct = __shfl(direct ? 1u : 0u, owner, 64), o_fo = __shfl(direct_fo, owner, 64);
      const uint32_t o_rank = __shfl(c.rank, owner, 64);
      const uint64_t pos = (static_cast<uint64_t>(static_cast<uint32_t>(__shfl(pos_hi, owner, 64))) << 32) | static_cast<uint32_t>(__shfl(pos_lo, owner, 64));
      bool ok = false;
      uint32_t mode = 0, tag = 0, mode_rank = 0;
      if (t < total) {
        HgWindow win{o_folded, o_fo};
        if (!o_direct) win = a.db.windows2[o_j0 + (t - o_start)];
        if (win.value == o_folded) {
          const uint32_t off = win.factor_off & 0xffu;
          const HgFactor *f = &a.db.factors[win.factor_off >> 8];
          const uint4 hdr = *reinterpret_cast<const uint4 *>(f);  // pattern, len, mode, mode_rank
          const uint32_t len = hdr.y;
          if (pos >= off && pos - off + len <= a.nbytes) {
            const uint64_t fs = pos - off;
            const uint32_t diff = literal_occurs(f, len, a.text, fs, readable) ? 0u : 1u;
            if (diff == 0) {
              ok = true;
              mode = hdr.z;
              mode_rank = hdr.w;
              tag = hdr.x | (off << 24);
            }
          }
        }
      }
      if (!__builtin_amdgcn_ballot_w64(ok)) continue;
#pragma unroll
      for (uint32_t m = 0; m < HG_CONFIRM_MODES; m++) {
        const uint64_t mm = __builtin_amdgcn_ballot_w64(ok && mode == m);
        if (!mm) continue;
        if (STAGED && (m == 1 || m == 2)) {  // staged, keyed by pattern (a pattern set with few such patterns spreads each over several lists)
          uint32_t at0 = 0;
          if (lane == 0) at0 = atomicAdd(&s_stage.n, static_cast<uint32_t>(__popcll(mm)));
          at0 = __builtin_amdgcn_readfirstlane(at0);
          if (ok && mode == m) {
            const uint32_t at = at0 + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mm >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mm), 0u));
            // (by the pattern's rank in its mode, not its index: indices that share a factor with HG_DEFER_SHARDS piled several
            // patterns into the same lists, and a confirm wave works through the patterns of its batch one after the other)
            const uint32_t list = (mode_rank * a.list_spread[m] + blockIdx.x % a.list_spread[m]) % HG_DEFER_SHARDS;
            if (at < VERIFY_STAGE_CAP) {
              s_stage.item[at] = HgDeferred{pos, tag, o_rank};
              s_stage.bin[at] = static_cast<uint8_t>((m - 1u) * HG_DEFER_SHARDS + list);
            } else {  // staging full (a wave met very large buckets): file this one directly
              const uint32_t slot = atomicAdd(&a.defer_count[m * HG_DEFER_SHARDS + list], 1u);
              if (slot < a.defer_shard_cap) a.deferred[(static_cast<uint64_t>(a.list_of_mode[m]) * HG_DEFER_SHARDS + list) * a.defer_shard_cap + slot] = HgDeferred{pos, tag, o_rank};
              else atomicMax(&a.counters[HG_CNT_DEFER_NEED], slot + 1);
            }
          }
          continue;
        }
        uint32_t slot0 = 0;
        if (lane == 0) slot0 = atomicAdd(&a.defer_count[m * HG_DEFER_SHARDS + shard], static_cast<uint32_t>(__popcll(mm)));
        slot0 = __builtin_amdgcn_readfirstlane(slot0);
        if (ok && mode == m) {
          const uint32_t slot = slot0 + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mm >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mm), 0u));
          if (slot < a.defer_shard_cap) a.deferred[(static_cast<uint64_t>(a.list_of_mode[m]) * HG_DEFER_SHARDS + shard) * a.defer_shard_cap + slot] = HgDeferred{pos, tag, o_rank};
          else atomicMax(&a.counters[HG_CNT_DEFER_NEED], slot + 1);
        }
      }
    }
  }
  if (STAGED) verify_stage_flush(a, s_stage);
#ifdef HG_PROFILE_CONFIRM
  if (lane == 0) a.tmp_hits[(6u << 20) + 65536u + blockIdx.x * 4u + wave] = HgHit{pf_t0, static_cast<uint32_t>(wall_clock64() - pf_t0), pf_rounds | (pf_pairs << 8)};
#endif
}
// (the register budgets: beside two stream workgroups — 4 waves of 80 VGPRs per SIMD — a SIMD has 192 VGPRs left; at 64 the
// pass keeps three waves per SIMD there, at 80 two.  The pass lives on resident waves: every candidate is a chain of fetches.)
// Waves per SIMD of the side kernels.  They run beside the stream pass, and what they take from it is not issue slots (raising
// the stream waves' priority changes nothing) but room in the memory pipeline: every resident side wave keeps scattered 16-byte
// requests in flight.  Config 5, 32 GiB, same box: verify at 8 waves + literal confirm at 7 (its natural count once its line walks
// pack their byte masks with v_dot4) 14.95 ms, literal confirm capped at 4: 14.3, at 3 with verify at 4: 13.6-13.7, verify at 2: 14.4-14.8
// (then the side passes are the longer chain).  Config 3 does not notice (6.70-6.78 ms either way).
#ifndef HG_VERIFY_WAVES
#define HG_VERIFY_WAVES 4
#endif
#ifndef HG_LITERAL_WAVES
#define HG_LITERAL_WAVES 3
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(HG_VERIFY_WAVES, HG_VERIFY_WAVES))) void hg_verify_kernel(HgConfirmArgs a) {
  __shared__ VerifyStage s_stage;
  verify_body<true>(a, &s_stage);
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(HG_VERIFY_WAVES, HG_VERIFY_WAVES))) void hg_verify_lean_kernel(HgConfirmArgs a) { verify_body<false>(a, nullptr); }

// Confirm passes over the lists of verified occurrences, one routine per confirm mode so that the lanes of a wave do the
// same work:
//   MODE 0 literal-only expressions (the verified occurrence is the match: locate the piece, apply the NUL rules),
//   MODE 1 context-free single-word automaton, MODE 2 <= 2 state words with boundary conditions (confirm_tables_body),
//   MODE 3 the scalar reference routine (multi-word state, all-matches mode).
// vblock / vgrid: this block's index among the blocks working on MODE (several modes can share one launch).
template <int MODE>  // 0: literal-only, 3: generic
__device__ __forceinline__ void confirm_body(const HgConfirmArgs &a, uint32_t vblock, uint32_t vgrid) {
  __shared__ uint32_t s_n, s_base;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  const uint64_t seg0 = static_cast<uint64_t>(blockIdx.x) * a.hit_seg_cap;
  const HitSink sink{a.tmp_hits + seg0, a.tmp_aux + seg0, a.hit_seg_cap, &s_n};
  // block b walks list b % HG_DEFER_SHARDS with the blocks that share it
  const uint32_t shard = vblock % HG_DEFER_SHARDS, peer = vblock / HG_DEFER_SHARDS, peers = (vgrid + HG_DEFER_SHARDS - 1 - shard) / HG_DEFER_SHARDS;
  const HgDeferred *dlist = a.deferred + (static_cast<uint64_t>(a.list_of_mode[MODE]) * HG_DEFER_SHARDS + shard) * a.defer_shard_cap;
  uint32_t n = a.defer_count[MODE * HG_DEFER_SHARDS + shard];
  if (n > a.defer_shard_cap) n = a.defer_shard_cap;
  for (uint32_t i = peer * blockDim.x + threadIdx.x; i < n; i += peers * blockDim.x) {
    const HgDeferred d = dlist[i];
    const uint32_t pattern = d.pattern & (HG_MAX_PATTERNS - 1u);
    const HgPattern &p = a.db.patterns[pattern];
    const uint32_t id = p.id;
    const bool single = MODE != 3 || p.single != 0;  // modes 0..2 are SINGLEMATCH expressions by definition (hg_confirm_mode)
    auto emit = [&](uint64_t line_no, uint32_t to, uint64_t start, uint32_t len) { sink.push(a, line_no, id, to, start, len, pattern, single); };
    if (MODE == 0) hgdev::confirm_literal(a.text, a.nbytes, a.sums, a.bases, a.bs1, d.pos, d.rank, d.pos - (d.pattern >> 24), p.max_len, emit);
    else hg_confirm(a.db, a.text, a.nbytes, a.sums, a.bases, a.bs1, d.pos, pattern, d.rank, emit);
  }
  flush_hits(a, &s_n, &s_base);
}

// Modes 1 and 2 (automata of one or two state words).  Run with the tables in HBM, one such occurrence costs hundreds of
// scattered 4-byte loads (reach / follow / boundary masks per text byte) and the pass is bound by memory requests.  Here
// the lists are keyed by pattern (the verify pass files an occurrence under pattern % HG_DEFER_SHARDS), each block takes an
// equal slice of the concatenated lists, and a wave stages the tables of ONE pattern at a time in LDS (<= 3 KiB) and runs
// the lanes whose occurrence belongs to it: a slice holds a handful of patterns, and the only HBM traffic left is the text.
// A wave copies the automaton tables of ONE pattern into its LDS area (<= 3 KiB): reach[256][nw] as 16-byte pieces, the
// small tables a dword per lane.
template <bool WITH_CONTEXT>
__device__ __forceinline__ void stage_tables(hgdev::lds_u32 *tab, const uint32_t *pool, const HgPattern &p, uint32_t nw, uint32_t lane) {
  const uint4 *rsrc = reinterpret_cast<const uint4 *>(pool + p.reach_off);
  const uint4 r0 = rsrc[lane];
  tab[CT_REACH + 4 * lane + 0] = r0.x; tab[CT_REACH + 4 * lane + 1] = r0.y; tab[CT_REACH + 4 * lane + 2] = r0.z; tab[CT_REACH + 4 * lane + 3] = r0.w;
  if (nw == 2) {
    const uint4 r1 = rsrc[64 + lane];
    tab[CT_REACH + 256 + 4 * lane + 0] = r1.x; tab[CT_REACH + 256 + 4 * lane + 1] = r1.y; tab[CT_REACH + 256 + 4 * lane + 2] = r1.z; tab[CT_REACH + 256 + 4 * lane + 3] = r1.w;
  }
  const uint32_t nfollow = p.nnodes * nw;  // <= 128
  for (uint32_t i = lane; i < nfollow; i += 64) tab[CT_FOLLOW + i] = pool[p.follow_off + i];
  if (WITH_CONTEXT) {
    if (lane < nw) tab[CT_INIT + lane] = pool[p.init_off + lane];
    if (lane < 16 * nw) tab[CT_AMASK + lane] = pool[p.amask_off + lane];
    if (lane < 20 * nw) tab[CT_ACC + lane] = pool[p.acc_off + lane];
  }
}

template <int MODE>
__device__ __forceinline__ void confirm_tables_body(const HgConfirmArgs &a, uint32_t vblock, uint32_t vgrid, uint32_t *s_tab) {
  __shared__ uint32_t s_n, s_base;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  const uint64_t seg0 = static_cast<uint64_t>(blockIdx.x) * a.hit_seg_cap;
  const HitSink sink{a.tmp_hits + seg0, a.tmp_aux + seg0, a.hit_seg_cap, &s_n};
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  hgdev::lds_u32 *tab = (hgdev::lds_u32 *)(&s_tab[0]) + wave * CT_WORDS;
  // this block's slice [lo, hi) of the concatenation of the HG_DEFER_SHARDS lists (lane s holds list s)
  static_assert(HG_DEFER_SHARDS == 64, "one list per lane");
  uint32_t cnt = a.defer_count[MODE * HG_DEFER_SHARDS + lane];
  if (cnt > a.defer_shard_cap) cnt = a.defer_shard_cap;
  const uint32_t incl = wave_inclusive_scan(cnt, lane), start = incl - cnt;
  const uint32_t total = __shfl(incl, 63, 64);
  // no more blocks than give every wave a full batch of 64 (the others leave at once: an idle block is a resident block)
  const uint32_t busy = total ? (total + HG_CONFIRM_THREADS - 1) / HG_CONFIRM_THREADS : 0u;
  if (busy < vgrid) vgrid = busy;
  if (vblock >= vgrid) return;  // (block-uniform; nothing staged: nothing to flush)
  const uint32_t lo = static_cast<uint32_t>(static_cast<uint64_t>(total) * vblock / vgrid), hi = static_cast<uint32_t>(static_cast<uint64_t>(total) * (vblock + 1) / vgrid);
  const HgDeferred *lists = a.deferred + static_cast<uint64_t>(a.list_of_mode[MODE]) * HG_DEFER_SHARDS * a.defer_shard_cap;
#ifdef HG_PROFILE_CONFIRM
  uint64_t pf_load = 0, pf_stage = 0, pf_run = 0;
  uint32_t pf_pats = 0;
#endif
  for (uint32_t g0 = lo + wave * 64u; g0 < hi; g0 += HG_CONFIRM_THREADS) {  // wave-uniform
#ifdef HG_PROFILE_CONFIRM
    uint64_t pf_t = wall_clock64();
#endif
    const uint32_t g = g0 + lane;
    // list holding item g: the last list whose exclusive prefix is <= g
    uint32_t list = 0;
#pragma unroll
    for (uint32_t step = 32; step; step >>= 1) {
      const uint32_t probe = list + step;
      const uint32_t sp = __shfl(start, probe & 63u, 64);
      if (probe < 64u && sp <= g) list = probe;
    }
    const uint32_t l_start = __shfl(start, list, 64);
    const bool valid = g < hi;
    HgDeferred d{0, 0, 0};
    if (valid) d = lists[static_cast<uint64_t>(list) * a.defer_shard_cap + (g - l_start)];
    const uint32_t pattern = d.pattern & (HG_MAX_PATTERNS - 1u);
#ifdef HG_PROFILE_CONFIRM
    { const uint64_t n = wall_clock64(); pf_load += n - pf_t; pf_t = n; }
#endif
    for (uint64_t todo = __builtin_amdgcn_ballot_w64(valid); todo;) {  // one pattern at a time
      const uint32_t pat = __builtin_amdgcn_readlane(pattern, __builtin_ctzll(todo));
      const bool mine = valid && pattern == pat;
      todo &= ~__builtin_amdgcn_ballot_w64(mine);
      const HgPattern &p = a.db.patterns[pat];  // wave-uniform: scalar loads
      const uint32_t nw = MODE == 1 ? 1u : p.nw;
      stage_tables<MODE == 2>(tab, a.db.pool, p, nw, lane);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#ifdef HG_PROFILE_CONFIRM
      { const uint64_t n = wall_clock64(); pf_stage += n - pf_t; pf_t = n; pf_pats++; }
#endif
      if (mine) {
        const uint32_t id = p.id;
        auto emit = [&](uint64_t line_no, uint32_t to, uint64_t start_, uint32_t len) { sink.push(a, line_no, id, to, start_, len, pat, true); };
        const uint64_t fs = d.pos - (d.pattern >> 24);  // where the verified literal occurrence begins
        const uint32_t lead = p.lit_lead;
        if (MODE == 1) {
          hgdev::confirm_simple(a.text, a.nbytes, a.sums, a.bases, a.bs1, d.pos, d.rank, fs, lead, p.init_word, p.acc_all, tab + CT_REACH, tab + CT_FOLLOW, emit);
        } else if (nw == 1) {
          hgdev::confirm_ctx<1>(a.text, a.nbytes, a.sums, a.bases, a.bs1, d.pos, d.rank, fs, lead, tab + CT_REACH, tab + CT_FOLLOW, tab + CT_INIT, tab + CT_AMASK,
                                    tab + CT_ACC, emit);
        } else {
          hgdev::confirm_ctx<2>(a.text, a.nbytes, a.sums, a.bases, a.bs1, d.pos, d.rank, fs, lead, tab + CT_REACH, tab + CT_FOLLOW, tab + CT_INIT, tab + CT_AMASK,
                                    tab + CT_ACC, emit);
        }
      }
      __builtin_amdgcn_wave_barrier();  // the next pattern's tables overwrite this one's
#ifdef HG_PROFILE_CONFIRM
      { const uint64_t n = wall_clock64(); pf_run += n - pf_t; pf_t = n; }
#endif
    }
  }
#ifdef HG_PROFILE_CONFIRM
  if (lane == 0) a.tmp_hits[(6u << 20) + 32768u + blockIdx.x * (HG_CONFIRM_THREADS / 64) + wave] = HgHit{pf_load | (pf_stage << 32), static_cast<uint32_t>(pf_run), pf_pats | (MODE << 28)};
#endif
  flush_hits(a, &s_n, &s_base);
}

// Modes 0..2 in ONE launch (their items are few and each item is a chain of dependent loads: run back to back the three
// passes cost three latency tails, side by side one): blocks [k * blocks_per_mode, (k+1) * blocks_per_mode) work on the
// k-th mode present in the database.
#ifdef HG_CONFIRM_WAVES
__attribute__((amdgpu_waves_per_eu(HG_CONFIRM_WAVES, HG_CONFIRM_WAVES)))
#endif
__global__ __launch_bounds__(HG_CONFIRM_THREADS) void hg_confirm_fast_kernel(HgConfirmArgs a, uint32_t blocks_per_mode) {
  // Launch order = slowest routine first, mode by mode (per-wave timing, tools/confirm_waves.py: a batch of the routine with
  // boundary conditions takes 200-280 us, one of the context-free routine 70, a literal-only one 40; 9216 blocks are 3.6
  // rounds of resident blocks, and with the modes alternating the last long batch STARTED 214 us into a 405 us kernel):
  // the long batches all start in the first round and the short ones fill in behind them.
  uint32_t nmodes = 0;
  for (uint32_t m = 0; m < 3; m++) nmodes += a.mode_present[m] ? 1u : 0u;
  if (nmodes == 0) return;
  const uint32_t k = nmodes - 1u - blockIdx.x / blocks_per_mode, vblock = blockIdx.x % blocks_per_mode;  // block-uniform
  uint32_t mode = 0, seen = 0;
  for (uint32_t m = 0; m < 3; m++)
    if (a.mode_present[m] && seen++ == k) mode = m;
  __shared__ __attribute__((aligned(16))) uint32_t s_tab[(HG_CONFIRM_THREADS / 64) * CT_WORDS];  // a table area per wave (modes 1, 2)
#ifdef HG_PROFILE_CONFIRM
  const uint64_t t0 = wall_clock64();
#endif
  if (mode == 0) confirm_body<0>(a, vblock, blocks_per_mode);
  else if (mode == 1) confirm_tables_body<1>(a, vblock, blocks_per_mode, s_tab);
  else confirm_tables_body<2>(a, vblock, blocks_per_mode, s_tab);
#ifdef HG_PROFILE_CONFIRM
  if ((threadIdx.x & 63u) == 0) {  // per wave: start, end (100 MHz ticks), mode, block: into the unused end of the staging array
    const uint64_t t1 = wall_clock64();
    const uint32_t w = blockIdx.x * (HG_CONFIRM_THREADS / 64) + (threadIdx.x >> 6);
    a.tmp_hits[(6u << 20) + w] = HgHit{t0, static_cast<uint32_t>(t1 - t0), mode | (vblock << 4)};
  }
#endif
}
__global__ __launch_bounds__(256) void hg_confirm_generic_kernel(HgConfirmArgs a) { confirm_body<3>(a, blockIdx.x, gridDim.x); }
// Pattern sets whose anchored expressions are ALL literal-only (config 5's 4096 literals, a single keyword): the routine on its
// own, at a third of the registers of the three-routine kernel — the pass is a chain of dependent loads per occurrence, and
// resident waves are what hides them.
// (held at HG_LITERAL_WAVES waves per SIMD, above: fewer resident side waves leave the stream pass more of the memory pipeline)
__attribute__((amdgpu_waves_per_eu(HG_LITERAL_WAVES, HG_LITERAL_WAVES)))
__global__ __launch_bounds__(256) void hg_confirm_literal_kernel(HgConfirmArgs a) { confirm_body<0>(a, blockIdx.x, gridDim.x); }

// ------------------------------------------------------------------------------------------------
// Block mode (Face A, hs_scan): the whole buffer is ONE scan unit, newlines are ordinary bytes.
// Pass 1 marks the patterns whose required literal really occurs somewhere in the block; pass 2 runs each marked
// (or always-on) pattern's automaton over the whole block, one lane per pattern.
__global__ __launch_bounds__(256) void hg_block_mark_kernel(HgConfirmArgs a, uint32_t *pattern_flags) {
  const HgCand *cseg = a.cands + static_cast<uint64_t>(blockIdx.x) * a.cand_seg_cap;
  const uint32_t n = a.seg_count[blockIdx.x];
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    const HgCand c = cseg[i];
    hg_verify_window(a.db, a.text, a.nbytes, c.pos, c.word, [&](uint32_t pattern, uint64_t, uint32_t) { pattern_flags[pattern] = 1; });
  }
}

__global__ __launch_bounds__(256) void hg_block_scan_kernel(HgConfirmArgs a, const uint32_t *pattern_flags) {
  __shared__ uint32_t s_n, s_base;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  const uint64_t seg0 = static_cast<uint64_t>(blockIdx.x) * a.hit_seg_cap;
  const HitSink sink{a.tmp_hits + seg0, a.tmp_aux + seg0, a.hit_seg_cap, &s_n};
  for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < a.db.npatterns; p += gridDim.x * blockDim.x) {
    const HgPattern &pat = a.db.patterns[p];
    if (pat.tier == 0 && !pattern_flags[p]) continue;
    if (pat.nw > HG_MAX_W) continue;  // a huge automaton: hg_block_huge_kernel
    hg_nfa_scan(a.db.pool, pat, a.text, a.nbytes,
                [&](uint32_t to) { sink.push(a, 0, pat.id, to, 0, static_cast<uint32_t>(a.nbytes), p, pat.single != 0); });
  }
  flush_hits(a, &s_n, &s_base);
}

// ------------------------------------------------------------------------------------------------
// Block mode, short blocks (Face A on a line of text): ONE launch and no device-side bookkeeping.  The block's bytes sit in
// pinned host memory (the caller's copy of hs_scan's `data`); every workgroup stages them in LDS with coalesced 16-byte
// loads (one trip over the host link), then each lane runs ONE expression's automaton over the LDS copy (hg_nfa_scan, the
// reference routine) and writes its reports to the workgroup's segment of a pinned result array; the host applies the
// report rules (a handful of records).  Replaces copy + reset + stream + mark + scan + finalize + two copies for blocks of
// up to HG_BLOCK_SMALL_MAX bytes.
// Completion is signalled through pinned memory too: the last workgroup to finish writes the call's sequence number to
// *h_flag behind a system-scope fence, and the caller polls that word instead of sleeping in a stream synchronisation
// (an interrupt-driven wake-up costs more than the kernel).
// The automaton tables of a workgroup's expressions (consecutive in the pool: hg_compile.cpp lays an expression's reach /
// follow / init / amask / acc tables out back to back, expression after expression) are staged in LDS as well when they fit
// (`ppw` expressions per workgroup, 32 for sets of up to 2048 expressions): the per-byte step then is a chain of LDS reads
// instead of dependent global loads.
// A lone wave on an otherwise idle chip runs at idle clocks, and the automaton is a serial chain per byte (34 us per 100
// bytes with one lane per expression), so blocks of up to HG_BLOCK_SLICED_MAX bytes are also split over the lanes by START
// position: the step S' = (init | follow(S)) & reach[c] & mask is linear in (init, S), so the match ends of a block are the
// union over slices of the ends of matches that START in the slice.  Lane (expression, slice k) injects the start states only
// at the slice's bytes and runs on past its end until no state is alive (a literal's partial match dies within a few bytes;
// an expression with .* runs to the end of the block, as before).  The same end found from two slices is emitted once
// (a bitmap of ends per expression in LDS).
__global__ __launch_bounds__(256) void hg_block_small_kernel(HgDbView db, const uint8_t *h_text, uint32_t length, HgHit *h_out, uint32_t seg_cap, uint32_t *h_counts,
                                                             uint32_t *d_done, uint32_t *h_flag, uint32_t seq, uint32_t ppw) {
  __shared__ __attribute__((aligned(16))) uint8_t s_text[HG_BLOCK_SMALL_MAX + 16];
  __shared__ __attribute__((aligned(16))) uint32_t s_pool[HG_BLOCK_SMALL_POOL];
  __shared__ uint32_t s_seen[32 * HG_BLOCK_SLICED_WORDS];  // sliced mode: ends already emitted, per expression of the workgroup
  __shared__ uint32_t s_n;
  if (threadIdx.x == 0) s_n = 0;
  const uint32_t chunks = (length + 15u) >> 4;  // (the pinned buffer is readable up to its size rounded up to 16)
  for (uint32_t i = threadIdx.x; i < chunks; i += blockDim.x) reinterpret_cast<uint4 *>(s_text)[i] = reinterpret_cast<const uint4 *>(h_text)[i];
  const uint32_t first = blockIdx.x * ppw, last = min(first + ppw, db.npatterns);  // (first < npatterns: the grid is sized so)
  const uint32_t npat = last - first;
  const HgPattern &pl = db.patterns[last - 1];
  const uint32_t lo = db.patterns[first].reach_off, hi = pl.acc_off + 20u * pl.nw;
  const bool staged = hi - lo <= HG_BLOCK_SMALL_POOL;  // (uniform)
  if (staged)
    for (uint32_t i = threadIdx.x; i < hi - lo; i += blockDim.x) s_pool[i] = db.pool[lo + i];
  // slices of at least 8 start positions, as many as the lanes allow
  const bool sliced = ppw <= 32 && length <= HG_BLOCK_SLICED_MAX;
  uint32_t slice_len = length, nslices = 1;
  if (sliced) {
    const uint32_t most = blockDim.x / npat;
    slice_len = max(8u, (length + most - 1) / most);
    nslices = (length + slice_len - 1) / slice_len;
    const uint32_t words = (length >> 5) + 1;  // ends 0 .. length
    for (uint32_t i = threadIdx.x; i < npat * HG_BLOCK_SLICED_WORDS; i += blockDim.x)
      if ((i & (HG_BLOCK_SLICED_WORDS - 1)) < words) s_seen[i] = 0;
  }
  __syncthreads();
  const uint32_t j = threadIdx.x % npat, k = threadIdx.x / npat;  // expression of the workgroup, slice
  if (k < nslices) {
    const HgPattern pat = db.patterns[first + j];
    const uint32_t single = pat.single ? HG_HIT_SINGLE_BIT : 0u;
    HgHit *seg = h_out + static_cast<uint64_t>(blockIdx.x) * seg_cap;
    auto emit = [&](uint32_t to) {
      if (sliced && (atomicOr(&s_seen[j * HG_BLOCK_SLICED_WORDS + (to >> 5)], 1u << (to & 31)) >> (to & 31) & 1)) return;
      const uint32_t slot = atomicAdd(&s_n, 1u);
      if (slot < seg_cap) seg[slot] = HgHit{0, pat.id, to | single};
    };
    const uint32_t from = k * slice_len, upto = min(from + slice_len, length);
    if (staged) {
      HgPattern q = pat;  // the same tables, relative to the LDS copy
      q.reach_off -= lo, q.follow_off -= lo, q.init_off -= lo, q.amask_off -= lo, q.acc_off -= lo;
      hg_nfa_scan_slice(s_pool, q, s_text, length, from, upto, emit);
    } else {
      hg_nfa_scan_slice(db.pool, pat, s_text, length, from, upto, emit);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    h_counts[blockIdx.x] = s_n;  // (more than seg_cap: the caller takes the general path)
    __threadfence_system();       // this workgroup's reports and count are visible to the host ...
    if (atomicAdd(d_done, 1u) == gridDim.x - 1) {  // ... and it was the last one
      *d_done = 0;
      __threadfence_system();
      __hip_atomic_store(h_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// One launch puts the scanner's device state in place for a pass (it used to be half a dozen memsets and a host-to-device
// copy, each a launch of its own in front of the first stream kernel): counters, tile cursors, finalize totals and flags
// zeroed, the tile-scan state set to (carry-in line start, first line number: 0 and the caller's base for a whole
// buffer, the state at the segment's first tile for a segment of one), bucket fill levels and the
// verified-occurrence counts zeroed.
__global__ void hg_reset_kernel(uint32_t *state, uint32_t state_words, HgTileBase *final_state, uint64_t carry_start, uint64_t first_piece, uint32_t *fill, uint32_t nb,
                                uint32_t *defer_count, uint32_t ndefer) {
  const uint32_t i0 = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
  for (uint32_t i = i0; i < state_words; i += stride) state[i] = 0;
  for (uint32_t i = i0; i < HG_CNT_CURSORS; i += stride) state[HG_CNT_CURSOR0 + i] = 0;  // every pipeline chunk's tile cursor
  for (uint32_t i = i0; i < nb; i += stride) fill[i] = 0;
  for (uint32_t i = i0; i < ndefer; i += stride) defer_count[i] = 0;
  if (i0 == 0) *final_state = HgTileBase{carry_start, first_piece};
}

// ------------------------------------------------------------------------------------------------
// Finalize, bucketed (the default): order by (line, id, to), SINGLEMATCH / duplicate rules, compaction.
// The confirm passes put every hit straight into the bucket of its line's start (HitSink::push): a line's start position
// grows with its number, so the buckets are in final order among themselves and all reports of one line sit in ONE bucket,
// whichever pipeline chunk found them.  Per range of buckets (a chunk of the pipeline, as soon as its side passes are
// done): sort each bucket (a few dozen records: one wave, bitonic network over the lanes, no LDS; larger ones in LDS),
// apply the report rules on the sorted keys, scan the kept counts, gather the kept records to their final places.  Only
// the last chunk's buckets are finalized after the last scan kernel; no library sort, no host round trip in between.
// A bucket larger than HG_FIN_BUCKET_CAP (thousands of reports in one bucket: every match end of an all-matches expression
// on a long line) raises a flag and the engine repeats the pass with the compact array + library sort (hg_key/keep/scatter).
// key = line start inside the bucket | id | to | single (field widths id_bits / to_bits): the order of (line, id, to)
__device__ __forceinline__ uint64_t fin_key(const HgHit &h, uint32_t id_bits, uint32_t to_bits) {
  HgHit c = h;
  c.line_no = h.line_no >> HG_HIT_REL_SHIFT;
  c.to &= ~HG_HIT_SINGLE_BIT;
  return hg_sort_key_packed(c, (h.to & HG_HIT_SINGLE_BIT) != 0, id_bits, to_bits);
}
// Report rules on sorted packed keys (hg_keep_hit_at restated): equal (line, id, to) <=> keys equal above bit 0; a
// SINGLEMATCH report (bit 0) is kept only if it is the first one of its (line, id) = key >> group_shift.
// Buckets of up to 64 reports (nearly all of them): one wave, one report per lane, the rules as bit operations on ballots.
__global__ __launch_bounds__(256) void hg_fin_sort_small_kernel(const HgHit *hits, uint32_t *idx, const uint32_t *fill, uint32_t b_lo, uint32_t b_hi, uint32_t cap,
                                                                uint32_t id_bits, uint32_t to_bits, uint32_t *kept_count, uint32_t *big_list, uint32_t *big_count,
                                                                uint32_t big_stride) {
  const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, waves = (gridDim.x * blockDim.x) >> 6;
  const uint32_t group_shift = to_bits + 1;
  // A bucket's fill level and its first 64 records are loaded TOGETHER (lanes past the fill level read records of the same
  // region that are simply not used), and one bucket ahead: the loads of bucket b + waves are in flight while bucket b is
  // ordered.  Per bucket the wave then pays the sorting network, not two memory round trips.
  // (the record travels as the four dwords it is: a struct selected by a condition went through scratch memory)
  const uint4 *hits4 = reinterpret_cast<const uint4 *>(hits);
  static_assert(sizeof(HgHit) == 16, "one 16-byte load per record");
  uint32_t b = b_lo + wave;
  uint32_t n_next = b < b_hi ? fill[b] : 0u;
  uint4 h_next = make_uint4(0, 0, 0, 0);
  if (b < b_hi && lane < cap) h_next = hits4[static_cast<uint64_t>(b) * cap + lane];
  for (; b < b_hi; b += waves) {
    uint32_t n = n_next;
    const uint4 hv = h_next;
    const uint32_t bn = b + waves;
    n_next = bn < b_hi ? fill[bn] : 0u;
    h_next = make_uint4(0, 0, 0, 0);
    if (bn < b_hi && lane < cap) h_next = hits4[static_cast<uint64_t>(bn) * cap + lane];
    if (n > cap) n = cap;  // (overflowed: the pass is repeated anyway)
    if (n > 64) {  // hg_fin_sort_big_kernel's: noted in the work list of its size class (up to HG_FIN_MEDIUM_CAP records / more)
      if (lane == 0) {
        const uint32_t cls = n <= HG_FIN_MEDIUM_CAP ? 0u : 1u;
        big_list[cls * big_stride + atomicAdd(big_count + cls, 1u)] = b;  // (big_stride: the bucket arrays' allocated length)
      }
      continue;
    }
    if (n == 0) {
      if (lane == 0) kept_count[b] = 0;
      continue;
    }
    const uint32_t b0 = b * cap;
    const HgHit h{(static_cast<uint64_t>(hv.y) << 32) | hv.x, hv.z, hv.w};
    uint64_t k = lane < n ? fin_key(h, id_bits, to_bits) : ~0ull;  // (padding sorts last; a real key never has all bits set)
    uint32_t x = b0 + lane;
    if (n > 1) {
      // Order by RANK: every lane counts the keys below its own (equal keys: the lower lane first) against the n keys
      // broadcast one after the other (v_readlane, n is wave-uniform), then one permute puts key and index in their places.
      // ~8 instructions per key and no LDS traffic; the bitonic network over the lanes this replaces was 21 stages of three
      // cross-lane moves each, ~2500 instructions per bucket, and the kernel was bound by exactly that.
      const uint32_t klo = static_cast<uint32_t>(k), khi = static_cast<uint32_t>(k >> 32);
      const uint32_t nu = __builtin_amdgcn_readfirstlane(n);
      uint32_t rank = 0;
      for (uint32_t i = 0; i < nu; i++) {
        const uint64_t ki = (static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readlane(khi, i))) << 32) | static_cast<uint32_t>(__builtin_amdgcn_readlane(klo, i));
        rank += (ki < k || (ki == k && i < lane)) ? 1u : 0u;
      }
      if (lane >= nu) rank = lane;  // (padding stays where it is: ranks are a permutation of the lanes)
      const uint32_t slo = __builtin_amdgcn_ds_permute(rank << 2, klo), shi = __builtin_amdgcn_ds_permute(rank << 2, khi);
      x = __builtin_amdgcn_ds_permute(rank << 2, x);
      k = (static_cast<uint64_t>(shi) << 32) | slo;
    }
    const uint32_t plo = __shfl_up(static_cast<uint32_t>(k), 1, 64), phi = __shfl_up(static_cast<uint32_t>(k >> 32), 1, 64);
    const uint64_t prev = (static_cast<uint64_t>(phi) << 32) | plo;
    const bool valid = lane < n;
    const bool dup = valid && lane > 0 && (prev >> 1) == (k >> 1);
    const bool head = valid && (lane == 0 || (prev >> group_shift) != (k >> group_shift));
    const uint64_t head_mask = __builtin_amdgcn_ballot_w64(head), single_mask = __builtin_amdgcn_ballot_w64(valid && (k & 1u));
    const uint64_t upto = lane == 63 ? ~0ull : ((2ull << lane) - 1ull), below = (1ull << lane) - 1ull;
    const uint32_t group_start = 63u - static_cast<uint32_t>(__builtin_clzll((head_mask & upto) | 1ull));
    const bool earlier_single = (single_mask & below & ~((1ull << group_start) - 1ull)) != 0;
    const bool keep = valid && !dup && !((k & 1u) && earlier_single);
    const uint64_t km = __builtin_amdgcn_ballot_w64(keep);
    if (keep) idx[b0 + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(km >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(km), 0u))] = x;
    if (lane == 0) kept_count[b] = static_cast<uint32_t>(__popcll(km));
  }
}
// Larger buckets (hits clustered on few lines): one block sorts the bucket.  Two size classes, each with its own work list
// and launch: nearly all larger buckets hold a few hundred records at most (LCAP = HG_FIN_MEDIUM_CAP, sorted in 6 KiB of
// LDS); buckets of thousands of records are sorted in a scratch area in HBM (volatile accesses between block barriers: slow,
// and rare) — with 48 KiB of LDS per block the launch waited 370-450 us for room next to the stream pass even when its
// work list was empty.
template <uint32_t LCAP, bool IN_LDS>
__global__ __launch_bounds__(256) void hg_fin_sort_big_kernel(const HgHit *hits, uint32_t *idx, const uint32_t *fill, const uint32_t *big_list, const uint32_t *big_count,
                                                              uint32_t cap, uint32_t id_bits, uint32_t to_bits, uint32_t *kept_count, uint32_t *overflow, uint64_t *scratch_key,
                                                              uint32_t *scratch_idx) {
  __shared__ uint64_t l_key[IN_LDS ? LCAP : 1];
  __shared__ uint32_t l_idx[IN_LDS ? LCAP : 1];
  volatile uint64_t *s_key = IN_LDS ? l_key : scratch_key + static_cast<uint64_t>(blockIdx.x) * LCAP;
  volatile uint32_t *s_idx = IN_LDS ? l_idx : scratch_idx + static_cast<uint64_t>(blockIdx.x) * LCAP;
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t group_shift = to_bits + 1;  // key >> group_shift == (line, id)
  const uint32_t nbig = *big_count;
  for (uint32_t q = blockIdx.x; q < nbig; q += gridDim.x) {
    const uint32_t b = big_list[q];
    uint32_t n = fill[b];
    if (n > cap) n = cap;
    if (n > LCAP) {  // (only the large class can meet this: beyond what one block sorts)
      if (tid == 0) {
        kept_count[b] = 0;
        atomicMax(overflow, n);
      }
      continue;
    }
    const uint32_t b0 = b * cap;
    uint32_t p2 = 128;
    while (p2 < n) p2 <<= 1;
    for (uint32_t i = tid; i < p2; i += 256) {
      s_key[i] = i < n ? fin_key(hits[b0 + i], id_bits, to_bits) : ~0ull;
      s_idx[i] = b0 + i;
    }
    __syncthreads();
    for (uint32_t k = 2; k <= p2; k <<= 1)
      for (uint32_t j = k >> 1; j > 0; j >>= 1) {
        for (uint32_t t = tid; t < (p2 >> 1); t += 256) {
          const uint32_t lo = 2 * t - (t & (j - 1)), hi = lo + j;  // partner pairs at distance j
          const bool up = (lo & k) == 0;
          const uint64_t a = s_key[lo], c = s_key[hi];
          if ((a > c) == up) {
            s_key[lo] = c;
            s_key[hi] = a;
            const uint32_t x = s_idx[lo];
            s_idx[lo] = s_idx[hi];
            s_idx[hi] = x;
          }
        }
        __syncthreads();
      }
    // rules + compaction, 64 sorted records at a time by the first wave (the other waves wait: a large bucket is rare)
    if (tid < 64) {
      uint32_t kept = 0;
      for (uint32_t i0 = 0; i0 < n; i0 += 64) {
        const uint32_t i = i0 + lane;
        bool keep = false;
        if (i < n) {
          const uint64_t k = s_key[i];
          keep = !(i > 0 && (s_key[i - 1] >> 1) == (k >> 1));  // the same report is already there (either kind)
          if (keep && (k & 1u)) {  // only the first SINGLEMATCH report of this (line, id)
            for (uint32_t j = i; j > 0; j--) {
              const uint64_t q = s_key[j - 1];
              if ((q >> group_shift) != (k >> group_shift)) break;
              if (q & 1u) {
                keep = false;
                break;
              }
            }
          }
        }
        const uint64_t km = __builtin_amdgcn_ballot_w64(keep);
        if (keep) idx[b0 + kept + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(km >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(km), 0u))] = s_idx[i];
        kept += static_cast<uint32_t>(__popcll(km));
      }
      if (lane == 0) kept_count[b] = kept;
    }
    __syncthreads();
  }
}
template __global__ void hg_fin_sort_big_kernel<HG_FIN_MEDIUM_CAP, true>(const HgHit *, uint32_t *, const uint32_t *, const uint32_t *, const uint32_t *, uint32_t, uint32_t, uint32_t,
                                                                          uint32_t *, uint32_t *, uint64_t *, uint32_t *);
template __global__ void hg_fin_sort_big_kernel<HG_FIN_BUCKET_CAP, false>(const HgHit *, uint32_t *, const uint32_t *, const uint32_t *, const uint32_t *, uint32_t, uint32_t, uint32_t,
                                                                           uint32_t *, uint32_t *, uint64_t *, uint32_t *);
// kept_count[b_lo, b_hi) -> exclusive positions in the compact output, continuing from *total (the kept records of the bucket
// ranges finalized before); *total moves on.  total[0] += kept records of the range, total[1] += raw records of the range (fill
// levels, a bucket holds at most cap).
// A grid of blocks, each with a contiguous span of the range (round 2: ONE block walked the whole range — 440 us for config 5's
// million buckets, twice per pass): a block sums its span, publishes the sums (part[] + a flag holding this launch's epoch),
// adds up the sums of the blocks before it as they appear, and rescans its span into positions.  A block only ever waits for
// blocks with LOWER indices, which were dispatched before it: no deadlock whatever shares the chip.
// part: 3 * gridDim.x words {kept sum, raw sum, flag}; epoch: a value no earlier launch on these words has used.
template <uint32_t HG_FIN_SCAN_THREADS>
__global__ __launch_bounds__(HG_FIN_SCAN_THREADS) void hg_fin_scan_kernel(uint32_t *kept_count, uint32_t b_lo, uint32_t b_hi, uint32_t *total, const uint32_t *fill,
                                                                          uint32_t cap, uint32_t *part, uint32_t epoch) {
  __shared__ uint32_t s_wave[HG_FIN_SCAN_THREADS / 64], s_raw[HG_FIN_SCAN_THREADS / 64], s_base[2];
  constexpr uint32_t WAVES = HG_FIN_SCAN_THREADS / 64;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t carry = total[0], carry_raw = total[1];
  // this block's span (a multiple of 64 buckets per wave), wave w a contiguous share of it
  const uint32_t nb = b_hi - b_lo;
  const uint32_t span = (((nb + gridDim.x - 1) / gridDim.x + WAVES * 64u - 1) / (WAVES * 64u)) * (WAVES * 64u);
  const uint32_t s_lo = b_lo + (blockIdx.x * span < nb ? blockIdx.x * span : nb);
  const uint32_t s_hi = s_lo + span < b_hi ? s_lo + span : b_hi;
  const uint32_t share = span / WAVES;
  const uint32_t w_lo = s_lo + wave * share < s_hi ? s_lo + wave * share : s_hi, w_hi = w_lo + share < s_hi ? w_lo + share : s_hi;
  uint32_t sum = 0, raw = 0;
#pragma unroll 4
  for (uint32_t i = w_lo + lane; i < w_hi; i += 64) {
    sum += kept_count[i];
    const uint32_t f = fill[i];
    raw += f < cap ? f : cap;
  }
  sum = wave_inclusive_scan(sum, lane);
  raw = wave_inclusive_scan(raw, lane);
  if (lane == 63) {
    s_wave[wave] = sum;
    s_raw[wave] = raw;
  }
  __syncthreads();  // (also: every thread has read total[] before anyone can rewrite it — the last block does, after all flags are up)
  uint32_t run = 0, all = 0, all_raw = 0;
#pragma unroll
  for (uint32_t w = 0; w < WAVES; w++) {
    const uint32_t t = s_wave[w];
    run += w < wave ? t : 0u;
    all += t;
    all_raw += s_raw[w];
  }
  if (threadIdx.x == 0) {
    part[3 * blockIdx.x] = all;
    part[3 * blockIdx.x + 1] = all_raw;
    __threadfence();
    __hip_atomic_store(&part[3 * blockIdx.x + 2], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (wave == 0) {  // the sums of the blocks before this one
    uint32_t before = 0, before_raw = 0;
    for (uint32_t p = lane; p < blockIdx.x; p += 64) {
      while (__hip_atomic_load(&part[3 * p + 2], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != epoch) __builtin_amdgcn_s_sleep(2);
      before += __hip_atomic_load(&part[3 * p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      before_raw += __hip_atomic_load(&part[3 * p + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    before = wave_inclusive_scan(before, lane);
    before_raw = wave_inclusive_scan(before_raw, lane);
    if (lane == 63) {
      s_base[0] = before;
      s_base[1] = before_raw;
    }
  }
  __syncthreads();
  run += carry + s_base[0];
  // ... then the exclusive positions (the entries come from the cache this time)
  for (uint32_t i0 = w_lo; i0 < w_hi; i0 += 64) {
    const uint32_t i = i0 + lane;
    const uint32_t c = i < w_hi ? kept_count[i] : 0u;
    const uint32_t incl = wave_inclusive_scan(c, lane);
    if (i < w_hi) kept_count[i] = run + incl - c;
    run += __builtin_amdgcn_readlane(incl, 63);
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {  // (every block has published, hence read total[], by now)
    total[0] = carry + s_base[0] + all;
    total[1] = carry_raw + s_base[1] + all_raw;
  }
}
template __global__ void hg_fin_scan_kernel<512u>(uint32_t *, uint32_t, uint32_t, uint32_t *, const uint32_t *, uint32_t, uint32_t *, uint32_t);
template __global__ void hg_fin_scan_kernel<1024u>(uint32_t *, uint32_t, uint32_t, uint32_t *, const uint32_t *, uint32_t, uint32_t *, uint32_t);
// kept records of bucket b (idx[b * cap ...] in final order) -> out[kept_base[b] ...]; bucket b + 1's base (or *total for the
// last bucket of the range) ends the run.  One wave per bucket, grid-stride.
__global__ void hg_fin_gather_kernel(const HgHit *hits, const HgHitAux *aux, const uint32_t *idx, const uint32_t *kept_base, const uint32_t *total, uint32_t b_lo,
                                     uint32_t b_hi, uint32_t cap, HgHit *oh, HgHitAux *oa) {
  const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, waves = (gridDim.x * blockDim.x) >> 6;
  // (one bucket ahead, like the sort: positions and the first 64 record indices of bucket b + waves are in flight while
  // bucket b's records move)
  auto base_of = [&](uint32_t b) { return b < b_hi ? kept_base[b] : *total; };
  uint32_t b = b_lo + wave;
  uint32_t k0_next = b < b_hi ? kept_base[b] : 0u, k1_next = b < b_hi ? base_of(b + 1) : 0u;
  uint32_t src_next = (b < b_hi && lane < cap) ? idx[b * cap + lane] : 0u;
  for (; b < b_hi; b += waves) {
    const uint32_t k0 = k0_next, n = k1_next - k0, b0 = b * cap;
    const uint32_t src0 = src_next;
    const uint32_t bn = b + waves;
    k0_next = bn < b_hi ? kept_base[bn] : 0u;
    k1_next = bn < b_hi ? base_of(bn + 1) : 0u;
    src_next = (bn < b_hi && lane < cap) ? idx[bn * cap + lane] : 0u;
    for (uint32_t i = lane; i < n; i += 64) {
      const uint32_t src = i < 64 ? src0 : idx[b0 + i];
      HgHit h = hits[src];
      h.to &= ~HG_HIT_SINGLE_BIT;
      h.line_no &= (1ull << HG_HIT_REL_SHIFT) - 1ull;
      oh[k0 + i] = h;
      oa[k0 + i] = aux[src];
    }
  }
}

// ------------------------------------------------------------------------------------------------
__global__ void hg_key_kernel(const HgHit *hits, const HgHitAux *aux, const HgPattern *patterns, uint32_t n, uint64_t *key, uint32_t *idx) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  key[i] = hg_sort_key(hits[i], patterns[aux[i].pattern].single);
  idx[i] = i;
}
__global__ void hg_line_key_kernel(const HgHit *hits, const uint32_t *perm, uint32_t n, uint64_t *key) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) key[i] = hits[perm[i]].line_no;
}
// indices of a later run are positions in the whole raw array
__global__ void hg_offset_kernel(uint32_t *idx, uint32_t n, uint32_t add) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) idx[i] += add;
}
__global__ void hg_key_packed_kernel(const HgHit *hits, const HgHitAux *aux, const HgPattern *patterns, uint32_t n, uint32_t id_bits, uint32_t to_bits,
                                     uint64_t *key, uint32_t *idx) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  key[i] = hg_sort_key_packed(hits[i], patterns[aux[i].pattern].single, id_bits, to_bits);
  idx[i] = i;
}
// SINGLEMATCH / duplicate rules on the sorted order, read through the permutation (no sorted copy of the records)
__global__ void hg_keep_kernel(const HgHit *hits, const HgHitAux *aux, const uint32_t *perm, const HgPattern *patterns, uint32_t n, uint8_t *keep) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  keep[i] = hg_keep_hit_at([&](size_t j) { return hits[perm[j]]; }, [&](size_t j) { return patterns[aux[perm[j]].pattern].single != 0; }, i) ? 1 : 0;
}
// pos = exclusive prefix sum of keep: the kept records go to their final places; the last thread leaves the count
__global__ void hg_scatter_kernel(const HgHit *hits, const HgHitAux *aux, const uint32_t *perm, const uint8_t *keep, const uint32_t *pos, uint32_t n,
                                  HgHit *oh, HgHitAux *oa, uint32_t *count) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (keep[i]) {
    oh[pos[i]] = hits[perm[i]];
    oa[pos[i]] = aux[perm[i]];
  }
  if (i == n - 1) *count = pos[i] + keep[i];
}
