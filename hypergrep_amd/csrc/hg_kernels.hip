// gfx950 kernels of the line-scan path.  Replaces the reference's hot loop
// (hypergrep/lib/c/hyperscanner.c:198-226: gzgets -> strlen -> hs_scan -> hs_callback per line) with
//
//   hg_stream_kernel      one pass over the text in HBM: 16 B per lane coalesced loads, per-dword window
//                         fingerprint (v_dot4_u32_u8) probed in an LDS cuckoo filter, exact newline counts
//                         per 16 KiB wave tile, window hits ranked (wave prefix sums) and appended to HBM
//                         with one atomic per wave
//   hg_tile_*             3-launch scan of the tile newline summaries -> global piece numbers
//   hg_confirm_kernel     one lane per window hit: literal verify, locate the line piece, run the pattern automaton
//   hg_always_on_kernel   patterns without a long enough required literal: every line, one wave per tile
//   hg_key/gather/keep    ordering + SINGLEMATCH / duplicate rules (sort itself: rocPRIM radix sort)
//
// Byte/integer work, HBM-bound: no MFMA anywhere.  Wave64 only.
#include <hip/hip_runtime.h>
#include <cstdlib>

#include "hg_confirm_dev.h"
#include "hg_core.h"
#include "hg_engine.h"
#include "hg_post.h"

namespace {

constexpr int WG_WAVES = 8;
constexpr int WG_THREADS = WG_WAVES * 64;
constexpr int ITERS = HG_TILE_BYTES / 1024;  // 1 KiB per wave-iteration

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
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
// Stream pass.  One wave owns one 16 KiB tile at a time; tiles are dealt round-robin over all resident
// waves so that at any moment the chip streams one contiguous window of the text.
//
// Per 16 bytes of text a lane spends: 1 coalesced 16 B load, 4 x (4 ops: exact newline count) and
// 4 x (fold, 3 x v_dot4_u32_u8, 2 LDS u16 reads, 2 compares) for the window filter.  Ranks and the append
// to the candidate buffer run only in iterations where some lane's fingerprint matched (~1e-5 per dword
// plus the real occurrences), so the steady state is pure streaming.
namespace {

// "not a newline" bits: bit 7 of each byte is CLEAR iff that byte is '\n' (exact, no carries between bytes)
__device__ __forceinline__ uint32_t not_newline_bits(uint32_t w) {
  uint32_t b = ((w & 0x7f7f7f7fu) ^ 0x0a0a0a0au) + 0x7f7f7f7fu;  // bit 7 set iff the low 7 bits differ from 0x0a
  return b | w | 0x7f7f7f7fu;                                     // ... or the byte's own bit 7 is set
}

// Rare path of one iteration: newline rank of every matching dword, then one LDS atomic per wave reserves
// slots in the workgroup's private segment of the candidate buffer (a single global counter would cap the whole
// kernel at ~90 atomics/us: measured 4.2 ms per 4 GiB for 0.36 M appends).
__device__ __noinline__ void append_matches(HgCand *__restrict__ seg, uint32_t seg_cap, uint32_t *lds_count, uint64_t chunk_pos, uint32_t lane,
                                            uint4 cur, uint32_t tot, uint32_t hits) {
  const uint32_t words[4] = {cur.x, cur.y, cur.z, cur.w};
  uint32_t cnt[4];
#pragma unroll
  for (int k = 0; k < 4; k++) cnt[k] = __popc(~not_newline_bits(words[k]));
  const uint32_t c = cnt[0] + cnt[1] + cnt[2] + cnt[3];
  uint32_t rank = wave_sum(tot) + wave_inclusive_scan(c, lane) - c;  // newlines in [tile start, this lane's chunk)
  const uint32_t mine = __popc(hits);
  const uint32_t incl = wave_inclusive_scan(mine, lane);
  const uint32_t total = __shfl(incl, 63, 64);
  uint32_t base = 0;
  if (lane == 0) base = atomicAdd(lds_count, total);
  uint32_t slot = __shfl(base, 0, 64) + incl - mine;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if ((hits >> k) & 1u) {
      if (slot < seg_cap) seg[slot] = HgCand{chunk_pos + 4u * k, words[k], rank};
      slot++;
    }
    rank += cnt[k];
  }
}

template <int LOG2, bool WIDE>
struct Probe {
  static constexpr uint32_t BYTE_MASK = ((1u << LOG2) - 1u) << 2;
  // First level for the lane's four dwords: a slot matches if it holds the window's hash C.
  // ANY_ONLY: non-zero iff any of the four windows matched (hot path); else per-window, per-slot match bits.
  template <bool ANY_ONLY>
  __device__ __forceinline__ static uint32_t probe4(const uint32_t *filter, uint32_t fold, uint32_t wa, uint32_t wb, uint4 v) {
    const uint32_t f0 = v.x | fold, f1 = v.y | fold, f2 = v.z | fold, f3 = v.w | fold;
    // the hashes of the four windows first (independent v_dot4), then the eight LDS reads
    const uint32_t a0 = hg_dot4(f0, wa), a1 = hg_dot4(f1, wa), a2 = hg_dot4(f2, wa), a3 = hg_dot4(f3, wa);
    const uint32_t b0 = hg_dot4(f0, wb), b1 = hg_dot4(f1, wb), b2 = hg_dot4(f2, wb), b3 = hg_dot4(f3, wb);
    const uint32_t c0 = hg_dot4(f0, HG_HASH_WEIGHTS), c1 = hg_dot4(f1, HG_HASH_WEIGHTS);
    const uint32_t c2 = hg_dot4(f2, HG_HASH_WEIGHTS), c3 = hg_dot4(f3, HG_HASH_WEIGHTS);
    const uint8_t *base = reinterpret_cast<const uint8_t *>(filter);
    auto at = [&](uint32_t h) { return *reinterpret_cast<const uint32_t *>(base + (h & BYTE_MASK)); };
    if (WIDE) {  // slots mix both sums (hg_slot_wide)
      auto wide = [&](uint32_t x, uint32_t y) { return *reinterpret_cast<const uint32_t *>(base + hg_slot_wide(x, y, BYTE_MASK)); };
      const uint32_t ta0 = wide(a0, b0), tb0 = wide(b0, a0), ta1 = wide(a1, b1), tb1 = wide(b1, a1);
      const uint32_t ta2 = wide(a2, b2), tb2 = wide(b2, a2), ta3 = wide(a3, b3), tb3 = wide(b3, a3);  // two 16-bit fingerprints per slot; no per-slot detail needed (no second level in wide mode)
      auto m = [](uint32_t t, uint32_t c) { return static_cast<uint16_t>(t) == static_cast<uint16_t>(c) || static_cast<uint16_t>(t >> 16) == static_cast<uint16_t>(c); };
      const bool m0 = m(ta0, c0) || m(tb0, c0), m1 = m(ta1, c1) || m(tb1, c1), m2 = m(ta2, c2) || m(tb2, c2), m3 = m(ta3, c3) || m(tb3, c3);
      if (ANY_ONLY) return (m0 || m1 || m2 || m3) ? 1u : 0u;
      return (m0 ? 1u : 0u) | (m1 ? 2u : 0u) | (m2 ? 4u : 0u) | (m3 ? 8u : 0u);
    }
    const uint32_t ta0 = at(a0), tb0 = at(b0), ta1 = at(a1), tb1 = at(b1), ta2 = at(a2), tb2 = at(b2), ta3 = at(a3), tb3 = at(b3);
    if (ANY_ONLY) return (ta0 == c0 || tb0 == c0 || ta1 == c1 || tb1 == c1 || ta2 == c2 || tb2 == c2 || ta3 == c3 || tb3 == c3) ? 1u : 0u;
    // bits 0..3: slot A of window k matched; bits 4..7: slot B
    return (ta0 == c0 ? 1u : 0u) | (ta1 == c1 ? 2u : 0u) | (ta2 == c2 ? 4u : 0u) | (ta3 == c3 ? 8u : 0u) | (tb0 == c0 ? 16u : 0u) |
           (tb1 == c1 ? 32u : 0u) | (tb2 == c2 ? 64u : 0u) | (tb3 == c3 ? 128u : 0u);
  }
};

// Second level, entered when some lane's first level matched: the 4 bytes before and after the window must
// agree (byte-masked) with what the matching slot's literals have there.  `l1`: bits 0..3 = slot A of window k
// matched, bits 4..7 = slot B.  Returns bit k set iff window k survives.
template <int LOG2>
__device__ __forceinline__ uint32_t level2_filter(const HgFilterExt *ext, uint32_t fold, uint32_t wa, uint32_t wb, uint4 v, uint32_t left,
                                                  uint32_t right, uint32_t lane, uint32_t l1) {
  constexpr uint32_t BYTE_MASK = ((1u << LOG2) - 1u) << 2;
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
  uint32_t out = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const bool ha = (l1 >> k) & 1u, hb = (l1 >> (k + 4)) & 1u;
    if (ha || hb) {
      const uint32_t f = w[k] | fold;
      // both conditions are fetched at once (one LDS round trip); a slot that did not match counts as failed
      const HgFilterExt ea = ext[(hg_dot4(f, wa) & BYTE_MASK) >> 2], eb = ext[(hg_dot4(f, wb) & BYTE_MASK) >> 2];
      const uint32_t prev = (k == 0 ? left : w[k - 1]) | fold;
      const uint32_t next_dword = k == 3 ? right : w[k + 1];
      const uint32_t next = (HG_WINDOW_BYTES == 4 ? next_dword : ((w[k] >> 24) | (next_dword << 8))) | fold;
      // the first lane has no left neighbour and the last lane no right neighbour inside this 1 KiB segment
      const uint32_t pmask = (k == 0 && lane == 0) ? 0u : 0xFFFFFFFFu;
      const uint32_t nmask = (k == 3 && lane == 63) ? (HG_WINDOW_BYTES == 4 ? 0u : 0xFFu) : 0xFFFFFFFFu;
      const bool oka = ha && ((((prev ^ ea.pv) & ea.pm & pmask) | ((next ^ ea.nv) & ea.nm & nmask)) == 0);
      const bool okb = hb && ((((prev ^ eb.pv) & eb.pm & pmask) | ((next ^ eb.nv) & eb.nm & nmask)) == 0);
      if (oka || okb) out |= 1u << k;
    }
  }
  return out;
}

// One tile.  FULL: the tile lies entirely inside the text (no bounds checks on the hot path).
template <int LOG2, bool WIDE, bool FULL>
__device__ __forceinline__ void stream_tile(const uint4 *__restrict__ text16, uint64_t nbytes, uint64_t tile, const uint32_t *filter,
                                            const HgFilterExt *ext, uint32_t fold, uint32_t wa, uint32_t wb, HgTileSum *__restrict__ sums, HgCand *__restrict__ seg, uint32_t seg_cap,
                                            uint32_t *lds_count, uint32_t lane) {
  const uint64_t chunk0 = tile * (HG_TILE_BYTES / 16) + lane;
  const uint64_t nchunks = (nbytes + 15) >> 4;  // 16-byte chunks holding at least one valid byte

  auto load_chunk = [&](int it) -> uint4 {
    const uint64_t g = chunk0 + static_cast<uint64_t>(it) * 64u;
    if (FULL) return text16[g];
    uint4 v = make_uint4(0, 0, 0, 0);
    if (g < nchunks) {
      v = text16[g];
      const uint64_t byte0 = g << 4;
      if (byte0 + 16 > nbytes) {  // zero the bytes past the end of the text
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
  };

  uint32_t tot = 0;                            // newlines this lane saw in earlier iterations of the tile
  uint32_t first_it = HG_NONE32, last_it = 0;  // wave-uniform: iterations holding the first / last newline
  uint32_t first_lane = 0, last_lane = 0;

  auto body = [&](int it, uint4 cur) {
    // exact newline count of this lane's 16 bytes: 128 - popcount of the "not a newline" bits
    uint32_t notnl = __popc(not_newline_bits(cur.x));
    notnl += __popc(not_newline_bits(cur.y));
    notnl += __popc(not_newline_bits(cur.z));
    notnl += __popc(not_newline_bits(cur.w));
    const uint32_t c = 128u - notnl;

    const bool any = Probe<LOG2, WIDE>::template probe4<true>(filter, fold, wa, wb, cur) != 0;

    const uint64_t nlm = __ballot(c != 0);
    if (nlm) {
      if (first_it == HG_NONE32) {
        first_it = it;
        first_lane = __builtin_ctzll(nlm);
      }
      last_it = it;
      last_lane = 63u - __builtin_clzll(nlm);
    }
    if (__ballot(any)) {
      const uint32_t l1 = Probe<LOG2, WIDE>::template probe4<false>(filter, fold, wa, wb, cur);
      if (WIDE) {  // wide mode: fingerprint hits go straight to the verify pass
        append_matches(seg, seg_cap, lds_count, (chunk0 + static_cast<uint64_t>(it) * 64u) << 4, lane, cur, tot, l1);
        tot += c;
        return;
      }
      // neighbours across the lane edge: DPP wave shifts (lane i gets lane i-1 / i+1; the edge lanes' values are masked out above)
      const uint32_t left = __builtin_amdgcn_update_dpp(0u, cur.w, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
      const uint32_t right = __builtin_amdgcn_update_dpp(0u, cur.x, 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
      const uint32_t hits = level2_filter<LOG2>(ext, fold, wa, wb, cur, left, right, lane, l1);
      if (__ballot(hits != 0))
        append_matches(seg, seg_cap, lds_count, (chunk0 + static_cast<uint64_t>(it) * 64u) << 4, lane, cur, tot, hits);
    }
    tot += c;
  };

  constexpr int DEPTH = 3;  // 16-byte loads in flight per lane
  if constexpr (FULL) {
    uint4 buf[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; d++) buf[d] = load_chunk(d);
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
      const uint4 cur = buf[it % DEPTH];
      if (it + DEPTH < ITERS) buf[it % DEPTH] = load_chunk(it + DEPTH);
      body(it, cur);
    }
  } else {
#pragma unroll 1
    for (int it = 0; it < ITERS; it++) body(it, load_chunk(it));
  }

  // tile summary: exact offsets of the first / last newline (re-read two 16-byte chunks, L2-resident)
  const uint32_t nl_count = wave_sum(tot);
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

template <int LOG2, bool WIDE>
__global__ __launch_bounds__(WG_THREADS, 8) void hg_stream_kernel(const uint4 *__restrict__ text16, uint64_t nbytes, uint64_t tile_begin, uint64_t tile_end,
                                                                  const uint4 *__restrict__ filter16, const uint4 *__restrict__ ext16,
                                                                  uint32_t fold, uint32_t wa, uint32_t wb, HgTileSum *__restrict__ sums,
                                                                  HgCand *__restrict__ cands, uint32_t seg_cap,
                                                                  uint32_t *__restrict__ seg_count, uint32_t *__restrict__ counters) {
  // LDS: window hash slots (4 B each) and, while they fit, the slots' neighbour conditions (16 B each)
  constexpr bool EXT_IN_LDS = LOG2 <= 12 && !WIDE;
  __shared__ __attribute__((aligned(16))) uint32_t s_filter[1u << LOG2];
  __shared__ __attribute__((aligned(16))) HgFilterExt s_ext[EXT_IN_LDS ? (1u << LOG2) : 1];
  __shared__ uint32_t s_cand_n;
  {
    uint4 *dst = reinterpret_cast<uint4 *>(s_filter);
    for (uint32_t i = threadIdx.x; i < (4u << LOG2) / 16; i += WG_THREADS) dst[i] = filter16[i];
    if (EXT_IN_LDS) {
      uint4 *edst = reinterpret_cast<uint4 *>(s_ext);
      for (uint32_t i = threadIdx.x; i < (1u << LOG2); i += WG_THREADS) edst[i] = ext16[i];
    }
    if (threadIdx.x == 0) s_cand_n = 0;
  }
  __syncthreads();
  const HgFilterExt *ext = EXT_IN_LDS ? s_ext : reinterpret_cast<const HgFilterExt *>(ext16);
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint64_t tile_stride = static_cast<uint64_t>(gridDim.x) * WG_WAVES;
  const uint64_t full_tiles = nbytes >> HG_TILE_SHIFT;
  HgCand *seg = cands + static_cast<uint64_t>(blockIdx.x) * seg_cap;  // this workgroup's private output segment
  for (uint64_t tile = tile_begin + static_cast<uint64_t>(blockIdx.x) * WG_WAVES + wave; tile < tile_end; tile += tile_stride) {
    if (tile < full_tiles) stream_tile<LOG2, WIDE, true>(text16, nbytes, tile, s_filter, ext, fold, wa, wb, sums, seg, seg_cap, &s_cand_n, lane);
    else stream_tile<LOG2, WIDE, false>(text16, nbytes, tile, s_filter, ext, fold, wa, wb, sums, seg, seg_cap, &s_cand_n, lane);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t n = s_cand_n;
    seg_count[blockIdx.x] = n < seg_cap ? n : seg_cap;
    atomicAdd(&counters[HG_CNT_CANDS], n < seg_cap ? n : seg_cap);
    if (n > seg_cap) atomicMax(&counters[HG_CNT_CAND_NEED], n);
  }
}

// Host-side launcher: picks the instantiation for the database's filter size / mode.
namespace {
template <int L, bool W>
void launch_one(const HgStreamArgs &a, uint32_t grid, hipStream_t stream) {
  const uint4 *t = reinterpret_cast<const uint4 *>(a.text);
  const uint4 *f = reinterpret_cast<const uint4 *>(a.filter);
  const uint4 *x = reinterpret_cast<const uint4 *>(a.ext);
  hipLaunchKernelGGL((hg_stream_kernel<L, W>), dim3(grid), dim3(WG_THREADS), 0, stream, t, a.nbytes, a.tile_begin, a.tile_end, f, x, a.db.fold_mask,
                     a.weights_a, a.weights_b, a.sums, a.cands, a.cand_seg_cap, a.seg_count, a.counters);
}
template <int L, bool W>
int blocks_one() {
  int n = 0;
  (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (hg_stream_kernel<L, W>), WG_THREADS, 0);
  return n > 0 ? n : 1;
}
}  // namespace
void hg_launch_stream(const HgStreamArgs &a, uint32_t grid, hipStream_t stream) {
  if (a.filter_wide) {
    switch (a.filter_log2) {
      case 13: launch_one<13, true>(a, grid, stream); break;
      case 14: launch_one<14, true>(a, grid, stream); break;
      case 15: launch_one<15, true>(a, grid, stream); break;
      default: break;
    }
    return;
  }
  switch (a.filter_log2) {
    case 11: launch_one<11, false>(a, grid, stream); break;
    case 12: launch_one<12, false>(a, grid, stream); break;
    case 13: launch_one<13, false>(a, grid, stream); break;
    case 14: launch_one<14, false>(a, grid, stream); break;
    case 15: launch_one<15, false>(a, grid, stream); break;
    default: break;
  }
}
int hg_stream_blocks_per_cu(uint32_t filter_log2, uint32_t filter_wide) {
  if (filter_wide) return filter_log2 == 13 ? blocks_one<13, true>() : (filter_log2 == 14 ? blocks_one<14, true>() : blocks_one<15, true>());
  switch (filter_log2) {
    case 11: return blocks_one<11, false>();
    case 12: return blocks_one<12, false>();
    case 13: return blocks_one<13, false>();
    case 14: return blocks_one<14, false>();
    default: return blocks_one<15, false>();
  }
}

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

// Hits are first appended to a block-private segment (LDS counter), then each block reserves one contiguous
// range of the compact output with a single global atomic and copies its segment there.
struct HitSink {
  HgHit *seg_hits;
  HgHitAux *seg_aux;
  uint32_t seg_cap;
  uint32_t *lds_count;
  __device__ __forceinline__ void push(uint64_t line_no, uint32_t id, uint32_t to, uint64_t start, uint32_t len, uint32_t pattern) const {
    const uint32_t slot = atomicAdd(lds_count, 1u);
    if (slot < seg_cap) {
      seg_hits[slot] = HgHit{line_no, id, to};
      seg_aux[slot] = HgHitAux{start, len, pattern};
    }
  }
};

__device__ __forceinline__ void flush_hits(const HgConfirmArgs &a, uint32_t *lds_count, uint32_t *lds_base) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t n = *lds_count;
    const uint32_t kept = n < a.hit_seg_cap ? n : a.hit_seg_cap;
    *lds_base = atomicAdd(&a.counters[HG_CNT_HITS], kept);
    if (n > a.hit_seg_cap) atomicMax(&a.counters[HG_CNT_HIT_NEED], n);
    *lds_count = kept;
  }
  __syncthreads();
  const uint32_t n = *lds_count, base = *lds_base;
  const uint64_t seg0 = static_cast<uint64_t>(blockIdx.x) * a.hit_seg_cap;
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    if (base + i < a.hit_cap) {
      a.hits[base + i] = a.tmp_hits[seg0 + i];
      a.aux[base + i] = a.tmp_aux[seg0 + i];
    }
  }
}

// One lane per window hit of the stream pass.  Candidate segment s (written by stream workgroup s) is consumed by
// the CONFIRM_SPLIT blocks s*CONFIRM_SPLIT .. +CONFIRM_SPLIT-1, so that even a few hundred thousand candidates
// keep every CU busy: the work is all memory latency.
// Verify pass: one lane per window hit of the stream pass.  Candidate segment s (written by stream workgroup s) is
// consumed by the HG_CONFIRM_SPLIT blocks s*HG_CONFIRM_SPLIT .. +HG_CONFIRM_SPLIT-1, so that even a few hundred
// thousand candidates keep every CU busy (the work is all memory latency).  Literal-only expressions are finished
// here; the other verified (position, pattern) pairs go to HG_DEFER_SHARDS append-only lists for the automaton passes.
__global__ __launch_bounds__(256) void hg_verify_kernel(HgConfirmArgs a) {
  __shared__ uint32_t s_n, s_base;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  const uint64_t seg0 = static_cast<uint64_t>(blockIdx.x) * a.hit_seg_cap;
  const HitSink sink{a.tmp_hits + seg0, a.tmp_aux + seg0, a.hit_seg_cap, &s_n};
  const uint32_t seg = blockIdx.x / HG_CONFIRM_SPLIT, sub = blockIdx.x % HG_CONFIRM_SPLIT;
  const HgCand *cseg = a.cands + static_cast<uint64_t>(seg) * a.cand_seg_cap;
  const uint32_t n = a.seg_count[seg];
  const uint32_t shard = blockIdx.x % HG_DEFER_SHARDS;
  HgDeferred *dlist = a.deferred + static_cast<uint64_t>(shard) * a.defer_shard_cap;
  for (uint32_t i = sub * blockDim.x + threadIdx.x; i < n; i += HG_CONFIRM_SPLIT * blockDim.x) {
    const HgCand c = cseg[i];
    hgdev::verify_window(a.db, a.text, a.nbytes, c.pos, c.word, [&](uint32_t pattern, uint64_t fs, uint32_t len) {
      const HgPattern &p = a.db.patterns[pattern];
      if (hg_confirm_mode(p) == 0) {
        const uint32_t id = p.id;
        hgdev::confirm_literal(a.text, a.nbytes, a.sums, a.bases, a.bs1, c.pos, c.rank, fs, len,
                               [&](uint64_t line_no, uint32_t to, uint64_t start, uint32_t l) { sink.push(line_no, id, to, start, l, pattern); });
      } else {
        const uint32_t slot = atomicAdd(&a.defer_count[shard], 1u);  // wave-aggregated by the compiler; HG_DEFER_SHARDS addresses
        if (slot < a.defer_shard_cap) dlist[slot] = HgDeferred{c.pos, pattern, c.rank};
        else atomicMax(&a.counters[HG_CNT_DEFER_NEED], slot + 1);
      }
    });
  }
  flush_hits(a, &s_n, &s_base);
}

// Automaton passes over the deferred lists, one launch per confirm mode present in the database so that lanes of a
// wave run the same routine and the common modes keep a small register footprint:
//   MODE 1 context-free single-word automaton (follow table in LDS), MODE 2 <= 2 state words with boundary conditions,
//   MODE 3 the scalar reference routine (multi-word state, all-matches mode).
template <int MODE>
__device__ __forceinline__ void confirm_body(const HgConfirmArgs &a) {
  __shared__ uint32_t s_n, s_base;
  __shared__ uint32_t s_follow[MODE == 1 ? 32 * 256 : 1];
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  const uint64_t seg0 = static_cast<uint64_t>(blockIdx.x) * a.hit_seg_cap;
  const HitSink sink{a.tmp_hits + seg0, a.tmp_aux + seg0, a.hit_seg_cap, &s_n};
  uint32_t *follow_lds = MODE == 1 ? s_follow + (threadIdx.x >> 6) * (32 * 64) + (threadIdx.x & 63u) : s_follow;
  // block b walks shard b % HG_DEFER_SHARDS with the blocks that share it
  const uint32_t shard = blockIdx.x % HG_DEFER_SHARDS, peer = blockIdx.x / HG_DEFER_SHARDS, peers = (gridDim.x + HG_DEFER_SHARDS - 1 - shard) / HG_DEFER_SHARDS;
  const HgDeferred *dlist = a.deferred + static_cast<uint64_t>(shard) * a.defer_shard_cap;
  uint32_t n = a.defer_count[shard];
  if (n > a.defer_shard_cap) n = a.defer_shard_cap;
  for (uint32_t i = peer * blockDim.x + threadIdx.x; i < n; i += peers * blockDim.x) {
    const HgDeferred d = dlist[i];
    const HgPattern &p = a.db.patterns[d.pattern];
    if (hg_confirm_mode(p) != MODE) continue;
    const uint32_t id = p.id, pattern = d.pattern;
    auto emit = [&](uint64_t line_no, uint32_t to, uint64_t start, uint32_t len) { sink.push(line_no, id, to, start, len, pattern); };
    if (MODE == 1) {
      hgdev::confirm_simple(a.db, a.text, a.nbytes, a.sums, a.bases, a.bs1, d.pos, p, d.rank, follow_lds, emit);
    } else if (MODE == 2) {
      if (p.nw == 1) hgdev::confirm_ctx<1>(a.db, a.text, a.nbytes, a.sums, a.bases, a.bs1, d.pos, p, d.rank, emit);
      else hgdev::confirm_ctx<2>(a.db, a.text, a.nbytes, a.sums, a.bases, a.bs1, d.pos, p, d.rank, emit);
    } else {
      hg_confirm(a.db, a.text, a.nbytes, a.sums, a.bases, a.bs1, d.pos, pattern, d.rank, emit);
    }
  }
  flush_hits(a, &s_n, &s_base);
}
__global__ __launch_bounds__(256) void hg_confirm_kernel(HgConfirmArgs a) { confirm_body<1>(a); }
__global__ __launch_bounds__(256) void hg_confirm_ctx_kernel(HgConfirmArgs a) { confirm_body<2>(a); }
__global__ __launch_bounds__(256) void hg_confirm_generic_kernel(HgConfirmArgs a) { confirm_body<3>(a); }

// Always-on tier: one wave per tile, each lane owns 256 bytes and handles the lines that START there.
__global__ __launch_bounds__(256) void hg_always_on_kernel(HgConfirmArgs a) {
  __shared__ uint32_t s_n, s_base;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  const uint64_t seg0 = static_cast<uint64_t>(blockIdx.x) * a.hit_seg_cap;
  const HitSink sink{a.tmp_hits + seg0, a.tmp_aux + seg0, a.hit_seg_cap, &s_n};
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t waves = (static_cast<uint64_t>(gridDim.x) * blockDim.x) >> 6;
  for (uint64_t tile = a.tile_begin + ((static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6); tile < a.tile_end; tile += waves) {
    const uint64_t lo = (tile << HG_TILE_SHIFT) + lane * 256ull;
    uint64_t hi = lo + 256;
    if (hi > a.nbytes) hi = a.nbytes;
    uint32_t cnt = 0;
    for (uint64_t s = lo; s < hi; s++) cnt += a.text[s] == '\n';
    uint32_t rank = wave_inclusive_scan(cnt, lane) - cnt;
    for (uint64_t s = lo; s < hi; s++) {
      const bool starts = s == 0 || a.text[s - 1] == '\n';
      if (starts)
        hg_scan_line_always_on(a.db, a.text, a.nbytes, a.sums, a.bases, a.bs1, s, rank,
                               [&](uint32_t pi, uint64_t line_no, uint32_t to, uint64_t start, uint32_t len) {
                                 sink.push(line_no, a.db.patterns[pi].id, to, start, len, pi);
                               });
      rank += a.text[s] == '\n';
    }
  }
  flush_hits(a, &s_n, &s_base);
}

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
    hg_nfa_scan(a.db.pool, pat, a.text, a.nbytes,
                [&](uint32_t to) { sink.push(0, pat.id, to, 0, static_cast<uint32_t>(a.nbytes), p); });
  }
  flush_hits(a, &s_n, &s_base);
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
__global__ void hg_gather_kernel(const HgHit *hits, const HgHitAux *aux, const uint32_t *perm, uint32_t n, HgHit *oh, HgHitAux *oa) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  oh[i] = hits[perm[i]];
  oa[i] = aux[perm[i]];
}
__global__ void hg_keep_kernel(const HgHit *hits, const HgHitAux *aux, const HgPattern *patterns, uint32_t n, uint8_t *keep) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) keep[i] = hg_keep_hit(hits, aux, patterns, i) ? 1 : 0;
}
