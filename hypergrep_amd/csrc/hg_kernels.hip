// gfx950 kernels of the line-scan path.  Replaces the reference's hot loop
// (hypergrep/lib/c/hyperscanner.c:198-226: gzgets -> strlen -> hs_scan -> hs_callback per line) with
//
//   hg_stream_kernel      one pass over the text in HBM: 16 B per lane coalesced loads, per-dword window
//                         hash (v_dot4_u32_u8) probed in a 32 KiB LDS bitmap, exact newline counts per
//                         16 KiB wave tile, candidates compacted through a per-wave LDS queue
//                         (ballot + mbcnt), literal verify, wave-aggregated append to HBM
//   hg_tile_*             3-launch scan of the tile newline summaries -> global piece numbers
//   hg_confirm_kernel     one lane per verified candidate: locate the line piece, run the pattern automaton
//   hg_always_on_kernel   patterns without a long enough required literal: every line, one wave per tile
//   hg_key/gather/keep    ordering + SINGLEMATCH / duplicate rules (sort itself: rocPRIM radix sort)
//
// Byte/integer work, HBM-bound: no MFMA anywhere.  Wave64 only.
#include <hip/hip_runtime.h>

#include "hg_core.h"
#include "hg_engine.h"
#include "hg_post.h"

namespace {

constexpr int WG_WAVES = 8;
constexpr int WG_THREADS = WG_WAVES * 64;
constexpr int QCAP = 512;        // queue entries per wave (8 B each)
constexpr int QDRAIN = QCAP - 256;  // one iteration can push at most 64 lanes x 4 dwords
constexpr int ITERS = HG_TILE_BYTES / 1024;  // 1 KiB per wave-iteration

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ uint32_t mbcnt64(uint64_t m) {
  return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
}
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

struct Queue {
  uint2 *q;
  uint32_t n;
};

// Lane-parallel verify of queued window hits; verified ones are appended to the candidate buffer in HBM.
__device__ __noinline__ void drain_queue(const HgStreamArgs &a, Queue &qu, uint64_t tile_base, uint32_t lane) {
  const uint8_t *text = a.text;
  for (uint32_t i = lane; i < qu.n; i += 64) {
    uint2 e = qu.q[i];
    uint64_t pos = tile_base + static_cast<uint64_t>(e.x & 0xFFFu) * 4u;
    uint32_t rank = e.x >> 12;
    hg_verify_window(a.db, text, a.nbytes, pos, e.y, [&](uint32_t pattern) {
      uint32_t idx = atomicAdd(&a.counters[HG_CNT_CANDS], 1u);
      if (idx < a.cand_cap) a.cands[idx] = HgCand{pos, pattern, rank};
    });
  }
  qu.n = 0;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// Stream pass.  One wave owns one 16 KiB tile at a time; tiles are dealt round-robin over all resident
// waves so that at any moment the chip streams one contiguous window of the text.
__global__ __launch_bounds__(WG_THREADS) void hg_stream_kernel(HgStreamArgs a) {
  __shared__ uint32_t s_bitmap[HG_BITMAP_WORDS];
  __shared__ uint2 s_queue[WG_WAVES][QCAP];

  {
    const uint4 *src = reinterpret_cast<const uint4 *>(a.bitmap);
    uint4 *dst = reinterpret_cast<uint4 *>(s_bitmap);
    for (uint32_t i = threadIdx.x; i < HG_BITMAP_WORDS / 4; i += WG_THREADS) dst[i] = src[i];
  }
  __syncthreads();

  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  Queue qu{s_queue[wave], 0};
  const uint32_t fold = a.db.fold_mask;
  const uint64_t nchunks = (a.nbytes + 15) >> 4;  // 16-byte chunks holding at least one valid byte
  const uint4 *text16 = reinterpret_cast<const uint4 *>(a.text);
  const uint64_t tile_stride = static_cast<uint64_t>(gridDim.x) * WG_WAVES;

  for (uint64_t tile = static_cast<uint64_t>(blockIdx.x) * WG_WAVES + wave; tile < a.ntiles; tile += tile_stride) {
    const uint64_t tile_base = tile << HG_TILE_SHIFT;
    const uint64_t chunk0 = tile * (HG_TILE_BYTES / 16) + lane;
    const bool full = tile_base + HG_TILE_BYTES <= a.nbytes;  // wave-uniform

    auto load_chunk = [&](int it) -> uint4 {
      uint64_t g = chunk0 + static_cast<uint64_t>(it) * 64u;
      if (full) return text16[g];
      uint4 v = make_uint4(0, 0, 0, 0);
      if (g < nchunks) {
        v = text16[g];
        uint64_t byte0 = g << 4;
        if (byte0 + 16 > a.nbytes) {  // zero the bytes past the end of the text
          uint32_t valid = static_cast<uint32_t>(a.nbytes - byte0);
          uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int k = 0; k < 4; k++) {
            uint32_t lo = k * 4u;
            if (valid <= lo) w[k] = 0;
            else if (valid < lo + 4) w[k] &= (1u << ((valid - lo) * 8)) - 1u;
          }
          v = make_uint4(w[0], w[1], w[2], w[3]);
        }
      }
      return v;
    };

    uint32_t tot = 0;                              // newlines this lane saw in earlier iterations of the tile
    uint32_t first_it = HG_NONE32, last_it = 0;    // wave-uniform: iterations holding the first / last newline
    uint32_t first_lane = 0, last_lane = 0;

    uint4 cur = load_chunk(0);
    uint4 nxt = load_chunk(1);
#pragma unroll 2
    for (int it = 0; it < ITERS; it++) {
      uint4 nn = make_uint4(0, 0, 0, 0);
      if (it + 2 < ITERS) nn = load_chunk(it + 2);

      const uint32_t w0 = cur.x, w1 = cur.y, w2 = cur.z, w3 = cur.w;
      const uint32_t m0 = hg_newline_mask(w0), m1 = hg_newline_mask(w1), m2 = hg_newline_mask(w2), m3 = hg_newline_mask(w3);
      const uint32_t c0 = __popc(m0), c1 = c0 + __popc(m1), c2 = c1 + __popc(m2), c = c2 + __popc(m3);

      const uint32_t h0 = hg_hash_window(w0 | fold), h1 = hg_hash_window(w1 | fold);
      const uint32_t h2 = hg_hash_window(w2 | fold), h3 = hg_hash_window(w3 | fold);
      const uint32_t b0 = (s_bitmap[h0 >> 5] >> (h0 & 31u)) & 1u, b1 = (s_bitmap[h1 >> 5] >> (h1 & 31u)) & 1u;
      const uint32_t b2 = (s_bitmap[h2 >> 5] >> (h2 & 31u)) & 1u, b3 = (s_bitmap[h3 >> 5] >> (h3 & 31u)) & 1u;

      const uint64_t nlm = __ballot(c != 0);
      if (nlm) {
        if (first_it == HG_NONE32) {
          first_it = it;
          first_lane = __builtin_ctzll(nlm);
        }
        last_it = it;
        last_lane = 63u - __builtin_clzll(nlm);
      }

      if (__ballot((b0 | b1 | b2 | b3) != 0)) {
        // rare path: rank = newlines in [tile start, this dword)
        const uint32_t before = wave_sum(tot) + wave_inclusive_scan(c, lane) - c;
        const uint32_t didx = static_cast<uint32_t>(it) * 256u + lane * 4u;
        const uint32_t bits[4] = {b0, b1, b2, b3};
        const uint32_t ranks[4] = {before, before + c0, before + c1, before + c2};
        const uint32_t words[4] = {w0, w1, w2, w3};
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const uint64_t mk = __ballot(bits[k] != 0);
          if (mk) {
            if (bits[k]) qu.q[qu.n + mbcnt64(mk)] = make_uint2((didx + k) | (ranks[k] << 12), words[k]);
            qu.n += __popcll(mk);
          }
        }
        if (qu.n >= QDRAIN) drain_queue(a, qu, tile_base, lane);
      }
      tot += c;
      cur = nxt;
      nxt = nn;
    }
    if (qu.n) drain_queue(a, qu, tile_base, lane);

    // tile summary: exact offsets of the first / last newline (re-read two 16-byte chunks, L2-resident)
    const uint32_t nl_count = wave_sum(tot);
    uint32_t first_nl = HG_NONE32, last_nl = HG_NONE32;
    if (nl_count) {
      auto chunk_masks = [&](uint32_t it_, uint32_t lane_) -> uint32_t {  // bit b set: byte b of the chunk is '\n'
        uint64_t g = tile * (HG_TILE_BYTES / 16) + it_ * 64u + lane_;
        uint4 v = text16[g];
        uint64_t byte0 = g << 4;
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t bitsm = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          uint32_t m = hg_newline_mask(w[k]);
#pragma unroll
          for (int b = 0; b < 4; b++)
            if ((m >> (8 * b + 7)) & 1u) bitsm |= 1u << (k * 4 + b);
        }
        if (byte0 + 16 > a.nbytes) bitsm &= (1u << static_cast<uint32_t>(a.nbytes - byte0)) - 1u;
        return bitsm;
      };
      uint32_t fm = chunk_masks(first_it, first_lane), lm = chunk_masks(last_it, last_lane);
      first_nl = first_it * 1024u + first_lane * 16u + (__ffs(fm) - 1);
      last_nl = last_it * 1024u + last_lane * 16u + (31 - __clz(lm));
    }
    if (lane == 0) a.sums[tile] = HgTileSum{nl_count, first_nl, last_nl, nl_count ? nl_count - 1 : 0};
  }
}

// ------------------------------------------------------------------------------------------------
// Tile scan (reduce / spine / apply).  256 threads x 4 tiles per block.
namespace {
constexpr int TS_THREADS = 256;
constexpr int TS_PER_THREAD = 4;
constexpr int TS_BLOCK_TILES = TS_THREADS * TS_PER_THREAD;

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
__device__ HgTileElem thread_elems(const HgTileSum *sums, uint64_t ntiles, uint64_t first, uint64_t bs1, HgTileElem *each) {
  HgTileElem acc = identity_elem();
  for (int k = 0; k < TS_PER_THREAD; k++) {
    uint64_t t = first + k;
    HgTileElem e = identity_elem();
    if (t < ntiles) e = hg_tile_elem(sums[t], t << HG_TILE_SHIFT);
    if (each) each[k] = e;
    acc = hg_tile_combine(acc, e, bs1);
  }
  return acc;
}
}  // namespace

__global__ __launch_bounds__(TS_THREADS) void hg_tile_reduce_kernel(const HgTileSum *sums, uint64_t ntiles, uint64_t bs1, HgTileElem *agg) {
  __shared__ HgTileElem sh[TS_THREADS];
  uint64_t first = (static_cast<uint64_t>(blockIdx.x) * TS_THREADS + threadIdx.x) * TS_PER_THREAD;
  HgTileElem v = thread_elems(sums, ntiles, first, bs1, nullptr);
  v = block_scan(v, sh, bs1);
  if (threadIdx.x == TS_THREADS - 1) agg[blockIdx.x] = v;
}

// One block: exclusive scan of the block aggregates -> state at the start of each block; also the final state.
__global__ __launch_bounds__(TS_THREADS) void hg_tile_spine_kernel(const HgTileElem *agg, uint32_t nblocks, uint64_t bs1,
                                                                   HgTileBase init, HgTileBase *block_base, HgTileBase *final_state) {
  __shared__ HgTileElem sh[TS_THREADS];
  __shared__ HgTileBase carry;
  if (threadIdx.x == 0) carry = init;
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
  if (threadIdx.x == 0) *final_state = carry;
}

__global__ __launch_bounds__(TS_THREADS) void hg_tile_apply_kernel(const HgTileSum *sums, uint64_t ntiles, uint64_t bs1,
                                                                   const HgTileBase *block_base, HgTileBase *bases) {
  __shared__ HgTileElem sh[TS_THREADS];
  uint64_t first = (static_cast<uint64_t>(blockIdx.x) * TS_THREADS + threadIdx.x) * TS_PER_THREAD;
  HgTileElem each[TS_PER_THREAD];
  HgTileElem v = thread_elems(sums, ntiles, first, bs1, each);
  block_scan(v, sh, bs1);
  HgTileElem excl = identity_elem();
  if (threadIdx.x > 0) excl = sh[threadIdx.x - 1];
  HgTileBase st = hg_tile_apply(block_base[blockIdx.x], excl, bs1);
  for (int k = 0; k < TS_PER_THREAD; k++) {
    uint64_t t = first + k;
    if (t < ntiles) bases[t] = st;
    st = hg_tile_apply(st, each[k], bs1);
  }
}

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void append_hit(const HgConfirmArgs &a, uint64_t line_no, uint32_t id, uint32_t to, uint64_t start,
                                           uint32_t len, uint32_t pattern) {
  uint32_t idx = atomicAdd(&a.counters[HG_CNT_HITS], 1u);
  if (idx < a.hit_cap) {
    a.hits[idx] = HgHit{line_no, id, to};
    a.aux[idx] = HgHitAux{start, len, pattern};
  }
}

// Small-buffer mode only (buffer_size - 1 < tile): lines inside a tile may split into several pieces, so the
// per-tile count of inner pieces is recomputed by walking the tile (one thread per tile; slow path).
__global__ __launch_bounds__(256) void hg_tile_inner_kernel(const uint8_t *text, HgTileSum *sums, uint64_t ntiles, uint64_t bs1) {
  for (uint64_t t = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < ntiles; t += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
    HgTileSum s = sums[t];
    if (s.nl_count < 2) continue;
    uint64_t base = t << HG_TILE_SHIFT;
    sums[t].inner = static_cast<uint32_t>(hg_inner_pieces(text, base + s.first_nl + 1, base + s.last_nl + 1, bs1));
  }
}

// One lane per verified candidate.
__global__ __launch_bounds__(256) void hg_confirm_kernel(HgConfirmArgs a) {
  uint32_t n = a.counters[HG_CNT_CANDS];
  if (n > a.cand_cap) n = a.cand_cap;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    HgCand c = a.cands[i];
    const uint32_t id = a.db.patterns[c.pattern].id;
    hg_confirm(a.db, a.text, a.nbytes, a.sums, a.bases, a.bs1, c.pos, c.pattern, c.rank,
               [&](uint64_t line_no, uint32_t to, uint64_t start, uint32_t len) { append_hit(a, line_no, id, to, start, len, c.pattern); });
  }
}

// Always-on tier: one wave per tile, each lane owns 256 bytes and handles the lines that START there.
__global__ __launch_bounds__(256) void hg_always_on_kernel(HgConfirmArgs a) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t waves = (static_cast<uint64_t>(gridDim.x) * blockDim.x) >> 6;
  for (uint64_t tile = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6; tile < a.ntiles; tile += waves) {
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
                                 append_hit(a, line_no, a.db.patterns[pi].id, to, start, len, pi);
                               });
      rank += a.text[s] == '\n';
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
