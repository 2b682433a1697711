// gfx950 kernels for HUGE automata: expressions of more than HG_MAX_NODES (1024) positions — unrolled bounded repeats such
// as foo.{0,3000}bar, [a-z]{2000}x, a{32767} — which hs_compile_multi accepts (hypergrep/lib/c/hyperscanner.c:136; Hyperscan's
// bounded-repeat limit is 32767) and whose state does not fit a lane's registers (up to 16384 state words).
//
// One WAVE runs one (scan unit, expression): the state words live in LDS, every lane owns the words w = lane, lane + 64, ...
// and a text byte is three short phases over them (hg_db.h HgHugeHeader has the table layout):
//   1  accept test; next = init | (state & smask) << 1 with the carry taken from the neighbour word          (shift edges)
//   2  the few set bits of state & xsrc OR their target ranges into next                                     (exception edges)
//   3  state = next & reach[class of the byte] & entry mask of the byte's context
// Only the words up to the highest live one are touched (a bounded repeat that has just started occupies its first word),
// so the usual cost of a byte is one word per phase on one lane's row, whatever the size of the automaton.
// hg_core.h hg_huge_scan_slice is the host mirror of huge_run (the tests replay it against the oracle).
//
// Three callers, mirroring the dense routines in hg_kernels.hip:
//   hg_confirm_huge_kernel     tier 0: verified occurrences of the expression's required literal (confirm mode 4).  A piece is
//                              run ONCE per expression, whichever occurrence comes first: (piece start, expression) pairs are
//                              claimed in a hash table, because a literal that overlaps itself (the a...a of a{32767}) gives a
//                              candidate per byte and each run covers the whole piece.
//   hg_always_on_huge_kernel   tier 1: every piece of every line.
//   hg_block_huge_kernel       block mode (hs_scan): the whole buffer is one scan unit.
// Byte/integer work, no MFMA.  Wave64 only; workgroups are ONE wave.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "hg_core.h"
#include "hg_engine.h"
#include "hg_sink_dev.h"

namespace {

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
  for (int o = 32; o; o >>= 1) {
    const uint32_t u = static_cast<uint32_t>(__shfl_xor(static_cast<int>(v), o, 64));
    v = u > v ? u : v;
  }
  return v;
}

// First position in [from, limit) whose byte satisfies pred, else limit.  All 64 lanes call it with the same arguments.
template <typename Pred>
__device__ __forceinline__ uint64_t wave_find(const uint8_t *text, uint64_t from, uint64_t limit, uint32_t lane, Pred &&pred) {
  for (uint64_t base = from; base < limit; base += 64) {
    const uint64_t p = base + lane;
    const bool hit = p < limit && pred(static_cast<uint32_t>(text[p]));
    const uint64_t m = __builtin_amdgcn_ballot_w64(hit);
    if (m) return base + static_cast<uint32_t>(__builtin_ctzll(m));
  }
  return limit;
}
// Start of the line that contains `pos`, given that the previous '\n' (if any) lies in [floor, pos): one past it, else floor.
__device__ __forceinline__ uint64_t wave_line_start(const uint8_t *text, uint64_t floor, uint64_t pos, uint32_t lane) {
  for (uint64_t top = pos; top > floor;) {
    const bool inside = top - floor > lane;  // p = top - 1 - lane >= floor
    const bool hit = inside && text[top - 1 - lane] == '\n';
    const uint64_t m = __builtin_amdgcn_ballot_w64(hit);
    if (m) return top - static_cast<uint32_t>(__builtin_ctzll(m));
    top = top - floor > 64 ? top - 64 : floor;
  }
  return floor;
}
// hg_trim_piece (hg_core.h), wave-parallel: the bytes hs_scan would see of the piece [ps, limit).
__device__ __forceinline__ void wave_trim_piece(const uint8_t *text, uint64_t ps, uint64_t limit, uint32_t lane, uint64_t &a, uint64_t &z) {
  a = wave_find(text, ps, limit, lane, [](uint32_t c) { return c != 0; });
  z = wave_find(text, a, limit, lane, [](uint32_t c) { return c == 0 || c == '\n'; });
  if (z < limit && text[z] == '\n') z++;
}

// The automaton of `v` over data[0, len), matches that START in [from, upto) (the whole unit: from 0, upto len).  emit(to) is
// called on lane 0 for every distinct match end in ascending order; returns after the first when `single`.
// S, T: nw words of LDS each.  Uniform control flow: every lane of the (single-wave) workgroup calls it with the same arguments.
template <typename Emit>
__device__ __forceinline__ void huge_run(const HgHugeView &v, uint32_t *S, uint32_t *T, const uint8_t *data, uint64_t len, uint64_t from, uint64_t upto, bool single,
                                         uint32_t lane, Emit &&emit) {
  const uint32_t nw = v.nw;
  for (uint32_t w = lane; w < nw; w += 64) {
    S[w] = 0;
    T[w] = 0;
  }
  __syncthreads();
  uint32_t whi = 0;  // S[w] == 0 for every w >= whi; T is all zero between steps
  uint32_t pc = from ? hg_prev_ctx(data[from - 1]) : HG_PC_START;
  uint32_t cv = 0;  // the text, 64 bytes at a time: lane l holds data[i0 + l]
  uint64_t i0 = from;
  for (uint64_t i = from; i < len; i++) {
    if (i >= upto && whi == 0) return;  // no start left and nothing alive
    if (i == from || i - i0 == 64) {
      i0 = i;
      cv = i0 + lane < len ? data[i0 + lane] : 0u;
    }
    const uint32_t c = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(cv), static_cast<int>(i - i0)));
    const uint32_t cc = c == '\n' ? (i + 1 == len ? HG_NC_NLFINAL : HG_NC_NL) : (hg_is_word(c) ? HG_NC_WORD : HG_NC_OTHER);
    const bool inject = i < upto;
    // ---- 1: accept test on the current state, shift edges into T
    uint32_t lim1 = whi + 1;
    if (inject && v.init_hi > lim1) lim1 = v.init_hi;
    if (lim1 > nw) lim1 = nw;
    const uint32_t *accw = hg_huge_acc(v, pc, cc);
    uint32_t any = 0;
    for (uint32_t w = lane; w < lim1; w += 64) {
      const uint32_t s = S[w];
      any |= s & accw[w];
      uint32_t t = inject ? v.init[w] : 0u;
      t |= (s & v.smask[w]) << 1;
      if (w) t |= (S[w - 1] & v.smask[w - 1]) >> 31;
      T[w] = t;
    }
    if (__builtin_amdgcn_ballot_w64(any != 0)) {
      if (lane == 0) emit(static_cast<uint32_t>(i));
      if (single) return;
    }
    __syncthreads();
    // ---- 2: exception edges (targets may lie anywhere: atomics on LDS)
    uint32_t xhi = 0;
    for (uint32_t w = lane; w < whi; w += 64) {
      const uint32_t xs = v.xsrc[w];
      uint32_t x = S[w] & xs;
      if (!x) continue;
      const uint32_t base = v.xrank[w];
      while (x) {
        const uint32_t b = hg_ctz(x);
        x &= x - 1;
        const uint32_t k = base + hg_popc(xs & ((1u << b) - 1u));
        for (uint32_t r = v.xlist[k], r1 = v.xlist[k + 1]; r < r1; r++) {
          const uint32_t lo = v.xt[2 * r], hi = v.xt[2 * r + 1];
          const uint32_t w0 = lo >> 5, w1 = hi >> 5;
          for (uint32_t tw = w0; tw <= w1; tw++)
            atomicOr(&T[tw], (tw == w0 ? hg_bits_from(lo) : 0xFFFFFFFFu) & (tw == w1 ? hg_bits_upto(hi) : 0xFFFFFFFFu));
          if (w1 + 1 > xhi) xhi = w1 + 1;
        }
      }
    }
    uint32_t lim3 = lim1;
    if (__builtin_amdgcn_ballot_w64(xhi != 0)) {  // (uniform) some exception fired: T may be set beyond lim1
      xhi = wave_max_u32(xhi);
      if (xhi > lim3) lim3 = xhi;
    }
    __syncthreads();
    // ---- 3: the byte's class and context select who survives
    const uint32_t *r = v.reach + hg_huge_class(v, c) * nw;
    const uint32_t *m = v.amask + (pc * 4 + cc) * nw;
    whi = 0;
    for (uint32_t base = 0; base < lim3; base += 64) {  // (uniform trip count: the highest live word comes from ballots, no cross-lane reduction)
      const uint32_t w = base + lane;
      uint32_t x = 0;
      if (w < lim3) {
        x = T[w] & r[w];
        if (!v.ctxfree) x &= m[w];
        T[w] = 0;
        S[w] = x;
      }
      const uint64_t live = __builtin_amdgcn_ballot_w64(x != 0);
      if (live) whi = base + 64u - static_cast<uint32_t>(__builtin_clzll(live));
    }
    __syncthreads();
    pc = hg_prev_ctx(c);
  }
  const uint32_t *accw = hg_huge_acc(v, pc, HG_NC_END);
  uint32_t any = 0;
  for (uint32_t w = lane; w < whi; w += 64) any |= S[w] & accw[w];
  if (__builtin_amdgcn_ballot_w64(any != 0) && lane == 0) emit(static_cast<uint32_t>(len));
}

// The tables huge_run reads for every byte — class map, reach rows, init, shift mask, exception sources / ranks, accept and entry
// masks — copied into LDS when they fit `cap_words` (the exception lists stay in HBM / L2: they are read when an exception
// fires).  From L2 every phase of a byte waits for a fetch (4.3 us per byte and wave measured on [a-z]{2000}x: 1.8 GiB/s for
// the always-on pass); staged, the phases wait for LDS.  Returns the view to run with.
__device__ __forceinline__ uint32_t huge_stage_words(const HgHugeView &v) { return 64u + v.nw * (v.ncls + 4u + (v.ctxfree ? 1u : 36u)); }
__device__ __forceinline__ HgHugeView huge_stage(const HgHugeView &v, uint32_t *lds, uint32_t cap_words, uint32_t lane) {
  if (huge_stage_words(v) > cap_words) return v;
  HgHugeView s = v;
  uint32_t at = 0;
  auto put = [&](const uint32_t *src, uint32_t words) -> const uint32_t * {
    for (uint32_t i = lane; i < words; i += 64) lds[at + i] = src[i];
    const uint32_t *dst = lds + at;
    at += words;
    return dst;
  };
  s.cls = put(v.cls, 64u);
  s.reach = put(v.reach, v.ncls * v.nw);
  s.init = put(v.init, v.nw);
  s.smask = put(v.smask, v.nw);
  s.xsrc = put(v.xsrc, v.nw);
  s.xrank = put(v.xrank, v.nw);
  s.acc = put(v.acc, (v.ctxfree ? 1u : 20u) * v.nw);
  if (!v.ctxfree) s.amask = put(v.amask, 16u * v.nw);
  __syncthreads();
  return s;
}

// ---- the register-resident routine: automata of at most 64 K words (K = 1, 2, 4: up to 8192 nodes) whose tables are staged ------
// Lane l holds the state words l, l + 64, ... in REGISTERS, with their init / shift / exception-source masks (and the accept mask
// of a condition-free automaton): they never change during a run.  A byte is then: accept test (ballot), shift edges (the carry
// between neighbouring words is a DPP wave shift), one LDS read per word for the byte's reach row (plus the context rows when the
// automaton has boundary conditions), ballots for liveness.  Exception edges are rare: only when some lane has a live exception
// source does the step go through the LDS copy of the next state (the atomic-OR procedure of huge_run).  Measured on
// [a-z]{2000}x, always-on: the LDS-resident routine above ran at 0.75 us per byte and wave (2.6 GiB/s).
using lds_u32 = __attribute__((address_space(3))) uint32_t;
struct HugeStaged {  // staged copies (huge_stage's layout), as LDS pointers
  const lds_u32 *cls, *reach, *acc, *amask;
};
template <int K, typename Emit>
__device__ __forceinline__ void huge_run_reg(const HgHugeView &v, const HugeStaged &t, lds_u32 *T, const uint8_t *data, uint64_t len, uint64_t from, uint64_t upto, bool single,
                                             uint32_t lane, Emit &&emit) {
  const uint32_t nw = v.nw;
  uint32_t s[K], c_init[K], c_smask[K], c_xsrc[K], c_acc[K];
#pragma unroll
  for (int k = 0; k < K; k++) {
    const uint32_t w = lane + 64u * k;
    const bool in = w < nw;
    s[k] = 0;
    c_init[k] = in ? v.init[w] : 0u;
    c_smask[k] = in ? v.smask[w] : 0u;
    c_xsrc[k] = in ? v.xsrc[w] : 0u;
    c_acc[k] = (in && v.ctxfree) ? v.acc[w] : 0u;
  }
  uint32_t pc = from ? hg_prev_ctx(data[from - 1]) : HG_PC_START;
  uint32_t cv = 0;
  uint64_t i0 = from;
  bool alive = false;
  for (uint64_t i = from; i < len; i++) {
    if (i >= upto && !alive) return;
    if (i == from || i - i0 == 64) {
      i0 = i;
      cv = i0 + lane < len ? data[i0 + lane] : 0u;
    }
    const uint32_t c = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(cv), static_cast<int>(i - i0)));
    const uint32_t cc = c == '\n' ? (i + 1 == len ? HG_NC_NLFINAL : HG_NC_NL) : (hg_is_word(c) ? HG_NC_WORD : HG_NC_OTHER);
    const uint32_t cls = __builtin_amdgcn_readfirstlane((t.cls[c >> 2] >> ((c & 3u) * 8u)) & 0xFFu);
    const bool inject = i < upto;
    // the byte's rows, one LDS read per word
    uint32_t r[K], m[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
      const uint32_t w = lane + 64u * k, wc = w < nw ? w : 0u;
      r[k] = w < nw ? t.reach[cls * nw + wc] : 0u;
      m[k] = v.ctxfree ? 0xFFFFFFFFu : t.amask[(pc * 4 + cc) * nw + wc];
    }
    // accept test on the current state
    if (alive) {
      uint32_t any = 0;
#pragma unroll
      for (int k = 0; k < K; k++) {
        const uint32_t w = lane + 64u * k, wc = w < nw ? w : 0u;
        any |= s[k] & (v.ctxfree ? c_acc[k] : t.acc[(pc * 5 + cc) * nw + wc]);
      }
      if (__builtin_amdgcn_ballot_w64(any != 0)) {
        if (lane == 0) emit(static_cast<uint32_t>(i));
        if (single) return;
      }
    }
    // shift edges: word w's top bit moves into word w + 1 (the next lane; lane 63's into lane 0 of the next slab)
    uint32_t nx[K], exc = 0;
    uint32_t carry_slab = 0;  // top bit of the previous slab's lane 63
#pragma unroll
    for (int k = 0; k < K; k++) {
      const uint32_t x = s[k] & c_smask[k];
      const uint32_t top = x >> 31;
      uint32_t carry = __builtin_amdgcn_update_dpp(0u, top, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);  // lane l gets lane l - 1's, lane 0 gets 0
      if (lane == 0) carry = carry_slab;
      carry_slab = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(top), 63));
      nx[k] = (inject ? c_init[k] : 0u) | (x << 1) | carry;
      exc |= s[k] & c_xsrc[k];
    }
    if (__builtin_amdgcn_ballot_w64(exc != 0)) {  // (rare) exception edges: through the LDS copy, targets may lie anywhere
#pragma unroll
      for (int k = 0; k < K; k++)
        if (lane + 64u * k < nw) T[lane + 64u * k] = nx[k];
      __syncthreads();
#pragma unroll
      for (int k = 0; k < K; k++) {
        const uint32_t w = lane + 64u * k;
        uint32_t x = s[k] & c_xsrc[k];
        if (!x) continue;
        const uint32_t base = v.xrank[w];
        while (x) {
          const uint32_t b = hg_ctz(x);
          x &= x - 1;
          const uint32_t kk = base + hg_popc(c_xsrc[k] & ((1u << b) - 1u));
          for (uint32_t q = v.xlist[kk], q1 = v.xlist[kk + 1]; q < q1; q++) {
            const uint32_t lo = v.xt[2 * q], hi = v.xt[2 * q + 1];
            const uint32_t w0 = lo >> 5, w1 = hi >> 5;
            for (uint32_t tw = w0; tw <= w1; tw++)
              __hip_atomic_fetch_or(&T[tw], (tw == w0 ? hg_bits_from(lo) : 0xFFFFFFFFu) & (tw == w1 ? hg_bits_upto(hi) : 0xFFFFFFFFu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < K; k++)
        if (lane + 64u * k < nw) nx[k] = T[lane + 64u * k];
      __syncthreads();
    }
    uint32_t live = 0;
#pragma unroll
    for (int k = 0; k < K; k++) {
      s[k] = nx[k] & r[k] & m[k];
      live |= s[k];
    }
    alive = __builtin_amdgcn_ballot_w64(live != 0) != 0;
    pc = hg_prev_ctx(c);
  }
  if (alive) {
    uint32_t any = 0;
#pragma unroll
    for (int k = 0; k < K; k++) {
      const uint32_t w = lane + 64u * k, wc = w < nw ? w : 0u;
      any |= s[k] & (v.ctxfree ? c_acc[k] : t.acc[(pc * 5 + HG_NC_END) * nw + wc]);
    }
    if (__builtin_amdgcn_ballot_w64(any != 0) && lane == 0) emit(static_cast<uint32_t>(len));
  }
}

// One expression made ready for a wave: its view (staged where that fits) and which routine runs it.
struct HugeUnit {
  HgHugeView v;
  HugeStaged t;
  uint32_t k;  // words per lane of the register-resident routine (1, 2, 4), 0: the LDS-resident routine
};
__device__ __forceinline__ HugeUnit huge_prepare(const HgHugeView &view, uint32_t *stage, uint32_t cap_words, uint32_t lane) {
  HugeUnit u;
  const bool fits = huge_stage_words(view) <= cap_words;
  u.v = huge_stage(view, stage, cap_words, lane);
  u.k = 0;
  u.t = HugeStaged{nullptr, nullptr, nullptr, nullptr};
  if (fits && view.nw <= 256u) {
    u.k = view.nw <= 64u ? 1u : (view.nw <= 128u ? 2u : 4u);
    // huge_stage's layout: cls[64] reach[ncls * nw] init smask xsrc xrank [nw each] acc[(1 | 20) * nw] amask[16 * nw]
    const lds_u32 *base = (const lds_u32 *)stage;
    u.t.cls = base;
    u.t.reach = base + 64u;
    u.t.acc = base + 64u + view.nw * (view.ncls + 4u);
    u.t.amask = u.t.acc + (view.ctxfree ? 1u : 20u) * view.nw;
  }
  return u;
}
template <typename Emit>
__device__ __forceinline__ void huge_run_unit(const HugeUnit &u, uint32_t *S, uint32_t *T, const uint8_t *data, uint64_t len, bool single, uint32_t lane, Emit &&emit) {
  if (u.k == 1) huge_run_reg<1>(u.v, u.t, (lds_u32 *)T, data, len, 0, len, single, lane, emit);
  else if (u.k == 2) huge_run_reg<2>(u.v, u.t, (lds_u32 *)T, data, len, 0, len, single, lane, emit);
  else if (u.k == 4) huge_run_reg<4>(u.v, u.t, (lds_u32 *)T, data, len, 0, len, single, lane, emit);
  else huge_run(u.v, S, T, data, len, 0, len, single, lane, emit);
}

// (piece start, expression) -> claimed by the first caller.  Open addressing, linear probing; the table has at least twice as
// many slots as there can be callers in a pass.
__device__ __forceinline__ bool claim_piece(unsigned long long *table, uint32_t mask, uint64_t ps, uint32_t pattern) {
  const unsigned long long key = ((static_cast<unsigned long long>(ps) << 24) | pattern) + 1ull;  // ps < 2^40, pattern < 2^24; 0 = empty
  uint32_t h = static_cast<uint32_t>((key * 0x9E3779B97F4A7C15ull) >> 32) & mask;
  for (;;) {
    const unsigned long long old = atomicCAS(&table[h], 0ull, key);
    if (old == 0ull) return true;
    if (old == key) return false;
    h = (h + 1u) & mask;
  }
}

extern __shared__ uint32_t s_dyn[];  // S[nw_max], T[nw_max], staged tables [stage_cap]

// Tier 0 (confirm mode 4).  Same unit of work as hg_confirm (hg_core.h): locate the piece that holds the verified occurrence,
// trim it, run the expression over the whole piece — once per (piece, expression).
__global__ __launch_bounds__(64) void hg_confirm_huge_kernel(HgConfirmArgs a, uint32_t nw_max, uint32_t stage_cap, unsigned long long *claim, uint32_t claim_mask) {
  __shared__ uint32_t s_n, s_base;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  uint32_t *S = s_dyn, *T = s_dyn + nw_max;
  const uint32_t lane = threadIdx.x;
  const uint64_t seg0 = static_cast<uint64_t>(blockIdx.x) * a.hit_seg_cap;
  const HitSink sink{a.tmp_hits + seg0, a.tmp_aux + seg0, a.hit_seg_cap, &s_n};
  constexpr uint32_t MODE = 4;
  uint32_t staged_pattern = HG_NONE32;  // the expression whose tables sit in LDS
  HugeUnit unit{};
  const uint32_t shard = blockIdx.x % HG_DEFER_SHARDS, peer = blockIdx.x / HG_DEFER_SHARDS, peers = (gridDim.x + HG_DEFER_SHARDS - 1 - shard) / HG_DEFER_SHARDS;
  const HgDeferred *dlist = a.deferred + (static_cast<uint64_t>(a.list_of_mode[MODE]) * HG_DEFER_SHARDS + shard) * a.defer_shard_cap;
  uint32_t n = a.defer_count[MODE * HG_DEFER_SHARDS + shard];
  if (n > a.defer_shard_cap) n = a.defer_shard_cap;
  for (uint32_t i = peer; i < n; i += peers) {  // one occurrence at a time, the whole wave on it
    const HgDeferred d = dlist[i];
    const uint32_t pattern = d.pattern & (HG_MAX_PATTERNS - 1u);
    const uint64_t pos = d.pos, t = pos >> HG_TILE_SHIFT, tile_start = t << HG_TILE_SHIFT;
    const uint64_t s = d.rank == 0 ? a.bases[t].cs : wave_line_start(a.text, tile_start, pos, lane);
    const uint64_t k = (pos - s) / a.bs1, ps = s + k * a.bs1;
    uint32_t mine = 0;
    if (lane == 0) mine = claim_piece(claim, claim_mask, ps, pattern) ? 1u : 0u;
    if (!__builtin_amdgcn_readfirstlane(mine)) continue;
    const uint64_t line_no = hg_line_index(a.text, a.sums[t], a.bases[t], tile_start, d.rank, s, a.bs1, a.bs1 < HG_TILE_BYTES) + k;
    const uint64_t limit = ps + a.bs1 < a.nbytes ? ps + a.bs1 : a.nbytes;
    uint64_t pa, pz;
    wave_trim_piece(a.text, ps, limit, lane, pa, pz);
    if (pz <= pa) continue;
    const HgPattern &p = a.db.patterns[pattern];
    if (pattern != staged_pattern) {
      __syncthreads();  // (the previous expression's tables are no longer read)
      unit = huge_prepare(hg_huge_view(a.db.pool, p), s_dyn + 2 * nw_max, stage_cap, lane);
      staged_pattern = pattern;
    }
    const uint32_t id = p.id, len = static_cast<uint32_t>(pz - pa);
    const bool single = p.single != 0;
    huge_run_unit(unit, S, T, a.text + pa, len, single, lane, [&](uint32_t to) { sink.push(a, line_no, id, to, pa, len, pattern, single); });
  }
  flush_hits(a, &s_n, &s_base);
}

// Tier 1: the huge entries [first, last) of the always-on list on every piece of every line that starts in the wave's tiles
// (hg_scan_line_always_on, hg_core.h, with the whole wave on one piece).
__global__ __launch_bounds__(64) void hg_always_on_huge_kernel(HgConfirmArgs a, uint32_t nw_max, uint32_t stage_cap, uint32_t first, uint32_t last) {
  __shared__ uint32_t s_n, s_base;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  uint32_t *S = s_dyn, *T = s_dyn + nw_max;
  const uint32_t lane = threadIdx.x;
  const uint64_t seg0 = static_cast<uint64_t>(blockIdx.x) * a.hit_seg_cap;
  const HitSink sink{a.tmp_hits + seg0, a.tmp_aux + seg0, a.hit_seg_cap, &s_n};
  const bool small = a.bs1 < HG_TILE_BYTES;
  uint32_t staged_pattern = HG_NONE32;  // (ONE huge always-on expression, the usual case: staged once for the whole kernel)
  HugeUnit unit{};
  for (uint64_t tile = a.tile_begin + blockIdx.x; tile < a.tile_end; tile += gridDim.x) {
    const uint64_t tile_start = tile << HG_TILE_SHIFT;
    uint32_t rank_base = 0;  // '\n' bytes in [tile_start, tile_start + off)
    for (uint32_t off = 0; off < HG_TILE_BYTES && tile_start + off < a.nbytes; off += 64) {
      const uint64_t p = tile_start + off + lane;
      const bool inside = p < a.nbytes;
      const bool starts = inside && (p == 0 || a.text[p - 1] == '\n');
      const bool is_nl = inside && a.text[p] == '\n';
      const uint64_t nlm = __builtin_amdgcn_ballot_w64(is_nl);
      for (uint64_t sm = __builtin_amdgcn_ballot_w64(starts); sm; sm &= sm - 1) {  // the lines that start in these 64 bytes
        const uint32_t b = static_cast<uint32_t>(__builtin_ctzll(sm));
        const uint64_t s = tile_start + off + b;
        const uint32_t rank = rank_base + static_cast<uint32_t>(__popcll(nlm & ((1ull << b) - 1ull)));
        uint64_t line_no = hg_line_index(a.text, a.sums[tile], a.bases[tile], tile_start, rank, s, a.bs1, small);
        for (uint64_t ps = s;;) {
          const uint64_t limit = ps + a.bs1 < a.nbytes ? ps + a.bs1 : a.nbytes;
          uint64_t pa, pz;
          wave_trim_piece(a.text, ps, limit, lane, pa, pz);
          if (pz > pa) {
            const uint32_t len = static_cast<uint32_t>(pz - pa);
            for (uint32_t j = first; j < last; j++) {
              const uint32_t pi = a.db.slow[j];
              const HgPattern &pat = a.db.patterns[pi];
              if (pi != staged_pattern) {
                __syncthreads();
                unit = huge_prepare(hg_huge_view(a.db.pool, pat), s_dyn + 2 * nw_max, stage_cap, lane);
                staged_pattern = pi;
              }
              const uint32_t id = pat.id;
              const bool single = pat.single != 0;
              huge_run_unit(unit, S, T, a.text + pa, len, single, lane, [&](uint32_t to) { sink.push(a, line_no, id, to, pa, len, pi, single); });
            }
          }
          // where does the piece end?  after its '\n', else at limit (a forced break: the line goes on as the next piece).
          // [ps, pa) holds NULs and [pa, pz) no '\n' before its last byte, so the first '\n' is that byte or lies behind pz.
          uint64_t q;
          if (pz > pa && a.text[pz - 1] == '\n') q = pz - 1;
          else q = wave_find(a.text, pz, limit, lane, [](uint32_t c) { return c == '\n'; });
          const bool nl = q < limit;
          const uint64_t e = nl ? q + 1 : limit;
          if (nl || e >= a.nbytes) break;
          ps = e;
          line_no++;
        }
      }
      rank_base += static_cast<uint32_t>(__popcll(nlm));
    }
  }
  flush_hits(a, &s_n, &s_base);
}

// Block mode (Face A, hs_scan): the whole buffer is ONE scan unit.  A wave per huge expression that is always-on or whose
// required literal occurs in the block (pattern_flags, hg_block_mark_kernel).
__global__ __launch_bounds__(64) void hg_block_huge_kernel(HgConfirmArgs a, uint32_t nw_max, uint32_t stage_cap, const uint32_t *pattern_flags) {
  __shared__ uint32_t s_n, s_base;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  uint32_t *S = s_dyn, *T = s_dyn + nw_max;
  const uint32_t lane = threadIdx.x;
  const uint64_t seg0 = static_cast<uint64_t>(blockIdx.x) * a.hit_seg_cap;
  const HitSink sink{a.tmp_hits + seg0, a.tmp_aux + seg0, a.hit_seg_cap, &s_n};
  uint32_t seen = 0;  // huge expressions met so far: the k-th one belongs to block k % gridDim.x
  for (uint32_t p = 0; p < a.db.npatterns; p++) {
    const HgPattern &pat = a.db.patterns[p];
    if (pat.nw <= HG_MAX_W) continue;
    if (seen++ % gridDim.x != blockIdx.x) continue;
    if (pat.tier == 0 && !pattern_flags[p]) continue;
    __syncthreads();
    const HugeUnit unit = huge_prepare(hg_huge_view(a.db.pool, pat), s_dyn + 2 * nw_max, stage_cap, lane);
    const uint32_t id = pat.id;
    const bool single = pat.single != 0;
    huge_run_unit(unit, S, T, a.text, a.nbytes, single, lane, [&](uint32_t to) { sink.push(a, 0, id, to, 0, static_cast<uint32_t>(a.nbytes), p, single); });
  }
  flush_hits(a, &s_n, &s_base);
}

template <typename K>
bool allow_lds(K kernel, size_t bytes) {
  if (bytes <= 48 * 1024) return true;
  return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes)) == hipSuccess;
}

}  // namespace

// LDS of a huge-automaton workgroup: two copies of the largest automaton's state words + the staged tables.  A workgroup may take
// the whole 160 KiB of a CU: the staging area shrinks to what the state words leave (an expression whose tables then do not fit
// reads them from L2).
static uint32_t clamp_stage(uint32_t nw_max, uint32_t stage_cap) {
  const size_t room = (160u * 1024u - 256u) / 4u;  // words; a little is kept for the kernels' static LDS
  const size_t state = static_cast<size_t>(nw_max) * 2u;
  return static_cast<uint32_t>(state >= room ? 0u : std::min<size_t>(stage_cap, room - state));
}
size_t hg_huge_lds_bytes(uint32_t nw_max, uint32_t stage_cap) { return (static_cast<size_t>(nw_max) * 2u + clamp_stage(nw_max, stage_cap)) * 4u; }

bool hg_launch_confirm_huge(const HgConfirmArgs &a, uint32_t grid, uint32_t nw_max, uint32_t stage_cap, void *claim, uint32_t claim_mask, hipStream_t stream) {
  const size_t lds = hg_huge_lds_bytes(nw_max, stage_cap);
  if (!allow_lds(hg_confirm_huge_kernel, lds)) return false;
  hipLaunchKernelGGL(hg_confirm_huge_kernel, dim3(grid), dim3(64), lds, stream, a, nw_max, clamp_stage(nw_max, stage_cap), static_cast<unsigned long long *>(claim), claim_mask);
  return true;
}
bool hg_launch_always_on_huge(const HgConfirmArgs &a, uint32_t grid, uint32_t nw_max, uint32_t stage_cap, uint32_t first, uint32_t last, hipStream_t stream) {
  const size_t lds = hg_huge_lds_bytes(nw_max, stage_cap);
  if (!allow_lds(hg_always_on_huge_kernel, lds)) return false;
  hipLaunchKernelGGL(hg_always_on_huge_kernel, dim3(grid), dim3(64), lds, stream, a, nw_max, clamp_stage(nw_max, stage_cap), first, last);
  return true;
}
bool hg_launch_block_huge(const HgConfirmArgs &a, uint32_t grid, uint32_t nw_max, uint32_t stage_cap, const uint32_t *pattern_flags, hipStream_t stream) {
  const size_t lds = hg_huge_lds_bytes(nw_max, stage_cap);
  if (!allow_lds(hg_block_huge_kernel, lds)) return false;
  hipLaunchKernelGGL(hg_block_huge_kernel, dim3(grid), dim3(64), lds, stream, a, nw_max, clamp_stage(nw_max, stage_cap), pattern_flags);
  return true;
}
