// Device-only fast paths of the confirm stage (hg_confirm_fast_kernel).  Semantics are exactly those of the scalar
// reference routine in hg_core.h (hg_confirm), which stays in use for multi-word / all-matches patterns and is what the
// host tests replay; here the same work is arranged for memory latency:
//   * the line is located and scanned in aligned 16-byte chunks (SWAR newline / NUL detection), four loads in flight,
//   * the automaton tables of the pattern a wave is working on are staged in LDS (hg_kernels.hip, confirm_tables_body).
#pragma once
#include <hip/hip_runtime.h>

#include "hg_core.h"

namespace hgdev {

using lds_u32 = __attribute__((address_space(3))) uint32_t;

__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t *text, uint64_t off) {
  const uint32_t *p = reinterpret_cast<const uint32_t *>(text + (off & ~3ull));
  const uint32_t sh = static_cast<uint32_t>(off & 3u) * 8u;
  const uint32_t lo = p[0];
  if (sh == 0) return lo;
  return (lo >> sh) | (p[1] << (32u - sh));
}

__device__ __forceinline__ uint32_t byte_of(const uint4 &v, uint32_t i) {
  const uint32_t w = i < 8 ? (i < 4 ? v.x : v.y) : (i < 12 ? v.z : v.w);
  return (w >> ((i & 3u) * 8u)) & 0xFFu;
}
// 16-bit mask of the bytes of v equal to `c`
// (the 0x80 flags of a dword become four adjacent bits with ONE v_dot4_u32_u8: sum of 0x80 * weight over the flagged bytes,
// weights 1 2 4 8 for the even dword of a pair and 16 32 64 128 for the odd one = the byte's bit << 7; the round-2 version
// extracted and merged the flags bit by bit, 28 vector instructions per chunk in the walks of every confirm routine)
__device__ __forceinline__ uint32_t eq_mask16(const uint4 &v, uint32_t c4) {
  const uint32_t m0 = hg_zero_bytes(v.x ^ c4), m1 = hg_zero_bytes(v.y ^ c4), m2 = hg_zero_bytes(v.z ^ c4), m3 = hg_zero_bytes(v.w ^ c4);
#ifdef HG_EQMASK_BITWISE  // (experiment builds: round 2's packing)
  auto pack = [](uint32_t m) { return ((m >> 7) & 1u) | ((m >> 14) & 2u) | ((m >> 21) & 4u) | ((m >> 28) & 8u); };
  return pack(m0) | (pack(m1) << 4) | (pack(m2) << 8) | (pack(m3) << 12);
#endif
  const uint32_t lo = __builtin_amdgcn_udot4(m1, 0x80402010u, __builtin_amdgcn_udot4(m0, 0x08040201u, 0u, false), false);
  const uint32_t hi = __builtin_amdgcn_udot4(m3, 0x80402010u, __builtin_amdgcn_udot4(m2, 0x08040201u, 0u, false), false);
  return (lo >> 7) | (hi << 1);
}

// Walks over the line in aligned 16-byte chunks, up to FOUR loads in flight per step: a walk is a chain of dependent memory
// round trips (each decides whether the next one is needed), and a log line is a handful of chunks, so four at a time
// turn most walks into one round trip.  visit(chunk address, chunk) returns true to stop.
// A step never leaves its 64-byte cache line (round 2 took the four chunks at and below the start wherever they lay: two
// lines per step, and the pass is bound by the lines it pulls from HBM — 4.9 per hit on config 5, rocprofv3 TCC_MISS).
template <typename Visit>
__device__ __forceinline__ void walk_back4(const uint8_t *text, uint64_t chunk, uint64_t lowest_chunk, Visit &&visit) {
  for (;;) {
    const uint64_t base = chunk & ~63ull;  // the chunks of this step: base .. chunk, never below the walk's floor
    uint4 v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      uint64_t c = base + 16u * j;
      if (c > chunk || c < lowest_chunk) c = chunk;  // (not visited: any address that is)
      v[j] = *reinterpret_cast<const uint4 *>(text + c);
    }
#pragma unroll
    for (int j = 3; j >= 0; j--) {
      const uint64_t c = base + 16u * j;
      if (c > chunk) continue;
      if (c < lowest_chunk) return;
      if (visit(c, v[j])) return;
    }
    if (base <= lowest_chunk) return;
    chunk = base - 16u;
  }
}
// forward over [chunk, end): `end` <= the readable size of the buffer (nbytes rounded up to 16)
template <typename Visit>
__device__ __forceinline__ void walk_fwd4(const uint8_t *text, uint64_t chunk, uint64_t end, Visit &&visit) {
  while (chunk < end) {
    const uint64_t base = chunk & ~63ull;  // the chunks of this step: chunk .. the end of its cache line
    uint4 v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      uint64_t c = base + 16u * j;
      if (c < chunk || c >= end) c = chunk;
      v[j] = *reinterpret_cast<const uint4 *>(text + c);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint64_t c = base + 16u * j;
      if (c < chunk) continue;
      if (c >= end) return;
      if (visit(c, v[j])) return;
    }
    chunk = base + 64u;
  }
}

// Start of the line containing `pos` when the previous '\n' lies in [tile_start, pos) (rank > 0).
__device__ __forceinline__ uint64_t line_start_in_tile(const uint8_t *text, uint64_t tile_start, uint64_t pos) {
  uint64_t s = tile_start;  // not reached for rank > 0
  walk_back4(text, (pos - 1) & ~15ull, tile_start, [&](uint64_t chunk, const uint4 &v) {
    uint32_t m = eq_mask16(v, 0x0a0a0a0au);
    if (chunk + 16 > pos) m &= (1u << static_cast<uint32_t>(pos - chunk)) - 1u;  // only bytes before pos
    if (!m) return false;
    s = chunk + (31 - __clz(m)) + 1;
    return true;
  });
  return s;
}

// End of the bytes hs_scan sees when they run on from `from` (no NUL in [a, from)): the first NUL, or just past the first '\n',
// else `limit`.
__device__ __forceinline__ uint64_t scanned_end(const uint8_t *text, uint64_t from, uint64_t limit) {
  uint64_t z = limit;
  walk_fwd4(text, from & ~15ull, limit, [&](uint64_t chunk, const uint4 &v) {
    const uint32_t lo = chunk < from ? static_cast<uint32_t>(from - chunk) : 0u;
    const uint32_t hi = limit - chunk < 16 ? static_cast<uint32_t>(limit - chunk) : 16u;
    const uint32_t range = ((1u << hi) - 1u) & ~((1u << lo) - 1u);
    const uint32_t nl = eq_mask16(v, 0x0a0a0a0au) & range, nul = eq_mask16(v, 0u) & range;
    if (!(nl | nul)) return false;
    const uint32_t e = __ffs(nl | nul) - 1;
    z = chunk + (((nl >> e) & 1u) ? e + 1 : e);
    return true;
  });
  return z;
}

// What the confirm routines need to know about the line up to `upto` (a byte of a verified literal occurrence, so no
// newline lies in [occurrence start, upto)): ONE backward walk finds the line start (rank > 0: the previous newline lies
// in the tile) and applies the NUL rules of the reference (hyperscanner.c:207-217: leading NULs are skipped, the first
// later NUL ends the scanned bytes) to [line start, upto).
struct LineHead {
  uint64_t s;    // line start
  uint64_t a;    // first scanned byte when the piece starts at s: the lowest non-NUL byte of [s, upto), else upto
  bool blocked;  // a NUL lies between a and upto: the scanned bytes end before upto
};
__device__ __forceinline__ LineHead line_head(const uint8_t *text, uint64_t tile_start, uint64_t carry_start, uint32_t rank, uint64_t upto) {
  const uint64_t floor = rank == 0 ? carry_start : tile_start;
  LineHead h{floor, upto, false};
  bool nul_above = false;
  if (upto > floor) {
    walk_back4(text, (upto - 1) & ~15ull, floor & ~15ull, [&](uint64_t chunk, const uint4 &v) {
      const uint32_t hi = upto - chunk < 16 ? static_cast<uint32_t>(upto - chunk) : 16u;
      uint32_t range = (1u << hi) - 1u;
      if (chunk < floor) range &= ~((1u << static_cast<uint32_t>(floor - chunk)) - 1u);
      bool stop = false;
      if (rank > 0) {
        const uint32_t nl = eq_mask16(v, 0x0a0a0a0au) & range;
        if (nl) {
          const uint32_t q = 31 - __clz(nl);
          h.s = chunk + q + 1;
          range &= ~((2u << q) - 1u);
          stop = true;
        }
      }
      const uint32_t zero = eq_mask16(v, 0u);
      const uint32_t nul = zero & range, data = ~zero & range;
      if (data) {
        if (nul_above || (nul && (data & ((1u << (31 - __clz(nul))) - 1u)))) h.blocked = true;
        h.a = chunk + (__ffs(data) - 1);
      }
      if (nul) nul_above = true;
      // (a blocked walk may only end early when the line start is known without it: the caller needs s for the piece index)
      return stop || (h.blocked && rank == 0);
    });
  }
  return h;
}

// Piece geometry shared by the confirm routines.  Returns false when nothing of the line can be scanned at `upto`.
struct PieceView {
  uint64_t line_no, a, limit;
  bool whole;  // a came from the line head (the piece starts at the line start); else: a later piece of an over-long line
};
__device__ __forceinline__ bool piece_view(const uint8_t *text, uint64_t nbytes, const HgTileSum *sums, const HgTileBase *bases, uint64_t bs1, uint64_t pos,
                                           uint32_t rank, uint64_t upto, PieceView *out) {
  const uint64_t t = pos >> HG_TILE_SHIFT, tile_start = t << HG_TILE_SHIFT;
  const HgTileBase tb = bases[t];
  const LineHead h = line_head(text, tile_start, tb.cs, rank, upto);
  const uint64_t k = (pos - h.s) / bs1;
  const uint64_t ps = h.s + k * bs1;
  out->limit = ps + bs1 < nbytes ? ps + bs1 : nbytes;
  out->line_no = hg_line_index(text, sums[t], tb, tile_start, rank, h.s, bs1, bs1 < HG_TILE_BYTES) + k;
  out->whole = k == 0;
  if (k == 0) {
    if (h.blocked) return false;
    out->a = h.a;
    return true;
  }
  // a later piece of a line longer than the scan buffer (rare): leading NULs of [ps, ...) are skipped byte by byte
  uint64_t a = ps;
  while (a < out->limit && text[a] == 0) a++;
  out->a = a;
  return a < out->limit;
}

// Literal-only SINGLEMATCH expression whose literal was verified at [fs, fs + len): the match is that occurrence if it
// lies inside the bytes hs_scan would see for its line piece (after the leading-NUL skip, before the first NUL).
template <typename Emit>
__device__ __forceinline__ void confirm_literal(const uint8_t *text, uint64_t nbytes, const HgTileSum *sums, const HgTileBase *bases, uint64_t bs1,
                                                uint64_t pos, uint32_t rank, uint64_t fs, uint32_t len, Emit &&emit) {
  const uint32_t last_byte = text[fs + len - 1];  // independent of the walk: issued with its first loads
  PieceView pv;
  if (!piece_view(text, nbytes, sums, bases, bs1, pos, rank, fs, &pv)) return;
  const uint64_t from = fs + len;
  if (from > pv.limit) return;  // the occurrence straddles a forced break: no piece contains it
  uint64_t a = pv.a;
  if (!pv.whole) {
    if (a > fs) return;  // the piece starts (after its leading NULs) inside or after the occurrence
    for (uint64_t i = a; i < fs; i++)
      if (text[i] == 0) return;  // a NUL between the first scanned byte and the occurrence
  }
  const uint64_t z = last_byte == '\n' ? from : scanned_end(text, from, pv.limit);
  emit(pv.line_no, static_cast<uint32_t>(from - a), a, static_cast<uint32_t>(z - a));
}

// The automaton confirm routines work on a WINDOW of the line, not on the line.
// A SINGLEMATCH expression reports the smallest match end of its line.  Every match contains an occurrence of the
// expression's required literal at most `lead` bytes after the match's start (hg_compile.cpp: Info::lead; 0xFFFFFFFF =
// no bound), every such occurrence is a candidate of its own, and the finalize keeps the smallest end per (line, id).  So
// the candidate whose verified occurrence starts at `fs` only has to answer for the matches that START in
// [fs - lead, fs]: the automaton starts at max(first scanned byte, fs - lead) with the left context read from the text,
// start states are injected up to fs, and the run ends at the first accept or as soon as no state is alive past fs.
// A wave's 64 lanes then run for a literal's length plus a few bytes each, instead of for their whole lines (the lanes
// wait for the longest one: per-wave timing put a batch at 90-350 us, tools/confirm_waves.py).
struct MatchWindow {
  uint64_t line_no, a, limit, q;  // piece index, first scanned byte of the piece, end of the piece's buffer, first byte the automaton sees
  uint32_t pc;                    // context left of q
};
__device__ __forceinline__ bool match_window(const uint8_t *text, uint64_t nbytes, const HgTileSum *sums, const HgTileBase *bases, uint64_t bs1, uint64_t pos,
                                             uint32_t rank, uint64_t fs, uint32_t lead, MatchWindow *w) {
  PieceView pv;
  if (!piece_view(text, nbytes, sums, bases, bs1, pos, rank, pos, &pv)) return false;  // (a NUL between the first scanned byte and pos: pos is not scanned)
  if (!pv.whole) {  // a later piece of an over-long line (rare): the NUL rule byte by byte
    if (pv.a > pos) return false;
    for (uint64_t i = pv.a; i < pos; i++)
      if (text[i] == 0) return false;
  }
  if (fs < pv.a) return false;  // the occurrence begins before the scanned bytes: no match inside them contains it
  w->line_no = pv.line_no;
  w->a = pv.a;
  w->limit = pv.limit;
  w->q = (lead != 0xFFFFFFFFu && fs - pv.a > lead) ? fs - lead : pv.a;
  w->pc = w->q == pv.a ? static_cast<uint32_t>(HG_PC_START) : hg_prev_ctx(text[w->q - 1]);  // (a scanned byte: neither NUL nor newline)
  return true;
}
// end of the scanned bytes given a match that ends at e (> a): just past a newline that is its last byte, else the first
// NUL / just past the first newline from e on
__device__ __forceinline__ uint64_t window_scanned_end(const uint8_t *text, uint64_t e, uint64_t limit) {
  return text[e - 1] == '\n' ? e : scanned_end(text, e, limit);
}

// Confirm one (candidate, pattern) for a "simple" SINGLEMATCH pattern (one state word, no boundary conditions).  reach[256] /
// follow[nodes]: the pattern's tables, staged by the wave in LDS (no table traffic to HBM / L2; the 16 reach lookups of a chunk
// are independent).  Reports at most one hit: the smallest end of a match that starts in the window.
template <typename Emit>
__device__ __forceinline__ void confirm_simple(const uint8_t *text, uint64_t nbytes, const HgTileSum *sums, const HgTileBase *bases, uint64_t bs1, uint64_t pos,
                                                   uint32_t rank, uint64_t fs, uint32_t lead, uint32_t init, uint32_t acc, const lds_u32 *reach, const lds_u32 *follow,
                                                   Emit &&emit) {
  MatchWindow w;
  if (!match_window(text, nbytes, sums, bases, bs1, pos, rank, fs, lead, &w)) return;
  const uint64_t a = w.a, limit = w.limit, q = w.q;
  uint32_t S = 0, first_to = HG_NONE32;
  bool alive = true;
  for (uint64_t chunk = q & ~15ull; chunk < limit && alive; chunk += 16) {
    const uint4 v = *reinterpret_cast<const uint4 *>(text + chunk);
    const uint32_t lo = chunk < q ? static_cast<uint32_t>(q - chunk) : 0u;
    const uint32_t hi = limit - chunk < 16 ? static_cast<uint32_t>(limit - chunk) : 16u;
    const uint32_t below_lo = (1u << lo) - 1u;
    const uint32_t nl = eq_mask16(v, 0x0a0a0a0au) & ~below_lo, nul = eq_mask16(v, 0u) & ~below_lo;
    // bytes [lo, end) of this chunk belong to the scanned bytes; they end inside the chunk if stop
    uint32_t end = hi;
    bool stop = false;
    const uint32_t stops = (nl | nul) & ((1u << hi) - 1u);
    if (stops) {
      const uint32_t e = __ffs(stops) - 1;
      end = ((nl >> e) & 1u) ? e + 1 : e;  // a newline is part of the line, a NUL is not
      stop = true;
    }
    uint32_t r[16];
#pragma unroll
    for (int i = 0; i < 16; i++) r[i] = reach[byte_of(v, i)];  // 16 independent loads
#pragma unroll
    for (int i = 0; i < 16; i++) {
      if (static_cast<uint32_t>(i) >= lo && static_cast<uint32_t>(i) < end && alive) {
        const uint64_t at = chunk + i;
        uint32_t T = at <= fs ? init : 0u;  // matches that start after fs belong to later occurrences
        for (uint32_t x = S; x; x &= x - 1) T |= follow[__ffs(x) - 1];
        S = T & r[i];
        if (S & acc) {
          first_to = static_cast<uint32_t>(at + 1 - a);
          alive = false;
        } else if (S == 0 && at >= fs) {
          alive = false;  // nothing alive and no start left
        }
      }
    }
    if (stop || hi < 16) break;
  }
  if (first_to != HG_NONE32) {
    const uint64_t z = window_scanned_end(text, a + first_to, limit);
    emit(w.line_no, first_to, a, static_cast<uint32_t>(z - a));
  }
}

// Confirm one (candidate, pattern) for a SINGLEMATCH pattern with up to NW <= 2 state words and arbitrary boundary
// conditions (^ $ \b ...): the same windowed run, with the per-context entry / accept tables.
template <int NW, typename Emit>
__device__ __forceinline__ void confirm_ctx(const uint8_t *text, uint64_t nbytes, const HgTileSum *sums, const HgTileBase *bases, uint64_t bs1, uint64_t pos,
                                                uint32_t rank, uint64_t fs, uint32_t lead, const lds_u32 *reach, const lds_u32 *follow, const lds_u32 *init,
                                                const lds_u32 *amask, const lds_u32 *acct, Emit &&emit) {
  MatchWindow w;
  if (!match_window(text, nbytes, sums, bases, bs1, pos, rank, fs, lead, &w)) return;
  const uint64_t a = w.a, limit = w.limit, q = w.q;
  uint32_t S[NW], I[NW];
#pragma unroll
  for (int u = 0; u < NW; u++) { S[u] = 0; I[u] = init[u]; }
  uint32_t pc = w.pc, first_to = HG_NONE32;
  bool alive = true;
  uint64_t z = limit;  // end of the scanned bytes, once the walk has reached it
  for (uint64_t chunk = q & ~15ull; chunk < limit && alive; chunk += 16) {
    const uint4 v = *reinterpret_cast<const uint4 *>(text + chunk);
    const uint32_t lo = chunk < q ? static_cast<uint32_t>(q - chunk) : 0u;
    const uint32_t hi = limit - chunk < 16 ? static_cast<uint32_t>(limit - chunk) : 16u;
    const uint32_t below_lo = (1u << lo) - 1u;
    const uint32_t nl = eq_mask16(v, 0x0a0a0a0au) & ~below_lo, nul = eq_mask16(v, 0u) & ~below_lo;
    uint32_t end = hi;
    bool stop = false;
    const uint32_t stops = (nl | nul) & ((1u << hi) - 1u);
    if (stops) {
      const uint32_t e = __ffs(stops) - 1;
      end = ((nl >> e) & 1u) ? e + 1 : e;
      stop = true;
    }
#pragma unroll
    for (int i = 0; i < 16; i++) {
      if (static_cast<uint32_t>(i) >= lo && static_cast<uint32_t>(i) < end && alive) {
        const uint64_t at = chunk + i;
        const uint32_t c = byte_of(v, i);
        // inside one line a '\n' is always the last scanned byte
        const uint32_t cc = c == '\n' ? HG_NC_NLFINAL : (hg_is_word(c) ? HG_NC_WORD : HG_NC_OTHER);
        const lds_u32 *ac = acct + (pc * 5 + cc) * NW, *am = amask + (pc * 4 + cc) * NW, *rc = reach + c * NW;
        uint32_t hit = 0;
#pragma unroll
        for (int u = 0; u < NW; u++) hit |= S[u] & ac[u];
        if (hit) {  // a match ends before this byte
          first_to = static_cast<uint32_t>(at - a);
          alive = false;
        } else {
          uint32_t T[NW];
#pragma unroll
          for (int u = 0; u < NW; u++) T[u] = at <= fs ? I[u] : 0u;  // matches that start after fs belong to later occurrences
#pragma unroll
          for (int u = 0; u < NW; u++)
            for (uint32_t x = S[u]; x; x &= x - 1) {
              const lds_u32 *f = follow + (u * 32 + (__ffs(x) - 1)) * NW;
#pragma unroll
              for (int k = 0; k < NW; k++) T[k] |= f[k];
            }
          uint32_t any = 0;
#pragma unroll
          for (int u = 0; u < NW; u++) any |= S[u] = T[u] & rc[u] & am[u];
          pc = hg_prev_ctx(c);
          if (any == 0 && at >= fs) alive = false;  // nothing alive and no start left
        }
      }
    }
    if (stop || hi < 16) {
      z = chunk + end;
      break;
    }
  }
  if (first_to == HG_NONE32 && alive) {  // the walk reached the end of the scanned bytes: a match may end exactly there
    const lds_u32 *ac = acct + (pc * 5 + HG_NC_END) * NW;
    uint32_t hit = 0;
#pragma unroll
    for (int u = 0; u < NW; u++) hit |= S[u] & ac[u];
    if (hit) {
      first_to = static_cast<uint32_t>(z - a);
      emit(w.line_no, first_to, a, static_cast<uint32_t>(z - a));
    }
    return;
  }
  if (first_to != HG_NONE32) {
    const uint64_t e = a + first_to;
    emit(w.line_no, first_to, a, static_cast<uint32_t>((e > a ? window_scanned_end(text, e, limit) : scanned_end(text, e, limit)) - a));
  }
}

}  // namespace hgdev
