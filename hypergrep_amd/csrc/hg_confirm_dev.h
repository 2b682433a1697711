// Device-only fast paths of the confirm stage (hg_confirm_kernel).  Semantics are exactly those of the scalar
// reference routines in hg_core.h (hg_verify_window, hg_confirm), which stay in use for multi-word / assertion
// patterns and are what the host tests replay; here the same work is arranged for memory latency:
//   * the literal verify compares 4 bytes at a time from aligned dword loads,
//   * the line is located and scanned in aligned 16-byte chunks (SWAR newline / NUL detection),
//   * for "simple" patterns (one state word, no boundary conditions) the automaton step needs only reach[c],
//     whose 16 loads per chunk are independent, and the follow table, which is staged in LDS per lane.
#pragma once
#include <hip/hip_runtime.h>

#include "hg_core.h"

namespace hgdev {

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
__device__ __forceinline__ uint32_t eq_mask16(const uint4 &v, uint32_t c4) {
  const uint32_t m0 = hg_zero_bytes(v.x ^ c4), m1 = hg_zero_bytes(v.y ^ c4), m2 = hg_zero_bytes(v.z ^ c4), m3 = hg_zero_bytes(v.w ^ c4);
  auto pack = [](uint32_t m) { return ((m >> 7) & 1u) | ((m >> 14) & 2u) | ((m >> 21) & 4u) | ((m >> 28) & 8u); };
  return pack(m0) | (pack(m1) << 4) | (pack(m2) << 8) | (pack(m3) << 12);
}

// Start of the line containing `pos` when the previous '\n' lies in [tile_start, pos) (rank > 0).
__device__ __forceinline__ uint64_t line_start_in_tile(const uint8_t *text, uint64_t tile_start, uint64_t pos) {
  uint64_t chunk = (pos - 1) & ~15ull;
  for (;;) {
    const uint4 v = *reinterpret_cast<const uint4 *>(text + chunk);
    uint32_t m = eq_mask16(v, 0x0a0a0a0au);
    if (chunk + 16 > pos) m &= (1u << static_cast<uint32_t>(pos - chunk)) - 1u;  // only bytes before pos
    if (m) return chunk + (31 - __clz(m)) + 1;
    if (chunk <= tile_start) return tile_start;  // cannot happen for rank > 0
    chunk -= 16;
  }
}

// Literal-only SINGLEMATCH expression whose literal was verified at [fs, fs + len): the match is that occurrence if it
// lies inside the bytes hs_scan would see for its line piece (after the leading-NUL skip, before the first NUL).
template <typename Emit>
__device__ __forceinline__ void confirm_literal(const uint8_t *text, uint64_t nbytes, const HgTileSum *sums, const HgTileBase *bases, uint64_t bs1,
                                                uint64_t pos, uint32_t rank, uint64_t fs, uint32_t len, Emit &&emit) {
  const uint64_t t = pos >> HG_TILE_SHIFT, tile_start = t << HG_TILE_SHIFT;
  const uint64_t s = rank == 0 ? bases[t].cs : line_start_in_tile(text, tile_start, pos);
  const uint64_t k = (pos - s) / bs1;
  const uint64_t ps = s + k * bs1;
  const uint64_t limit = ps + bs1 < nbytes ? ps + bs1 : nbytes;
  if (fs < ps || fs + len > limit) return;  // the occurrence straddles a forced break: no piece contains it
  const uint64_t line_no = hg_line_index(text, sums[t], bases[t], tile_start, rank, s, bs1, bs1 < HG_TILE_BYTES) + k;
  // [ps, fs): leading NULs are skipped, any later NUL ends the scanned bytes before the occurrence
  uint64_t a = ps;
  bool seen_data = false, blocked = false;
  for (uint64_t chunk = ps & ~15ull; chunk < fs && !blocked; chunk += 16) {
    const uint4 v = *reinterpret_cast<const uint4 *>(text + chunk);
    const uint32_t lo = chunk < ps ? static_cast<uint32_t>(ps - chunk) : 0u;
    const uint32_t hi = fs - chunk < 16 ? static_cast<uint32_t>(fs - chunk) : 16u;
    const uint32_t range = ((1u << hi) - 1u) & ~((1u << lo) - 1u);
    const uint32_t nul = eq_mask16(v, 0u) & range, data = ~eq_mask16(v, 0u) & range;
    if (!seen_data) {
      if (data) {
        const uint32_t first = __ffs(data) - 1;
        a = chunk + first;
        seen_data = true;
        if (nul >> first) blocked = true;  // a NUL after the first data byte
      }
    } else if (nul) {
      blocked = true;
    }
  }
  if (blocked) return;
  if (!seen_data) a = fs;  // only NULs (or nothing) before the occurrence
  // end of the scanned bytes: first NUL, or just past the first '\n', at or after the occurrence's end
  uint64_t z = limit;
  const uint64_t from = fs + len;
  if (from > fs && text[from - 1] == '\n') {
    z = from;  // the literal ends with the line's newline
  } else {
    for (uint64_t chunk = from & ~15ull; chunk < limit; chunk += 16) {
      const uint4 v = *reinterpret_cast<const uint4 *>(text + chunk);
      const uint32_t lo = chunk < from ? static_cast<uint32_t>(from - chunk) : 0u;
      const uint32_t hi = limit - chunk < 16 ? static_cast<uint32_t>(limit - chunk) : 16u;
      const uint32_t range = ((1u << hi) - 1u) & ~((1u << lo) - 1u);
      const uint32_t nl = eq_mask16(v, 0x0a0a0a0au) & range, nul = eq_mask16(v, 0u) & range;
      if (nl | nul) {
        const uint32_t e = __ffs(nl | nul) - 1;
        z = chunk + (((nl >> e) & 1u) ? e + 1 : e);
        break;
      }
    }
  }
  emit(line_no, static_cast<uint32_t>(fs + len - a), a, static_cast<uint32_t>(z - a));
}

// Confirm one (candidate, pattern) for a "simple" SINGLEMATCH pattern.  follow_lds: this lane's private LDS slot
// (entries interleaved by lane: index v * 64).  Reports at most one hit: the smallest match end offset.
template <typename Emit>
__device__ __forceinline__ void confirm_simple(const HgDbView &db, const uint8_t *text, uint64_t nbytes, const HgTileSum *sums, const HgTileBase *bases,
                                               uint64_t bs1, uint64_t pos, const HgPattern &p, uint32_t rank, uint32_t *follow_lds, Emit &&emit) {
  const uint64_t t = pos >> HG_TILE_SHIFT, tile_start = t << HG_TILE_SHIFT;
  const uint64_t s = rank == 0 ? bases[t].cs : line_start_in_tile(text, tile_start, pos);
  const uint64_t k = (pos - s) / bs1;
  const uint64_t ps = s + k * bs1;
  const uint64_t line_no = hg_line_index(text, sums[t], bases[t], tile_start, rank, s, bs1, bs1 < HG_TILE_BYTES) + k;
  const uint64_t limit = ps + bs1 < nbytes ? ps + bs1 : nbytes;

  const uint32_t *reach = db.pool + p.reach_off, *follow = db.pool + p.follow_off;
  for (uint32_t v = 0; v < p.nnodes; v++) follow_lds[v * 64] = follow[v];
  const uint32_t init = p.init_word, acc = p.acc_all;

  // leading NULs are skipped (hyperscanner.c:207-214): a = first non-NUL byte of the piece
  uint64_t a = ps;
  while (a < limit && text[a] == 0) a++;
  if (a >= limit) return;  // empty or all-NUL piece

  uint32_t S = 0, first_to = HG_NONE32;
  uint64_t z = limit;
  for (uint64_t chunk = a & ~15ull; chunk < limit; chunk += 16) {
    const uint4 v = *reinterpret_cast<const uint4 *>(text + chunk);
    const uint32_t lo = chunk < a ? static_cast<uint32_t>(a - chunk) : 0u;
    const uint32_t hi = limit - chunk < 16 ? static_cast<uint32_t>(limit - chunk) : 16u;
    const uint32_t below_lo = (1u << lo) - 1u;
    const uint32_t nl = eq_mask16(v, 0x0a0a0a0au) & ~below_lo, nul = eq_mask16(v, 0u) & ~below_lo;
    // bytes [lo, end) of this chunk belong to the scanned line; the piece ends inside the chunk if stop
    uint32_t end = hi;
    bool stop = false;
    const uint32_t stops = (nl | nul) & ((1u << hi) - 1u);
    if (stops) {
      const uint32_t e = __ffs(stops) - 1;
      end = ((nl >> e) & 1u) ? e + 1 : e;  // a newline is part of the line, a NUL is not
      stop = true;
    }
    if (first_to == HG_NONE32) {
      uint32_t r[16];
#pragma unroll
      for (int i = 0; i < 16; i++) r[i] = reach[byte_of(v, i)];  // 16 independent loads
#pragma unroll
      for (int i = 0; i < 16; i++) {
        if (static_cast<uint32_t>(i) >= lo && static_cast<uint32_t>(i) < end && first_to == HG_NONE32) {
          uint32_t T = init;
          for (uint32_t x = S; x; x &= x - 1) T |= follow_lds[(__ffs(x) - 1) * 64];
          S = T & r[i];
          if (S & acc) first_to = static_cast<uint32_t>(chunk + i + 1 - a);
        }
      }
    }
    if (stop || hi < 16) {
      z = chunk + end;
      break;
    }
  }
  if (first_to != HG_NONE32) emit(line_no, first_to, a, static_cast<uint32_t>(z - a));
}

// Confirm one (candidate, pattern) for a SINGLEMATCH pattern with up to NW <= 2 state words and arbitrary boundary
// conditions (^ $ \b ...): same chunked walk as confirm_simple, automaton tables read from HBM/L2 (the handful of
// 4-byte lookups per text byte depend on the text only, except follow[], so they pipeline).
template <int NW, typename Emit>
__device__ __forceinline__ void confirm_ctx(const HgDbView &db, const uint8_t *text, uint64_t nbytes, const HgTileSum *sums, const HgTileBase *bases,
                                            uint64_t bs1, uint64_t pos, const HgPattern &p, uint32_t rank, Emit &&emit) {
  const uint64_t t = pos >> HG_TILE_SHIFT, tile_start = t << HG_TILE_SHIFT;
  const uint64_t s = rank == 0 ? bases[t].cs : line_start_in_tile(text, tile_start, pos);
  const uint64_t k = (pos - s) / bs1;
  const uint64_t ps = s + k * bs1;
  const uint64_t line_no = hg_line_index(text, sums[t], bases[t], tile_start, rank, s, bs1, bs1 < HG_TILE_BYTES) + k;
  const uint64_t limit = ps + bs1 < nbytes ? ps + bs1 : nbytes;

  const uint32_t *reach = db.pool + p.reach_off, *follow = db.pool + p.follow_off, *init = db.pool + p.init_off;
  const uint32_t *amask = db.pool + p.amask_off, *acct = db.pool + p.acc_off;

  uint64_t a = ps;
  while (a < limit && text[a] == 0) a++;
  if (a >= limit) return;

  uint32_t S[NW], I[NW];
#pragma unroll
  for (int w = 0; w < NW; w++) { S[w] = 0; I[w] = init[w]; }
  uint32_t pc = HG_PC_START, first_to = HG_NONE32;
  uint64_t z = limit;
  for (uint64_t chunk = a & ~15ull; chunk < limit; chunk += 16) {
    const uint4 v = *reinterpret_cast<const uint4 *>(text + chunk);
    const uint32_t lo = chunk < a ? static_cast<uint32_t>(a - chunk) : 0u;
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
    if (first_to == HG_NONE32) {
#pragma unroll
      for (int i = 0; i < 16; i++) {
        if (static_cast<uint32_t>(i) >= lo && static_cast<uint32_t>(i) < end && first_to == HG_NONE32) {
          const uint32_t c = byte_of(v, i);
          // inside one line a '\n' is always the last scanned byte
          const uint32_t cc = c == '\n' ? HG_NC_NLFINAL : (hg_is_word(c) ? HG_NC_WORD : HG_NC_OTHER);
          const uint32_t *ac = acct + (pc * 5 + cc) * NW, *am = amask + (pc * 4 + cc) * NW, *rc = reach + c * NW;
          uint32_t hit = 0;
#pragma unroll
          for (int w = 0; w < NW; w++) hit |= S[w] & ac[w];
          if (hit) {
            first_to = static_cast<uint32_t>(chunk + i - a);
          } else {
            uint32_t T[NW];
#pragma unroll
            for (int w = 0; w < NW; w++) T[w] = I[w];
#pragma unroll
            for (int w = 0; w < NW; w++)
              for (uint32_t x = S[w]; x; x &= x - 1) {
                const uint32_t *f = follow + (w * 32 + (__ffs(x) - 1)) * NW;
#pragma unroll
                for (int q = 0; q < NW; q++) T[q] |= f[q];
              }
#pragma unroll
            for (int w = 0; w < NW; w++) S[w] = T[w] & rc[w] & am[w];
            pc = hg_prev_ctx(c);
          }
        }
      }
    }
    if (stop || hi < 16) {
      z = chunk + end;
      break;
    }
  }
  if (first_to == HG_NONE32) {  // match ending exactly at the end of the scanned bytes
    const uint32_t *ac = acct + (pc * 5 + HG_NC_END) * NW;
    uint32_t hit = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) hit |= S[w] & ac[w];
    if (hit) first_to = static_cast<uint32_t>(z - a);
  }
  if (first_to != HG_NONE32) emit(line_no, first_to, a, static_cast<uint32_t>(z - a));
}

}  // namespace hgdev
