// gfx950 kernels of the always-on tier: expressions without a usable required literal, whose automata see every byte.
// (hyperscanner.c:198-226 runs hs_scan over every line for every expression; here only these expressions pay per byte.)
//
//   hg_always_on_fast_kernel    segment-parallel: a lane owns 256 bytes of a tile and walks them, after a lead-in, with the
//                               unit's tables in LDS (units = groups of expressions sharing one state word, or one expression);
//                               matches are only NOTED (end offset, expression, newlines of the tile before the last byte)
//   hg_always_on_finish_kernel  noted matches -> hits: the line geometry (start, NUL rules, end of the scanned bytes)
//   hg_always_on_kernel         the scalar routine, line by line, for automata of more than two state words
//
// Byte/integer work bound by vector-instruction issue and LDS lookups: no MFMA anywhere.  Wave64 only.
#include <hip/hip_runtime.h>

#include "hg_confirm_dev.h"
#include "hg_core.h"
#include "hg_engine.h"
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

// Scalar always-on pass over the entries [first, last) of the always-on list (patterns of more than two state words).
__global__ __launch_bounds__(256) void hg_always_on_kernel(HgConfirmArgs a, uint32_t first, uint32_t last) {
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
        hg_scan_line_always_on(a.db, a.text, a.nbytes, a.sums, a.bases, a.bs1, s, rank, first, last,
                               [&](uint32_t pi, uint64_t line_no, uint32_t to, uint64_t start, uint32_t len) {
                                 sink.push(a, line_no, a.db.patterns[pi].id, to, start, len, pi, a.db.patterns[pi].single != 0);
                               });
      rank += a.text[s] == '\n';
    }
  }
  flush_hits(a, &s_n, &s_base);
}

// Always-on patterns of bounded length (and <= 2 state words), segment-parallel.  A match of such a pattern that ends in a
// lane's 256-byte segment starts at most max_len - 1 bytes before it, so every lane runs the automaton over its own
// segment plus that much lead-in, independently of the others; '\n' and NUL reset the automaton (a line is scanned on its
// own, a NUL ends the scanned bytes), and what the reset cannot know — whether an earlier NUL already ended the line — the
// line geometry decides when a match is reported (LineHead.blocked).  A SINGLEMATCH pattern can report once per lane and
// line; the ordering pass keeps the smallest end offset.  Expressions of long or unbounded match length start at the
// line's start (the piece's start, past a forced break) instead of a fixed lead-in.
// Tables: one pattern at a time, staged per wave in LDS like the automaton confirm routines.
// The scan loop only notes a match (end offset, pattern, newlines of the tile before its last byte) in the block's private
// list; hg_always_on_finish_kernel locates the lines afterwards, every lane busy.  Keeping the line geometry (and its
// registers) out of the scan kernel is what lets eight waves per SIMD hide the LDS latency of the automaton steps.
struct AlwaysOnCtx {
  const uint8_t *text;
  uint64_t nbytes;
  HgDeferred *list;            // this block's segment of the match list
  uint32_t list_cap;
  hgdev::lds_u32 *list_count;  // LDS counter of the block
};
// (one LDS atomic per noted match.  Round 3 tried one atomic per wave step — ballot of the lanes that are at the call together,
// mbcnt, readlane: no difference, 142.6 against 143.0 GiB/s on the hit-heavy set [0-9]+\.[0-9]+; what bounded that set was the
// finalize: 76 reports per bucket with the bucket count capped at 2^20, all of them sorted by the block-per-bucket kernel)
__device__ __forceinline__ void always_on_note(const AlwaysOnCtx &cx, uint32_t pi, uint64_t end, uint32_t rank_at_last) {
  const uint32_t slot = __hip_atomic_fetch_add(cx.list_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if (slot < cx.list_cap) cx.list[slot] = HgDeferred{end, pi, rank_at_last};
}

template <int NW, bool SIMPLE>
__device__ __forceinline__ void always_on_segment(const AlwaysOnCtx &a, const HgPattern &p, uint32_t pi, const hgdev::lds_u32 *tab, uint64_t lo, uint64_t hi,
                                                  uint64_t line_start, uint64_t bs1, uint32_t rank_lo) {
  if (lo >= hi) return;
  const hgdev::lds_u32 *reach = tab + CT_REACH, *follow = tab + CT_FOLLOW, *amask = tab + CT_AMASK, *acct = tab + CT_ACC;
  const uint8_t *text = a.text;
  // first byte the automaton sees: a match ending in [lo, hi) starts at most max_len - 1 bytes before lo — or, for long /
  // unbounded expressions, anywhere from the start of the line that contains lo
  uint64_t q = line_start;
  if (p.max_len && p.max_len <= HG_ALWAYS_ON_FAST_MAX_LEN) {
    const uint32_t lead = p.max_len - 1;
    q = lo > lead ? lo - lead : 0;
  } else if (lo - line_start >= bs1) {
    q = line_start + (lo - line_start) / bs1 * bs1;  // the piece that contains lo (forced breaks, below)
  }
  const uint64_t base = q & ~15ull;
  // offsets relative to base (all < 16 + 63 + 256 + 1): [first, stop) are consumed; `own` = first offset of the own segment;
  // the byte at `stop` (if inside the text) only lends its context to a match ending there
  const uint32_t first = static_cast<uint32_t>(q - base), own = static_cast<uint32_t>(lo - base), stop = static_cast<uint32_t>(hi - base);
  const bool text_ends = hi >= a.nbytes;  // the byte at `stop` does not exist
  uint32_t S[NW], I[NW];
#pragma unroll
  for (int u = 0; u < NW; u++) { S[u] = 0; I[u] = SIMPLE ? p.init_word : tab[CT_INIT + u]; }
  uint32_t pc = HG_PC_START;
  if (q) {
    const uint32_t before = text[q - 1];
    pc = (before == '\n' || before == 0) ? HG_PC_START : hg_prev_ctx(before);
  }
  // forced breaks: a line longer than the scan buffer continues as a new piece every bs1 bytes (hyperscanner.c:199); the
  // automaton starts afresh there.  next_break: offset (relative to base) of the next one, NO_BREAK until the line is known
  constexpr uint32_t NO_BREAK = 0xFFFFFFFFu;
  uint32_t next_break = NO_BREAK;
  auto break_after = [&](uint64_t piece_start) {  // first break after a piece that starts at piece_start
    const uint64_t at = piece_start + bs1 - base;
    return at < 0x7FFFFFFFull ? static_cast<uint32_t>(at) : NO_BREAK;
  };
  if (q >= line_start) {  // (else a newline inside the lead-in starts the line whose breaks matter)
    const uint64_t d = q - line_start, k = d < bs1 ? 0 : d / bs1;
    next_break = break_after(line_start + k * bs1);
    if (k && d == k * bs1) pc = HG_PC_START;  // q is itself the first byte of a piece
  }
  uint32_t rank = rank_lo;  // newlines in [tile start, current byte) once the walk is inside the own segment
  bool reported = false;    // SINGLEMATCH: this lane already reported the current line
  const bool single = p.single != 0;
  const uint32_t acc_all = p.acc_all;
  // the lane's text arrives 16 bytes at a time, one load ahead (each is a memory round trip of its own: the lanes of a wave
  // read 256 bytes apart)
  auto load16 = [&](uint32_t at) { return (at <= stop && base + at < a.nbytes) ? *reinterpret_cast<const uint4 *>(text + base + at) : make_uint4(0, 0, 0, 0); };
  uint4 chunk = make_uint4(0, 0, 0, 0), ahead = load16(0);
#pragma unroll 1
  for (uint32_t off = 0; off <= stop; off += 4) {  // 16-byte loads, a dword per trip: the unrolled body (and its registers) stays small
    if ((off & 15u) == 0) {
      chunk = ahead;
      ahead = load16(off + 16);
    }
    const uint32_t sel = (off >> 2) & 3u;
    const uint32_t v = sel == 0 ? chunk.x : (sel == 1 ? chunk.y : (sel == 2 ? chunk.z : chunk.w));
    // the reach sets of the four bytes do not depend on the automaton state: fetch them ahead of the dependent chain
    uint32_t rc[4][NW];
#pragma unroll
    for (uint32_t i = 0; i < 4; i++)
#pragma unroll
      for (int u = 0; u < NW; u++) rc[i][u] = reach[((v >> (8 * i)) & 0xFFu) * NW + u];
#pragma unroll
    for (uint32_t i = 0; i < 4; i++) {
      const uint32_t r = off + i;
      if (r < first || r > stop) continue;
      const bool beyond = r == stop;  // context only (or the end of the text)
      const uint32_t c = (beyond && text_ends) ? 0u : ((v >> (8 * i)) & 0xFFu);
      if (r == next_break) {  // the piece ends before this byte: END context for a match ending here, then a fresh start
        if (!SIMPLE) {
          uint32_t hit = 0;
#pragma unroll
          for (int u = 0; u < NW; u++) hit |= S[u] & acct[(pc * 5 + HG_NC_END) * NW + u];
          if (hit && r > own && !(single && reported)) always_on_note(a, pi, base + r, rank);
        }
#pragma unroll
        for (int u = 0; u < NW; u++) S[u] = 0;
        pc = HG_PC_START;
        reported = false;
        next_break = break_after(base + r);
      }
      if (!SIMPLE) {
        // a match can end before this byte; the byte decides the right-hand context (END at a NUL / the end of the text)
        const uint32_t cc = c == 0 ? HG_NC_END : (c == '\n' ? HG_NC_NLFINAL : (hg_is_word(c) ? HG_NC_WORD : HG_NC_OTHER));
        uint32_t hit = 0;
#pragma unroll
        for (int u = 0; u < NW; u++) hit |= S[u] & acct[(pc * 5 + cc) * NW + u];
        if (hit && !(single && reported)) {
          reported = true;  // (a match that ends in an earlier segment is that segment's to report; it also settles this line)
          if (r > own) always_on_note(a, pi, base + r, rank);
        }
      }
      if (beyond) continue;
      if (c == 0) {  // scanned bytes end here (or leading NULs are skipped): start afresh after it
#pragma unroll
        for (int u = 0; u < NW; u++) S[u] = 0;
        pc = HG_PC_START;
        continue;
      }
      uint32_t T[NW];
#pragma unroll
      for (int u = 0; u < NW; u++) T[u] = I[u];
#pragma unroll
      for (int u = 0; u < NW; u++)
        for (uint32_t x = S[u]; x; x &= x - 1) {
          const hgdev::lds_u32 *f = follow + (u * 32 + (__ffs(x) - 1)) * NW;
#pragma unroll
          for (int t = 0; t < NW; t++) T[t] |= f[t];
        }
      if (SIMPLE) {
        S[0] = T[0] & rc[i][0];
        if ((S[0] & acc_all) && !(single && reported)) {
          reported = true;  // (a match that ends in an earlier segment is that segment's to report; it also settles this line)
          if (r >= own) always_on_note(a, pi, base + r + 1, rank);
        }
      } else {
        const uint32_t cc = c == '\n' ? HG_NC_NLFINAL : (hg_is_word(c) ? HG_NC_WORD : HG_NC_OTHER);
#pragma unroll
        for (int u = 0; u < NW; u++) S[u] = T[u] & rc[i][u] & amask[(pc * 4 + cc) * NW + u];
      }
      if (c == '\n') {
        if (!SIMPLE) {  // a match that includes the newline ends the line: END context
          uint32_t hit = 0;
#pragma unroll
          for (int u = 0; u < NW; u++) hit |= S[u] & acct[(HG_PC_NL * 5 + HG_NC_END) * NW + u];
          if (hit && r >= own && !(single && reported)) always_on_note(a, pi, base + r + 1, rank);
        }
        if (r >= own) rank++;
#pragma unroll
        for (int u = 0; u < NW; u++) S[u] = 0;
        pc = HG_PC_START;
        reported = false;
        next_break = break_after(base + r + 1);
      } else {
        pc = hg_prev_ctx(c);
      }
    }
  }
}

// What a pass of the always-on kernel advances: one expression, or several packed into one state word (HgSlowGroup).
// The members' pattern indices and node masks sit in the wave's table area (CT_MEMBER).
struct AoUnit {
  uint32_t init_word, acc_all, max_len, nmembers, single_mask, nnodes;
};

// ---- single-word units: lean dword steps; an exact per-byte walk where they do not apply ------------------------------
// Round 2's routines spent 30 (context-free) to 55 (with boundary conditions) vector instructions per byte and lane on what
// almost never happens inside a dword: a match to note, a forced break, the edge of the text.  The lean step advances the
// automaton over the four bytes of a dword with table lookups only:
//   context-free:  S' = fuI(S) & reachL[c]; accept if S' & acc.  reachL = reach, but 0 at NUL and '\n' (a line ends there)
//   with conditions: the per-context masks are folded into tables indexed by (class of the previous byte, byte):
//                  RX = reach[c] & amask[pc][cc(c)] (0 at NUL and '\n'),  AX = acct[pc][cc(c)] (a match may end before this byte)
// fuI = the follow unions with the init nodes folded into table 0.  A match met on the way is noted from what the step holds
// (which byte, which members, the newlines before it).  What the lean steps leave out — forced breaks (lines longer than the
// scan buffer), the end of the text, units whose matches can include the newline (an accepting node that consumes '\n') —
// is decided per wave before the walk: such a wave takes the exact routine (round 2's per-byte code) for every dword.
struct AoWalk {
  uint64_t base;                 // text offset of r = 0
  uint32_t own, stop, bs1c;      // own segment = [own, stop); bytes r < stop are consumed, r == stop only lends its context
  bool text_ends;                // the byte at `stop` does not exist
};
__device__ __forceinline__ uint32_t ao_break_after(uint32_t piece_start_rel, uint32_t bs1c) {
  const uint32_t at = piece_start_rel + bs1c;  // (both below 2^31)
  return at < 0x7FFFFFFFu ? at : 0xFFFFFFFFu;
}
// Follow step without tables: in position order most follow sets are "the next node" (concatenation), "the node itself" (+ / *)
// and "the node after an optional one" (a b* c: a -> c), so
//   init | follow(S) = init | (S & M0) | ((S & M1) << 1) | ((S & M2) << 2) | ((S & M3) << 3) | the follow rest of <= 2 exception nodes
// (the head of an alternation, the way back in a repeated group).  Units of more than 8 nodes that fit take this form in their
// lean steps: three or four LDS lookups per byte less, which is what a unit with boundary conditions is bound by.
struct AoShift {
  uint32_t M0, M1, M2, M3, I, nexc, src0, F0, src1, F1;
};
template <int NT>
__device__ __forceinline__ uint32_t ao_follow(const hgdev::lds_u32 *fu, uint32_t S) {  // init | follow(S)
  uint32_t T = fu[S & 0xFFu];
  if (NT > 1) T |= fu[256 + ((S >> 8) & 0xFFu)];
  if (NT > 2) T |= fu[512 + ((S >> 16) & 0xFFu)];
  if (NT > 3) T |= fu[768 + (S >> 24)];
  return T;
}

// The exact routine: the four bytes at walk offset `off`, byte by byte, from the state the previous dword left.
template <int NT>  // NT == 0: the shift form
__device__ __forceinline__ uint32_t ao_follow_lean(const hgdev::lds_u32 *fu, const AoShift &sh, uint32_t S) {
  if (NT != 0) return ao_follow<(NT ? NT : 1)>(fu, S);
  uint32_t T = ((S & sh.M1) << 1) | sh.I;
  T |= S & sh.M0;
  if (sh.M2) T |= (S & sh.M2) << 2;  // (wave-uniform branches)
  if (sh.M3) T |= (S & sh.M3) << 3;
  if (sh.nexc > 0) T |= static_cast<uint32_t>(__builtin_amdgcn_sbfe(static_cast<int32_t>(S), sh.src0, 1u)) & sh.F0;
  if (sh.nexc > 1) T |= static_cast<uint32_t>(__builtin_amdgcn_sbfe(static_cast<int32_t>(S), sh.src1, 1u)) & sh.F1;
  return T;
}

template <bool CTX, int NT>
__device__ __forceinline__ void ao_exact_dword(const AlwaysOnCtx &a, const AoUnit &p, const hgdev::lds_u32 *tab, const AoWalk &w, uint32_t off, uint32_t v, uint32_t prevc,
                                               uint32_t &S, uint32_t &rank, uint32_t &reported, uint32_t &nb) {
  const hgdev::lds_u32 *reach = tab + CT_REACH, *fu = tab + CT_FU, *amask = tab + CT_AMASK, *acct = tab + CT_ACC;
  const hgdev::lds_u32 *member = tab + CT_MEMBER, *member_nodes = tab + CT_MEMBER + HG_GROUP_MAX_MEMBERS;
  // nodes `hit` -> the members they belong to: mark, and note the match when it is this segment's to report
  auto settle = [&](uint32_t hit, bool mark, bool report, uint64_t end, uint32_t rank_at_last) {
    for (uint32_t m = 0; m < p.nmembers; m++) {
      if (!(hit & member_nodes[m]) || ((p.single_mask & reported) >> m & 1u)) continue;
      if (mark) reported |= 1u << m;
      if (report) always_on_note(a, member[m], end, rank_at_last);
    }
  };
  uint32_t pc = (prevc == '\n' || prevc == 0) ? static_cast<uint32_t>(HG_PC_START) : hg_prev_ctx(prevc);
#pragma unroll
  for (uint32_t i = 0; i < 4; i++) {
    const uint32_t r = off + i;
    const bool consume = r < w.stop;
    if (CTX) {
      const bool at_stop = r == w.stop;
      const bool inside = consume || at_stop;
      const uint32_t c = (at_stop && w.text_ends) ? 0u : ((v >> (8 * i)) & 0xFFu);
      if (inside && r == nb) {  // the piece ends before this byte: END context for a match ending here, then a fresh start
        const uint32_t hit = S & acct[pc * 5 + HG_NC_END];
        if (hit && r > w.own) settle(hit, false, true, w.base + r, rank);
        S = 0;
        pc = HG_PC_START;
        reported = 0;
        nb = ao_break_after(r, w.bs1c);
      }
      // a match can end before this byte; the byte decides the right-hand context (END at a NUL / the end of the text)
      const uint32_t cc = c == '\n' ? HG_NC_NLFINAL : (hg_is_word(c) ? HG_NC_WORD : HG_NC_OTHER);
      const uint32_t hit = inside ? (S & acct[pc * 5 + (c == 0 ? static_cast<uint32_t>(HG_NC_END) : cc)]) : 0u;
      if (hit) settle(hit, true, r > w.own, w.base + r, rank);  // (a match that ends in an earlier segment is that segment's to report; it also settles this line)
      uint32_t Sn = ao_follow<NT>(fu, S) & reach[c] & amask[pc * 4 + cc];
      if (c == 0) Sn = 0;  // scanned bytes end here (or leading NULs are skipped): start afresh after it
      S = consume ? Sn : S;
      const bool nl = consume && c == '\n';
      if (nl) {  // a match that includes the newline ends the line: END context
        const uint32_t hit_nl = S & acct[HG_PC_NL * 5 + HG_NC_END];
        if (hit_nl && r >= w.own) settle(hit_nl, false, true, w.base + r + 1, rank);
      }
      if (consume) pc = (nl || c == 0) ? static_cast<uint32_t>(HG_PC_START) : hg_prev_ctx(c);
      rank += (nl && r >= w.own) ? 1u : 0u;
      S = nl ? 0u : S;
      reported = nl ? 0u : reported;
      nb = nl ? ao_break_after(r + 1u, w.bs1c) : nb;
    } else {
      const uint32_t c = (v >> (8 * i)) & 0xFFu;
      if (r == nb) {  // a forced break: the automaton starts afresh with this byte
        S = 0;
        reported = 0;
        nb = ao_break_after(r, w.bs1c);
      }
      const uint32_t Sn = c ? (ao_follow<NT>(fu, S) & reach[c]) : 0u;  // a NUL ends the scanned bytes (or is a skipped leading one): start afresh after it
      S = consume ? Sn : S;
      if (consume && (Sn & p.acc_all)) {  // some member accepts: which ones?  `reported`: bit m = member m has reported on this line
        for (uint32_t m = 0; m < p.nmembers; m++) {
          if (!(Sn & member_nodes[m]) || ((p.single_mask & reported) >> m & 1u)) continue;
          reported |= 1u << m;  // (a match that ends in an earlier segment is that segment's to report; it also settles this line)
          if (r >= w.own) always_on_note(a, member[m], w.base + r + 1, rank);
        }
      }
      const bool nl = consume && c == '\n';
      rank += (nl && r >= w.own) ? 1u : 0u;
      S = nl ? 0u : S;
      reported = nl ? 0u : reported;
      nb = nl ? ao_break_after(r + 1u, w.bs1c) : nb;
    }
  }
}

// One lean step: the dword `v` at walk offset `off`.  OWN: the step lies in the lanes' own segments (matches are noted);
// else in the lead-in (a match only marks its member as reported on its line).
struct AoLane {
  uint32_t S, pcs;
  uint32_t nlc;                 // newlines met since the walk began
  uint32_t reported, rep_nlc;   // bit m: member m has reported on the line that began after newline number rep_nlc
};
template <bool CTX, int NT, bool OWN>
__device__ __forceinline__ void ao_step(const AlwaysOnCtx &a, const AoUnit &p, const AoShift &sh, const hgdev::lds_u32 *tab, const AoWalk &w, uint32_t off, uint32_t v,
                                        AoLane &s, uint32_t rank_base, uint32_t nlacc) {
  const hgdev::lds_u32 *fu = tab + CT_FU, *rxa = tab + CT_RXA;
  const __attribute__((address_space(3))) uint8_t *rxa8 = reinterpret_cast<const __attribute__((address_space(3))) uint8_t *>(rxa);
  uint32_t S = s.S;
  uint32_t hb[4];  // accepting nodes met at each byte (CTX: before it)
  const uint32_t m = hg_newline_mask(v);
  if (CTX) {
    uint32_t cb[4], k[4];
#pragma unroll
    for (uint32_t i = 0; i < 4; i++) {
      cb[i] = ((v >> (8 * i)) & 0xFFu) << 3;  // byte offset of the byte's entry within a class
      k[i] = *reinterpret_cast<const hgdev::lds_u32 *>(rxa8 + 2048u + cb[i]);
    }
    uint2 e[4];
#pragma unroll
    for (uint32_t i = 0; i < 4; i++) {
      const hgdev::lds_u32 *ep = reinterpret_cast<const hgdev::lds_u32 *>(rxa8 + ((i == 0 ? s.pcs : k[i - 1]) | cb[i]));
      e[i] = make_uint2(ep[0], ep[1]);
    }
#pragma unroll
    for (uint32_t i = 0; i < 4; i++) {
      hb[i] = S & e[i].y;
      S = ao_follow_lean<NT>(fu, sh, S) & e[i].x;
    }
    s.pcs = k[3];
  } else {
    uint32_t rc[4];
#pragma unroll
    for (uint32_t i = 0; i < 4; i++) rc[i] = rxa[(v >> (8 * i)) & 0xFFu];
    uint32_t Sb[4];  // the state before each byte
#pragma unroll
    for (uint32_t i = 0; i < 4; i++) {
      Sb[i] = S;
      S = ao_follow_lean<NT>(fu, sh, S) & rc[i];
      hb[i] = S & p.acc_all;
    }
    // (wave-uniform, rare: an accepting node consumes the newline, e.g. [0-9]+\s — a match may end WITH its line.  reachL['\n'] is 0,
    // so what would survive a newline of this dword is looked at here: nlacc = reach['\n'] & the accepting nodes)
    if (nlacc != 0 && m != 0) {
#pragma unroll
      for (uint32_t i = 0; i < 4; i++)
        if ((m >> (8 * i + 7)) & 1u) hb[i] |= ao_follow_lean<NT>(fu, sh, Sb[i]) & nlacc;
    }
  }
  s.S = S;
  if (hb[0] | hb[1] | hb[2] | hb[3]) {  // a match (rare)
    const hgdev::lds_u32 *member = tab + CT_MEMBER, *member_nodes = tab + CT_MEMBER + HG_GROUP_MAX_MEMBERS;
#pragma unroll
    for (uint32_t i = 0; i < 4; i++) {
      const uint32_t h = hb[i], r = off + i;
      if (!h) continue;
      const uint32_t line = s.nlc + __popc(m & ((1u << (8 * i)) - 1u));  // newlines before this byte: its line
      if (s.rep_nlc != line) s.reported = 0;
      for (uint32_t mm = 0; mm < p.nmembers; mm++) {
        if (!(h & member_nodes[mm]) || ((p.single_mask & s.reported) >> mm & 1u)) continue;
        s.reported |= 1u << mm;  // (a match that ends in an earlier segment is that segment's to report; it also settles this line)
        s.rep_nlc = line;
        if (!OWN) continue;
        if (CTX) {
          if (r > w.own) always_on_note(a, member[mm], w.base + r, rank_base + line);
        } else {
          always_on_note(a, member[mm], w.base + r + 1, rank_base + line);
        }
      }
    }
  }
  s.nlc += __popc(m);
}

// All lanes of the wave walk in step: r = 0 lies `own` bytes before every lane's segment, `own` = the longest lead-in of the
// wave rounded up to 16 bytes; a lane whose lead-in is shorter reads zeros up to its first chunk (a NUL leaves the automaton
// in its start state with the START context, which is what the start of a line or piece is).  So the loop bounds, the chunk
// loads' cadence and "inside the own segment" are wave-uniform.  Lanes past the end of the text take part with an empty segment.
template <bool CTX, int NT>
__device__ __forceinline__ void always_on_word(const AlwaysOnCtx &a, const AoUnit &p, const hgdev::lds_u32 *tab, uint64_t lo, uint64_t hi, uint64_t line_start,
                                               uint64_t bs1, uint32_t rank_lo) {
  const uint8_t *text = a.text;
  const bool bounded = p.max_len && p.max_len <= HG_ALWAYS_ON_FAST_MAX_LEN;  // wave-uniform
  uint64_t q = line_start;  // first byte the automaton must see: a match ending in [lo, hi) starts at most max_len - 1 bytes before lo —
  if (bounded) {            // or, for long / unbounded expressions, anywhere from the start of the line (piece) that contains lo
    const uint32_t lead = p.max_len - 1;
    q = lo > lead ? lo - lead : 0;
  } else if (lo - line_start >= bs1) {
    q = line_start + (lo - line_start) / bs1 * bs1;
  }
  // The text comes 64 bytes per lane at a time (the lanes of a wave read 256 bytes apart: every 16-byte load of a lane is a
  // request to the L2 of its own — the L1 holds a wave's 64 lines for no longer than the other waves take to bring in theirs;
  // 16 bytes at a time the kernel made 18 times the L2 requests of a streaming read of the same text).
  const uint64_t q64 = q & ~63ull;  // the lane's first block
  const uint32_t lead64 = lo < hi ? static_cast<uint32_t>(lo - q64) : 0u;  // (a lane past the end of the text reads nothing)
  uint32_t own = lead64;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t other = __shfl_xor(own, o, 64);
    own = other > own ? other : own;
  }
  own = __builtin_amdgcn_readfirstlane(own);
  AoWalk w;
  w.base = lo - own;  // (may wrap below zero for the first lanes of the text: only offsets >= start64 are turned into addresses)
  w.own = own;
  w.stop = own + static_cast<uint32_t>(hi - lo);
  w.bs1c = bs1 < 0x7FFFFFFFull ? static_cast<uint32_t>(bs1) : 0x7FFFFFFFu;
  w.text_ends = hi >= a.nbytes;
  const uint32_t start64 = own - lead64;
  // forced breaks: a line longer than the scan buffer continues as a new piece every bs1 bytes (hyperscanner.c:199) and the
  // automaton starts afresh there.  nb = walk offset of the next one at or after the lane's first block (the start of the line
  // itself is none); unknown while the walk is still in the line before line_start: the newline that ends it sets nb
  uint32_t nb = 0xFFFFFFFFu, nb_first;  // nb_first: the first break the walk can meet
  if (q64 >= line_start) {
    const uint64_t d = q64 - line_start, k = d <= bs1 ? 1 : (d + bs1 - 1) / bs1;
    const uint64_t at = line_start + k * bs1 - w.base;
    nb_first = nb = at < 0x7FFFFFFFull ? static_cast<uint32_t>(at) : 0xFFFFFFFFu;
  } else {
    const uint64_t at = line_start + bs1 - w.base;
    nb_first = at < 0x7FFFFFFFull ? static_cast<uint32_t>(at) : 0xFFFFFFFFu;
  }
  // The exact routine for the whole walk (wave-uniform) if a lane can meet a forced break (a newline inside the walk puts the next
  // break bs1 bytes past it: beyond `stop` unless the scan buffer is tiny), if the text ends in the wave's tile, or if a match of
  // a unit with boundary conditions can include the newline (a context-free unit handles that in its lean steps).
  const uint32_t nlacc = __builtin_amdgcn_readfirstlane(tab[CT_NL_ACCEPTS]);
  const bool careful = w.bs1c <= 512u || (CTX && nlacc != 0) || __builtin_amdgcn_ballot_w64(nb_first <= w.stop + 4u || w.text_ends) != 0;
  // the left context of the first byte: a bounded lead-in begins in the middle of a line (every lane of the wave at walk offset 0,
  // but the lanes whose lead-in the start of the text cuts short); a line or piece begins with the START context, which the
  // zeros before it leave behind
  uint32_t pv = 0;  // the previous dword (its last byte is the left context)
  if (bounded && q64 > 0 && start64 == 0) pv = static_cast<uint32_t>(text[q64 - 1]) << 24;
  auto load16 = [&](uint32_t at) {
    return (at >= start64 && at <= w.stop && w.base + at < a.nbytes) ? *reinterpret_cast<const uint4 *>(text + w.base + at) : make_uint4(0, 0, 0, 0);
  };
  if (careful) {
    const uint32_t lim = own + 256u + (CTX ? 4u : 0u);  // wave-uniform (a lane with a shorter segment is at the end of the text)
    uint32_t S = 0, rank = rank_lo, reported = 0;
    uint4 ahead = load16(0), chunk = ahead;
#pragma unroll 1
    for (uint32_t off = 0; off < lim; off += 4) {
      if ((off & 15u) == 0) {
        chunk = ahead;
        ahead = load16(off + 16);
      }
      const uint32_t sel = (off >> 2) & 3u;
      const uint32_t v = sel == 0 ? chunk.x : (sel == 1 ? chunk.y : (sel == 2 ? chunk.z : chunk.w));
      ao_exact_dword<CTX, (NT ? NT : 4)>(a, p, tab, w, off, v, pv >> 24, S, rank, reported, nb);
      pv = v;
    }
    return;
  }
  struct Block { uint4 p[4]; };
  auto load64 = [&](uint32_t at) {
    Block b;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) b.p[k] = load16(at + 16 * k);
    return b;
  };
  AoShift sh{};
  if (NT == 0) {
    const hgdev::lds_u32 *q = tab + CT_SHIFT;
    auto uni = [&](uint32_t i) { return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(q[i])); };
    sh = AoShift{uni(2), uni(1), uni(3), uni(8), p.init_word, uni(0), uni(4), uni(5), uni(6), uni(7)};
  }
  AoLane s;
  s.S = 0; s.nlc = 0; s.reported = 0; s.rep_nlc = 0;
  s.pcs = CTX ? tab[CT_RXA + 512 + 2 * (pv >> 24)] : 0u;
  uint32_t off = 0;
  Block ahead = load64(0);
  for (; off < own; off += 64) {  // the lead-in
    const Block blk = ahead;
    ahead = load64(off + 64);
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
      ao_step<CTX, NT, false>(a, p, sh, tab, w, off + 16 * k, blk.p[k].x, s, 0, nlacc);
      ao_step<CTX, NT, false>(a, p, sh, tab, w, off + 16 * k + 4, blk.p[k].y, s, 0, nlacc);
      ao_step<CTX, NT, false>(a, p, sh, tab, w, off + 16 * k + 8, blk.p[k].z, s, 0, nlacc);
      ao_step<CTX, NT, false>(a, p, sh, tab, w, off + 16 * k + 12, blk.p[k].w, s, 0, nlacc);
    }
  }
  const uint32_t rank_base = rank_lo - s.nlc;  // rank of a byte of the own segment = rank_lo + the newlines met since `own`
  for (; off < own + 256u; off += 64) {
    const Block blk = ahead;
    ahead = load64(off + 64);
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
      ao_step<CTX, NT, true>(a, p, sh, tab, w, off + 16 * k, blk.p[k].x, s, rank_base, nlacc);
      ao_step<CTX, NT, true>(a, p, sh, tab, w, off + 16 * k + 4, blk.p[k].y, s, rank_base, nlacc);
      ao_step<CTX, NT, true>(a, p, sh, tab, w, off + 16 * k + 8, blk.p[k].z, s, rank_base, nlacc);
      ao_step<CTX, NT, true>(a, p, sh, tab, w, off + 16 * k + 12, blk.p[k].w, s, rank_base, nlacc);
    }
  }
  if (CTX) {  // the byte after the segment lends its context to a match that ends with the segment
    const uint32_t c = ahead.p[0].x & 0xFFu;
    const uint32_t h = s.S & tab[CT_RXA + ((s.pcs >> 2) | (2 * c)) + 1];
    if (h) {
      const hgdev::lds_u32 *member = tab + CT_MEMBER, *member_nodes = tab + CT_MEMBER + HG_GROUP_MAX_MEMBERS;
      if (s.rep_nlc != s.nlc) s.reported = 0;
      for (uint32_t mm = 0; mm < p.nmembers; mm++) {
        if (!(h & member_nodes[mm]) || ((p.single_mask & s.reported) >> mm & 1u)) continue;
        always_on_note(a, member[mm], w.base + off, rank_base + s.nlc);
      }
    }
  }
}

// ---- two state words (33..64 nodes), context-free, follow step in shift form: the lean walk with 64-bit states --------------
// (a SHA-1 in hex, a UUID, [^ ]{40}: 250 GiB/s through always_on_segment<2, false>, whose follow step walks the set bits of the
// state through LDS).  What does not fit — boundary conditions, follow sets outside the shift form, a match that can include the
// newline — and every wave the lean steps leave out (always_on_word) keeps that routine.
struct AoShift2 {
  uint2 M0, M1, M2, M3, I, F0, F1, acc;
  uint32_t nexc, src0, src1;
};
__device__ __forceinline__ uint2 ao_follow2(const AoShift2 &sh, uint2 S) {
  uint2 a = make_uint2(S.x & sh.M1.x, S.y & sh.M1.y);
  uint2 T = make_uint2((a.x << 1) | sh.I.x, ((a.y << 1) | (a.x >> 31)) | sh.I.y);
  T.x |= S.x & sh.M0.x;
  T.y |= S.y & sh.M0.y;
  if (sh.M2.x | sh.M2.y) {  // (wave-uniform branches)
    a = make_uint2(S.x & sh.M2.x, S.y & sh.M2.y);
    T.x |= a.x << 2;
    T.y |= (a.y << 2) | (a.x >> 30);
  }
  if (sh.M3.x | sh.M3.y) {
    a = make_uint2(S.x & sh.M3.x, S.y & sh.M3.y);
    T.x |= a.x << 3;
    T.y |= (a.y << 3) | (a.x >> 29);
  }
  if (sh.nexc > 0) {
    const uint32_t e = static_cast<uint32_t>(__builtin_amdgcn_sbfe(static_cast<int32_t>(sh.src0 < 32 ? S.x : S.y), sh.src0 & 31u, 1u));
    T.x |= e & sh.F0.x;
    T.y |= e & sh.F0.y;
  }
  if (sh.nexc > 1) {
    const uint32_t e = static_cast<uint32_t>(__builtin_amdgcn_sbfe(static_cast<int32_t>(sh.src1 < 32 ? S.x : S.y), sh.src1 & 31u, 1u));
    T.x |= e & sh.F1.x;
    T.y |= e & sh.F1.y;
  }
  return T;
}
struct AoLane2 {
  uint2 S;
  uint32_t nlc, reported, rep_nlc;
};
template <bool OWN>
__device__ __forceinline__ void ao_step2(const AlwaysOnCtx &a, const AoShift2 &sh, uint32_t pi, bool single, const hgdev::lds_u32 *tab, const AoWalk &w, uint32_t off, uint32_t v,
                                         AoLane2 &s, uint32_t rank_base) {
  const __attribute__((address_space(3))) uint8_t *r8 = reinterpret_cast<const __attribute__((address_space(3))) uint8_t *>(tab + CT_RXA);
  uint2 rc[4];
#pragma unroll
  for (uint32_t i = 0; i < 4; i++) {
    const hgdev::lds_u32 *e = reinterpret_cast<const hgdev::lds_u32 *>(r8 + (((v >> (8 * i)) & 0xFFu) << 3));
    rc[i] = make_uint2(e[0], e[1]);
  }
  uint2 S = s.S;
  uint32_t hb[4];
#pragma unroll
  for (uint32_t i = 0; i < 4; i++) {
    const uint2 T = ao_follow2(sh, S);
    S = make_uint2(T.x & rc[i].x, T.y & rc[i].y);
    hb[i] = (S.x & sh.acc.x) | (S.y & sh.acc.y);
  }
  s.S = S;
  const uint32_t m = hg_newline_mask(v);
  if (hb[0] | hb[1] | hb[2] | hb[3]) {  // a match (rare)
#pragma unroll
    for (uint32_t i = 0; i < 4; i++) {
      if (!hb[i]) continue;
      const uint32_t line = s.nlc + __popc(m & ((1u << (8 * i)) - 1u));  // newlines before this byte: its line
      if (s.rep_nlc != line) s.reported = 0;
      if (single && s.reported) continue;
      s.reported = 1;
      s.rep_nlc = line;
      if (OWN) always_on_note(a, pi, w.base + off + i + 1, rank_base + line);
    }
  }
  s.nlc += __popc(m);
}
__device__ __forceinline__ void always_on_word2(const AlwaysOnCtx &a, const HgPattern &p, uint32_t pi, const hgdev::lds_u32 *tab, uint64_t lo, uint64_t hi,
                                                uint64_t line_start, uint64_t bs1, uint32_t rank_lo) {
  // the walk's geometry and the decision for the exact routine: as always_on_word
  const uint8_t *text = a.text;
  const bool bounded = p.max_len && p.max_len <= HG_ALWAYS_ON_FAST_MAX_LEN;  // wave-uniform
  uint64_t q = line_start;
  if (bounded) {
    const uint32_t lead = p.max_len - 1;
    q = lo > lead ? lo - lead : 0;
  } else if (lo - line_start >= bs1) {
    q = line_start + (lo - line_start) / bs1 * bs1;
  }
  const uint64_t q64 = q & ~63ull;
  const uint32_t lead64 = lo < hi ? static_cast<uint32_t>(lo - q64) : 0u;
  uint32_t own = lead64;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t other = __shfl_xor(own, o, 64);
    own = other > own ? other : own;
  }
  own = __builtin_amdgcn_readfirstlane(own);
  AoWalk w;
  w.base = lo - own;
  w.own = own;
  w.stop = own + static_cast<uint32_t>(hi - lo);
  w.bs1c = bs1 < 0x7FFFFFFFull ? static_cast<uint32_t>(bs1) : 0x7FFFFFFFu;
  w.text_ends = hi >= a.nbytes;
  const uint32_t start64 = own - lead64;
  uint32_t nb_first;
  if (q64 >= line_start) {
    const uint64_t d = q64 - line_start, k = d <= bs1 ? 1 : (d + bs1 - 1) / bs1;
    const uint64_t at = line_start + k * bs1 - w.base;
    nb_first = at < 0x7FFFFFFFull ? static_cast<uint32_t>(at) : 0xFFFFFFFFu;
  } else {
    const uint64_t at = line_start + bs1 - w.base;
    nb_first = at < 0x7FFFFFFFull ? static_cast<uint32_t>(at) : 0xFFFFFFFFu;
  }
  const hgdev::lds_u32 *q2 = tab + CT_SHIFT + 16;
  auto uni = [&](uint32_t i) { return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(q2[i])); };
  const uint32_t nexc = uni(0);
  const bool careful = nexc > 2u || w.bs1c <= 512u || __builtin_amdgcn_ballot_w64(nb_first <= w.stop + 4u || w.text_ends) != 0;
  if (careful) {
    always_on_segment<2, false>(a, p, pi, tab, lo, hi, line_start, bs1, rank_lo);
    return;
  }
  AoShift2 sh;
  sh.nexc = nexc;
  sh.M0 = make_uint2(uni(1), uni(2)); sh.M1 = make_uint2(uni(3), uni(4)); sh.M2 = make_uint2(uni(5), uni(6)); sh.M3 = make_uint2(uni(7), uni(8));
  sh.src0 = uni(9); sh.F0 = make_uint2(uni(10), uni(11)); sh.src1 = uni(12); sh.F1 = make_uint2(uni(13), uni(14));
  sh.I = make_uint2(static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(tab[CT_INIT])), static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(tab[CT_INIT + 1])));
  sh.acc = make_uint2(static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(tab[CT_ACC])), static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(tab[CT_ACC + 1])));
  auto load16 = [&](uint32_t at) {
    return (at >= start64 && at <= w.stop && w.base + at < a.nbytes) ? *reinterpret_cast<const uint4 *>(text + w.base + at) : make_uint4(0, 0, 0, 0);
  };
  struct Block { uint4 p[4]; };
  auto load64 = [&](uint32_t at) {
    Block b;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) b.p[k] = load16(at + 16 * k);
    return b;
  };
  const bool single = p.single != 0;
  AoLane2 s;
  s.S = make_uint2(0, 0); s.nlc = 0; s.reported = 0; s.rep_nlc = 0;
  uint32_t off = 0;
  Block ahead = load64(0);
  for (; off < own; off += 64) {  // the lead-in
    const Block blk = ahead;
    ahead = load64(off + 64);
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
      ao_step2<false>(a, sh, pi, single, tab, w, off + 16 * k, blk.p[k].x, s, 0);
      ao_step2<false>(a, sh, pi, single, tab, w, off + 16 * k + 4, blk.p[k].y, s, 0);
      ao_step2<false>(a, sh, pi, single, tab, w, off + 16 * k + 8, blk.p[k].z, s, 0);
      ao_step2<false>(a, sh, pi, single, tab, w, off + 16 * k + 12, blk.p[k].w, s, 0);
    }
  }
  const uint32_t rank_base = rank_lo - s.nlc;
  for (; off < own + 256u; off += 64) {
    const Block blk = ahead;
    ahead = load64(off + 64);
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
      ao_step2<true>(a, sh, pi, single, tab, w, off + 16 * k, blk.p[k].x, s, rank_base);
      ao_step2<true>(a, sh, pi, single, tab, w, off + 16 * k + 4, blk.p[k].y, s, rank_base);
      ao_step2<true>(a, sh, pi, single, tab, w, off + 16 * k + 8, blk.p[k].z, s, rank_base);
      ao_step2<true>(a, sh, pi, single, tab, w, off + 16 * k + 12, blk.p[k].w, s, rank_base);
    }
  }
}

// Tables of one unit, staged by the whole workgroup (the caller brackets this with __syncthreads): reach / follow / context
// tables as the database holds them, then what the lean steps read.
__device__ __forceinline__ void always_on_stage(hgdev::lds_u32 *tab, const HgConfirmArgs &a, uint32_t u, uint32_t tid) {
  const uint32_t ngroups = a.db.ngroups;
  uint32_t nnodes, init_word, acc_all;
  bool one_word, ctx;
  if (u < ngroups) {
    const HgSlowGroup &g = a.db.groups[u];
    tab[CT_REACH + tid] = a.db.pool[g.reach_off + tid];
    if (tid < g.nnodes) tab[CT_FOLLOW + tid] = a.db.pool[g.follow_off + tid];
    if (tid < HG_GROUP_MAX_MEMBERS) {
      tab[CT_MEMBER + tid] = g.member[tid];
      tab[CT_MEMBER + HG_GROUP_MAX_MEMBERS + tid] = tid < g.nmembers ? g.acc[tid] : 0u;
    }
    if (g.ctx_off) {  // a group with boundary conditions: the union of the members' per-context tables
      if (tid < 16) tab[CT_AMASK + tid] = a.db.pool[g.ctx_off + tid];
      if (tid < 20) tab[CT_ACC + tid] = a.db.pool[g.ctx_off + 16 + tid];
    }
    nnodes = g.nnodes; init_word = g.init_word; acc_all = g.acc_all;
    one_word = true;
    ctx = g.ctx_off != 0;
  } else {
    const uint32_t pi = a.db.slow[a.db.nslow_grouped + (u - ngroups)];
    const HgPattern &p = a.db.patterns[pi];
    const uint32_t nw = p.simple ? 1u : p.nw;
    for (uint32_t i = tid; i < 256 * nw; i += 256) tab[CT_REACH + i] = a.db.pool[p.reach_off + i];
    if (tid < p.nnodes * nw) tab[CT_FOLLOW + tid] = a.db.pool[p.follow_off + tid];
    if (!p.simple) {
      if (tid < nw) tab[CT_INIT + tid] = a.db.pool[p.init_off + tid];
      if (tid < 16 * nw) tab[CT_AMASK + tid] = a.db.pool[p.amask_off + tid];
      if (tid < 20 * nw) tab[CT_ACC + tid] = a.db.pool[p.acc_off + tid];
    }
    if (tid == 0) {
      tab[CT_MEMBER] = pi;
      tab[CT_MEMBER + HG_GROUP_MAX_MEMBERS] = p.simple ? p.acc_all : 0xFFFFFFFFu;  // (with conditions: the nodes of the member, i.e. all)
    }
    nnodes = p.nnodes; init_word = p.init_word; acc_all = p.acc_all;
    one_word = nw == 1;
    ctx = !p.simple;
  }
  if (!one_word) {  // two state words: can the unit take the lean walk (always_on_word2)?
    __syncthreads();
    if (tid == 0) {
      // context-free: every context enters and accepts the same nodes
      bool same = true;
      for (uint32_t k = 1; k < 16; k++) same = same && tab[CT_AMASK + 2 * k] == tab[CT_AMASK] && tab[CT_AMASK + 2 * k + 1] == tab[CT_AMASK + 1];
      for (uint32_t k = 1; k < 20; k++) same = same && tab[CT_ACC + 2 * k] == tab[CT_ACC] && tab[CT_ACC + 2 * k + 1] == tab[CT_ACC + 1];
      // ... and no match includes the newline
      const uint32_t nl0 = tab[CT_REACH + 2 * '\n'] & tab[CT_AMASK] & tab[CT_ACC], nl1 = tab[CT_REACH + 2 * '\n' + 1] & tab[CT_AMASK + 1] & tab[CT_ACC + 1];
      uint64_t M0 = 0, M1 = 0, M2 = 0, M3 = 0, F0 = 0, F1 = 0;
      uint32_t nexc = 0, src0 = 0, src1 = 0;
      for (uint32_t i = 0; i < nnodes; i++) {
        uint64_t f = tab[CT_FOLLOW + 2 * i] | (static_cast<uint64_t>(tab[CT_FOLLOW + 2 * i + 1]) << 32);
        const uint64_t self = 1ull << i;
        if (f & self) { M0 |= self; f &= ~self; }
        if (i + 1 < 64 && (f >> (i + 1) & 1ull)) { M1 |= self; f &= ~(1ull << (i + 1)); }
        if (i + 2 < 64 && (f >> (i + 2) & 1ull)) { M2 |= self; f &= ~(1ull << (i + 2)); }
        if (i + 3 < 64 && (f >> (i + 3) & 1ull)) { M3 |= self; f &= ~(1ull << (i + 3)); }
        if (f) {
          if (nexc == 0) { src0 = i; F0 = f; }
          if (nexc == 1) { src1 = i; F1 = f; }
          nexc++;
        }
      }
      hgdev::lds_u32 *q = tab + CT_SHIFT + 16;
      q[0] = (same && !(nl0 | nl1) && nexc <= 2) ? nexc : 0xFFFFFFFFu;
      q[1] = static_cast<uint32_t>(M0); q[2] = static_cast<uint32_t>(M0 >> 32); q[3] = static_cast<uint32_t>(M1); q[4] = static_cast<uint32_t>(M1 >> 32);
      q[5] = static_cast<uint32_t>(M2); q[6] = static_cast<uint32_t>(M2 >> 32); q[7] = static_cast<uint32_t>(M3); q[8] = static_cast<uint32_t>(M3 >> 32);
      q[9] = src0; q[10] = static_cast<uint32_t>(F0); q[11] = static_cast<uint32_t>(F0 >> 32); q[12] = src1; q[13] = static_cast<uint32_t>(F1); q[14] = static_cast<uint32_t>(F1 >> 32);
    }
    // reachL2[c] = the nodes byte c may enter (0 at NUL and the newline: a line ends there)
    const uint32_t c = tid;
    tab[CT_RXA + 2 * c] = (c == 0 || c == '\n') ? 0u : (tab[CT_REACH + 2 * c] & tab[CT_AMASK]);
    tab[CT_RXA + 2 * c + 1] = (c == 0 || c == '\n') ? 0u : (tab[CT_REACH + 2 * c + 1] & tab[CT_AMASK + 1]);
    return;
  }
  __syncthreads();
  // follow unions fu[t][x] = union of follow[8t + b] over the set bits b of x; the init nodes ride in table 0
  for (uint32_t e = tid; e < 1024; e += 256) {
    const uint32_t t = e >> 8, x = e & 255u;
    uint32_t f = t == 0 ? init_word : 0u;
#pragma unroll
    for (uint32_t b = 0; b < 8; b++)
      if (((x >> b) & 1u) && 8 * t + b < nnodes) f |= tab[CT_FOLLOW + 8 * t + b];
    tab[CT_FU + e] = f;
  }
  if (tid == 0) {  // the shift form of the follow step, if the unit has one (ao_follow_lean)
    uint32_t M[4] = {0, 0, 0, 0}, nexc = 0, src[2] = {0, 0}, F[2] = {0, 0};
    for (uint32_t i = 0; i < nnodes; i++) {
      uint32_t f = tab[CT_FOLLOW + i];
      for (uint32_t k = 0; k < 4 && i + k < 32; k++)
        if (f >> (i + k) & 1u) {
          M[k] |= 1u << i;
          f &= ~(1u << (i + k));
        }
      if (f) {
        if (nexc < 2) { src[nexc] = i; F[nexc] = f; }
        nexc++;
      }
    }
    tab[CT_SHIFT + 0] = nexc <= 2 ? nexc : 0xFFFFFFFFu;
    tab[CT_SHIFT + 1] = M[1]; tab[CT_SHIFT + 2] = M[0]; tab[CT_SHIFT + 3] = M[2]; tab[CT_SHIFT + 8] = M[3];
    tab[CT_SHIFT + 4] = src[0]; tab[CT_SHIFT + 5] = F[0]; tab[CT_SHIFT + 6] = src[1]; tab[CT_SHIFT + 7] = F[1];
  }
  if (!ctx) {  // reachL
    const uint32_t c = tid, r = tab[CT_REACH + c];
    tab[CT_RXA + c] = (c == 0 || c == '\n') ? 0u : r;
    if (tid == 0) tab[CT_NL_ACCEPTS] = tab[CT_REACH + '\n'] & acc_all;
    return;
  }
  if (tid == 0) tab[CT_NL_ACCEPTS] = tab[CT_REACH + '\n'] & tab[CT_ACC + HG_PC_NL * 5 + HG_NC_END];
  for (uint32_t e = tid; e < 1024; e += 256) {
    const uint32_t pc = e >> 8, c = e & 255u;
    uint32_t rx, ax;
    if (pc == HG_PC_NL) {  // (no byte leaves this class behind: a newline starts a line.)  Its slots hold the class of every byte
      rx = ((c == '\n' || c == 0) ? static_cast<uint32_t>(HG_PC_START) : hg_prev_ctx(c)) << 11;
      ax = 0;
    } else {
      const uint32_t cc = c == '\n' ? HG_NC_NLFINAL : (hg_is_word(c) ? HG_NC_WORD : HG_NC_OTHER);
      ax = tab[CT_ACC + pc * 5 + (c == 0 ? static_cast<uint32_t>(HG_NC_END) : cc)];
      rx = (c == 0 || c == '\n') ? 0u : (tab[CT_REACH + c] & tab[CT_AMASK + pc * 4 + cc]);
    }
    tab[CT_RXA + 2 * e] = rx;
    tab[CT_RXA + 2 * e + 1] = ax;
  }
}

#ifndef HG_AO_WAVES
#define HG_AO_WAVES 4
#endif
// (more waves per SIMD measured slower: the L1 serves the lanes' strided reads only while few waves share it — 5 waves 11.8 ms
// per 8 GiB, 6 waves 14.6, against 8.8 at 4, tools/ao_ab.sh)
__attribute__((amdgpu_waves_per_eu(HG_AO_WAVES, HG_AO_WAVES)))
__global__ __launch_bounds__(256) void hg_always_on_fast_kernel(HgConfirmArgs a) {
  __shared__ uint32_t s_n;
  __shared__ __attribute__((aligned(16))) uint32_t s_tab[AO_TAB_WORDS];
  if (threadIdx.x == 0) s_n = 0;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  hgdev::lds_u32 *tab = (hgdev::lds_u32 *)(&s_tab[0]);
  // Units of the pass: the groups (several expressions in one state word, HgSlowGroup) first, then the other entries of the
  // always-on list one by one.  A unit's tables are staged once per workgroup and four tiles (one per wave).
  const uint32_t ngroups = a.db.ngroups, nunits = ngroups + (a.db.nslow_fast - a.db.nslow_grouped);
  const bool one_unit = nunits == 1;  // its tables stay staged for the whole kernel
  if (one_unit) always_on_stage(tab, a, 0, threadIdx.x);
  __syncthreads();
  // the verified-occurrence lists are free again (their confirm passes ran before this kernel): one segment per block
  const uint32_t list_cap = a.always_list_cap;
  const AlwaysOnCtx cx{a.text, a.nbytes, a.deferred + static_cast<uint64_t>(blockIdx.x) * list_cap, list_cap, (hgdev::lds_u32 *)(&s_n)};
  for (uint64_t tile0 = a.tile_begin + static_cast<uint64_t>(blockIdx.x) * 4u; tile0 < a.tile_end; tile0 += static_cast<uint64_t>(gridDim.x) * 4u) {  // block-uniform
    const uint64_t tile = tile0 + wave;
    const bool have = tile < a.tile_end;  // wave-uniform
    uint64_t lo = 0, hi = 0, line_start = 0;
    uint32_t rank_lo = 0;
    if (have) {
      const uint64_t tile_start = tile << HG_TILE_SHIFT;
      lo = tile_start + lane * 256ull < a.nbytes ? tile_start + lane * 256ull : a.nbytes;
      hi = lo + 256 < a.nbytes ? lo + 256 : a.nbytes;
      // newlines per segment -> rank of the segment start; last newline of the segment -> line start of the later segments
      // (aligned 16-byte chunks, SWAR)
      uint32_t cnt = 0, after_last = 0;  // after_last: offset in the tile just past the segment's last newline, 0 = none
      const uint32_t seg = static_cast<uint32_t>(hi - lo), seg_off = static_cast<uint32_t>(lo - tile_start);
      if (__builtin_amdgcn_ballot_w64(seg != 256u) == 0) {  // (wave-uniform) whole segments: a dword at a time, the last newline located once
        uint32_t last_m = 0, last_at = 0;
        for (uint32_t b = 0; b < 256; b += 64) {  // 64 bytes per lane at a time (see always_on_word)
          uint4 v[4];
#pragma unroll
          for (uint32_t k = 0; k < 4; k++) v[k] = *reinterpret_cast<const uint4 *>(a.text + lo + b + 16 * k);
#pragma unroll
          for (uint32_t k = 0; k < 4; k++) {
            const uint32_t dw[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
            for (uint32_t d = 0; d < 4; d++) {
              const uint32_t m = hg_newline_mask(dw[d]);
              cnt += __popc(m);
              last_at = m ? b + 16 * k + 4 * d : last_at;
              last_m = m ? m : last_m;
            }
          }
        }
        if (last_m) after_last = seg_off + last_at + (4u - (static_cast<uint32_t>(__clz(last_m)) >> 3));
      } else {
        for (uint32_t b = 0; b < 256; b += 64) {
          uint4 v[4];
#pragma unroll
          for (uint32_t k = 0; k < 4; k++) v[k] = b + 16 * k < seg ? *reinterpret_cast<const uint4 *>(a.text + lo + b + 16 * k) : make_uint4(0, 0, 0, 0);
#pragma unroll
          for (uint32_t k = 0; k < 4; k++) {
            const uint32_t at = b + 16 * k;
            uint32_t m = hgdev::eq_mask16(v[k], 0x0a0a0a0au);
            if (at + 16 > seg) m &= at < seg ? (1u << (seg - at)) - 1u : 0u;
            cnt += __popc(m);
            if (m) after_last = seg_off + at + (31 - __clz(m)) + 1;
          }
        }
      }
      rank_lo = wave_inclusive_scan(cnt, lane) - cnt;
      const HgTileBase tb = a.bases[tile];
      // start of the line that contains lo: past the last newline of the earlier segments, else the carry-in line's start
      uint32_t prev = after_last;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(prev, o, 64);
        if (lane >= static_cast<uint32_t>(o) && up > prev) prev = up;
      }
      prev = __shfl_up(prev, 1, 64);
      if (lane == 0) prev = 0;
      line_start = prev ? tile_start + prev : tb.cs;
    }
    for (uint32_t u = 0; u < nunits; u++) {  // block-uniform
      if (!one_unit) {
        __syncthreads();  // the previous unit's tables are no longer read
        always_on_stage(tab, a, u, threadIdx.x);
        __syncthreads();
      }
      if (!have) continue;
      AoUnit unit;
      bool ctx;
      if (u < ngroups) {
        const HgSlowGroup &g = a.db.groups[u];
        unit = AoUnit{g.init_word, g.acc_all, g.max_len, g.nmembers, g.single_mask, g.nnodes};
        ctx = g.ctx_off != 0;
      } else {
        const uint32_t pi = a.db.slow[a.db.nslow_grouped + (u - ngroups)];
        const HgPattern &p = a.db.patterns[pi];
        if (!p.simple && p.nw != 1) {
          always_on_word2(cx, p, pi, tab, lo, hi, line_start, a.bs1, rank_lo);
          continue;
        }
        unit = AoUnit{p.init_word, p.acc_all, p.max_len, 1u, p.single ? 1u : 0u, p.nnodes};
        ctx = !p.simple;
      }
      const uint32_t nt = (unit.nnodes + 7u) >> 3;  // follow-union tables in use (wave-uniform)
      const bool shift_form = nt >= 2 && tab[CT_SHIFT] <= 2u;
      if (shift_form) {
        if (ctx) always_on_word<true, 0>(cx, unit, tab, lo, hi, line_start, a.bs1, rank_lo);
        else always_on_word<false, 0>(cx, unit, tab, lo, hi, line_start, a.bs1, rank_lo);
      } else if (ctx) {
        if (nt <= 1) always_on_word<true, 1>(cx, unit, tab, lo, hi, line_start, a.bs1, rank_lo);
        else if (nt == 2) always_on_word<true, 2>(cx, unit, tab, lo, hi, line_start, a.bs1, rank_lo);
        else if (nt == 3) always_on_word<true, 3>(cx, unit, tab, lo, hi, line_start, a.bs1, rank_lo);
        else always_on_word<true, 4>(cx, unit, tab, lo, hi, line_start, a.bs1, rank_lo);
      } else {
        if (nt <= 1) always_on_word<false, 1>(cx, unit, tab, lo, hi, line_start, a.bs1, rank_lo);
        else if (nt == 2) always_on_word<false, 2>(cx, unit, tab, lo, hi, line_start, a.bs1, rank_lo);
        else if (nt == 3) always_on_word<false, 3>(cx, unit, tab, lo, hi, line_start, a.bs1, rank_lo);
        else always_on_word<false, 4>(cx, unit, tab, lo, hi, line_start, a.bs1, rank_lo);
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t n = s_n;
    a.always_count[blockIdx.x] = n < list_cap ? n : list_cap;
    // DEFER_NEED is in entries per list of HG_DEFER_SHARDS lists (the workspace grows to need * HG_DEFER_SHARDS entries)
    if (n > list_cap) atomicMax(&a.counters[HG_CNT_DEFER_NEED], static_cast<uint32_t>((static_cast<uint64_t>(n) * gridDim.x + HG_DEFER_SHARDS - 1) / HG_DEFER_SHARDS));
  }
}

// Matches noted by hg_always_on_fast_kernel -> hits: the line of the match's last byte (start, first scanned byte, NUL
// rules: LineHead), the end of its scanned bytes, the hit record.
__global__ __launch_bounds__(256) void hg_always_on_finish_kernel(HgConfirmArgs a) {
  __shared__ uint32_t s_n, s_base;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  const uint64_t seg0 = static_cast<uint64_t>(blockIdx.x) * a.hit_seg_cap;
  const HitSink sink{a.tmp_hits + seg0, a.tmp_aux + seg0, a.hit_seg_cap, &s_n};
  const HgDeferred *list = a.deferred + static_cast<uint64_t>(blockIdx.x) * a.always_list_cap;
  const uint32_t n = a.always_count[blockIdx.x];
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    const HgDeferred d = list[i];
    const uint64_t end = d.pos, pos = end - 1;  // the match's last byte; d.rank = newlines of the tile before it
    hgdev::PieceView pv;
    if (!hgdev::piece_view(a.text, a.nbytes, a.sums, a.bases, a.bs1, pos, d.rank, pos, &pv)) continue;
    if (!pv.whole) {  // a later piece of an over-long line: the NUL rules byte by byte (rare)
      if (pv.a > pos) continue;
      bool blocked = false;
      for (uint64_t i = pv.a; i < pos && !blocked; i++) blocked = a.text[i] == 0;
      if (blocked) continue;
    }
    const uint64_t z = a.text[pos] == '\n' ? end : hgdev::scanned_end(a.text, end, pv.limit);
    const HgPattern &pat = a.db.patterns[d.pattern];
    sink.push(a, pv.line_no, pat.id, static_cast<uint32_t>(end - pv.a), pv.a, static_cast<uint32_t>(z - pv.a), d.pattern, pat.single != 0);
  }
  flush_hits(a, &s_n, &s_base);
}

