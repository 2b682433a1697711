// Scalar building blocks shared by the gfx950 kernels (hg_kernels.hip) and the host-side unit
// tests (tests/native/hostsim.cpp compiles this header for x86 to exercise the exact same code).
// Nothing here is a CPU scan path of the product: the shipped library only calls these from device code.
//
// Line / piece rules restated from hypergrep/lib/c/hyperscanner.c:198-226 (gzgets pieces of at most
// buffer_size-1 bytes ending after '\n'; leading NULs skipped :207-214; scan stops at the first NUL
// because the shim passes strlen() :217; line_number is the 0-based piece index :225).
#pragma once
#include "hg_db.h"

constexpr uint32_t HG_TILE_BYTES = 16384;  // bytes one wavefront streams per tile (64 lanes x 16 B x 16 iterations)
constexpr uint32_t HG_TILE_SHIFT = 14;
constexpr uint32_t HG_NONE32 = 0xFFFFFFFFu;

// Newline summary of one tile, written by the stream kernel.
struct HgTileSum {
  uint32_t nl_count;  // '\n' bytes in the tile
  uint32_t first_nl;  // tile-relative offset of the first / last '\n' (HG_NONE32 if none)
  uint32_t last_nl;
  uint32_t inner;     // pieces of the lines between the first and the last '\n' (nl_count - 1 unless lines are split)
};
// Prefix state at the start of a tile, written by the tile-scan kernel.
struct HgTileBase {
  uint64_t cs;  // absolute start of the line that is open at the tile start ("carry-in" line)
  uint64_t L;   // piece index (= reference line_number) of the piece starting at cs
};
// A window-filter hit from the stream pass: the dword `word` at byte `pos` has the fingerprint of some
// pattern's required-literal window (not yet compared with the literal itself).
struct HgCand {
  uint64_t pos;
  uint32_t word;
  uint32_t rank;  // '\n' bytes in [tile start, pos)
};
// Final records (16 B + 16 B): one per (line piece, report).
struct HgHit {
  uint64_t line_no;  // 0-based piece index: hyperscanner_result_t.line_number
  uint32_t id;       // hyperscanner_result_t.id
  uint32_t to;       // match end offset inside the scanned bytes (Hyperscan's `to`)
};
struct HgHitAux {
  uint64_t start;  // absolute offset of the bytes Result.line holds
  uint32_t len;
  uint32_t pattern;
};

struct HgDbView {
  const HgPattern *patterns;
  const uint32_t *pool;
  const HgFactor *factors;
  const HgWindow *windows;
  const uint32_t *bucket_off;
  const uint16_t *disc;          // GPU verify pass: discriminated buckets (hg_db.h)
  const uint32_t *bucket_off2;
  const HgWindow *windows2;
  const HgWinBucket *wtab;       // direct window table (hg_db.h)
  uint32_t wtab_mask;
  uint32_t wtab_first;           // the verify pass asks the table first (sets whose windows mostly have ONE owner); else it goes straight to the discriminated buckets
  const uint32_t *slow;
  uint32_t npatterns, nslow, fold_mask;
  uint32_t window_mask;  // 0xFFFFFFFF, or 0x00FFFFFF for 3-byte windows
  uint32_t nslow_fast;  // the first nslow_fast entries of `slow` have <= 2 state words: hg_always_on_fast_kernel takes them
  uint32_t nslow_grouped;  // ... and the first nslow_grouped of those run as members of `groups` (HgSlowGroup), not one by one
  uint32_t nslow_huge;     // the LAST nslow_huge entries of `slow` are huge automata (hg_always_on_huge_kernel)
  uint32_t ngroups;
  const HgSlowGroup *groups;
};

// 0x80 in every byte of x that is zero, exact (no borrow between bytes).
HG_HD uint32_t hg_zero_bytes(uint32_t x) {
  uint32_t y = ((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x;
  return ~y & 0x80808080u;
}
HG_HD uint32_t hg_newline_mask(uint32_t w) { return hg_zero_bytes(w ^ 0x0a0a0a0au); }

HG_HD uint32_t hg_popc(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __popc(x);
#else
  return static_cast<uint32_t>(__builtin_popcount(x));
#endif
}
HG_HD uint32_t hg_clz32(uint32_t x) {  // x != 0
#if defined(__HIP_DEVICE_COMPILE__)
  return static_cast<uint32_t>(__clz(static_cast<int>(x)));
#else
  return static_cast<uint32_t>(__builtin_clz(x));
#endif
}
HG_HD uint32_t hg_ctz(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __ffs(x) - 1;
#else
  return static_cast<uint32_t>(__builtin_ctz(x));
#endif
}

// Which confirm routine handles a pattern:
//   0  literal-only SINGLEMATCH expression: the verified literal occurrence is the match (no automaton run)
//   1  context-free single-word SINGLEMATCH automaton (follow table in LDS)
//   2  SINGLEMATCH automaton with <= 2 state words, boundary conditions allowed
//   3  everything else of at most HG_MAX_W state words (scalar reference routine)
//   4  huge automata (sparse tables, wave-cooperative routine: hg_huge.hip)
constexpr uint32_t HG_CONFIRM_MODES = 5;
HG_HD uint32_t hg_confirm_mode(const HgPattern &p) {
  if (p.nw > HG_MAX_W) return 4;
  if (p.single && p.literal_only) return 0;
  if (p.single && p.simple) return 1;
  if (p.single && p.nw <= 2) return 2;
  return 3;
}

HG_HD uint32_t hg_prev_ctx(uint32_t c) { return c == '\n' ? HG_PC_NL : (hg_is_word(c) ? HG_PC_WORD : HG_PC_OTHER); }

// Bucket probe + literal verify for one window hit at absolute byte `pos` (dword aligned) holding raw
// dword `w`.  Calls emit(pattern, literal start, literal length) for every factor whose literal really occurs around pos.
// Range [*j0, *j1) of db.windows2 that can contain the window `folded` found at `pos`: the group's discriminator dword is
// read from the text (hg_db.h).  Empty when a discriminator byte would lie outside the text (no literal fits there).
HG_HD void hg_disc_range(const HgDbView &db, const uint8_t *text, uint64_t nbytes, uint64_t pos, uint32_t folded, uint32_t *j0, uint32_t *j1) {
  const uint32_t h = hg_hash_window(folded);
  const uint32_t d = db.disc[h];
  const uint32_t sel = d >> 8;
  uint32_t key = 0;
  *j0 = *j1 = 0;
  if (sel) {
    const int64_t at = static_cast<int64_t>(pos) + static_cast<int8_t>(d & 0xFFu);
    const uint32_t lo = hg_ctz(sel), hi = 31u - hg_clz32(sel);  // first / last selected byte
    if (at + static_cast<int64_t>(lo) < 0 || static_cast<uint64_t>(at + hi) >= nbytes) return;
    uint32_t v = 0;
    for (uint32_t b = lo; b <= hi; b++) v |= static_cast<uint32_t>(text[at + b]) << (8 * b);
    key = (v | db.fold_mask) & hg_disc_bytes(sel);
  }
  const uint32_t h2 = hg_disc_bucket(h, key);
  *j0 = db.bucket_off2[h2];
  *j1 = db.bucket_off2[h2 + 1];
}

template <typename Emit>
HG_HD void hg_verify_window(const HgDbView &db, const uint8_t *text, uint64_t nbytes, uint64_t pos, uint32_t w,
                            Emit &&emit) {
  uint32_t folded = (w | db.fold_mask) & db.window_mask;
  auto check = [&](uint32_t factor_off) {
    uint32_t off = factor_off & 0xff;
    const HgFactor &f = db.factors[factor_off >> 8];
    if (pos < off) return;
    uint64_t start = pos - off;
    if (start + f.len > nbytes) return;
    for (uint32_t b = 0; b < f.len; b++)
      if ((text[start + b] ^ f.lit[b]) & hg_factor_cmask(f, b)) return;
    emit(f.pattern, start, f.len);
  };
  // the direct table first: a value that is not in it belongs to no literal, a value with one owner names it
  const uint32_t owner = hg_wtab_find(db.wtab, db.wtab_mask, folded);
  if (owner == HG_WTAB_EMPTY) return;
  if (owner != HG_WTAB_SHARED) {
    check(owner);
    return;
  }
  uint32_t j0, j1;
  hg_disc_range(db, text, nbytes, pos, folded, &j0, &j1);
  for (uint32_t j = j0; j < j1; j++) {
    HgWindow win = db.windows2[j];
    if (win.value != folded) continue;
    check(win.factor_off);
  }
}

// Pieces of the carry-in line of a tile: ceil(len / bs1), len >= 1.
HG_HD uint64_t hg_pieces(uint64_t len, uint64_t bs1) { return (len + bs1 - 1) / bs1; }

// Pieces of the whole lines lying in [from, to): both are line starts (to is just past a '\n').
HG_HD uint64_t hg_inner_pieces(const uint8_t *text, uint64_t from, uint64_t to, uint64_t bs1) {
  uint64_t total = 0, start = from;
  for (uint64_t i = from; i < to; i++)
    if (text[i] == '\n') {
      total += hg_pieces(i + 1 - start, bs1);
      start = i + 1;
    }
  return total;
}

// Piece index of the line that starts at absolute offset `s`, which has `rank` newlines before it in
// tile `t` (rank == 0: the carry-in line itself).  When bs1 >= HG_TILE_BYTES every line lying inside one
// tile is a single piece and the answer is arithmetic; otherwise (`small`) the lines between the tile's
// first newline and `s` are walked.
HG_HD uint64_t hg_line_index(const uint8_t *text, const HgTileSum &ts, const HgTileBase &tb, uint64_t tile_start, uint32_t rank,
                             uint64_t s, uint64_t bs1, bool small) {
  if (rank == 0) return tb.L;
  uint64_t after_first = tile_start + ts.first_nl + 1;
  uint64_t base = tb.L + hg_pieces(after_first - tb.cs, bs1);
  if (!small) return base + (rank - 1);
  return base + hg_inner_pieces(text, after_first, s, bs1);
}

// Bytes hs_scan would see for the piece [ps, limit): skip leading NULs, cut at the first NUL or after the
// first '\n'.  Returns [a, z).
HG_HD void hg_trim_piece(const uint8_t *text, uint64_t ps, uint64_t limit, uint64_t &a, uint64_t &z) {
  a = ps;
  while (a < limit && text[a] == 0) a++;
  z = a;
  while (z < limit) {
    uint32_t c = text[z];
    if (c == 0) break;
    z++;
    if (c == '\n') break;
  }
}

// ---- huge automata (nw > HG_MAX_W): sparse tables (HgHugeHeader, hg_db.h) -------------------------------------------------
struct HgHugeView {
  uint32_t nw, ncls, ctxfree, init_hi;
  const uint32_t *cls;    // 64 words: the class of each byte value, one byte each
  const uint32_t *reach;  // reach[ncls][nw]
  const uint32_t *init, *smask, *xsrc, *xrank, *xlist, *xt;
  const uint32_t *amask;  // amask[4][4][nw]   (unless ctxfree)
  const uint32_t *acc;    // acc[4][5][nw], or acc[nw] when ctxfree
};
HG_HD HgHugeView hg_huge_view(const uint32_t *pool, const HgPattern &p) {
  const HgHugeHeader *h = reinterpret_cast<const HgHugeHeader *>(pool + p.follow_off);
  HgHugeView v;
  v.nw = p.nw;
  v.ncls = h->ncls;
  v.ctxfree = h->ctxfree;
  v.init_hi = h->init_hi;
  v.cls = pool + h->cls_off;
  v.reach = pool + p.reach_off;
  v.init = pool + p.init_off;
  v.smask = pool + h->smask_off;
  v.xsrc = pool + h->xsrc_off;
  v.xrank = pool + h->xrank_off;
  v.xlist = pool + h->xlist_off;
  v.xt = pool + h->xt_off;
  v.amask = pool + p.amask_off;
  v.acc = pool + p.acc_off;
  return v;
}
HG_HD uint32_t hg_huge_class(const HgHugeView &v, uint32_t c) { return (v.cls[c >> 2] >> ((c & 3u) * 8u)) & 0xFFu; }
// the accepting-node mask for (context of the previous byte, context of the next) and the entry mask for (previous, own byte)
HG_HD const uint32_t *hg_huge_acc(const HgHugeView &v, uint32_t pc, uint32_t nc) { return v.ctxfree ? v.acc : v.acc + (pc * 5 + nc) * v.nw; }
// bits [lo & 31, 31] and [0, hi & 31] of a word
HG_HD uint32_t hg_bits_from(uint32_t lo) { return 0xFFFFFFFFu << (lo & 31u); }
HG_HD uint32_t hg_bits_upto(uint32_t hi) { return 0xFFFFFFFFu >> (31u - (hi & 31u)); }

#if !defined(__HIP_DEVICE_COMPILE__)
// HOST mirror of the wave-cooperative device routine (hg_huge.hip huge_run): the same tables, the same step, one word at a
// time.  The tests replay it against the oracle (tests/native/hostsim.cpp); the product never runs it.
template <typename Emit>
inline void hg_huge_scan_slice(const uint32_t *pool, const HgPattern &p, const uint8_t *data, uint64_t len, uint64_t from, uint64_t upto, Emit &&emit) {
  const HgHugeView v = hg_huge_view(pool, p);
  const uint32_t nw = v.nw;
  const bool single = p.single != 0;
  uint32_t *S = new uint32_t[2 * static_cast<size_t>(nw)](), *T = S + nw;
  uint32_t pc = from ? hg_prev_ctx(data[from - 1]) : HG_PC_START;
  bool alive = false;
  auto any_accept = [&](const uint32_t *a) {
    uint32_t any = 0;
    for (uint32_t w = 0; w < nw; w++) any |= S[w] & a[w];
    return any != 0;
  };
  for (uint64_t i = from; i < len; i++) {
    if (i >= upto && !alive) { delete[] S; return; }
    const uint32_t c = data[i];
    const uint32_t cc = c == '\n' ? (i + 1 == len ? HG_NC_NLFINAL : HG_NC_NL) : (hg_is_word(c) ? HG_NC_WORD : HG_NC_OTHER);
    if (alive && any_accept(hg_huge_acc(v, pc, cc))) {
      emit(static_cast<uint32_t>(i));
      if (single) { delete[] S; return; }
    }
    for (uint32_t w = 0; w < nw; w++) T[w] = i < upto ? v.init[w] : 0u;
    for (uint32_t w = 0; w < nw; w++) {
      const uint32_t x = S[w] & v.smask[w];
      T[w] |= x << 1;
      if (w + 1 < nw) T[w + 1] |= x >> 31;
    }
    for (uint32_t w = 0; w < nw; w++)
      for (uint32_t x = S[w] & v.xsrc[w]; x; x &= x - 1) {
        const uint32_t b = hg_ctz(x), k = v.xrank[w] + hg_popc(v.xsrc[w] & ((1u << b) - 1u));
        for (uint32_t r = v.xlist[k]; r < v.xlist[k + 1]; r++) {
          const uint32_t lo = v.xt[2 * r], hi = v.xt[2 * r + 1];
          for (uint32_t tw = lo >> 5; tw <= (hi >> 5); tw++)
            T[tw] |= (tw == (lo >> 5) ? hg_bits_from(lo) : 0xFFFFFFFFu) & (tw == (hi >> 5) ? hg_bits_upto(hi) : 0xFFFFFFFFu);
        }
      }
    const uint32_t *r = v.reach + hg_huge_class(v, c) * nw, *m = v.ctxfree ? nullptr : v.amask + (pc * 4 + cc) * nw;
    uint32_t any = 0;
    for (uint32_t w = 0; w < nw; w++) any |= S[w] = T[w] & r[w] & (m ? m[w] : 0xFFFFFFFFu);
    alive = any != 0;
    pc = hg_prev_ctx(c);
  }
  if (alive && any_accept(hg_huge_acc(v, pc, HG_NC_END))) emit(static_cast<uint32_t>(len));
  delete[] S;
}
#endif

// Run one pattern's automaton over data[0, len) (one trimmed piece).  emit(to) per distinct match end
// offset in ascending order; returns after the first when `single`.
template <typename Emit>
HG_HD void hg_nfa_scan(const uint32_t *pool, const HgPattern &p, const uint8_t *data, uint64_t len, Emit &&emit) {
  const uint32_t nw = p.nw;
  const uint32_t *reach = pool + p.reach_off, *follow = pool + p.follow_off, *init = pool + p.init_off;
  const uint32_t *amask = pool + p.amask_off, *acc = pool + p.acc_off;
  const bool single = p.single != 0;
  if (nw > HG_MAX_W) {  // a huge automaton: the device runs these wave-cooperatively (hg_huge.hip), never through this routine
#if !defined(__HIP_DEVICE_COMPILE__)
    hg_huge_scan_slice(pool, p, data, len, 0, len, emit);
#endif
    return;
  }
  uint32_t pc = HG_PC_START;
  if (nw == 1) {
    uint32_t S = 0;
    const uint32_t init0 = init[0];
    for (uint64_t i = 0; i < len; i++) {
      uint32_t c = data[i];
      uint32_t cc = c == '\n' ? (i + 1 == len ? HG_NC_NLFINAL : HG_NC_NL) : (hg_is_word(c) ? HG_NC_WORD : HG_NC_OTHER);
      if (S & acc[pc * 5 + cc]) {
        emit(static_cast<uint32_t>(i));
        if (single) return;
      }
      uint32_t T = init0;
      for (uint32_t x = S; x; x &= x - 1) T |= follow[hg_ctz(x)];
      S = T & reach[c] & amask[pc * 4 + cc];
      pc = hg_prev_ctx(c);
    }
    if (S & acc[pc * 5 + HG_NC_END]) emit(static_cast<uint32_t>(len));
    return;
  }
  uint32_t S[HG_MAX_W], T[HG_MAX_W];
  for (uint32_t w = 0; w < nw; w++) S[w] = 0;
  for (uint64_t i = 0; i < len; i++) {
    uint32_t c = data[i];
    uint32_t cc = c == '\n' ? (i + 1 == len ? HG_NC_NLFINAL : HG_NC_NL) : (hg_is_word(c) ? HG_NC_WORD : HG_NC_OTHER);
    const uint32_t *a = acc + (pc * 5 + cc) * nw;
    uint32_t any = 0;
    for (uint32_t w = 0; w < nw; w++) any |= S[w] & a[w];
    if (any) {
      emit(static_cast<uint32_t>(i));
      if (single) return;
    }
    for (uint32_t w = 0; w < nw; w++) T[w] = init[w];
    for (uint32_t w = 0; w < nw; w++)
      for (uint32_t x = S[w]; x; x &= x - 1) {
        const uint32_t *f = follow + (w * 32 + hg_ctz(x)) * nw;
        for (uint32_t k = 0; k < nw; k++) T[k] |= f[k];
      }
    const uint32_t *r = reach + c * nw, *m = amask + (pc * 4 + cc) * nw;
    for (uint32_t w = 0; w < nw; w++) S[w] = T[w] & r[w] & m[w];
    pc = hg_prev_ctx(c);
  }
  const uint32_t *a = acc + (pc * 5 + HG_NC_END) * nw;
  uint32_t any = 0;
  for (uint32_t w = 0; w < nw; w++) any |= S[w] & a[w];
  if (any) emit(static_cast<uint32_t>(len));
}

// The same automaton restricted to matches that START in data[from, upto): the step S' = (init | follow(S)) & reach[c] & mask
// is linear in (init, S), so the start states are injected only at those bytes, the left context at `from` comes from
// data[from - 1], and the run ends as soon as no state is alive past `upto`.  emit(to) per distinct end of such a match
// (ascending); returns after the first when `single`.  Used where one scan unit is split by match start (hg_block_small_kernel)
// and by the windowed confirm (hg_confirm_window below, hg_confirm_dev.h on the device).
template <typename Emit>
HG_HD void hg_nfa_scan_slice(const uint32_t *pool, const HgPattern &p, const uint8_t *data, uint32_t len, uint32_t from, uint32_t upto, Emit &&emit) {
  const uint32_t nw = p.nw;
  const uint32_t *reach = pool + p.reach_off, *follow = pool + p.follow_off, *init = pool + p.init_off;
  const uint32_t *amask = pool + p.amask_off, *acc = pool + p.acc_off;
  const bool single = p.single != 0;
  if (nw > HG_MAX_W) {
#if !defined(__HIP_DEVICE_COMPILE__)
    hg_huge_scan_slice(pool, p, data, len, from, upto, emit);
#endif
    return;
  }
  uint32_t pc = from ? hg_prev_ctx(data[from - 1]) : HG_PC_START;
  if (nw == 1) {
    uint32_t S = 0;
    const uint32_t init0 = init[0];
    for (uint32_t i = from; i < len; i++) {
      if (i >= upto && S == 0) return;  // no start left and nothing alive
      const uint32_t c = data[i];
      const uint32_t cc = c == '\n' ? (i + 1 == len ? HG_NC_NLFINAL : HG_NC_NL) : (hg_is_word(c) ? HG_NC_WORD : HG_NC_OTHER);
      if (S & acc[pc * 5 + cc]) {
        emit(i);
        if (single) return;
      }
      uint32_t T = i < upto ? init0 : 0u;
      for (uint32_t x = S; x; x &= x - 1) T |= follow[hg_ctz(x)];
      S = T & reach[c] & amask[pc * 4 + cc];
      pc = hg_prev_ctx(c);
    }
    if (S & acc[pc * 5 + HG_NC_END]) emit(len);
    return;
  }
  uint32_t S[HG_MAX_W], T[HG_MAX_W];
  for (uint32_t w = 0; w < nw; w++) S[w] = 0;
  uint32_t alive = 0;
  for (uint32_t i = from; i < len; i++) {
    if (i >= upto && alive == 0) return;
    const uint32_t c = data[i];
    const uint32_t cc = c == '\n' ? (i + 1 == len ? HG_NC_NLFINAL : HG_NC_NL) : (hg_is_word(c) ? HG_NC_WORD : HG_NC_OTHER);
    const uint32_t *a = acc + (pc * 5 + cc) * nw;
    uint32_t any = 0;
    for (uint32_t w = 0; w < nw; w++) any |= S[w] & a[w];
    if (any) {
      emit(i);
      if (single) return;
    }
    for (uint32_t w = 0; w < nw; w++) T[w] = i < upto ? init[w] : 0u;
    for (uint32_t w = 0; w < nw; w++)
      for (uint32_t x = S[w]; x; x &= x - 1) {
        const uint32_t *f = follow + (w * 32 + hg_ctz(x)) * nw;
        for (uint32_t k = 0; k < nw; k++) T[k] |= f[k];
      }
    const uint32_t *r = reach + c * nw, *m = amask + (pc * 4 + cc) * nw;
    alive = 0;
    for (uint32_t w = 0; w < nw; w++) alive |= S[w] = T[w] & r[w] & m[w];
    pc = hg_prev_ctx(c);
  }
  const uint32_t *a = acc + (pc * 5 + HG_NC_END) * nw;
  uint32_t any = 0;
  for (uint32_t w = 0; w < nw; w++) any |= S[w] & a[w];
  if (any) emit(len);
}


// Confirm by WINDOW: what the device's confirm routines for SINGLEMATCH automata compute (hg_confirm_dev.h has the argument).
// Every match contains an occurrence of the pattern's required literal that begins at most `lit_lead` bytes after the
// match's start; the candidate whose verified occurrence begins at `fs` answers for the matches that start in
// [fs - lit_lead, fs].  emit(line_no, to, a, len) for the smallest end of such a match.
template <typename Emit>
HG_HD void hg_confirm_window(const HgDbView &db, const uint8_t *text, uint64_t nbytes, const HgTileSum *sums, const HgTileBase *bases, uint64_t bs1, uint64_t pos,
                             uint64_t fs, uint32_t pattern, uint32_t rank, Emit &&emit) {
  uint64_t t = pos >> HG_TILE_SHIFT, tile_start = t << HG_TILE_SHIFT;
  uint64_t s;
  if (rank == 0) {
    s = bases[t].cs;
  } else {
    s = pos;  // the previous '\n' lies inside this tile
    while (s > tile_start && text[s - 1] != '\n') s--;
  }
  uint64_t k = (pos - s) / bs1;
  uint64_t ps = s + k * bs1;
  uint64_t line_no = hg_line_index(text, sums[t], bases[t], tile_start, rank, s, bs1, bs1 < HG_TILE_BYTES) + k;
  uint64_t limit = ps + bs1 < nbytes ? ps + bs1 : nbytes;
  uint64_t a, z;
  hg_trim_piece(text, ps, limit, a, z);
  if (z <= a || pos < a || pos >= z || fs < a) return;  // the occurrence does not lie in the scanned bytes
  const HgPattern &p = db.patterns[pattern];
  const uint64_t q = (p.lit_lead != 0xFFFFFFFFu && fs - a > p.lit_lead) ? fs - p.lit_lead : a;
  hg_nfa_scan_slice(db.pool, p, text + a, static_cast<uint32_t>(z - a), static_cast<uint32_t>(q - a), static_cast<uint32_t>(fs - a + 1),
                    [&](uint32_t to) { emit(line_no, to, a, static_cast<uint32_t>(z - a)); });
}

// Confirm one candidate: locate the piece containing byte `pos`, trim it, run the pattern.
// emit(line_no, to, a, len) per report.
// claim(piece start) -> false: the piece is not run (the caller has seen this (expression, piece) before).
template <typename Emit, typename Claim>
HG_HD void hg_confirm(const HgDbView &db, const uint8_t *text, uint64_t nbytes, const HgTileSum *sums,
                      const HgTileBase *bases, uint64_t bs1, uint64_t pos, uint32_t pattern, uint32_t rank, Emit &&emit, Claim &&claim) {
  uint64_t t = pos >> HG_TILE_SHIFT, tile_start = t << HG_TILE_SHIFT;
  uint64_t s;
  if (rank == 0) {
    s = bases[t].cs;
  } else {
    s = pos;  // the previous '\n' lies inside this tile
    while (s > tile_start && text[s - 1] != '\n') s--;
  }
  uint64_t k = (pos - s) / bs1;
  uint64_t ps = s + k * bs1;
  uint64_t line_no = hg_line_index(text, sums[t], bases[t], tile_start, rank, s, bs1, bs1 < HG_TILE_BYTES) + k;
  uint64_t limit = ps + bs1 < nbytes ? ps + bs1 : nbytes;
  uint64_t a, z;
  hg_trim_piece(text, ps, limit, a, z);
  if (z <= a) return;
  if (!claim(ps)) return;
  const HgPattern &p = db.patterns[pattern];
  hg_nfa_scan(db.pool, p, text + a, z - a, [&](uint32_t to) { emit(line_no, to, a, static_cast<uint32_t>(z - a)); });
}
template <typename Emit>
HG_HD void hg_confirm(const HgDbView &db, const uint8_t *text, uint64_t nbytes, const HgTileSum *sums,
                      const HgTileBase *bases, uint64_t bs1, uint64_t pos, uint32_t pattern, uint32_t rank, Emit &&emit) {
  hg_confirm(db, text, nbytes, sums, bases, bs1, pos, pattern, rank, emit, [](uint64_t) { return true; });
}

// Always-on tier: process every piece of the line that starts at absolute offset `s` (which has `rank`
// newlines before it in its tile) with every tier-1 pattern.  emit(pattern, line_no, to, a, len).
template <typename Emit>
HG_HD void hg_scan_line_always_on(const HgDbView &db, const uint8_t *text, uint64_t nbytes, const HgTileSum *sums,
                                  const HgTileBase *bases, uint64_t bs1, uint64_t s, uint32_t rank, uint32_t first, uint32_t last, Emit &&emit) {
  uint64_t t = s >> HG_TILE_SHIFT, tile_start = t << HG_TILE_SHIFT;
  uint64_t line_no = hg_line_index(text, sums[t], bases[t], tile_start, rank, s, bs1, bs1 < HG_TILE_BYTES);
  uint64_t ps = s;
  for (;;) {
    uint64_t limit = ps + bs1 < nbytes ? ps + bs1 : nbytes;
    uint64_t a, z;
    hg_trim_piece(text, ps, limit, a, z);
    if (z > a) {
      for (uint32_t j = first; j < last; j++) {  // entries [first, last) of the always-on list
        uint32_t pi = db.slow[j];
        hg_nfa_scan(db.pool, db.patterns[pi], text + a, z - a,
                    [&](uint32_t to) { emit(pi, line_no, to, a, static_cast<uint32_t>(z - a)); });
      }
    }
    // where does the piece end?  after its '\n', else at limit
    uint64_t e = ps;
    bool nl = false;
    while (e < limit) {
      if (text[e++] == '\n') { nl = true; break; }
    }
    if (nl || e >= nbytes) return;
    ps = e;  // forced break: the line continues as the next piece
    line_no++;
  }
}
