// Tile-scan monoid and hit ordering / de-duplication rules, shared by device kernels and host tests.
#pragma once
#include "hg_core.h"

// ---- tile scan -------------------------------------------------------------------------------
// Effect of one tile (or a run of tiles) on the (carry-in line start, piece index) state.
struct HgTileElem {
  uint64_t k1;      // absolute offset just past the first '\n' of the run
  uint64_t d;       // pieces added after the carry-in line of the run has been closed
  uint64_t new_cs;  // absolute offset just past the last '\n' of the run
  uint32_t has_nl, pad;
};

HG_HD HgTileElem hg_tile_elem(const HgTileSum &s, uint64_t tile_start) {
  HgTileElem e;
  e.has_nl = s.nl_count != 0;
  e.pad = 0;
  e.k1 = e.has_nl ? tile_start + s.first_nl + 1 : 0;
  e.d = e.has_nl ? s.inner : 0;
  e.new_cs = e.has_nl ? tile_start + s.last_nl + 1 : 0;
  return e;
}
// a then b.  Lines inside each run are already priced in d; the line that crosses from a's last newline to
// b's first newline is priced here.
HG_HD HgTileElem hg_tile_combine(const HgTileElem &a, const HgTileElem &b, uint64_t bs1) {
  if (!a.has_nl) return b;
  if (!b.has_nl) return a;
  HgTileElem r;
  r.has_nl = 1;
  r.pad = 0;
  r.k1 = a.k1;
  r.d = a.d + hg_pieces(b.k1 - a.new_cs, bs1) + b.d;
  r.new_cs = b.new_cs;
  return r;
}
HG_HD HgTileBase hg_tile_apply(const HgTileBase &st, const HgTileElem &e, uint64_t bs1) {
  if (!e.has_nl) return st;
  HgTileBase r;
  r.L = st.L + hg_pieces(e.k1 - st.cs, bs1) + e.d;
  r.cs = e.new_cs;
  return r;
}

// ---- ordering and de-duplication ---------------------------------------------------------------
// Hits are ordered by (line_no, id, to, single-after-multi).  `to` < 2^31 (a piece is at most an int).
HG_HD uint64_t hg_sort_key(const HgHit &h, uint32_t single) {
  return (static_cast<uint64_t>(h.id) << 32) | (static_cast<uint64_t>(h.to) << 1) | (single ? 1u : 0u);
}
// The whole order in ONE 64-bit key when line, id and `to` fit: line | id | to | single, fields to_bits / id_bits wide.
HG_HD uint64_t hg_sort_key_packed(const HgHit &h, uint32_t single, uint32_t id_bits, uint32_t to_bits) {
  return (((h.line_no << id_bits) | h.id) << (to_bits + 1)) | (static_cast<uint64_t>(h.to) << 1) | (single ? 1u : 0u);
}
// Report rules per (line, id) — restated Hyperscan behaviour, see oracle/ohs.c header:
//  * expressions with HS_FLAG_SINGLEMATCH sharing the id yield ONE report (smallest end offset);
//  * other expressions yield every distinct end offset;
//  * identical (id, to) reports are delivered once.
// `i` indexes arrays sorted by (line_no, hg_sort_key).
// hit_at(j) / single_at(j): the j-th report in that order and whether its expression has HS_FLAG_SINGLEMATCH.
template <typename HitAt, typename SingleAt>
HG_HD bool hg_keep_hit_at(HitAt &&hit_at, SingleAt &&single_at, size_t i) {
  const HgHit h = hit_at(i);
  bool single = single_at(i);
  if (i > 0) {
    const HgHit p = hit_at(i - 1);
    if (p.line_no == h.line_no && p.id == h.id && p.to == h.to) {
      bool psingle = single_at(i - 1);
      if (!single) return false;   // twin non-single kept
      if (!psingle) return false;  // non-single with same `to` kept: report delivered once
    }
  }
  if (!single) return true;
  for (size_t j = i; j > 0; j--) {  // first single report of this (line, id)?
    const HgHit p = hit_at(j - 1);
    if (p.line_no != h.line_no || p.id != h.id) break;
    if (single_at(j - 1)) return false;
  }
  return true;
}
HG_HD bool hg_keep_hit(const HgHit *hits, const HgHitAux *aux, const HgPattern *patterns, size_t i) {
  const HgHit &h = hits[i];
  bool single = patterns[aux[i].pattern].single != 0;
  if (i > 0) {
    const HgHit &p = hits[i - 1];
    if (p.line_no == h.line_no && p.id == h.id && p.to == h.to) {
      // same report already present; the earlier one is kept unless it is a dropped single — a dropped single at
      // this `to` means an earlier single exists, and then this one (single) is dropped too; a non-single
      // duplicate is dropped because its twin (non-single, sorted first) was kept.
      bool psingle = patterns[aux[i - 1].pattern].single != 0;
      if (!single) return false;        // twin non-single kept
      if (!psingle) return false;       // non-single with same `to` kept: report delivered once
    }
  }
  if (!single) return true;
  for (size_t j = i; j > 0; j--) {  // first single report of this (line, id)?
    const HgHit &p = hits[j - 1];
    if (p.line_no != h.line_no || p.id != h.id) break;
    if (patterns[aux[j - 1].pattern].single) return false;
  }
  return true;
}
