// TEST-ONLY host harness (never shipped, never loaded by the product): compiles the product's pattern
// compiler and hg_core.h for x86 and replays the GPU pipeline's stages in scalar form —
//   stream pass (newline summaries + window fingerprint filter + literal verify)  ~ hg_stream_kernel
//   tile scan                                                          ~ hg_tile_scan_*
//   confirm / always-on                                                ~ hg_confirm_kernel / hg_always_on_kernel
//   order + dedupe                                                     ~ hg_finalize
// so the compiler's tables and the shared device logic can be checked against the oracle without a GPU.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "../../hypergrep_amd/csrc/hg_compile.h"
#include "../../hypergrep_amd/csrc/hg_core.h"
#include "../../hypergrep_amd/csrc/hg_post.h"

extern "C" {

struct SimHit {
  uint64_t line_no;
  uint32_t id, to;
  uint64_t start;
  uint32_t len, pattern;
};

void *hgsim_compile(const char *const *exprs, const unsigned *flags, const unsigned *ids, unsigned n, char *err, size_t errlen) {
  HgDb *db = nullptr;
  std::string e;
  int bad = -1;
  if (hgc_compile(exprs, flags, ids, n, &db, &e, &bad) != 0) {
    if (err && errlen) snprintf(err, errlen, "%d: %s", bad, e.c_str());
    return nullptr;
  }
  return db;
}
int hgsim_tune(void *h, const uint8_t *sample, size_t n) {  // the handle keeps its identity: the tuned copy's tables move in
  std::string e;
  HgDb *tuned = nullptr;
  int rc = hgc_tune(static_cast<const HgDb *>(h), sample, n, &tuned, &e);
  if (rc != 0) return rc;
  *static_cast<HgDb *>(h) = std::move(*tuned);
  hgc_free(tuned);
  return 0;
}
void hgsim_free(void *h) { hgc_free(static_cast<HgDb *>(h)); }

void hgsim_info(void *h, uint32_t *out) {  // npatterns, nfactors, nwindows, nslow, fold_mask, max_nw
  HgDb *db = static_cast<HgDb *>(h);
  out[0] = db->patterns.size();
  out[1] = db->nreal_factors;
  out[2] = db->nreal_factors ? db->windows.size() : 0;
  out[3] = db->slow.size();
  out[4] = db->fold_mask;
  out[5] = db->max_nw;
}
static HgDbView view_of(HgDb *db);
// Invariants of the compiled prefilter, checked on the tables themselves (no text): every window of every literal
//  (1) passes the first level at its own slot, (2) passes the second level when the literal's own bytes surround it,
//  (3) is found by the verify pass's discriminated bucket lookup when the literal itself is the text.
// Returns the number of violations; out[0] = filter log2, out[1] = wide, out[2] = slots holding more than two values.
uint32_t hgsim_selfcheck(void *h, uint32_t *out) {  // out[6]
  HgDb *db = static_cast<HgDb *>(h);
  const HgDbView v = view_of(db);
  uint32_t bad = 0, many = 0;
  const uint32_t byte_mask = ((1u << db->filter_log2) - 1u) << 2;
  if (!db->filter_wide)
    for (const HgSlotInfo &s : db->ext) many += s.many ? 1 : 0;
  for (size_t wi = 0; wi < db->windows.size() && db->nreal_factors; wi++) {
    const HgWindow &w = db->windows[wi];
    const HgFactor &f = db->factors[w.factor_off >> 8];
    const uint32_t off = w.factor_off & 0xff;
    const uint32_t key = hg_hash_window(w.value);
    // the literal as a text of its own, lower-cased where it is case-insensitive (what folding maps both cases to)
    std::vector<uint8_t> text(f.lit, f.lit + f.len);
    if (f.len < db->window_bytes) text.push_back(static_cast<uint8_t>(w.value >> 24));  // enumerated byte after a 3-byte literal
    auto dword_at = [&](int64_t p) {
      uint32_t x = 0;
      for (int b = 0; b < 4; b++)
        if (p + b >= 0 && p + b < static_cast<int64_t>(text.size())) x |= static_cast<uint32_t>(text[p + b]) << (8 * b);
      return x;
    };
    if (db->filter_wide) {
      const uint32_t a = hg_dot4(w.value, db->weights_a), b = hg_dot4(w.value, db->weights_b), fp = key & 0xFFFFu;
      const uint32_t ta = db->filter[hg_slot_wide(a, b, byte_mask) >> 2], tb = db->filter[hg_slot_wide(b, a, byte_mask) >> 2];
      if (!((ta & 0xFFFFu) == fp || (ta >> 16) == fp || (tb & 0xFFFFu) == fp || (tb >> 16) == fp)) bad++;
    } else {
      const uint32_t sl = hg_slot(w.value, db->weights_a, byte_mask) >> 2;
      if (!hg_slot_match(db->filter[sl], key)) bad++;
      const uint32_t prev = dword_at(static_cast<int64_t>(off) - 4) | db->fold_mask, next = dword_at(static_cast<int64_t>(off) + db->window_bytes) | db->fold_mask;
      // bytes outside the literal are unknown in a real text: try both extremes
      for (uint32_t fill : {0u, 0xFFFFFFFFu}) {
        uint32_t p2 = prev, n2 = next;
        for (int b = 0; b < 4; b++) {
          if (static_cast<int>(off) - 4 + b < 0) p2 = (p2 & ~(0xFFu << (8 * b))) | (fill & (0xFFu << (8 * b)));
          if (off + db->window_bytes + b >= f.len) n2 = (n2 & ~(0xFFu << (8 * b))) | (fill & (0xFFu << (8 * b)));
        }
        if (!hg_slot_pass(db->ext[sl], w.value, p2 | db->fold_mask, n2 | db->fold_mask, 0xFFFFFFFFu, 0xFFFFFFFFu)) bad++;
      }
    }
    uint32_t j0, j1;
    hg_disc_range(v, text.data(), text.size(), off, w.value, &j0, &j1);
    bool found = false;
    for (uint32_t j = j0; j < j1; j++) found = found || (db->windows2[j].value == w.value && db->windows2[j].factor_off == w.factor_off);
    if (!found) bad++;
    // (4) the direct table holds the window's value; with one owner it names exactly this (literal, offset)
    const uint32_t owner = hg_wtab_find(db->wtab.data(), db->wtab_mask, w.value);
    if (owner == HG_WTAB_EMPTY || (owner != HG_WTAB_SHARED && owner != w.factor_off)) bad++;
  }
  if (out) { out[0] = db->filter_log2; out[1] = db->filter_wide; out[2] = many; out[3] = db->dense; }
  // the table's buckets fill front to back and are never more than half full over all
  size_t used = 0;
  for (const HgWinBucket &b : db->wtab)
    for (uint32_t k = 0; k < HG_WTAB_WAYS; k++) {
      used += b.factor_off[k] != HG_WTAB_EMPTY ? 1 : 0;
      if (k && b.factor_off[k] != HG_WTAB_EMPTY && b.factor_off[k - 1] == HG_WTAB_EMPTY) bad++;
    }
  if (used * 2 > db->wtab.size() * HG_WTAB_WAYS || db->wtab.size() != static_cast<size_t>(db->wtab_mask) + 1) bad++;
  if (out) { out[4] = db->shared_windows; out[5] = db->wtab_first; }
  return bad;
}
uint32_t hgsim_pattern_tier(void *h, uint32_t i) { return static_cast<HgDb *>(h)->patterns[i].tier; }
uint32_t hgsim_pattern_nodes(void *h, uint32_t i) { return static_cast<HgDb *>(h)->patterns[i].nnodes; }

// Run one pattern's automaton over one already-trimmed piece; returns number of reports, fills tos[<=cap].
size_t hgsim_nfa(void *h, uint32_t pattern, const uint8_t *data, size_t len, uint32_t *tos, size_t cap) {
  HgDb *db = static_cast<HgDb *>(h);
  size_t n = 0;
  hg_nfa_scan(db->pool.data(), db->patterns[pattern], data, len, [&](uint32_t to) {
    if (n < cap) tos[n] = to;
    n++;
  });
  return n;
}

static HgDbView view_of(HgDb *db) {
  HgDbView v{};
  v.patterns = db->patterns.data();
  v.pool = db->pool.data();
  v.factors = db->factors.data();
  v.windows = db->windows.data();
  v.bucket_off = db->bucket_off.data();
  v.disc = db->disc.data();
  v.bucket_off2 = db->bucket_off2.data();
  v.windows2 = db->windows2.data();
  v.wtab = db->wtab.data();
  v.wtab_mask = db->wtab_mask;
  v.wtab_first = db->wtab_first;
  v.slow = db->slow.data();
  v.npatterns = db->patterns.size();
  v.nslow = db->slow.size();
  v.nslow_fast = db->nslow_fast;
  v.nslow_grouped = 0;  // (the host replay runs every always-on expression on its own)
  v.nslow_huge = db->nslow_huge;
  v.ngroups = 0;
  v.groups = nullptr;
  v.fold_mask = db->fold_mask;
  v.window_mask = db->window_mask;
  return v;
}

// Whole pipeline on a memory buffer.  Returns number of hits (after dedupe); *out is malloc'ed.
// stats[4] = hits surviving the neighbour-dword check; stats[0] = window fingerprint filter hits, stats[1] = verified candidates, stats[2] = raw hits before dedupe, stats[3] = pieces
long hgsim_scan(void *h, const uint8_t *data, uint64_t nbytes, int buffer_size, SimHit **out, uint64_t *stats) {
  HgDb *db = static_cast<HgDb *>(h);
  HgDbView v = view_of(db);
  if (buffer_size < 2) { *out = nullptr; return 0; }
  uint64_t bs1 = static_cast<uint64_t>(buffer_size) - 1;
  uint64_t ntiles = (nbytes + HG_TILE_BYTES - 1) / HG_TILE_BYTES;
  std::vector<HgTileSum> sums(ntiles ? ntiles : 1);
  std::vector<HgTileBase> bases(ntiles + 1);
  std::vector<HgCand> cands;
  uint64_t bitmap_hits = 0, level2_hits = 0;
  std::map<uint32_t, uint64_t> l1_hist;
  // ---- stream pass
  for (uint64_t t = 0; t < ntiles; t++) {
    uint64_t base = t * HG_TILE_BYTES;
    HgTileSum s{0, HG_NONE32, HG_NONE32, 0};
    const uint32_t step = 1;  // newlines are counted byte by byte; windows are probed every `probe_step` bytes
    const uint32_t probe_step = db->dense ? db->dense : 4;
    for (uint32_t d = 0; d < HG_TILE_BYTES / step; d++) {
      uint64_t pos = base + static_cast<uint64_t>(d) * step;
      if (pos >= nbytes) break;
      uint32_t w = 0;
      uint64_t avail = nbytes - pos < 4 ? nbytes - pos : 4;
      std::memcpy(&w, data + pos, avail);  // bytes past the end read as zero, as the kernel masks them
      uint32_t rank_here = s.nl_count;
      uint32_t m = hg_newline_mask(w);
      m &= 0x80u;  // only the byte at pos itself
      if (m) {
        for (uint32_t b = 0; b < 4; b++)
          if (m >> (8 * b + 7) & 1) {
            if (s.first_nl == HG_NONE32) s.first_nl = d * step + b;
            s.last_nl = d * step + b;
          }
        s.nl_count += hg_popc(m);
      }
      if (pos % probe_step) continue;  // windows start every probe_step bytes (absolute offsets: chunks are 16-byte aligned)
      const uint32_t folded = (w | v.fold_mask) & db->window_mask, byte_mask = ((1u << db->filter_log2) - 1u) << 2;
      // bytes outside the text read as zero, like the kernel's masked tail
      auto dword_at = [&](int64_t p) -> uint32_t {
        uint32_t x = 0;
        for (int b = 0; b < 4; b++)
          if (p + b >= 0 && static_cast<uint64_t>(p + b) < nbytes) x |= static_cast<uint32_t>(data[p + b]) << (8 * b);
        return x;
      };
      // first level: window hash
      const uint32_t key0 = hg_hash_window(folded);
      uint32_t sla = hg_slot(folded, db->weights_a, byte_mask) >> 2, slb = hg_slot(folded, db->weights_b, byte_mask) >> 2;
      if (db->filter_wide) {
        const uint32_t a = hg_dot4(folded, db->weights_a), b = hg_dot4(folded, db->weights_b);
        sla = hg_slot_wide(a, b, byte_mask) >> 2, slb = hg_slot_wide(b, a, byte_mask) >> 2;
      }
      bool ha = hg_slot_match(db->filter[sla], key0), hb = false;  // single probe (slot A)
      if (db->filter_wide) {
        const uint32_t fp = key0 & 0xFFFFu, ta = db->filter[sla], tb = db->filter[slb];
        ha = (ta & 0xFFFFu) == fp || (ta >> 16) == fp;
        hb = (tb & 0xFFFFu) == fp || (tb >> 16) == fp;
      }
      if (ha || hb) {
        bitmap_hits++;
        if (getenv("HGSIM_DUMP")) l1_hist[folded]++;
        const uint32_t pf = dword_at(static_cast<int64_t>(pos) - 4) | v.fold_mask, nf = dword_at(static_cast<int64_t>(pos) + db->window_bytes) | v.fold_mask;
        // the kernel cannot see across its 16 KiB tile edge or the first/last lane of a 1 KiB segment: treat as pass there
        const bool edge_prev = !db->dense && (pos % 1024) == 0, edge_next = !db->dense && (pos % 1024) == 1020;
        auto pass = [&](uint32_t sl) {
          // the last lane sees at most its own dword's top byte
          return hg_slot_pass(db->ext[sl], folded, pf, nf, edge_prev ? 0u : 0xFFFFFFFFu, edge_next ? (HG_WINDOW_BYTES == 4 ? 0u : 0xFFu) : 0xFFFFFFFFu);
        };
        if (!db->filter_wide && !((ha && pass(sla)) || (hb && pass(slb)))) continue;
        level2_hits++;
        cands.push_back(HgCand{pos, w, rank_here});
      }
    }
    s.inner = s.nl_count ? s.nl_count - 1 : 0;
    if (bs1 < HG_TILE_BYTES && s.nl_count >= 2)  // ~ hg_tile_inner_kernel
      s.inner = static_cast<uint32_t>(hg_inner_pieces(data, base + s.first_nl + 1, base + s.last_nl + 1, bs1));
    sums[t] = s;
  }
  if (getenv("HGSIM_DUMP")) {
    std::vector<std::pair<uint64_t, uint32_t>> top;
    for (auto &kv : l1_hist) top.push_back({kv.second, kv.first});
    std::sort(top.rbegin(), top.rend());
    for (size_t i = 0; i < top.size() && i < 12; i++) {
      uint32_t x = top[i].second;
      fprintf(stderr, "L1 window '%c%c%c%c' x%llu\n", x & 0xff, (x >> 8) & 0xff, (x >> 16) & 0xff, x >> 24, (unsigned long long)top[i].first);
    }
  }
  // ---- tile scan
  HgTileBase st{0, 0};
  for (uint64_t t = 0; t < ntiles; t++) {
    bases[t] = st;
    st = hg_tile_apply(st, hg_tile_elem(sums[t], t * HG_TILE_BYTES), bs1);
  }
  bases[ntiles] = st;
  uint64_t pieces = st.L + (nbytes > st.cs ? hg_pieces(nbytes - st.cs, bs1) : 0);
  // ---- confirm
  std::vector<HgHit> hits;
  std::vector<HgHitAux> aux;
  uint64_t verified = 0;
  std::set<std::pair<uint64_t, uint32_t>> huge_done;  // (piece start, expression) pairs a huge automaton has run on
  for (auto &c : cands) {
    hg_verify_window(v, data, nbytes, c.pos, c.word, [&](uint32_t pattern, uint64_t fs, uint32_t) {
      verified++;
      auto emit = [&](uint64_t line_no, uint32_t to, uint64_t a, uint32_t len) {
        hits.push_back(HgHit{line_no, db->patterns[pattern].id, to});
        aux.push_back(HgHitAux{a, len, pattern});
      };
      // SINGLEMATCH automata of <= 2 state words are confirmed by window on the device (confirm modes 1 and 2): the same here
      const uint32_t mode = hg_confirm_mode(db->patterns[pattern]);
      if ((mode == 1 || mode == 2) && !getenv("HGSIM_NO_WINDOW")) {
        hg_confirm_window(v, data, nbytes, sums.data(), bases.data(), bs1, c.pos, fs, pattern, c.rank, emit);
      } else if (mode == 4) {
        // huge automata: ONE run per (expression, piece), whichever occurrence of the literal comes first — the device claims
        // the pair in a hash table (hg_huge.hip); a literal that overlaps itself (a{32767}) gives a candidate per byte
        bool fresh = false;
        hg_confirm(v, data, nbytes, sums.data(), bases.data(), bs1, c.pos, pattern, c.rank, emit,
                   [&](uint64_t ps) { return fresh = huge_done.insert({ps, pattern}).second; });
        (void)fresh;
      } else {
        hg_confirm(v, data, nbytes, sums.data(), bases.data(), bs1, c.pos, pattern, c.rank, emit);
      }
    });
  }
  // ---- always-on tier: every line start
  if (v.nslow) {
    for (uint64_t t = 0; t < ntiles; t++) {
      uint64_t base = t * HG_TILE_BYTES, end = std::min<uint64_t>(base + HG_TILE_BYTES, nbytes);
      uint32_t rank = 0;
      for (uint64_t s = base; s < end; s++) {
        bool starts = (s == 0) || data[s - 1] == '\n';
        if (starts)
          hg_scan_line_always_on(v, data, nbytes, sums.data(), bases.data(), bs1, s, rank, 0u, v.nslow,
                                 [&](uint32_t pi, uint64_t line_no, uint32_t to, uint64_t a, uint32_t len) {
                                   hits.push_back(HgHit{line_no, db->patterns[pi].id, to});
                                   aux.push_back(HgHitAux{a, len, pi});
                                 });
        if (data[s] == '\n') rank++;
      }
    }
  }
  uint64_t raw = hits.size();
  // ---- order + dedupe
  std::vector<uint32_t> order(hits.size());
  for (uint32_t i = 0; i < order.size(); i++) order[i] = i;
  std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
    uint64_t kx = hg_sort_key(hits[x], db->patterns[aux[x].pattern].single), ky = hg_sort_key(hits[y], db->patterns[aux[y].pattern].single);
    if (hits[x].line_no != hits[y].line_no) return hits[x].line_no < hits[y].line_no;
    return kx < ky;
  });
  std::vector<HgHit> sh(hits.size());
  std::vector<HgHitAux> sa(hits.size());
  for (size_t i = 0; i < order.size(); i++) { sh[i] = hits[order[i]]; sa[i] = aux[order[i]]; }
  std::vector<SimHit> kept;
  for (size_t i = 0; i < sh.size(); i++) {
    if (!hg_keep_hit(sh.data(), sa.data(), v.patterns, i)) continue;
    kept.push_back(SimHit{sh[i].line_no, sh[i].id, sh[i].to, sa[i].start, sa[i].len, sa[i].pattern});
  }
  SimHit *res = static_cast<SimHit *>(malloc(sizeof(SimHit) * (kept.size() ? kept.size() : 1)));
  std::memcpy(res, kept.data(), sizeof(SimHit) * kept.size());
  *out = res;
  if (stats) { stats[0] = bitmap_hits; stats[1] = verified; stats[2] = raw; stats[3] = pieces; stats[4] = level2_hits; }
  return static_cast<long>(kept.size());
}
void hgsim_free_hits(SimHit *p) { free(p); }
}
