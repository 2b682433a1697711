// Deterministic synthetic log generator (bench + parity tests), identical on host and device.
//
// The text is a sequence of independent blocks of HG_SYNTH_BLOCK bytes (deliberately NOT a multiple of
// the 16 KiB scan tile, so lines straddle tile boundaries); block b depends only on (seed, b), so any
// shard [first_block, ...) can be produced in place on any GPU without moving bytes over PCIe.
// Every block holds whole lines, '\n' terminated, printable ASCII:
//     2026-10-03T12:34:56.789Z host-123 svc-12 INFO key=value word word ...
// With probability hit_per_million / 1e6 a line carries one "needle" token from the caller's table
// (the bench derives needles from its pattern set), so hit counts are reproducible by construction.
// SURVEY.md §8(d) "Synthetic input".
#pragma once
#include <cstdint>

#include "hg_db.h"

constexpr uint32_t HG_SYNTH_BLOCK = 16000;
constexpr uint32_t HG_SYNTH_MAX_NEEDLE = 96;

struct HgSynthSpec {
  uint64_t seed;
  uint64_t first_block;
  uint32_t hit_per_million;
  uint32_t n_needles;
  const uint8_t *needles;
  const uint32_t *needle_off;
};

// Uniform pick in [0, n) from 32 random bits without any division (identical on host and device).
HG_HD uint32_t hg_pick(uint32_t r32, uint32_t n) { return static_cast<uint32_t>((static_cast<uint64_t>(r32) * n) >> 32); }

HG_HD uint64_t hg_splitmix(uint64_t &s) {
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

namespace hg_synth_detail {
// filler vocabulary: common log words; none of them is a needle
HG_HD const char *word(uint32_t i) {
  switch (i & 31u) {
    case 0: return "request";
    case 1: return "response";
    case 2: return "session";
    case 3: return "cache";
    case 4: return "lookup";
    case 5: return "queue";
    case 6: return "worker";
    case 7: return "retry";
    case 8: return "status=200";
    case 9: return "status=204";
    case 10: return "status=404";
    case 11: return "method=GET";
    case 12: return "method=POST";
    case 13: return "path=/api/v1/items";
    case 14: return "path=/healthz";
    case 15: return "user=guest";
    case 16: return "connection";
    case 17: return "accepted";
    case 18: return "closed";
    case 19: return "timeout";
    case 20: return "backend";
    case 21: return "upstream";
    case 22: return "latency_ms=";
    case 23: return "bytes=";
    case 24: return "trace_id=";
    case 25: return "shard=";
    case 26: return "region=us-east";
    case 27: return "region=eu-west";
    case 28: return "tenant=";
    case 29: return "attempt=";
    case 30: return "ok";
    default: return "done";
  }
}
HG_HD const char *level(uint32_t i) {
  switch (i & 3u) {
    case 0: return "INFO";
    case 1: return "DEBUG";
    case 2: return "WARN";
    default: return "INFO";
  }
}
struct Out {
  uint8_t *p;
  uint32_t n, cap;
  HG_HD void put(uint8_t c) {
    if (n < cap) p[n] = c;
    n++;
  }
  HG_HD void str(const char *s) {
    while (*s) put(static_cast<uint8_t>(*s++));
  }
  HG_HD void num(uint32_t v, int width) {  // zero padded decimal, width <= 5; constant divisions only
    const uint32_t d0 = v % 10u, d1 = (v / 10u) % 10u, d2 = (v / 100u) % 10u, d3 = (v / 1000u) % 10u, d4 = (v / 10000u) % 10u;
    if (width >= 5) put(static_cast<uint8_t>('0' + d4));
    if (width >= 4) put(static_cast<uint8_t>('0' + d3));
    if (width >= 3) put(static_cast<uint8_t>('0' + d2));
    if (width >= 2) put(static_cast<uint8_t>('0' + d1));
    put(static_cast<uint8_t>('0' + d0));
  }
};
}  // namespace hg_synth_detail

// Writes block `b` (absolute block index) into out[0, len), len <= HG_SYNTH_BLOCK (shorter only for the
// last, truncated block of a buffer).  Returns the number of lines that carry a needle.
HG_HD uint32_t hg_synth_block(const HgSynthSpec &sp, uint64_t b, uint8_t *out, uint32_t len) {
  using namespace hg_synth_detail;
  uint64_t s = sp.seed ^ (b * 0xD1B54A32D192ED03ull);
  Out o{out, 0, len};
  uint32_t needle_lines = 0;
  // a line never exceeds 24+9+7+6 + 12 tokens x 20 + needle 97 < 400 bytes
  while (o.n + 400 < HG_SYNTH_BLOCK) {
    uint64_t r = hg_splitmix(s);
    const uint32_t ra = static_cast<uint32_t>(r), rb = static_cast<uint32_t>(r >> 32);
    o.str("2026-10-");
    o.num(1 + hg_pick(ra, 28), 2);
    o.put('T');
    o.num(hg_pick(ra * 0x9E3779B1u, 24), 2);
    o.put(':');
    o.num(hg_pick(ra * 0x85EBCA6Bu, 60), 2);
    o.put(':');
    o.num(hg_pick(ra * 0xC2B2AE35u, 60), 2);
    o.put('.');
    o.num(hg_pick(rb, 1000), 3);
    o.str("Z host-");
    o.num(hg_pick(rb * 0x9E3779B1u, 1000), 3);
    o.str(" svc-");
    o.num(hg_pick(rb * 0x85EBCA6Bu, 100), 2);
    o.put(' ');
    uint64_t r2 = hg_splitmix(s);
    const uint32_t rc = static_cast<uint32_t>(r2), rd = static_cast<uint32_t>(r2 >> 32);
    o.str(level(rc >> 30));
    uint32_t ntok = 1 + hg_pick(rc * 0x9E3779B1u, 12);
    bool has_needle = sp.n_needles && (hg_pick(rd, 1000000u) < sp.hit_per_million);
    uint32_t needle_at = hg_pick(rd * 0x9E3779B1u, ntok);
    uint32_t needle_ix = sp.n_needles ? hg_pick(rd * 0x85EBCA6Bu, sp.n_needles) : 0;
    for (uint32_t t = 0; t < ntok; t++) {
      o.put(' ');
      if (has_needle && t == needle_at) {
        for (uint32_t k = sp.needle_off[needle_ix]; k < sp.needle_off[needle_ix + 1]; k++) o.put(sp.needles[k]);
        continue;
      }
      uint64_t r3 = hg_splitmix(s);
      const uint32_t re = static_cast<uint32_t>(r3), rf = static_cast<uint32_t>(r3 >> 32);
      const char *w = word(re >> 27);
      o.str(w);
      // words ending in '=' take a numeric value
      const char *e = w;
      while (*e) e++;
      if (e[-1] == '=') o.num(hg_pick(rf, 100000), 1 + static_cast<int>(hg_pick(re * 0x9E3779B1u, 5)));
    }
    o.put('\n');
    needle_lines += has_needle ? 1u : 0u;
  }
  // closing comment line pads the block to its exact size
  o.put('#');
  while (o.n + 1 < HG_SYNTH_BLOCK) o.put('.');
  o.put('\n');
  return needle_lines;
}
