// Compiled pattern database layout shared by the host compiler and the gfx950 kernels.
//
// Replaces what hs_compile_multi builds for the reference (hypergrep/lib/c/hyperscanner.c:126-142):
// instead of Hyperscan's opaque bytecode the database is a set of flat arrays that the engine uploads
// to HBM once; the window filter (8-128 KiB, normally 16 KiB) is the only part staged in LDS by the streaming kernel.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define HG_HD __host__ __device__ __forceinline__
#else
#define HG_HD inline
#endif

// Flag values (hypergrep/utils.py:10-13).
constexpr uint32_t HG_FLAG_CASELESS = 1, HG_FLAG_DOTALL = 2, HG_FLAG_MULTILINE = 4, HG_FLAG_SINGLEMATCH = 8;
constexpr uint32_t HG_FLAGS_SUPPORTED = 15;

// Boundary contexts for zero-width assertions.  Every assertion the compiler accepts (^ $ \A \z \Z \b \B)
// is a boolean function of (context of the previous byte, context of the next byte); a 20-bit truth
// table indexed [prev * 5 + next] represents any conjunction / disjunction of them exactly.
enum : uint32_t { HG_PC_START = 0, HG_PC_NL = 1, HG_PC_WORD = 2, HG_PC_OTHER = 3 };                      // previous byte
enum : uint32_t { HG_NC_NL = 0, HG_NC_NLFINAL = 1, HG_NC_WORD = 2, HG_NC_OTHER = 3, HG_NC_END = 4 };     // next byte
constexpr uint32_t HG_TT_ALL = (1u << 20) - 1;

constexpr uint32_t HG_MAX_NODES = 1024;  // per pattern with DENSE tables (32 state words): the lane-private routines' limit
constexpr uint32_t HG_MAX_W = HG_MAX_NODES / 32;
// Larger automata ("huge": unrolled bounded repeats such as foo.{0,3000}bar, [a-z]{2000}x, a{32767}) keep the same node
// semantics with SPARSE tables (HgHugeHeader below) and run wave-cooperatively with the state words in LDS (hg_huge.hip).
// Limits: the expression's Thompson program size (what the oracle and Hyperscan's graph limits bound) and the node count.
constexpr uint32_t HG_HUGE_MAX_PROGRAM = 400000;   // Thompson instructions of the expression (oracle/orx.c ORX_MAX_INST restates the same bound)
constexpr uint32_t HG_HUGE_MAX_NODES = 1u << 19;   // (position, entry condition) nodes: 16384 state words = 64 KiB of LDS per state copy
constexpr uint32_t HG_HUGE_STAGE_MAX = 12288;      // words (48 KiB) of per-byte tables a huge routine stages in LDS
constexpr uint32_t HG_HUGE_MAX_EDGES = 1u << 22;   // automaton edges the compiler will hold (quadratic constructions such as (a?b?c?...){n} stop here)
constexpr uint32_t HG_MAX_PATTERNS = 1u << 24;  // pattern index and window offset share one word in the verified-occurrence records
constexpr uint32_t HG_ALWAYS_ON_FAST_MAX_LEN = 64;  // always-on patterns up to this match length use a fixed lead-in, longer / unbounded ones their line's start
constexpr uint32_t HG_FACTOR_MAX = 32;   // bytes of a required literal kept for the in-stream verify
// A window = the low HG_WINDOW_BYTES bytes of a dword-aligned text dword.  4 and 3 are supported; measured on the
// round-1 workload a 3-byte window admits 6-byte literals to the fast tier but is hit 1.5x more often by filler text.
constexpr uint32_t HG_WINDOW_BYTES = 4;
constexpr uint32_t HG_WINDOW_MASK = HG_WINDOW_BYTES == 4 ? 0xFFFFFFFFu : 0x00FFFFFFu;
constexpr uint32_t HG_FAST_MIN_FACTOR = HG_WINDOW_BYTES + 3;  // a window on every residue mod 4
constexpr uint32_t HG_DENSE_MIN_FACTOR = HG_WINDOW_BYTES - 1;  // byte-aligned probing: one window anywhere; one byte short: the byte after the literal enumerated
constexpr uint32_t HG_HASH_BITS = 18;       // v_dot4_u32_u8 of four bytes with weights < 256 fits 18 bits
// Byte-weighted sums of the (case-folded) window dword, one v_dot4_u32_u8 each:
//   hash C  -> bucket index of the window table in HBM, and (low 16 bits) the fingerprint
//   hash A  (weights are multiples of 4, so the sum is already the byte offset of a 4-byte slot) -> the window's slot
//           in the LDS filter: ONE probe per text dword (two LDS reads per dword measured 5 % slower on the stream pass).
// A slot word = care mask << 16 | fingerprint: windows that share a slot keep the fingerprint bits they agree on
// (hg_slot_match), so a lookup can never miss a window; shared slots only admit more false positives, which the
// second level and the verify pass remove.  (wide mode, large sets: hash A and B name two slots of two 16-bit cells)
// (with 3-byte windows the weight of the dword's top byte is zero in all three sums: that byte is not part of the window)
constexpr uint32_t HG_TOP_WEIGHT_MASK = HG_WINDOW_BYTES == 4 ? 0xFFFFFFFFu : 0x00FFFFFFu;
constexpr uint32_t HG_HASH_WEIGHTS = 0xfbf1efe9u & HG_TOP_WEIGHT_MASK;  // C: 233, 239, 241, 251
// Candidate weight vectors (A and B of each pair: the single-probe filter takes whichever spreads the windows best, wide
// mode uses a pair for its two slots; bytes are multiples of 4
// and deliberately not in arithmetic progression, so that no small byte difference cancels in both sums).
constexpr uint32_t HG_SLOT_WEIGHT_CHOICES[][2] = {
    {0x2cec94fcu & HG_TOP_WEIGHT_MASK, 0xbc34f474u & HG_TOP_WEIGHT_MASK},  // A: 252,148,236,44   B: 116,244,52,188
    {0x74d43cb4u & HG_TOP_WEIGHT_MASK, 0xe40c9c5cu & HG_TOP_WEIGHT_MASK},  // A: 180,60,212,116   B: 92,156,12,228
    {0xa41cf86cu & HG_TOP_WEIGHT_MASK, 0x54c4247cu & HG_TOP_WEIGHT_MASK},  // A: 108,248,28,164   B: 124,36,196,84
    {0xdc4484f4u & HG_TOP_WEIGHT_MASK, 0x1cac6cccu & HG_TOP_WEIGHT_MASK},  // A: 244,132,68,220   B: 204,108,172,28
};
constexpr uint32_t HG_SLOT_WEIGHT_NCHOICES = 4;
constexpr uint32_t HG_FILTER_MIN_LOG2 = 11, HG_FILTER_MAX_LOG2 = 15;  // 4-byte slots: 8 KiB .. 128 KiB of LDS
constexpr uint32_t HG_FILTER_EMPTY = 0xFFFFFFFFu;

// One compiled expression: a position (Glushkov) automaton whose nodes are (position, entry condition).
// Tables live in one u32 pool; *_off are indices into it.
// Tables of a huge automaton (nw > HG_MAX_W).  HgPattern::follow_off points at this header in the pool; reach_off / init_off
// / amask_off / acc_off keep their meaning with these shapes: reach[ncls][nw] (indexed by the byte's CLASS, cls[] below),
// init[nw], and — unless ctxfree — amask[4][4][nw], acc[4][5][nw]; a ctxfree automaton (no boundary conditions) has no
// amask and ONE acc[nw].  The follow relation is sparse: node v -> v + 1 edges are one bit of smask ("shift" edges: with
// nodes numbered in expression order an unrolled repeat is almost all of them), every other edge is an "exception": the
// sources are the bits of xsrc, source number k (xrank[w] + bits of xsrc[w] below it) owns the target ranges
// xt[2 r], xt[2 r + 1] (first and last node, inclusive) for r in [xlist[k], xlist[k + 1]).
struct HgHugeHeader {
  uint32_t cls_off;    // pool: 64 words = 256 bytes, the class of each byte value
  uint32_t ncls;
  uint32_t smask_off;  // pool: smask[nw]
  uint32_t xsrc_off;   // pool: xsrc[nw]
  uint32_t xrank_off;  // pool: xrank[nw]
  uint32_t xlist_off;  // pool: xlist[sources + 1]
  uint32_t xt_off;     // pool: xt[2 * ranges]
  uint32_t ctxfree;    // 1: no boundary conditions anywhere
  uint32_t init_hi;    // one past the last non-zero word of init[]
  uint32_t nsources, nranges;
  uint32_t reserved[5];
};
static_assert(sizeof(HgHugeHeader) == 64, "HgHugeHeader layout");

struct HgPattern {
  // the four words every confirm routine reads, in one 16-byte piece (the record is 80 bytes: no such piece straddles a cache line)
  uint32_t id;          // report id given by the caller
  uint32_t single;      // HS_FLAG_SINGLEMATCH set
  uint32_t max_len;     // longest possible match in bytes, 0 = unbounded (literal_only: the literal's length)
  uint32_t lit_lead;    // tier 0: every match contains one of the pattern's required literals starting at most this many bytes
                        // after the match's start (0xFFFFFFFF: no bound) — the confirm routines' window (hg_confirm_dev.h)
  uint32_t flags;       // HS_FLAG_* bits
  uint32_t nnodes;      // automaton nodes
  uint32_t nw;          // state words = ceil(nnodes / 32); more than HG_MAX_W: a huge automaton (HgHugeHeader)
  uint32_t reach_off;   // reach[256][nw]   nodes whose byte class contains c
  uint32_t follow_off;  // follow[nnodes][nw]  (huge: the HgHugeHeader)
  uint32_t init_off;    // init[nw]         nodes enterable from the (always active) start state
  uint32_t amask_off;   // amask[4][4][nw]  nodes whose entry condition holds for (prev ctx, ctx of own byte)
  uint32_t acc_off;     // acc[4][5][nw]    nodes that accept for (ctx of own byte, next ctx)
  uint32_t tier;        // 0: anchored by a required literal (stream prefilter + confirm); 1: always-on
  uint32_t simple;      // one state word and no boundary conditions: S' = (init | follow(S)) & reach[c], accept = S & acc_all
  uint32_t acc_all;     // accepting nodes when `simple`
  uint32_t init_word;   // init[0] when `simple`
  uint32_t literal_only;  // the whole expression is one literal (its factor): a verified occurrence IS the match
  uint32_t reserved[3];
};
static_assert(sizeof(HgPattern) == 80, "HgPattern layout");

// A required literal of one pattern ("factor"): any match of the pattern contains an occurrence.  ONE 64-byte cache line
// (round 2: 80 bytes with a mask byte per literal byte — three to five 16-byte fetches from two lines per verified candidate).
struct HgFactor {
  uint32_t pattern;              // index into patterns[]
  uint32_t len;                  // <= HG_FACTOR_MAX
  uint32_t mode;                 // confirm routine of the pattern (hg_confirm_mode), copied here so the verify pass needs no pattern load
  uint32_t mode_rank;            // the pattern's rank among the patterns of its confirm mode: names its verified-occurrence lists (no two
                                 // patterns of a mode share a list while the mode has at most HG_DEFER_SHARDS of them)
  uint8_t lit[HG_FACTOR_MAX];    // literal bytes (case-insensitive letters in lower case)
  uint32_t casebits;             // bit b: byte b is a case-insensitive letter (compared under the mask 0xDF)
  uint32_t id;                   // the pattern's report id (a literal-only expression needs nothing else of its pattern)
  uint32_t reserved[2];
};
static_assert(sizeof(HgFactor) == 64, "HgFactor layout");
HG_HD uint32_t hg_factor_cmask(const HgFactor &f, uint32_t b) { return ((f.casebits >> b) & 1u) ? 0xDFu : 0xFFu; }
// Compare mask of the literal's dword j (bytes 4 j .. 4 j + 3): 0xDF on case-insensitive letters, zero past the literal's end.
HG_HD uint32_t hg_factor_mask_dword(uint32_t casebits, uint32_t len, uint32_t j) {
  const uint32_t nib = (casebits >> (4u * j)) & 15u;
  const uint32_t spread = (nib & 1u) | ((nib & 2u) << 7) | ((nib & 4u) << 14) | ((nib & 8u) << 21);
  const uint32_t m = ~(spread << 5);
  const uint32_t valid = len > 4u * j ? len - 4u * j : 0u;
  return valid >= 4u ? m : (m & ((1u << (8u * valid)) - 1u));
}

// Direct window table: the folded value of a literal window -> the (literal, offset) it belongs to.  A bucket = four values
// (one 16-byte fetch) + their four payloads, 32 bytes; open addressing over buckets, filled front to back, never more than
// half full over all.  Payload: factor_off of the only (literal, offset) with this window; HG_WTAB_SHARED: several literals
// share the window, the discriminated buckets (disc / bucket_off2 / windows2 below) name them; HG_WTAB_EMPTY: no entry.  A
// window value that is not in the table belongs to no literal: the stream filter's false positives end here.
struct HgWinBucket {
  uint32_t value[4];
  uint32_t factor_off[4];
};
static_assert(sizeof(HgWinBucket) == 32, "HgWinBucket layout");
constexpr uint32_t HG_WTAB_WAYS = 4, HG_WTAB_EMPTY = 0xFFFFFFFFu, HG_WTAB_SHARED = 0xFFFFFFFEu;
HG_HD uint32_t hg_wtab_bucket(uint32_t folded, uint32_t bucket_mask) { return ((folded * 0x9E3779B1u) >> 9) & bucket_mask; }
// Returns HG_WTAB_EMPTY (not in the table), HG_WTAB_SHARED, or the factor_off of the window's only owner.
HG_HD uint32_t hg_wtab_find(const HgWinBucket *tab, uint32_t bucket_mask, uint32_t folded) {
  for (uint32_t b = hg_wtab_bucket(folded, bucket_mask);; b = (b + 1u) & bucket_mask) {
    const HgWinBucket &e = tab[b];
    for (uint32_t k = 0; k < HG_WTAB_WAYS; k++) {
      if (e.factor_off[k] == HG_WTAB_EMPTY) return HG_WTAB_EMPTY;
      if (e.value[k] == folded) return e.factor_off[k];
    }
  }
}

// One window (HG_WINDOW_BYTES bytes) of a factor, placed at literal offset `off`; keyed by the folded, masked dword value.
struct HgWindow {
  uint32_t value;
  uint32_t factor_off;  // factor index << 8 | off
};

// Several always-on expressions in ONE state word.  Single-word automata whose nodes add up to <= 32 are renumbered into
// one word (their follow sets stay disjoint), with the union of their reach / init tables: one pass of the always-on kernel
// then advances all of them for the price of one, and only a match looks at which member accepted.  A group of context-free
// members runs the cheapest routine; as soon as one member has boundary conditions (\b, ^, $ ...) the group carries the
// union of the members' per-context tables (ctx_off) and runs the routine with conditions — members without conditions
// ride along for free.
constexpr uint32_t HG_GROUP_MAX_MEMBERS = 8;
struct HgSlowGroup {
  uint32_t reach_off;   // pool: reach[256] over the group's nodes
  uint32_t follow_off;  // pool: follow[nnodes]
  uint32_t nnodes, init_word, acc_all;
  uint32_t max_len;      // longest match of any member, 0 = unbounded (some member is): the lead-in of the pass
  uint32_t nmembers;
  uint32_t single_mask;  // bit m: member m has HS_FLAG_SINGLEMATCH
  uint32_t member[HG_GROUP_MAX_MEMBERS];  // pattern indices
  uint32_t acc[HG_GROUP_MAX_MEMBERS];     // context-free group: accepting nodes of each member; with conditions: ALL nodes of each member
  uint32_t ctx_off;      // 0: context-free group; else pool: amask[4 previous][4 next] then acc[4 previous][5 next] over the group's nodes
  uint32_t reserved;
};
static_assert(sizeof(HgSlowGroup) == 104, "HgSlowGroup layout");

HG_HD uint32_t hg_dot4(uint32_t v, uint32_t w) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_udot4(v, w, 0u, false);
#else
  return (v & 0xff) * (w & 0xff) + ((v >> 8) & 0xff) * ((w >> 8) & 0xff) + ((v >> 16) & 0xff) * ((w >> 16) & 0xff) + (v >> 24) * (w >> 24);
#endif
}
HG_HD uint32_t hg_hash_window(uint32_t folded) { return hg_dot4(folded, HG_HASH_WEIGHTS); }
// Byte offsets of the two candidate slots; byte_mask = (slots - 1) << 2.
HG_HD uint32_t hg_slot(uint32_t folded, uint32_t weights, uint32_t byte_mask) { return hg_dot4(folded, weights) & byte_mask; }
// Single-probe slot test: every fingerprint bit the slot cares about agrees with hash C (the mask has 16 bits, so the
// upper bits of hash C drop out by themselves).  The empty slot 0xFFFFFFFF only admits fingerprint 0xFFFF.
HG_HD bool hg_slot_match(uint32_t slot_word, uint32_t hash_c) { return (hash_c & (slot_word >> 16)) == (slot_word & 0xFFFFu); }
// Wide-mode slots (large pattern sets): one byte-weighted sum spans too few values for text over a small alphabet
// (hex digits: ~9000 distinct sums), so each slot mixes the low bits of both sums, sum_x + (sum_y << 8).
HG_HD uint32_t hg_slot_wide(uint32_t sum_x, uint32_t sum_y, uint32_t byte_mask) { return (sum_x + (sum_y << 8)) & byte_mask; }

// Second-level check of a filter slot: what the dwords just before / after the window must look like
// (folded, byte-masked) for any of the slot's windows to be part of its literal.  Conservative union.
struct HgFilterExt {
  uint32_t pv, pm;  // the 4 bytes before the window: (prev | fold) & pm == pv
  uint32_t nv, nm;  // the 4 bytes after the window (with 3-byte windows they start with the top byte of the window's own dword)
};
HG_HD bool hg_ext_pass(const HgFilterExt &e, uint32_t prev_folded, uint32_t next_folded) {
  return (((prev_folded ^ e.pv) & e.pm) | ((next_folded ^ e.nv) & e.nm)) == 0;
}

// Verify-pass discriminator.  Windows are grouped by hash C; a group can be large when many literals share a window
// ("status=5xx" alternatives, common stems).  Per group the compiler picks the dword of the literals, `delta` bytes from
// the window, whose known bytes split the group best; the verify pass reads that dword of the text and goes straight to
// the (usually one) literal that agrees with it.  disc[h] = (delta as int8) | byte-select bits << 8.
constexpr uint32_t HG_DISC_WEIGHTS_1 = 0xc5a36b1du, HG_DISC_WEIGHTS_2 = 0x3b9d47e1u;
HG_HD uint32_t hg_disc_bytes(uint32_t sel4) {
  return ((sel4 & 1u) ? 0xFFu : 0u) | ((sel4 & 2u) ? 0xFF00u : 0u) | ((sel4 & 4u) ? 0xFF0000u : 0u) | ((sel4 & 8u) ? 0xFF000000u : 0u);
}
HG_HD uint32_t hg_disc_bucket(uint32_t h, uint32_t masked_dword) {
  return (h ^ (hg_dot4(masked_dword, HG_DISC_WEIGHTS_1) + (hg_dot4(masked_dword, HG_DISC_WEIGHTS_2) << 7))) & ((1u << HG_HASH_BITS) - 1u);
}

// Second level of a single-probe slot (64 B, HBM / L2): the window values that live in the slot, exactly, each with its
// own neighbour conditions; a third and further value of a crowded slot share `rest` (byte-wise agreement).
struct HgSlotInfo {
  uint32_t value[2];    // folded window values
  uint32_t nvalues;     // 0..2 of them valid
  uint32_t many;        // more than two values map to the slot: any other dword is judged by `rest`
  HgFilterExt cond[2];
  HgFilterExt rest;
};
static_assert(sizeof(HgSlotInfo) == 64, "HgSlotInfo layout");
// pm_keep / nm_keep: 0 where the caller has no left / right neighbour dword (condition skipped), else all ones.
HG_HD bool hg_slot_pass(const HgSlotInfo &s, uint32_t folded, uint32_t prev_folded, uint32_t next_folded, uint32_t pm_keep, uint32_t nm_keep) {
  HgFilterExt e;
  if (s.nvalues > 0 && s.value[0] == folded) e = s.cond[0];
  else if (s.nvalues > 1 && s.value[1] == folded) e = s.cond[1];
  else if (s.many) e = s.rest;
  else return false;
  e.pm &= pm_keep;
  e.nm &= nm_keep;
  return hg_ext_pass(e, prev_folded, next_folded);
}

HG_HD bool hg_is_word(uint32_t b) {
  return (b - '0' < 10u) || ((b | 0x20) - 'a' < 26u) || b == '_';
}
