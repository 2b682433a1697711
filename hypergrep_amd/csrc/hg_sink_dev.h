// Device-side hit emission shared by the confirm / always-on kernels (hg_kernels.hip, hg_always_on.hip) and the huge-automaton kernels
// (hg_huge.hip): a report goes straight into the finalize bucket of its line's start, or — compact-array mode — into the
// block's private staging segment that flush_hits() then copies to the compact output.
#pragma once
#include <hip/hip_runtime.h>

#include "hg_core.h"
#include "hg_engine.h"

// Hits are first appended to a block-private segment (LDS counter), then each block reserves one contiguous
// range of the compact output with a single global atomic and copies its segment there.
struct HitSink {
  HgHit *seg_hits;
  HgHitAux *seg_aux;
  uint32_t seg_cap;
  uint32_t *lds_count;
  // a.hit_direct: a block whose private segment is full appends straight to the compact array (one global atomic per hit).
  // The engine turns it on when equal segments for every block would have to grow past any sensible size because ONE block
  // holds most of the hits (all match ends of an all-matches expression on a very long line).  (The compact array's
  // fields are read from the kernel arguments at the call, not kept here: a larger struct went through scratch.)
  __device__ __forceinline__ void push(const HgConfirmArgs &a, uint64_t line_no, uint32_t id, uint32_t to, uint64_t start, uint32_t len, uint32_t pattern,
                                       bool single) const {
    if (start < a.own_lo || start >= a.own_hi) return;  // (segmented scans: the piece is another segment's to report)
    if (a.bucket_cap) {  // straight into the bucket of the line's start; the finalize kernels order each bucket
      const uint64_t rel_start = start - a.own_lo;
      const uint32_t b = static_cast<uint32_t>(rel_start >> a.bucket_shift);
      const uint32_t slot = atomicAdd(&a.bucket_fill[b], 1u);
      if (slot < a.bucket_cap) {
        const uint64_t at = static_cast<uint64_t>(b) * a.bucket_cap + slot;
        // raw record: the line's start inside its bucket rides in the top bits of the line number — it orders the bucket's
        // lines like the line number does, in far fewer key bits (hg_fin_*; the gather strips both extras again)
        const uint64_t rel = rel_start & ((1ull << a.bucket_shift) - 1ull);
        a.hits[at] = HgHit{line_no | (rel << HG_HIT_REL_SHIFT), id, to | (single ? HG_HIT_SINGLE_BIT : 0u)};
        a.aux[at] = HgHitAux{start, len, pattern};
      } else {
        atomicMax(&a.counters[HG_CNT_HIT_NEED], slot + 1u);  // the engine grows the buckets (or leaves bucketed emission) and repeats the pass
      }
      return;
    }
    const uint32_t slot = atomicAdd(lds_count, 1u);
    if (slot < seg_cap) {
      seg_hits[slot] = HgHit{line_no, id, to};
      seg_aux[slot] = HgHitAux{start, len, pattern};
    } else if (a.hit_direct) {
      const uint32_t at = atomicAdd(&a.counters[HG_CNT_HITS], 1u);
      if (at == 0xFFFFFFFFu) a.counters[HG_CNT_HITS_WRAPPED] = 1u;
      if (at < a.hit_cap) {
        a.hits[at] = HgHit{line_no, id, to};
        a.aux[at] = HgHitAux{start, len, pattern};
      }
    }
  }
};

__device__ __forceinline__ void flush_hits(const HgConfirmArgs &a, uint32_t *lds_count, uint32_t *lds_base) {
  if (a.bucket_cap) return;  // (block-uniform) nothing was staged
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t n = *lds_count;
    const uint32_t kept = n < a.hit_seg_cap ? n : a.hit_seg_cap;
    *lds_base = atomicAdd(&a.counters[HG_CNT_HITS], kept);
    if (n > a.hit_seg_cap && !a.hit_direct) atomicMax(&a.counters[HG_CNT_HIT_NEED], n);  // (hits past the segment were dropped: the pass is repeated)
    *lds_count = kept;
  }
  __syncthreads();
  const uint32_t n = *lds_count, base = *lds_base;
  const uint64_t seg0 = static_cast<uint64_t>(blockIdx.x) * a.hit_seg_cap;
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    if (base + i < a.hit_cap) {
      a.hits[base + i] = a.tmp_hits[seg0 + i];
      a.aux[base + i] = a.tmp_aux[seg0 + i];
    }
  }
}

