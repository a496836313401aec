// Pippenger multi-scalar multiplication kernels for gfx950 (G1 and G2 via the field tag T).
//
// Replaces the reference's prover loops `acc = Add(acc, MulScalar(base[i], scalar[i]))`
// (groth16/groth16.go:243-250,269-271; snark.go:265-286; bn128/g1.go:140-155, g2.go:142-181):
// per term ~254 doublings + ~127 additions there, W = ceil(254/c) mixed additions here.
//
// Pipeline (all on the library stream, no host round trips):
//   plan (once per scalar vector, shared by every base array multiplied by it; no host round trip):
//     k_digits        scalars -> signed c-bit digit matrix (u16)
//     k_hist          per (window, slice, bucket range) workgroup: histogram of 2^15 buckets in LDS (128 KiB of the 160 KiB)
//     k_colscan+scan  exclusive prefix sums -> bucket offsets, per-slice cursors
//     k_scatter       counting sort with LDS cursors: entries[] = term index (sign in bit 31) grouped by bucket
//     k_chunk_map     cut the sorted entry list into equal chunks of 32 entries (load balance)
//   per base array (each has a resident window table rows[j][i] = 2^(c j) P_i):
//     k_bucket_accumulate  one thread per CHUNK: XYZZ += +-affine table point (8M+2S), flushing at bucket
//                          boundaries; cut buckets leave head/tail partials                      <- dominant
//     k_heavy_combine      block-wide tree-sum for buckets cut into many chunks
//     k_bucket_combine     merged[b] = sum of the chunk partials of bucket b
//     k_block_reduce       sum_b (b+1) merged[b] per 2048-bucket workgroup (running sums + LDS scan)
//   host: adds the <= 16 workgroup pairs, converts to affine / from Montgomery
//
// HBM layout: bases are AoS packed canonical Montgomery words (G1: 16 x u32 = 64 B/point,
// G2: 128 B), so a bucket thread gathers each point with 4 (8) 16-byte loads; scalars are the
// ABI's 8 x u32 words, read fully coalesced; buckets are raw 29-bit limbs (36 / 72 words).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ec.h"
#include "point_io.h"

namespace gs {

struct PlanParams {
  uint32_t n;        // scalars
  int c;             // window bits (<= 20)
  int W;             // windows = floor(254 / c) + 1
  uint32_t B;        // buckets per window = 2^(c-1)   (digits are signed: [-B+1, B])
  uint32_t S;        // slices of the scalar vector (one histogram/scatter workgroup per (window, slice))
  uint32_t slice;    // scalars per slice
  uint32_t stride;   // row stride of the digit matrix (elements)
  uint32_t R;        // bucket ranges of kRangeBuckets counters each (B <= 2^15: one range; c = 20: sixteen)
  uint32_t tf;       // table-free route: a bucket set PER WINDOW (bucket id = w * B + |d| - 1; msm.h, MsmPlan) instead of one for all
};
// One histogram / scatter workgroup keeps the counters of ONE bucket range in LDS (2^15 u32 = 128 KiB of the CU's 160 KiB) and
// counts the digits of its (window, slice) that fall into it; wide windows (c > 16) take R = B / 2^15 such workgroups per
// (window, slice), each re-reading the slice's digits (sequential 4-byte reads: the sort is bandwidth-cheap, the accumulation is
// not -- fewer, wider windows trade R passes over a 50 MiB digit matrix for 3 fewer point additions per term).
constexpr uint32_t kRangeLog = 15;
constexpr uint32_t kRangeBuckets = 1u << kRangeLog;

// signed digit of window w with the running carry (digit in [-B+1, B]; 0 <-> the term is skipped in that window)
GS_HD int32_t next_digit(const uint32_t (&k)[8], const PlanParams& pp, int w, uint32_t& carry) {
  uint32_t raw = scalar_bits(k, w * pp.c, pp.c) + carry;
  if (raw > pp.B) { carry = 1; return (int32_t)raw - (int32_t)(2u * pp.B); }
  carry = 0;
  return (int32_t)raw;
}

// ---- plan, step 1: scalars -> digit matrix digits[w][i] = d + B - 1, read once, coalesced ------------------------
using digit_t = uint32_t;
// term_index (optional): the plan covers only the listed terms -- compact term i of the plan is term term_index[i] - index_bias of the
// scalar vector (and of the base arrays: k_scatter writes that id into the entries).  For keys with sparse B arrays (prove.h,
// GrothPkObj::b_index): the terms whose base points are the point at infinity never enter the digit matrix.
__global__ void __launch_bounds__(256) k_digits(const uint32_t* __restrict__ scalars, PlanParams pp, digit_t* __restrict__ digits,
                                                 const uint32_t* __restrict__ term_index, uint32_t index_bias) {
  wave_priority<GS_PRIO_PLAN>();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= pp.n) return;
  const uint32_t src = term_index ? term_index[i] - index_bias : i;
  uint32_t k[8];
  const uint4* s4 = reinterpret_cast<const uint4*>(scalars + (size_t)src * 8);
  uint4 lo = s4[0], hi = s4[1];
  k[0] = lo.x; k[1] = lo.y; k[2] = lo.z; k[3] = lo.w; k[4] = hi.x; k[5] = hi.y; k[6] = hi.z; k[7] = hi.w;
  scalar_canon(k);
  uint32_t carry = 0;
  for (int w = 0; w < pp.W; ++w) {
    const int32_t d = next_digit(k, pp, w, carry);
    digits[(size_t)w * pp.stride + i] = (digit_t)(d + (int32_t)pp.B - 1);
  }
}

// mask[i >> 5] bit (i & 31) = point i of `a` OR of `b` (either may be null) is not the point at infinity (packed affine: all words
// zero); *count += finite points.  The mask must be zero beforehand.
__global__ void __launch_bounds__(256) k_finite_mask(const uint32_t* __restrict__ a, uint32_t wa, const uint32_t* __restrict__ b, uint32_t wb, uint32_t n,
                                                      uint32_t* __restrict__ mask, uint32_t* __restrict__ count) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  bool finite = false;
  if (i < n) {
    uint32_t acc = 0;
    if (a) for (uint32_t k = 0; k < wa; ++k) acc |= a[(size_t)i * wa + k];
    if (b) for (uint32_t k = 0; k < wb; ++k) acc |= b[(size_t)i * wb + k];
    finite = acc != 0;
  }
  if (finite) { atomicOr(&mask[i >> 5], 1u << (i & 31u)); atomicAdd(count, 1u); }
}

// ---- plan, step 2: per-(window, slice, range) bucket histogram in LDS -----------------------------------------
// grid = (W, S, R): blockIdx.x = window, so that all slices of a window run on XCD (w % 8) and the window's
// 4n-byte region of `entries` is assembled in ONE L2 by the scatter pass below.
constexpr int kSortBlock = 1024;      // upper bound; the launch picks 256 .. 1024 threads (msm.hip: sort_block)

__global__ void __launch_bounds__(kSortBlock) k_hist(const digit_t* __restrict__ digits, PlanParams pp, uint32_t* __restrict__ hist) {
  wave_priority<GS_PRIO_PLAN>();
  extern __shared__ uint32_t sh[];
  const uint32_t w = blockIdx.x, s = blockIdx.y, r = blockIdx.z;
  const uint32_t nb = min(pp.B, kRangeBuckets), base = r << kRangeLog;
  for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) sh[b] = 0;
  __syncthreads();
  const uint32_t lo = s * pp.slice, hi = min(pp.n, lo + pp.slice);
  const digit_t* row = digits + (size_t)w * pp.stride;
  const int32_t zero = (int32_t)pp.B - 1;
  for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    const int32_t d = (int32_t)row[i] - zero;
    if (d != 0) {
      const uint32_t b = (uint32_t)(d < 0 ? -d : d) - 1u;
      if ((b >> kRangeLog) == r) atomicAdd(&sh[b - base], 1u);
    }
  }
  __syncthreads();
  uint32_t* out = hist + ((size_t)w * pp.S + s) * pp.B + base;
  for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) out[b] = sh[b];
}

// hist[q][b], q = w*S + s  ->  exclusive prefix over q (in place); totals[b] = sum over q.
// (All windows share one bucket set -- see the window tables below -- so a bucket's entries are the
// concatenation of its (window, slice) runs.)
__global__ void __launch_bounds__(256) k_colscan(uint32_t* __restrict__ hist, PlanParams pp, uint32_t* __restrict__ totals) {
  wave_priority<GS_PRIO_PLAN>();
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= pp.B) return;
  uint32_t* col = hist + b;
  const uint32_t nq = (uint32_t)pp.W * pp.S;
  uint32_t run = 0;
  for (uint32_t q = 0; q < nq; ++q) { const uint32_t t = col[(size_t)q * pp.B]; col[(size_t)q * pp.B] = run; run += t; }
  totals[b] = run;
}

// The table-free route keeps the windows apart: hist[q][b], q = w*S + s  ->  exclusive prefix over the SLICES of window w only;
// totals[w * B + b] = the entries of bucket (w, b).  (B <= 2^15 on that route: one range.)
__global__ void __launch_bounds__(256) k_colscan_windows(uint32_t* __restrict__ hist, PlanParams pp, uint32_t* __restrict__ totals) {
  wave_priority<GS_PRIO_PLAN>();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (uint32_t)pp.W * pp.B) return;
  const uint32_t w = i / pp.B, b = i % pp.B;
  uint32_t* col = hist + (size_t)w * pp.S * pp.B + b;
  uint32_t run = 0;
  for (uint32_t q = 0; q < pp.S; ++q) { const uint32_t t = col[(size_t)q * pp.B]; col[(size_t)q * pp.B] = run; run += t; }
  totals[i] = run;
}

// ---- plan, step 4: counting-sort scatter; the cursors of the workgroup's bucket range live in LDS -------------------
// entry = sign (bit 31) | window (bits 30..26) | term index (bits 25..0)
__global__ void __launch_bounds__(kSortBlock) k_scatter(const digit_t* __restrict__ digits, PlanParams pp, const uint32_t* __restrict__ hist,
                                                         const uint32_t* __restrict__ offsets, uint32_t* __restrict__ entries,
                                                         const uint32_t* __restrict__ term_index, uint32_t index_bias) {
  wave_priority<GS_PRIO_PLAN>();
  extern __shared__ uint32_t sh[];
  const uint32_t w = blockIdx.x, s = blockIdx.y, r = blockIdx.z;
  const uint32_t nb = min(pp.B, kRangeBuckets), base = r << kRangeLog;
  const uint32_t* pre = hist + ((size_t)w * pp.S + s) * pp.B + base;
  const uint32_t* off = offsets + (pp.tf ? (size_t)w * pp.B : (size_t)0);      // table-free: window w's own bucket set
  for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) sh[b] = off[base + b] + pre[b];
  __syncthreads();
  const uint32_t lo = s * pp.slice, hi = min(pp.n, lo + pp.slice);
  const digit_t* row = digits + (size_t)w * pp.stride;
  const int32_t zero = (int32_t)pp.B - 1;
  for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    const int32_t d = (int32_t)row[i] - zero;
    if (d != 0) {
      const uint32_t b = (uint32_t)(d < 0 ? -d : d) - 1u;
      if ((b >> kRangeLog) == r) {
        const uint32_t pos = atomicAdd(&sh[b - base], 1u);
        const uint32_t id = term_index ? term_index[i] - index_bias : i;       // (k_digits: the plan's compact term i)
        entries[pos] = id | (w << kWindowShift) | (d < 0 ? kSignBit : 0u);
      }
    }
  }
}

// ---- wide windows (R > 1 bucket ranges): partition first, then the same LDS counting sort per partition --------------
// Filtering a (window, slice) of the digit matrix once per range would read it R times (16 x 50 MiB at c = 20).  Instead
// the (term, digit) pairs are first PARTITIONED by (window, range) -- W * R <= 512 partitions of ~n / R records each, sizes
// counted by k_part_count, bases by k_part_scan, records written by k_part_scatter -- and the LDS histogram / scatter
// workgroup of (window, range, slice) then reads only its own partition: every pair is read twice, whatever R is.
// record = bucket (high word) | sign (bit 31) term index (low word); the window is implied by the partition.
constexpr int kPartBlock = 256;
constexpr int kPartPerThread = 4;        // scalars per thread of k_part_scatter: runs of ~kPartBlock * 4 / R records per partition
constexpr uint32_t kMaxParts = 512;

GS_HD void load_scalar_canon(const uint32_t* __restrict__ scalars, uint32_t i, uint32_t (&k)[8]) {
  const uint4* s4 = reinterpret_cast<const uint4*>(scalars + (size_t)i * 8);
  const uint4 lo = s4[0], hi = s4[1];
  k[0] = lo.x; k[1] = lo.y; k[2] = lo.z; k[3] = lo.w; k[4] = hi.x; k[5] = hi.y; k[6] = hi.z; k[7] = hi.w;
  scalar_canon(k);
}

__global__ void __launch_bounds__(kPartBlock) k_part_count(const uint32_t* __restrict__ scalars, PlanParams pp, uint32_t* __restrict__ part_count) {
  wave_priority<GS_PRIO_PLAN>();
  __shared__ uint32_t cnt[kMaxParts];
  const uint32_t nparts = (uint32_t)pp.W * pp.R;
  for (uint32_t t = threadIdx.x; t < nparts; t += kPartBlock) cnt[t] = 0;
  __syncthreads();
  const uint32_t i = blockIdx.x * kPartBlock + threadIdx.x;
  if (i < pp.n) {
    uint32_t k[8];
    load_scalar_canon(scalars, i, k);
    uint32_t carry = 0;
    for (int w = 0; w < pp.W; ++w) {
      const int32_t d = next_digit(k, pp, w, carry);
      if (d != 0) atomicAdd(&cnt[(uint32_t)w * pp.R + (((uint32_t)(d < 0 ? -d : d) - 1u) >> kRangeLog)], 1u);
    }
  }
  __syncthreads();
  for (uint32_t t = threadIdx.x; t < nparts; t += kPartBlock) if (cnt[t]) atomicAdd(&part_count[t], cnt[t]);
}

// part_base[0 .. nparts] = exclusive prefix of the partition sizes; cursors start at zero
__global__ void __launch_bounds__(64) k_part_scan(uint32_t* __restrict__ part_count, uint32_t nparts, uint32_t* __restrict__ part_base,
                                                    uint32_t* __restrict__ part_cursor) {
  wave_priority<GS_PRIO_PLAN>();
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  uint32_t run = 0;
  for (uint32_t t = 0; t < nparts; ++t) { part_base[t] = run; run += part_count[t]; part_cursor[t] = 0; part_count[t] = 0; }
  part_base[nparts] = run;
}

__global__ void __launch_bounds__(kPartBlock) k_part_scatter(const uint32_t* __restrict__ scalars, PlanParams pp, const uint32_t* __restrict__ part_base,
                                                              uint32_t* __restrict__ part_cursor, uint2* __restrict__ recs) {
  wave_priority<GS_PRIO_PLAN>();
  __shared__ uint32_t cnt[kMaxParts];      // this workgroup's records per partition, then its write cursor in each
  const uint32_t nparts = (uint32_t)pp.W * pp.R;
  for (uint32_t t = threadIdx.x; t < nparts; t += kPartBlock) cnt[t] = 0;
  __syncthreads();
  const uint32_t first = blockIdx.x * (kPartBlock * kPartPerThread);
  for (int j = 0; j < kPartPerThread; ++j) {
    const uint32_t i = first + j * kPartBlock + threadIdx.x;
    if (i >= pp.n) break;
    uint32_t k[8];
    load_scalar_canon(scalars, i, k);
    uint32_t carry = 0;
    for (int w = 0; w < pp.W; ++w) {
      const int32_t d = next_digit(k, pp, w, carry);
      if (d != 0) atomicAdd(&cnt[(uint32_t)w * pp.R + (((uint32_t)(d < 0 ? -d : d) - 1u) >> kRangeLog)], 1u);
    }
  }
  __syncthreads();
  for (uint32_t t = threadIdx.x; t < nparts; t += kPartBlock) {            // reserve this workgroup's run in every partition
    const uint32_t mine = cnt[t];
    cnt[t] = part_base[t] + (mine ? atomicAdd(&part_cursor[t], mine) : 0u);
  }
  __syncthreads();
  for (int j = 0; j < kPartPerThread; ++j) {
    const uint32_t i = first + j * kPartBlock + threadIdx.x;
    if (i >= pp.n) break;
    uint32_t k[8];
    load_scalar_canon(scalars, i, k);
    uint32_t carry = 0;
    for (int w = 0; w < pp.W; ++w) {
      const int32_t d = next_digit(k, pp, w, carry);
      if (d != 0) {
        const uint32_t b = (uint32_t)(d < 0 ? -d : d) - 1u;
        const uint32_t pos = atomicAdd(&cnt[(uint32_t)w * pp.R + (b >> kRangeLog)], 1u);
        recs[pos] = make_uint2(i | (d < 0 ? kSignBit : 0u), b);
      }
    }
  }
}

// Workgroup -> (partition, slice).  Workgroups go to the 8 XCDs round-robin by linear id, and every XCD has its own L2: all
// slices of a partition are given ids with the same (id mod 8), so the partition's window of `entries` (B / R buckets worth
// of 4-byte scattered writes) is assembled in ONE L2 instead of bouncing between eight.
GS_HD bool part_of_block(uint32_t id, const PlanParams& pp, uint32_t& w, uint32_t& r, uint32_t& s) {
  const uint32_t xcd = id & 7u, j = id >> 3;
  const uint32_t p = xcd + 8u * (j / pp.S);
  s = j % pp.S;
  w = p / pp.R; r = p % pp.R;
  return p < (uint32_t)pp.W * pp.R;
}

// the records of slice s of partition (w, r)
GS_HD void part_slice(const uint32_t* __restrict__ part_base, const PlanParams& pp, uint32_t w, uint32_t r, uint32_t s, uint32_t& lo, uint32_t& hi) {
  const uint32_t p0 = part_base[w * pp.R + r], p1 = part_base[w * pp.R + r + 1];
  const uint32_t len = p1 - p0, per = (len + pp.S - 1) / pp.S;
  lo = min(p1, p0 + s * per);
  hi = min(p1, lo + per);
}

__global__ void __launch_bounds__(kSortBlock) k_hist_part(const uint2* __restrict__ recs, const uint32_t* __restrict__ part_base, PlanParams pp,
                                                           uint32_t* __restrict__ hist) {
  wave_priority<GS_PRIO_PLAN>();
  extern __shared__ uint32_t sh[];
  uint32_t w, r, s;
  if (!part_of_block(blockIdx.x, pp, w, r, s)) return;
  const uint32_t base = r << kRangeLog;
  for (uint32_t b = threadIdx.x; b < kRangeBuckets; b += blockDim.x) sh[b] = 0;
  __syncthreads();
  uint32_t lo, hi;
  part_slice(part_base, pp, w, r, s, lo, hi);
  for (uint32_t e = lo + threadIdx.x; e < hi; e += blockDim.x) atomicAdd(&sh[recs[e].y - base], 1u);
  __syncthreads();
  uint32_t* out = hist + ((size_t)w * pp.S + s) * pp.B + base;
  for (uint32_t b = threadIdx.x; b < kRangeBuckets; b += blockDim.x) out[b] = sh[b];
}

__global__ void __launch_bounds__(kSortBlock) k_scatter_part(const uint2* __restrict__ recs, const uint32_t* __restrict__ part_base, PlanParams pp,
                                                              const uint32_t* __restrict__ hist, const uint32_t* __restrict__ offsets,
                                                              uint32_t* __restrict__ entries) {
  wave_priority<GS_PRIO_PLAN>();
  extern __shared__ uint32_t sh[];
  uint32_t w, r, s;
  if (!part_of_block(blockIdx.x, pp, w, r, s)) return;
  const uint32_t base = r << kRangeLog;
  const uint32_t* pre = hist + ((size_t)w * pp.S + s) * pp.B + base;
  for (uint32_t b = threadIdx.x; b < kRangeBuckets; b += blockDim.x) sh[b] = offsets[base + b] + pre[b];
  __syncthreads();
  uint32_t lo, hi;
  part_slice(part_base, pp, w, r, s, lo, hi);
  for (uint32_t e = lo + threadIdx.x; e < hi; e += blockDim.x) {
    const uint2 rec = recs[e];
    entries[atomicAdd(&sh[rec.y - base], 1u)] = rec.x | (w << kWindowShift);
  }
}

// ---- exclusive scan over uint32 (three small kernels) ----------------------------------------------
constexpr int kScanBlock = 256;
constexpr int kScanPerThread = 8;
constexpr int kScanTile = kScanBlock * kScanPerThread;

__global__ void __launch_bounds__(kScanBlock) k_scan_tiles(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                            uint32_t* __restrict__ tile_sums, uint32_t n) {
  wave_priority<GS_PRIO_PLAN>();
  __shared__ uint32_t sh[kScanBlock];
  const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * kScanPerThread;
  uint32_t v[kScanPerThread], sum = 0;
#pragma unroll
  for (int j = 0; j < kScanPerThread; ++j) { v[j] = (base + j < n) ? in[base + j] : 0u; sum += v[j]; }
  sh[threadIdx.x] = sum;
  __syncthreads();
  for (int off = 1; off < kScanBlock; off <<= 1) {     // Hillis-Steele inclusive scan of thread sums
    uint32_t t = (threadIdx.x >= (uint32_t)off) ? sh[threadIdx.x - off] : 0u;
    __syncthreads();
    sh[threadIdx.x] += t;
    __syncthreads();
  }
  uint32_t run = sh[threadIdx.x] - sum;                // exclusive prefix of this thread within the tile
#pragma unroll
  for (int j = 0; j < kScanPerThread; ++j) { if (base + j < n) out[base + j] = run; run += v[j]; }
  if (threadIdx.x == kScanBlock - 1) tile_sums[blockIdx.x] = sh[threadIdx.x];
}

__global__ void __launch_bounds__(1024) k_scan_tile_sums(uint32_t* __restrict__ tile_sums, uint32_t ntiles, uint32_t* __restrict__ total) {
  wave_priority<GS_PRIO_PLAN>();
  // single block: sequential-per-thread chunks + block scan; ntiles <= 1024 * 64
  __shared__ uint32_t sh[1024];
  const uint32_t per = (ntiles + 1023u) / 1024u;
  const uint32_t b0 = threadIdx.x * per;
  uint32_t sum = 0;
  for (uint32_t j = 0; j < per; ++j) if (b0 + j < ntiles) sum += tile_sums[b0 + j];
  sh[threadIdx.x] = sum;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    uint32_t t = (threadIdx.x >= (uint32_t)off) ? sh[threadIdx.x - off] : 0u;
    __syncthreads();
    sh[threadIdx.x] += t;
    __syncthreads();
  }
  uint32_t run = sh[threadIdx.x] - sum;
  for (uint32_t j = 0; j < per; ++j) if (b0 + j < ntiles) { uint32_t t = tile_sums[b0 + j]; tile_sums[b0 + j] = run; run += t; }
  if (threadIdx.x == 1023) *total = sh[1023];
}

__global__ void __launch_bounds__(kScanBlock) k_scan_add(uint32_t* __restrict__ out, const uint32_t* __restrict__ tile_sums, uint32_t n) {
  wave_priority<GS_PRIO_PLAN>();
  const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * kScanPerThread;
  const uint32_t add = tile_sums[blockIdx.x];
#pragma unroll
  for (int j = 0; j < kScanPerThread; ++j) if (base + j < n) out[base + j] += add;
}

// ---- chunk map (load balance) ------------------------------------------------------------------------------
// The sorted entry list is cut into chunks of `chunk` entries (16 or 32, a multiple of 4 chosen per plan -- see choose_chunk
// in msm.hip), ONE accumulate thread per chunk, whatever the
// bucket sizes are (uniform scalars: Poisson-sized buckets; real witnesses: 0/1-heavy ones).  A bucket that is
// cut by chunk boundaries is summed from per-chunk partials:  sum_{t = first}^{last-1} tail[t] + head[last].
// chunk_bucket[t] = bucket holding entry t*chunk.  Buckets cut into more than kHeavySpan + 1 pieces are listed
// for a block-wide tree combine (k_heavy_combine), the others are combined by whoever reads them.
constexpr uint32_t kHeavySpan = 64;
// (The heavy list has one slot per BUCKET -- a bucket is listed at most once -- so it cannot overflow; round 2's fixed cap of 2^16
// entries silently dropped buckets beyond it at c >= 18.)

__global__ void __launch_bounds__(256) k_chunk_map(const uint32_t* __restrict__ offsets, uint32_t nbuckets, uint32_t chunk,
                                                    uint32_t* __restrict__ chunk_bucket, uint32_t* __restrict__ heavy_list,
                                                    uint32_t* __restrict__ heavy_count) {
  wave_priority<GS_PRIO_PLAN>();
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nbuckets) return;
  const uint32_t o0 = offsets[b], o1 = offsets[b + 1];
  if (o1 == o0) return;
  const uint32_t t0 = (o0 + chunk - 1) / chunk, t1 = (o1 + chunk - 1) / chunk;   // chunks starting inside [o0, o1)
  for (uint32_t t = t0; t < t1; ++t) chunk_bucket[t] = b;
  if ((o1 - 1) / chunk - o0 / chunk > kHeavySpan) {
    heavy_list[atomicAdd(heavy_count, 1u)] = b;          // < nbuckets slots are ever taken
  }
}

// ---- bucket accumulation (dominant kernel) ---------------------------------------------------------
// grid.y = base array (job): up to 8 arrays share one plan (Groth16: At, BACGamma, BACDelta on w's plan).
struct AccJob {
  const uint32_t* table;      // window table rows[j][i] = 2^(c j) * P_i, packed affine, already offset to term 0's point
  uint32_t row_stride;        // points per table row
  uint32_t* buckets;          // nbuckets * kXyzzWords: (window, bucket) sums that live inside one chunk (or were tree-combined)
  uint32_t* heads;            // maxchunks * kXyzzWords: partial of the bucket that began in an earlier chunk and ends here
  uint32_t* tails;            // maxchunks * kXyzzWords: partial of the bucket that continues into the next chunk
  uint32_t* merged;           // B * kXyzzWords: bucket sums over all windows
  uint32_t* out;              // nblocks * 2 * kXyzzWords: per reduce-workgroup (A, S) pairs
  uint32_t* final_out;        // kXyzzWords: the job's result when the pairs are folded on the device (k_pair_reduce)
};
constexpr int kMaxJobs = 8;
struct AccJobs { AccJob j[kMaxJobs]; };

// waves per SIMD the register allocator must allow: G1 needs ~150 VGPRs (3 waves), G2 must fit 256 (2 waves;
// left alone it takes 264 and drops to ONE wave per SIMD, which halves the v_mad_u64_u32 issue rate)
template <class T> struct AccTuning;
#ifndef GS_G1_WAVES
#define GS_G1_WAVES 3
#endif
#ifndef GS_G1_PREFETCH
#define GS_G1_PREFETCH 1
#endif
#ifndef GS_G2_TOUCH
#define GS_G2_TOUCH 0            // measured: a one-word touch of the next point's line does not pay for G2 (3.96 vs 4.07 ms)
#endif
template <> struct AccTuning<FqTag> { static constexpr int kMinWaves = GS_G1_WAVES; static constexpr bool kRegisterPrefetch = GS_G1_PREFETCH != 0; static constexpr bool kTouch = true; };
#ifndef GS_G2_WAVES
#define GS_G2_WAVES 2
#endif
#ifndef GS_G2_PREFETCH
#define GS_G2_PREFETCH 0
#endif
template <> struct AccTuning<Fq2Tag> { static constexpr int kMinWaves = GS_G2_WAVES; static constexpr bool kRegisterPrefetch = GS_G2_PREFETCH != 0; static constexpr bool kTouch = GS_G2_TOUCH != 0; };

// Threads per workgroup of the accumulation kernels.  The kernel uses no LDS and no barrier (one thread per chunk), so the workgroup is only
// the unit the dispatcher places: a 256-thread group needs a free wave slot on all four SIMDs of a CU at once, a 64-thread group refills
// any slot the moment it frees (round 5's counters showed 2.65 of 3 wave slots occupied on average).
#ifndef GS_ACC_BLOCK
#define GS_ACC_BLOCK 256
#endif
constexpr int kAccBlock = GS_ACC_BLOCK;
template <class T>
__global__ void __launch_bounds__(kAccBlock, AccTuning<T>::kMinWaves) k_bucket_accumulate(AccJobs jobs, const uint32_t* __restrict__ offsets,
                                                            const uint32_t* __restrict__ entries,
                                                            const uint32_t* __restrict__ chunk_bucket, uint32_t nbuckets, uint32_t chunk) {
  constexpr int pw = PointIO<T>::kXyzzWords;
  constexpr int aw = PointIO<T>::kAffineWords;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t total = offsets[nbuckets];
  const uint32_t beg = t * chunk;
  if (beg >= total) return;
  const uint32_t end = min(beg + chunk, total);
  const AccJob job = jobs.j[blockIdx.y];
  uint32_t b = chunk_bucket[t];
  uint32_t bend = offsets[b + 1];
  bool started_before = offsets[b] < beg;
  // G2: the accumulator with y typed below 2p (ec.h, XyzzAcc: R then needs no reduction); G1 has no reductions to save
  constexpr bool kTight = GS_PAIR != 0 && T::kWords != 8;
  using Acc = std::conditional_t<kTight, XyzzAcc<T>, Xyzz<T>>;
  auto acc_inf = [] { if constexpr (kTight) return xyzz_acc_inf<T>(); else return xyzz_inf<T>(); };
  auto acc_store = [](uint32_t* p, const Acc& a) { if constexpr (kTight) store_xyzz<T>(p, to_xyzz(a)); else store_xyzz<T>(p, a); };
  Acc acc = acc_inf();
  const uint4* e4 = reinterpret_cast<const uint4*>(entries + beg);      // beg is a multiple of 4 entries = 16 B
  // Software pipeline: the table point of entry e+1 is requested before the 8M+2S of entry e, so the random
  // 64/128-byte gather (HBM miss ~900 cycles) is covered by ~4000 cycles of arithmetic of the same wave.
  // G1 keeps the whole next point in registers (5.9 vs 6.3 ms without); G2 is register-bound (256 VGPRs at 2 waves/SIMD) and
  // loads its point where it needs it.
  constexpr bool kPre = AccTuning<T>::kRegisterPrefetch;
  uint4 q = e4[0];
  uint32_t v = q.x;
  uint32_t nb = b;                                                       // bucket of entry `v`
  // the digit of window w multiplies 2^(c w) P_i = row w of the table: every window feeds the SAME bucket set
  const uint32_t* np = job.table + ((size_t)((v >> kWindowShift) & 31u) * job.row_stride + (v & kIndexMask)) * aw;
  RawAffine<T> nextp;
  uint32_t touch = 0;
  if constexpr (kPre) nextp = load_raw_affine<T>(np);
  for (uint32_t e = beg; e < end; ++e) {
    const uint32_t cur = v;
    const uint32_t* cp = np;
    RawAffine<T> curp;
    if constexpr (kPre) curp = nextp;
    if (e >= bend) {                                                     // bucket b is complete
      acc_store((started_before ? job.heads + (size_t)t * pw : job.buckets + (size_t)b * pw), acc);
      started_before = false;
      acc = acc_inf();
      b = nb; bend = offsets[b + 1];
    }
    if (e + 1 < end) {                                                   // issue the next gather
      const uint32_t k = (e + 1 - beg) & 3u;
      if (k == 0) q = e4[(e + 1 - beg) >> 2];
      v = k == 0 ? q.x : k == 1 ? q.y : k == 2 ? q.z : q.w;
      uint32_t nbend = offsets[nb + 1];
      while (nbend <= e + 1) { ++nb; nbend = offsets[nb + 1]; }
      np = job.table + ((size_t)((v >> kWindowShift) & 31u) * job.row_stride + (v & kIndexMask)) * aw;
      if constexpr (kPre) nextp = load_raw_affine<T>(np);
      else if constexpr (AccTuning<T>::kTouch) touch = *np;
    }
    if constexpr (!kPre) curp = load_raw_affine<T>(cp);
    const Affine<T> p = unpack_affine<T>(curp);
    xyzz_madd(acc, p, (cur & kSignBit) != 0);
    if constexpr (!kPre && AccTuning<T>::kTouch) asm volatile("" ::"v"(touch));   // keep the touch load alive until here
  }
  uint32_t* dst = (bend > end) ? job.tails + (size_t)t * pw
                               : (started_before ? job.heads + (size_t)t * pw : job.buckets + (size_t)b * pw);
  acc_store(dst, acc);
}

// ---- the tails: bucket combine + reduction ---------------------------------------------------------------------------------
// Every point addition of the tail kernels takes its second operand straight from memory (ec.h, xyzz_add_mem): a thread holds
// ONE point in registers plus the temporaries of the formula, whatever it adds up.  Round 3's form (both operands and, in the
// reduction, two running points in registers) took 317-512 VGPRs for G2 -- one wave per SIMD, beside which no accumulation wave
// fits; see DESIGN.md section 5 for what the new kernels take.

// the sum of bucket b, wherever its pieces are
template <class T>
GS_HD Xyzz<T> load_bucket(const AccJob& job, const uint32_t* __restrict__ offsets, uint32_t b, uint32_t chunk) {
  constexpr int pw = PointIO<T>::kXyzzWords;
  const uint32_t o0 = offsets[b], o1 = offsets[b + 1];
  if (o1 == o0) return xyzz_inf<T>();
  const uint32_t tf = o0 / chunk, tl = (o1 - 1) / chunk;
  if (tf == tl || tl - tf > kHeavySpan) return load_xyzz<T>(job.buckets + (size_t)b * pw);
  Xyzz<T> acc = load_xyzz<T>(job.heads + (size_t)tl * pw);
  for (uint32_t t = tf; t < tl; ++t) xyzz_add_mem<T>(acc, job.tails + (size_t)t * pw);
  return acc;
}

// Two instances of every tail kernel (round 4, profiles/r04_ab_tails_alone.txt):
//   kAlone = false  <= 256 VGPRs, so a tail wave shares its SIMD with an accumulation wave: right when the chip is FULL (2^19 terms and
//                   up: the tails are throughput work in the shadow of the accumulations);
//   kAlone = true   the backend pads the register count so that at most ONE such wave lives on a SIMD: right when the chip is mostly
//                   EMPTY (small MSMs: the tails are chains of dependent additions that set the pace, and the workgroup dispatcher
//                   otherwise packs several tail waves -- and the next proof's sort / NTT waves -- onto the same SIMDs while other
//                   CUs idle: 2^16 / 2^17 / 2^18 proofs 1.00 / 1.58 / 2.69 -> 0.93 / 1.42 / 2.46 ms).
#ifndef GS_TAIL_WAVES
#define GS_TAIL_WAVES 2
#endif
#define GS_TAIL_KERNEL(block, alone) \
  __global__ void __attribute__((amdgpu_flat_work_group_size(1, block), amdgpu_waves_per_eu((alone) ? 1 : GS_TAIL_WAVES, (alone) ? 1 : 8)))
// Heavy buckets (cut into more than kHeavySpan + 1 chunks: the buckets of the digits 0/1-heavy witnesses are full of) are summed by
// the whole grid, in two steps.  Round 4: a witness of the shape the reference's CalculateWitness produces puts a quarter of all entries
// into ONE bucket (262 545 entries = 8 205 chunks at 2^20); with one workgroup per bucket that was a chain of 64 dependent additions per
// thread on a single workgroup while 63 others idled -- 1.78 ms for G2, the longest kernel of such a proof.  Now every heavy bucket is cut
// into kHeavySlices slices of consecutive chunks; k_heavy_combine sums (bucket, slice) items grid-wide and parks each slice's sum in
// the slice's first chunk slot (its own, by then consumed), k_heavy_finish adds the <= 16 slice sums of a bucket with 16 lanes.
constexpr int kHeavyBlock = 128;
constexpr int kHeavySlices = 16;
// chunks [c0, c1) of slice s of a bucket whose pieces are chunks tf .. tl (empty when c0 > tl)
GS_HD void heavy_slice(uint32_t tf, uint32_t tl, uint32_t s, uint32_t& c0, uint32_t& c1) {
  const uint32_t len = (tl - tf + 1 + kHeavySlices - 1) / kHeavySlices;
  c0 = tf + s * len;
  c1 = min(c0 + len, tl + 1);
}
// where chunk t keeps its piece of a bucket that spans tf .. tl: the last chunk holds a head, the others tails
GS_HD uint32_t* heavy_slot(const AccJob& job, uint32_t t, uint32_t tl, int pw) { return (t == tl ? job.heads : job.tails) + (size_t)t * pw; }

template <class T, bool kAlone>
GS_TAIL_KERNEL(kHeavyBlock, kAlone) k_heavy_combine(AccJobs jobs, const uint32_t* __restrict__ offsets,
                                                                const uint32_t* __restrict__ heavy_list,
                                                                const uint32_t* __restrict__ heavy_count, uint32_t chunk) {
  wave_priority<GS_PRIO_TAIL>();
  constexpr int pw = PointIO<T>::kXyzzWords;
  __shared__ uint32_t sh[kHeavyBlock * pw];
  const AccJob job = jobs.j[blockIdx.y];
  const uint32_t nitems = *heavy_count * (uint32_t)kHeavySlices;
  for (uint32_t item = blockIdx.x; item < nitems; item += gridDim.x) {
    const uint32_t b = heavy_list[item / kHeavySlices];
    const uint32_t tf = offsets[b] / chunk, tl = (offsets[b + 1] - 1) / chunk;
    uint32_t c0, c1;
    heavy_slice(tf, tl, item % kHeavySlices, c0, c1);
    if (c0 > tl) continue;                                         // (uniform over the workgroup)
    Xyzz<T> acc = xyzz_inf<T>();
    for (uint32_t t = c0 + threadIdx.x; t < c1; t += kHeavyBlock) xyzz_add_mem<T>(acc, heavy_slot(job, t, tl, pw));
    store_xyzz<T>(sh + threadIdx.x * pw, acc);
    __syncthreads();
    for (int half = kHeavyBlock / 2; half >= 1; half >>= 1) {
      if ((int)threadIdx.x < half) {                           // reads [half, 2 half), writes [0, half): no hazard inside a level
        xyzz_add_mem<T>(acc, sh + (threadIdx.x + half) * pw);
        store_xyzz<T>(sh + threadIdx.x * pw, acc);
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) store_xyzz<T>(heavy_slot(job, c0, tl, pw), acc);      // the slice's own first slot: every piece of it has been read
    __syncthreads();
  }
}
// buckets[b] = the sum of the slice sums of heavy bucket b: 16 lanes per bucket, 8 buckets per workgroup
template <class T, bool kAlone>
GS_TAIL_KERNEL(kHeavyBlock, kAlone) k_heavy_finish(AccJobs jobs, const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ heavy_list,
                                                   const uint32_t* __restrict__ heavy_count, uint32_t chunk) {
  wave_priority<GS_PRIO_TAIL>();
  constexpr int pw = PointIO<T>::kXyzzWords;
  constexpr uint32_t per = kHeavyBlock / kHeavySlices;
  __shared__ uint32_t sh[kHeavyBlock * pw];
  const AccJob job = jobs.j[blockIdx.y];
  const uint32_t nheavy = *heavy_count;
  const uint32_t lane = threadIdx.x % kHeavySlices;
  for (uint32_t g = blockIdx.x; g * per < nheavy; g += gridDim.x) {
    const uint32_t h = g * per + threadIdx.x / kHeavySlices;
    Xyzz<T> acc = xyzz_inf<T>();
    uint32_t b = 0;
    if (h < nheavy) {
      b = heavy_list[h];
      const uint32_t tf = offsets[b] / chunk, tl = (offsets[b + 1] - 1) / chunk;
      uint32_t c0, c1;
      heavy_slice(tf, tl, lane, c0, c1);
      if (c0 <= tl) acc = load_xyzz<T>(heavy_slot(job, c0, tl, pw));
    }
    store_xyzz<T>(sh + threadIdx.x * pw, acc);
    __syncthreads();
    for (int half = kHeavySlices / 2; half >= 1; half >>= 1) {
      if ((int)lane < half) {
        xyzz_add_mem<T>(acc, sh + (threadIdx.x + half) * pw);
        store_xyzz<T>(sh + threadIdx.x * pw, acc);
      }
      __syncthreads();
    }
    if (lane == 0 && h < nheavy) store_xyzz<T>(job.buckets + (size_t)b * pw, acc);
    __syncthreads();
  }
}

// merged[b] = the pieces of bucket b (it spans ~ n W / (B * chunk) chunks).  Thanks to the window tables all W
// digit positions share one bucket set, so the MSM is simply sum_b (b + 1) * merged[b]: no per-window
// reduction and no Horner recombination.  Thread (0, 0, 0) also leaves what the plan found for gs_timing:
// stats[0] = bucket entries (= non-zero digits = additions one base array costs), stats[1] = buckets combined by the heavy tree.
template <class T, bool kAlone>
GS_TAIL_KERNEL(256, kAlone) k_bucket_combine(AccJobs jobs, const uint32_t* __restrict__ offsets, uint32_t B, uint32_t chunk,
                                                                     const uint32_t* __restrict__ heavy_count, uint32_t* __restrict__ stats) {
  wave_priority<GS_PRIO_TAIL>();
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b == 0 && blockIdx.y == 0) { stats[0] = offsets[B]; stats[1] = *heavy_count; }
  if (b >= B) return;
  const AccJob job = jobs.j[blockIdx.y];
  store_xyzz<T>(job.merged + (size_t)b * PointIO<T>::kXyzzWords, load_bucket<T>(job, offsets, b, chunk));
}

// One workgroup of 256 threads reduces 256 * L consecutive buckets to the pair
//   A = sum_j (j + 1) * merged[base + j],  S = sum_j merged[base + j]        (j local to the workgroup)
// Thread t owns L buckets.  Pass 1 turns them IN PLACE into their suffix sums run_j = sum_{j' >= j} merged[j'] (L - 1 additions),
// pass 2 adds the suffix sums up: acc_t = sum_j run_j = sum_j (j + 1) merged[j] (L - 1 additions, parked in the thread's last
// slot) -- one point in registers in either pass.  Then a suffix scan of run_0 over the workgroup in LDS gives
// R_t = sum_{t' >= t} run_0(t'), and sum_t t * run_0(t) = sum_{t >= 1} R_t; value_t = acc_t + L R_t, and a tree sum.  The host adds
// the (at most 32) pairs: result = sum_blk A_blk + 256 L * sum_blk blk * S_blk.
constexpr int kReduceBlock = 256;
template <class T, bool kAlone>
GS_TAIL_KERNEL(kReduceBlock, kAlone) k_block_reduce(AccJobs jobs, uint32_t B, int L) {
  wave_priority<GS_PRIO_TAIL>();
  constexpr int pw = PointIO<T>::kXyzzWords;
  __shared__ uint32_t sh[kReduceBlock * pw];
  const AccJob job = jobs.j[blockIdx.y];
  const uint32_t t = threadIdx.x;
  const uint32_t base = (blockIdx.x * kReduceBlock + t) * (uint32_t)L;
  const bool any = base < B;                                   // (B and L are powers of two: all L buckets exist or none)
  uint32_t* mine = job.merged + (size_t)base * pw;
  Xyzz<T> R = xyzz_inf<T>();
  if (any) {
    R = load_xyzz<T>(mine + (size_t)(L - 1) * pw);
    for (int j = L - 2; j >= 0; --j) {                         // pass 1: suffix sums, in place
      xyzz_add_mem<T>(R, mine + (size_t)j * pw);
      store_xyzz<T>(mine + (size_t)j * pw, R);
    }
    if (L > 1) {                                               // pass 2: their sum, parked in the last slot (L = 1: it is there already)
      GS_MEM_FENCE();
      Xyzz<T> acc = load_xyzz<T>(mine + (size_t)(L - 1) * pw);
      for (int j = L - 2; j >= 0; --j) xyzz_add_mem<T>(acc, mine + (size_t)j * pw);
      store_xyzz<T>(mine + (size_t)(L - 1) * pw, acc);
      GS_MEM_FENCE();
      R = load_xyzz<T>(mine);                                  // run_0 again
    }
  }
  // suffix (inclusive) scan of run_0 over the workgroup: the additions read their operand from LDS, the writes follow a barrier
  store_xyzz<T>(sh + t * pw, R);
  __syncthreads();
  for (int off = 1; off < kReduceBlock; off <<= 1) {
    const bool has = t + (uint32_t)off < (uint32_t)kReduceBlock;
    if (has) xyzz_add_mem<T>(R, sh + (t + off) * pw);
    __syncthreads();
    if (has) store_xyzz<T>(sh + t * pw, R);
    __syncthreads();
  }
  if (t == 0) store_xyzz<T>(job.out + ((size_t)blockIdx.x * 2 + 1) * pw, R);     // S = R_0
  // value_t = acc_t + L * R_t (t >= 1), acc_0 for t = 0
  if (t >= 1) { for (int l = L; l > 1; l >>= 1) xyzz_dbl(R); }
  else R = xyzz_inf<T>();
  if (any) xyzz_add_mem<T>(R, mine + (size_t)(L - 1) * pw);
  __syncthreads();
  store_xyzz<T>(sh + t * pw, R);
  __syncthreads();
  for (int half = kReduceBlock / 2; half >= 1; half >>= 1) {
    if ((int)t < half) {
      xyzz_add_mem<T>(R, sh + (t + half) * pw);
      store_xyzz<T>(sh + t * pw, R);
    }
    __syncthreads();
  }
  if (t == 0) store_xyzz<T>(job.out + (size_t)blockIdx.x * 2 * pw, R);          // A
}

// Second level, for wide windows (more than 32 reduce workgroups per job): one workgroup per job folds the nblk <= 256
// pairs (A_blk, S_blk) into the job's result  sum_blk A_blk + (256 L) * sum_blk blk * S_blk  -- the suffix-scan identity
// again (sum_blk blk * S_blk = sum_{blk >= 1} R_blk with R_blk = sum_{b' >= blk} S_b'), then log2(256 L) doublings -- so the
// host receives ONE point per job however many buckets there were.
template <class T, bool kAlone>
GS_TAIL_KERNEL(kReduceBlock, kAlone) k_pair_reduce(AccJobs jobs, uint32_t nblk, int log2_span) {
  wave_priority<GS_PRIO_TAIL>();
  constexpr int pw = PointIO<T>::kXyzzWords;
  __shared__ uint32_t sh[kReduceBlock * pw];
  const AccJob job = jobs.j[blockIdx.y];
  const uint32_t t = threadIdx.x;
  Xyzz<T> R = xyzz_inf<T>();
  if (t < nblk) R = load_xyzz<T>(job.out + ((size_t)t * 2 + 1) * pw);
  store_xyzz<T>(sh + t * pw, R);
  __syncthreads();
  for (int off = 1; off < kReduceBlock; off <<= 1) {                 // inclusive suffix scan of S
    const bool has = t + (uint32_t)off < (uint32_t)kReduceBlock;
    if (has) xyzz_add_mem<T>(R, sh + (t + off) * pw);
    __syncthreads();
    if (has) store_xyzz<T>(sh + t * pw, R);
    __syncthreads();
  }
  if (t == 0) R = xyzz_inf<T>();                                     // the sum runs over blk >= 1
  for (int pass = 0; pass < 2; ++pass) {                             // tree sums: first of R (the weighted part), then of A
    if (pass == 1) {
      if (t == 0) {                                                  // (256 L) * sum_{blk >= 1} R_blk, kept in LDS slot 0's place: park it in `final_out`
        for (int k = 0; k < log2_span; ++k) xyzz_dbl(R);
        store_xyzz<T>(job.final_out, R);
      }
      R = xyzz_inf<T>();
      if (t < nblk) R = load_xyzz<T>(job.out + (size_t)t * 2 * pw);
    }
    __syncthreads();
    store_xyzz<T>(sh + t * pw, R);
    __syncthreads();
    for (int half = kReduceBlock / 2; half >= 1; half >>= 1) {
      if ((int)t < half) {
        xyzz_add_mem<T>(R, sh + (t + half) * pw);
        store_xyzz<T>(sh + t * pw, R);
      }
      __syncthreads();
    }
  }
  if (t == 0) {
    GS_MEM_FENCE();
    xyzz_add_mem<T>(R, job.final_out);
    store_xyzz<T>(job.final_out, R);
  }
}

// ---- window tables ----------------------------------------------------------------------------------------
// rows[j][i] = 2^(c j) * P_i for j < W, packed affine.  Built once per (base array, c) and kept in HBM: a
// 2^20-point G1 array costs 64 MiB per row, 1 GiB for c = 16 -- the trade the 288 GB of HBM3E is there for.
template <class T>
__global__ void __launch_bounds__(256) k_build_table(const uint32_t* __restrict__ row0, uint32_t n, int c, int W, uint32_t* __restrict__ rows) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr int aw = PointIO<T>::kAffineWords;
  Affine<T> a = PointIO<T>::load_affine(row0 + (size_t)i * aw);
  if (rows != row0) PointIO<T>::store_affine(rows + (size_t)i * aw, a);
  for (int j = 1; j < W; ++j) {
    if (!is_inf(a)) {
      Xyzz<T> x = xyzz_dbl_affine<T>(a.x, relax<2>(a.y));
      for (int k = 1; k < c; ++k) xyzz_dbl(x);
      a = xyzz_to_affine(x);
    }
    PointIO<T>::store_affine(rows + ((size_t)j * n + i) * aw, a);
  }
}

// The same table with ONE field inversion per point instead of one per row (Montgomery's trick along the rows of a
// point): the forward sweep keeps doubling in XYZZ and parks every row's point and the running product of the ZZZ's in
// a scratch slab, the backward sweep peels the individual inverses off the inverted product and writes the affine rows.
// ~2000 field products per point instead of ~6000 (15 Fermat inversions cost more than the 240 doublings).
// Points that reach infinity while doubling (only possible outside the order-r subgroup) take the per-row path above.
template <class T>
__global__ void __launch_bounds__(256) k_build_table_batched(const uint32_t* __restrict__ row0, uint32_t n, uint32_t first, uint32_t count, int c, int W,
                                                              uint32_t* __restrict__ rows, uint32_t* __restrict__ scratch) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count) return;
  const uint32_t i = first + t;
  constexpr int aw = PointIO<T>::kAffineWords, pw = PointIO<T>::kXyzzWords, ew = pw / 4, sw = pw + ew;
  using E2 = typename T::template E<2>;
  Affine<T> a = PointIO<T>::load_affine(row0 + (size_t)i * aw);
  if (rows != row0) PointIO<T>::store_affine(rows + (size_t)i * aw, a);
  if (W <= 1) return;
  if (is_inf(a)) {
    for (int j = 1; j < W; ++j) PointIO<T>::store_affine(rows + ((size_t)j * n + i) * aw, a);
    return;
  }
  // Round 6: JACOBIAN doublings (ec.h, jac_dbl: 3M + 4S = 990 multiply-adds against the XYZZ doubling's 1350) -- a table row is 14 x 17
  // doublings of one point and never an addition; per row the point is parked as (X, Y, Z) with the running product of the Z's, one
  // inversion per point turns all rows affine (x = X / Z^2, y = Y / Z^3).
  constexpr int cw = pw / 4;                                     // words per coordinate; a parked row = X | Y | Z | prefix (4 of the sw = 5 cw)
  Jac<T> x = jac_from_affine<T>(a);
  for (int k = 0; k < c; ++k) jac_dbl(x);
  bool degenerate = is_inf(x);
  E2 pref = reduce2(x.z);
  auto park = [&](int j) {
    uint32_t* s = scratch + ((size_t)(j - 1) * count + t) * sw;
    PointIO<T>::store_limbs(s, x.x); PointIO<T>::store_limbs(s + cw, x.y); PointIO<T>::store_limbs(s + 2 * cw, x.z);
    PointIO<T>::store_limbs(s + 3 * cw, pref);
  };
  park(1);
  for (int j = 2; j < W && !degenerate; ++j) {
    for (int k = 0; k < c; ++k) jac_dbl(x);
    degenerate = is_inf(x);
    pref = smul<T>(pref, x.z);
    park(j);
  }
  if (degenerate) {                                             // rare: redo this point row by row
    for (int j = 1; j < W; ++j) {
      if (!is_inf(a)) {
        Xyzz<T> y = xyzz_dbl_affine<T>(a.x, relax<2>(a.y));
        for (int k = 1; k < c; ++k) xyzz_dbl(y);
        a = xyzz_to_affine(y);
      }
      PointIO<T>::store_affine(rows + ((size_t)j * n + i) * aw, a);
    }
    return;
  }
  E2 itot = inv(pref);                                          // 1 / (Z_1 ... Z_{W-1})
  for (int j = W - 1; j >= 1; --j) {
    const uint32_t* s = scratch + ((size_t)(j - 1) * count + t) * sw;
    Jac<T> xj;
    PointIO<T>::load_limbs(s, xj.x); PointIO<T>::load_limbs(s + cw, xj.y); PointIO<T>::load_limbs(s + 2 * cw, xj.z);
    E2 iz = itot;                                               // 1 / Z_j
    if (j > 1) {
      E2 before;
      PointIO<T>::load_limbs(scratch + ((size_t)(j - 2) * count + t) * sw + 3 * cw, before);
      iz = smul<T>(itot, before);
      itot = smul<T>(itot, xj.z);
    }
    PointIO<T>::store_affine(rows + ((size_t)j * n + i) * aw, jac_to_affine_with_inverse<T>(xj, iz));
  }
}

// ---- base-array preparation ----------------------------------------------------------------------------
// Jacobian standard-form triples -> packed canonical Montgomery affine       [g1.go:157-170]
// off_curve[0] counts the points that are not on their curve, off_curve[1] keeps the smallest such index
template <class T>
__global__ void __launch_bounds__(256) k_jacobian_to_affine(const uint32_t* __restrict__ jac, uint32_t n, uint32_t* __restrict__ out,
                                                             uint32_t* __restrict__ off_curve) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr int cw = PointIO<T>::kCoordWords;
  const uint32_t* p = jac + (size_t)i * 3 * cw;
  auto X = PointIO<T>::load_std(p), Y = PointIO<T>::load_std(p + cw), Z = PointIO<T>::load_std(p + 2 * cw);
  // Z = 1 (a key that was normalised when it was written -- utils.GrothSetupToBinary, gs_g*_download): x = X, y = Y, no inversion.  A wave
  // whose 64 points all have it skips the ~380 products of the Fermat inversion altogether: 2^20 G1 points 2.6 -> 0.3 ms; the reference's
  // own setup leaves Z != 1 (groth16.go:139-175) and pays the inversion per point as before.
  bool z_is_one = p[2 * cw] == 1u;
#pragma unroll
  for (int k = 1; k < cw; ++k) z_is_one = z_is_one && p[2 * cw + k] == 0u;
  Affine<T> a;
  if (z_is_one) { a.x = canon(X); a.y = canon(Y); }
  else a = jacobian_to_affine<T>(X, Y, Z);
  if (!on_curve(a)) {
    atomicAdd(off_curve, 1u);
    atomicMin(off_curve + 1, i);
  }
  PointIO<T>::store_affine(out + (size_t)i * PointIO<T>::kAffineWords, a);
}

// packed affine -> standard-form Jacobian triple [x, y, 1] / [0, 0, 0]
template <class T>
__global__ void __launch_bounds__(256) k_affine_to_jacobian_std(const uint32_t* __restrict__ aff, uint32_t n, uint32_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr int cw = PointIO<T>::kCoordWords;
  Affine<T> a = PointIO<T>::load_affine(aff + (size_t)i * PointIO<T>::kAffineWords);
  uint32_t* o = out + (size_t)i * 3 * cw;
  for (int k = 0; k < 3 * cw; ++k) o[k] = 0;
  if (is_inf(a)) return;
  PointIO<T>::store_std(o, a.x);
  PointIO<T>::store_std(o + cw, a.y);
  o[2 * cw] = 1u;
}

// fixed-base batch: out[i] = k_i * G with 8-bit windows: win[w][d] = d * 2^(8 w) * G (affine, packed; d = 0 is infinity),
// 32 windows x 256 entries = 512 KiB for G1 (cache resident): 32 mixed additions per scalar instead of one per scalar bit
// (a wave of the bit-serial loop paid all 254, since some lane always has the bit set).
template <class T>
__global__ void __launch_bounds__(256) k_fixed_base_mul(const uint32_t* __restrict__ scalars, uint32_t n,
                                                         const uint32_t* __restrict__ win, uint32_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t k[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) k[j] = scalars[(size_t)i * 8 + j];
  scalar_canon(k);
  Xyzz<T> acc = xyzz_inf<T>();
  for (int w = 0; w < 32; ++w) {
    const uint32_t d = (k[w >> 2] >> ((w & 3) * 8)) & 0xffu;
    if (d) {
      Affine<T> p = PointIO<T>::load_affine(win + ((size_t)w * 256 + d) * PointIO<T>::kAffineWords);
      xyzz_madd(acc, p, false);
    }
  }
  Affine<T> a = xyzz_to_affine(acc);
  PointIO<T>::store_affine(out + (size_t)i * PointIO<T>::kAffineWords, a);
}
// win[w][d] = d * 2^(8 w) * G from pow2[j] = 2^j * G (8192 threads, once per process)
template <class T>
__global__ void __launch_bounds__(256) k_build_fixed_window_table(const uint32_t* __restrict__ pow2, uint32_t* __restrict__ win) {
  const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 32u * 256u) return;
  const uint32_t w = idx >> 8, d = idx & 255u;
  Xyzz<T> acc = xyzz_inf<T>();
  for (int j = 0; j < 8; ++j)
    if ((d >> j) & 1u) xyzz_madd(acc, PointIO<T>::load_affine(pow2 + (size_t)(8 * w + j) * PointIO<T>::kAffineWords), false);
  PointIO<T>::store_affine(win + (size_t)idx * PointIO<T>::kAffineWords, xyzz_to_affine(acc));
}

// table[j] = 2^j * G, j < 256 (once per process).  One workgroup of 256 threads: thread 0 walks the 255 doublings in XYZZ
// (no inversion on the chain), then every thread normalises one row -- 256 inversions side by side instead of in sequence.
template <class T>
__global__ void __launch_bounds__(256) k_build_pow2_table(uint32_t* __restrict__ table, uint32_t* __restrict__ chain /* 256 x kXyzzWords */) {
  if (blockIdx.x != 0) return;
  constexpr int pw = PointIO<T>::kXyzzWords;
  if (threadIdx.x == 0) {
    Affine<T> g;
    if constexpr (PointIO<T>::kAffineWords == 16) {
#pragma unroll
      for (int i = 0; i < NL; ++i) { g.x.l[i] = Gen::g1x(i); g.y.l[i] = Gen::g1y(i); }
    } else {
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        g.x.c0.l[i] = Gen::g2x0(i); g.x.c1.l[i] = Gen::g2x1(i);
        g.y.c0.l[i] = Gen::g2y0(i); g.y.c1.l[i] = Gen::g2y1(i);
      }
    }
    Xyzz<T> acc = xyzz_from_affine(g);
    for (int j = 0; j < 256; ++j) {
      store_xyzz<T>(chain + (size_t)j * pw, acc);
      xyzz_dbl(acc);
    }
  }
  __threadfence_block();
  __syncthreads();
  const Xyzz<T> mine = load_xyzz<T>(chain + (size_t)threadIdx.x * pw);
  PointIO<T>::store_affine(table + (size_t)threadIdx.x * PointIO<T>::kAffineWords, xyzz_to_affine(mine));
}

}  // namespace gs
