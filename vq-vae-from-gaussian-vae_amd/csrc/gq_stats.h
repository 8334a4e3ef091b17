// gq_stats.h -- order-independent accumulation of the GroupNorm statistics.
//
// The conv-stack kernels leave the statistics of the NEXT GroupNorm (sum and sum of squares per (image, group)) with
// their output, accumulated across threads and blocks by atomics.  A floating-point atomic sum depends on the order
// in which the adds land, so the statistics -- and with them the encoder's z and, at a near-tie, a token -- could
// differ between two runs of the same input (the reference's CPU path is deterministic: pit/quantization/
// gaussian.py:136-150 sees one z per image).  Here every addend (one thread's fp32 partial sum) is converted EXACTLY
// to a 120-bit fixed-point number held in three signed 64-bit limbs of 40 payload bits each, and the limbs are summed
// with integer atomics: integer addition is associative, so the total is the exact sum of the addends whatever the
// order, and the statistics are bit-reproducible by construction.
//
//   value = q[0] * 2^-56 + q[1] * 2^-16 + q[2] * 2^24,   |q[k]| < 2^40 per addend
//
// Range: |addend| < 2^64 (a sum of squares of fp32 activations up to ~4e9 per element at 1 element per thread);
// bits below 2^-56 of an addend are truncated toward zero (a function of the addend alone, so still order-independent;
// 2^-56 is 36 binary orders below the eps = 1e-6 of every GroupNorm of this UNet).  24 spare bits per limb allow 2^23
// addends per statistic.  An addend that is not finite or not below 2^64 poisons the record (q[6] != 0) and every
// reader then sees NaN -- loud, like the fp64 sum it replaces.
//
// Record per (image, group): 8 x int64 = {sum: q0 q1 q2, sum of squares: q0 q1 q2, poison, unused} (64 bytes).
#pragma once
#ifndef GQ_STATS_HOST_TEST          // tests/stats_host_test.cpp compiles the arithmetic for the host with its own shims
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

namespace gqhip {

constexpr int kStatWords = 8;   // int64 words per (image, group) record

// the three limbs of one addend
__device__ __forceinline__ void stat_split(double v, int64_t (&q)[3], bool &poison) {
  const double a = __builtin_fabs(v);
  poison = !(a < 18446744073709551616.0);                 // NaN, inf, >= 2^64
  const double a2 = __builtin_trunc(a * 5.9604644775390625e-08);             // a / 2^24
  double r = a - a2 * 16777216.0;                                           // exact: < 2^24
  const double a1 = __builtin_trunc(r * 65536.0);                           // r / 2^-16
  r = r * 65536.0 - a1;                                                     // exact: < 1 (in units of 2^-16)
  const double a0 = __builtin_trunc(r * 1099511627776.0);                   // units of 2^-56
  const bool neg = v < 0.0;
  q[2] = poison ? 0 : (neg ? -(int64_t)a2 : (int64_t)a2);
  q[1] = poison ? 0 : (neg ? -(int64_t)a1 : (int64_t)a1);
  q[0] = poison ? 0 : (neg ? -(int64_t)a0 : (int64_t)a0);
}

__device__ __forceinline__ void stat_atomic_add(int64_t *dst, int64_t v) {
  if (v != 0) atomicAdd(reinterpret_cast<unsigned long long *>(dst), (unsigned long long)v);
}

// Add one addend pair (a thread's fp32 partial sums, or a fixed-order combination of them) to a record in LDS or
// global memory.
__device__ __forceinline__ void stat_add(int64_t *rec, double s, double ss) {
  int64_t q[3];
  bool bad, bad2;
  stat_split(s, q, bad);
  stat_atomic_add(rec + 0, q[0]);
  stat_atomic_add(rec + 1, q[1]);
  stat_atomic_add(rec + 2, q[2]);
  stat_split(ss, q, bad2);
  stat_atomic_add(rec + 3, q[0]);
  stat_atomic_add(rec + 4, q[1]);
  stat_atomic_add(rec + 5, q[2]);
  if (bad) atomicAdd(reinterpret_cast<unsigned long long *>(rec + 6), 1ull);
  if (bad2) atomicAdd(reinterpret_cast<unsigned long long *>(rec + 6), 1ull);
}

// The same for a thread's fp32 partial sums (the common case), by integer arithmetic on the fp32 bits: the 24-bit
// significand m of v = m 2^(e - 150) lands at bit (e - 94) of the fixed-point number, i.e. in limb k = (e - 94) / 40 and,
// when it straddles, limb k + 1 -- two shifts and at most two atomics instead of ~80 fp64 instructions.  Bit-identical
// to stat_split((double)v) (tests/test_gpu_convstack_kernels.py compares the kernels that use either form).
__device__ __forceinline__ void stat_add_one_f32(int64_t *limbs, int64_t *poison, float v) {
  const unsigned bits = __float_as_uint(v);
  const int e = (int)((bits >> 23) & 0xffu);
  if (e >= 191) {                                     // |v| >= 2^64, inf, NaN
    atomicAdd(reinterpret_cast<unsigned long long *>(poison), 1ull);
    return;
  }
  const unsigned long long m = (unsigned long long)((bits & 0x7fffffu) | (e ? 0x800000u : 0u));
  const int shift = (e ? e : 1) - 94;                 // v = +- m 2^(shift - 56)
  if (shift <= -24) return;                           // entirely below 2^-56
  unsigned long long lo, hi = 0ull;
  int k = 0;
  if (shift < 0) {
    lo = m >> (-shift);                               // truncation toward zero, as stat_split
  } else {
    k = shift >= 80 ? 2 : (shift >= 40 ? 1 : 0);
    const unsigned long long t = m << (shift - 40 * k);          // < 2^64: the in-limb offset is < 40 and m < 2^24
    lo = t & 0xffffffffffull;
    hi = t >> 40;                                     // 0 when k == 2 (offset <= 16)
  }
  const bool neg = (bits >> 31) != 0u;
  stat_atomic_add(limbs + k, neg ? -(int64_t)lo : (int64_t)lo);
  if (hi) stat_atomic_add(limbs + k + 1, neg ? -(int64_t)hi : (int64_t)hi);
}
__device__ __forceinline__ void stat_add_f32(int64_t *rec, float s, float ss) {
#ifdef GQHIP_STATS_SPLIT_F64   // diagnostic build (A/B timing): the fp64 split for fp32 addends too
  stat_add(rec, (double)s, (double)ss);
#else
  stat_add_one_f32(rec, rec + 6, s);
  stat_add_one_f32(rec + 3, rec + 6, ss);
#endif
}

// word `w` of a block's LDS records -> the same word of the global records (one thread per word)
__device__ __forceinline__ void stat_flush_word(int64_t *global_word, int64_t local_word) {
  stat_atomic_add(global_word, local_word);
}

// (sum, sum of squares) of a finished record, as doubles; a fixed evaluation order, the same for every reader
__device__ __forceinline__ void stat_load(const int64_t *rec, double &s, double &ss) {
  s = ((double)rec[2] * 16777216.0 + (double)rec[1] * 1.52587890625e-05) + (double)rec[0] * 1.3877787807814457e-17;
  ss = ((double)rec[5] * 16777216.0 + (double)rec[4] * 1.52587890625e-05) + (double)rec[3] * 1.3877787807814457e-17;
  if (rec[6] != 0) {
    s = __builtin_nan("");
    ss = __builtin_nan("");
  }
}

}  // namespace gqhip
