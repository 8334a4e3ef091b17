// Host-side check of csrc/gq_stats.h (CPU suite, no GPU): the integer split of an fp32 addend (stat_add_f32) is
// bit-identical to the fp64 split (stat_add), sums are order-independent, the value read back is the exact sum of the
// addends down to 2^-56, and out-of-range addends poison the record.  Built and run by tests/test_host.py.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#define GQ_STATS_HOST_TEST 1
#define __device__
#define __forceinline__ inline
static inline unsigned __float_as_uint(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }
static inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { const unsigned long long o = *p; *p = o + v; return o; }
#include "gq_stats.h"

using namespace gqhip;

int main() {
  std::mt19937_64 rng(12345);
  std::vector<float> vals;
  const float edges[] = {0.f, -0.f, 1.f, -1.f, 1e-30f, 1.17549435e-38f, 1e-45f, 5.4210109e-20f /* 2^-64 */, 1.3877788e-17f /* 2^-56 */,
                         2.7755576e-17f, 1.52587890625e-05f, 0.99999994f, 16777216.f, 16777217.f, 1.09951163e12f /* 2^40 */, 1.8446743e19f /* ~2^64- */,
                         9.223372e18f, -9.223372e18f, 3.4e38f, INFINITY, -INFINITY, NAN};
  for (float e : edges) vals.push_back(e);
  for (int i = 0; i < 200000; ++i) {
    const unsigned bits = (unsigned)rng();
    float f; std::memcpy(&f, &bits, 4);
    vals.push_back(f);
  }
  std::uniform_real_distribution<float> uni(-8.f, 8.f);
  for (int i = 0; i < 100000; ++i) vals.push_back(uni(rng) * uni(rng));
  // 1. per addend: integer split == fp64 split
  long bad = 0;
  for (float v : vals) {
    int64_t a[kStatWords] = {0}, b[kStatWords] = {0};
    stat_add(a, (double)v, (double)v);
    stat_add_f32(b, v, v);
    if (std::memcmp(a, b, sizeof(a)) != 0) {
      if (bad < 5) std::printf("mismatch at %a: f64 {%lld %lld %lld | p %lld} int {%lld %lld %lld | p %lld}\n", (double)v, (long long)a[0], (long long)a[1],
                               (long long)a[2], (long long)a[6], (long long)b[0], (long long)b[1], (long long)b[2], (long long)b[6]);
      ++bad;
    }
  }
  if (bad) { std::printf("FAIL: %ld addends split differently\n", bad); return 1; }
  // 2. order independence + exactness: finite moderate values, forward vs reversed vs shuffled; value == long double sum of truncated addends
  std::vector<float> fin;
  for (float v : vals) if (std::isfinite(v) && std::fabs(v) < 1e15f) fin.push_back(v);
  int64_t r0[kStatWords] = {0}, r1[kStatWords] = {0};
  for (size_t i = 0; i < fin.size(); ++i) stat_add_f32(r0, fin[i], fin[i] * fin[i] < 1e18f ? fin[i] * fin[i] : 0.f);
  for (size_t i = fin.size(); i-- > 0;) stat_add_f32(r1, fin[i], fin[i] * fin[i] < 1e18f ? fin[i] * fin[i] : 0.f);
  if (std::memcmp(r0, r1, sizeof(r0)) != 0) { std::printf("FAIL: order dependence\n"); return 1; }
  __int128 exact = 0;                                 // in units of 2^-56; |v| < 2^50 -> each addend < 2^106
  for (float v : fin) {
    int ex;
    const double fr = std::frexp((double)v, &ex);     // v = fr 2^ex, |fr| in [0.5, 1): 24 significant bits
    const long long m = (long long)std::ldexp(fr, 24);          // exact integer significand
    const int sh = ex - 24 + 56;
    if (sh >= 0) exact += (__int128)m << sh;
    else if (sh > -63) exact += (__int128)(m < 0 ? -((-m) >> -sh) : m >> -sh);   // truncation toward zero
  }
  const __int128 back = ((__int128)r0[2] << 80) + ((__int128)r0[1] << 40) + (__int128)r0[0];
  if (back != exact) { std::printf("FAIL: limbs != exact integer sum (diff %g units of 2^-56)\n", (double)(back - exact)); return 1; }
  double s, ss;
  stat_load(r0, s, ss);
  const double want = (double)exact * 0x1p-56;
  if (std::fabs(s - want) > std::fabs(want) * 0x1p-50 + 0x1p-56) { std::printf("FAIL: stat_load %.17g vs %.17g\n", s, want); return 1; }
  // 3. poison
  int64_t p[kStatWords] = {0};
  stat_add_f32(p, 1.0f, INFINITY);
  stat_load(p, s, ss);
  if (!(s != s) || !(ss != ss)) { std::printf("FAIL: poison not reported\n"); return 1; }
  std::printf("ok: %zu addends, %zu in the order test\n", vals.size(), fin.size());
  return 0;
}
