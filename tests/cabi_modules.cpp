// Second C++ consumer of the C ABI (include/gqhip.h), no Python / torch in the process: the module-level surface of ABI 7 / 8 --
//   gq_quantize_z_f32 (BCHW strided K = 2, BLC contiguous K = 2), gq_argmax_f32 at dim 4 WITH a codebook cache (then the codebook is
//   edited in place and the call repeated), vq_argmin_f32, vq_quantize_z_f32 (straight-through value + loss), lfq_pack_f32,
//   gq_quantize_z_gauss_f32 (statistics, lambda state) --
// each checked against the CPU oracle's C entry points (oracle/gq_oracle.c) on operands derived here the way gqhip.h says the
// kernels derive them.  Built and run by tests/test_gpu_cabi.py.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "gqhip.h"

extern "C" void gq_oracle_nlp(const float *cb, float *nlp, int64_t n, int64_t dim);
extern "C" void gq_oracle_argmax(const float *mu, const float *sd, const float *lsd, const float *cb, const float *nlp,
                                 int64_t *idx, float *zhat, float *best, float *second, int64_t dim, int64_t rows,
                                 int64_t n, float beta, int nthreads);
extern "C" void vq_oracle_argmin(const float *z, const float *emb, int64_t *idx, double *best_out, double *second_out,
                                 int64_t dim, int64_t rows, int64_t n, int nthreads);
extern "C" void lfq_oracle_pack(const float *x, int64_t *idx, int64_t rows, int64_t nbits);

#define CHECK_HIP(x)                                                              \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) { std::printf("HIP error %d at %s\n", (int)e_, #x); return 2; } \
  } while (0)
#define CHECK_GQ(x)                                                                        \
  do {                                                                                     \
    int rc_ = (x);                                                                         \
    if (rc_ != GQHIP_OK) { std::printf("%s -> %s\n", #x, gqhip_status_string(rc_)); return 3; } \
  } while (0)

static float gauss(uint64_t &s) {
  auto u = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return ((s >> 11) + 1) * (1.0 / 9007199254740993.0); };
  return (float)(std::sqrt(-2.0 * std::log(u())) * std::cos(6.283185307179586 * u()));
}

template <typename T>
struct Dev {
  T *p = nullptr;
  size_t n = 0;
  explicit Dev(size_t count) : n(count) { if (hipMalloc(&p, count * sizeof(T) + 256) != hipSuccess) p = nullptr; }
  ~Dev() { if (p) (void)hipFree(p); }
  int up(const std::vector<T> &h, hipStream_t st) { return hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, st) == hipSuccess ? 0 : 1; }
  int down(std::vector<T> &h, hipStream_t st) { h.resize(n); return hipMemcpyAsync(h.data(), p, n * sizeof(T), hipMemcpyDeviceToHost, st) == hipSuccess ? 0 : 1; }
};

// rows (mu, sd, lsd) of z [B, 2c, L] (bchw) / [B, L, 2c] (blc) as gqhip.h defines them: row = (b L + l) K + k, column g <- channel
// g K + k (strided) | k dim + g (contiguous); sd = float(exp(double(0.5 clamp(lv)))), lsd = float(log(double(sd)))
static void rows_of_z(const std::vector<float> &z, int64_t B, int64_t L, int64_t c, int64_t dim, int bchw, int contiguous, float lv_min,
                      float lv_max, std::vector<float> &mu, std::vector<float> &sd, std::vector<float> &lsd) {
  const int64_t K = c / dim;
  mu.resize(B * L * c); sd.resize(mu.size()); lsd.resize(mu.size());
  for (int64_t b = 0; b < B; ++b)
    for (int64_t l = 0; l < L; ++l)
      for (int64_t k = 0; k < K; ++k)
        for (int64_t g = 0; g < dim; ++g) {
          const int64_t ch = contiguous ? k * dim + g : g * K + k;
          const int64_t zm = bchw ? (b * 2 * c + ch) * L + l : (b * L + l) * 2 * c + ch;
          const int64_t zl = bchw ? (b * 2 * c + c + ch) * L + l : zm + c;
          float lv = z[zl];
          lv = lv < lv_min ? lv_min : lv;
          lv = lv > lv_max ? lv_max : lv;
          const float s = (float)std::exp((double)(0.5f * lv));
          const int64_t o = (((b * L + l) * K + k) * dim) + g;
          mu[o] = z[zm]; sd[o] = s; lsd[o] = (float)std::log((double)s);
        }
}

int main() {
  if (gqhip_abi_version() != GQHIP_ABI_VERSION || GQHIP_ABI_VERSION < 8) { std::printf("ABI version mismatch\n"); return 1; }
  hipStream_t st;
  CHECK_HIP(hipStreamCreate(&st));
  uint64_t seed = 7;
  long long bad_total = 0;

  // ------------------------------------------------------------------ 1. gq_quantize_z_f32: BCHW strided, BLC contiguous (dim 8, K = 2)
  {
    const int64_t B = 3, L = 70, c = 16, dim = 8, K = 2, n = 2100, rows = B * L * K;      // ragged everywhere
    std::vector<float> cb(n * dim), nlp(n * dim);
    for (auto &v : cb) v = gauss(seed);
    gq_oracle_nlp(cb.data(), nlp.data(), n, dim);
    Dev<float> d_cb(cb.size());
    if (d_cb.up(cb, st)) return 2;
    const int64_t wsb = gqhip_workspace_bytes(rows, n, dim), cab = gqhip_cb_cache_bytes(n, dim);
    Dev<char> d_ws((size_t)wsb), d_cache((size_t)(cab > 0 ? cab : 1));
    CHECK_HIP(hipMemsetAsync(d_cache.p, 0, (size_t)(cab > 0 ? cab : 1), st));
    for (int variant = 0; variant < 2; ++variant) {
      const int bchw = variant == 0, contiguous = variant == 1;
      std::vector<float> z(B * 2 * c * L);
      for (size_t i = 0; i < z.size(); ++i) z[i] = gauss(seed);
      // logvar half: trained-like, plus values beyond the clamp
      for (int64_t b = 0; b < B; ++b)
        for (int64_t l = 0; l < L; ++l)
          for (int64_t ch = 0; ch < c; ++ch) {
            const int64_t zl = bchw ? (b * 2 * c + c + ch) * L + l : (b * L + l) * 2 * c + c + ch;
            z[zl] = -1.5f + 0.3f * z[zl];
            if ((b + l + ch) % 97 == 0) z[zl] = 25.0f;
            if ((b + l + ch) % 89 == 0) z[zl] = -40.0f;
          }
      std::vector<float> mu, sd, lsd;
      rows_of_z(z, B, L, c, dim, bchw, contiguous, -30.0f, 20.0f, mu, sd, lsd);
      std::vector<int64_t> ref(rows);
      gq_oracle_argmax(mu.data(), sd.data(), lsd.data(), cb.data(), nlp.data(), ref.data(), nullptr, nullptr, nullptr, dim, rows, n, 1.0f, 0);
      Dev<float> d_z(z.size()), d_zhat(B * c * L), d_mu(rows * dim), d_sd(rows * dim);
      Dev<int64_t> d_idx(rows);
      if (d_z.up(z, st)) return 2;
      CHECK_GQ(gq_quantize_z_f32(d_z.p, nullptr, d_cb.p, d_idx.p, d_zhat.p, nullptr, d_mu.p, d_sd.p, B, L, c, dim, n,
                                 bchw ? GQHIP_LAYOUT_BCHW : GQHIP_LAYOUT_BLC, contiguous ? GQHIP_GROUP_CONTIGUOUS : GQHIP_GROUP_STRIDED,
                                 -30.0, 20.0, 1.0, d_ws.p, wsb, cab > 0 ? d_cache.p : nullptr, cab > 0 ? cab : 0, st));
      std::vector<int64_t> idx;
      std::vector<float> zhat, mu_k, sd_k;
      if (d_idx.down(idx, st) || d_zhat.down(zhat, st) || d_mu.down(mu_k, st) || d_sd.down(sd_k, st)) return 2;
      CHECK_HIP(hipStreamSynchronize(st));
      long long bad = 0, bad_ops = 0;
      for (int64_t r = 0; r < rows * dim; ++r) bad_ops += (mu_k[r] != mu[r]) + (sd_k[r] != sd[r]);
      for (int64_t b = 0; b < B; ++b)
        for (int64_t l = 0; l < L; ++l)
          for (int64_t k = 0; k < K; ++k) {
            const int64_t row = (b * L + l) * K + k;
            const int64_t io = bchw ? (b * K + k) * L + l : row;
            bad += idx[io] != ref[row];
            for (int64_t g = 0; g < dim; ++g) {
              const int64_t ch = contiguous ? k * dim + g : g * K + k;
              const int64_t zo = bchw ? (b * c + ch) * L + l : (b * L + l) * c + ch;
              bad += zhat[zo] != cb[ref[row] * dim + g];
            }
          }
      std::printf("gq_quantize_z_f32 %s: %lld rows, mismatches %lld, operand mismatches %lld\n",
                  bchw ? "bchw/strided" : "blc/contiguous", (long long)rows, bad, bad_ops);
      bad_total += bad + bad_ops;
    }
  }

  // ------------------------------------------------------------------ 2. dim 4 with a codebook cache; the codebook edited in place
  {
    const int64_t dim = 4, rows = 3000, n = 16384;
    if (!gqhip_grid_search_applies(n, dim)) { std::printf("dim-4 search does not apply?\n"); return 6; }
    std::vector<float> cb(n * dim), nlp(n * dim), mu(rows * dim), sd(rows * dim), lsd(rows * dim);
    for (auto &v : cb) v = gauss(seed);
    for (auto &v : mu) v = 0.9f * gauss(seed);
    for (size_t i = 0; i < sd.size(); ++i) { sd[i] = std::exp(0.5f * (-1.5f + 0.3f * gauss(seed))); lsd[i] = (float)std::log((double)sd[i]); }
    const int64_t wsb = gqhip_workspace_bytes(rows, n, dim), cab = gqhip_cb_cache_bytes(n, dim);
    if (cab <= 0) return 6;
    Dev<float> d_cb(cb.size()), d_mu(mu.size()), d_sd(sd.size()), d_zhat(rows * dim);
    Dev<int64_t> d_idx(rows);
    Dev<char> d_ws((size_t)wsb), d_cache((size_t)cab);
    CHECK_HIP(hipMemsetAsync(d_cache.p, 0xA5, (size_t)cab, st));             // garbage: the library must notice and build
    if (d_cb.up(cb, st) || d_mu.up(mu, st) || d_sd.up(sd, st)) return 2;
    for (int pass = 0; pass < 3; ++pass) {
      if (pass == 2) {                                                       // edit a slice of the codebook in place: same pointer, same cache
        for (int64_t j = 100; j < 1100; ++j) for (int64_t i = 0; i < dim; ++i) cb[j * dim + i] = 1.3f * gauss(seed);
        if (d_cb.up(cb, st)) return 2;
      }
      CHECK_GQ(gq_argmax_f32(d_mu.p, d_sd.p, nullptr, d_cb.p, d_idx.p, d_zhat.p, dim, rows, n, 1.0, d_ws.p, wsb, d_cache.p, cab, st));
      std::vector<int64_t> idx, ref(rows);
      std::vector<float> zhat;
      if (d_idx.down(idx, st) || d_zhat.down(zhat, st)) return 2;
      CHECK_HIP(hipStreamSynchronize(st));
      gq_oracle_nlp(cb.data(), nlp.data(), n, dim);
      gq_oracle_argmax(mu.data(), sd.data(), lsd.data(), cb.data(), nlp.data(), ref.data(), nullptr, nullptr, nullptr, dim, rows, n, 1.0f, 0);
      long long bad = 0;
      for (int64_t r = 0; r < rows; ++r) {
        bad += idx[r] != ref[r];
        for (int64_t i = 0; i < dim; ++i) bad += zhat[r * dim + i] != cb[ref[r] * dim + i];
      }
      int64_t g4[4];
      CHECK_GQ(gqhip_debug_grid(d_ws.p, d_cache.p, g4));
      std::printf("gq_argmax_f32 dim 4 + cache, pass %d%s: mismatches %lld, index current %lld\n", pass, pass == 2 ? " (codebook edited)" : "", bad, (long long)g4[3]);
      bad_total += bad + (g4[3] != 1);
    }
  }

  // ------------------------------------------------------------------ 3. vq_argmin_f32, vq_quantize_z_f32 (K = 2, bchw), lfq_pack_f32
  {
    const int64_t B = 2, L = 150, dim = 8, K = 2, c = 16, n = 3000, rows = B * L * K;
    std::vector<float> emb(n * dim), z(B * c * L);
    for (auto &v : emb) v = gauss(seed);
    for (auto &v : z) v = gauss(seed);
    std::vector<float> zr(rows * dim);                                       // row (b, l, k), column d <- channel d K + k (vq.py:53)
    for (int64_t b = 0; b < B; ++b) for (int64_t l = 0; l < L; ++l) for (int64_t k = 0; k < K; ++k) for (int64_t d = 0; d < dim; ++d)
      zr[((b * L + l) * K + k) * dim + d] = z[(b * c + d * K + k) * L + l];
    std::vector<int64_t> ref(rows);
    vq_oracle_argmin(zr.data(), emb.data(), ref.data(), nullptr, nullptr, dim, rows, n, 0);
    const int64_t wsb = gqhip_workspace_bytes(rows, n, dim), cab = gqhip_cb_cache_bytes(n, dim);
    Dev<float> d_emb(emb.size()), d_z(z.size()), d_zr(zr.size()), d_zq(z.size()), d_zq2(zr.size()), d_loss(2);
    Dev<int64_t> d_idx(rows), d_idx2(rows);
    Dev<char> d_ws((size_t)wsb), d_cache((size_t)(cab > 0 ? cab : 1));
    CHECK_HIP(hipMemsetAsync(d_cache.p, 0, (size_t)(cab > 0 ? cab : 1), st));
    if (d_emb.up(emb, st) || d_z.up(z, st) || d_zr.up(zr, st)) return 2;
    CHECK_GQ(vq_argmin_f32(d_zr.p, d_emb.p, d_idx2.p, d_zq2.p, dim, rows, n, d_ws.p, wsb, cab > 0 ? d_cache.p : nullptr, cab > 0 ? cab : 0, st));
    CHECK_GQ(vq_quantize_z_f32(d_z.p, d_emb.p, d_idx.p, d_zq.p, d_loss.p, B, L, c, dim, n, GQHIP_LAYOUT_BCHW, 0.25, 1, d_ws.p, wsb,
                               cab > 0 ? d_cache.p : nullptr, cab > 0 ? cab : 0, st));
    std::vector<int64_t> idx, idx2;
    std::vector<float> zq, zq2, loss;
    if (d_idx.down(idx, st) || d_idx2.down(idx2, st) || d_zq.down(zq, st) || d_zq2.down(zq2, st) || d_loss.down(loss, st)) return 2;
    CHECK_HIP(hipStreamSynchronize(st));
    long long bad = 0;
    double acc = 0.0;
    for (int64_t b = 0; b < B; ++b) for (int64_t l = 0; l < L; ++l) for (int64_t k = 0; k < K; ++k) {
      const int64_t row = (b * L + l) * K + k;
      bad += idx2[row] != ref[row];
      bad += idx[(b * K + k) * L + l] != ref[row];
      for (int64_t d = 0; d < dim; ++d) {
        const float e = emb[ref[row] * dim + d], zz = zr[row * dim + d];
        volatile float df = e - zz;                                          // z + (e - z), each op rounded (vq.py:89)
        const float want = zz + df;
        bad += zq[(b * c + d * K + k) * L + l] != want;
        bad += zq2[row * dim + d] != e;
        acc += (double)(df * df);
      }
    }
    const float m = (float)(acc / (double)(rows * dim));
    const float want_loss = m + 0.25f * m;
    const bool loss_ok = std::fabs(loss[0] - want_loss) <= 2e-6f * want_loss && std::fabs(loss[1] - m) <= 2e-6f * m;
    std::printf("vq_argmin_f32 / vq_quantize_z_f32: %lld rows, mismatches %lld, loss %.7f (want %.7f)\n", (long long)rows, bad, loss[0], want_loss);
    bad_total += bad + !loss_ok;

    const int64_t lrows = 777, nbits = 16;
    std::vector<float> x(lrows * nbits);
    for (auto &v : x) v = gauss(seed);
    x[5] = 0.0f; x[6] = -0.0f;
    std::vector<int64_t> lref(lrows), lidx;
    lfq_oracle_pack(x.data(), lref.data(), lrows, nbits);
    Dev<float> d_x(x.size()), d_q(x.size());
    Dev<int64_t> d_li(lrows);
    if (d_x.up(x, st)) return 2;
    CHECK_GQ(lfq_pack_f32(d_x.p, d_li.p, d_q.p, lrows, nbits, st));
    std::vector<float> q;
    if (d_li.down(lidx, st) || d_q.down(q, st)) return 2;
    CHECK_HIP(hipStreamSynchronize(st));
    long long lbad = 0;
    for (int64_t r = 0; r < lrows; ++r) {
      lbad += lidx[r] != lref[r];
      for (int64_t i = 0; i < nbits; ++i) lbad += q[r * nbits + i] != (x[r * nbits + i] > 0.0f ? 1.0f : -1.0f);
    }
    std::printf("lfq_pack_f32: %lld rows, mismatches %lld\n", (long long)lrows, lbad);
    bad_total += lbad;
  }

  // ------------------------------------------------------------------ 4. gq_quantize_z_gauss_f32 (GQ2 eval forward), two calls: the lambda state moves
  {
    const int64_t B = 2, L = 256, c = 16, dim = 16, n = 4096, rows = B * L;
    std::vector<float> cb(n * dim), nlp(n * dim), z(B * L * 2 * c), noise(B * L * c);
    for (auto &v : cb) v = gauss(seed);
    for (auto &v : noise) v = gauss(seed);
    for (int64_t p = 0; p < B * L; ++p) for (int64_t ch = 0; ch < c; ++ch) {
      z[p * 2 * c + ch] = 1.6f * gauss(seed);
      z[p * 2 * c + c + ch] = -1.5f + 0.3f * gauss(seed);
    }
    gq_oracle_nlp(cb.data(), nlp.data(), n, dim);
    std::vector<float> mu, sd, lsd;
    rows_of_z(z, B, L, c, dim, 0, 1, -30.0f, 20.0f, mu, sd, lsd);
    std::vector<int64_t> ref(rows);
    gq_oracle_argmax(mu.data(), sd.data(), lsd.data(), cb.data(), nlp.data(), ref.data(), nullptr, nullptr, nullptr, dim, rows, n, 1.0f, 0);
    // host statistics: the element in fp32 (gaussian.py:225), sums in fp64
    const double log2n = 12.0, tol = 0.5, f = 1.01;
    std::vector<float> kl2(rows);
    for (int64_t r = 0; r < rows; ++r) {
      double a = 0.0;
      for (int64_t g = 0; g < dim; ++g) {
        const float m = z[r * 2 * c + g];
        float lv = z[r * 2 * c + c + g];
        const float var = (float)std::exp((double)lv);
        volatile float t = m * m; t = t + var; t = t - 1.0f; t = t - lv;
        volatile float kt = (float)0.7213 * t;
        a += (double)kt;
      }
      kl2[r] = (float)a;
    }
    const int64_t wsb = gqhip_workspace_bytes(rows, n, dim), cab = gqhip_cb_cache_bytes(n, dim);
    Dev<float> d_cb(cb.size()), d_z(z.size()), d_noise(noise.size()), d_zhat(B * L * c), d_zq(B * L * c), d_noq(B * L * c), d_sd(B * L * c);
    Dev<int64_t> d_idx(rows);
    Dev<char> d_ws((size_t)wsb), d_cache((size_t)(cab > 0 ? cab : 1)), d_sc(64);
    Dev<double> d_lam(3);
    CHECK_HIP(hipMemsetAsync(d_cache.p, 0, (size_t)(cab > 0 ? cab : 1), st));
    std::vector<double> lam = {1.0, 1.0, 1.0};
    if (d_cb.up(cb, st) || d_z.up(z, st) || d_noise.up(noise, st) || d_lam.up(lam, st)) return 2;
    long long bad = 0;
    for (int call = 0; call < 2; ++call) {
      CHECK_GQ(gq_quantize_z_gauss_f32(d_z.p, d_noise.p, d_cb.p, d_idx.p, d_zhat.p, d_zq.p, d_noq.p, d_sd.p, d_sc.p, d_lam.p, B, L, c, dim, n,
                                       GQHIP_LAYOUT_BLC, GQHIP_GROUP_CONTIGUOUS, -30.0, 20.0, 1.0, 1, log2n, tol, f, 1e-7, 1e7, 0,
                                       d_ws.p, wsb, cab > 0 ? d_cache.p : nullptr, cab > 0 ? cab : 0, st));
      std::vector<int64_t> idx;
      std::vector<float> zhat, zq, noq, sdo;
      std::vector<char> sc;
      std::vector<double> lam_dev;
      if (d_idx.down(idx, st) || d_zhat.down(zhat, st) || d_zq.down(zq, st) || d_noq.down(noq, st) || d_sd.down(sdo, st) || d_sc.down(sc, st) ||
          d_lam.down(lam_dev, st)) return 2;
      CHECK_HIP(hipStreamSynchronize(st));
      float fs[4];
      double ds[3];
      std::memcpy(fs, sc.data(), 16);
      std::memcpy(ds, sc.data() + 32, 24);
      // expected scalars with the lambdas as they were before this call
      double s = 0.0, ws = 0.0;
      float mn = INFINITY, mx = -INFINITY;
      const float hi = (float)(log2n + tol), lo = (float)(log2n - tol);
      for (int64_t r = 0; r < rows; ++r) {
        const float k = kl2[r];
        s += k; mn = std::fmin(mn, k); mx = std::fmax(mx, k);
        const float w = k > hi ? (float)lam[2] : (k < lo ? (float)lam[1] : 1.0f);
        volatile float e = w * k;
        ws += (double)e;
      }
      const float mean = (float)(s / rows), kl_loss = (float)(ws / rows) * (float)lam[0];
      lam[0] = mean > (float)log2n ? lam[0] * f : lam[0] / f;
      if (mx > hi) lam[2] = lam[2] * f;
      lam[2] = std::fmax(std::fmin(lam[2], 1e7), 1.0);
      lam[1] = mn < lo ? lam[1] / f : lam[1] * f;
      lam[1] = std::fmax(std::fmin(lam[1], 1.0), 1e-7);
      auto close = [](float a, float b) { return std::fabs(a - b) <= 2e-6f * std::fmax(1.0f, std::fabs(b)); };
      bad += !close(fs[0], kl_loss) + !close(fs[1], mean) + (fs[2] != mn) + (fs[3] != mx);
      for (int i = 0; i < 3; ++i) bad += (ds[i] != lam[i]) + (lam_dev[i] != lam[i]);
      for (int64_t r = 0; r < rows; ++r) {
        bad += idx[r] != ref[r];
        for (int64_t g = 0; g < dim; ++g) {
          const int64_t o = r * c + g;
          volatile float e = noise[o] * sd[r * dim + g];
          const float nq = mu[r * dim + g] + e;
          bad += (zq[o] != cb[ref[r] * dim + g]) + (zhat[o] != cb[ref[r] * dim + g]) + (noq[o] != nq) + (sdo[o] != sd[r * dim + g]);
        }
      }
      std::printf("gq_quantize_z_gauss_f32 call %d: kl_loss %.6f mean %.5f min %.5f max %.5f, lambdas %.6f %.6f %.6f, mismatches so far %lld\n",
                  call, fs[0], fs[1], fs[2], fs[3], ds[0], ds[1], ds[2], bad);
    }
    bad_total += bad;
  }
  std::printf("cabi modules: total mismatches %lld\n", bad_total);
  return bad_total ? 10 : 0;
}
