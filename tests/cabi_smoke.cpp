// C++ consumer of the C ABI (include/gqhip.h) with no Python / torch in the process: device buffers from
// hipMalloc, a user stream, the fused arg-max, the compat score op and the dequant gather -- checked against
// the CPU oracle's C entry point (oracle/gq_oracle.c).  Built and run by tests/test_gpu_cabi.py:
//   hipcc -O2 -I include tests/cabi_smoke.cpp -L<csrc> -lgqhip -L oracle -lgq_oracle -o cabi_smoke
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gqhip.h"

extern "C" void gq_oracle_nlp(const float *cb, float *nlp, int64_t n, int64_t dim);
extern "C" void gq_oracle_argmax(const float *mu, const float *sd, const float *lsd, const float *cb, const float *nlp,
                                 int64_t *idx, float *zhat, float *best, float *second, int64_t dim, int64_t rows,
                                 int64_t n, float beta, int nthreads);

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

static float gauss(uint64_t &s) {  // Box-Muller on a 64-bit LCG: deterministic inputs without <random>
  auto u = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return ((s >> 11) + 1) * (1.0 / 9007199254740993.0); };
  return (float)(std::sqrt(-2.0 * std::log(u())) * std::cos(6.283185307179586 * u()));
}

int main() {
  const int64_t dim = 16, rows = 1000, n = 4100;  // ragged: rows % 128 != 0, n % 32 != 0
  const float beta = 1.0f;
  uint64_t seed = 42;
  std::vector<float> mu(rows * dim), sd(rows * dim), lsd(rows * dim), cb(n * dim), nlp(n * dim);
  for (auto &v : cb) v = gauss(seed);
  for (auto &v : mu) v = 0.9f * gauss(seed);
  for (size_t i = 0; i < sd.size(); ++i) {
    sd[i] = std::exp(0.5f * (-1.5f + 0.3f * gauss(seed)));
    lsd[i] = (float)std::log((double)sd[i]);  // what the kernels use when logsd is NULL
  }
  if (gqhip_abi_version() != GQHIP_ABI_VERSION) { std::printf("ABI version mismatch\n"); return 1; }

  float *d_mu, *d_sd, *d_cb, *d_zhat, *d_out, *d_deq;
  int64_t *d_idx;
  void *d_ws;
  hipStream_t st;
  CHECK_HIP(hipStreamCreate(&st));
  const int64_t ws_bytes = gqhip_workspace_bytes(rows, n, dim);
  CHECK_HIP(hipMalloc(&d_mu, mu.size() * 4));
  CHECK_HIP(hipMalloc(&d_sd, sd.size() * 4));
  CHECK_HIP(hipMalloc(&d_cb, cb.size() * 4));
  CHECK_HIP(hipMalloc(&d_zhat, rows * dim * 4));
  CHECK_HIP(hipMalloc(&d_deq, rows * dim * 4));
  CHECK_HIP(hipMalloc(&d_out, (size_t)64 * n * 4));
  CHECK_HIP(hipMalloc(&d_idx, rows * 8));
  CHECK_HIP(hipMalloc(&d_ws, ws_bytes));
  CHECK_HIP(hipMemcpyAsync(d_mu, mu.data(), mu.size() * 4, hipMemcpyHostToDevice, st));
  CHECK_HIP(hipMemcpyAsync(d_sd, sd.data(), sd.size() * 4, hipMemcpyHostToDevice, st));
  CHECK_HIP(hipMemcpyAsync(d_cb, cb.data(), cb.size() * 4, hipMemcpyHostToDevice, st));

  // error paths first: status codes, no crash
  if (gq_argmax_f32(d_mu, d_sd, nullptr, d_cb, d_idx, d_zhat, dim, rows, n, beta, d_ws, 16, nullptr, 0, st) != GQHIP_ERR_WORKSPACE) return 4;
  if (gq_argmax_f32(nullptr, d_sd, nullptr, d_cb, d_idx, d_zhat, dim, rows, n, beta, d_ws, ws_bytes, nullptr, 0, st) != GQHIP_ERR_INVALID_ARG) return 5;

  CHECK_GQ(gq_argmax_f32(d_mu, d_sd, nullptr, d_cb, d_idx, d_zhat, dim, rows, n, beta,
                         d_ws, ws_bytes, nullptr, 0, st));
  CHECK_GQ(gq_dequant_f32(d_idx, d_cb, d_deq, /*B*/ 1, /*L*/ rows, /*K*/ 1, dim, n, GQHIP_LAYOUT_BLC,
                          GQHIP_GROUP_STRIDED, st));
  CHECK_GQ(gq_scores_f32(d_mu, d_sd, d_cb, d_out, dim, 64, n, (double)beta, st));
  std::vector<int64_t> idx(rows);
  std::vector<float> zhat(rows * dim), deq(rows * dim), out((size_t)64 * n);
  CHECK_HIP(hipMemcpyAsync(idx.data(), d_idx, rows * 8, hipMemcpyDeviceToHost, st));
  CHECK_HIP(hipMemcpyAsync(zhat.data(), d_zhat, zhat.size() * 4, hipMemcpyDeviceToHost, st));
  CHECK_HIP(hipMemcpyAsync(deq.data(), d_deq, deq.size() * 4, hipMemcpyDeviceToHost, st));
  CHECK_HIP(hipMemcpyAsync(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost, st));
  CHECK_HIP(hipStreamSynchronize(st));

  gq_oracle_nlp(cb.data(), nlp.data(), n, dim);
  std::vector<int64_t> ref(rows);
  std::vector<float> best(rows), second(rows);
  gq_oracle_argmax(mu.data(), sd.data(), lsd.data(), cb.data(), nlp.data(), ref.data(), nullptr, best.data(), second.data(),
                   dim, rows, n, beta, 0);
  int64_t bad = 0, bad_z = 0, bad_am = 0;
  for (int64_t r = 0; r < rows; ++r) {
    bad += idx[r] != ref[r];
    for (int64_t i = 0; i < dim; ++i) {
      bad_z += zhat[r * dim + i] != cb[ref[r] * dim + i];
      bad_z += deq[r * dim + i] != zhat[r * dim + i];
    }
  }
  for (int64_t r = 0; r < 64; ++r) {  // compat op: same arg-max wherever the gap is not a rounding tie
    int64_t am = 0;
    for (int64_t j = 1; j < n; ++j) if (out[r * n + j] > out[r * n + am]) am = j;
    if (best[r] - second[r] > 1e-3f && am != ref[r]) ++bad_am;
  }
  // the other filter kernel (fp32 MFMA instead of the default split-bf16): new workspace size, same indices
  int64_t bad_f = 0;
  if (gqhip_get_filter() != GQHIP_FILTER_AUTO) return 6;
  CHECK_GQ(gqhip_set_filter(GQHIP_FILTER_FP32));
  const int64_t ws2_bytes = gqhip_workspace_bytes(rows, n, dim);
  void *d_ws2;
  int64_t *d_idx2;
  CHECK_HIP(hipMalloc(&d_ws2, ws2_bytes));
  CHECK_HIP(hipMalloc(&d_idx2, rows * 8));
  CHECK_GQ(gq_argmax_f32(d_mu, d_sd, nullptr, d_cb, d_idx2, nullptr, dim, rows, n, beta, d_ws2, ws2_bytes, nullptr, 0, st));
  CHECK_GQ(gqhip_set_filter(GQHIP_FILTER_AUTO));
  if (gqhip_set_filter(7) != GQHIP_ERR_INVALID_ARG) return 7;
  std::vector<int64_t> idx2(rows);
  CHECK_HIP(hipMemcpyAsync(idx2.data(), d_idx2, rows * 8, hipMemcpyDeviceToHost, st));
  CHECK_HIP(hipStreamSynchronize(st));
  for (int64_t r = 0; r < rows; ++r) bad_f += idx2[r] != ref[r];
  std::printf("cabi smoke: %lld rows, index mismatches %lld (split-bf16 filter) / %lld (fp32 filter), zhat/dequant mismatches "
              "%lld, compat arg-max mismatches %lld, workspace %lld / %lld bytes\n",
              (long long)rows, (long long)bad, (long long)bad_f, (long long)bad_z, (long long)bad_am,
              (long long)ws_bytes, (long long)ws2_bytes);
  return (bad || bad_f || bad_z || bad_am) ? 10 : 0;
}
