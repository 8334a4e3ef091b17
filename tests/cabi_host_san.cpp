// Host half of libgqhip.so under AddressSanitizer + UndefinedBehaviorSanitizer, on the CPU (no GPU in the process: `make -C
// vq-vae-from-gaussian-vae_amd/csrc san` builds libgqhip_san.so with host-side sanitizers and device code untouched; sanitizers
// never run on the GPU).  Drives everything the entry points do BEFORE their first launch: the launch plan / workspace layout
// arithmetic over a sweep of shapes (the place where an overflow or a division by zero would sit), the filter selection, and the
// argument validation of every entry point (NULL pointers, negative / zero / huge sizes must come back as a status, not a crash).
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "gqhip.h"

int main() {
  if (gqhip_abi_version() != GQHIP_ABI_VERSION) { std::printf("FAIL abi\n"); return 1; }
  long plans = 0;
  const int64_t rows_v[] = {1, 2, 31, 32, 33, 127, 128, 1000, 1024, 4096, 8191, 8192, 16384, 65536, 262144, 1048576, 0x3fffffff};
  const int64_t n_v[] = {1, 2, 31, 32, 33, 64, 1000, 4096, 65536, 65537, 1 << 20, (1 << 21) + 1, 1 << 22, (1 << 22) + 33, 1 << 23, (1 << 26) - 7,
                         (1 << 27) + 5, 1 << 28, 0x3fffffff};
  const int64_t dim_v[] = {1, 3, 4, 7, 8, 16, 24, 32, 33, 64};
  for (int kind = 0; kind <= 3; ++kind) {
    if (gqhip_set_filter(kind) != GQHIP_OK || gqhip_get_filter() != kind) { std::printf("FAIL set_filter %d\n", kind); return 1; }
    for (int64_t rows : rows_v)
      for (int64_t n : n_v)
        for (int64_t dim : dim_v) {
          const int64_t b = gqhip_workspace_bytes(rows, n, dim);
          if (b <= 0) { std::printf("FAIL workspace_bytes(%lld, %lld, %lld) = %lld\n", (long long)rows, (long long)n, (long long)dim, (long long)b); return 1; }
          int64_t out8[8];
          std::memset(out8, 0, sizeof(out8));
          if (gqhip_debug_plan(rows, n, dim, out8) != GQHIP_OK) { std::printf("FAIL debug_plan\n"); return 1; }
          if (out8[1] < 0 || out8[1] > 64 || out8[0] < 0 || out8[0] > b) { std::printf("FAIL plan fields\n"); return 1; }
          if (out8[1] > 0) {
            // the records hold 16-bit half-group ids relative to their split (csrc/gq_common.h:Rec), and the splits cover the codebook
            const int64_t gt = out8[2], tps = out8[3], tiles = (n + 31) / 32;
            const int64_t halves = (out8[4] == 2 || out8[4] == 3) ? 2 : 1;
            if (gt < 1 || tps % gt != 0 || 2 * (tps / gt) > 65536 || (out8[1] / halves) * tps < tiles) {
              std::printf("FAIL id range / coverage: rows %lld n %lld dim %lld: sets %lld gt %lld tps %lld\n", (long long)rows, (long long)n,
                          (long long)dim, (long long)out8[1], (long long)gt, (long long)tps);
              return 1;
            }
          }
          ++plans;
        }
  }
  gqhip_set_filter(GQHIP_FILTER_AUTO);
  if (gqhip_set_filter(7) == GQHIP_OK || gqhip_workspace_bytes(-1, 16, 16) != -1 || gqhip_workspace_bytes(16, 0, 16) != -1 ||
      gqhip_workspace_bytes(16, 16, 65) != -1 || gqhip_debug_plan(0, 16, 16, nullptr) == GQHIP_OK) {
    std::printf("FAIL invalid sizes accepted\n");
    return 1;
  }
  // every entry point with NULL pointers / bad sizes: a status, never a dereference (rows == 0 returns OK before any check of the pointers)
  int bad = 0;
  bad += gq_argmax_f32(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 16, 4, 1024, 1.0, nullptr, 0, nullptr, 0, nullptr) == GQHIP_OK;
  bad += gq_argmax_f32(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 16, -1, 1024, 1.0, nullptr, 0, nullptr, 0, nullptr) == GQHIP_OK;
  bad += gq_argmax_f32(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 16, 0, 1024, 1.0, nullptr, 0, nullptr, 0, nullptr) != GQHIP_OK;
  bad += gq_scores_f32(nullptr, nullptr, nullptr, nullptr, 16, 4, 1024, 1.0, nullptr) == GQHIP_OK;
  bad += gq_quantize_z_f32(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1, 16, 16, 16, 1024, 0, 0, -30.0, 20.0,
                           1.0, nullptr, 0, nullptr, 0, nullptr) == GQHIP_OK;
  bad += gq_dequant_f32(nullptr, nullptr, nullptr, 1, 16, 1, 16, 1024, 0, 0, nullptr) == GQHIP_OK;
  bad += vq_argmin_f32(nullptr, nullptr, nullptr, nullptr, 16, 4, 1024, nullptr, 0, nullptr, 0, nullptr) == GQHIP_OK;
  bad += vq_quantize_z_f32(nullptr, nullptr, nullptr, nullptr, nullptr, 1, 16, 16, 16, 1024, 0, 0.25, 1, nullptr, 0, nullptr, 0, nullptr) == GQHIP_OK;
  {
    char buf[64];                 // non-NULL host addresses: validation must stop at the first NULL device pointer / bad size
    bad += vq_quantize_z_f32((const float *)buf, (const float *)buf, (int64_t *)buf, (float *)buf, nullptr, 1, 16, 16, 5, 1024, 0, 0.25, 1, nullptr, 0,
                             nullptr, 0, nullptr) == GQHIP_OK;                                      // c % dim != 0
    bad += vq_quantize_z_f32((const float *)buf, (const float *)buf, (int64_t *)buf, (float *)buf, nullptr, 0, 16, 16, 16, 1024, 0, 0.25, 1, nullptr,
                             0, nullptr, 0, nullptr) != GQHIP_OK;                                  // empty batch: OK, nothing launched
    bad += gq_quantize_z_gauss_f32(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1, 16, 16, 16, 1024, 0, 1,
                                   -30.0, 20.0, 1.0, 1, 10.0, 0.5, 1.01, 1e-7, 1e7, 0, nullptr, 0, nullptr, 0, nullptr) == GQHIP_OK;
    bad += gq_quantize_z_gauss_f32((const float *)buf, (const float *)buf, (const float *)buf, (int64_t *)buf, (float *)buf, nullptr, (float *)buf,
                                   nullptr, buf + 4, (double *)buf, 1, 16, 16, 16, 1024, 0, 1, -30.0, 20.0, 1.0, 1, 10.0, 0.5, 1.01, 1e-7, 1e7, 0,
                                   nullptr, 0, nullptr, 0, nullptr) == GQHIP_OK;                  // scalars_out not 8-byte aligned
    bad += gq_quantize_z_gauss_f32((const float *)buf, (const float *)buf, (const float *)buf, (int64_t *)buf, (float *)buf, nullptr, (float *)buf,
                                   nullptr, buf, (double *)buf, 1, 16, 16, 16, 1024, 0, 1, -30.0, 20.0, 1.0, 1, 10.0, 0.5, 1.01, 1e-7, 1e7, 0,
                                   nullptr, 0, nullptr, 0, nullptr) != GQHIP_ERR_WORKSPACE;       // valid arguments, no workspace
  }
  bad += gqhip_cb_cache_degenerate(nullptr, 65536, 4) != -1;
  bad += lfq_pack_f32(nullptr, nullptr, nullptr, 4, 16, nullptr) == GQHIP_OK;
  bad += lfq_pack_f32(nullptr, nullptr, nullptr, 4, 63, nullptr) == GQHIP_OK;
  bad += lfq_unpack_f32(nullptr, nullptr, 4, 0, nullptr) == GQHIP_OK;
  bad += gn_silu_f32(nullptr, nullptr, nullptr, nullptr, nullptr, 1, 128, 64, 32, 1e-6, 1, GQHIP_LAYOUT_NHWC, nullptr, nullptr) == GQHIP_OK;
  bad += gn_silu_f32(nullptr, nullptr, nullptr, nullptr, nullptr, 1, 128, 64, 0, 1e-6, 1, GQHIP_LAYOUT_NHWC, nullptr, nullptr) == GQHIP_OK;
  bad += add_bias_f32(nullptr, nullptr, nullptr, nullptr, 1, 128, 64, GQHIP_LAYOUT_NHWC, nullptr) == GQHIP_OK;
  bad += conv3x3_f32(nullptr, nullptr, nullptr, nullptr, nullptr, 32, 1e-6, 1, nullptr, nullptr, nullptr, 1, 32, 32, 512, 32, nullptr) == GQHIP_OK;
  bad += conv3x3_f32(nullptr, nullptr, nullptr, nullptr, nullptr, 32, 1e-6, 1, nullptr, nullptr, nullptr, 1, 32, 32, 512, 3, nullptr) == GQHIP_OK;
  bad += gqhip_checksum_tensors(nullptr, 4, nullptr, nullptr) == GQHIP_OK;
  bad += gqhip_checksum_tensors(nullptr, -1, nullptr, nullptr) == GQHIP_OK;
  if (bad) { std::printf("FAIL %d argument checks\n", bad); return 1; }
  std::printf("ok: %ld launch plans, argument validation of the entry points, under ASan + UBSan (host side)\n", plans);
  return 0;
}
