// gqhip_unet.hip -- C-ABI entry points of libgqhip.so for the conv stack (include/gqhip.h): fused GroupNorm, residual adds,
// Winograd transforms and GEMMs, the direct fp16 x 3 convolutions, attention operand splits.
// gfx950 only; built by `make -C vq-vae-from-gaussian-vae_amd/csrc` (its own translation unit: the two halves compile in parallel).
#include "gqhip.h"

#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "gqhip_internal.h"
#include "gq_unet_aux.h"
#include "gq_wino_gemm.h"
#include "gq_conv3.h"
#include "gq_conv_f32.h"

using namespace gqhip;

namespace {
// Statistics records are zeroed by the entry point that fills them -- unless the caller has said that they already are
// (gqhip_stats_prezeroed: the conv stack's modules carve all records of a forward out of one arena zeroed by ONE fill, instead of
// ~60 separate 32-KB memset launches per step).  Thread-local: the flag belongs to the calling thread's sequence of calls.
thread_local int g_stats_prezeroed = 0;
inline hipError_t stats_zero(void *p, size_t bytes, hipStream_t st) {
  return g_stats_prezeroed ? hipSuccess : hipMemsetAsync(p, 0, bytes, st);
}
}  // namespace

extern "C" {

int gqhip_stats_prezeroed(int on) {
  g_stats_prezeroed = on != 0;
  return GQHIP_OK;
}

int gn_silu_f32(const float *x, const float *gamma, const float *beta, const float *pre_bias_or_null, float *y,
                int64_t B, int64_t C, int64_t HW, int64_t groups, double eps, int apply_silu, int layout,
                int64_t *stats_ws, void *stream) {
  if (B < 0 || C < 1 || HW < 1 || groups < 1 || C % groups != 0) return GQHIP_ERR_INVALID_ARG;
  if (B == 0) return GQHIP_OK;
  if (!x || !gamma || !beta || !y || !stats_ws) return GQHIP_ERR_INVALID_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t bg = B * groups, cpg = C / groups, chunk = cpg * HW;
  if (layout == GQHIP_LAYOUT_NHWC) {
    // thread <-> channel-quad mapping needs cpg % 4 == 0, (C/4) | 256, <= 64 groups
    if (cpg % 4 != 0 || 256 % (C / 4) != 0 || groups > 64) return GQHIP_ERR_INVALID_ARG;
    if (stats_zero(stats_ws, sizeof(int64_t) * kStatWords * bg, st) != hipSuccess) return check_launch();
    const int lanes = (int)(256 / (C / 4));
    int slabs = (int)((HW + (int64_t)lanes * 16 - 1) / ((int64_t)lanes * 16));   // ~16 pixels per thread
    if (slabs > 1024) slabs = 1024;
    if (slabs < 1) slabs = 1;
    hipLaunchKernelGGL(gn_stats_nhwc_kernel, dim3((unsigned)(B * slabs)), dim3(256), 0, st, x, pre_bias_or_null,
                       stats_ws, (int)C, (long)HW, (int)cpg, slabs);
    int rc = check_launch();
    if (rc != GQHIP_OK) return rc;
    if (apply_silu)
      hipLaunchKernelGGL((gn_apply_nhwc_kernel<1>), dim3((unsigned)(B * slabs)), dim3(256), 0, st, x, gamma, beta, y,
                         stats_ws, pre_bias_or_null, (int)C, (long)HW, (int)cpg, eps, slabs);
    else
      hipLaunchKernelGGL((gn_apply_nhwc_kernel<0>), dim3((unsigned)(B * slabs)), dim3(256), 0, st, x, gamma, beta, y,
                         stats_ws, pre_bias_or_null, (int)C, (long)HW, (int)cpg, eps, slabs);
    return check_launch();
  }
  if (layout != GQHIP_LAYOUT_NCHW || HW % 4 != 0) return GQHIP_ERR_INVALID_ARG;   // callers fall back to torch
  if (stats_zero(stats_ws, sizeof(int64_t) * kStatWords * bg, st) != hipSuccess) return check_launch();
  // ~16 KiB of input per block keeps >= 2k blocks in flight at the big resolutions
  int slices = (int)((chunk + 4095) / 4096);
  if (slices > 256) slices = 256;
  if (slices < 1) slices = 1;
  hipLaunchKernelGGL(gn_stats_kernel, dim3((unsigned)(bg * slices)), dim3(256), 0, st, x, pre_bias_or_null, stats_ws,
                     (long)chunk, slices, (long)HW, (int)cpg, (int)groups);
  int rc = check_launch();
  if (rc != GQHIP_OK) return rc;
  int segs = (int)((HW + 8191) / 8192);
  if (segs < 1) segs = 1;
  const dim3 grid((unsigned)(B * C * segs));
  if (apply_silu)
    hipLaunchKernelGGL((gn_apply_kernel<1>), grid, dim3(256), 0, st, x, gamma, beta, y, stats_ws, pre_bias_or_null,
                       (int)C, (long)HW, (int)cpg, eps, segs);
  else
    hipLaunchKernelGGL((gn_apply_kernel<0>), grid, dim3(256), 0, st, x, gamma, beta, y, stats_ws, pre_bias_or_null,
                       (int)C, (long)HW, (int)cpg, eps, segs);
  return check_launch();
}

int add_bias_f32(const float *a, const float *b, const float *bias_or_null, float *y, int64_t B, int64_t C,
                 int64_t HW, int layout, void *stream) {
  if (B < 0 || C < 1 || HW < 1) return GQHIP_ERR_INVALID_ARG;
  if (B == 0) return GQHIP_OK;
  if (!a || !b || !y) return GQHIP_ERR_INVALID_ARG;
  const long total4 = (long)(B * C * HW / 4);
  long blocks = (total4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (layout == GQHIP_LAYOUT_NHWC) {
    if (C % 4 != 0) return GQHIP_ERR_INVALID_ARG;
    hipLaunchKernelGGL(add_bias_nhwc_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a, b, bias_or_null, y, (int)C,
                       total4);
  } else {
    if (layout != GQHIP_LAYOUT_NCHW || HW % 4 != 0) return GQHIP_ERR_INVALID_ARG;
    hipLaunchKernelGGL(add_bias_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a, b, bias_or_null, y, (int)C,
                       (long)HW, total4);
  }
  return check_launch();
}

// slabs of ~16 pixels per thread, as gn_silu_f32's NHWC path
static int nhwc_slabs(int64_t C, int64_t HW) {
  const int lanes = (int)(256 / (C / 4));
  int slabs = (int)((HW + (int64_t)lanes * 16 - 1) / ((int64_t)lanes * 16));
  if (slabs > 1024) slabs = 1024;
  return slabs < 1 ? 1 : slabs;
}

int add_bias_stats_f32(const float *a, const float *b, const float *bias_or_null, float *y, int64_t B, int64_t C,
                       int64_t HW, int64_t groups, int64_t *stats_out, void *stream) {
  if (B < 0 || C < 1 || HW < 1 || groups < 1 || C % groups != 0) return GQHIP_ERR_INVALID_ARG;
  if (B == 0) return GQHIP_OK;
  if (!a || !b || !y || !stats_out) return GQHIP_ERR_INVALID_ARG;
  const int64_t cpg = C / groups;
  if (cpg % 4 != 0 || 256 % (C / 4) != 0 || groups > 64) return GQHIP_ERR_INVALID_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (stats_zero(stats_out, sizeof(int64_t) * kStatWords * B * groups, st) != hipSuccess) return check_launch();
  const int slabs = nhwc_slabs(C, HW);
  hipLaunchKernelGGL(add_bias_stats_nhwc_kernel, dim3((unsigned)(B * slabs)), dim3(256), 0, st, a, b, bias_or_null, y,
                     stats_out, (int)C, (long)HW, (int)cpg, slabs);
  return check_launch();
}

int gn_apply_f32(const float *x, const float *gamma, const float *beta, float *y, int64_t B, int64_t C, int64_t HW,
                 int64_t groups, double eps, int apply_silu, const int64_t *stats, void *stream) {
  if (B < 0 || C < 1 || HW < 1 || groups < 1 || C % groups != 0) return GQHIP_ERR_INVALID_ARG;
  if (B == 0) return GQHIP_OK;
  if (!x || !gamma || !beta || !y || !stats) return GQHIP_ERR_INVALID_ARG;
  const int64_t cpg = C / groups;
  if (cpg % 4 != 0 || 256 % (C / 4) != 0 || groups > 64) return GQHIP_ERR_INVALID_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int slabs = nhwc_slabs(C, HW);
  if (apply_silu)
    hipLaunchKernelGGL((gn_apply_nhwc_kernel<1>), dim3((unsigned)(B * slabs)), dim3(256), 0, st, x, gamma, beta, y, stats,
                       (const float *)nullptr, (int)C, (long)HW, (int)cpg, eps, slabs);
  else
    hipLaunchKernelGGL((gn_apply_nhwc_kernel<0>), dim3((unsigned)(B * slabs)), dim3(256), 0, st, x, gamma, beta, y, stats,
                       (const float *)nullptr, (int)C, (long)HW, (int)cpg, eps, slabs);
  return check_launch();
}

int wino_in_nhwc_f32(const float *x, float *V, int64_t B, int64_t H, int64_t W, int64_t C, void *stream) {
  if (B < 0 || H < 2 || W < 2 || H % 2 || W % 2 || C < 4 || C % 4 != 0) return GQHIP_ERR_INVALID_ARG;
  if (B == 0) return GQHIP_OK;
  if (!x || !V) return GQHIP_ERR_INVALID_ARG;
  const long tiles = (long)(B * (H / 2) * (W / 2)), total = tiles * (C / 4);
  long blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(wino_in_nhwc_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                     (void *)V, (int)H, (int)W, (int)(C / 4), tiles, total, 1.0f);
  return check_launch();
}

static int wino_in_f16_impl(int vm, const float *x, void *V, int64_t B, int64_t H, int64_t W, int64_t C, int tile,
                            float scale, void *stream) {
  if ((tile != 2 && tile != 4) || B < 0 || H < tile || W < tile || H % tile || W % tile || C < 4 || C % 4 != 0 ||
      !(scale > 0.f))
    return GQHIP_ERR_INVALID_ARG;
  if (B == 0) return GQHIP_OK;
  if (!x || !V) return GQHIP_ERR_INVALID_ARG;
  const long tiles = (long)(B * (H / tile) * (W / tile)), total = tiles * (C / 4);
  long blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipStream_t st = static_cast<hipStream_t>(stream);
#define GQ_WIN(K, VM) \
  hipLaunchKernelGGL(K<VM>, dim3((unsigned)blocks), dim3(256), 0, st, x, V, (int)H, (int)W, (int)(C / 4), tiles, total, scale)
  if (tile == 4) { if (vm == 2) GQ_WIN(wino4_in_nhwc_kernel, 2); else GQ_WIN(wino4_in_nhwc_kernel, 1); }
  else { if (vm == 2) GQ_WIN(wino_in_nhwc_kernel, 2); else GQ_WIN(wino_in_nhwc_kernel, 1); }
#undef GQ_WIN
  return check_launch();
}

int wino_in_nhwc_f16x3(const float *x, void *V3, int64_t B, int64_t H, int64_t W, int64_t C, int tile, float scale,
                       void *stream) {
  return wino_in_f16_impl(1, x, V3, B, H, W, C, tile, scale, stream);
}

int wino_in_nhwc_f16x2(const float *x, void *V2, int64_t B, int64_t H, int64_t W, int64_t C, int tile, float scale,
                       void *stream) {
  return wino_in_f16_impl(2, x, V2, B, H, W, C, tile, scale, stream);
}

int wino_gemm_f16x2(const void *V2, const void *Wf, float *M, int64_t P, int64_t tiles, int64_t Cin, int64_t Cout,
                    void *stream) {
  if (P < 1 || tiles < 0 || tiles > 0x3fffffff || tiles % 256 != 0 || Cin < 32 || Cin % 32 != 0 || Cin > 4096 || Cout < 128 ||
      Cout % 128 != 0 || Cout > 4096)
    return GQHIP_ERR_INVALID_ARG;
  if (tiles == 0) return GQHIP_OK;
  if (!V2 || !Wf || !M) return GQHIP_ERR_INVALID_ARG;
  WinoGemm2Params wp{};
  wp.V2 = static_cast<const _Float16 *>(V2); wp.Wf = static_cast<const _Float16 *>(Wf); wp.M = M; wp.tiles = tiles;
  wp.cin = (int)Cin; wp.cout = (int)Cout; wp.nnb = (int)(Cout / 128);
  wp.mtiles = tiles / 256; wp.ntile_total = P * wp.mtiles; wp.tiles_per_xcd = (wp.ntile_total + 7) / 8;
  // 256 x 256 tiles (8 waves, one block per CU) where the shape allows: Cin % 64 == 0, Cout % 256 == 0; GQHIP_WGEMM=128 keeps
  // the 256 x 128 form (A/B)
  static const int env_w = getenv("GQHIP_WGEMM") ? atoi(getenv("GQHIP_WGEMM")) : 0;
  const bool wide = Cin % 64 == 0 && Cout % 256 == 0 && env_w != 128;
  if (wide) wp.nnb = (int)(Cout / 256);
  const long blocks = 8 * wp.tiles_per_xcd * wp.nnb;
  if (blocks > 0x7fffffffL) return GQHIP_ERR_INVALID_ARG;
  if (wide)
    hipLaunchKernelGGL(wino_gemm_f16x2_w8_kernel, dim3((unsigned)blocks), dim3(512), 0, static_cast<hipStream_t>(stream), wp);
  else
    hipLaunchKernelGGL(wino_gemm_f16x2_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), wp);
  return check_launch();
}

int gn_stats_f32(const float *x, const float *pre_bias_or_null, int64_t B, int64_t C, int64_t HW, int64_t groups,
                 int64_t *stats_out, void *stream) {
  if (B < 0 || C < 1 || HW < 1 || groups < 1 || C % groups != 0) return GQHIP_ERR_INVALID_ARG;
  if (B == 0) return GQHIP_OK;
  if (!x || !stats_out) return GQHIP_ERR_INVALID_ARG;
  const int64_t cpg = C / groups;
  if (cpg % 4 != 0 || 256 % (C / 4) != 0 || groups > 64) return GQHIP_ERR_INVALID_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (stats_zero(stats_out, sizeof(int64_t) * kStatWords * B * groups, st) != hipSuccess) return check_launch();
  const int slabs = nhwc_slabs(C, HW);
  hipLaunchKernelGGL(gn_stats_nhwc_kernel, dim3((unsigned)(B * slabs)), dim3(256), 0, st, x, pre_bias_or_null, stats_out,
                     (int)C, (long)HW, (int)cpg, slabs);
  return check_launch();
}

static int wino_in_gn_impl(int tile, int f16, const float *x, const float *gamma, const float *beta,
                           const float *pre_bias_or_null, const int64_t *stats, void *V, int64_t B, int64_t H, int64_t W,
                           int64_t C, int64_t groups, double eps, int apply_silu, float scale, void *stream) {
  if (B < 0 || H < tile || W < tile || H % tile || W % tile || C < 4 || C % 4 != 0 || groups < 1 || C % groups != 0 ||
      (C / groups) % 4 != 0 || !(scale > 0.f))
    return GQHIP_ERR_INVALID_ARG;
  if (B == 0) return GQHIP_OK;
  if (!x || !gamma || !beta || !stats || !V) return GQHIP_ERR_INVALID_ARG;
  // F(4x4,3x3) on an fp16 operand: two channels per thread (register pressure: see the kernel); 4 otherwise
  const int vw = (tile == 4 && f16 != 0) ? 2 : 4;
  const long tiles = (long)(B * (H / tile) * (W / tile)), total = tiles * (C / vw);
  long blocks = (total + 255) / 256;
  if (blocks > 32768) blocks = 32768;
  hipStream_t st = static_cast<hipStream_t>(stream);
#define GQ_WGN(K, ...)                                                                                                  \
  hipLaunchKernelGGL((K<__VA_ARGS__>), dim3((unsigned)blocks), dim3(256), 0, st, x, gamma, beta, pre_bias_or_null, stats, \
                     V, (int)H, (int)W, (int)(C / vw), (int)(C / groups), eps, tiles, total, scale)
  if (tile == 4) {
    if (apply_silu) {
      if (f16 == 2) GQ_WGN(wino4_in_gn_nhwc_kernel, 1, 2, 2); else if (f16 == 1) GQ_WGN(wino4_in_gn_nhwc_kernel, 1, 1, 2);
      else GQ_WGN(wino4_in_gn_nhwc_kernel, 1, 0, 4);
    } else {
      if (f16 == 2) GQ_WGN(wino4_in_gn_nhwc_kernel, 0, 2, 2); else if (f16 == 1) GQ_WGN(wino4_in_gn_nhwc_kernel, 0, 1, 2);
      else GQ_WGN(wino4_in_gn_nhwc_kernel, 0, 0, 4);
    }
  } else {
    if (apply_silu) {
      if (f16 == 2) GQ_WGN(wino_in_gn_nhwc_kernel, 1, 2); else if (f16 == 1) GQ_WGN(wino_in_gn_nhwc_kernel, 1, 1);
      else GQ_WGN(wino_in_gn_nhwc_kernel, 1, 0);
    } else {
      if (f16 == 2) GQ_WGN(wino_in_gn_nhwc_kernel, 0, 2); else if (f16 == 1) GQ_WGN(wino_in_gn_nhwc_kernel, 0, 1);
      else GQ_WGN(wino_in_gn_nhwc_kernel, 0, 0);
    }
  }
#undef GQ_WGN
  return check_launch();
}

int wino_in_gn_nhwc_f32(const float *x, const float *gamma, const float *beta, const float *pre_bias_or_null,
                        const int64_t *stats, float *V, int64_t B, int64_t H, int64_t W, int64_t C, int64_t groups,
                        double eps, int apply_silu, void *stream) {
  return wino_in_gn_impl(2, 0, x, gamma, beta, pre_bias_or_null, stats, V, B, H, W, C, groups, eps, apply_silu, 1.0f, stream);
}

int wino4_in_gn_nhwc_f32(const float *x, const float *gamma, const float *beta, const float *pre_bias_or_null,
                         const int64_t *stats, float *V, int64_t B, int64_t H, int64_t W, int64_t C, int64_t groups,
                         double eps, int apply_silu, void *stream) {
  return wino_in_gn_impl(4, 0, x, gamma, beta, pre_bias_or_null, stats, V, B, H, W, C, groups, eps, apply_silu, 1.0f, stream);
}

int wino_in_gn_nhwc_f16x3(const float *x, const float *gamma, const float *beta, const float *pre_bias_or_null,
                          const int64_t *stats, void *V3, int64_t B, int64_t H, int64_t W, int64_t C, int64_t groups,
                          double eps, int apply_silu, int tile, float scale, void *stream) {
  if (tile != 2 && tile != 4) return GQHIP_ERR_INVALID_ARG;
  return wino_in_gn_impl(tile, 1, x, gamma, beta, pre_bias_or_null, stats, V3, B, H, W, C, groups, eps, apply_silu, scale,
                         stream);
}

int wino_in_gn_nhwc_f16x2(const float *x, const float *gamma, const float *beta, const float *pre_bias_or_null,
                          const int64_t *stats, void *V2, int64_t B, int64_t H, int64_t W, int64_t C, int64_t groups,
                          double eps, int apply_silu, int tile, float scale, void *stream) {
  if (tile != 2 && tile != 4) return GQHIP_ERR_INVALID_ARG;
  return wino_in_gn_impl(tile, 2, x, gamma, beta, pre_bias_or_null, stats, V2, B, H, W, C, groups, eps, apply_silu, scale,
                         stream);
}

static bool conv3_groups_ok(int64_t Cout, int64_t groups_out) {
  if (groups_out < 1 || Cout % groups_out != 0) return false;
  const int64_t cpg = Cout / groups_out;
  return cpg % 4 == 0 && 128 % cpg == 0;
}

static void conv3_fill(Conv3Params &cp, const void *Wf, const float *bias, const float *res, float *y, int64_t *stats, int64_t B,
                       int64_t H, int64_t W, int64_t Cin, int64_t Cout, int64_t groups_out, float mscale) {
  cp.Wf = static_cast<const _Float16 *>(Wf);
  cp.bias = bias; cp.res = res; cp.y = y; cp.stats = stats;
  cp.H = (int)H; cp.W = (int)W; cp.nch = (int)(Cin / 16); cp.cpg = stats ? (int)(Cout / groups_out) : 4;
  cp.cout = (int)Cout; cp.nnb = (int)(Cout / 128);
  cp.tiles_x = (int)(W / kC3TW); cp.tiles_y = (int)(H / kC3TH);
  cp.ntiles = (long)B * cp.tiles_x * cp.tiles_y;
  cp.tiles_per_xcd = (cp.ntiles + 7) / 8;
  cp.mscale = mscale;
}

int conv3x3_gn_f16x3(const float *x, const float *gamma, const float *beta, const float *pre_bias_or_null,
                     const int64_t *stats_in, int64_t groups_in, double eps, int apply_silu, float scale, const void *Wf,
                     const float *bias_or_null, const float *res_or_null, float *y, int64_t *stats_out_or_null, int64_t B,
                     int64_t H, int64_t W, int64_t Cin, int64_t Cout, int64_t groups_out, float mscale, void *stream) {
  if (B < 0 || H < kC3TH || W < kC3TW || H % kC3TH || W % kC3TW || Cin < 32 || Cin % 32 != 0 || Cin > 512 || (Cout != 128 && Cout != 256) ||
      H * W > (1 << 22) || groups_in < 1 || Cin % groups_in != 0 || !(scale > 0.f))
    return GQHIP_ERR_INVALID_ARG;
  if (stats_out_or_null && !conv3_groups_ok(Cout, groups_out)) return GQHIP_ERR_INVALID_ARG;
  if (B == 0) return GQHIP_OK;
  if (!x || !gamma || !beta || !stats_in || !Wf || !y) return GQHIP_ERR_INVALID_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (stats_out_or_null && stats_zero(stats_out_or_null, sizeof(int64_t) * kStatWords * B * groups_out, st) != hipSuccess)
    return check_launch();
  Conv3GnParams gp{};
  Conv3Params &cp = gp.c;
  cp.Xs = nullptr;
  conv3_fill(cp, Wf, bias_or_null, res_or_null, y, stats_out_or_null, B, H, W, Cin, Cout, groups_out, mscale);
  gp.x = x; gp.gamma = gamma; gp.beta = beta; gp.pre_bias = pre_bias_or_null; gp.stats_in = stats_in;
  gp.cin = (int)Cin; gp.cpg_in = (int)(Cin / groups_in); gp.eps = eps; gp.scale = scale;
  const dim3 grid((unsigned)(8 * cp.tiles_per_xcd * cp.nnb));
  static const int env_dyn = getenv("GQHIP_C3_DYNLDS") ? atoi(getenv("GQHIP_C3_DYNLDS")) : 0;   // diagnostic: extra LDS -> one block per CU
  if (Cout == 128) {
    if (apply_silu) hipLaunchKernelGGL((conv3x3_gn_f16x3_kernel<1, 128>), grid, dim3(256), env_dyn, st, gp);
    else hipLaunchKernelGGL((conv3x3_gn_f16x3_kernel<0, 128>), grid, dim3(256), 0, st, gp);
  } else {
    if (apply_silu) hipLaunchKernelGGL((conv3x3_gn_f16x3_kernel<1, 256>), grid, dim3(256), 0, st, gp);
    else hipLaunchKernelGGL((conv3x3_gn_f16x3_kernel<0, 256>), grid, dim3(256), 0, st, gp);
  }
  return check_launch();
}

#ifdef GQHIP_CLOCK_STAMPS
extern "C" int gqhip_debug_c3_stamps(unsigned long long *out, int64_t words) {   // diagnostic build only
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_c3_stamps), sizeof(unsigned long long) * words) == hipSuccess ? 0 : 1;
}
#endif

int conv1x1_f16x3(const float *x, const float *pre_bias_or_null, const void *Wf, const float *scales_dev_or_null, float scale,
                  float mscale, const float *bias_or_null, const float *res_or_null, float *y, int64_t *stats_out_or_null, int64_t B, int64_t HW,
                  int64_t Cin, int64_t Cout, int64_t groups_out, void *stream) {
  if (B < 0 || HW < 256 || HW % 256 != 0 || Cin < 32 || Cin % 32 != 0 || (Cout != 128 && Cout != 256 && Cout != 512 && Cout != 1536) ||
      HW > (1 << 24) || (!scales_dev_or_null && !(scale > 0.f)))
    return GQHIP_ERR_INVALID_ARG;
  if (stats_out_or_null && !conv3_groups_ok(Cout, groups_out)) return GQHIP_ERR_INVALID_ARG;
  if (B == 0) return GQHIP_OK;
  if (!x || !Wf || !y) return GQHIP_ERR_INVALID_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (stats_out_or_null && stats_zero(stats_out_or_null, sizeof(int64_t) * kStatWords * B * groups_out, st) != hipSuccess)
    return check_launch();
  Conv1Params gp{};
  Conv3Params &cp = gp.c;
  conv3_fill(cp, Wf, bias_or_null, res_or_null, y, stats_out_or_null, B, HW / 32, 32, Cin, Cout, groups_out, mscale);
  cp.nch = (int)(Cin / 16);
  gp.x = x; gp.pre_bias = pre_bias_or_null; gp.scales_dev = scales_dev_or_null; gp.scale = scale; gp.cin = (int)Cin;
  const dim3 grid((unsigned)(8 * cp.tiles_per_xcd * cp.nnb));
  if (Cout == 128) hipLaunchKernelGGL(conv1x1_f16x3_kernel<128>, grid, dim3(256), 0, st, gp);
  else if (Cout == 256) hipLaunchKernelGGL(conv1x1_f16x3_kernel<256>, grid, dim3(256), 0, st, gp);
  else if (Cout == 512) hipLaunchKernelGGL(conv1x1_f16x3_kernel<512>, grid, dim3(256), 0, st, gp);
  else hipLaunchKernelGGL(conv1x1_f16x3_kernel<1536>, grid, dim3(256), 0, st, gp);
  return check_launch();
}

int conv3x3s2_f16x3(const float *x, const void *Wf, const float *scales_dev_or_null, float scale, float mscale,
                    const float *bias_or_null, float *y, int64_t *stats_out_or_null, int64_t B, int64_t Hin, int64_t Win,
                    int64_t Cin, int64_t Cout, int64_t groups_out, void *stream) {
  // output H = Hin / 2, W = Win / 2 (the reference pads one zero row / column at the bottom / right: unet.py:92-95)
  if (B < 0 || Hin < 2 * kC3TH || Win < 2 * kC3TW || Hin % (2 * kC3TH) || Win % (2 * kC3TW) || Cin < 16 || Cin % 16 != 0 ||
      (Cout != 128 && Cout != 256 && Cout != 512) || Hin * Win > (1 << 24) || (!scales_dev_or_null && !(scale > 0.f)))
    return GQHIP_ERR_INVALID_ARG;
  if (stats_out_or_null && !conv3_groups_ok(Cout, groups_out)) return GQHIP_ERR_INVALID_ARG;
  if (B == 0) return GQHIP_OK;
  if (!x || !Wf || !y) return GQHIP_ERR_INVALID_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (stats_out_or_null && stats_zero(stats_out_or_null, sizeof(int64_t) * kStatWords * B * groups_out, st) != hipSuccess)
    return check_launch();
  Conv3S2Params gp{};
  Conv3Params &cp = gp.c;
  conv3_fill(cp, Wf, bias_or_null, nullptr, y, stats_out_or_null, B, Hin / 2, Win / 2, Cin, Cout, groups_out, mscale);
  gp.x = x; gp.scales_dev = scales_dev_or_null; gp.scale = scale; gp.cin = (int)Cin; gp.Hin = (int)Hin; gp.Win = (int)Win;
  const dim3 grid((unsigned)(8 * cp.tiles_per_xcd * cp.nnb));
  if (Cout == 128) hipLaunchKernelGGL(conv3x3s2_f16x3_kernel<128>, grid, dim3(256), 0, st, gp);
  else if (Cout == 256) hipLaunchKernelGGL(conv3x3s2_f16x3_kernel<256>, grid, dim3(256), 0, st, gp);
  else hipLaunchKernelGGL(conv3x3s2_f16x3_kernel<512>, grid, dim3(256), 0, st, gp);
  return check_launch();
}

int upconv2x_f16x3(const float *x, const void *Wf, const float *scales_dev_or_null, float scale, float mscale,
                   const float *bias_or_null, float *y, int64_t *stats_out_or_null, int64_t B, int64_t H, int64_t W,
                   int64_t Cin, int64_t Cout, int64_t groups_out, void *stream) {
  if (B < 0 || H < kC3TH || W < kC3TW || H % kC3TH || W % kC3TW || Cin < 16 || Cin % 16 != 0 ||
      (Cout != 128 && Cout != 256 && Cout != 512) || H * W > (1 << 22) || (!scales_dev_or_null && !(scale > 0.f)))
    return GQHIP_ERR_INVALID_ARG;
  if (stats_out_or_null && !conv3_groups_ok(Cout, groups_out)) return GQHIP_ERR_INVALID_ARG;
  if (B == 0) return GQHIP_OK;
  if (!x || !Wf || !y) return GQHIP_ERR_INVALID_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (stats_out_or_null && stats_zero(stats_out_or_null, sizeof(int64_t) * kStatWords * B * groups_out, st) != hipSuccess)
    return check_launch();
  Upconv2Params gp{};
  Conv3Params &cp = gp.c;
  conv3_fill(cp, Wf, bias_or_null, nullptr, y, stats_out_or_null, B, H, W, Cin, Cout, groups_out, mscale);
  gp.x = x; gp.scales_dev = scales_dev_or_null; gp.scale = scale; gp.cin = (int)Cin;
  const dim3 grid((unsigned)(8 * cp.tiles_per_xcd * cp.nnb), 4);
  if (Cout == 128) hipLaunchKernelGGL(upconv2x_f16x3_kernel<128>, grid, dim3(256), 0, st, gp);
  else if (Cout == 256) hipLaunchKernelGGL(upconv2x_f16x3_kernel<256>, grid, dim3(256), 0, st, gp);
  else hipLaunchKernelGGL(upconv2x_f16x3_kernel<512>, grid, dim3(256), 0, st, gp);
  return check_launch();
}

int conv3x3_gn_small_f32(const float *x, const float *gamma, const float *beta, const float *pre_bias_or_null,
                         const int64_t *stats_in, int64_t groups_in, double eps, int apply_silu, const float *w_ohwi,
                         const float *bias_or_null, float *y, int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t Cout,
                         void *stream) {
  if (B < 0 || H < 16 || W < 16 || H % 16 || W % 16 || Cin < 32 || Cin % 32 != 0 || Cin > 512 || Cout < 1 || Cout > 4 ||
      groups_in < 1 || Cin % groups_in != 0 || B * (H / 16) * (W / 16) > 0x7fffffffL)
    return GQHIP_ERR_INVALID_ARG;
  if (B == 0) return GQHIP_OK;
  if (!x || !gamma || !beta || !stats_in || !w_ohwi || !y) return GQHIP_ERR_INVALID_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)(B * (H / 16) * (W / 16)));
  const int cpg = (int)(Cin / groups_in);
#define GQ_CS(S, CO)                                                                                                      \
  hipLaunchKernelGGL((conv3x3_gn_small_kernel<S, CO>), grid, dim3(256), 0, st, x, gamma, beta, pre_bias_or_null, stats_in, \
                     w_ohwi, bias_or_null, y, (int)H, (int)W, (int)Cin, cpg, eps)
#define GQ_CS2(CO) do { if (apply_silu) GQ_CS(1, CO); else GQ_CS(0, CO); } while (0)
  switch (Cout) {
    case 1: GQ_CS2(1); break;
    case 2: GQ_CS2(2); break;
    case 3: GQ_CS2(3); break;
    default: GQ_CS2(4); break;
  }
#undef GQ_CS2
#undef GQ_CS
  return check_launch();
}

int conv3x3_cin_small_f32(const float *x, const float *wk, const float *bias_or_null, float *y, int64_t *stats_out_or_null,
                          int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int64_t groups_out, void *stream) {
  if (B < 0 || H < 8 || W < 32 || H % 8 || W % 32 || Cin < 1 || Cin > 4 || Cout != 128 || B * (H / 8) * (W / 32) > 0x7fffffffL)
    return GQHIP_ERR_INVALID_ARG;
  if (stats_out_or_null && groups_out != 32) return GQHIP_ERR_INVALID_ARG;
  if (B == 0) return GQHIP_OK;
  if (!x || !wk || !y) return GQHIP_ERR_INVALID_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (stats_out_or_null && stats_zero(stats_out_or_null, sizeof(int64_t) * kStatWords * B * 32, st) != hipSuccess)
    return check_launch();
  ConvInParams cp{};
  cp.x = x; cp.wk = wk; cp.bias = bias_or_null; cp.y = y; cp.stats = stats_out_or_null; cp.H = (int)H; cp.W = (int)W;
  const dim3 grid((unsigned)(B * (H / 8) * (W / 32)));
  switch (Cin) {
    case 1: hipLaunchKernelGGL(conv3x3_cin_small_kernel<1>, grid, dim3(256), 0, st, cp); break;
    case 2: hipLaunchKernelGGL(conv3x3_cin_small_kernel<2>, grid, dim3(256), 0, st, cp); break;
    case 3: hipLaunchKernelGGL(conv3x3_cin_small_kernel<3>, grid, dim3(256), 0, st, cp); break;
    default: hipLaunchKernelGGL(conv3x3_cin_small_kernel<4>, grid, dim3(256), 0, st, cp); break;
  }
  return check_launch();
}

int conv3x3_f32(const float *x, const float *gamma_or_null, const float *beta_or_null, const float *pre_bias_or_null,
                const int64_t *stats_or_null, int64_t groups_in, double eps, int apply_silu, const float *wk,
                const float *bias_or_null, float *y, int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t Cout, void *stream) {
  if (B < 0 || H < 1 || W < 1 || Cin < 8 || Cout < 4 || Cout % 4 != 0 || B * H * ((W + 31) / 32) > 0x7fffffffL)
    return GQHIP_ERR_INVALID_ARG;
  if (B == 0) return GQHIP_OK;
  if (!x || !wk || !y) return GQHIP_ERR_INVALID_ARG;
  const bool gn = stats_or_null != nullptr;
  if (gn && (!gamma_or_null || !beta_or_null || groups_in < 1 || Cin % groups_in != 0 || Cin > 1024)) return GQHIP_ERR_INVALID_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  ConvF32Params cp{};
  cp.x = x; cp.gamma = gamma_or_null; cp.beta = beta_or_null; cp.pre_bias = pre_bias_or_null; cp.stats = stats_or_null;
  cp.wk = wk; cp.bias = bias_or_null; cp.y = y;
  cp.H = (int)H; cp.W = (int)W; cp.Cin = (int)Cin; cp.Cout = (int)Cout; cp.cpg = gn ? (int)(Cin / groups_in) : 1; cp.eps = eps;
  const unsigned segs = (unsigned)(B * H * ((W + 31) / 32));
  if (Cin % 64 == 0) {
    // many input channels: K split over the waves of a block, partial tiles added in wave order
    const dim3 grid(segs, (unsigned)((Cout + 31) / 32));
    if (gn && apply_silu) hipLaunchKernelGGL((conv3x3_f32_ksplit_kernel<true, 1>), grid, dim3(256), 0, st, cp);
    else if (gn) hipLaunchKernelGGL((conv3x3_f32_ksplit_kernel<true, 0>), grid, dim3(256), 0, st, cp);
    else hipLaunchKernelGGL((conv3x3_f32_ksplit_kernel<false, 0>), grid, dim3(256), 0, st, cp);
    return check_launch();
  }
  if (gn) return GQHIP_ERR_INVALID_ARG;
  switch (Cin) {
    case 8: hipLaunchKernelGGL((conv3x3_f32_nsplit_kernel<8>), dim3(segs), dim3(256), 0, st, cp); break;
    case 16: hipLaunchKernelGGL((conv3x3_f32_nsplit_kernel<16>), dim3(segs), dim3(256), 0, st, cp); break;
    case 32: hipLaunchKernelGGL((conv3x3_f32_nsplit_kernel<32>), dim3(segs), dim3(256), 0, st, cp); break;
    default: return GQHIP_ERR_INVALID_ARG;
  }
  return check_launch();
}

int wino_out_nhwc_f32(const float *M, float *y, int64_t B, int64_t H, int64_t W, int64_t C, float mscale, void *stream) {
  if (B < 0 || H < 2 || W < 2 || H % 2 || W % 2 || C < 4 || C % 4 != 0) return GQHIP_ERR_INVALID_ARG;
  if (B == 0) return GQHIP_OK;
  if (!M || !y) return GQHIP_ERR_INVALID_ARG;
  const long tiles = (long)(B * (H / 2) * (W / 2)), total = tiles * (C / 4);
  long blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(wino_out_nhwc_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), M, y,
                     (int)H, (int)W, (int)(C / 4), tiles, total, mscale);
  return check_launch();
}

int wino4_in_nhwc_f32(const float *x, float *V, int64_t B, int64_t H, int64_t W, int64_t C, void *stream) {
  if (B < 0 || H < 4 || W < 4 || H % 4 || W % 4 || C < 4 || C % 4 != 0) return GQHIP_ERR_INVALID_ARG;
  if (B == 0) return GQHIP_OK;
  if (!x || !V) return GQHIP_ERR_INVALID_ARG;
  const long tiles = (long)(B * (H / 4) * (W / 4)), total = tiles * (C / 4);
  long blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(wino4_in_nhwc_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                     (void *)V, (int)H, (int)W, (int)(C / 4), tiles, total, 1.0f);
  return check_launch();
}

int wino4_out_nhwc_f32(const float *M, float *y, int64_t B, int64_t H, int64_t W, int64_t C, float mscale, void *stream) {
  if (B < 0 || H < 4 || W < 4 || H % 4 || W % 4 || C < 4 || C % 4 != 0) return GQHIP_ERR_INVALID_ARG;
  if (B == 0) return GQHIP_OK;
  if (!M || !y) return GQHIP_ERR_INVALID_ARG;
  const long tiles = (long)(B * (H / 4) * (W / 4)), total = tiles * (C / 4);
  long blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(wino4_out_nhwc_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), M, y,
                     (int)H, (int)W, (int)(C / 4), tiles, total, mscale);
  return check_launch();
}

int wino_out_res_nhwc_f32(const float *M, const float *res, const float *bias_or_null, float *y, int64_t *stats_out,
                          int64_t B, int64_t H, int64_t W, int64_t C, int64_t groups, int tile, float mscale,
                          void *stream) {
  if ((tile != 2 && tile != 4) || B < 0 || H < tile || W < tile || H % tile || W % tile || C < 4 || C % 4 != 0 ||
      groups < 1 || C % groups != 0)
    return GQHIP_ERR_INVALID_ARG;
  if (B == 0) return GQHIP_OK;
  if (!M || !y || !stats_out) return GQHIP_ERR_INVALID_ARG;   // res may be NULL: bias + statistics only
  const int64_t cpg = C / groups;
  if (cpg % 4 != 0 || 256 % (C / 4) != 0 || groups > 64) return GQHIP_ERR_INVALID_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (stats_zero(stats_out, sizeof(int64_t) * kStatWords * B * groups, st) != hipSuccess) return check_launch();
  const long tpi = (long)((H / tile) * (W / tile)), tiles = (long)B * tpi;
  // F(4x4,3x3): two channels per thread (see the kernel) wherever a block still spans whole pixels
  const int vw = (tile == 4 && 256 % (C / 2) == 0) ? 2 : 4;
  const int lanes = (int)(256 / (C / vw));
  long slabs = (tpi + (long)lanes * 4 - 1) / ((long)lanes * 4);    // ~4 tiles per thread
  if (slabs > 1024) slabs = 1024;
  if (slabs < 1) slabs = 1;
  const dim3 grid((unsigned)(B * slabs));
  if (tile == 4 && vw == 2)
    hipLaunchKernelGGL((wino_out_res_nhwc_kernel<4, 2>), grid, dim3(256), 0, st, M, res, bias_or_null, y, stats_out, (int)H,
                       (int)W, (int)(C / 2), (int)cpg, tiles, (int)slabs, mscale);
  else if (tile == 4)
    hipLaunchKernelGGL((wino_out_res_nhwc_kernel<4, 4>), grid, dim3(256), 0, st, M, res, bias_or_null, y, stats_out, (int)H,
                       (int)W, (int)(C / 4), (int)cpg, tiles, (int)slabs, mscale);
  else
    hipLaunchKernelGGL((wino_out_res_nhwc_kernel<2, 4>), grid, dim3(256), 0, st, M, res, bias_or_null, y, stats_out, (int)H,
                       (int)W, (int)(C / 4), (int)cpg, tiles, (int)slabs, mscale);
  return check_launch();
}

int attn_split_qkv_f16x3(const float *qkv, void *Q3, void *K3, void *V3, int64_t B, int64_t L, int64_t C, float sq, float sv,
                         void *stream) {
  if (B < 0 || L < 1 || C < 4 || C % 4 != 0 || !(sq > 0.f) || !(sv > 0.f)) return GQHIP_ERR_INVALID_ARG;
  if (B == 0) return GQHIP_OK;
  if (!qkv || !Q3 || !K3 || !V3) return GQHIP_ERR_INVALID_ARG;
  const long total = (long)(B * L * (C / 4));
  long blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(attn_split_qkv_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), qkv,
                     static_cast<_Float16 *>(Q3), static_cast<_Float16 *>(K3), static_cast<_Float16 *>(V3), (long)L,
                     (int)(C / 4), sq, sv, total);
  return check_launch();
}

int attn_softmax_split_f16x3(const float *S, void *P3, int64_t rows, int64_t L, float factor, void *stream) {
  if (rows < 0 || L < 64 || L % 64 != 0 || L > 4096 || !(factor > 0.f)) return GQHIP_ERR_INVALID_ARG;
  if (rows == 0) return GQHIP_OK;
  if (!S || !P3) return GQHIP_ERR_INVALID_ARG;
  const dim3 grid((unsigned)((rows + 3) / 4));
  hipStream_t st = static_cast<hipStream_t>(stream);
  _Float16 *p = static_cast<_Float16 *>(P3);
#define GQ_SM(N) hipLaunchKernelGGL(attn_softmax_split_kernel<N>, grid, dim3(256), 0, st, S, p, (long)rows, factor)
  switch (L / 64) {
    case 1: GQ_SM(1); break;
    case 4: GQ_SM(4); break;
    case 16: GQ_SM(16); break;
    case 36: GQ_SM(36); break;
    case 64: GQ_SM(64); break;
    default: return GQHIP_ERR_INVALID_ARG;
  }
#undef GQ_SM
  return check_launch();
}

int f16_scales_from_gn_stats(const int64_t *stats, int64_t n_bg, double amp, double u_scale, float *scales_out,
                             void *stream) {
  if (!stats || !scales_out || n_bg < 1 || n_bg > 0x7fffffff || !(amp > 0.0) || !(u_scale > 0.0)) return GQHIP_ERR_INVALID_ARG;
  hipLaunchKernelGGL(f16_scales_from_stats_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), stats, (int)n_bg,
                     (float)amp, (float)u_scale, scales_out);
  return check_launch();
}

int upsample2x_nhwc_f32(const float *x, float *y, int64_t B, int64_t H, int64_t W, int64_t C, void *stream) {
  if (B < 0 || H < 1 || W < 1 || C < 4 || C % 4 != 0) return GQHIP_ERR_INVALID_ARG;
  if (B == 0) return GQHIP_OK;
  if (!x || !y) return GQHIP_ERR_INVALID_ARG;
  const long total = (long)(B * H * W * (C / 4));
  long blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(upsample2x_nhwc_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                     y, (int)H, (int)W, (int)(C / 4), total);
  return check_launch();
}

int gqhip_checksum_tensors(const void *table_dev, int64_t count, uint64_t *sums_dev, void *stream) {
  if (count < 0 || count > 65535) return GQHIP_ERR_INVALID_ARG;
  if (count == 0) return GQHIP_OK;
  if (!table_dev || !sums_dev) return GQHIP_ERR_INVALID_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(sums_dev, 0, sizeof(uint64_t) * count, st) != hipSuccess) return check_launch();
  hipLaunchKernelGGL(checksum_tensors_kernel, dim3((unsigned)count, kChecksumSlices), dim3(256), 0, st,
                     static_cast<const ChecksumEntry *>(table_dev), reinterpret_cast<unsigned long long *>(sums_dev));
  return check_launch();
}

}  // extern "C"
