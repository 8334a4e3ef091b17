// gqhip.hip -- C-ABI entry points of libgqhip.so (see include/gqhip.h).
// gfx950 only; built by `make -C vq-vae-from-gaussian-vae_amd/csrc`.
#include "gqhip.h"

#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>
#include <mutex>
#include <utility>
#include <vector>

#include "gq_aux.h"
#include "gq_common.h"
#include "gq_filter.h"
#include "gq_filter_bf16.h"
#include "gq_rerank.h"

using namespace gqhip;

namespace {

thread_local int g_last_hip_error = 0;

inline int check_launch() {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    g_last_hip_error = (int)e;
    return GQHIP_ERR_LAUNCH;
  }
  return GQHIP_OK;
}

// ---- launch plan: identical on the sizing and the launching side -----------
struct Plan {
  bool mfma;            // filter kernel applies (dim in {4,8,16,32}, n >= 32)
  int rt;               // row tiles per wave
  int rows_per_block;   // 128 * rt
  int row_blocks;
  int nsplit;           // code splits
  int tiles_total;
  int tiles_per_split;
  int gt;               // tiles per candidate group: 4 for dim <= 8 and for the split-bf16 filter, else 2
  bool bf16;            // split-bf16 filter (gq_filter_bf16.h) instead of the fp32 MFMA one
  int ct;               // tiles per LDS chunk of the split-bf16 filter
  int waves;            // waves per block: 8 (one block per CU) for the split-bf16 filter, else 4
  // second level of the cascade behind the split-bf16 filter: the fp32 MFMA filter (RT 1) on the undecided rows
  int nsplit2, tiles_per_split2, gt2;
};

// Filter selection: 0 = auto (split-bf16 where it applies: dims 8/16/32), 1 = always the fp32 MFMA filter.
// Initial value from GQHIP_FILTER=fp32|bf16, changed at run time by gqhip_set_filter().  Both filters feed the
// same exact re-rank, so the choice never changes an index.
std::atomic<int> g_filter_kind{[] {
  const char *e = getenv("GQHIP_FILTER");
  return (e && (e[0] == 'f' || e[0] == 'F')) ? 1 : 0;
}()};
bool want_bf16_filter() { return g_filter_kind.load(std::memory_order_relaxed) == 0; }

Plan make_plan(int64_t rows, int64_t n, int64_t dim) {
  Plan pl{};
  pl.mfma = (dim == 4 || dim == 8 || dim == 16 || dim == 32) && n >= 1 && rows >= 1;
  pl.tiles_total = (int)((n + kTileCodes - 1) / kTileCodes);
  static const int env_rt = getenv("GQHIP_RT") ? atoi(getenv("GQHIP_RT")) : 0;
  static const int env_blocks = getenv("GQHIP_TARGET_BLOCKS") ? atoi(getenv("GQHIP_TARGET_BLOCKS")) : 0;
  pl.rt = rows >= 8192 ? 2 : 1;
  if (env_rt == 1 || env_rt == 2) pl.rt = env_rt;
  pl.bf16 = pl.mfma && want_bf16_filter();
  static const int env_waves = getenv("GQHIP_BF16_WAVES") ? atoi(getenv("GQHIP_BF16_WAVES")) : 0;
  pl.waves = pl.bf16 ? (env_waves == 4 ? 4 : 8) : 4;
  pl.rows_per_block = 32 * pl.waves * pl.rt;
  pl.row_blocks = (int)((rows + pl.rows_per_block - 1) / pl.rows_per_block);
  // 4-wave blocks: ~2 blocks per CU on 256 CUs; 8-wave blocks: one per CU.  Splits in multiples of 8 so that
  // blockIdx % 8 (XCD) == split % 8.
  const int target = env_blocks > 0 ? env_blocks : (pl.waves == 8 ? 256 : 512);
  int s = (target + pl.row_blocks - 1) / (pl.row_blocks > 0 ? pl.row_blocks : 1);
  s = ((s + 7) / 8) * 8;
  static const int env_nsplit = getenv("GQHIP_NSPLIT") ? atoi(getenv("GQHIP_NSPLIT")) : 0;
  if (env_nsplit > 0) s = env_nsplit;
  if (s > kMaxSplit) s = kMaxSplit;
  if (s > pl.tiles_total) s = pl.tiles_total;
  if (s < 1) s = 1;
  pl.tiles_per_split = (pl.tiles_total + s - 1) / s;
  // tiles per LDS chunk: 16 at dim 16 with one block per CU (2 x 64 KiB of LDS, half the chunk barriers: +1-2 %)
  static const int env_ct = getenv("GQHIP_BF16_CT") ? atoi(getenv("GQHIP_BF16_CT")) : 0;
  pl.ct = dim == 32 ? 4 : ((dim == 16 && pl.waves == 8 && env_ct != 8) ? 16 : 8);
  static const int env_bgt = getenv("GQHIP_BF16_GT") ? atoi(getenv("GQHIP_BF16_GT")) : 0;
  // split-bf16: the tracker's VALU work overlaps the bf16 MFMAs, so the finest candidate (one tile half =
  // 16 codes) is free in the filter and halves / quarters the exact re-rank work
  // (dim 4: two MFMAs per tile, the epilogue dominates -> the coarse 64-code candidate keeps the tracker cheap)
  pl.gt = pl.bf16 ? (dim == 4 ? 4 : ((pl.waves == 4 && (env_bgt == 2 || env_bgt == 4)) ? env_bgt : 1)) : (dim <= 8 ? 4 : 2);
  pl.tiles_per_split = (pl.tiles_per_split + pl.gt - 1) / pl.gt * pl.gt;   // a tile group never straddles two splits
  pl.nsplit = (pl.tiles_total + pl.tiles_per_split - 1) / pl.tiles_per_split;
  pl.gt2 = dim <= 8 ? 4 : 2;
  int s2 = pl.tiles_total < 16 ? pl.tiles_total : 16;          // 16 splits: 4096 codes per block at N = 65 536
  pl.tiles_per_split2 = ((pl.tiles_total + s2 - 1) / s2 + pl.gt2 - 1) / pl.gt2 * pl.gt2;
  pl.nsplit2 = (pl.tiles_total + pl.tiles_per_split2 - 1) / pl.tiles_per_split2;
  return pl;
}

inline int64_t align256(int64_t v) { return (v + 255) / 256 * 256; }

struct WsLayout {
  int64_t hdr, rec, fb, fb2, rec2, spread, mu, sd, lsd, cbimg, rowimg, total;
};

WsLayout ws_layout(int64_t rows, int64_t n, int64_t dim) {
  const Plan pl = make_plan(rows, n, dim);
  WsLayout w{};
  int64_t off = 0;
  w.hdr = off; off += (int64_t)sizeof(WsHeader);
  w.rec = off; off += align256((int64_t)sizeof(Rec) * rows * (pl.mfma ? pl.nsplit : 0));
  w.fb = off;  off += align256(4 * rows);
  // cascade behind the split-bf16 filter: list B + the second-level fp32 filter's records (kCascadeSplit splits)
  w.fb2 = off;  off += pl.bf16 ? align256(4 * rows) : 0;
  w.rec2 = off; off += pl.bf16 ? align256((int64_t)sizeof(Rec) * rows * pl.nsplit2) : 0;
  w.spread = off; off += align256((int64_t)sizeof(SpreadSlot) * kSpreadRows);
  w.mu = off;  off += align256(4 * rows * dim);
  w.sd = off;  off += align256(4 * rows * dim);
  w.lsd = off; off += align256(4 * rows * dim);
  // split-bf16 operand images: 2*NV vectors of 16 B per (code, half) / (row, half), NV = dim / 8
  const int64_t nvec = dim == 4 ? 2 : dim / 4;
  w.cbimg = off;  off += pl.bf16 ? align256((int64_t)(pl.tiles_total + pl.ct) * nvec * 64 * 16) : 0;
  w.rowimg = off; off += pl.bf16 ? align256(rows * nvec * 2 * 16) : 0;
  w.total = off;
  return w;
}

// ---- profiling recorder ------------------------------------------------------
std::mutex g_prof_mu;
bool g_prof_on = false;
int g_debug_stats = 0;
std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_events;

// When profiling is on, the filter is launched through hipExtLaunchKernelGGL with a start and a
// stop event attached to the dispatch itself, so the elapsed time is the kernel's own duration on
// its stream (what rocprofv3 --kernel-trace reports), not a marker-to-marker bracket.
struct ProfScope {
  hipEvent_t a = nullptr, b = nullptr;
  bool on;
  explicit ProfScope(bool enable = true) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    on = g_prof_on && enable;
    if (on) {
      (void)hipEventCreate(&a);
      (void)hipEventCreate(&b);
    }
  }
  ~ProfScope() {
    if (on) {
      std::lock_guard<std::mutex> lk(g_prof_mu);
      g_prof_events.emplace_back(a, b);
    }
  }
};

template <int MODE>
int launch_filter(const Plan &pl, const FilterParams &fp, int dim, hipStream_t st, bool profile = true) {
  const dim3 grid((unsigned)(pl.row_blocks * pl.nsplit)), block(256);
  ProfScope prof(profile);
#define GQ_LAUNCH(D, R, C, G)                                                                           \
  do {                                                                                                \
    if (prof.on)                                                                                      \
      hipExtLaunchKernelGGL((gq_filter_kernel<D, R, C, MODE, G>), grid, block, 0, st, prof.a, prof.b, 0, fp); \
    else                                                                                              \
      hipLaunchKernelGGL((gq_filter_kernel<D, R, C, MODE, G>), grid, block, 0, st, fp);                  \
  } while (0)
  if (pl.rt == 2) {
    switch (dim) {
      case 4: GQ_LAUNCH(4, 2, 8, 4); break;
      case 8: GQ_LAUNCH(8, 2, 8, 4); break;
      case 16: GQ_LAUNCH(16, 2, 8, 2); break;
      case 32: GQ_LAUNCH(32, 2, 4, 2); break;
      default: return GQHIP_ERR_INVALID_ARG;
    }
  } else {
    switch (dim) {
      case 4: GQ_LAUNCH(4, 1, 8, 4); break;
      case 8: GQ_LAUNCH(8, 1, 8, 4); break;
      case 16: GQ_LAUNCH(16, 1, 8, 2); break;
      case 32: GQ_LAUNCH(32, 1, 4, 2); break;
      default: return GQHIP_ERR_INVALID_ARG;
    }
  }
#undef GQ_LAUNCH
  return check_launch();
}

// re-rank: GROUP = codes per candidate = lanes per row
template <int MODE>
void launch_rerank(const RerankParams &rp, int64_t rows, hipStream_t st) {
  if (rp.gt == 1)
    hipLaunchKernelGGL((gq_rerank_kernel<MODE, 16>), dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, st, rp);
  else if (rp.gt == 2)
    hipLaunchKernelGGL((gq_rerank_kernel<MODE, 32>), dim3((unsigned)((rows + 7) / 8)), dim3(256), 0, st, rp);
  else
    hipLaunchKernelGGL((gq_rerank_kernel<MODE, 64>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, rp);
}

template <int MODE>
int launch_filter_bf16(const Plan &pl, const FilterBfParams &fp, const SplitParams &sp, int dim, hipStream_t st) {
  const long split_threads = (long)pl.tiles_total * 64 + (long)sp.rows * 2;
  const dim3 sgrid((unsigned)((split_threads + 255) / 256)), grid((unsigned)(pl.row_blocks * pl.nsplit)), block(256);
  const dim3 fblock((unsigned)(64 * pl.waves));
  switch (dim) {
    case 4: hipLaunchKernelGGL((bf16_split_kernel<MODE, 4>), sgrid, block, 0, st, sp); break;
    case 8: hipLaunchKernelGGL((bf16_split_kernel<MODE, 8>), sgrid, block, 0, st, sp); break;
    case 16: hipLaunchKernelGGL((bf16_split_kernel<MODE, 16>), sgrid, block, 0, st, sp); break;
    case 32: hipLaunchKernelGGL((bf16_split_kernel<MODE, 32>), sgrid, block, 0, st, sp); break;
    default: return GQHIP_ERR_INVALID_ARG;
  }
  int rc = check_launch();
  if (rc != GQHIP_OK) return rc;
  ProfScope prof;
#define GQ_LAUNCH_BF1(NV, R, C, G, W)                                                                       \
  do {                                                                                                    \
    if (prof.on)                                                                                          \
      hipExtLaunchKernelGGL((gq_filter_bf16_kernel<NV, R, C, G, W>), grid, fblock, 0, st, prof.a, prof.b, 0, fp); \
    else                                                                                                  \
      hipLaunchKernelGGL((gq_filter_bf16_kernel<NV, R, C, G, W>), grid, fblock, 0, st, fp);                  \
  } while (0)
#define GQ_LAUNCH_BF(NV, R, C)                                                                            \
  do {                                                                                                    \
    if (pl.waves == 8 && pl.gt == 4) GQ_LAUNCH_BF1(NV, R, C, 4, 8);                                       \
    else if (pl.waves == 8) GQ_LAUNCH_BF1(NV, R, C, 1, 8);                                                \
    else if (pl.gt == 1) GQ_LAUNCH_BF1(NV, R, C, 1, 4);                                                   \
    else if (pl.gt == 2) GQ_LAUNCH_BF1(NV, R, C, 2, 4);                                                   \
    else GQ_LAUNCH_BF1(NV, R, C, 4, 4);                                                                   \
  } while (0)
  if (pl.rt == 2) {
    switch (dim) {
      case 4: GQ_LAUNCH_BF(0, 2, 8); break;
      case 8: GQ_LAUNCH_BF(1, 2, 8); break;
      case 16: if (pl.ct == 16) GQ_LAUNCH_BF1(2, 2, 16, 1, 8); else GQ_LAUNCH_BF(2, 2, 8); break;
      default: GQ_LAUNCH_BF(4, 2, 4); break;
    }
  } else {
    switch (dim) {
      case 4: GQ_LAUNCH_BF(0, 1, 8); break;
      case 8: GQ_LAUNCH_BF(1, 1, 8); break;
      case 16: GQ_LAUNCH_BF(2, 1, 8); break;
      default: GQ_LAUNCH_BF(4, 1, 4); break;
    }
  }
#undef GQ_LAUNCH_BF
#undef GQ_LAUNCH_BF1
  return check_launch();
}

// filter -> re-rank -> exhaustive, shared by GQ and VQ.
template <int MODE>
int run_argmax(const float *mu, const float *sd, const float *lsd, const float *cb, int64_t *idx,
               float *zhat, int64_t dim, int64_t rows, int64_t n, double beta, float cb_absmax,
               void *workspace, int64_t workspace_bytes, const OutMap &omap, hipStream_t st) {
  if (dim < 1 || dim > kMaxDim || rows < 0 || n < 1 || n > 0x3fffffff || rows > 0x3fffffff)
    return GQHIP_ERR_INVALID_ARG;
  if (rows == 0) return GQHIP_OK;   // empty batch: nothing to do (pointers may be NULL)
  if (!mu || !cb || !idx || (MODE == kModeGQ && !sd)) return GQHIP_ERR_INVALID_ARG;
  const WsLayout w = ws_layout(rows, n, dim);
  if (!workspace || workspace_bytes < w.total) return GQHIP_ERR_WORKSPACE;
  char *ws = static_cast<char *>(workspace);
  WsHeader *hdr = reinterpret_cast<WsHeader *>(ws + w.hdr);
  const Plan pl = make_plan(rows, n, dim);

  auto launch_absmax = [&]() {
    const long count = (long)n * dim;
    const int blocks = (int)((count + 256 * 16 - 1) / (256 * 16));
    hipLaunchKernelGGL(absmax_kernel, dim3(blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks)), dim3(256), 0,
                       st, cb, count, &hdr->absmax);
  };
  const bool need_absmax = !(cb_absmax > 0.f);

  RerankParams rp{};
  rp.mu = mu; rp.sd = sd; rp.lsd = lsd; rp.cb = cb;
  rp.rec = reinterpret_cast<const Rec *>(ws + w.rec);
  rp.idx = idx; rp.zhat = zhat; rp.hdr = hdr;
  rp.fb_list = reinterpret_cast<int *>(ws + w.fb);
  rp.fb2_list = reinterpret_cast<int *>(ws + w.fb2);
  rp.cascade = pl.bf16 ? 1 : 0;
  rp.level = 1;
  rp.spread = reinterpret_cast<SpreadSlot *>(ws + w.spread);
  rp.rows = (int)rows; rp.n = (int)n; rp.dim = (int)dim;
  rp.ef_coeff = pl.bf16 ? (float)(dim == 4 ? 332 : 220 + 24 * dim) : (float)(2 * dim + 4);
  static const double env_ef = getenv("GQHIP_EF_COEFF") ? atof(getenv("GQHIP_EF_COEFF")) : 0.0;   // diagnostics
  if (env_ef > 0.0) rp.ef_coeff = (float)env_ef;
  rp.beta = (float)beta; rp.nsplit = pl.nsplit; rp.gt = pl.gt; rp.all_rows = pl.mfma ? 0 : 1; rp.stats = g_debug_stats;
  rp.omap = omap;

  if (pl.bf16) {
    SplitParams sp{};
    sp.mu = mu; sp.sd = sd; sp.cb = cb;
    sp.cbimg = reinterpret_cast<u32x4 *>(ws + w.cbimg); sp.rowimg = reinterpret_cast<u32x4 *>(ws + w.rowimg);
    sp.rows = (int)rows; sp.n = (int)n; sp.tiles_total = pl.tiles_total; sp.beta = (float)beta;
    FilterBfParams fp{};
    fp.cbimg = sp.cbimg; fp.rowimg = sp.rowimg;
    fp.rec = reinterpret_cast<Rec *>(ws + w.rec);
    fp.rows = (int)rows; fp.n = (int)n;
    fp.nsplit = pl.nsplit; fp.tiles_total = pl.tiles_total; fp.tiles_per_split = pl.tiles_per_split;
    fp.hdr = hdr; fp.absmax = cb_absmax;
    fp.dbg = ws + w.mu;   // scratch rows area (unused by gq_argmax_f32); read by diagnostic builds only
    if (need_absmax && hipMemsetAsync(&hdr->absmax, 0, sizeof(float), st) != hipSuccess) return check_launch();
    int rc = launch_filter_bf16<MODE>(pl, fp, sp, (int)dim, st);
    if (rc != GQHIP_OK) return rc;
    if (need_absmax) launch_absmax();
    launch_rerank<MODE>(rp, rows, st);
    rc = check_launch();
    if (rc != GQHIP_OK) return rc;
    // Cascade: when more than kCascadeMin rows are undecided (ill-conditioned inputs: the split-bf16 margin is ~16x
    // the fp32 one), they go through the fp32 MFMA filter + re-rank before the fp64 second stage.  Both launches
    // read the list length on the device and return at once when it is short.
    FilterParams f2{};
    f2.mu = mu; f2.sd = sd; f2.cb = cb;
    f2.rec = reinterpret_cast<Rec *>(ws + w.rec2);
    f2.rows = (int)rows; f2.n = (int)n; f2.beta = (float)beta;
    f2.nsplit = pl.nsplit2; f2.tiles_total = pl.tiles_total; f2.tiles_per_split = pl.tiles_per_split2;
    f2.hdr = hdr; f2.absmax = 0.f; f2.dbg = nullptr;
    f2.row_list = rp.fb_list; f2.row_count = &hdr->fb_count; f2.min_count = kCascadeMin;
    Plan p2 = pl;
    p2.rt = 1; p2.rows_per_block = 128; p2.row_blocks = (int)((rows + 127) / 128); p2.nsplit = pl.nsplit2;
    rc = launch_filter<MODE>(p2, f2, (int)dim, st, /*profile=*/false);
    if (rc != GQHIP_OK) return rc;
    RerankParams r2 = rp;
    r2.rec = reinterpret_cast<const Rec *>(ws + w.rec2);
    r2.nsplit = pl.nsplit2; r2.gt = pl.gt2; r2.ef_coeff = (float)(2 * dim + 4); r2.level = 2;
    launch_rerank<MODE>(r2, rows, st);
    rc = check_launch();
    if (rc != GQHIP_OK) return rc;
  } else if (pl.mfma) {
    FilterParams fp{};
    fp.mu = mu; fp.sd = sd; fp.cb = cb;
    fp.rec = reinterpret_cast<Rec *>(ws + w.rec);
    fp.rows = (int)rows; fp.n = (int)n; fp.beta = (float)beta;
    fp.nsplit = pl.nsplit; fp.tiles_total = pl.tiles_total; fp.tiles_per_split = pl.tiles_per_split;
    fp.hdr = hdr; fp.absmax = cb_absmax;
    fp.dbg = ws + w.mu;   // scratch rows area (unused by gq_argmax_f32); read by diagnostic builds only
    if (need_absmax && hipMemsetAsync(&hdr->absmax, 0, sizeof(float), st) != hipSuccess) return check_launch();
    int rc = launch_filter<MODE>(pl, fp, (int)dim, st);
    if (rc != GQHIP_OK) return rc;
    if (need_absmax) launch_absmax();   // after the filter (it owns the header init), before the re-rank
    launch_rerank<MODE>(rp, rows, st);
    rc = check_launch();
    if (rc != GQHIP_OK) return rc;
  }
  if (pl.mfma) {
    // rows the filter could not decide: fp64 second-stage filter + exact re-rank (usually none or a handful).
    // One launch; on the device the list length picks the variant: short lists are spread over kSpreadSlices blocks
    // per row, long ones take 8 rows per block.
    const int64_t groups = (rows + kFallbackRows - 1) / kFallbackRows;
    const int64_t sp_rows = rows < kSpreadRows ? rows : kSpreadRows;
    int64_t fb_blocks = groups > sp_rows * kSpreadSlices ? groups : sp_rows * kSpreadSlices;
    if (fb_blocks > 2048) fb_blocks = 2048;
    switch (dim) {
      case 4: hipLaunchKernelGGL((gq_fallback64_kernel<MODE, 4>), dim3((unsigned)fb_blocks), dim3(256), 0, st, rp); break;
      case 8: hipLaunchKernelGGL((gq_fallback64_kernel<MODE, 8>), dim3((unsigned)fb_blocks), dim3(256), 0, st, rp); break;
      case 16: hipLaunchKernelGGL((gq_fallback64_kernel<MODE, 16>), dim3((unsigned)fb_blocks), dim3(256), 0, st, rp); break;
      default: hipLaunchKernelGGL((gq_fallback64_kernel<MODE, 32>), dim3((unsigned)fb_blocks), dim3(256), 0, st, rp); break;
    }
    return check_launch();
  }
  if (hipMemsetAsync(hdr, 0, sizeof(WsHeader), st) != hipSuccess) return check_launch();
  const int ex_blocks = (int)(rows < 4096 ? rows : 4096);
  hipLaunchKernelGGL((gq_exhaustive_kernel<MODE>), dim3((unsigned)ex_blocks), dim3(256), 0, st, rp);
  return check_launch();
}

}  // namespace

extern "C" {

int gqhip_abi_version(void) { return GQHIP_ABI_VERSION; }

const char *gqhip_status_string(int s) {
  switch (s) {
    case GQHIP_OK: return "ok";
    case GQHIP_ERR_INVALID_ARG: return "invalid argument";
    case GQHIP_ERR_WORKSPACE: return "workspace missing or too small";
    case GQHIP_ERR_LAUNCH: return "kernel launch failed";
    case GQHIP_ERR_NO_DEVICE: return "no HIP device";
    default: return "unknown status";
  }
}

int gqhip_last_hip_error(void) { return g_last_hip_error; }

int gqhip_set_filter(int kind) {
  if (kind != GQHIP_FILTER_AUTO && kind != GQHIP_FILTER_FP32) return GQHIP_ERR_INVALID_ARG;
  g_filter_kind.store(kind, std::memory_order_relaxed);
  return GQHIP_OK;
}

int gqhip_get_filter(void) { return g_filter_kind.load(std::memory_order_relaxed); }

int gqhip_debug_plan(int64_t rows, int64_t n, int64_t dim, int64_t *out8) {
  if (!out8 || rows < 1 || n < 1 || dim < 1 || dim > kMaxDim) return GQHIP_ERR_INVALID_ARG;
  const Plan pl = make_plan(rows, n, dim);
  const WsLayout w = ws_layout(rows, n, dim);
  out8[0] = w.rec; out8[1] = pl.mfma ? pl.nsplit : 0; out8[2] = pl.gt; out8[3] = pl.tiles_per_split;
  out8[4] = pl.bf16 ? 1 : 0; out8[5] = pl.bf16 ? (dim == 4 ? 332 : 220 + 24 * dim) : 2 * dim + 4; out8[6] = pl.rt; out8[7] = pl.waves;
  return GQHIP_OK;
}

int64_t gqhip_workspace_bytes(int64_t rows, int64_t n, int64_t dim) {
  if (rows < 0 || n < 1 || dim < 1 || dim > kMaxDim) return -1;
  return ws_layout(rows < 1 ? 1 : rows, n, dim).total;
}

int gqhip_codebook_absmax(const float *cb, int64_t n, int64_t dim, float *absmax_out, void *stream) {
  if (!cb || !absmax_out || n < 1 || dim < 1) return GQHIP_ERR_INVALID_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(absmax_out, 0, sizeof(float), st) != hipSuccess) return check_launch();
  const long count = (long)n * dim;
  int blocks = (int)((count + 256 * 16 - 1) / (256 * 16));
  blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
  hipLaunchKernelGGL(absmax_kernel, dim3(blocks), dim3(256), 0, st, cb, count, absmax_out);
  return check_launch();
}

int gq_scores_f32(const float *mu, const float *sd, const float *cb, float *out, int64_t dim,
                  int64_t rows, int64_t n, double beta, void *stream) {
  if (dim < 1 || rows < 0 || n < 1 || n > 0x7fffffff || rows > 0x7fffffff) return GQHIP_ERR_INVALID_ARG;
  if (rows == 0) return GQHIP_OK;
  if (!mu || !sd || !cb || !out) return GQHIP_ERR_INVALID_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int cpt = 1;   // codes per thread of gq_scores_kernel (CPT there)
  const unsigned gx = (unsigned)((n + 256 * cpt - 1) / (256 * cpt));
  constexpr int ROWS = 16;
  const unsigned gy = (unsigned)((rows + ROWS - 1) / ROWS);
  if (gy > 65535u * 32u) return GQHIP_ERR_INVALID_ARG;
#define GQ_SC(D)                                                                                          \
  do {                                                                                                    \
    if (beta == 1.0)                                                                                      \
      hipLaunchKernelGGL((gq_scores_kernel<D, ROWS, true>), dim3(gx, gy), dim3(256), 0, st, mu, sd, cb, out, \
                         (int)rows, (int)n, beta);                                                        \
    else                                                                                                  \
      hipLaunchKernelGGL((gq_scores_kernel<D, ROWS, false>), dim3(gx, gy), dim3(256), 0, st, mu, sd, cb, out, \
                         (int)rows, (int)n, beta);                                                        \
  } while (0)
  switch (dim) {
    case 4: GQ_SC(4); break;
    case 8: GQ_SC(8); break;
    case 16: GQ_SC(16); break;
    case 32: GQ_SC(32); break;
    default:
      hipLaunchKernelGGL(gq_scores_generic_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)rows), dim3(256), 0, st, mu, sd, cb,
                         out, (int)dim, (int)rows, (int)n, beta);
  }
#undef GQ_SC
  return check_launch();
}

int gq_argmax_f32(const float *mu, const float *sd, const float *logsd_or_null, const float *cb,
                  int64_t *idx, float *zhat_or_null, int64_t dim, int64_t rows, int64_t n, double beta,
                  float cb_absmax, void *workspace, int64_t workspace_bytes, void *stream) {
  OutMap om{};
  om.mode = 0;
  return run_argmax<kModeGQ>(mu, sd, logsd_or_null, cb, idx, zhat_or_null, dim, rows, n, beta, cb_absmax,
                             workspace, workspace_bytes, om, static_cast<hipStream_t>(stream));
}

int gq_quantize_z_f32(const float *z, const float *cb, int64_t *idx, float *zhat_or_null,
                      float *mu_out_or_null, float *sd_out_or_null, int64_t B, int64_t L, int64_t c,
                      int64_t dim, int64_t n, int layout, int grouping, double lv_min, double lv_max,
                      double beta, float cb_absmax, void *workspace, int64_t workspace_bytes,
                      void *stream) {
  if (!z || !cb || !idx || B < 0 || L < 1 || c < 1 || dim < 1 || dim > kMaxDim || c % dim != 0)
    return GQHIP_ERR_INVALID_ARG;
  if ((layout != GQHIP_LAYOUT_BCHW && layout != GQHIP_LAYOUT_BLC) ||
      (grouping != GQHIP_GROUP_STRIDED && grouping != GQHIP_GROUP_CONTIGUOUS))
    return GQHIP_ERR_INVALID_ARG;
  const int64_t K = c / dim, rows = B * L * K;
  if (rows == 0) return GQHIP_OK;
  if (rows > 0x3fffffff) return GQHIP_ERR_INVALID_ARG;
  const WsLayout w = ws_layout(rows, n, dim);
  if (!workspace || workspace_bytes < w.total) return GQHIP_ERR_WORKSPACE;
  char *ws = static_cast<char *>(workspace);
  hipStream_t st = static_cast<hipStream_t>(stream);
  PrepParams pp{};
  pp.z = z;
  pp.mu = mu_out_or_null ? mu_out_or_null : reinterpret_cast<float *>(ws + w.mu);
  pp.sd = sd_out_or_null ? sd_out_or_null : reinterpret_cast<float *>(ws + w.sd);
  pp.lsd = reinterpret_cast<float *>(ws + w.lsd);
  pp.rows = rows; pp.dim = (int)dim; pp.K = (int)K; pp.L = (int)L; pp.c = (int)c;
  pp.layout = layout; pp.grouping = grouping;
  pp.lv_min = (float)lv_min; pp.lv_max = (float)lv_max;
  hipLaunchKernelGGL(prep_kernel, dim3((unsigned)((rows * dim + 255) / 256)), dim3(256), 0, st, pp);
  int rc = check_launch();
  if (rc != GQHIP_OK) return rc;
  OutMap om{};
  om.mode = layout == GQHIP_LAYOUT_BCHW ? 1 : 2;
  om.K = (int)K; om.L = (int)L; om.c = (int)c; om.grouping = grouping;
  return run_argmax<kModeGQ>(pp.mu, pp.sd, pp.lsd, cb, idx, zhat_or_null, dim, rows, n, beta, cb_absmax,
                             workspace, workspace_bytes, om, st);
}

int gq_dequant_f32(const int64_t *idx, const float *cb, float *zhat, int64_t B, int64_t L, int64_t K,
                   int64_t dim, int64_t n, int layout, int grouping, void *stream) {
  if (!idx || !cb || !zhat || B < 0 || L < 1 || K < 1 || dim < 1 || n < 1) return GQHIP_ERR_INVALID_ARG;
  const int64_t rows = B * L * K;
  if (rows == 0) return GQHIP_OK;
  DequantParams dp{};
  dp.idx = idx; dp.cb = cb; dp.zhat = zhat; dp.rows = rows; dp.dim = (int)dim; dp.n = (int)n;
  dp.omap.mode = layout == GQHIP_LAYOUT_BCHW ? 1 : 2;
  dp.omap.K = (int)K; dp.omap.L = (int)L; dp.omap.c = (int)(K * dim); dp.omap.grouping = grouping;
  hipLaunchKernelGGL(dequant_kernel, dim3((unsigned)((rows * dim + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), dp);
  return check_launch();
}

int vq_argmin_f32(const float *z, const float *emb, int64_t *idx, float *zq_or_null, int64_t dim,
                  int64_t rows, int64_t n, float emb_absmax, void *workspace, int64_t workspace_bytes,
                  void *stream) {
  OutMap om{};
  om.mode = 0;
  return run_argmax<kModeVQ>(z, nullptr, nullptr, emb, idx, zq_or_null, dim, rows, n, 0.0, emb_absmax,
                             workspace, workspace_bytes, om, static_cast<hipStream_t>(stream));
}

int lfq_pack_f32(const float *x, int64_t *idx, float *q_or_null, int64_t rows, int64_t nbits,
                 void *stream) {
  if (!x || !idx || rows < 0 || nbits < 1 || nbits > 62) return GQHIP_ERR_INVALID_ARG;
  if (rows == 0) return GQHIP_OK;
  hipLaunchKernelGGL(lfq_pack_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, idx, q_or_null, (long)rows, (int)nbits);
  return check_launch();
}

int lfq_unpack_f32(const int64_t *idx, float *q, int64_t rows, int64_t nbits, void *stream) {
  if (!idx || !q || rows < 0 || nbits < 1 || nbits > 62) return GQHIP_ERR_INVALID_ARG;
  if (rows == 0) return GQHIP_OK;
  hipLaunchKernelGGL(lfq_unpack_kernel, dim3((unsigned)((rows * nbits + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), idx, q, (long)rows, (int)nbits);
  return check_launch();
}

int gn_silu_f32(const float *x, const float *gamma, const float *beta, const float *pre_bias_or_null, float *y,
                int64_t B, int64_t C, int64_t HW, int64_t groups, double eps, int apply_silu, int layout,
                double *stats_ws, void *stream) {
  if (B < 0 || C < 1 || HW < 1 || groups < 1 || C % groups != 0) return GQHIP_ERR_INVALID_ARG;
  if (B == 0) return GQHIP_OK;
  if (!x || !gamma || !beta || !y || !stats_ws) return GQHIP_ERR_INVALID_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t bg = B * groups, cpg = C / groups, chunk = cpg * HW;
  if (layout == GQHIP_LAYOUT_NHWC) {
    // thread <-> channel-quad mapping needs cpg % 4 == 0, (C/4) | 256, <= 64 groups
    if (cpg % 4 != 0 || 256 % (C / 4) != 0 || groups > 64) return GQHIP_ERR_INVALID_ARG;
    if (hipMemsetAsync(stats_ws, 0, sizeof(double) * 2 * bg, st) != hipSuccess) return check_launch();
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
  if (hipMemsetAsync(stats_ws, 0, sizeof(double) * 2 * bg, st) != hipSuccess) return check_launch();
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
                       int64_t HW, int64_t groups, double *stats_out, void *stream) {
  if (B < 0 || C < 1 || HW < 1 || groups < 1 || C % groups != 0) return GQHIP_ERR_INVALID_ARG;
  if (B == 0) return GQHIP_OK;
  if (!a || !b || !y || !stats_out) return GQHIP_ERR_INVALID_ARG;
  const int64_t cpg = C / groups;
  if (cpg % 4 != 0 || 256 % (C / 4) != 0 || groups > 64) return GQHIP_ERR_INVALID_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(stats_out, 0, sizeof(double) * 2 * B * groups, st) != hipSuccess) return check_launch();
  const int slabs = nhwc_slabs(C, HW);
  hipLaunchKernelGGL(add_bias_stats_nhwc_kernel, dim3((unsigned)(B * slabs)), dim3(256), 0, st, a, b, bias_or_null, y,
                     stats_out, (int)C, (long)HW, (int)cpg, slabs);
  return check_launch();
}

int gn_apply_f32(const float *x, const float *gamma, const float *beta, float *y, int64_t B, int64_t C, int64_t HW,
                 int64_t groups, double eps, int apply_silu, const double *stats, void *stream) {
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
  hipLaunchKernelGGL(wino_in_nhwc_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x, V,
                     (int)H, (int)W, (int)(C / 4), tiles, total);
  return check_launch();
}

int gn_stats_f32(const float *x, const float *pre_bias_or_null, int64_t B, int64_t C, int64_t HW, int64_t groups,
                 double *stats_out, void *stream) {
  if (B < 0 || C < 1 || HW < 1 || groups < 1 || C % groups != 0) return GQHIP_ERR_INVALID_ARG;
  if (B == 0) return GQHIP_OK;
  if (!x || !stats_out) return GQHIP_ERR_INVALID_ARG;
  const int64_t cpg = C / groups;
  if (cpg % 4 != 0 || 256 % (C / 4) != 0 || groups > 64) return GQHIP_ERR_INVALID_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(stats_out, 0, sizeof(double) * 2 * B * groups, st) != hipSuccess) return check_launch();
  const int slabs = nhwc_slabs(C, HW);
  hipLaunchKernelGGL(gn_stats_nhwc_kernel, dim3((unsigned)(B * slabs)), dim3(256), 0, st, x, pre_bias_or_null, stats_out,
                     (int)C, (long)HW, (int)cpg, slabs);
  return check_launch();
}

static int wino_in_gn_nhwc_f32_impl(int tile, const float *x, const float *gamma, const float *beta, const float *pre_bias_or_null,
                        const double *stats, float *V, int64_t B, int64_t H, int64_t W, int64_t C, int64_t groups,
                        double eps, int apply_silu, void *stream) {
  if (B < 0 || H < tile || W < tile || H % tile || W % tile || C < 4 || C % 4 != 0 || groups < 1 || C % groups != 0 ||
      (C / groups) % 4 != 0)
    return GQHIP_ERR_INVALID_ARG;
  if (B == 0) return GQHIP_OK;
  if (!x || !gamma || !beta || !stats || !V) return GQHIP_ERR_INVALID_ARG;
  const long tiles = (long)(B * (H / tile) * (W / tile)), total = tiles * (C / 4);
  long blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (tile == 4) {
    if (apply_silu)
      hipLaunchKernelGGL((wino4_in_gn_nhwc_kernel<1>), dim3((unsigned)blocks), dim3(256), 0, st, x, gamma, beta,
                         pre_bias_or_null, stats, V, (int)H, (int)W, (int)(C / 4), (int)(C / groups), eps, tiles, total);
    else
      hipLaunchKernelGGL((wino4_in_gn_nhwc_kernel<0>), dim3((unsigned)blocks), dim3(256), 0, st, x, gamma, beta,
                         pre_bias_or_null, stats, V, (int)H, (int)W, (int)(C / 4), (int)(C / groups), eps, tiles, total);
    return check_launch();
  }
  if (apply_silu)
    hipLaunchKernelGGL((wino_in_gn_nhwc_kernel<1>), dim3((unsigned)blocks), dim3(256), 0, st, x, gamma, beta,
                       pre_bias_or_null, stats, V, (int)H, (int)W, (int)(C / 4), (int)(C / groups), eps, tiles, total);
  else
    hipLaunchKernelGGL((wino_in_gn_nhwc_kernel<0>), dim3((unsigned)blocks), dim3(256), 0, st, x, gamma, beta,
                       pre_bias_or_null, stats, V, (int)H, (int)W, (int)(C / 4), (int)(C / groups), eps, tiles, total);
  return check_launch();
}

int wino_in_gn_nhwc_f32(const float *x, const float *gamma, const float *beta, const float *pre_bias_or_null,
                        const double *stats, float *V, int64_t B, int64_t H, int64_t W, int64_t C, int64_t groups,
                        double eps, int apply_silu, void *stream) {
  return wino_in_gn_nhwc_f32_impl(2, x, gamma, beta, pre_bias_or_null, stats, V, B, H, W, C, groups, eps, apply_silu, stream);
}

int wino4_in_gn_nhwc_f32(const float *x, const float *gamma, const float *beta, const float *pre_bias_or_null,
                         const double *stats, float *V, int64_t B, int64_t H, int64_t W, int64_t C, int64_t groups,
                         double eps, int apply_silu, void *stream) {
  return wino_in_gn_nhwc_f32_impl(4, x, gamma, beta, pre_bias_or_null, stats, V, B, H, W, C, groups, eps, apply_silu, stream);
}

int wino_out_nhwc_f32(const float *M, float *y, int64_t B, int64_t H, int64_t W, int64_t C, void *stream) {
  if (B < 0 || H < 2 || W < 2 || H % 2 || W % 2 || C < 4 || C % 4 != 0) return GQHIP_ERR_INVALID_ARG;
  if (B == 0) return GQHIP_OK;
  if (!M || !y) return GQHIP_ERR_INVALID_ARG;
  const long tiles = (long)(B * (H / 2) * (W / 2)), total = tiles * (C / 4);
  long blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(wino_out_nhwc_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), M, y,
                     (int)H, (int)W, (int)(C / 4), tiles, total);
  return check_launch();
}

int wino4_in_nhwc_f32(const float *x, float *V, int64_t B, int64_t H, int64_t W, int64_t C, void *stream) {
  if (B < 0 || H < 4 || W < 4 || H % 4 || W % 4 || C < 4 || C % 4 != 0) return GQHIP_ERR_INVALID_ARG;
  if (B == 0) return GQHIP_OK;
  if (!x || !V) return GQHIP_ERR_INVALID_ARG;
  const long tiles = (long)(B * (H / 4) * (W / 4)), total = tiles * (C / 4);
  long blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(wino4_in_nhwc_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x, V,
                     (int)H, (int)W, (int)(C / 4), tiles, total);
  return check_launch();
}

int wino4_out_nhwc_f32(const float *M, float *y, int64_t B, int64_t H, int64_t W, int64_t C, void *stream) {
  if (B < 0 || H < 4 || W < 4 || H % 4 || W % 4 || C < 4 || C % 4 != 0) return GQHIP_ERR_INVALID_ARG;
  if (B == 0) return GQHIP_OK;
  if (!M || !y) return GQHIP_ERR_INVALID_ARG;
  const long tiles = (long)(B * (H / 4) * (W / 4)), total = tiles * (C / 4);
  long blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(wino4_out_nhwc_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), M, y,
                     (int)H, (int)W, (int)(C / 4), tiles, total);
  return check_launch();
}

int wino_out_res_nhwc_f32(const float *M, const float *res, const float *bias_or_null, float *y, double *stats_out,
                          int64_t B, int64_t H, int64_t W, int64_t C, int64_t groups, int tile, void *stream) {
  if ((tile != 2 && tile != 4) || B < 0 || H < tile || W < tile || H % tile || W % tile || C < 4 || C % 4 != 0 ||
      groups < 1 || C % groups != 0)
    return GQHIP_ERR_INVALID_ARG;
  if (B == 0) return GQHIP_OK;
  if (!M || !y || !stats_out) return GQHIP_ERR_INVALID_ARG;   // res may be NULL: bias + statistics only
  const int64_t cpg = C / groups;
  if (cpg % 4 != 0 || 256 % (C / 4) != 0 || groups > 64) return GQHIP_ERR_INVALID_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(stats_out, 0, sizeof(double) * 2 * B * groups, st) != hipSuccess) return check_launch();
  const long tpi = (long)((H / tile) * (W / tile)), tiles = (long)B * tpi;
  const int lanes = (int)(256 / (C / 4));
  long slabs = (tpi + (long)lanes * 4 - 1) / ((long)lanes * 4);    // ~4 tiles per thread
  if (slabs > 1024) slabs = 1024;
  if (slabs < 1) slabs = 1;
  const dim3 grid((unsigned)(B * slabs));
  if (tile == 4)
    hipLaunchKernelGGL((wino_out_res_nhwc_kernel<4>), grid, dim3(256), 0, st, M, res, bias_or_null, y, stats_out, (int)H,
                       (int)W, (int)(C / 4), (int)cpg, tiles, (int)slabs);
  else
    hipLaunchKernelGGL((wino_out_res_nhwc_kernel<2>), grid, dim3(256), 0, st, M, res, bias_or_null, y, stats_out, (int)H,
                       (int)W, (int)(C / 4), (int)cpg, tiles, (int)slabs);
  return check_launch();
}

int upconv_im2col_nhwc_f32(const float *x, float *A, int64_t B, int64_t H, int64_t W, int64_t C, void *stream) {
  if (B < 0 || H < 1 || W < 1 || C < 4 || C % 4 != 0) return GQHIP_ERR_INVALID_ARG;
  if (B == 0) return GQHIP_OK;
  if (!x || !A) return GQHIP_ERR_INVALID_ARG;
  const long total = (long)(B * (H + 1) * (W + 1) * 4 * (C / 4));
  long blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(upconv_im2col_nhwc_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                     A, (int)H, (int)W, (int)(C / 4), total);
  return check_launch();
}

int upconv_shuffle_nhwc_f32(const float *src, float *y, int64_t B, int64_t H, int64_t W, int64_t C, void *stream) {
  if (B < 0 || H < 1 || W < 1 || C < 4 || C % 4 != 0) return GQHIP_ERR_INVALID_ARG;
  if (B == 0) return GQHIP_OK;
  if (!src || !y) return GQHIP_ERR_INVALID_ARG;
  const long total = (long)(B * 2 * H * 2 * W * (C / 4));
  long blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(upconv_shuffle_nhwc_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     src, y, (int)H, (int)W, (int)(C / 4), total);
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

static int fsq_levels(const int32_t *levels_host, int64_t nlev, FsqLevels *L) {
  if (!levels_host || nlev < 1 || nlev > 16) return GQHIP_ERR_INVALID_ARG;
  L->n = (int)nlev;
  long prod = 1;
  for (int i = 0; i < 16; ++i) L->lev[i] = 1;
  for (int i = 0; i < nlev; ++i) {
    if (levels_host[i] < 2) return GQHIP_ERR_INVALID_ARG;
    L->lev[i] = levels_host[i];
    prod *= levels_host[i];
    if (prod > 0x7fffffffL) return GQHIP_ERR_INVALID_ARG;   // the packed index is an int32
  }
  return GQHIP_OK;
}

int fsq_quantize_f32(const float *z, const int32_t *levels_host, int64_t nlev, float *zhat, int32_t *idx,
                     int64_t rows, void *stream) {
  FsqLevels L;
  if (!z || !idx || rows < 0 || fsq_levels(levels_host, nlev, &L) != GQHIP_OK) return GQHIP_ERR_INVALID_ARG;
  if (rows == 0) return GQHIP_OK;
  hipLaunchKernelGGL(fsq_quantize_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), z, L, zhat, idx, (long)rows);
  return check_launch();
}

int fsq_dequant_f32(const int32_t *idx, const int32_t *levels_host, int64_t nlev, float *zhat, int64_t rows,
                    void *stream) {
  FsqLevels L;
  if (!idx || !zhat || rows < 0 || fsq_levels(levels_host, nlev, &L) != GQHIP_OK) return GQHIP_ERR_INVALID_ARG;
  if (rows == 0) return GQHIP_OK;
  hipLaunchKernelGGL(fsq_dequant_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), idx, L, zhat, (long)rows);
  return check_launch();
}

int gq_index_histogram(const int64_t *idx, int64_t count, int64_t n, int32_t *hist, void *stream) {
  if (!idx || !hist || count < 0 || n < 1) return GQHIP_ERR_INVALID_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(hist, 0, sizeof(int32_t) * n, st) != hipSuccess) return check_launch();
  if (count == 0) return GQHIP_OK;
  int blocks = (int)((count + 255) / 256);
  blocks = blocks > 2048 ? 2048 : blocks;
  hipLaunchKernelGGL(hist_kernel, dim3(blocks), dim3(256), 0, st, idx, (long)count, (int)n, hist);
  return check_launch();
}

int gq_indices_to_u16(const int64_t *idx, uint16_t *out, int64_t count, void *stream) {
  if (!idx || !out || count < 0) return GQHIP_ERR_INVALID_ARG;
  if (count == 0) return GQHIP_OK;
  hipLaunchKernelGGL(to_u16_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), idx, out, (long)count);
  return check_launch();
}

int gq_indices_from_u16(const uint16_t *in, int64_t *idx, int64_t count, void *stream) {
  if (!in || !idx || count < 0) return GQHIP_ERR_INVALID_ARG;
  if (count == 0) return GQHIP_OK;
  hipLaunchKernelGGL(from_u16_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), in, idx, (long)count);
  return check_launch();
}

int gqhip_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_on = on != 0;
  return GQHIP_OK;
}

int gqhip_debug_enable(int on) {
  g_debug_stats = on != 0;
  return GQHIP_OK;
}

int gqhip_profile_collect(int *launches_host, double *total_ms_host) {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
  {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ev.swap(g_prof_events);
  }
  double total = 0.0;
  int cnt = 0;
  for (auto &pr : ev) {
    float ms = 0.f;
    if (hipEventSynchronize(pr.second) == hipSuccess &&
        hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
      total += ms;
      ++cnt;
    }
    (void)hipEventDestroy(pr.first);
    (void)hipEventDestroy(pr.second);
  }
  if (launches_host) *launches_host = cnt;
  if (total_ms_host) *total_ms_host = total;
  return GQHIP_OK;
}

int gqhip_debug_counters(const void *workspace, int64_t *fallback_rows_host,
                         int64_t *reranked_halftiles_host) {
  if (!workspace) return GQHIP_ERR_INVALID_ARG;
  WsHeader h;
  if (hipMemcpy(&h, workspace, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return check_launch();
  if (fallback_rows_host) *fallback_rows_host = h.fb_count;
  if (reranked_halftiles_host) *reranked_halftiles_host = (int64_t)h.reranked;
  return GQHIP_OK;
}

}  // extern "C"
