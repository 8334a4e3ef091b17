// gqhip.hip -- C-ABI entry points of libgqhip.so (see include/gqhip.h): the quantiser path (fused arg-max, compat score
// op, dequant, LFQ / FSQ, wire format) and the library-wide services.  The conv-stack entry points are in gqhip_unet.hip.
// gfx950 only; built by `make -C vq-vae-from-gaussian-vae_amd/csrc`.
#include "gqhip.h"

#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "gqhip_internal.h"
#include "gq_aux.h"
#include "gq_common.h"
#include "gq_filter.h"
#include "gq_filter_bf16.h"
#include "gq_grid.h"
#include "gq_prep.h"
#include "gq_rerank.h"
#include "gq_scores.h"
#include "gq_scores_f16.h"

using namespace gqhip;

namespace gqhip {
thread_local int g_last_hip_error = 0;

int check_launch() {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    g_last_hip_error = (int)e;
    return GQHIP_ERR_LAUNCH;
  }
  return GQHIP_OK;
}
}  // namespace gqhip

namespace {

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
  bool mixed;           // dim 16, Gaussian score: fp16 main product + fp8 corrections instead of three bf16 products
  bool f16;             // fp16 main product only + the data-dependent bound of the re-rank (round 3: the default filter)
};

// gq_filter_bf16.h / DESIGN.md section 3: 2 x 1057 (the two fp8 correction types: (2^-3 + 2^-8) relative on a term of at most
// 2^-11 (1 + 2^-11) |A s|) + 4 (dropped A_l s_l) + 1 (fp32 square) + 42 (operands in the fp8 / fp16 subnormal ranges: absolute
// errors, bounded against T for 1 <= max|cb| <= 16) + 4 * 32 + 2 * 64 (accumulation steps of the main product / of the
// corrections) = 2417, rounded up
constexpr float kMixedEfCoeff = 2450.0f;
constexpr float kMixedN1Limit = 16.0f;   // ... and max|cb| >= 1 (gq_rerank.h)
// fp16 main-product filter (gq_filter_bf16.h, F16): per product the two fp16 roundings, (1 + 2^-11)^2 - 1 = 16384 u + 4 u, the fp32
// roundings of A / B, of n^2 and of the row normalisation's inputs (3 u), 4 u per accumulation step of up to 2 x 64 slots
// ... charged for the widest layout (dim 32: 64 products): 16384 + 4 + 3 + 4 * 64 = 16647, rounded up.  Operands in fp16's
// subnormal range are charged separately, as an absolute term (gq_rerank.h:f16_bound).
constexpr float kF16EfCoeff = 16700.0f;
constexpr float kF16N1Limit = 255.0f;    // n^2 must stay a finite fp16

// Filter selection: 0 = auto (the fp16 main-product filter at every MFMA dim, GQ and VQ), 1 = always the fp32 MFMA filter,
// 2 = split-bf16 wherever it applies, 3 = fp16 + fp8 (round 2's default: dim 16 / Gaussian score, split-bf16 elsewhere).
// Initial value from GQHIP_FILTER=fp32|bf16, changed at run time by gqhip_set_filter().  Both filters feed the
// same exact re-rank, so the choice never changes an index.
std::atomic<int> g_filter_kind{[] {
  const char *e = getenv("GQHIP_FILTER");
  if (e && (e[0] == 'f' || e[0] == 'F') && (e[1] == 'p' || e[1] == 'P') && e[2] == '3') return 1;   // fp32
  if (e && (e[0] == 'b' || e[0] == 'B')) return 2;                                                   // bf16
  if (e && (e[0] == 'm' || e[0] == 'M')) return 3;                                                   // mixed = fp16 + fp8
  return 0;
}()};
bool want_bf16_filter() { return g_filter_kind.load(std::memory_order_relaxed) != 1; }
bool want_mixed_filter() { return g_filter_kind.load(std::memory_order_relaxed) == 3; }
bool want_f16_filter() { return g_filter_kind.load(std::memory_order_relaxed) == 0; }

// Grid search (gq_grid.h) instead of filter + re-rank: dims 4 / 8, filter selection AUTO, 2^14 <= n <= 2^20 codes, and the caller
// passed a codebook cache of gqhip_cb_cache_bytes().  GQHIP_GRID=0 disables it, =4 / =8 restricts it to one dim (A/B timing).
bool grid_dim_enabled(int64_t dim) {
  static const int env = getenv("GQHIP_GRID") ? atoi(getenv("GQHIP_GRID")) : 4;     // default: dim 4 only (dim 8 is slower than the
  return env == 48 ? (dim == 4 || dim == 8) : (env != 0 && dim == env);              // dense path today: =8 / =48 for experiments)
}
int64_t grid_cache_bytes(int64_t n, int64_t dim) {
  if (!grid_dim_enabled(dim) || n < 16384 || n > (1 << 20)) return 0;
  return grid_layout(n, dim).total;
}
bool grid_applies(int64_t n, int64_t dim, const void *cache, int64_t cache_bytes) {
  const int64_t need = grid_cache_bytes(n, dim);
  return need > 0 && cache && cache_bytes >= need && g_filter_kind.load(std::memory_order_relaxed) == 0;
}
// Dims 8 / 16 / 32 (the fp16 main-product filter): the codebook's fp16 operand image is kept in the cache, every 1/256 slice of it
// validated against -- and, when stale, rebuilt and restamped by -- the code block of the first launch that owns it (gq_prep.h).
// GQHIP_IMG_CACHE=0 disables it (A/B timing): the image is then rebuilt in the workspace on every call, as before round 5.
int64_t image_cache_bytes(int64_t n, int64_t dim);   // (needs make_plan)

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

  // Candidate granularity: GT tiles per half-group.  Behind the bf16 / fp16 filters 4 (64-code candidates): the tracker's top-4
  // insert (12 VALU) runs once per GT tiles, and VALU issue is what those loops are short of (measured at config 2 with the
  // split-bf16 filter: GT 1 162 us, 2 157, 4 153; GT 8: filter -2 us, re-rank +14 us); the re-rank's fp32 pre-filter makes their
  // 16 GT codes cheap to go through.
  // dim 4 keeps the packed split-bf16 filter: 65 536 codes are dense in 4-d -- 3.5 candidate groups per row inside the fp16
  // margin and a quarter of the rows undecided (measured, profiles/r03) -- and its kernel is not MFMA-bound in the first place
  pl.f16 = pl.bf16 && want_f16_filter() && pl.waves == 8 && dim != 4;
  pl.gt = pl.bf16 ? 4 : (dim <= 8 ? 4 : 2);
  // (Measured in round 4 and not kept: groups of 8 tiles at dim 8, where the tracker's 12 VALU per group weigh twice what they do at
  // dim 16 -- filter 78.9 -> 75.1 us, but the re-rank's 128-code candidates give it back: whole call 128.4 -> 130.2 us.)
  pl.tiles_per_split = (pl.tiles_per_split + pl.gt - 1) / pl.gt * pl.gt;   // a tile group never straddles two splits
  pl.nsplit = (pl.tiles_total + pl.tiles_per_split - 1) / pl.tiles_per_split;
  pl.mixed = pl.bf16 && want_mixed_filter() && dim == 16 && pl.waves == 8 && pl.ct == 16 && pl.gt == 4;
  if (pl.f16) pl.ct = dim == 32 ? 8 : 16;   // one 16-byte vector per MFMA, lane and tile: 16-tile chunks are 16 / 32 KiB
  if ((pl.mixed || pl.f16) && pl.nsplit > kMaxSplit / 2) {
    // the fp16 + fp8 filter leaves one record per lane half: 2 nsplit record sets for the re-rank (<= kMaxSplit)
    const int s3 = kMaxSplit / 2;
    pl.tiles_per_split = ((pl.tiles_total + s3 - 1) / s3 + pl.gt - 1) / pl.gt * pl.gt;
    pl.nsplit = (pl.tiles_total + pl.tiles_per_split - 1) / pl.tiles_per_split;
  }
  // The records carry half-group ids RELATIVE to their split in 16 bits (gq_common.h:Rec): 2 * tiles_per_split / GT <= 65536, i.e.
  // at most 2^20 GT codes per split.  More splits where that is not so (n > 2^22 at the bench's 8..16 splits); codebooks beyond what
  // the split cap allows (n > 2^26 GT / 2: 134 M codes at GT 4) leave the MFMA path for the exhaustive kernel.
  const int64_t max_tps = 32768LL * pl.gt;
  if (pl.mfma && pl.tiles_per_split > max_tps) {
    const int cap = (pl.mixed || pl.f16) ? kMaxSplit / 2 : kMaxSplit;
    const int64_t need = (pl.tiles_total + max_tps - 1) / max_tps;
    if (need > cap) {
      pl.mfma = pl.bf16 = pl.f16 = pl.mixed = false;
    } else {
      pl.tiles_per_split = (int)(((pl.tiles_total + need - 1) / need + pl.gt - 1) / pl.gt * pl.gt);
      pl.nsplit = (pl.tiles_total + pl.tiles_per_split - 1) / pl.tiles_per_split;
    }
  }
  return pl;
}

inline int64_t align256(int64_t v) { return (v + 255) / 256 * 256; }

int64_t image_cache_bytes(int64_t n, int64_t dim) {
  static const bool off = getenv("GQHIP_IMG_CACHE") && atoi(getenv("GQHIP_IMG_CACHE")) == 0;
  if (off || (dim != 8 && dim != 16 && dim != 32) || n < 1 || g_filter_kind.load(std::memory_order_relaxed) != 0) return 0;
  const Plan pl = make_plan(65536, n, dim);       // (what the image depends on -- tiles, chunk padding, vectors -- does not depend on rows)
  if (!pl.f16) return 0;
  return (int64_t)sizeof(GridHdr) + align256((int64_t)(pl.tiles_total + pl.ct) * (dim / 8) * 64 * 16);
}

struct WsLayout {
  int64_t hdr, rec, fb, dbg, mu, sd, lsd, rowsum, coef, cbimg, rowimg, rowscale, rowaux, kl2, total;
};

WsLayout ws_layout(int64_t rows, int64_t n, int64_t dim) {
  const Plan pl = make_plan(rows, n, dim);
  WsLayout w{};
  int64_t off = 0;
  w.hdr = off; off += (int64_t)sizeof(WsHeader);
  w.rec = off; off += align256((int64_t)sizeof(Rec) * rows * (pl.mfma ? pl.nsplit * ((pl.mixed || pl.f16) ? 2 : 1) : 0));
  w.fb = off;  off += align256(4 * rows);
  w.dbg = off; off += 128 * 1024;     // diagnostic builds only (GQHIP_CLOCK_STAMPS: per-block timeline records of the filter | of the re-rank)
  w.mu = off;  off += align256(4 * rows * dim);
  w.sd = off;  off += align256(4 * rows * dim);
  w.lsd = off; off += align256(4 * rows * dim);
  w.rowsum = off; off += pl.mfma ? align256(8 * 4 * rows) : 0;
  w.coef = off; off += pl.mfma ? align256(4 * 2 * rows * dim) : 0;
  // split-bf16 operand images: 2*NV vectors of 16 B per (code, half) / (row, half), NV = dim / 8
  const int64_t nvec = pl.f16 ? (dim == 4 ? 1 : dim / 8) : (dim == 4 ? 2 : dim / 4);
  w.cbimg = off;  off += pl.bf16 ? align256((int64_t)(pl.tiles_total + pl.ct) * nvec * 64 * 16) : 0;
  w.rowimg = off; off += pl.bf16 ? align256(rows * nvec * 2 * 16) : 0;
  w.rowscale = off; off += (pl.mixed || pl.f16) ? align256(4 * rows) : 0;
  w.rowaux = off; off += pl.f16 ? align256(32 * rows) : 0;
  w.kl2 = off; off += align256(4 * rows);       // per-row KL bits of gq_quantize_z_gauss_f32
  w.total = off;
  return w;
}

// ---- profiling recorder ------------------------------------------------------
std::mutex g_prof_mu;
bool g_prof_on = false;
int g_debug_stats = 0;
std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_events;   // attached to a dispatch, not yet collected
std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_pool;     // created ahead of time (gqhip_profile_reserve)

// When profiling is on, the filter is launched through hipExtLaunchKernelGGL with a start and a
// stop event attached to the dispatch itself, so the elapsed time is the kernel's own duration on
// its stream (what rocprofv3 --kernel-trace reports), not a marker-to-marker bracket.  The event pairs
// come from a pool filled by gqhip_profile_reserve(): nothing is created on the launch path, and a
// launch that finds the pool empty simply is not recorded.
struct ProfScope {
  hipEvent_t a = nullptr, b = nullptr;
  bool on = false;
  explicit ProfScope(bool enable = true) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (g_prof_on && enable && !g_prof_pool.empty()) {
      a = g_prof_pool.back().first;
      b = g_prof_pool.back().second;
      g_prof_pool.pop_back();
      on = true;
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

// re-rank: 16 lanes per row, 16 rows per block; NSI = record passes per lane (1 covers up to 16 record sets: every BASELINE shape)
template <int MODE>
int launch_rerank(const RerankParams &rp, int64_t rows, int dim, hipStream_t st, bool stats_block = false) {
  const dim3 grid((unsigned)((rows + 15) / 16 + (stats_block ? 1 : 0)));
#define GQ_RR(D, G)                                                                                  \
  do {                                                                                               \
    if (rp.nsplit <= 16) hipLaunchKernelGGL((gq_rerank_kernel<MODE, D, G, 1>), grid, dim3(256), 0, st, rp); \
    else hipLaunchKernelGGL((gq_rerank_kernel<MODE, D, G, 4>), grid, dim3(256), 0, st, rp);           \
  } while (0)
  switch (dim * 100 + rp.gt) {
    case 404: GQ_RR(4, 4); break;
    case 804: GQ_RR(8, 4); break;
    case 1602: GQ_RR(16, 2); break;
    case 1604: GQ_RR(16, 4); break;
    case 3202: GQ_RR(32, 2); break;
    case 3204: GQ_RR(32, 4); break;
    default: return GQHIP_ERR_INVALID_ARG;
  }
#undef GQ_RR
  return check_launch();
}

template <int MODE>
int launch_filter_bf16(const Plan &pl, const FilterBfParams &fp, int dim, bool mixed, hipStream_t st) {
  const dim3 grid((unsigned)(pl.row_blocks * pl.nsplit));
  const dim3 fblock((unsigned)(64 * pl.waves));
  ProfScope prof;
  if (pl.f16) {
#define GQ_LAUNCH_F16G(NV, R, C, G)                                                                                  \
  do {                                                                                                             \
    if (prof.on)                                                                                                   \
      hipExtLaunchKernelGGL((gq_filter_bf16_kernel<NV, R, C, G, 8, 2>), grid, fblock, 0, st, prof.a, prof.b, 0, fp); \
    else                                                                                                           \
      hipLaunchKernelGGL((gq_filter_bf16_kernel<NV, R, C, G, 8, 2>), grid, fblock, 0, st, fp);                      \
  } while (0)
#define GQ_LAUNCH_F16(NV, R, C) GQ_LAUNCH_F16G(NV, R, C, 4)
    switch (dim * 10 + pl.rt) {
      case 81: GQ_LAUNCH_F16(1, 1, 16); break;
      case 82: GQ_LAUNCH_F16(1, 2, 16); break;
      case 161: GQ_LAUNCH_F16(2, 1, 16); break;
      case 162: GQ_LAUNCH_F16(2, 2, 16); break;
      case 321: GQ_LAUNCH_F16(4, 1, 8); break;
      case 322: GQ_LAUNCH_F16(4, 2, 8); break;
      default: return GQHIP_ERR_INVALID_ARG;
    }
#undef GQ_LAUNCH_F16
#undef GQ_LAUNCH_F16G
    return check_launch();
  }
#define GQ_LAUNCH_BF1(NV, R, C, G, W)                                                                       \
  do {                                                                                                    \
    if (prof.on)                                                                                          \
      hipExtLaunchKernelGGL((gq_filter_bf16_kernel<NV, R, C, G, W>), grid, fblock, 0, st, prof.a, prof.b, 0, fp); \
    else                                                                                                  \
      hipLaunchKernelGGL((gq_filter_bf16_kernel<NV, R, C, G, W>), grid, fblock, 0, st, fp);                  \
  } while (0)
#define GQ_LAUNCH_BF(NV, R, C)                                                                            \
  do {                                                                                                    \
    if (pl.waves == 8) GQ_LAUNCH_BF1(NV, R, C, 4, 8);                                                     \
    else GQ_LAUNCH_BF1(NV, R, C, 4, 4);                                                                   \
  } while (0)
  if (pl.rt == 2) {
    switch (dim) {
      case 4: GQ_LAUNCH_BF(0, 2, 8); break;
      case 8: GQ_LAUNCH_BF(1, 2, 8); break;
      case 16:
        if (mixed) {
          if (prof.on)
            hipExtLaunchKernelGGL((gq_filter_bf16_kernel<2, 2, 16, 4, 8, 1>), grid, fblock, 0, st, prof.a, prof.b, 0, fp);
          else
            hipLaunchKernelGGL((gq_filter_bf16_kernel<2, 2, 16, 4, 8, 1>), grid, fblock, 0, st, fp);
        }
        else if (pl.ct == 16) GQ_LAUNCH_BF1(2, 2, 16, 4, 8);
        else GQ_LAUNCH_BF(2, 2, 8);
        break;
      default: GQ_LAUNCH_BF(4, 2, 4); break;
    }
  } else {
    switch (dim) {
      case 4: GQ_LAUNCH_BF(0, 1, 8); break;
      case 8: GQ_LAUNCH_BF(1, 1, 8); break;
      case 16:
        if (mixed) {
          if (prof.on)
            hipExtLaunchKernelGGL((gq_filter_bf16_kernel<2, 1, 16, 4, 8, 1>), grid, fblock, 0, st, prof.a, prof.b, 0, fp);
          else
            hipLaunchKernelGGL((gq_filter_bf16_kernel<2, 1, 16, 4, 8, 1>), grid, fblock, 0, st, fp);
        }
        else GQ_LAUNCH_BF(2, 1, 8);
        break;
      default: GQ_LAUNCH_BF(4, 1, 4); break;
    }
  }
#undef GQ_LAUNCH_BF
#undef GQ_LAUNCH_BF1
  return check_launch();
}

// What the first launch reads: either z in the module layout (FROM_Z) or ready-made rows.
struct PrepInput {
  const float *z = nullptr, *noise = nullptr;
  float *zhat_noquant = nullptr;
  float lv_min = 0.f, lv_max = 0.f;
  float *mu_out = nullptr, *sd_out = nullptr;   // FROM_Z: optional copies of the row operands for the caller
  float *sd_layout = nullptr;                   // FROM_Z: optional sd in the layout of zhat
  bool want_kl2 = false;                        // FROM_Z: leave the per-row KL bits in the workspace (ws_layout().kl2)
  int ste_kind = 0;                             // straight-through mix where zhat is stored (gq_common.h:WsHeader)
  const float *ste = nullptr;
  float *pure = nullptr;
  GaussStatsParams gs{};                        // want_kl2: the statistics block's parameters (kl2row is filled in by run_argmax)
  bool *stats_done = nullptr;                   // out: the statistics block ran inside the re-rank launch (no launch of its own needed)
};

template <int MODE, bool FROM_Z>
int launch_prep(const PrepParams &pp, int dim, bool mixed, bool f16, hipStream_t st) {
  const dim3 grid((unsigned)(pp.row_blocks + kPrepCodeBlocks));
  if constexpr (MODE == kModeGQ) {
    if (mixed && dim == 16) {
      hipLaunchKernelGGL((gq_prep_kernel<MODE, 16, FROM_Z, 1>), grid, dim3(256), 0, st, pp);
      return check_launch();
    }
  }
  if (f16) {
    switch (dim) {
      case 8: hipLaunchKernelGGL((gq_prep_kernel<MODE, 8, FROM_Z, 2>), grid, dim3(256), 0, st, pp); break;
      case 16: hipLaunchKernelGGL((gq_prep_kernel<MODE, 16, FROM_Z, 2>), grid, dim3(256), 0, st, pp); break;
      case 32: hipLaunchKernelGGL((gq_prep_kernel<MODE, 32, FROM_Z, 2>), grid, dim3(256), 0, st, pp); break;
      default: return GQHIP_ERR_INVALID_ARG;
    }
    return check_launch();
  }
  switch (dim) {
    case 4: hipLaunchKernelGGL((gq_prep_kernel<MODE, 4, FROM_Z>), grid, dim3(256), 0, st, pp); break;
    case 8: hipLaunchKernelGGL((gq_prep_kernel<MODE, 8, FROM_Z>), grid, dim3(256), 0, st, pp); break;
    case 16: hipLaunchKernelGGL((gq_prep_kernel<MODE, 16, FROM_Z>), grid, dim3(256), 0, st, pp); break;
    case 32: hipLaunchKernelGGL((gq_prep_kernel<MODE, 32, FROM_Z>), grid, dim3(256), 0, st, pp); break;
    default: return GQHIP_ERR_INVALID_ARG;
  }
  return check_launch();
}

// prep -> filter -> re-rank (MFMA dims) | prep -> exhaustive (other dims), shared by GQ and VQ.
// Three launches per call (round 4: the rows the candidates cannot decide are finished inside the re-rank, gq_rerank.h);
// nothing derived from the codebook or the rows survives the call.
template <int MODE>
int run_argmax(const PrepInput &in, const float *mu, const float *sd, const float *lsd, const float *cb, int64_t *idx,
               float *zhat, int64_t dim, int64_t rows, int64_t n, double beta, void *workspace,
               int64_t workspace_bytes, void *cb_cache, int64_t cb_cache_bytes, const OutMap &omap, hipStream_t st) {
  if (dim < 1 || dim > kMaxDim || rows < 0 || n < 1 || n > 0x3fffffff || rows > 0x3fffffff)
    return GQHIP_ERR_INVALID_ARG;
  if (rows == 0) return GQHIP_OK;   // empty batch: nothing to do (pointers may be NULL)
  const bool from_z = in.z != nullptr;
  if (!cb || !idx || (!from_z && (!mu || (MODE == kModeGQ && !sd)))) return GQHIP_ERR_INVALID_ARG;
  const WsLayout w = ws_layout(rows, n, dim);
  if (!workspace || workspace_bytes < w.total) return GQHIP_ERR_WORKSPACE;
  char *ws = static_cast<char *>(workspace);
  WsHeader *hdr = reinterpret_cast<WsHeader *>(ws + w.hdr);
  const Plan pl = make_plan(rows, n, dim);
  float *ws_mu = reinterpret_cast<float *>(ws + w.mu), *ws_sd = reinterpret_cast<float *>(ws + w.sd);
  float *ws_lsd = reinterpret_cast<float *>(ws + w.lsd);
  // the rows every later kernel reads
  const float *r_mu = from_z ? (in.mu_out ? in.mu_out : ws_mu) : mu;
  const float *r_sd = from_z ? (in.sd_out ? in.sd_out : ws_sd) : sd;
  const float *r_lsd = from_z ? ws_lsd : (MODE == kModeGQ ? (lsd ? lsd : (pl.mfma ? ws_lsd : nullptr)) : nullptr);

  RerankParams rp{};
  rp.mu = r_mu; rp.sd = MODE == kModeGQ ? r_sd : nullptr; rp.lsd = r_lsd; rp.cb = cb;
  rp.rowsum = reinterpret_cast<const double *>(ws + w.rowsum);
  rp.coef = reinterpret_cast<const float *>(ws + w.coef);
  rp.rec = reinterpret_cast<const Rec *>(ws + w.rec);
  rp.idx = idx; rp.zhat = zhat; rp.hdr = hdr;
  rp.fb_list = reinterpret_cast<int *>(ws + w.fb);
  rp.rows = (int)rows; rp.n = (int)n; rp.dim = (int)dim;
  const bool mixed = pl.mixed && MODE == kModeGQ;
  const bool f16 = pl.f16;
  rp.ef_coeff = pl.bf16 ? (float)(dim == 4 ? 332 : 220 + 24 * dim) : (float)(2 * dim + 4);
  if (mixed) { rp.ef_coeff = kMixedEfCoeff; rp.n1_limit = kMixedN1Limit; rp.n1_min = 1.0f; }
  if (f16) {
    rp.ef_coeff = kF16EfCoeff; rp.n1_limit = kF16N1Limit; rp.n1_min = 0.0f;
    rp.rowaux = reinterpret_cast<const float *>(ws + w.rowaux);
  }
  static const double env_ef = getenv("GQHIP_EF_COEFF") ? atof(getenv("GQHIP_EF_COEFF")) : 0.0;   // diagnostics
  if (env_ef > 0.0) rp.ef_coeff = (float)env_ef;
  rp.beta = (float)beta; rp.nsplit = (mixed || f16) ? 2 * pl.nsplit : pl.nsplit; rp.gt = pl.gt; rp.all_rows = pl.mfma ? 0 : 1; rp.stats = g_debug_stats;
  rp.omap = omap;
  rp.dbg = ws + w.dbg + 64 * 1024;
  rp.rec_halves = (mixed || f16) ? 2 : 1;
  rp.tiles_per_split = pl.tiles_per_split; rp.tiles_total = pl.tiles_total;

  if (!pl.mfma) {
    // dims outside {4, 8, 16, 32}: exact score of every code (gq_exhaustive_kernel)
    if (hipMemsetAsync(hdr, 0, sizeof(WsHeader), st) != hipSuccess) return check_launch();
    if (from_z) {
      PrepPlainParams pq{};
      pq.z = in.z; pq.noise = in.noise; pq.zhat_noquant = in.zhat_noquant;
      pq.sd_layout = in.sd_layout; pq.kl2row = in.want_kl2 ? reinterpret_cast<float *>(ws + w.kl2) : nullptr;
      pq.vq = MODE == kModeVQ ? 1 : 0;
      pq.hdr = hdr; pq.ste_kind = in.ste_kind; pq.ste = in.ste; pq.pure = in.pure;
      pq.mu = const_cast<float *>(r_mu); pq.sd = const_cast<float *>(r_sd); pq.lsd = ws_lsd;
      pq.rows = rows; pq.dim = (int)dim; pq.lv_min = in.lv_min; pq.lv_max = in.lv_max; pq.omap = omap;
      hipLaunchKernelGGL(prep_plain_kernel, dim3((unsigned)((rows * dim + 255) / 256)), dim3(256), 0, st, pq);
      const int rc = check_launch();
      if (rc != GQHIP_OK) return rc;
    }
    const int ex_blocks = (int)(rows < 4096 ? rows : 4096);
    hipLaunchKernelGGL((gq_exhaustive_kernel<MODE>), dim3((unsigned)ex_blocks), dim3(256), 0, st, rp);
    return check_launch();
  }

  // ---- launch 1: rows (+ zhat_noquant), operand images, bound sums, max|cb| partials, header -------------------
  PrepParams pp{};
  pp.z = in.z; pp.noise = in.noise; pp.zhat_noquant = in.zhat_noquant; pp.lv_min = in.lv_min; pp.lv_max = in.lv_max;
  pp.sd_layout = in.sd_layout; pp.kl2row = in.want_kl2 ? reinterpret_cast<float *>(ws + w.kl2) : nullptr;
  pp.ste_kind = in.ste_kind; pp.ste = in.ste; pp.pure = in.pure;
  const bool stats_in_rerank = MODE == kModeGQ && in.want_kl2 && !grid_applies(n, dim, cb_cache, cb_cache_bytes);
  if (stats_in_rerank) {
    pp.gs = in.gs;
    pp.gs.kl2row = pp.kl2row;
  }
  if (in.stats_done) *in.stats_done = stats_in_rerank;
  pp.mu = const_cast<float *>(r_mu); pp.sd = const_cast<float *>(r_sd);
  pp.lsd = const_cast<float *>(from_z ? ws_lsd : lsd);
  pp.lsd_out = (!from_z && MODE == kModeGQ && !lsd) ? ws_lsd : nullptr;
  pp.rowsum = reinterpret_cast<double *>(ws + w.rowsum);
  pp.coef = reinterpret_cast<float *>(ws + w.coef);
  pp.rowimg = pl.bf16 ? reinterpret_cast<u32x4 *>(ws + w.rowimg) : nullptr;
  pp.cb = cb;
  pp.cbimg = pl.bf16 ? reinterpret_cast<u32x4 *>(ws + w.cbimg) : nullptr;
  pp.hdr = hdr; pp.rows = rows; pp.n = (int)n; pp.tiles_total = pl.tiles_total;
  pp.row_blocks = (int)((rows * dim + 255) / 256);
  pp.beta = (float)beta; pp.omap = omap;
  pp.rowscale = (mixed || f16) ? reinterpret_cast<float *>(ws + w.rowscale) : nullptr;
  pp.rowaux = f16 ? reinterpret_cast<float *>(ws + w.rowaux) : nullptr;
  const bool grid = grid_applies(n, dim, cb_cache, cb_cache_bytes);
  if (grid) {
    // ---- dims 4 / 8: prep (rows, bound sums, max|cb|, codebook hash) -> index builder (exits unless the hash says the cache is
    //      stale) -> pruned exact search (gq_grid.h).  Three launches, no filter, no re-rank.
    GridHdr *gh = reinterpret_cast<GridHdr *>(cb_cache);
    pp.rowimg = nullptr; pp.cbimg = nullptr; pp.rowscale = nullptr; pp.rowaux = nullptr;
    pp.cache_sums = gh->blk_sum; pp.cache_stale = &gh->stale;
    int rc = from_z ? launch_prep<MODE, true>(pp, (int)dim, false, false, st) : launch_prep<MODE, false>(pp, (int)dim, false, false, st);
    if (rc != GQHIP_OK) return rc;
    GridBuildParams bp{};
    bp.cb = cb; bp.cache = static_cast<char *>(cb_cache); bp.hdr = hdr; bp.n = (int)n;
    if (dim == 4) hipLaunchKernelGGL((gq_grid_build_kernel<4>), dim3(1), dim3(kGridBuildThreads), 0, st, bp);
    else hipLaunchKernelGGL((gq_grid_build_kernel<8>), dim3(1), dim3(kGridBuildThreads), 0, st, bp);
    rc = check_launch();
    if (rc != GQHIP_OK) return rc;
    GridParams gp{};
    gp.mu = r_mu; gp.sd = r_sd; gp.lsd = r_lsd; gp.rowsum = pp.rowsum; gp.coef = pp.coef; gp.cb = cb;
    gp.cache = static_cast<const char *>(cb_cache);
    gp.idx = idx; gp.zhat = zhat; gp.hdr = hdr; gp.rows = (int)rows; gp.n = (int)n; gp.beta = (float)beta;
    static const int env_cap = getenv("GQHIP_GRID_CAP") ? atoi(getenv("GQHIP_GRID_CAP")) : 0;
    gp.leaf_cap = env_cap > 0 ? env_cap : 256;     // leaves a row may visit before it is handed to the scan (a flat score)
    static const int env_inwave = getenv("GQHIP_GRID_INWAVE") ? atoi(getenv("GQHIP_GRID_INWAVE")) : 0;   // diagnostics / tuning
    gp.inwave_cap = env_inwave > 0 ? env_inwave : kGridLeafCap;
    gp.stats = g_debug_stats; gp.omap = omap;
    static const int env_abl = getenv("GQHIP_GRID_ABL") ? atoi(getenv("GQHIP_GRID_ABL")) : 0;   // diagnostic builds (make abl) only
    gp.abl = env_abl;
    // the list of undecided rows lives in the (otherwise unused) candidate-record region: 16 B x rows x record sets >= 12 B x rows
    gp.und_row = reinterpret_cast<int *>(ws + w.rec);
    gp.und_thr = reinterpret_cast<float *>(ws + w.rec) + rows;
    gp.und_margin = reinterpret_cast<float *>(ws + w.rec) + 2 * rows;
    const int64_t nsets = (rows + 31) / 32;                        // (a block's eight waves fetch four rows at a time from a counter)
    static const int env_blocks = getenv("GQHIP_GRID_BLOCKS") ? atoi(getenv("GQHIP_GRID_BLOCKS")) : 0;   // diagnostics
    const int64_t max_blocks = env_blocks > 0 ? env_blocks : 512;
    const dim3 ggrid((unsigned)(nsets < max_blocks ? nsets : max_blocks));   // two 512-thread blocks per CU: one wave of blocks
    ProfScope prof;
#define GQ_GRID(D)                                                                                                   \
  do {                                                                                                               \
    if (prof.on) hipExtLaunchKernelGGL((gq_grid_kernel<MODE, D>), ggrid, dim3(kGridThreads), 0, st, prof.a, prof.b, 0, gp);    \
    else hipLaunchKernelGGL((gq_grid_kernel<MODE, D>), ggrid, dim3(kGridThreads), 0, st, gp);                                  \
  } while (0)
    if (dim == 4) GQ_GRID(4); else GQ_GRID(8);
#undef GQ_GRID
    rc = check_launch();
    if (rc != GQHIP_OK) return rc;
    // launch 4: the rows the search left undecided, a block per row (exits at once when there are none)
    static const int env_fblocks = getenv("GQHIP_FINISH_BLOCKS") ? atoi(getenv("GQHIP_FINISH_BLOCKS")) : 0;   // diagnostics
    const dim3 fgrid((unsigned)(env_fblocks > 0 ? env_fblocks : 256));
    if (dim == 4) hipLaunchKernelGGL((gq_grid_finish_kernel<MODE, 4>), fgrid, dim3(kGridThreads), 0, st, gp);
    else hipLaunchKernelGGL((gq_grid_finish_kernel<MODE, 8>), fgrid, dim3(kGridThreads), 0, st, gp);
    return check_launch();
  }
  if (f16 && cb_cache && image_cache_bytes(n, dim) > 0 && cb_cache_bytes >= image_cache_bytes(n, dim)) {
    GridHdr *ih = reinterpret_cast<GridHdr *>(cb_cache);          // (only its slice hashes are used: a slice is current iff its hash matches)
    pp.cbimg = reinterpret_cast<u32x4 *>(static_cast<char *>(cb_cache) + sizeof(GridHdr));
    pp.cache_sums = ih->blk_sum;
    pp.cache_stale = nullptr;
  }
  int rc = from_z ? launch_prep<MODE, true>(pp, (int)dim, mixed, f16, st) : launch_prep<MODE, false>(pp, (int)dim, mixed, f16, st);
  if (rc != GQHIP_OK) return rc;

  // ---- launch 2: the filter ---------------------------------------------------------------------------------
  if (pl.bf16) {
    FilterBfParams fp{};
    fp.cbimg = pp.cbimg; fp.rowimg = pp.rowimg;
    fp.rec = reinterpret_cast<Rec *>(ws + w.rec);
    fp.rows = (int)rows; fp.n = (int)n;
    fp.nsplit = pl.nsplit; fp.tiles_total = pl.tiles_total; fp.tiles_per_split = pl.tiles_per_split;
    fp.hdr = hdr;
    fp.dbg = ws + w.dbg;    // diagnostic builds only
    fp.rowscale = (mixed || f16) ? reinterpret_cast<const float *>(ws + w.rowscale) : nullptr;
    rc = launch_filter_bf16<MODE>(pl, fp, (int)dim, mixed, st);
    if (rc != GQHIP_OK) return rc;
  } else {
    FilterParams fp{};
    fp.mu = r_mu; fp.sd = r_sd; fp.cb = cb;
    fp.rec = reinterpret_cast<Rec *>(ws + w.rec);
    fp.rows = (int)rows; fp.n = (int)n; fp.beta = (float)beta;
    fp.nsplit = pl.nsplit; fp.tiles_total = pl.tiles_total; fp.tiles_per_split = pl.tiles_per_split;
    fp.hdr = hdr;
    fp.dbg = ws + w.dbg;      // diagnostic builds only
    rc = launch_filter<MODE>(pl, fp, (int)dim, st);
    if (rc != GQHIP_OK) return rc;
  }

  // ---- launch 3: exact re-rank of the candidates; rows they cannot decide are finished by their own block --------------
  return launch_rerank<MODE>(rp, rows, (int)dim, st, stats_in_rerank);
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
  if (kind != GQHIP_FILTER_AUTO && kind != GQHIP_FILTER_FP32 && kind != GQHIP_FILTER_BF16 && kind != GQHIP_FILTER_MIXED)
    return GQHIP_ERR_INVALID_ARG;
  g_filter_kind.store(kind, std::memory_order_relaxed);
  return GQHIP_OK;
}

int gqhip_get_filter(void) { return g_filter_kind.load(std::memory_order_relaxed); }

int gqhip_debug_plan(int64_t rows, int64_t n, int64_t dim, int64_t *out8) {
  if (!out8 || rows < 1 || n < 1 || dim < 1 || dim > kMaxDim) return GQHIP_ERR_INVALID_ARG;
  const Plan pl = make_plan(rows, n, dim);
  const WsLayout w = ws_layout(rows, n, dim);
  out8[0] = w.rec; out8[1] = pl.mfma ? pl.nsplit * ((pl.mixed || pl.f16) ? 2 : 1) : 0;   // record sets per row (fp16 + fp8: one per lane half)
  out8[2] = pl.gt; out8[3] = pl.tiles_per_split;
  out8[4] = pl.f16 ? 3 : (pl.mixed ? 2 : (pl.bf16 ? 1 : 0));   // 0 fp32 MFMA filter, 1 split-bf16, 2 fp16 + fp8 (Gaussian score), 3 fp16 main product
  out8[5] = pl.f16 ? (int)kF16EfCoeff : (pl.mixed ? (int)kMixedEfCoeff : (pl.bf16 ? (dim == 4 ? 332 : 220 + 24 * dim) : 2 * dim + 4));
  out8[6] = pl.rt; out8[7] = pl.waves;
  return GQHIP_OK;
}

int gqhip_grid_search_applies(int64_t n, int64_t dim) { return grid_cache_bytes(n, dim) > 0 && g_filter_kind.load(std::memory_order_relaxed) == 0; }

int gqhip_cb_cache_degenerate(const void *cb_cache, int64_t n, int64_t dim) {
  if (!cb_cache || grid_cache_bytes(n, dim) <= 0) return -1;
  GridHdr g;
  if (hipMemcpy(&g, cb_cache, sizeof(g), hipMemcpyDeviceToHost) != hipSuccess) { (void)check_launch(); return -1; }
  if (g.magic != kGridMagic || g.n != (int)n || g.dim != (int)dim || g.stale != 0) return -1;
  return g.max_sub > 255 ? 1 : 0;
}

int64_t gqhip_cb_cache_bytes(int64_t n, int64_t dim) {
  if (n < 1 || dim < 1 || dim > kMaxDim) return -1;
  const int64_t g = grid_cache_bytes(n, dim);
  return g > 0 ? g : image_cache_bytes(n, dim);
}

int64_t gqhip_workspace_bytes(int64_t rows, int64_t n, int64_t dim) {
  if (rows < 0 || n < 1 || dim < 1 || dim > kMaxDim) return -1;
  return ws_layout(rows < 1 ? 1 : rows, n, dim).total;
}

int gq_scores_f32(const float *mu, const float *sd, const float *cb, float *out, int64_t dim,
                  int64_t rows, int64_t n, double beta, void *stream) {
  if (dim < 1 || rows < 0 || n < 1 || n > 0x7fffffff || rows > 0x7fffffff) return GQHIP_ERR_INVALID_ARG;
  if (rows == 0) return GQHIP_OK;
  if (!mu || !sd || !cb || !out) return GQHIP_ERR_INVALID_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // Default: the score matrix on the matrix cores, HBM-write bound -- dims 16 / 32 as three fp16 products of two-term splits
  // (gq_scores_f16.h), dims 4 / 8 on the fp32 matrix cores (gq_scores.h; GQHIP_SCORES=f32: at every dim).  GQHIP_SCORES=direct:
  // the per-pair restatement of the CUDA kernel's formula (VALU bound, ~3x slower).  Non-finite beta, dims outside
  // {4, 8, 16, 32}: per-pair kernels.
  static const bool env_direct = getenv("GQHIP_SCORES") && getenv("GQHIP_SCORES")[0] == 'd';
  if (!env_direct && (dim == 4 || dim == 8 || dim == 16 || dim == 32) && beta == beta && n >= 32) {
    ScoresParams sp{};
    sp.mu = mu; sp.sd = sd; sp.cb = cb; sp.out = out; sp.rows = (int)rows; sp.n = (int)n; sp.beta = beta;
    sp.tiles_total = (int)((n + kTileCodes - 1) / kTileCodes);
    // diagnostics (tools/scores_sweep.sh): code splits per row block; bit 0 / 1 of the rotation (gq_scores.h), default both
    static const int env_nsplit = getenv("GQHIP_SCORES_NSPLIT") ? atoi(getenv("GQHIP_SCORES_NSPLIT")) : 0;
    static const int env_rot = getenv("GQHIP_SCORES_ROT") ? atoi(getenv("GQHIP_SCORES_ROT")) : 3;
    sp.rot = env_rot;
    constexpr int RT = 2;
    const int row_blocks = (int)((rows + 128 * RT - 1) / (128 * RT));
    int s = (512 + row_blocks - 1) / row_blocks;      // ~2 blocks per CU; splits in multiples of 8 (XCD = blockIdx % 8)
    s = ((s + 7) / 8) * 8;
    if (env_nsplit > 0) s = env_nsplit;
    if (s > sp.tiles_total) s = sp.tiles_total;
    if (s < 1) s = 1;
    sp.tiles_per_split = (sp.tiles_total + s - 1) / s;
    sp.nsplit = (sp.tiles_total + sp.tiles_per_split - 1) / sp.tiles_per_split;
    const dim3 grid((unsigned)(row_blocks * sp.nsplit));
    // dims 16 / 32: three fp16 products of two-term splits (gq_scores_f16.h) unless GQHIP_SCORES=f32
    static const bool env_f32 = getenv("GQHIP_SCORES") && getenv("GQHIP_SCORES")[0] == 'f';
    if (!env_f32 && dim == 16) {
      hipLaunchKernelGGL((gq_scores_f16x3_kernel<16, RT, 8>), grid, dim3(256), 0, st, sp);
      return check_launch();
    }
    if (!env_f32 && dim == 32) {
      hipLaunchKernelGGL((gq_scores_f16x3_kernel<32, RT, 4>), grid, dim3(256), 0, st, sp);
      return check_launch();
    }
    switch (dim) {
      case 4: hipLaunchKernelGGL((gq_scores_mfma_kernel<4, RT, 8>), grid, dim3(256), 0, st, sp); break;
      case 8: hipLaunchKernelGGL((gq_scores_mfma_kernel<8, RT, 8>), grid, dim3(256), 0, st, sp); break;
      case 16: hipLaunchKernelGGL((gq_scores_mfma_kernel<16, RT, 8>), grid, dim3(256), 0, st, sp); break;
      default: hipLaunchKernelGGL((gq_scores_mfma_kernel<32, RT, 4>), grid, dim3(256), 0, st, sp); break;
    }
    return check_launch();
  }
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
                  void *workspace, int64_t workspace_bytes, void *cb_cache_or_null, int64_t cb_cache_bytes, void *stream) {
  OutMap om{};
  om.mode = 0; om.K = 1; om.L = 1; om.c = (int)dim;
  return run_argmax<kModeGQ>(PrepInput{}, mu, sd, logsd_or_null, cb, idx, zhat_or_null, dim, rows, n, beta,
                             workspace, workspace_bytes, cb_cache_or_null, cb_cache_bytes, om, static_cast<hipStream_t>(stream));
}

int gq_quantize_z_f32(const float *z, const float *noise_or_null, const float *cb, int64_t *idx,
                      float *zhat_or_null, float *zhat_noquant_or_null, float *mu_out_or_null,
                      float *sd_out_or_null, int64_t B, int64_t L, int64_t c, int64_t dim, int64_t n, int layout,
                      int grouping, double lv_min, double lv_max, double beta, void *workspace,
                      int64_t workspace_bytes, void *cb_cache_or_null, int64_t cb_cache_bytes, void *stream) {
  if (!z || !cb || !idx || B < 0 || L < 1 || c < 1 || dim < 1 || dim > kMaxDim || c % dim != 0)
    return GQHIP_ERR_INVALID_ARG;
  if ((layout != GQHIP_LAYOUT_BCHW && layout != GQHIP_LAYOUT_BLC) ||
      (grouping != GQHIP_GROUP_STRIDED && grouping != GQHIP_GROUP_CONTIGUOUS))
    return GQHIP_ERR_INVALID_ARG;
  if (zhat_noquant_or_null && !noise_or_null) return GQHIP_ERR_INVALID_ARG;
  const int64_t K = c / dim, rows = B * L * K;
  if (rows == 0) return GQHIP_OK;
  if (rows > 0x3fffffff) return GQHIP_ERR_INVALID_ARG;
  PrepInput in;
  in.z = z; in.noise = noise_or_null; in.zhat_noquant = zhat_noquant_or_null;
  in.lv_min = (float)lv_min; in.lv_max = (float)lv_max;
  in.mu_out = mu_out_or_null; in.sd_out = sd_out_or_null;
  OutMap om{};
  om.mode = layout == GQHIP_LAYOUT_BCHW ? 1 : 2;
  om.K = (int)K; om.L = (int)L; om.c = (int)c; om.grouping = grouping;
  return run_argmax<kModeGQ>(in, nullptr, nullptr, nullptr, cb, idx, zhat_or_null, dim, rows, n, beta, workspace,
                             workspace_bytes, cb_cache_or_null, cb_cache_bytes, om, static_cast<hipStream_t>(stream));
}

int gq_quantize_z_gauss_f32(const float *z, const float *noise, const float *cb, int64_t *idx, float *zhat,
                            float *zhat_quant_or_null, float *zhat_noquant, float *sd_out_or_null, void *scalars_out, double *lam_state, int64_t B,
                            int64_t L, int64_t c, int64_t dim, int64_t n, int layout, int grouping, double lv_min,
                            double lv_max, double beta, int use_ste, double log2n, double tolerance, double lam_factor,
                            double lam_lo, double lam_hi, int lam_max_decreases, void *workspace, int64_t workspace_bytes,
                            void *cb_cache_or_null, int64_t cb_cache_bytes, void *stream) {
  if (!z || !noise || !cb || !idx || !zhat || !zhat_noquant || !scalars_out || !lam_state || B < 0 || L < 1 || c < 1 ||
      dim < 1 || dim > kMaxDim || c % dim != 0)
    return GQHIP_ERR_INVALID_ARG;
  if ((layout != GQHIP_LAYOUT_BCHW && layout != GQHIP_LAYOUT_BLC) ||
      (grouping != GQHIP_GROUP_STRIDED && grouping != GQHIP_GROUP_CONTIGUOUS))
    return GQHIP_ERR_INVALID_ARG;
  if ((reinterpret_cast<uintptr_t>(scalars_out) & 7u) || (reinterpret_cast<uintptr_t>(lam_state) & 7u)) return GQHIP_ERR_INVALID_ARG;
  const int64_t K = c / dim, rows = B * L * K;
  if (rows == 0) return GQHIP_OK;      // (the reference's means of an empty tensor are NaN; nothing is written here)
  if (rows > 0x3fffffff) return GQHIP_ERR_INVALID_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  PrepInput in;
  in.z = z; in.noise = noise; in.zhat_noquant = zhat_noquant;
  in.lv_min = (float)lv_min; in.lv_max = (float)lv_max;
  in.sd_layout = sd_out_or_null; in.want_kl2 = true;
  OutMap om{};
  om.mode = layout == GQHIP_LAYOUT_BCHW ? 1 : 2;
  om.K = (int)K; om.L = (int)L; om.c = (int)c; om.grouping = grouping;
  if (use_ste) { in.ste_kind = 1; in.ste = zhat_noquant; in.pure = zhat_quant_or_null; }
  GaussStatsParams gp{};
  gp.rows = (long)rows; gp.lam_state = lam_state; gp.scalars = scalars_out;
  gp.thr_hi = (float)(log2n + tolerance); gp.thr_lo = (float)(log2n - tolerance); gp.log2n = (float)log2n;
  gp.lam_factor = lam_factor; gp.lam_lo = lam_lo; gp.lam_hi = lam_hi; gp.lam_max_decreases = lam_max_decreases;
  bool stats_done = false;
  in.gs = gp; in.stats_done = &stats_done;
  int rc = run_argmax<kModeGQ>(in, nullptr, nullptr, nullptr, cb, idx, zhat, dim, rows, n, beta, workspace, workspace_bytes,
                               cb_cache_or_null, cb_cache_bytes, om, st);
  if (rc != GQHIP_OK || stats_done) return rc;       // dims 8 / 16 / 32: the statistics block ran as one extra block of the re-rank launch
  gp.kl2row = reinterpret_cast<const float *>(static_cast<char *>(workspace) + ws_layout(rows, n, dim).kl2);
  hipLaunchKernelGGL(gauss_stats_finalize_kernel, dim3(1), dim3(256), 0, st, gp);
  return check_launch();
}

int vq_quantize_z_f32(const float *z, const float *emb, int64_t *idx, float *zq, float *loss2_or_null, int64_t B, int64_t L,
                      int64_t c, int64_t dim, int64_t n, int layout, double beta, int legacy, void *workspace,
                      int64_t workspace_bytes, void *cb_cache_or_null, int64_t cb_cache_bytes, void *stream) {
  if (!z || !emb || !idx || !zq || B < 0 || L < 1 || c < 1 || dim < 1 || dim > kMaxDim || c % dim != 0)
    return GQHIP_ERR_INVALID_ARG;
  if (layout != GQHIP_LAYOUT_BCHW && layout != GQHIP_LAYOUT_BLC) return GQHIP_ERR_INVALID_ARG;
  const int64_t K = c / dim, rows = B * L * K;
  if (rows == 0) return GQHIP_OK;
  if (rows > 0x3fffffff) return GQHIP_ERR_INVALID_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  PrepInput in;
  in.z = z;
  OutMap om{};
  om.mode = layout == GQHIP_LAYOUT_BCHW ? 1 : 2;
  om.K = (int)K; om.L = (int)L; om.c = (int)c; om.grouping = GQHIP_GROUP_STRIDED;     // channel = d * K + k (vq.py:53)
  in.ste_kind = 2; in.ste = z;                                                        // z_q = z + (z_q - z) (vq.py:89)
  int rc = run_argmax<kModeVQ>(in, nullptr, nullptr, nullptr, emb, idx, zq, dim, rows, n, 0.0, workspace, workspace_bytes,
                               cb_cache_or_null, cb_cache_bytes, om, st);
  if (rc != GQHIP_OK || !loss2_or_null) return rc;
  const WsLayout w = ws_layout(rows, n, dim);
  VqLossParams lp{};
  lp.zrows = reinterpret_cast<const float *>(static_cast<char *>(workspace) + w.mu);
  lp.idx = idx; lp.emb = emb; lp.loss = loss2_or_null;
  lp.hdr = reinterpret_cast<WsHeader *>(static_cast<char *>(workspace) + w.hdr);
  lp.rows = (long)rows; lp.dim = (int)dim; lp.n = (int)n; lp.beta = (float)beta; lp.legacy = legacy; lp.omap = om;
  int64_t blocks = (rows * dim + 256 * 16 - 1) / (256 * 16);
  blocks = blocks < 1 ? 1 : (blocks > 256 ? 256 : blocks);
  hipLaunchKernelGGL(vq_loss_kernel, dim3((unsigned)blocks), dim3(256), 0, st, lp);
  return check_launch();
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
                  int64_t rows, int64_t n, void *workspace, int64_t workspace_bytes, void *cb_cache_or_null,
                  int64_t cb_cache_bytes, void *stream) {
  OutMap om{};
  om.mode = 0; om.K = 1; om.L = 1; om.c = (int)dim;
  return run_argmax<kModeVQ>(PrepInput{}, z, nullptr, nullptr, emb, idx, zq_or_null, dim, rows, n, 0.0, workspace,
                             workspace_bytes, cb_cache_or_null, cb_cache_bytes, om, static_cast<hipStream_t>(stream));
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

int64_t gq_step_record_workspace_bytes(int64_t B, int64_t per_image) {
  if (B < 0 || per_image < 1) return -1;
  const int64_t chunks = (per_image + kPsnrChunk - 1) / kPsnrChunk;
  return B * chunks * 8 + ((B * 4 + 7) / 8) * 8;
}

int gq_step_record_f32(const float *x, const float *x_rec, const int64_t *idx, int32_t *rec, int64_t B, int64_t per_image,
                       int64_t n_idx, void *workspace_zeroed, int64_t workspace_bytes, void *stream) {
  if (B < 0 || per_image < 1 || n_idx < 0 || B > 0x3fffffff) return GQHIP_ERR_INVALID_ARG;
  if (B == 0 && n_idx == 0) return GQHIP_OK;
  if (!rec || (B > 0 && (!x || !x_rec)) || (n_idx > 0 && !idx)) return GQHIP_ERR_INVALID_ARG;
  if (B > 0 && (!workspace_zeroed || workspace_bytes < gq_step_record_workspace_bytes(B, per_image))) return GQHIP_ERR_WORKSPACE;
  StepRecordParams p{};
  p.x = x; p.x_rec = x_rec; p.idx = idx; p.rec = rec;
  p.per_image = (long)per_image; p.n_idx = (long)n_idx; p.B = (int)B;
  p.chunks = (int)((per_image + kPsnrChunk - 1) / kPsnrChunk);
  if ((int64_t)p.chunks * B > 0x3fffffff) return GQHIP_ERR_INVALID_ARG;
  p.psnr_blocks = p.chunks * (int)B;
  p.partial = static_cast<double *>(workspace_zeroed);
  p.ticket = reinterpret_cast<int *>(static_cast<char *>(workspace_zeroed) + B * p.chunks * 8);
  const int64_t words = (n_idx + 1) / 2;
  const int64_t blocks = p.psnr_blocks + (words + 255) / 256;
  hipLaunchKernelGGL(step_record_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), p);
  return check_launch();
}

int gqhip_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_on = on != 0;
  if (g_prof_on && g_prof_pool.empty()) {   // callers that never reserve still get a (small) pool
    for (int i = 0; i < 64; ++i) {
      hipEvent_t a, b;
      if (hipEventCreate(&a) != hipSuccess) break;
      if (hipEventCreate(&b) != hipSuccess) { (void)hipEventDestroy(a); break; }
      g_prof_pool.emplace_back(a, b);
    }
  }
  return GQHIP_OK;
}

int gqhip_profile_reserve(int pairs) {
  if (pairs < 0) return GQHIP_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  while ((int)g_prof_pool.size() < pairs) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess) return check_launch();
    if (hipEventCreate(&b) != hipSuccess) { (void)hipEventDestroy(a); return check_launch(); }
    g_prof_pool.emplace_back(a, b);
  }
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
  }
  {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto &pr : ev) g_prof_pool.push_back(pr);   // recycled: the next profiled region creates nothing
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

int gqhip_debug_grid(const void *workspace, const void *cb_cache, int64_t *out4_host) {
  if (!workspace || !out4_host) return GQHIP_ERR_INVALID_ARG;
  WsHeader h;
  if (hipMemcpy(&h, workspace, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return check_launch();
  out4_host[0] = (int64_t)h.grid_leaves;
  out4_host[1] = (int64_t)h.reranked;
  out4_host[2] = h.fb_count;
  out4_host[3] = -1;
  if (cb_cache) {
    GridHdr g;
    if (hipMemcpy(&g, cb_cache, sizeof(g), hipMemcpyDeviceToHost) != hipSuccess) return check_launch();
    out4_host[3] = (g.magic == kGridMagic && g.stale == 0) ? 1 : 0;
  }
  return GQHIP_OK;
}

}  // extern "C"
