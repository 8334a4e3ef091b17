// Calibration: what write rate does THIS device sustain for the compat op's output pattern?  The score matrix is
// rows x n fp32 (16 384 x 65 536: rows 256 KiB apart); an MFMA tile hands a wave TR rows x a few hundred bytes at a time, so the
// stream a wave emits is "a short run in each of TR rows", not the linear sweep a fill makes (6.9 TB/s).  This program emits
// pure stores (no loads, no matrix work) in parameterised patterns:
//   TR    rows per wave tile            TC   bytes per row and tile        NT  tiles a wave walks along its rows
//   vec   1: one dword per lane (256-byte run per instruction), 4: dwordx4 (TC >= 1024: 1 KiB run of one row; else 4 rows x 256 B)
//   order 0: rows inner (all rows of a 256-byte / 1 KiB column piece, then the next piece)   1: columns inner
//   rot   1: row block b starts at tile (5 b) % NT of its split and wraps
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/store_pattern tools/store_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct P {
  float *out;
  long row_bytes;
  int rows, TR, TC, NT, vec, order, rot, nsplit;
};

// Store cache policies (round 4): 0 plain, 1 __builtin_nontemporal_store (the `nt` bit), 2 `sc0 sc1` (system-scope write-through),
// 3 `nt sc0 sc1`.  A write-once 4.29 GB stream through a write-back L2 + the 256 MB Infinity Cache is what these bits exist for.
template <int POL>
__device__ __forceinline__ void st4(f32x4 *p, f32x4 v) {
  if constexpr (POL == 0) *p = v;
  else if constexpr (POL == 1) __builtin_nontemporal_store(v, p);
  else if constexpr (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
}

// "Code-stationary" pattern: block b owns the column segment [b W, (b + 1) W) bytes of EVERY row and walks the rows in order,
// wave w of the block writing rows rt * 4 TR + w TR .. + TR of its segment in 1 KiB (or W-byte) runs.  All blocks advance through the
// rows together, so at any time the chip writes a window of a few dozen complete rows -- a linear sweep through memory, like a
// fill -- instead of 64 rows x 256 B per wave scattered over the whole matrix.
struct Q {
  float *out;
  long row_bytes;
  int rows, TR, W;
};
template <int POL>
__global__ __launch_bounds__(256) void colstat_kernel(const Q q) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char *seg = reinterpret_cast<char *>(q.out) + (long)blockIdx.x * q.W;
  const f32x4 vv = {(float)lane, 1.f, 2.f, 3.f};
  const int lanes_per_row = q.W >= 1024 ? 64 : q.W / 16, per = 64 / lanes_per_row;    // rows per store instruction
  for (int r0 = wave * q.TR; r0 < q.rows; r0 += 4 * q.TR) {
    for (int r = 0; r < q.TR; r += per) {
      char *rowp = seg + (long)(r0 + r + lane / lanes_per_row) * q.row_bytes + (lane % lanes_per_row) * 16;
      for (int c = 0; c < q.W; c += 1024) st4<POL>(reinterpret_cast<f32x4 *>(rowp + c), vv);
    }
  }
}

// Plain linear fill, many short blocks (what a torch fill_ launches): block b writes bytes [b BB, (b + 1) BB).
template <int POL>
__global__ __launch_bounds__(256) void linear_kernel(float *out, int BB) {
  char *base = reinterpret_cast<char *>(out) + (long)blockIdx.x * BB + threadIdx.x * 16;
  const f32x4 vv = {1.f, 2.f, 3.f, 4.f};
  for (int c = 0; c < BB; c += 4096) st4<POL>(reinterpret_cast<f32x4 *>(base + c), vv);
}

// Page-placement probe (round 4): every block iteration writes ONE 4 KiB page (256 threads x 16 B).  Which page:
//   mode 0: page = b + G it                    -- block b keeps the class b % 8 of its pages (blocks are dealt round-robin over the
//                                                 8 XCDs: the XCD of block b writes pages = b (mod 8) only)
//   mode 1: as 0 with the class rotated by s:  page = (b & ~7 | (b + s) & 7) + G it
//   mode 2: page = b iters + it                -- a block writes a contiguous run of pages (every class in turn)
//   mode 3: as 0, but the class is taken from the hardware XCC id instead of b % 8 (slot = b / 8)
template <int POL>
__global__ __launch_bounds__(256) void paged_kernel(float *out, int G, int iters, int mode, int s) {
  const int b = blockIdx.x;
  const f32x4 vv = {1.f, 2.f, 3.f, 4.f};
  int cls = b & 7;
  if (mode == 3) cls = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;      // HW_REG_XCC_ID, bits [3:0]
  for (int it = 0; it < iters; ++it) {
    long page;
    if (mode == 0) page = (long)b + (long)G * it;
    else if (mode == 1) page = (long)((b & ~7) | ((b + s) & 7)) + (long)G * it;
    else if (mode == 2) page = (long)b * iters + it;
    else page = (long)((b & ~7) | cls) + (long)G * it;
    st4<POL>(reinterpret_cast<f32x4 *>(reinterpret_cast<char *>(out) + page * 4096 + threadIdx.x * 16), vv);
  }
}

__global__ __launch_bounds__(256) void store_kernel(const P p) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int split = blockIdx.x % p.nsplit, rowblk = blockIdx.x / p.nsplit;
  const long row0 = ((long)rowblk * 4 + wave) * p.TR;
  if (row0 >= p.rows) return;
  char *base = reinterpret_cast<char *>(p.out) + row0 * p.row_bytes + (long)split * p.NT * p.TC;
  const int start = p.rot ? (int)((unsigned)rowblk * 5u % (unsigned)p.NT) : 0;
  const float v = (float)lane;
  for (int t0 = 0; t0 < p.NT; ++t0) {
    int t = t0 + start;
    t = t >= p.NT ? t - p.NT : t;
    char *tb = base + (long)t * p.TC;
    if (p.vec == 1) {
      const int pieces = p.TC / 256;
      if (p.order == 0) {
        for (int c = 0; c < pieces; ++c)
          for (int r = 0; r < p.TR; ++r) *reinterpret_cast<float *>(tb + r * p.row_bytes + c * 256 + lane * 4) = v;
      } else {
        for (int r = 0; r < p.TR; ++r)
          for (int c = 0; c < pieces; ++c) *reinterpret_cast<float *>(tb + r * p.row_bytes + c * 256 + lane * 4) = v;
      }
    } else if (p.TC >= 1024) {
      const int pieces = p.TC / 1024;
      const f32x4 vv = {v, v, v, v};
      if (p.order == 0) {
        for (int c = 0; c < pieces; ++c)
          for (int r = 0; r < p.TR; ++r) *reinterpret_cast<f32x4 *>(tb + r * p.row_bytes + c * 1024 + lane * 16) = vv;
      } else {
        for (int r = 0; r < p.TR; ++r)
          for (int c = 0; c < pieces; ++c) *reinterpret_cast<f32x4 *>(tb + r * p.row_bytes + c * 1024 + lane * 16) = vv;
      }
    } else {
      const int per = 1024 / p.TC;         // rows per instruction
      const int lanes_per_row = 64 / per;
      const f32x4 vv = {v, v, v, v};
      for (int r = 0; r < p.TR; r += per)
        *reinterpret_cast<f32x4 *>(tb + (r + lane / lanes_per_row) * p.row_bytes + (lane % lanes_per_row) * 16) = vv;
    }
  }
}

int main(int argc, char **argv) {
  const int rows = 16384;
  const long row_bytes = 65536L * 4;
  float *out;
  if (hipMalloc(&out, rows * row_bytes) != hipSuccess) return 1;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  // plain fill for reference
  {
    hipMemsetAsync(out, 0, rows * row_bytes, 0);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 5; ++i) hipMemsetAsync(out, 0, rows * row_bytes, 0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("hipMemsetAsync: %.3f ms -> %.0f GB/s\n", ms / 5, rows * row_bytes / (ms / 5) / 1e6);
  }
  struct Cfg { int TR, TC, span, vec, order, rot; };   // span: bytes of a row one wave walks
  const Cfg cfgs[] = {
      {64, 256, 32768, 1, 0, 0},    // the shipped kernel: 64 rows x 256 B, 8 splits
      {64, 256, 32768, 1, 0, 1},
      {32, 256, 32768, 1, 0, 1},
      {64, 256, 32768, 4, 0, 1},    // WIDE
      {32, 512, 32768, 1, 0, 1},    // two pieces, rows inner
      {32, 512, 32768, 1, 1, 1},    // two pieces, columns inner
      {32, 1024, 32768, 1, 0, 1},
      {32, 1024, 32768, 1, 1, 1},
      {32, 1024, 32768, 4, 0, 1},   // 1 KiB runs in one instruction
      {16, 1024, 32768, 4, 0, 1},
      {32, 2048, 32768, 4, 1, 1},
      {32, 4096, 32768, 4, 1, 1},
      {16, 4096, 32768, 4, 1, 1},
      {8, 4096, 32768, 4, 1, 1},
      {32, 1024, 8192, 4, 0, 1},
      {32, 1024, 262144, 4, 0, 0},  // one wave walks whole rows
      {32, 1024, 1024, 4, 0, 0},    // one tile per wave (256 splits)
      {64, 256, 1024, 1, 0, 0},
      {4, 262144, 262144, 4, 1, 0}, // whole rows, columns inner: nearly linear
      {1, 262144, 262144, 4, 1, 0},
  };
  for (const Cfg &c : cfgs) {
    P p{};
    p.out = out; p.row_bytes = row_bytes; p.rows = rows;
    p.TR = c.TR; p.TC = c.TC; p.NT = c.span / c.TC; p.vec = c.vec; p.order = c.order; p.rot = c.rot;
    p.nsplit = (int)(row_bytes / c.span);
    const int row_blocks = rows / (4 * c.TR);
    const unsigned grid = (unsigned)(row_blocks * p.nsplit);
    hipLaunchKernelGGL(store_kernel, dim3(grid), dim3(256), 0, 0, p);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(store_kernel, dim3(grid), dim3(256), 0, 0, p);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("TR %3d  TC %6d  span %6d  vec %d  order %d  rot %d  grid %5u: %.3f ms -> %.0f GB/s\n", c.TR, c.TC, c.span, c.vec, c.order,
           c.rot, grid, ms / 5, rows * row_bytes / (ms / 5) / 1e6);
  }
  // ---- round 4: linear fills and the code-stationary pattern, every store policy ----
  auto timeit = [&](auto launch, const char *label) {
    launch();
    hipEventRecord(e0, 0);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%s: %.3f ms -> %.0f GB/s\n", label, ms / 5, rows * row_bytes / (ms / 5) / 1e6);
  };
  const long total = rows * row_bytes;
  char label[256];
  for (int BB : {4096, 16384, 65536, 1 << 20}) {
    const unsigned grid = (unsigned)(total / BB);
#define LIN(POL)                                                                                          \
    snprintf(label, sizeof label, "linear fill, %7d B per block, grid %7u, policy %d", BB, grid, POL);    \
    timeit([&] { hipLaunchKernelGGL(linear_kernel<POL>, dim3(grid), dim3(256), 0, 0, out, BB); }, label);
    LIN(0) LIN(1) LIN(2) LIN(3)
#undef LIN
  }
  {
    const long pages = total / 4096;
    for (int G : {2048, 16384, 131072, (int)pages})
      for (int mode : {0, 1, 1, 2, 3}) {
        static int flip = 0;
        const int sh = mode == 1 ? ((flip++ & 1) ? 4 : 1) : 0;
        const int iters = (int)(pages / G);
        snprintf(label, sizeof label, "paged: grid %7d x %4d pages per block, mode %d shift %d, policy 0", G, iters, mode, sh);
        timeit([&] { hipLaunchKernelGGL(paged_kernel<0>, dim3(G), dim3(256), 0, 0, out, G, iters, mode, sh); }, label);
        if (mode == 0) {
          snprintf(label, sizeof label, "paged: grid %7d x %4d pages per block, mode %d shift %d, policy 2", G, iters, mode, sh);
          timeit([&] { hipLaunchKernelGGL(paged_kernel<2>, dim3(G), dim3(256), 0, 0, out, G, iters, mode, sh); }, label);
        }
      }
  }
  for (int W : {256, 512, 1024, 2048, 4096})
    for (int TR : {8, 32}) {
      Q q{out, row_bytes, rows, TR, W};
      const unsigned grid = (unsigned)(row_bytes / W);
#define CS(POL)                                                                                                           \
      snprintf(label, sizeof label, "code-stationary, segment %5d B, %2d rows per wave step, grid %4u, policy %d", W, TR, grid, POL); \
      timeit([&] { hipLaunchKernelGGL(colstat_kernel<POL>, dim3(grid), dim3(256), 0, 0, q); }, label);
      CS(0) CS(1) CS(2) CS(3)
#undef CS
    }
  return 0;
}
