// Calibration 6: would a mixed-precision filter tile loop pay?  Per 32-code x 64-row tile pair:
//   MODE 0: 12 x v_mfma_f32_32x32x16_bf16              (today's split-bf16 filter: hh + lh + hl, K = 96 slots, 96 passes)
//   MODE 1: 4 x v_mfma_f32_32x32x16_f16 + 2 x v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 operands)
//           = main product of the fp16 h parts (K = 32) + both correction types (lh | hl, K = 64) on the block-scaled fp8
//           instruction, whose scale operand applies the 2^-11 of the l parts: 32 + 32 = 64 passes
//   MODE 2: as 1 with fp6 (e2m3) correction operands (8 passes per x64 MFMA): 32 + 16 = 48 passes
// Same LDS traffic (4 x ds_read_b128 per tile), same v_max3 epilogue, one 512-thread block per CU.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/mfma_mixed_f16_f8 tools/calibration/mfma_mixed_f16_f8.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8 as_bf(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ f16x8 as_h(u32x4 v) { return __builtin_bit_cast(f16x8, v); }
__device__ __forceinline__ i32x8 cat8(u32x4 a, u32x4 b) {
  return i32x8{(int)a.x, (int)a.y, (int)a.z, (int)a.w, (int)b.x, (int)b.y, (int)b.z, (int)b.w};
}

template <int MODE>
__global__ __launch_bounds__(512, 1) void tile_loop(float *out, const u32x4 *in, int iters, unsigned long long *clk) {
  __shared__ u32x4 lds[2048];   // 32 KB: 8 tiles x 4 vectors x 64 lanes
  for (int i = threadIdx.x; i < 2048; i += 512) lds[i] = in[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  u32x4 b[8];   // row operands
  for (int s = 0; s < 8; ++s) b[s] = in[(threadIdx.x * 9 + s) & 2047];
  float t0 = -1e30f, t1 = -1e30f, t2 = 0.f;
  const int sc = 127 | (116 << 8);   // E8M0 scale bytes: 2^0, 2^-11
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  f32x16 dp[2];
  for (int r = 0; r < 2; ++r) for (int k = 0; k < 16; ++k) dp[r][k] = -1e30f;
  for (int it = 0; it < (MODE == 3 ? iters / 2 : iters); ++it) {
    const u32x4 *p = lds + (it & 7) * 256 + lane;
    const u32x4 a0 = p[0], a1 = p[64], a2 = p[128], a3 = p[192];
    f32x16 d[2];
    d[0] = d[1] = f32x16{0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f};
    if constexpr (MODE == 0) {
      const u32x4 a[4] = {a0, a1, a2, a3};
      constexpr int ca[6] = {0, 1, 2, 3, 0, 1}, cb[6] = {0, 1, 0, 1, 2, 3};
#pragma unroll
      for (int s = 0; s < 6; ++s) {
        d[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a[ca[s]]), as_bf(b[cb[s]]), d[0], 0, 0, 0);
        d[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a[ca[s]]), as_bf(b[4 + cb[s]]), d[1], 0, 0, 0);
      }
    } else {
      d[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h(a0), as_h(b[0]), d[0], 0, 0, 0);
      d[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h(a0), as_h(b[4]), d[1], 0, 0, 0);
      d[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h(a1), as_h(b[1]), d[0], 0, 0, 0);
      d[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h(a1), as_h(b[5]), d[1], 0, 0, 0);
      constexpr int FMT = MODE == 2 ? 2 : 0;   // 0: fp8 e4m3, 2: fp6 e2m3 (reads the low 24 bytes of the operand)
      d[0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(cat8(a2, a3), cat8(b[2], b[3]), d[0], FMT, FMT, 1, sc, 0, sc);
      d[1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(cat8(a2, a3), cat8(b[6], b[7]), d[1], FMT, FMT, 1, sc, 0, sc);
    }
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
      t0 = __builtin_fmaxf(__builtin_fmaxf(t0, dp[0][k]), dp[0][k + 1]);
      t1 = __builtin_fmaxf(__builtin_fmaxf(t1, dp[1][k]), dp[1][k + 1]);
    }
    dp[0] = d[0]; dp[1] = d[1];
    if constexpr (MODE == 3) {   // a second tile in the same iteration: four independent accumulator chains in flight
      const u32x4 *p2 = lds + ((it + 4) & 7) * 256 + lane;
      const u32x4 c0 = p2[0], c1 = p2[64], c2 = p2[128], c3 = p2[192];
      f32x16 e[2];
      e[0] = e[1] = f32x16{0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f,0.f};
      e[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h(c0), as_h(b[0]), e[0], 0, 0, 0);
      e[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h(c0), as_h(b[4]), e[1], 0, 0, 0);
      e[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h(c1), as_h(b[1]), e[0], 0, 0, 0);
      e[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h(c1), as_h(b[5]), e[1], 0, 0, 0);
      e[0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(cat8(c2, c3), cat8(b[2], b[3]), e[0], 0, 0, 1, sc, 0, sc);
      e[1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(cat8(c2, c3), cat8(b[6], b[7]), e[1], 0, 0, 1, sc, 0, sc);
#pragma unroll
      for (int k = 0; k < 16; k += 2) {
        t0 = __builtin_fmaxf(__builtin_fmaxf(t0, e[0][k]), e[0][k + 1]);
        t1 = __builtin_fmaxf(__builtin_fmaxf(t1, e[1][k]), e[1][k + 1]);
      }
    }
  }
  for (int r = 0; r < 2; ++r) for (int k = 0; k < 16; ++k) t2 += dp[r][k];
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 512 + threadIdx.x] = t0 + t1 + t2;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int MODE>
void run(int blocks, float *out, const u32x4 *in, unsigned long long *clk, int rounds) {
  const int iters = 256;   // tiles per wave (config 2: 65536 codes / 8 splits / 32)
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int w = 0; w < 50; ++w) hipLaunchKernelGGL((tile_loop<MODE>), dim3(blocks), dim3(512), 0, 0, out, in, iters, clk);
  (void)hipEventRecord(e0);
  for (int w = 0; w < rounds; ++w) hipLaunchKernelGGL((tile_loop<MODE>), dim3(blocks), dim3(512), 0, 0, out, in, iters, clk);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks * 2);
  (void)hipMemcpy(h.data(), clk, blocks * 16, hipMemcpyDeviceToHost);
  std::vector<double> ghz;
  for (int b = 0; b < blocks; ++b) ghz.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 0.1);
  std::sort(ghz.begin(), ghz.end());
  const double us = ms / rounds * 1e3;
  const int passes = MODE == 0 ? 96 : (MODE == 2 ? 48 : 64);
  printf("mode=%d (%s) blocks=%d: %.1f us/launch, in-kernel clock %.3f GHz, %.1f cycles per tile pair (nominal %d)\n", MODE,
         MODE == 0 ? "12 bf16 MFMAs" : (MODE == 1 ? "4 f16 + 2 scaled fp8 x64" : (MODE == 2 ? "4 f16 + 2 scaled fp6 x64" : "fp8 form, two tiles per iteration")), blocks, us,
         ghz[blocks / 2], us * 1e-6 * ghz[blocks / 2] * 1e9 / (iters * 2.0), passes * 4);
}

int main() {
  float *out; u32x4 *in; unsigned long long *clk;
  (void)hipMalloc(&out, 1024 * 512 * 4); (void)hipMalloc(&clk, 1024 * 16); (void)hipMalloc(&in, 2048 * 16);
  std::vector<unsigned> h(8192);
  unsigned s = 12345u;
  for (int i = 0; i < 8192; ++i) {   // two random bf16 / fp16-ish values per word (as fp8 bytes: random finite patterns)
    s = s * 1664525u + 1013904223u; const unsigned lo = ((s >> 16) & 0x807fu) | 0x3f00u;
    s = s * 1664525u + 1013904223u; const unsigned hi = ((s >> 16) & 0x807fu) | 0x3f80u;
    h[i] = (lo | (hi << 16)) & 0x3f7f3f7fu;   // keeps every byte a finite e4m3 / e2m3 pattern and fp16 values small
  }
  (void)hipMemcpy(in, h.data(), 8192 * 4, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 3; ++rep) {
    run<0>(256, out, in, clk, 2000);
    run<1>(256, out, in, clk, 2000);
    run<2>(256, out, in, clk, 2000);
    run<3>(256, out, in, clk, 2000);
  }
  return 0;
}
