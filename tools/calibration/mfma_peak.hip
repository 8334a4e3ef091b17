// Calibration: sustained fp32-MFMA rate and in-kernel clock of THIS device under a bare
// v_mfma_f32_32x32x2_f32 loop (no memory traffic), by waves/SIMD, accumulators, unroll and
// VALU filler per MFMA.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int UNROLL, int VALU>
__global__ __launch_bounds__(256) void mfma_loop(float *out, int iters, unsigned long long *clk) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = 0.37f + threadIdx.x * 1e-3f, b = -0.61f + threadIdx.x * 7e-4f;
  float v[4] = {a, b, a + b, a - b};
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it += UNROLL) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < VALU; ++k) v[k & 3] = __builtin_amdgcn_fmed3f(v[k & 3], v[(k + 1) & 3], v[(k + 2) & 3]);
        if (VALU) { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x2, VALU, 0); }
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = v[0] + v[1] + v[2] + v[3];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int NACC, int UNROLL, int VALU>
void run(int blocks, float *out, unsigned long long *clk) {
  const int iters = 4096 * 4 / NACC;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int w = 0; w < 10; ++w) hipLaunchKernelGGL((mfma_loop<NACC, UNROLL, VALU>), dim3(blocks), dim3(256), 0, 0, out, iters, clk);
  (void)hipEventRecord(e0);
  const int L = 30;
  for (int w = 0; w < L; ++w) hipLaunchKernelGGL((mfma_loop<NACC, UNROLL, VALU>), dim3(blocks), dim3(256), 0, 0, out, iters, clk);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double flops = 2.0 * 32 * 32 * 2 * (double)NACC * iters * blocks * 4 * L;
  std::vector<unsigned long long> h(blocks * 2);
  (void)hipMemcpy(h.data(), clk, blocks * 16, hipMemcpyDeviceToHost);
  std::vector<double> ghz;
  for (int b = 0; b < blocks; ++b) ghz.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 0.1);
  std::sort(ghz.begin(), ghz.end());
  const int wps = blocks / 256;
  printf("waves/SIMD=%d nacc=%d unroll=%d valu/mfma=%d: %.1f us/launch, %.1f TFLOP/s (%.1f%%); clock %.3f GHz; "
         "cycles/MFMA/SIMD %.1f\n", wps, NACC, UNROLL, VALU, ms / L * 1e3, flops / (ms * 1e-3) / 1e12,
         flops / (ms * 1e-3) / 1e12 / 157.3 * 100, ghz[blocks / 2], (double)h[0] / ((double)NACC * iters * wps));
}

int main() {
  float *out; unsigned long long *clk;
  (void)hipMalloc(&out, 1024 * 256 * 4); (void)hipMalloc(&clk, 1024 * 16);
  for (int blocks : {256, 512, 1024}) {
    run<2, 1, 0>(blocks, out, clk);
    run<2, 8, 0>(blocks, out, clk);
    run<4, 1, 0>(blocks, out, clk);
    run<4, 4, 0>(blocks, out, clk);
    run<2, 8, 2>(blocks, out, clk);
    run<2, 8, 4>(blocks, out, clk);
    run<4, 4, 2>(blocks, out, clk);
    run<4, 4, 6>(blocks, out, clk);
  }
  return 0;
}
