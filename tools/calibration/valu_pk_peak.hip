// Calibration 3: issue rate of packed (v_pk_fma_f32) vs scalar (v_fma_f32) fp32 VALU, and the write bandwidth of the
// compat op's store pattern (256-B runs per wave, rows 4*n bytes apart).  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <bool PK>
__global__ __launch_bounds__(256) void valu_loop(float *out, int iters, float s) {
  f32x2 a[8];
  for (int k = 0; k < 8; ++k) a[k] = f32x2{(float)threadIdx.x + k, (float)k};
  const f32x2 m = {s, s * 0.5f}, c = {0.25f, 0.125f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (PK) {
        a[k] = __builtin_elementwise_fma(a[k], m, c);
      } else {
        a[k][0] = __builtin_fmaf(a[k][0], m[0], c[0]);
        a[k][1] = __builtin_fmaf(a[k][1], m[1], c[1]);
      }
    }
  }
  float r = 0.f;
  for (int k = 0; k < 8; ++k) r += a[k][0] + a[k][1];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

// Forced instruction forms (inline asm, so SLP cannot repack them): 8 independent chains per thread.
//   0 v_fma_f32 (scalar, 16/iter)   1 v_pk_fma_f32   2 v_pk_fma_f32 with an op_sel-broadcast operand
//   3 v_pk_mul_f32 + v_pk_add_f32   4 v_pk_fma_f32, ONE dependent chain (latency)   5 v_fma_f32, ONE chain
template <int MODE>
__global__ __launch_bounds__(256) void valu_forms(float *out, int iters, float s) {
  f32x2 a[8];
  for (int k = 0; k < 8; ++k) a[k] = f32x2{(float)threadIdx.x + k, (float)k};
  f32x2 m = {s, s * 0.5f}, c = {0.25f, 0.125f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      f32x2 &x = (MODE >= 4) ? a[0] : a[k];
      if (MODE == 0 || MODE == 5) {
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[0]) : "v"(m[0]), "v"(c[0]));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[1]) : "v"(m[1]), "v"(c[1]));
      } else if (MODE == 1 || MODE == 4) {
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(c));
      } else if (MODE == 2) {
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(x) : "v"(m), "v"(c));
      } else if (MODE == 3) {
        asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(m));
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(c));
      }
    }
  }
  float r = 0.f;
  for (int k = 0; k < 8; ++k) r += a[k][0] + a[k][1];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

// every block writes ROWS rows x 512 floats, like gq_scores_kernel's stores
__global__ __launch_bounds__(256) void store_pattern(float *out, int rows, int n) {
  const int j0 = blockIdx.x * 512 + threadIdx.x, r0 = blockIdx.y * 16;
  for (int r = 0; r < 16 && r0 + r < rows; ++r) {
    out[(long)(r0 + r) * n + j0] = (float)r;
    out[(long)(r0 + r) * n + j0 + 256] = (float)r;
  }
}

__global__ __launch_bounds__(256) void store_linear(float4 *out, long n4) {
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < n4; t += (long)gridDim.x * 256) out[t] = float4{1.f, 2.f, 3.f, 4.f};
}

template <typename F>
float timed(F f, int reps) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) f();
  (void)hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) f();
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main() {
  float *out; const int rows = 4096, n = 65536;
  (void)hipMalloc(&out, (size_t)rows * n * 4);
  const int iters = 4096, blocks = 256 * 8;
  for (int pk = 0; pk < 2; ++pk) {
    const float ms = pk ? timed([&] { hipLaunchKernelGGL(valu_loop<true>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f); }, 10)
                        : timed([&] { hipLaunchKernelGGL(valu_loop<false>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f); }, 10);
    const double fma = (double)blocks * 256 * iters * 16;   // scalar FMAs
    printf("%s: %.3f ms, %.1f TFLOP/s (2 flop/FMA)\n", pk ? "v_pk_fma_f32" : "v_fma_f32   ", ms, fma * 2 / (ms * 1e-3) / 1e12);
  }
  const char *names[6] = {"v_fma_f32 x16 (8 chains x2)", "v_pk_fma_f32 x8", "v_pk_fma_f32 op_sel bcast x8",
                          "v_pk_mul+v_pk_add x8", "v_pk_fma_f32 one chain x8", "v_fma_f32 one chain x16"};
  for (int wps = 1; wps <= 8; wps *= 2) {   // waves per SIMD
    const int nb = 256 * wps;
    for (int mode = 0; mode < 6; ++mode) {
      auto go = [&] {
        switch (mode) {
          case 0: hipLaunchKernelGGL(valu_forms<0>, dim3(nb), dim3(256), 0, 0, out, iters, 1.0001f); break;
          case 1: hipLaunchKernelGGL(valu_forms<1>, dim3(nb), dim3(256), 0, 0, out, iters, 1.0001f); break;
          case 2: hipLaunchKernelGGL(valu_forms<2>, dim3(nb), dim3(256), 0, 0, out, iters, 1.0001f); break;
          case 3: hipLaunchKernelGGL(valu_forms<3>, dim3(nb), dim3(256), 0, 0, out, iters, 1.0001f); break;
          case 4: hipLaunchKernelGGL(valu_forms<4>, dim3(nb), dim3(256), 0, 0, out, iters, 1.0001f); break;
          default: hipLaunchKernelGGL(valu_forms<5>, dim3(nb), dim3(256), 0, 0, out, iters, 1.0001f); break;
        }
      };
      const float ms = timed(go, 5);
      const int per_iter = (mode == 0 || mode == 5 || mode == 3) ? 16 : 8;   // VALU instructions per thread-iteration
      // cycles per instruction per SIMD at 2.4 GHz: time * f / (instructions issued per SIMD)
      const double instr_per_simd = (double)iters * per_iter * wps;
      printf("waves/SIMD %d  %-30s %.3f ms  %.2f cycles/instr/SIMD (at 2.4 GHz)\n", wps, names[mode], ms,
             ms * 1e-3 * 2.4e9 / instr_per_simd);
    }
  }
  const double gb = (double)rows * n * 4 / 1e9;
  float ms = timed([&] { hipLaunchKernelGGL(store_pattern, dim3(n / 512, rows / 16), dim3(256), 0, 0, out, rows, n); }, 10);
  printf("store pattern of gq_scores (rows=%d): %.3f ms -> %.0f GB/s\n", rows, ms, gb / ms * 1e3);
  ms = timed([&] { hipLaunchKernelGGL(store_linear, dim3(256 * 16), dim3(256), 0, 0, (float4 *)out, (long)rows * n / 4); }, 10);
  printf("linear float4 stores: %.3f ms -> %.0f GB/s\n", ms, gb / ms * 1e3);
  ms = timed([&] { (void)hipMemsetAsync(out, 0, (size_t)rows * n * 4, 0); }, 10);
  printf("hipMemsetAsync: %.3f ms -> %.0f GB/s\n", ms, gb / ms * 1e3);
  return 0;
}
