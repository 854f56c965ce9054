// Two waves per SIMD: does one wave's VALU work overlap the other wave's fp32 MFMAs?  512 threads: waves 0-3 (one per
// SIMD) issue only v_mfma_f32_32x32x2_f32, waves 4-7 only v_fma_f32 / v_exp_f32.  Time of each role alone and together.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>   // 1: MFMA waves only, 2: VALU waves only, 3: both
__global__ void __launch_bounds__(512, 1) probe(float* out, int iters, int nfma) {
  const int wave = threadIdx.x >> 6;
  float x = threadIdx.x * 1e-3f, y = 1.0001f, s = 0.f;
  if (wave < 4) {
    if (MODE & 1) {
      f32x16 a0, a1;
      for (int i = 0; i < 16; ++i) { a0[i] = 0.f; a1[i] = 0.f; }
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(x), "v"(y));
          asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(a1) : "v"(x), "v"(y));
        }
      }
      for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
    }
  } else {
    if (MODE & 2) {
      float f[8];
      for (int i = 0; i < 8; ++i) f[i] = x + i;
      for (int it = 0; it < nfma; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[k]) : "v"(y));
      }
      for (int i = 0; i < 8; ++i) s += f[i];
    }
  }
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE> float run(float* out, int iters, int nfma) {
  probe<MODE><<<256, 512>>>(out, iters, nfma);
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  (void)hipEventRecord(a);
  probe<MODE><<<256, 512>>>(out, iters, nfma);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  return ms;
}

int main() {
  float* out; (void)hipMalloc(&out, 256 * 512 * 4);
  const int iters = 2000;            // 16000 MFMAs per wave = 1.02 M cycles
  for (int nfma : {8000, 16000, 32000}) {   // x 8 v_fma per iteration, 4 cycles each
    float m = run<1>(out, iters, nfma), v = run<2>(out, iters, nfma), b = run<3>(out, iters, nfma);
    printf("MFMA alone %.3f ms | VALU alone (%d x 8 fma) %.3f ms | both %.3f ms  (sum %.3f, max %.3f)\n", m, nfma, v, b, m + v,
           m > v ? m : v);
  }
  return 0;
}
