// How many independent vector instructions hide in the shadow of one v_mfma_f32_32x32x2_f32 (fp32 in, 64 cyc/SIMD)?
// One wave per SIMD (256 threads, one workgroup per CU); each loop body is 8 MFMAs (2 accumulator chains), each followed by
// F fillers (v_fma_f32 / v_exp_f32 / ds_read_b32 on distinct registers).  Cycles per MFMA from s_memtime.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_filler_probe.hip -o gpurun_out/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int F, int KIND>
__global__ void __launch_bounds__(256, 1) probe(float* out, long long* cyc, int iters) {
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i * 1e-3f;
  __syncthreads();
  f32x16 a0, a1;
  for (int i = 0; i < 16; ++i) { a0[i] = 0.f; a1[i] = 0.f; }
  float x = threadIdx.x * 1e-3f, y = 1.0001f;
  float f[16];
  for (int i = 0; i < 16; ++i) f[i] = x + i;
  const float* lp = lds + threadIdx.x;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (m & 1) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(a1) : "v"(x), "v"(y));
      else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(x), "v"(y));
#pragma unroll
      for (int k = 0; k < F; ++k) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[k]) : "v"(y));
        else if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(f[k]));
        else asm volatile("ds_read_b32 %0, %1" : "=v"(f[k]) : "v"((unsigned)(size_t)lp + 4 * k));
      }
    }
    if (KIND == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + f[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int F, int KIND> void run(const char* name, float* out, long long* cyc) {
  const int iters = 2000, nb = 256;
  probe<F, KIND><<<nb, 256>>>(out, cyc, iters);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a);
  probe<F, KIND><<<nb, 256>>>(out, cyc, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  std::vector<long long> h(nb);
  hipMemcpy(h.data(), cyc, nb * 8, hipMemcpyDeviceToHost);
  double mean = 0; for (auto v : h) mean += v; mean /= nb;
  printf("%s F=%2d  cycles/MFMA %.1f   wall %.3f ms  -> %.2f GHz-equivalent, %.1f TF\n", name, F, mean / (iters * 8.0), ms,
         mean / (ms * 1e6), 2.0 * 32 * 32 * 2 * 8.0 * iters * nb * 4 / (ms * 1e9));
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
  run<0, 0>("fma", out, cyc); run<4, 0>("fma", out, cyc); run<8, 0>("fma", out, cyc); run<10, 0>("fma", out, cyc);
  run<12, 0>("fma", out, cyc); run<14, 0>("fma", out, cyc); run<16, 0>("fma", out, cyc);
  run<2, 1>("exp", out, cyc); run<4, 1>("exp", out, cyc); run<6, 1>("exp", out, cyc); run<8, 1>("exp", out, cyc);
  run<2, 2>("ds_read", out, cyc); run<4, 2>("ds_read", out, cyc); run<8, 2>("ds_read", out, cyc);
  return 0;
}
