// Build: hipcc --offload-arch=gfx950 -O2 -fPIC -shared tools/lds_canary.hip -o tools/_build/liblds_canary.so   (loaded by
// tools/t2b_canary.py, tools/vgpr_canary.py, tools/lds_poison_step.py; investigation tools, not part of the product)
// Debug tool: workgroups that fill their LDS with a pattern and keep checking it for `iters` rounds; a word that changes under
// them (another workgroup's stray LDS write) is counted.  Run beside the kernel under suspicion (tools/t2b_canary.py).
#include <hip/hip_runtime.h>
extern "C" __global__ void __launch_bounds__(256) lds_canary(int words, int iters, unsigned* bad, unsigned* first) {
  extern __shared__ unsigned sm[];
  const unsigned tag = 0xC0DE0000u ^ (blockIdx.x << 4);
  for (int i = threadIdx.x; i < words; i += 256) sm[i] = tag + i;
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    for (int i = threadIdx.x; i < words; i += 256) {
      const unsigned v = ((volatile unsigned*)sm)[i];
      if (v != tag + i) {
        const unsigned k = atomicAdd(bad, 1u);
        if (k < 16) { first[4 * k] = blockIdx.x; first[4 * k + 1] = i; first[4 * k + 2] = v; first[4 * k + 3] = it; }
        sm[i] = tag + i;
      }
    }
    __builtin_amdgcn_s_sleep(20);
  }
}
extern "C" int lds_canary_launch(int grid, int lds_bytes, int iters, unsigned* bad, unsigned* first, void* stream) {
  hipFuncSetAttribute((const void*)lds_canary, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipLaunchKernelGGL(lds_canary, dim3(grid), dim3(256), lds_bytes, (hipStream_t)stream, lds_bytes / 4, iters, bad, first);
  return (int)hipGetLastError();
}

// Fill the whole LDS of (about) every CU with `pattern`: one workgroup per CU-sized LDS allocation.  A later kernel that
// reads LDS it never wrote then computes with the pattern instead of its predecessor's leftovers.
extern "C" __global__ void __launch_bounds__(1024) lds_poison(unsigned pattern, int words, unsigned* sink) {
  extern __shared__ unsigned sm[];
  for (int i = threadIdx.x; i < words; i += 1024) sm[i] = pattern;
  __syncthreads();
  __builtin_amdgcn_s_sleep(100);
  if (sm[(threadIdx.x * 37) % words] == 0x12345u) sink[0] = 1;      // keep the stores alive
}
extern "C" int lds_poison_launch(unsigned pattern, int grid, void* sink, void* stream) {
  static bool once = false;
  if (!once) { (void)hipFuncSetAttribute((const void*)lds_poison, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); once = true; }
  hipLaunchKernelGGL(lds_poison, dim3(grid), dim3(1024), 160 * 1024, (hipStream_t)stream, pattern, 160 * 256, (unsigned*)sink);
  return (int)hipGetLastError();
}

// Register canary: every thread runs the same fp32 recurrence twice in separate registers (16 + 16 chains, fma + exp as in the
// step's element-wise kernels) and compares the copies at the end.  The copies can only differ if a register changed under it.
extern "C" __global__ void __launch_bounds__(256) vgpr_canary(int iters, unsigned* bad, unsigned* first) {
  float x[16], y[16];
  const float seed = 0.001f * (float)(threadIdx.x + 1) + 0.000001f * (float)blockIdx.x;
#pragma unroll
  for (int k = 0; k < 16; ++k) { x[k] = seed + 0.01f * k; y[k] = x[k]; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      x[k] = fmaf(x[k], 0.999f, 0.0007f * __expf(-x[k]));
      asm volatile("" : "+v"(x[k]));
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      y[k] = fmaf(y[k], 0.999f, 0.0007f * __expf(-y[k]));
      asm volatile("" : "+v"(y[k]));
    }
  }
#pragma unroll
  for (int k = 0; k < 16; ++k)
    if (__float_as_uint(x[k]) != __float_as_uint(y[k])) {
      const unsigned n = atomicAdd(bad, 1u);
      if (n < 16) { first[4 * n] = blockIdx.x; first[4 * n + 1] = threadIdx.x; first[4 * n + 2] = k; first[4 * n + 3] = __float_as_uint(x[k]) ^ __float_as_uint(y[k]); }
    }
}
extern "C" int vgpr_canary_launch(int grid, int iters, unsigned* bad, unsigned* first, void* stream) {
  hipLaunchKernelGGL(vgpr_canary, dim3(grid), dim3(256), 0, (hipStream_t)stream, iters, bad, first);
  return (int)hipGetLastError();
}
