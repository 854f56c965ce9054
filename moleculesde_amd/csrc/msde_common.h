// msde_common.h — shared device helpers for libmsde_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/msde_hip.h"

#define MSDE_WAVE 64

#define MSDE_CHECK_LAUNCH()                      \
  do {                                           \
    hipError_t _e = hipGetLastError();           \
    if (_e != hipSuccess) return (int)_e;        \
  } while (0)

// hipGetLastError() is per-thread sticky state shared with every other HIP user in the process
// (torch included): clear it right before our launch so MSDE_CHECK_LAUNCH reports only our error.
#define MSDE_LAUNCH(...)            \
  do {                              \
    (void)hipGetLastError();        \
    hipLaunchKernelGGL(__VA_ARGS__); \
  } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// fixed-order sum of per-workgroup partial slabs (defined in linear.hip)
int msde_reduce_slabs(const float* slabs, int splits, size_t n, float* out, const float* cs, size_t nb, float* outb,
                      hipStream_t st);

// p[0 .. n_words) := 0 as a KERNEL (defined in linear.hip).  The library never uses hipMemsetAsync: a memset NODE of a
// captured step was observed not to take effect in the first replay that follows an eager step on the same stream
// (tools/replay_growth_debug2.py is the 3-second reproduction), a kernel node is always executed.
int msde_zero_words(void* p, size_t n_words, hipStream_t st);

// Row bounds: kernels that REDUCE over rows take the device address of the TRUE row count of their operand (`rows_dev`,
// NULL = all rows) and clamp their row range with it, so one captured hipGraph serves batches of different sizes padded to
// the same capacities (include/msde_hip.h, "Row bounds").  The pointer is an explicit argument of every such entry point:
// the library holds no table, no global state.
__device__ __forceinline__ int msde_true_rows(int cap, const int* __restrict__ dev) {
  return dev ? min(cap, dev[0]) : cap;
}

// compute units of the current device (256 on MI355X); queried once -- one process drives one GPU model
static inline int msde_num_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
      cus = n;
    else
      cus = 256;
  }
  return cus;
}

// ---- tiny vector abstraction: V = 4 (float4, 16 B/lane) or V = 1 (scalar fallback) -------------
template <int V> struct VecT;
template <> struct VecT<4> { using type = float4; };
template <> struct VecT<1> { using type = float; };

__device__ __forceinline__ float4 vzero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
template <int V> __device__ __forceinline__ typename VecT<V>::type vzero();
template <> __device__ __forceinline__ float4 vzero<4>() { return vzero4(); }
template <> __device__ __forceinline__ float vzero<1>() { return 0.f; }

__device__ __forceinline__ float4 vadd(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float vadd(float a, float b) { return a + b; }
__device__ __forceinline__ float4 vmul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float vmul(float a, float b) { return a * b; }
__device__ __forceinline__ float4 vscale(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float vscale(float a, float s) { return a * s; }
__device__ __forceinline__ float4 vfma(float4 a, float4 b, float4 c) {
  return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}
__device__ __forceinline__ float vfma(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ float4 vrelu(float4 a) { return make_float4(fmaxf(a.x, 0.f), fmaxf(a.y, 0.f), fmaxf(a.z, 0.f), fmaxf(a.w, 0.f)); }
__device__ __forceinline__ float vrelu(float a) { return fmaxf(a, 0.f); }
// g where a > 0 else 0
__device__ __forceinline__ float4 vgate(float4 g, float4 a) {
  return make_float4(a.x > 0.f ? g.x : 0.f, a.y > 0.f ? g.y : 0.f, a.z > 0.f ? g.z : 0.f, a.w > 0.f ? g.w : 0.f);
}
__device__ __forceinline__ float vgate(float g, float a) { return a > 0.f ? g : 0.f; }
__device__ __forceinline__ float vhsum(float4 a) { return (a.x + a.y) + (a.z + a.w); }
__device__ __forceinline__ float vhsum(float a) { return a; }

// sum across the `width` (power of two <= 64) consecutive lanes of a row group
__device__ __forceinline__ float group_sum(float v, int width) {
  for (int o = width >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float group_max(float v, int width) {
  for (int o = width >> 1; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// threads-per-row: smallest power of two >= cols (cols = D / V), clamped to [4, 64]
static inline int pick_tpr(int cols) {
  int t = 4;
  while (t < cols && t < 64) t <<= 1;
  return t;
}

// counter-based RNG for dropout masks: the same (seed, index) always gives the same uniform, so
// backward regenerates the forward mask without storing it.  (splitmix64 finaliser.)
__device__ __forceinline__ float msde_uniform(unsigned long long seed, unsigned long long idx) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (idx + 1ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (float)(z >> 40) * (1.0f / 16777216.0f);  // 24 random bits -> [0,1)
}

// N(0,1) from two counter-based uniforms (Box-Muller); the same (seed, index) always gives the same draw
__device__ __forceinline__ float msde_randn(unsigned long long seed, unsigned long long idx) {
  float u1 = msde_uniform(seed, 2ull * idx), u2 = msde_uniform(seed, 2ull * idx + 1ull);
  u1 = fmaxf(u1, 5.9604645e-8f);
  return sqrtf(-2.f * logf(u1)) * cosf(6.283185307179586f * u2);
}

// launch a row kernel templated on the vector width; defines cols/tpr for the argument list
#define LAUNCH_ROWS(KERNEL, ROWS, D, ...)                                                                          \
  if ((D) % 4 == 0) {                                                                                              \
    int cols = (D) / 4, tpr = pick_tpr(cols), rpb = 256 / tpr;                                                     \
    MSDE_LAUNCH(KERNEL<4>, dim3(((ROWS) + rpb - 1) / rpb), dim3(256), 0, as_stream(stream), __VA_ARGS__);    \
  } else {                                                                                                         \
    int cols = (D), tpr = pick_tpr(cols), rpb = 256 / tpr;                                                         \
    MSDE_LAUNCH(KERNEL<1>, dim3(((ROWS) + rpb - 1) / rpb), dim3(256), 0, as_stream(stream), __VA_ARGS__);    \
  }

