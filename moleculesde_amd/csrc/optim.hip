// optim.hip — Adam over one flat fp32 parameter buffer (examples/pretrain_MoleculeSDE.py:331-337,156).
// torch.optim.Adam (no amsgrad) semantics:
//   g += wd * p;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
//   p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
#include "msde_common.h"

// ONE update routine for the flat and the chunk-table kernels (scalar and 16-byte paths): the data-parallel step runs
// the flat kernel on the all-reduced buffer, the single-GPU step the chunk kernel on the gradients in place, and the two
// must produce bit-identical parameters from identical gradients (tests/test_gpu_dp.py) -- so no expression here is
// left to the compiler's context-dependent fma contraction.
__device__ __forceinline__ void adam_update(float& pi, float gi, float& mi, float& vi, float beta1, float beta2, float eps,
                                            float wd, float grad_scale, float inv_sqrt_bc2, float step) {
#pragma clang fp contract(off)
  gi = gi * grad_scale;
  if (wd != 0.f) gi = fmaf(wd, pi, gi);
  mi = beta1 * mi + (1.f - beta1) * gi;
  vi = beta2 * vi + ((1.f - beta2) * gi) * gi;
  const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
  pi = pi - step * (mi / denom);
}

__global__ void adam_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                 float* __restrict__ v, long long n, const int* __restrict__ step_dev,
                                 const long long* __restrict__ seg_end, const float* __restrict__ seg_lr, int S,
                                 float beta1, float beta2, float eps, float wd, float grad_scale) {
  int t = step_dev[0];
  float bc1 = 1.f - powf(beta1, (float)t);
  float bc2 = 1.f - powf(beta2, (float)t);
  float inv_sqrt_bc2 = 1.f / sqrtf(bc2);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    int s = 0;
    while (s < S - 1 && i >= seg_end[s]) ++s;
    float lr = seg_lr[s];
    float pi = p[i], mi = m[i], vi = v[i];
    adam_update(pi, g[i], mi, vi, beta1, beta2, eps, wd, grad_scale, inv_sqrt_bc2, lr / bc1);
    m[i] = mi;
    v[i] = vi;
    p[i] = pi;
  }
}

extern "C" int msde_adam_flat(float* p, const float* g, float* m, float* v, long long n, const int* step_dev,
                              const long long* seg_end, const float* seg_lr, int S, float beta1, float beta2, float eps,
                              float weight_decay, float grad_scale, void* stream) {
  if (n < 0 || S <= 0 || !p || !g || !m || !v || !step_dev || !seg_end || !seg_lr) return MSDE_EINVAL;
  if (n == 0) return 0;
  long long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  MSDE_LAUNCH(adam_flat_kernel, dim3((int)blocks), dim3(256), 0, as_stream(stream), p, g, m, v, n, step_dev,
                     seg_end, seg_lr, S, beta1, beta2, eps, weight_decay, grad_scale);
  MSDE_CHECK_LAUNCH();
  return 0;
}

// ---- chunk-table variants: the gradients stay where autograd left them -------------------------------------
// table[c] = {address of the chunk's first gradient element (0: the parameter received no gradient -> zeros),
//             offset of the chunk in the flat buffers, element count (<= MSDE_CHUNK)}, three int64 per chunk.
// One workgroup per chunk; chunks never straddle tensors.
#define MSDE_CHUNK 2048

__global__ void __launch_bounds__(256)
gather_chunks_kernel(const long long* __restrict__ table, float* __restrict__ flat) {
  const long long* e = table + (size_t)blockIdx.x * 3;
  const float* src = reinterpret_cast<const float*>(e[0]);
  float* dst = flat + e[1];
  const int cnt = (int)e[2];
  for (int i = threadIdx.x; i < cnt; i += 256) dst[i] = src ? src[i] : 0.f;
}

__global__ void __launch_bounds__(256)
adam_chunks_kernel(float* __restrict__ p, const long long* __restrict__ table, float* __restrict__ m,
                   float* __restrict__ v, const int* __restrict__ step_dev, const long long* __restrict__ seg_end,
                   const float* __restrict__ seg_lr, int S, float beta1, float beta2, float eps, float wd,
                   float grad_scale) {
  const long long* e = table + (size_t)blockIdx.x * 3;
  const float* g = reinterpret_cast<const float*>(e[0]);
  const long long off = e[1];
  const int cnt = (int)e[2];
  const int t = step_dev[0];
  const float bc1 = 1.f - powf(beta1, (float)t);
  const float bc2 = 1.f - powf(beta2, (float)t);
  const float inv_sqrt_bc2 = 1.f / sqrtf(bc2);
  int s = 0;
  while (s < S - 1 && off >= seg_end[s]) ++s;       // a chunk lies inside one tensor, hence inside one group
  const float lr = seg_lr[s];
  const float step = lr / bc1;
  auto upd = [&](float& pi, float gi, float& mi, float& vi) {
    adam_update(pi, gi, mi, vi, beta1, beta2, eps, wd, grad_scale, inv_sqrt_bc2, step);
  };
  // 16-byte path: chunk start, length and gradient address all float4-aligned (every chunk of a tensor whose element
  // count and flat offset are multiples of 4 -- all but a few bias / scalar tails); same arithmetic per element
  const bool vec = ((off | (long long)cnt) & 3) == 0 && (reinterpret_cast<uintptr_t>(g) & 15) == 0 &&
                   ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0;
  if (vec) {
    float4* p4 = reinterpret_cast<float4*>(p + off);
    float4* m4 = reinterpret_cast<float4*>(m + off);
    float4* v4 = reinterpret_cast<float4*>(v + off);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (int i = threadIdx.x; i < cnt / 4; i += 256) {
      float4 pi = p4[i], mi = m4[i], vi = v4[i];
      const float4 gi = g ? g4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      upd(pi.x, gi.x, mi.x, vi.x); upd(pi.y, gi.y, mi.y, vi.y); upd(pi.z, gi.z, mi.z, vi.z); upd(pi.w, gi.w, mi.w, vi.w);
      m4[i] = mi; v4[i] = vi; p4[i] = pi;
    }
    return;
  }
  for (int i = threadIdx.x; i < cnt; i += 256) {
    const long long k = off + i;
    float pi = p[k], mi = m[k], vi = v[k];
    upd(pi, g ? g[i] : 0.f, mi, vi);
    m[k] = mi;
    v[k] = vi;
    p[k] = pi;
  }
}

extern "C" int msde_chunk_elems(void) { return MSDE_CHUNK; }

// both device-side counters of a training step -- the step counter that re-seeds the in-kernel noise / dropout masks and the
// optimiser's step count -- advanced by ONE one-thread launch at the head of the step (they were two elementwise-add launches of
// the tensor library, the second one in the serial tail between the slab reduction and Adam)
__global__ void step_counters_kernel(long long* __restrict__ a, int* __restrict__ b) {
  if (a) a[0] += 1;
  if (b) b[0] += 1;
}
extern "C" int msde_step_counters(long long* step_counter, int* optimiser_step, void* stream) {
  if (!step_counter && !optimiser_step) return 0;
  MSDE_LAUNCH(step_counters_kernel, dim3(1), dim3(1), 0, as_stream(stream), step_counter, optimiser_step);
  MSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int msde_gather_chunks(const long long* table, int n_chunks, float* flat, void* stream) {
  if (n_chunks < 0 || !table || !flat) return MSDE_EINVAL;
  if (n_chunks == 0) return 0;
  MSDE_LAUNCH(gather_chunks_kernel, dim3(n_chunks), dim3(256), 0, as_stream(stream), table, flat);
  MSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int msde_adam_chunks(float* p, const long long* table, int n_chunks, float* m, float* v, const int* step_dev,
                                const long long* seg_end, const float* seg_lr, int S, float beta1, float beta2,
                                float eps, float weight_decay, float grad_scale, void* stream) {
  if (n_chunks < 0 || S <= 0 || !p || !table || !m || !v || !step_dev || !seg_end || !seg_lr) return MSDE_EINVAL;
  if (n_chunks == 0) return 0;
  MSDE_LAUNCH(adam_chunks_kernel, dim3(n_chunks), dim3(256), 0, as_stream(stream), p, table, m, v, step_dev, seg_end,
              seg_lr, S, beta1, beta2, eps, weight_decay, grad_scale);
  MSDE_CHECK_LAUNCH();
  return 0;
}
