// cfconv_fused.hip — fused CFConv forward for gfx950 (schnet.py:141-145,185-195):
//   rbf(d) -> Linear(G,F) -> ShiftedSoftplus -> Linear(F,F) -> * C(d) -> * x1[src] -> segmented sum
// in ONE kernel with fp32 MFMA (v_mfma_f32_32x32x2_f32: exact fp32 fma chain in k order).
//
// Layout / mapping (F = 128):
//   * a workgroup (4 waves) owns a contiguous range of target nodes, hence a contiguous edge range
//     of the by-target CSR; it walks that range in chunks of 64 edges (two 32-row MFMA blocks);
//   * wave w owns filter columns [32w, 32w+32): its slices of W1 (G x 32) and W2 (128 x 32) live
//     in VGPRs for the whole kernel as MFMA B operands (lane l: B[k = 2kk + (l>>5)][col = l&31]),
//     so weights are read from L2 once per workgroup and never touch LDS;
//   * the A operands (rbf tile, then the softplus'd hidden tile) are shared by the 4 waves through
//     LDS, row-major with ODD row strides (2*KK1+1, 129): both the row-per-lane A reads and the
//     column-per-lane epilogue writes are bank-conflict free;
//   * the message tile reuses the hidden tile's LDS; per-target sums are accumulated in LDS in edge
//     order (bitwise deterministic, same order as the reference's scatter) and written once.
#include "msde_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CF_F 128
#define CF_TE 64          // edges per chunk
#define CF_HS 129         // hidden/message tile row stride (floats)
#define CF_MAX_NPW 32     // max target nodes per workgroup

__device__ __forceinline__ float ssp_f(float x) {
  // F.softplus (beta=1, threshold=20) - ln 2   (schnet.py:210-216)
  float sp = x > 20.f ? x : log1pf(expf(x));
  return sp - 0.69314718246459961f;  // torch.log(torch.tensor(2.0)).item() as fp32
}

template <int KK1>
__global__ void __launch_bounds__(256)
cfconv_fused_fwd_kernel(const float* __restrict__ x1, const float* __restrict__ dist, const int* __restrict__ rowptr,
                        const int* __restrict__ src, const int* __restrict__ dst, const float* __restrict__ W1,
                        const float* __restrict__ b1, const float* __restrict__ W2, const float* __restrict__ b2,
                        const float* __restrict__ offset, int N, int G, float coeff, float cutoff, int npw,
                        float* __restrict__ agg) {
  constexpr int RS = 2 * KK1 + 1;  // rbf tile row stride (odd)
  extern __shared__ float lds[];
  float* rbf_t = lds;                          // [64][RS]
  float* hid_t = rbf_t + CF_TE * RS;           // [64][129]  (hidden tile, then message tile)
  float* out_acc = hid_t + CF_TE * CF_HS;      // [npw][128]
  float* c_s = out_acc + CF_MAX_NPW * CF_F;    // [64] cutoff value per edge row (0 for padding rows)
  int* src_s = reinterpret_cast<int*>(c_s + CF_TE);  // [64]
  int* tl_s = src_s + CF_TE;                   // [64] local target index, -1 for padding rows

  const float PI_F = 3.14159265358979323846f;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lcol = lane & 31, lhalf = lane >> 5;
  const int col = wave * 32 + lcol;  // this lane's filter column

  const int n0 = blockIdx.x * npw;
  const int n1 = min(n0 + npw, N);
  if (n0 >= N) return;
  const int e0 = rowptr[n0], e1 = rowptr[n1];

  // weights -> registers (B operands).  torch Linear weight is [out, in].
  float w1r[KK1], w2r[CF_F / 2];
#pragma unroll
  for (int kk = 0; kk < KK1; ++kk) {
    int g = 2 * kk + lhalf;
    w1r[kk] = g < G ? W1[(size_t)col * G + g] : 0.f;
  }
#pragma unroll
  for (int kk = 0; kk < CF_F / 2; ++kk) w2r[kk] = W2[(size_t)col * CF_F + 2 * kk + lhalf];
  const float b1c = b1[col], b2c = b2[col];

  for (int t = tid; t < CF_MAX_NPW * CF_F; t += 256) out_acc[t] = 0.f;

  for (int ec = e0; ec < e1; ec += CF_TE) {
    __syncthreads();  // previous chunk's reduction is done with hid_t / meta
    if (tid < CF_TE) {
      int e = ec + tid;
      bool ok = e < e1;
      float d = ok ? dist[e] : 0.f;
      c_s[tid] = ok ? 0.5f * (cosf(d * PI_F / cutoff) + 1.0f) : 0.f;
      src_s[tid] = ok ? src[e] : -1;
      tl_s[tid] = ok ? dst[e] - n0 : -1;
    }
    // Gaussian smearing tile: rbf[r][g] = exp(coeff * (d_r - mu_g)^2), zero for padding
    for (int idx = tid; idx < CF_TE * 2 * KK1; idx += 256) {
      int r = idx / (2 * KK1), g = idx % (2 * KK1);
      int e = ec + r;
      float v = 0.f;
      if (e < e1 && g < G) {
        float diff = dist[e] - offset[g];
        v = expf(coeff * (diff * diff));
      }
      rbf_t[r * RS + g] = v;
    }
    __syncthreads();

    // ---- GEMM1: [64 x 2KK1] . [2KK1 x 32] per wave, two 32-row blocks
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
#pragma unroll
    for (int kk = 0; kk < KK1; ++kk) {
      float a0 = rbf_t[lcol * RS + 2 * kk + lhalf];
      float a1 = rbf_t[(32 + lcol) * RS + 2 * kk + lhalf];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, w1r[kk], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, w1r[kk], acc1, 0, 0, 0);
    }
    // epilogue 1: bias + shifted softplus -> hidden tile (C/D map: row = (i&3)+8(i>>2)+4*lhalf)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      int row = (i & 3) + 8 * (i >> 2) + 4 * lhalf;
      hid_t[row * CF_HS + col] = ssp_f(acc0[i] + b1c);
      hid_t[(32 + row) * CF_HS + col] = ssp_f(acc1[i] + b1c);
    }
    __syncthreads();

    // ---- GEMM2: [64 x 128] . [128 x 32] per wave
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
#pragma unroll
    for (int kk = 0; kk < CF_F / 2; ++kk) {
      float a0 = hid_t[lcol * CF_HS + 2 * kk + lhalf];
      float a1 = hid_t[(32 + lcol) * CF_HS + 2 * kk + lhalf];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, w2r[kk], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, w2r[kk], acc1, 0, 0, 0);
    }
    __syncthreads();  // every wave is done reading the hidden tile: reuse it for messages

    // epilogue 2: filter = (acc + b2) * C ; message = x1[src] * filter
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      int row = (i & 3) + 8 * (i >> 2) + 4 * lhalf;
      int s0 = src_s[row], s1 = src_s[32 + row];
      float x0 = s0 >= 0 ? x1[(size_t)s0 * CF_F + col] : 0.f;
      float xx1 = s1 >= 0 ? x1[(size_t)s1 * CF_F + col] : 0.f;
      hid_t[row * CF_HS + col] = x0 * ((acc0[i] + b2c) * c_s[row]);
      hid_t[(32 + row) * CF_HS + col] = xx1 * ((acc1[i] + b2c) * c_s[32 + row]);
    }
    __syncthreads();

    // segmented sum in edge order: thread (column, parity) owns the targets of its parity
    {
      int c = tid & 127, par = tid >> 7;
      for (int r = 0; r < CF_TE; ++r) {
        int tl = tl_s[r];
        if (tl >= 0 && (tl & 1) == par) out_acc[tl * CF_F + c] += hid_t[r * CF_HS + c];
      }
    }
  }
  __syncthreads();
  for (int t = tid; t < (n1 - n0) * CF_F; t += 256) agg[(size_t)n0 * CF_F + t] = out_acc[t];
}

extern "C" int msde_cfconv_fused_fwd(const float* x1, const float* dist, const int* rowptr, const int* src,
                                     const int* dst, const float* W1, const float* b1, const float* W2,
                                     const float* b2, const float* offset, int N, int F, int G, float coeff,
                                     float cutoff, int nodes_per_wg, float* agg, void* stream) {
  if (N < 0 || !x1 || !dist || !rowptr || !src || !dst || !W1 || !b1 || !W2 || !b2 || !offset || !agg)
    return MSDE_EINVAL;
  if (F != CF_F || G <= 0 || G > 64) return MSDE_EUNSUP;
  if (nodes_per_wg <= 0) nodes_per_wg = 16;
  if (nodes_per_wg > CF_MAX_NPW) nodes_per_wg = CF_MAX_NPW;
  if (N == 0) return 0;
  int kk1 = (G + 1) / 2;
  int grid = (N + nodes_per_wg - 1) / nodes_per_wg;
  auto lds_bytes = [](int KK1) {
    return (size_t)(CF_TE * (2 * KK1 + 1) + CF_TE * CF_HS + CF_MAX_NPW * CF_F + 3 * CF_TE) * sizeof(float);
  };
#define CF_LAUNCH(KK)                                                                                              \
  {                                                                                                                \
    static bool attr_done = false;                                                                                 \
    if (!attr_done) {                                                                                              \
      hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void*>(&cfconv_fused_fwd_kernel<KK>),             \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes(KK));         \
      if (ae != hipSuccess) return (int)ae;                                                                        \
      attr_done = true;                                                                                            \
    }                                                                                                              \
  }                                                                                                                \
  MSDE_LAUNCH(cfconv_fused_fwd_kernel<KK>, dim3(grid), dim3(256), lds_bytes(KK), as_stream(stream), x1, dist, \
                     rowptr, src, dst, W1, b1, W2, b2, offset, N, G, coeff, cutoff, nodes_per_wg, agg)
  if (kk1 == 26) { CF_LAUNCH(26); }
  else if (kk1 == 25) { CF_LAUNCH(25); }
  else { CF_LAUNCH(32); }
#undef CF_LAUNCH
  MSDE_CHECK_LAUNCH();
  return 0;
}
