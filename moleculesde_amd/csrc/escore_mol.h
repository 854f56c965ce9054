// escore_mol.h — shared pieces of the one-workgroup-per-molecule score-network kernels (escore_mol.hip: forward,
// escore_mol_bwd.hip: backward): sizes, the parameter table, MFMA / DPP helpers.
#pragma once
#include "msde_common.h"

#define ES_D 32
#define ES_HC 128
#define ES_NMAX 32
#define ES_ECH 192            // edges per attention chunk (a molecule of <= 14 atoms: one chunk)
#define ES_LDX 36             // LDS row stride of the [., 32] tiles (16-byte aligned rows, conflict-free b128 fragments)
#define ES_LDQ 132            // LDS row stride of qkvs [., 128]
#define ES_LAYERS 4
#define ES_SV 176             // saved floats per (layer, atom): att | y1 | h0 | x2 | out | softmax max[8] | 1/sum[8]
#define ES_NPTR 76
#define ES_EMAX (ES_NMAX * (ES_NMAX - 1))    // 992 edges at most
#define ES_EAL 384                           // molecules of up to this many edges keep their edge features in LDS

typedef float es_f4 __attribute__((ext_vector_type(4)));

// Parameter table: ES_NPTR device pointers (nn.Linear layouts [out][in]) in DEVICE memory -- 17 per GAT layer (lin_query,
// lin_key, lin_value, lin_skip weights [32,32]; their four biases; lin_edge weight; ln1_g, ln1_b, W0, b0, W3, b3, ln2_g, ln2_b),
// then 4 per basis MLP (W1, b1, W2, b2): every entry is a PARAMETER (stable address), never a concatenation.  Read with scalar
// loads at a dynamic layer index (a by-value struct indexed by the layer went through scratch memory).  q|k|v|skip form the
// [128, 32] projection "Wqkvs" of the kernels: row `col` lives in block col >> 5.
struct EsW {
  const float* const* __restrict__ p;
  __device__ __forceinline__ const float* at(int field, int i) const { return p[i * 17 + field]; }
  __device__ __forceinline__ const float* Wq(int l, int block) const { return at(block, l); }          // rows 32 block ..
  __device__ __forceinline__ const float* bq(int l, int block) const { return at(4 + block, l); }
  __device__ __forceinline__ const float* Wedge(int l) const { return at(8, l); }
  __device__ __forceinline__ const float* ln1g(int l) const { return at(9, l); }
  __device__ __forceinline__ const float* ln1b(int l) const { return at(10, l); }
  __device__ __forceinline__ const float* W0(int l) const { return at(11, l); }
  __device__ __forceinline__ const float* b0(int l) const { return at(12, l); }
  __device__ __forceinline__ const float* W3(int l) const { return at(13, l); }
  __device__ __forceinline__ const float* b3(int l) const { return at(14, l); }
  __device__ __forceinline__ const float* ln2g(int l) const { return at(15, l); }
  __device__ __forceinline__ const float* ln2b(int l) const { return at(16, l); }
  __device__ __forceinline__ const float* bW1(int m) const { return p[68 + 4 * m]; }
  __device__ __forceinline__ const float* bb1(int m) const { return p[69 + 4 * m]; }
  __device__ __forceinline__ const float* bW2(int m) const { return p[70 + 4 * m]; }
  __device__ __forceinline__ const float* bb2(int m) const { return p[71 + 4 * m]; }
};

__device__ __forceinline__ es_f4 es_mfma(float a, float b, es_f4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// 1 / (1 + 2^(-x log2 e)) on the transcendental unit (v_exp_f32, v_rcp_f32: ~1 ulp each, no range branches; x -> -inf gives
// rcp(inf) = 0, x -> +inf gives 1)
__device__ __forceinline__ float es_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}
__device__ __forceinline__ float es_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ void es_ld8(const float* __restrict__ p, float (&v)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void es_ld16(const float* __restrict__ p, float (&v)[16]) {
#pragma unroll
  for (int t = 0; t < 16; t += 4) {
    const float4 a = *reinterpret_cast<const float4*>(p + t);
    v[t] = a.x; v[t + 1] = a.y; v[t + 2] = a.z; v[t + 3] = a.w;
  }
}
__device__ __forceinline__ float es_dot4(float4 a, float4 b, float4 c) {      // a . (b + c), channel order
  return ((a.x * (b.x + c.x) + a.y * (b.y + c.y)) + a.z * (b.z + c.z)) + a.w * (b.w + c.w);
}
// sum over the 8 lanes of an atom row: xor 1, xor 2 (quad permutes), then the mirror image inside the half row -- three DPP
// moves on the vector ALU instead of three ds_bpermute round trips
template <int CTRL>
__device__ __forceinline__ float es_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float es_row8_sum(float s) {
  s += es_dpp<0xB1>(s);          // quad_perm [1,0,3,2]
  s += es_dpp<0x4E>(s);          // quad_perm [2,3,0,1]
  return s + es_dpp<0x141>(s);   // row_half_mirror: lane i <-> 7 - i of its group of 8
}
__device__ __forceinline__ void es_layernorm(const float (&v)[4], float eps, float& mu, float& rs) {
  mu = es_row8_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.f / 32.f);
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) q = fmaf(v[k] - mu, v[k] - mu, q);
  rs = rsqrtf(es_row8_sum(q) * (1.f / 32.f) + eps);
}

// targets [t0, t1) whose in-edges [rp[t0], rp[t1]) fit one chunk of ES_ECH edges
__device__ __forceinline__ int es_chunk_end(const int* rp, int t0, int n) {
  int t1 = t0 + 1;
  while (t1 < n && rp[t1 + 1] - rp[t0] <= ES_ECH) ++t1;
  return t1;
}


// D[m0 + 4g + r][n0 + c] = sum over the 32 rows k of X[k][m0 + .] Y[k][n0 + .]  (contraction over atom rows: weight gradients
// of the per-atom Linear layers).  A operand: lane (m = c, g), step t: X[8g + t][m0 + c]; B operand: Y[8g + t][n0 + c].
__device__ __forceinline__ es_f4 es_xty(const float* X, int ldx, int m0, const float* Y, int ldy, int n0, int lane) {
  const int c = lane & 15, g = lane >> 4;
  es_f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 8; ++t) acc = es_mfma(X[(8 * g + t) * ldx + m0 + c], Y[(8 * g + t) * ldy + n0 + c], acc);
  return acc;
}
__device__ __forceinline__ float es_dsilu(float z, float s) { return s * (1.f + z * (1.f - s)); }   // s = sigmoid(z)

// slab of weight gradients one workgroup writes (floats): 4 x [gWqkvs 4096 | gbqkvs 128 | gWedge 1024 | gln1_g 32 | gln1_b 32 |
// gW0 1024 | gb0 32 | gW3 1024 | gb3 32 | gln2_g 32 | gln2_b 32], then 2 x [gW1 8192 | gb1 128 | gW2 384 | gb2 3 + 1 pad]
#define ES_SL_LAYER 7488
#define ES_SL_BQ 4096
#define ES_SL_WE 4224
#define ES_SL_LN1G 5248
#define ES_SL_LN1B 5280
#define ES_SL_W0 5312
#define ES_SL_B0 6336
#define ES_SL_W3 6368
#define ES_SL_B3 7392
#define ES_SL_LN2G 7424
#define ES_SL_LN2B 7456
#define ES_SL_BASIS 8708
#define ES_SL_B1 8192
#define ES_SL_W2 8320
#define ES_SL_B2 8704
#define ES_SLAB (4 * ES_SL_LAYER + 2 * ES_SL_BASIS)

#ifdef ES_TIMING            // tools/escore_phases.py: wall-clock stamps (100 MHz) of workgroup 0 at phase boundaries
static __device__ long long es_stamps[128];   // one array per translation unit (no relocatable device code)
#define ES_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) es_stamps[i] = wall_clock64(); } while (0)
#else
#define ES_STAMP(i)
#endif
