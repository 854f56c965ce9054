// escore_mol.hip — the whole EquivariantScoreNetwork of the 2D->3D model, ONE WORKGROUP PER MOLECULE
// (equivariant_scorenetwork.py:13-40 GATLayer, :121-169 forward; SDE_model_2D_to_3D.py:386-391, :393-445 get_score).
//
// Every extended edge joins two atoms of one molecule (<= 32 atoms, <= 992 edges, hidden size 32), so a molecule's four
// GAT layers, two basis MLPs and the frame mix never need data of another workgroup:
//     per layer   qkvs = x Wqkvs^T + b                   [n, 128]   fp32 MFMA 16x16x4, A from LDS, B (weights) in registers
//                 ee   = edge_attr Wedge^T               [E_m, 32]  fp32 MFMA, A straight from global, result in LDS
//                 attention: one wave per target, lane = 8 * edge slot + head (the recipe of edge_attention_fwd_wave_kernel
//                            reading LDS), softmax in registers, dropout from the counter mask of the operator path
//                 tail: y1 = x + LN1(att); FFN; out = y1 + LN2(.) [+ SiLU]: 8 lanes per atom row (the recipe of gat_tail.hip)
//     per block   Z = [h_src + h_dst | edge_attr] W1^T + b1  [E_m, 128]  fp32 MFMA, each wave owns 32 columns (weights resident
//                 in registers for all edge tiles), SiLU, the 3-wide head as per-lane partial dots + a 16-lane reduction,
//                 frame mix per edge, mean over the in-edges of each atom in edge order (fixed order: bit-reproducible)
// The operator path (msde_edge_attention_*, msde_gat_tail_*, msde_mlp_head_mix_*, ~18 launches forward) stays as the
// cross-check and for shapes this kernel does not take (hidden != 32, heads != 8, basis-MLP width != 128, > 32 atoms).
// Dropout masks are functions of (seed, GLOBAL edge / element index) exactly as in those kernels, so both paths draw the
// same masks from the same seed.
//
// Training: the forward stores, per layer, the rows the backward kernel cannot cheaply rebuild (attention output, y1, h0, x2,
// the layer output: 5 x [N, 32]; the softmax weights [E, 8]); everything per-edge and 128 wide is recomputed there.
#include "msde_common.h"

#define ES_D 32
#define ES_HC 128
#define ES_NMAX 32
#define ES_ECH 384            // edges per attention chunk (whole molecules of <= 20 atoms: one chunk)
#define ES_LDX 36             // LDS row stride of the [., 32] tiles (16-byte aligned rows, conflict-free b128 fragments)
#define ES_LDQ 132            // LDS row stride of qkvs [., 128]
#define ES_LAYERS 4
#define ES_SV 160             // saved floats per (layer, atom): att | y1 | h0 | x2 | out
#define ES_NPTR 52

typedef float es_f4 __attribute__((ext_vector_type(4)));

struct EsW {                  // device pointers, nn.Linear layouts ([out][in])
  const float *Wqkvs[4], *bqkvs[4], *Wedge[4], *ln1g[4], *ln1b[4], *W0[4], *b0[4], *W3[4], *b3[4], *ln2g[4], *ln2b[4];
  const float *bW1[2], *bb1[2], *bW2[2], *bb2[2];
};

__device__ __forceinline__ es_f4 es_mfma(float a, float b, es_f4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float es_sigmoid(float x) {
  float e = expf(-fabsf(x));
  float r = 1.f / (1.f + e);
  return x >= 0.f ? r : e * r;
}
__device__ __forceinline__ void es_ld8(const float* __restrict__ p, float (&v)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ float es_red8_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 8, 64)); v = fmaxf(v, __shfl_xor(v, 16, 64)); return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float es_red8_sum(float v) {
  v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64);
}
__device__ __forceinline__ float es_red16_sum(float v) {      // over the 16 lanes that share lane >> 4
  v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); return v + __shfl_xor(v, 8, 64);
}
__device__ __forceinline__ float es_dot4(float4 a, float4 b, float4 c) {      // a . (b + c), channel order
  return ((a.x * (b.x + c.x) + a.y * (b.y + c.y)) + a.z * (b.z + c.z)) + a.w * (b.w + c.w);
}

// ---- the tail of a GAT layer on 8 lanes per atom row (4 columns per lane), as gat_tail.hip with GT_LPR = 8 --------------
__device__ __forceinline__ float es_row_sum(float s) {
  s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); return s + __shfl_xor(s, 4);
}
__device__ __forceinline__ void es_layernorm(const float (&v)[4], float eps, float& mu, float& rs) {
  mu = es_row_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.f / 32.f);
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) q = fmaf(v[k] - mu, v[k] - mu, q);
  rs = rsqrtf(es_row_sum(q) * (1.f / 32.f) + eps);
}
__device__ __forceinline__ void es_gather(const float (&v)[4], float (&full)[32]) {
  const int base = (threadIdx.x & 63) & ~7;
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) full[r * 4 + c] = __shfl(v[c], base + r);
}
// out[jj] = bias[4q + jj] + sum_k W[4q + jj][k] full[k]; Wp = permuted image [jj][q][k]
__device__ __forceinline__ void es_matvec(const float* __restrict__ Wp, const float* __restrict__ bs, int q,
                                          const float (&full)[32], float (&out)[4]) {
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    float acc = bs[q * 4 + jj];
    const float* wr = Wp + (jj * 8 + q) * 32;
#pragma unroll
    for (int k = 0; k < 32; k += 4) {
      const float4 w = *reinterpret_cast<const float4*>(wr + k);
      acc = fmaf(w.x, full[k], acc); acc = fmaf(w.y, full[k + 1], acc);
      acc = fmaf(w.z, full[k + 2], acc); acc = fmaf(w.w, full[k + 3], acc);
    }
    out[jj] = acc;
  }
}

struct EsMol { int n0, n, e0, Em; };

// qkvs = xs Wqkvs^T + b into LDS (rows >= n hold the bias: finite, never read)
__device__ __forceinline__ void es_qkvs(const float* __restrict__ Wq, const float* __restrict__ bq, const float* xs, float* qk,
                                        int n, int wave, int lane) {
  const int c = lane & 15, g = lane >> 4;
  const int ntile = (n + 15) >> 4;
#pragma unroll
  for (int cti = 0; cti < 2; ++cti) {
    const int col = 16 * (2 * wave + cti) + c;
    float b[8];
    es_ld8(Wq + (size_t)col * ES_D + 8 * g, b);
    const float bias = bq[col];
    for (int rt = 0; rt < ntile; ++rt) {
      float a[8];
      es_ld8(xs + (16 * rt + c) * ES_LDX + 8 * g, a);
      es_f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 8; ++t) acc = es_mfma(a[t], b[t], acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) qk[(16 * rt + 4 * g + r) * ES_LDQ + col] = acc[r] + bias;
    }
  }
}

// ee[0 .. cn) = edge_attr[eg0 .. eg0 + cn) Wedge^T into LDS
__device__ __forceinline__ void es_edge_proj(const float* __restrict__ We, const float* __restrict__ ea, int ld_ea, size_t eg0,
                                             int cn, float* ee, int wave, int lane) {
  const int c = lane & 15, g = lane >> 4, ct = wave & 1;
  float b[8];
  es_ld8(We + (size_t)(16 * ct + c) * ES_D + 8 * g, b);
  const int ntile = (cn + 15) >> 4;
  for (int rt = wave >> 1; rt < ntile; rt += 2) {
    const int el = 16 * rt + c;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (el < cn) es_ld8(ea + (eg0 + el) * ld_ea + 8 * g, a);
    es_f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 8; ++t) acc = es_mfma(a[t], b[t], acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) ee[(16 * rt + 4 * g + r) * ES_LDX + 16 * ct + c] = acc[r];
  }
}

// targets [t0, t1) whose in-edges [rp[t0], rp[t1]) fit one chunk of ES_ECH edges
__device__ __forceinline__ int es_chunk_end(const int* rp, int t0, int n) {
  int t1 = t0 + 1;
  while (t1 < n && rp[t1 + 1] - rp[t0] <= ES_ECH) ++t1;
  return t1;
}

template <bool TRAIN>
__global__ void __launch_bounds__(256)
escore_mol_fwd_kernel(EsW W, const float* __restrict__ x0, const float* __restrict__ ea, int ld_ea,
                      const float* __restrict__ basis, const int* __restrict__ mol_ptr, int B, const int* __restrict__ rowptr,
                      const int* __restrict__ src, const int* __restrict__ dst, int N, float p_att, float p_ffn,
                      unsigned long long seed0, const unsigned long long* __restrict__ seed_dev, float eps1, float eps2,
                      float* __restrict__ out, float* __restrict__ sv, float* __restrict__ alpha_sv, int E_total) {
  __shared__ __attribute__((aligned(16))) float xs[ES_NMAX * ES_LDX];
  __shared__ __attribute__((aligned(16))) float att[ES_NMAX * ES_LDX];
  __shared__ __attribute__((aligned(16))) float qk[ES_NMAX * ES_LDQ];
  __shared__ __attribute__((aligned(16))) float ee[ES_ECH * ES_LDX];       // basis phase: per-wave partial head sums
  __shared__ __attribute__((aligned(16))) float al[ES_ECH * 8];            // basis phase: per-edge mixed vectors
  __shared__ __attribute__((aligned(16))) float W0p[ES_D * ES_D], W3p[ES_D * ES_D];
  __shared__ float prm[6 * ES_D];
  __shared__ float gacc[ES_NMAX * 3];
  __shared__ int rp[ES_NMAX + 1];
  __shared__ int srcl[ES_ECH];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if ((int)blockIdx.x >= B) {
    // rows behind the last molecule (capacity padding): finite outputs
    const int nt = mol_ptr[B];
    for (int t = nt * 3 + tid; t < N * 3; t += 256) out[t] = 0.f;
    return;
  }
  const int n0 = mol_ptr[blockIdx.x], n = min(mol_ptr[blockIdx.x + 1] - n0, ES_NMAX);
  if (n <= 0) return;
  const int e0 = rowptr[n0], Em = rowptr[n0 + n] - e0;
  unsigned long long sdev = seed_dev ? seed_dev[0] * 0x100000001B3ull : 0ull;
  for (int t = tid; t <= n; t += 256) rp[t] = rowptr[n0 + t] - e0;
  for (int t = tid; t < ES_NMAX * 8; t += 256) {          // xs <- node_attr rows (zero beyond n), one float4 per thread
    const int row = t >> 3, q = t & 7;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < n) v = *reinterpret_cast<const float4*>(x0 + (size_t)(n0 + row) * ES_D + 4 * q);
    *reinterpret_cast<float4*>(xs + row * ES_LDX + 4 * q) = v;
  }
  if (tid < ES_NMAX * 3) gacc[tid] = 0.f;
  __syncthreads();

  for (int layer = 0; layer < ES_LAYERS; ++layer) {
    const int mi = layer >> 1, ci = layer & 1;
    // tail weights of this layer -> LDS (read after two barriers)
    for (int t = tid; t < ES_D * ES_D; t += 256) {
      const int j = t >> 5, k = t & 31;
      const int d = ((j & 3) * 8 + (j >> 2)) * ES_D + k;
      W0p[d] = W.W0[layer][t]; W3p[d] = W.W3[layer][t];
    }
    if (tid < ES_D) {
      prm[tid] = W.ln1g[layer][tid]; prm[ES_D + tid] = W.ln1b[layer][tid]; prm[2 * ES_D + tid] = W.b0[layer][tid];
      prm[3 * ES_D + tid] = W.b3[layer][tid]; prm[4 * ES_D + tid] = W.ln2g[layer][tid]; prm[5 * ES_D + tid] = W.ln2b[layer][tid];
    }
    es_qkvs(W.Wqkvs[layer], W.bqkvs[layer], xs, qk, n, wave, lane);
    const unsigned long long seed_l = seed0 + (unsigned long long)(mi * 4 + ci);
    const unsigned long long seed_att = seed_l + sdev, seed_ffn = (seed_l ^ 0x46464Eull) + sdev;
    const float keep_att = p_att > 0.f ? 1.f / (1.f - p_att) : 1.f;
    for (int t0 = 0; t0 < n;) {
      const int t1 = es_chunk_end(rp, t0, n);
      const int ce0 = rp[t0], cn = rp[t1] - ce0;
      es_edge_proj(W.Wedge[layer], ea, ld_ea, (size_t)e0 + ce0, cn, ee, wave, lane);
      for (int t = tid; t < cn; t += 256) srcl[t] = src[e0 + ce0 + t] - n0;
      __syncthreads();
      {
        const int h = lane & 7, l = lane >> 3;
        for (int i = t0 + wave; i < t1; i += 4) {
          const float4 q4 = *reinterpret_cast<const float4*>(qk + i * ES_LDQ + h * 4);
          const int s0 = rp[i] - ce0, s1 = rp[i + 1] - ce0;
          float m = -INFINITY;
          for (int e = s0 + l; e < s1; e += 8) {
            const float4 k4 = *reinterpret_cast<const float4*>(qk + srcl[e] * ES_LDQ + ES_D + h * 4);
            const float4 e4 = *reinterpret_cast<const float4*>(ee + e * ES_LDX + h * 4);
            const float sc = es_dot4(q4, k4, e4) * 0.5f;
            al[e * 8 + h] = sc;
            m = fmaxf(m, sc);
          }
          m = es_red8_max(m);
          float sum = 0.f;
          for (int e = s0 + l; e < s1; e += 8) {
            const float p = expf(al[e * 8 + h] - m);
            al[e * 8 + h] = p;
            sum += p;
          }
          sum = es_red8_sum(sum);
          const float inv = 1.f / (sum + 1e-16f);
          float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
          for (int e = s0 + l; e < s1; e += 8) {
            const float4 v4 = *reinterpret_cast<const float4*>(qk + srcl[e] * ES_LDQ + 2 * ES_D + h * 4);
            const float4 e4 = *reinterpret_cast<const float4*>(ee + e * ES_LDX + h * 4);
            float a = al[e * 8 + h] * inv;
            const unsigned long long ge = (unsigned long long)(e0 + ce0 + e);
            if (TRAIN) alpha_sv[((size_t)layer * E_total + ge) * 8 + h] = a;
            if (p_att > 0.f) a = (msde_uniform(seed_att, ge * 8 + h) >= p_att) ? a * keep_att : 0.f;
            acc.x = fmaf(a, v4.x + e4.x, acc.x); acc.y = fmaf(a, v4.y + e4.y, acc.y);
            acc.z = fmaf(a, v4.z + e4.z, acc.z); acc.w = fmaf(a, v4.w + e4.w, acc.w);
          }
          acc.x = es_red8_sum(acc.x); acc.y = es_red8_sum(acc.y); acc.z = es_red8_sum(acc.z); acc.w = es_red8_sum(acc.w);
          if (l == 0) {
            const float4 s4 = *reinterpret_cast<const float4*>(qk + i * ES_LDQ + 3 * ES_D + h * 4);   // + lin_skip(x_i)
            acc.x += s4.x; acc.y += s4.y; acc.z += s4.z; acc.w += s4.w;
            *reinterpret_cast<float4*>(att + i * ES_LDX + h * 4) = acc;
          }
        }
      }
      __syncthreads();
      t0 = t1;
    }
    // tail: 8 lanes per atom row
    {
      const int row = tid >> 3, q = tid & 7;
      const bool live = row < n;
      const int rr = live ? row : 0;
      const float scale = p_ffn > 0.f ? 1.f / (1.f - p_ffn) : 1.f;
      float v[4], y1[4], h[4], full[32];
      {
        const float4 a = *reinterpret_cast<const float4*>(att + rr * ES_LDX + 4 * q);
        const float4 r4 = *reinterpret_cast<const float4*>(xs + rr * ES_LDX + 4 * q);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; y1[0] = r4.x; y1[1] = r4.y; y1[2] = r4.z; y1[3] = r4.w;
      }
      float* svr = TRAIN ? sv + ((size_t)layer * N + n0 + rr) * ES_SV + 4 * q : nullptr;
      if (TRAIN && live) *reinterpret_cast<float4*>(svr) = make_float4(v[0], v[1], v[2], v[3]);
      float mu, rs;
      es_layernorm(v, eps1, mu, rs);
#pragma unroll
      for (int k = 0; k < 4; ++k) y1[k] += fmaf((v[k] - mu) * rs, prm[q * 4 + k], prm[ES_D + q * 4 + k]);
      es_gather(y1, full);
      es_matvec(W0p, prm + 2 * ES_D, q, full, h);
      if (TRAIN && live) {
        *reinterpret_cast<float4*>(svr + 32) = make_float4(y1[0], y1[1], y1[2], y1[3]);
        *reinterpret_cast<float4*>(svr + 64) = make_float4(h[0], h[1], h[2], h[3]);
      }
      const unsigned long long off = (unsigned long long)(n0 + rr) * ES_D + q * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float s = h[j] * es_sigmoid(h[j]);
        if (p_ffn > 0.f) s = msde_uniform(seed_ffn, off + j) >= p_ffn ? s * scale : 0.f;
        h[j] = s;
      }
      es_gather(h, full);
      es_matvec(W3p, prm + 3 * ES_D, q, full, v);      // v = x2
      if (TRAIN && live) *reinterpret_cast<float4*>(svr + 96) = make_float4(v[0], v[1], v[2], v[3]);
      es_layernorm(v, eps2, mu, rs);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float o = y1[k] + fmaf((v[k] - mu) * rs, prm[4 * ES_D + q * 4 + k], prm[5 * ES_D + q * 4 + k]);
        v[k] = ci == 0 ? o * es_sigmoid(o) : o;        // SiLU between the two convolutions of a block (:142)
      }
      if (live) {
        *reinterpret_cast<float4*>(xs + row * ES_LDX + 4 * q) = make_float4(v[0], v[1], v[2], v[3]);
        if (TRAIN) *reinterpret_cast<float4*>(svr + 128) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
    __syncthreads();
    if (ci == 1) {
      // basis MLP of block mi on every edge + frame mix + mean over the in-edges of the target (:150-166)
      const int c = lane & 15, g = lane >> 4;
      float b[2][16], bias1[2], w2[2][3];
#pragma unroll
      for (int cti = 0; cti < 2; ++cti) {
        const int col = 32 * wave + 16 * cti + c;
        const float* wr = W.bW1[mi] + (size_t)col * (2 * ES_D) + 16 * g;
        es_ld8(wr, *reinterpret_cast<float(*)[8]>(&b[cti][0]));
        es_ld8(wr + 8, *reinterpret_cast<float(*)[8]>(&b[cti][8]));
        bias1[cti] = W.bb1[mi][col];
#pragma unroll
        for (int k = 0; k < 3; ++k) w2[cti][k] = W.bW2[mi][k * ES_HC + col];
      }
      const int ntile = (Em + 15) >> 4, Ep = ntile * 16;
      float* part = ee;                                 // [4 waves][Ep][3]
      for (int rt = 0; rt < ntile; ++rt) {
        const int el = 16 * rt + c;
        float a[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) a[t] = 0.f;
        if (el < Em) {
          const size_t ge = (size_t)e0 + el;
          if (g < 2) {
            const float* pj = xs + (src[ge] - n0) * ES_LDX + 16 * g;
            const float* pi = xs + (dst[ge] - n0) * ES_LDX + 16 * g;
#pragma unroll
            for (int t = 0; t < 16; t += 4) {
              const float4 u = *reinterpret_cast<const float4*>(pj + t), w = *reinterpret_cast<const float4*>(pi + t);
              a[t] = u.x + w.x; a[t + 1] = u.y + w.y; a[t + 2] = u.z + w.z; a[t + 3] = u.w + w.w;
            }
          } else {
            const float* pe = ea + ge * ld_ea + 16 * (g - 2);
            es_ld8(pe, *reinterpret_cast<float(*)[8]>(&a[0]));
            es_ld8(pe + 8, *reinterpret_cast<float(*)[8]>(&a[8]));
          }
        }
        es_f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 16; ++t) { acc0 = es_mfma(a[t], b[0][t], acc0); acc1 = es_mfma(a[t], b[1][t], acc1); }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float z0 = acc0[r] + bias1[0], z1 = acc1[r] + bias1[1];
          const float a0 = z0 * es_sigmoid(z0), a1 = z1 * es_sigmoid(z1);
          const float p0 = es_red16_sum(fmaf(a0, w2[0][0], a1 * w2[1][0]));
          const float p1 = es_red16_sum(fmaf(a0, w2[0][1], a1 * w2[1][1]));
          const float p2 = es_red16_sum(fmaf(a0, w2[0][2], a1 * w2[1][2]));
          if (c == 0) {
            float* pp = part + ((size_t)wave * Ep + 16 * rt + 4 * g + r) * 3;
            pp[0] = p0; pp[1] = p1; pp[2] = p2;
          }
        }
      }
      __syncthreads();
      const float b20 = W.bb2[mi][0], b21 = W.bb2[mi][1], b22 = W.bb2[mi][2];
      for (int e = tid; e < Em; e += 256) {
        float cf[3];
#pragma unroll
        for (int k = 0; k < 3; ++k)
          cf[k] = (((part[(size_t)e * 3 + k] + part[((size_t)Ep + e) * 3 + k]) + part[((size_t)2 * Ep + e) * 3 + k]) +
                   part[((size_t)3 * Ep + e) * 3 + k]) + (k == 0 ? b20 : k == 1 ? b21 : b22);
        const float* bs = basis + 9 * ((size_t)e0 + e);
        al[e * 3] = (cf[0] * bs[0] + cf[1] * bs[3]) + cf[2] * bs[6];
        al[e * 3 + 1] = (cf[0] * bs[1] + cf[1] * bs[4]) + cf[2] * bs[7];
        al[e * 3 + 2] = (cf[0] * bs[2] + cf[1] * bs[5]) + cf[2] * bs[8];
      }
      __syncthreads();
      if (tid < n * 3) {
        const int i = tid / 3, k = tid - 3 * i;
        const int s0 = rp[i], s1 = rp[i + 1];
        float s = 0.f;
        for (int e = s0; e < s1; ++e) s += al[e * 3 + k];
        gacc[tid] += s * (1.f / (float)max(s1 - s0, 1));
      }
      __syncthreads();
    }
  }
  if (tid < n * 3) out[(size_t)n0 * 3 + tid] = gacc[tid];
}

extern "C" long long msde_escore_mol_saved_floats(int N) { return (long long)ES_LAYERS * (long long)N * ES_SV; }

// params: HOST array of ES_NPTR device pointers in the order of struct EsW (per field: the 4 layers / the 2 blocks)
extern "C" int msde_escore_mol_fwd(const void* const* params, const float* x0, const float* edge_attr, int ld_ea,
                                   const float* basis, const int* mol_ptr, int B, const int* rowptr, const int* src,
                                   const int* dst, int N, int E, int hidden, int heads, int hidden_coff, float p_att,
                                   float p_ffn, unsigned long long seed0, const unsigned long long* seed_dev, float eps1,
                                   float eps2, float* out, float* saved, float* alpha_saved, void* stream) {
  if (!params || !x0 || !edge_attr || !basis || !mol_ptr || !rowptr || !src || !dst || !out || N < 0 || B < 0 || E < 0)
    return MSDE_EINVAL;
  if (hidden != ES_D || heads != 8 || hidden_coff != ES_HC) return MSDE_EUNSUP;
  if (ld_ea < ES_D || ld_ea % 4 || (reinterpret_cast<uintptr_t>(edge_attr) & 15) || (reinterpret_cast<uintptr_t>(x0) & 15))
    return MSDE_EINVAL;
  if (p_att < 0.f || p_att >= 1.f || p_ffn < 0.f || p_ffn >= 1.f) return MSDE_EINVAL;
  if ((saved == nullptr) != (alpha_saved == nullptr)) return MSDE_EINVAL;
  if (saved && (reinterpret_cast<uintptr_t>(saved) & 15)) return MSDE_EINVAL;
  EsW W;
  const float** wp = reinterpret_cast<const float**>(&W);
  for (int i = 0; i < ES_NPTR; ++i) {
    if (!params[i]) return MSDE_EINVAL;
    wp[i] = static_cast<const float*>(params[i]);
  }
  if (N == 0 || B == 0) return 0;
  if (saved)
    MSDE_LAUNCH(escore_mol_fwd_kernel<true>, dim3(B + 1), dim3(256), 0, as_stream(stream), W, x0, edge_attr, ld_ea, basis,
                mol_ptr, B, rowptr, src, dst, N, p_att, p_ffn, seed0, seed_dev, eps1, eps2, out, saved, alpha_saved, E);
  else
    MSDE_LAUNCH(escore_mol_fwd_kernel<false>, dim3(B + 1), dim3(256), 0, as_stream(stream), W, x0, edge_attr, ld_ea, basis,
                mol_ptr, B, rowptr, src, dst, N, p_att, p_ffn, seed0, seed_dev, eps1, eps2, out, saved, alpha_saved, E);
  MSDE_CHECK_LAUNCH();
  return 0;
}
