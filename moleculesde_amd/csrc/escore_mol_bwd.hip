// escore_mol_bwd.hip — backward of the one-workgroup-per-molecule EquivariantScoreNetwork (escore_mol.hip), same launch
// shape: input gradients w.r.t. the node features [N, 32] and the edge features [E, 32], and EVERY weight gradient of the
// network as one slab per workgroup (ES_SLAB floats; summed over workgroups by the caller's batched slab reduction in a fixed
// order).  Replaces the ~28 backward launches of the operator path (msde_mlp_head_mix_bwd, gemm dgrad/wgrad of the basis MLP,
// msde_gat_tail_bwd, msde_edge_attention_bwd, segment sums, projections' dgrad/wgrad; equivariant_scorenetwork.py:13-40,121-169).
//
// Per-atom rows the forward saved (attention output, y1, h0, x2, layer output, softmax max and 1/sum) are read back; everything
// per edge (lin_edge rows, scores, the basis MLP's hidden rows) is recomputed on chip.  Matrix products run on
// v_mfma_f32_16x16x4_f32; an accumulator tile (lane = column, registers = rows) can be contracted over its ROW index without
// moving data, which decides the orientation of every product below:
//   basis MLP, sweep A (a wave owns 32 hidden columns, walks all edge tiles): Z [edge, hid] -> gZ; contracted over edges with
//     edge_attr^T (-> gW1[:, 32:]^T), with the atom-incidence matrix (-> gP: the per-atom sum of gZ over incident edges) and, as
//     SiLU(Z), with gcoff^T (-> gW2);
//   basis MLP, sweep B (a wave owns every fourth edge tile, all 128 hidden rows): Z^T [hid, edge] -> gZ^T, contracted over the
//     hidden index with W1[:, 32:]^T (-> g_edge_attr^T);
//   attention: lane = (target, head) for the softmax backward, lane = (source, head) for the key / value gradients (by-source
//     lists: fixed order, no atomics); the gradient of the lin_edge rows is rank one per head and is rebuilt in operand layout
//     from two scalars per (edge, head).
#include "escore_mol.h"

#define EB_SCR 9984            // floats of phase scratch: {ee 6912 | gs 1536 | am 1536} / {5 row tiles + LN partials} / {gcoff [992][4]}

__device__ __forceinline__ float eb_rows_sum(float v) {   // over the 8 atom rows of a wave (lanes with equal lane & 7)
  v += es_dpp<0x128>(v);                                   // row_ror:8 (xor 8 inside a row of 16)
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

#ifdef ES_TIMING
extern "C" int msde_escore_debug_stamps_bwd(long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(es_stamps), sizeof(long long) * 128);
}
#endif

__global__ void __launch_bounds__(256)
escore_mol_bwd_kernel(EsW W, const float* __restrict__ x0, const float* __restrict__ ea, int ld_ea,
                      const float* __restrict__ basis, const int* __restrict__ mol_ptr, int B, const int* __restrict__ rowptr,
                      const int* __restrict__ src, const int* __restrict__ dst, const int* __restrict__ rowptr_s,
                      const int* __restrict__ perm_s, int N, int E, float p_att, float p_ffn, unsigned long long seed0,
                      const unsigned long long* __restrict__ seed_dev, float eps1, float eps2, const float* __restrict__ sv,
                      const float* __restrict__ g_out, float* __restrict__ g_x0, float* __restrict__ g_ea, int ld_gea,
                      float* __restrict__ slabs) {
  __shared__ __attribute__((aligned(16))) float xs[ES_NMAX * ES_LDX];       // layer input X_l (basis phase: block output H)
  __shared__ __attribute__((aligned(16))) float gx[ES_NMAX * ES_LDX];       // gradient w.r.t. the layer output, then input
  __shared__ __attribute__((aligned(16))) float ga[ES_NMAX * ES_LDX];       // gradient w.r.t. the attention output
  __shared__ __attribute__((aligned(16))) float qk[ES_NMAX * ES_LDQ];       // q|k|v|skip (basis phase: P)
  __shared__ __attribute__((aligned(16))) float gqk[ES_NMAX * ES_LDQ];      // their gradients (basis phase: gP)
  __shared__ __attribute__((aligned(16))) float eal[ES_EAL * ES_LDX];       // edge features of the molecule (Em <= ES_EAL)
  __shared__ __attribute__((aligned(16))) float scr[EB_SCR];
  __shared__ __attribute__((aligned(16))) float hw[ES_HC + 3 * ES_HC];      // basis phase: b1 | W2
  __shared__ float prm[3 * ES_D];                                           // ln1_g | ln2_g | ln2_b of the layer
  __shared__ float gG[ES_NMAX * 3];                                         // g_out / in-degree
  __shared__ int rp[ES_NMAX + 1], rps[ES_NMAX + 1];
  __shared__ unsigned char sl[ES_EMAX + 16], dl[ES_EMAX + 16];
  __shared__ unsigned short es[ES_EMAX + 16];                               // by-source slot -> molecule-local edge
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 15, g = lane >> 4;
  const int nt = mol_ptr[B];
  {
    // capacity padding: zero gradients for atoms behind the last molecule and edge slots behind the last edge
    for (int t = nt * ES_D + (int)blockIdx.x * 256 + tid; t < N * ES_D; t += B * 256) g_x0[t] = 0.f;
    const int Et = rowptr[nt];
    for (int t = Et * 8 + (int)blockIdx.x * 256 + tid; t < E * 8; t += B * 256)
      *reinterpret_cast<float4*>(g_ea + (size_t)(t >> 3) * ld_gea + 4 * (t & 7)) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float* slab = slabs + (size_t)blockIdx.x * ES_SLAB;
  const int n0 = mol_ptr[blockIdx.x], n = min(mol_ptr[blockIdx.x + 1] - n0, ES_NMAX);
  if (n <= 0) {
    for (int t = tid; t < ES_SLAB; t += 256) slab[t] = 0.f;
    return;
  }
  const int e0 = rowptr[n0], Em = min(rowptr[n0 + n] - e0, ES_EMAX);
  const bool ea_lds = Em <= ES_EAL;
  const unsigned long long sdev = seed_dev ? seed_dev[0] * 0x100000001B3ull : 0ull;
  for (int t = tid; t <= n; t += 256) { rp[t] = rowptr[n0 + t] - e0; rps[t] = rowptr_s[n0 + t] - e0; }
  for (int t = tid; t < Em; t += 256) {
    sl[t] = (unsigned char)(src[e0 + t] - n0);
    dl[t] = (unsigned char)(dst[e0 + t] - n0);
    es[t] = (unsigned short)(perm_s[e0 + t] - e0);
  }
  const int row = tid >> 3, q = tid & 7;
  const bool live = row < n;
  {
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(gx + row * ES_LDX + 4 * q) = z4;
    for (int t = tid; t < ES_NMAX * ES_LDQ / 4; t += 256) reinterpret_cast<float4*>(gqk)[t] = z4;
    if (ea_lds)
      for (int r = tid >> 3; r < Em; r += 32)
        *reinterpret_cast<float4*>(eal + r * ES_LDX + 4 * q) = *reinterpret_cast<const float4*>(ea + ((size_t)e0 + r) * ld_ea + 4 * q);
  }
  if (tid < n * 3) {
    const int i = tid / 3;
    gG[tid] = g_out[(size_t)n0 * 3 + tid] / (float)max(rowptr[n0 + i + 1] - rowptr[n0 + i], 1);
  }
  __syncthreads();
  // A operand row of edge `e` (8 consecutive features starting at k0) from LDS or, for big molecules, from global
  auto ea_row8 = [&](int e, int k0, float (&a)[8]) {
    if (ea_lds) es_ld8(eal + e * ES_LDX + k0, a);
    else es_ld8(ea + ((size_t)e0 + e) * ld_ea + k0, a);
  };
  auto ea_at = [&](int e, int k) -> float { return ea_lds ? eal[e * ES_LDX + k] : ea[((size_t)e0 + e) * ld_ea + k]; };
  ES_STAMP(64);
  bool gea_first = true;          // the first pass over g_edge_attr stores, the later ones accumulate

#pragma unroll 1
  for (int layer = ES_LAYERS - 1; layer >= 0; --layer) {
    const int mi = layer >> 1, ci = layer & 1;
    float* sl_l = slab + layer * ES_SL_LAYER;
    if (ci == 1) {
      // ================= basis MLP of block mi: backward =================================================================
      float* sb = slab + 4 * ES_SL_LAYER + mi * ES_SL_BASIS;
      float* gc = scr;                                  // [Em][4]: gradient of the three coefficients of every edge
      const float* W1 = W.bW1(mi);
      {
        // H = output of this layer (saved) -> xs
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live) v = *reinterpret_cast<const float4*>(sv + ((size_t)layer * N + n0 + row) * ES_SV + 128 + 4 * q);
        *reinterpret_cast<float4*>(xs + row * ES_LDX + 4 * q) = v;
        if (tid < ES_HC) hw[tid] = W.bb1(mi)[tid];
        hw[ES_HC + tid] = W.bW2(mi)[tid];
        if (tid < ES_HC) hw[ES_HC + 256 + tid] = W.bW2(mi)[256 + tid];
        for (int e = tid; e < Em; e += 256) {
          const float* bs = basis + 9 * ((size_t)e0 + e);
          const int i = dl[e];
          const float gx_ = gG[3 * i], gy_ = gG[3 * i + 1], gz_ = gG[3 * i + 2];
          *reinterpret_cast<float4*>(gc + 4 * e) = make_float4(gx_ * bs[0] + gy_ * bs[1] + gz_ * bs[2], gx_ * bs[3] + gy_ * bs[4] + gz_ * bs[5],
                                                                gx_ * bs[6] + gy_ * bs[7] + gz_ * bs[8], 0.f);
        }
      }
      __syncthreads();
      ES_STAMP(65 + 14 * (3 - layer));
      {
        // P = H W1[:, :32]^T + b1 / 2 -> qk (as the forward); gb2 = sum over edges of gcoff (wave 0, fixed tree)
        float wp[2][8];
#pragma unroll
        for (int cti = 0; cti < 2; ++cti) es_ld8(W1 + (size_t)(32 * wave + 16 * cti + c) * (2 * ES_D) + 8 * g, wp[cti]);
        const int ntile_n = (n + 15) >> 4;
        const float hb0 = 0.5f * hw[32 * wave + c], hb1 = 0.5f * hw[32 * wave + 16 + c];
        for (int rt = 0; rt < ntile_n; ++rt) {
          float a[8];
          es_ld8(xs + (16 * rt + c) * ES_LDX + 8 * g, a);
          es_f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int t = 0; t < 8; ++t) { acc0 = es_mfma(a[t], wp[0][t], acc0); acc1 = es_mfma(a[t], wp[1][t], acc1); }
          float* o = qk + (16 * rt + 4 * g) * ES_LDQ + 32 * wave + c;
#pragma unroll
          for (int r = 0; r < 4; ++r) { o[r * ES_LDQ] = acc0[r] + hb0; o[r * ES_LDQ + 16] = acc1[r] + hb1; }
        }
        if (wave == 0) {
          float s0 = 0.f, s1 = 0.f, s2 = 0.f;
          for (int e = lane; e < Em; e += 64) { const float4 v = *reinterpret_cast<const float4*>(gc + 4 * e); s0 += v.x; s1 += v.y; s2 += v.z; }
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) { s0 += __shfl_xor(s0, o, 64); s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
          if (lane == 0) { sb[ES_SL_B2] = s0; sb[ES_SL_B2 + 1] = s1; sb[ES_SL_B2 + 2] = s2; sb[ES_SL_B2 + 3] = 0.f; }
        }
      }
      __syncthreads();
      ES_STAMP(66 + 14 * (3 - layer));
      const int ntile = (Em + 15) >> 4;
      {
        // ---- sweep A: the wave's 32 hidden columns, all edge tiles ----
        float bz[2][8], w2c[2][3];
#pragma unroll
        for (int cti = 0; cti < 2; ++cti) {
          const int hid = 32 * wave + 16 * cti + c;
          es_ld8(W1 + (size_t)hid * (2 * ES_D) + ES_D + 8 * g, bz[cti]);
#pragma unroll
          for (int k = 0; k < 3; ++k) w2c[cti][k] = hw[ES_HC + k * ES_HC + hid];
        }
        es_f4 gW1bT[2][2], gPn[2][2], gW2a[2];
        float gb1p[2] = {0.f, 0.f};
#pragma unroll
        for (int a_ = 0; a_ < 2; ++a_) {
          gW2a[a_] = es_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int b_ = 0; b_ < 2; ++b_) { gW1bT[a_][b_] = es_f4{0.f, 0.f, 0.f, 0.f}; gPn[a_][b_] = es_f4{0.f, 0.f, 0.f, 0.f}; }
        }
        for (int rt = 0; rt < ntile; ++rt) {
          // ---- every operand of the tile is requested first (none depends on a product) ----
          const int em = 16 * rt + c;
          float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          if (em < Em) ea_row8(em, 8 * g, a);
          float eaT0[4], eaT1[4], gct[4], inc0[4], inc1[4];
          float4 gcr[4];
          bool on[4];
          // (Z's initial value P[src] + P[dst] by 16 LDS reads: as an incidence-row x P product -- 16 more MFMAs per wave and
          // tile -- the sweep was 28.6 instead of 24.2 us for 166 edges)
          es_f4 z[2];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int e = 16 * rt + 4 * g + t;
            on[t] = e < Em;
            const int ec = on[t] ? e : 0;
            {
              const int sj_ = on[t] ? sl[ec] : 0, si_ = on[t] ? dl[ec] : 0;
              z[0][t] = qk[sj_ * ES_LDQ + 32 * wave + c] + qk[si_ * ES_LDQ + 32 * wave + c];
              z[1][t] = qk[sj_ * ES_LDQ + 32 * wave + 16 + c] + qk[si_ * ES_LDQ + 32 * wave + 16 + c];
            }
            eaT0[t] = on[t] ? ea_at(ec, c) : 0.f;                     // edge_attr^T: kin = c
            eaT1[t] = on[t] ? ea_at(ec, 16 + c) : 0.f;                //               kin = 16 + c
            const int sje = on[t] ? sl[ec] : 255, sie = on[t] ? dl[ec] : 255;
            inc0[t] = (float)(sje == c) + (float)(sie == c);
            inc1[t] = (float)(sje == 16 + c) + (float)(sie == 16 + c);
            gcr[t] = on[t] ? *reinterpret_cast<const float4*>(gc + 4 * ec) : make_float4(0.f, 0.f, 0.f, 0.f);
            gct[t] = c == 0 ? gcr[t].x : c == 1 ? gcr[t].y : c == 2 ? gcr[t].z : 0.f;   // gcoff^T: k = c
          }
          // ---- Z rows of this tile for the wave's columns: P[src] + P[dst] + edge_attr x W1[:, 32:]^T ----
#pragma unroll
          for (int t = 0; t < 8; ++t) { z[0] = es_mfma(a[t], bz[0][t], z[0]); z[1] = es_mfma(a[t], bz[1][t], z[1]); }
          es_f4 S[2], gZ[2];
#pragma unroll
          for (int cti = 0; cti < 2; ++cti)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float zz = z[cti][r], s = es_sigmoid(zz);
              const float gS = (gcr[r].x * w2c[cti][0] + gcr[r].y * w2c[cti][1]) + gcr[r].z * w2c[cti][2];
              S[cti][r] = on[r] ? zz * s : 0.f;
              gZ[cti][r] = on[r] ? gS * es_dsilu(zz, s) : 0.f;
              gb1p[cti] += gZ[cti][r];
            }
          // ---- contraction over the tile's 16 edges (k = 4 g + t) ----
#pragma unroll
          for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int cti = 0; cti < 2; ++cti) {
              gW1bT[0][cti] = es_mfma(eaT0[t], gZ[cti][t], gW1bT[0][cti]);
              gW1bT[1][cti] = es_mfma(eaT1[t], gZ[cti][t], gW1bT[1][cti]);
              gPn[0][cti] = es_mfma(inc0[t], gZ[cti][t], gPn[0][cti]);
              gPn[1][cti] = es_mfma(inc1[t], gZ[cti][t], gPn[1][cti]);
              gW2a[cti] = es_mfma(gct[t], S[cti][t], gW2a[cti]);
            }
          }
        }
#pragma unroll
        for (int cti = 0; cti < 2; ++cti) {
          const int hid = 32 * wave + 16 * cti + c;
          // gW1[hid][32 + 16 kt + 4 g + r] = gW1bT[kt][cti][r]
#pragma unroll
          for (int kt = 0; kt < 2; ++kt)
            *reinterpret_cast<float4*>(sb + (size_t)hid * 64 + 32 + 16 * kt + 4 * g) =
                make_float4(gW1bT[kt][cti][0], gW1bT[kt][cti][1], gW1bT[kt][cti][2], gW1bT[kt][cti][3]);
          float b1s = gb1p[cti];
          b1s += __shfl_xor(b1s, 16, 64); b1s += __shfl_xor(b1s, 32, 64);
          if (g == 0) {
            sb[ES_SL_B1 + hid] = b1s;
#pragma unroll
            for (int k = 0; k < 3; ++k) sb[ES_SL_W2 + k * ES_HC + hid] = gW2a[cti][k];
          }
#pragma unroll
          for (int nt_ = 0; nt_ < 2; ++nt_)
#pragma unroll
            for (int r = 0; r < 4; ++r) gqk[(16 * nt_ + 4 * g + r) * ES_LDQ + hid] = gPn[nt_][cti][r];
        }
      }
      __syncthreads();
      ES_STAMP(67 + 14 * (3 - layer));
      {
        // g_H += gP W1[:, :32]  (contraction over the 128 hidden columns): the wave's tile (atom tile wave >> 1, input tile wave & 1)
        const int trt = wave >> 1, tct = wave & 1;
        es_f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          float a[8], b[8];
          es_ld8(gqk + (16 * trt + c) * ES_LDQ + 32 * g + 8 * kk, a);
#pragma unroll
          for (int t = 0; t < 8; ++t) b[t] = W1[(size_t)(32 * g + 8 * kk + t) * (2 * ES_D) + 16 * tct + c];
#pragma unroll
          for (int t = 0; t < 8; ++t) acc = es_mfma(a[t], b[t], acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) gx[(16 * trt + 4 * g + r) * ES_LDX + 16 * tct + c] += acc[r];
        // gW1[hid][kin < 32] = sum over atoms gP[atom][hid] H[atom][kin]: 16 tiles, four per wave
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int tile = 4 * wave + u, ht = tile >> 1, kt = tile & 1;
          const es_f4 d = es_xty(gqk, ES_LDQ, 16 * ht, xs, ES_LDX, 16 * kt, lane);
#pragma unroll
          for (int r = 0; r < 4; ++r) sb[(size_t)(16 * ht + 4 * g + r) * 64 + 16 * kt + c] = d[r];
        }
      }
      ES_STAMP(68 + 14 * (3 - layer));
      {
        // ---- sweep B: every fourth edge tile, all 128 hidden rows: g_edge_attr^T = W1[:, 32:]^T gZ^T ----
        float b1w[8][8];
#pragma unroll
        for (int ht = 0; ht < 8; ++ht) es_ld8(W1 + (size_t)(16 * ht + c) * (2 * ES_D) + ES_D + 8 * g, b1w[ht]);
        // A operands of the second product, resident: W1[hid = 16 ht + 4 g + r][32 + kin], kin = c / 16 + c
        float wga[8][4], wgb[8][4];
#pragma unroll
        for (int ht = 0; ht < 8; ++ht)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float* wr = W1 + (size_t)(16 * ht + 4 * g + r) * (2 * ES_D) + ES_D;
            wga[ht][r] = wr[c]; wgb[ht][r] = wr[16 + c];
          }
        for (int rt = wave; rt < ntile; rt += 4) {
          const int el = 16 * rt + c;
          const bool on = el < Em;
          float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          if (on) ea_row8(el, 8 * g, a);
          const float* pj = qk + (on ? sl[el] : 0) * ES_LDQ + 4 * g;
          const float* pi = qk + (on ? dl[el] : 0) * ES_LDQ + 4 * g;
          const float4 gce = on ? *reinterpret_cast<const float4*>(gc + 4 * el) : make_float4(0.f, 0.f, 0.f, 0.f);
          es_f4 acc[8];
#pragma unroll
          for (int ht = 0; ht < 8; ++ht) {
            const float4 u = *reinterpret_cast<const float4*>(pj + 16 * ht), w = *reinterpret_cast<const float4*>(pi + 16 * ht);
            acc[ht] = es_f4{u.x + w.x, u.y + w.y, u.z + w.z, u.w + w.w};
          }
#pragma unroll
          for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int ht = 0; ht < 8; ++ht) acc[ht] = es_mfma(b1w[ht][t], a[t], acc[ht]);
          es_f4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ht = 0; ht < 8; ++ht) {
            const float4 w0 = *reinterpret_cast<const float4*>(hw + ES_HC + 16 * ht + 4 * g);
            const float4 w1 = *reinterpret_cast<const float4*>(hw + 2 * ES_HC + 16 * ht + 4 * g);
            const float4 w2 = *reinterpret_cast<const float4*>(hw + 3 * ES_HC + 16 * ht + 4 * g);
            const float w0v[4] = {w0.x, w0.y, w0.z, w0.w}, w1v[4] = {w1.x, w1.y, w1.z, w1.w}, w2v[4] = {w2.x, w2.y, w2.z, w2.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float zz = acc[ht][r], s = es_sigmoid(zz);
              const float gS = (gce.x * w0v[r] + gce.y * w1v[r]) + gce.z * w2v[r];
              const float gz = gS * es_dsilu(zz, s);
              o0 = es_mfma(wga[ht][r], gz, o0);
              o1 = es_mfma(wgb[ht][r], gz, o1);
            }
          }
          if (on) {
            float* po = g_ea + ((size_t)e0 + el) * ld_gea + 4 * g;
            float4 v0 = make_float4(o0[0], o0[1], o0[2], o0[3]), v1 = make_float4(o1[0], o1[1], o1[2], o1[3]);
            if (!gea_first) {
              const float4 p0 = *reinterpret_cast<const float4*>(po), p1 = *reinterpret_cast<const float4*>(po + 16);
              v0.x += p0.x; v0.y += p0.y; v0.z += p0.z; v0.w += p0.w; v1.x += p1.x; v1.y += p1.y; v1.z += p1.z; v1.w += p1.w;
            }
            *reinterpret_cast<float4*>(po) = v0;
            *reinterpret_cast<float4*>(po + 16) = v1;
          }
        }
        gea_first = false;
      }
      __syncthreads();
      ES_STAMP(69 + 14 * (3 - layer));
    }

    // ================= GAT layer `layer`: backward ========================================================================
    const unsigned long long seed_l = seed0 + (unsigned long long)(mi * 4 + ci);
    const unsigned long long seed_att = seed_l + sdev, seed_ffn = (seed_l ^ 0x46464Eull) + sdev;
    const float keep_att = p_att > 0.f ? 1.f / (1.f - p_att) : 1.f;
    const float* svl = sv + ((size_t)layer * N + n0) * ES_SV;
    float* t_a = scr;                     // gx2
    float* t_b = scr + 1152;              // go, then g_y1
    float* t_c = scr + 2 * 1152;          // g_h0
    float* t_d = scr + 3 * 1152;          // Dropout(SiLU(h0))
    float* t_e = scr + 4 * 1152;          // y1
    float* lnp = scr + 5 * 1152;          // [4 waves][128] partial LayerNorm-parameter gradients
    if (tid < 3 * ES_D) {
      const int f = tid >> 5, k = tid & 31;
      prm[tid] = (f == 0 ? W.ln1g(layer) : f == 1 ? W.ln2g(layer) : W.ln2b(layer))[k];
    }
    // weights of the layer in operand layouts
    float wq[2][8], bq[2], we[8], w3t[8], w0t[8], wet[8];
    const int trt = wave >> 1, tct = wave & 1, tcol = 16 * tct + c;
    {
#pragma unroll
      for (int cti = 0; cti < 2; ++cti) {
        const int colb = 16 * cti + c;
        es_ld8(W.Wq(layer, wave) + (size_t)colb * ES_D + 8 * g, wq[cti]);
        bq[cti] = W.bq(layer, wave)[colb];
      }
      es_ld8(W.Wedge(layer) + (size_t)(16 * tct + c) * ES_D + 8 * g, we);
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        w3t[t] = W.W3(layer)[(8 * g + t) * ES_D + tcol];
        w0t[t] = W.W0(layer)[(8 * g + t) * ES_D + tcol];
        wet[t] = W.Wedge(layer)[(8 * g + t) * ES_D + tcol];
      }
    }
    float lnacc[16];
    __syncthreads();                      // prm visible; scratch free
    {
      // T4': through out = y1 + LN2(x2) [and the SiLU between the convolutions]; X_l -> xs
      float4 y4 = make_float4(0.f, 0.f, 0.f, 0.f), x4 = y4, xin = y4;
      if (live) {
        y4 = *reinterpret_cast<const float4*>(svl + (size_t)row * ES_SV + 32 + 4 * q);
        x4 = *reinterpret_cast<const float4*>(svl + (size_t)row * ES_SV + 96 + 4 * q);
        xin = layer > 0 ? *reinterpret_cast<const float4*>(svl - (size_t)N * ES_SV + (size_t)row * ES_SV + 128 + 4 * q)
                        : *reinterpret_cast<const float4*>(x0 + (size_t)(n0 + row) * ES_D + 4 * q);
      }
      *reinterpret_cast<float4*>(xs + row * ES_LDX + 4 * q) = xin;
      const float4 g4 = *reinterpret_cast<const float4*>(gx + row * ES_LDX + 4 * q);
      float v[4] = {x4.x, x4.y, x4.z, x4.w}, y1[4] = {y4.x, y4.y, y4.z, y4.w}, go[4] = {g4.x, g4.y, g4.z, g4.w}, xh[4], gg[4];
      float mu, rs;
      es_layernorm(v, eps2, mu, rs);
      float c1 = 0.f, c2 = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        xh[k] = (v[k] - mu) * rs;
        if (ci == 0) {
          const float o = y1[k] + fmaf(xh[k], prm[ES_D + q * 4 + k], prm[2 * ES_D + q * 4 + k]);
          go[k] *= es_dsilu(o, es_sigmoid(o));
        }
        lnacc[k] = go[k] * xh[k];
        lnacc[4 + k] = go[k];
        gg[k] = go[k] * prm[ES_D + q * 4 + k];
        c1 += gg[k]; c2 = fmaf(gg[k], xh[k], c2);
      }
      c1 = es_row8_sum(c1) * (1.f / 32.f); c2 = es_row8_sum(c2) * (1.f / 32.f);
      *reinterpret_cast<float4*>(t_a + row * ES_LDX + 4 * q) =
          make_float4(rs * (gg[0] - c1 - xh[0] * c2), rs * (gg[1] - c1 - xh[1] * c2), rs * (gg[2] - c1 - xh[2] * c2), rs * (gg[3] - c1 - xh[3] * c2));
      *reinterpret_cast<float4*>(t_b + row * ES_LDX + 4 * q) = make_float4(go[0], go[1], go[2], go[3]);
      *reinterpret_cast<float4*>(t_e + row * ES_LDX + 4 * q) = y4;
    }
    __syncthreads();
    ES_STAMP(70 + 14 * (3 - layer));
    {
      // T3': g_a = gx2 W3; through Dropout and SiLU -> g_h0 (t_c); the activation itself -> t_d
      float a[8];
      es_ld8(t_a + (16 * trt + c) * ES_LDX + 8 * g, a);
      es_f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 8; ++t) acc = es_mfma(a[t], w3t[t], acc);
      const float scale = p_ffn > 0.f ? 1.f / (1.f - p_ffn) : 1.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rw = 16 * trt + 4 * g + r;
        float act = 0.f, gh = 0.f;
        if (rw < n) {
          const float h0 = svl[(size_t)rw * ES_SV + 64 + tcol];
          const float s = es_sigmoid(h0);
          float mk = 1.f;
          if (p_ffn > 0.f) mk = msde_uniform(seed_ffn, (unsigned long long)(n0 + rw) * ES_D + tcol) >= p_ffn ? scale : 0.f;
          act = h0 * s * mk;
          gh = acc[r] * mk * es_dsilu(h0, s);
        }
        t_d[rw * ES_LDX + tcol] = act;
        t_c[rw * ES_LDX + tcol] = gh;
      }
    }
    __syncthreads();
    {
      // T2': g_y1 = go + g_h0 W0 (in place over go); gW3 = gx2^T a; gW0 = g_h0^T y1; gb3, gb0
      float a[8];
      es_ld8(t_c + (16 * trt + c) * ES_LDX + 8 * g, a);
      es_f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 8; ++t) acc = es_mfma(a[t], w0t[t], acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) t_b[(16 * trt + 4 * g + r) * ES_LDX + tcol] += acc[r];
      const es_f4 d3 = es_xty(t_a, ES_LDX, 16 * trt, t_d, ES_LDX, 16 * tct, lane);
      const es_f4 d0 = es_xty(t_c, ES_LDX, 16 * trt, t_e, ES_LDX, 16 * tct, lane);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sl_l[ES_SL_W3 + (16 * trt + 4 * g + r) * ES_D + tcol] = d3[r];
        sl_l[ES_SL_W0 + (16 * trt + 4 * g + r) * ES_D + tcol] = d0[r];
      }
      if (tid < 2 * ES_D) {
        const float* src_ = tid < ES_D ? t_a : t_c;
        const int col = tid & 31;
        float s = 0.f;
        for (int r = 0; r < ES_NMAX; ++r) s += src_[r * ES_LDX + col];
        sl_l[(tid < ES_D ? ES_SL_B3 : ES_SL_B0) + col] = s;
      }
    }
    __syncthreads();
    ES_STAMP(71 + 14 * (3 - layer));
    {
      // T1': through y1 = X_l + LN1(att): residual gradient -> gx, g_att -> ga; LayerNorm-parameter gradients
      const float4 g4 = *reinterpret_cast<const float4*>(t_b + row * ES_LDX + 4 * q);
      float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (live) a4 = *reinterpret_cast<const float4*>(svl + (size_t)row * ES_SV + 4 * q);
      float v[4] = {a4.x, a4.y, a4.z, a4.w}, gy[4] = {g4.x, g4.y, g4.z, g4.w}, xh[4], gg[4];
      float mu, rs;
      es_layernorm(v, eps1, mu, rs);
      float c1 = 0.f, c2 = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        xh[k] = (v[k] - mu) * rs;
        lnacc[8 + k] = gy[k] * xh[k];
        lnacc[12 + k] = gy[k];
        gg[k] = gy[k] * prm[q * 4 + k];
        c1 += gg[k]; c2 = fmaf(gg[k], xh[k], c2);
      }
      c1 = es_row8_sum(c1) * (1.f / 32.f); c2 = es_row8_sum(c2) * (1.f / 32.f);
      float4 gat = make_float4(rs * (gg[0] - c1 - xh[0] * c2), rs * (gg[1] - c1 - xh[1] * c2), rs * (gg[2] - c1 - xh[2] * c2), rs * (gg[3] - c1 - xh[3] * c2));
      if (!live) gat = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(ga + row * ES_LDX + 4 * q) = gat;
      *reinterpret_cast<float4*>(gx + row * ES_LDX + 4 * q) = live ? g4 : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const float s = eb_rows_sum(live ? lnacc[k] : 0.f);
        if (lane < 8) lnp[wave * 128 + (k >> 2) * 32 + 4 * q + (k & 3)] = s;
      }
    }
    __syncthreads();
    if (tid < 128) {
      const float s = ((lnp[tid] + lnp[128 + tid]) + lnp[256 + tid]) + lnp[384 + tid];
      const int quant = tid >> 5, col = tid & 31;
      sl_l[(quant == 0 ? ES_SL_LN2G : quant == 1 ? ES_SL_LN2B : quant == 2 ? ES_SL_LN1G : ES_SL_LN1B) + col] = s;
    }
    // q|k|v|skip of this layer (as the forward)
    {
      const int ntile_n = (n + 15) >> 4;
      for (int rt = 0; rt < ntile_n; ++rt) {
        float a[8];
        es_ld8(xs + (16 * rt + c) * ES_LDX + 8 * g, a);
        es_f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 8; ++t) { acc0 = es_mfma(a[t], wq[0][t], acc0); acc1 = es_mfma(a[t], wq[1][t], acc1); }
        float* o = qk + (16 * rt + 4 * g) * ES_LDQ + 32 * wave + c;
#pragma unroll
        for (int r = 0; r < 4; ++r) { o[r * ES_LDQ] = acc0[r] + bq[0]; o[r * ES_LDQ + 16] = acc1[r] + bq[1]; }
      }
    }
    float* ee = scr;                      // [ES_ECH][36]
    float* gsb = scr + ES_ECH * ES_LDX;   // [ES_ECH][8]: d loss / d score
    float* amb = gsb + ES_ECH * 8;        // [ES_ECH][8]: softmax weight x dropout factor
    float4 gk4 = make_float4(0.f, 0.f, 0.f, 0.f), gv4 = gk4;          // lane = (source atom, head) accumulators over the chunks
    es_f4 gWe = {0.f, 0.f, 0.f, 0.f};                                 // the wave's tile of gWedge (out tile trt, in tile tct)
    const int ai = tid >> 3, ah = tid & 7;
    float sm = 0.f, sinv = 0.f;
    if (ai < n) { sm = svl[(size_t)ai * ES_SV + 160 + ah]; sinv = svl[(size_t)ai * ES_SV + 168 + ah]; }
    __syncthreads();                      // qk ready; lnp consumed (ee aliases the tail's scratch)
    ES_STAMP(72 + 14 * (3 - layer));
    for (int t0 = 0; t0 < n;) {
      const int t1 = es_chunk_end(rp, t0, n);
      const int ce0 = rp[t0], cn = rp[t1] - ce0;
      {
        const int ntile_c = (cn + 15) >> 4;
        for (int rt = wave >> 1; rt < ntile_c; rt += 2) {
          const int el = 16 * rt + c;
          float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          if (el < cn) ea_row8(ce0 + el, 8 * g, a);
          es_f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int t = 0; t < 8; ++t) acc = es_mfma(a[t], we[t], acc);
#pragma unroll
          for (int r = 0; r < 4; ++r) ee[(16 * rt + 4 * g + r) * ES_LDX + tcol] = acc[r];
        }
      }
      __syncthreads();
      ES_STAMP(73 + 14 * (3 - layer));
      if (ai >= t0 && ai < t1) {
        // softmax backward of (target ai, head ah)
        const float4 q4 = *reinterpret_cast<const float4*>(qk + ai * ES_LDQ + ah * 4);
        const float4 go = *reinterpret_cast<const float4*>(ga + ai * ES_LDX + ah * 4);
        const int s0 = rp[ai], s1 = rp[ai + 1];
        const float* kb = qk + ES_D + ah * 4;
        const float* eb = ee + ah * 4 - ce0 * ES_LDX;
        float dsum = 0.f;
#pragma unroll 2
        for (int e = s0; e < s1; ++e) {
          const int j = sl[e];
          const float4 k4 = *reinterpret_cast<const float4*>(kb + j * ES_LDQ);
          const float4 v4 = *reinterpret_cast<const float4*>(kb + ES_D + j * ES_LDQ);
          const float4 e4 = *reinterpret_cast<const float4*>(eb + e * ES_LDX);
          const float a = es_exp(es_dot4(q4, k4, e4) * 0.5f - sm) * sinv;
          float ms = 1.f;
          if (p_att > 0.f) ms = (msde_uniform(seed_att, (unsigned long long)(e0 + e) * 8 + ah) >= p_att) ? keep_att : 0.f;
          const float gav = es_dot4(go, v4, e4);
          dsum = fmaf(a, gav * ms, dsum);
          gsb[(e - ce0) * 8 + ah] = gav * ms;          // (finished below)
          amb[(e - ce0) * 8 + ah] = a;
        }
        float4 gq = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 2
        for (int e = s0; e < s1; ++e) {
          const int j = sl[e];
          const float4 k4 = *reinterpret_cast<const float4*>(kb + j * ES_LDQ);
          const float4 e4 = *reinterpret_cast<const float4*>(eb + e * ES_LDX);
          const float a = amb[(e - ce0) * 8 + ah];
          const float gam = gsb[(e - ce0) * 8 + ah];
          const float gs = a * (gam - dsum) * 0.5f;
          float ms = 1.f;
          if (p_att > 0.f) ms = (msde_uniform(seed_att, (unsigned long long)(e0 + e) * 8 + ah) >= p_att) ? keep_att : 0.f;
          gsb[(e - ce0) * 8 + ah] = gs;
          amb[(e - ce0) * 8 + ah] = a * ms;
          gq.x = fmaf(gs, k4.x + e4.x, gq.x); gq.y = fmaf(gs, k4.y + e4.y, gq.y);
          gq.z = fmaf(gs, k4.z + e4.z, gq.z); gq.w = fmaf(gs, k4.w + e4.w, gq.w);
        }
        *reinterpret_cast<float4*>(gqk + ai * ES_LDQ + ah * 4) = gq;
        *reinterpret_cast<float4*>(gqk + ai * ES_LDQ + 3 * ES_D + ah * 4) = go;      // d out / d skip = 1
      }
      __syncthreads();
      ES_STAMP(74 + 14 * (3 - layer));
      {
        // gradient of the lin_edge rows of this chunk, rank one per head, written OVER the lin_edge rows (no longer needed):
        //   gee[e][col] = gs[e][h] q[dst][col] + am[e][h] g_att[dst][col]
        for (int t = tid; t < cn * 8; t += 256) {
          const int el = t >> 3, h = t & 7, i = dl[ce0 + el];
          const float gs = gsb[el * 8 + h], am = amb[el * 8 + h];
          const float4 qd = *reinterpret_cast<const float4*>(qk + i * ES_LDQ + 4 * h);
          const float4 gd = *reinterpret_cast<const float4*>(ga + i * ES_LDX + 4 * h);
          *reinterpret_cast<float4*>(ee + el * ES_LDX + 4 * h) =
              make_float4(gs * qd.x + am * gd.x, gs * qd.y + am * gd.y, gs * qd.z + am * gd.z, gs * qd.w + am * gd.w);
        }
        // key / value gradients: lane = (source atom ai, head ah) over its out-edges that lie in this chunk
        if (ai < n) {
          for (int s = rps[ai]; s < rps[ai + 1]; ++s) {
            const int e = es[s];
            if (e >= ce0 && e < ce0 + cn) {
              const int i = dl[e];
              const float gs = gsb[(e - ce0) * 8 + ah], am = amb[(e - ce0) * 8 + ah];
              const float4 qd = *reinterpret_cast<const float4*>(qk + i * ES_LDQ + ah * 4);
              const float4 gd = *reinterpret_cast<const float4*>(ga + i * ES_LDX + ah * 4);
              gk4.x = fmaf(gs, qd.x, gk4.x); gk4.y = fmaf(gs, qd.y, gk4.y); gk4.z = fmaf(gs, qd.z, gk4.z); gk4.w = fmaf(gs, qd.w, gk4.w);
              gv4.x = fmaf(am, gd.x, gv4.x); gv4.y = fmaf(am, gd.y, gv4.y); gv4.z = fmaf(am, gd.z, gv4.z); gv4.w = fmaf(am, gd.w, gv4.w);
            }
          }
        }
      }
      __syncthreads();
      {
        const int ntile_c = (cn + 15) >> 4;
        // gWedge[out][kin] += sum_e gee[e][out] edge_attr[e][kin]: the wave's tile (out tile trt, in tile tct), all tiles of the
        // chunk, two independent accumulators
        es_f4 gWe2 = {0.f, 0.f, 0.f, 0.f};
        for (int rt = 0; rt < ntile_c; rt += 2) {
          float av0[4], bv0[4], av1[4], bv1[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int el0 = 16 * rt + 4 * g + t, el1 = el0 + 16;
            const bool ok0 = el0 < cn, ok1 = el1 < cn;
            av0[t] = ok0 ? ee[el0 * ES_LDX + 16 * trt + c] : 0.f;
            bv0[t] = ok0 ? ea_at(ce0 + el0, tcol) : 0.f;
            av1[t] = ok1 ? ee[el1 * ES_LDX + 16 * trt + c] : 0.f;
            bv1[t] = ok1 ? ea_at(ce0 + el1, tcol) : 0.f;
          }
#pragma unroll
          for (int t = 0; t < 4; ++t) { gWe = es_mfma(av0[t], bv0[t], gWe); gWe2 = es_mfma(av1[t], bv1[t], gWe2); }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) gWe[r] += gWe2[r];
        // g_edge_attr[e][kin] += sum_out gee[e][out] Wedge[out][kin]: in tile tct, edge tiles trt, trt + 2, ...
        for (int rt = trt; rt < ntile_c; rt += 2) {
          const int el = 16 * rt + c;
          float av[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          if (el < cn) es_ld8(ee + el * ES_LDX + 8 * g, av);
          es_f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int t = 0; t < 8; ++t) acc = es_mfma(av[t], wet[t], acc);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int er = 16 * rt + 4 * g + r;
            if (er < cn) g_ea[((size_t)e0 + ce0 + er) * ld_gea + tcol] += acc[r];
          }
        }
      }
      __syncthreads();
      ES_STAMP(75 + 14 * (3 - layer));
      t0 = t1;
    }
    if (ai < n) {
      *reinterpret_cast<float4*>(gqk + ai * ES_LDQ + ES_D + ah * 4) = gk4;
      *reinterpret_cast<float4*>(gqk + ai * ES_LDQ + 2 * ES_D + ah * 4) = gv4;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) sl_l[ES_SL_WE + (16 * trt + 4 * g + r) * ES_D + tcol] = gWe[r];
    __syncthreads();
    {
      // through q|k|v|skip = X_l Wqkvs^T + b: gx += gqk Wqkvs (contraction over 128 columns), gWqkvs = gqk^T X_l, gb = column sums
      es_f4 acc = {0.f, 0.f, 0.f, 0.f};
      const float* Wq = W.Wq(layer, g);               // column 32 g + .. of q|k|v|skip = row .. of block g
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        float a[8], b[8];
        es_ld8(gqk + (16 * trt + c) * ES_LDQ + 32 * g + 8 * kk, a);
#pragma unroll
        for (int t = 0; t < 8; ++t) b[t] = Wq[(size_t)(8 * kk + t) * ES_D + tcol];
#pragma unroll
        for (int t = 0; t < 8; ++t) acc = es_mfma(a[t], b[t], acc);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) gx[(16 * trt + 4 * g + r) * ES_LDX + tcol] += acc[r];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int tile = 4 * wave + u, colt = tile >> 1, kt = tile & 1;
        const es_f4 d = es_xty(gqk, ES_LDQ, 16 * colt, xs, ES_LDX, 16 * kt, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) sl_l[(size_t)(16 * colt + 4 * g + r) * ES_D + 16 * kt + c] = d[r];
      }
      if (tid < 4 * ES_D) {
        float s = 0.f;
        for (int r = 0; r < ES_NMAX; ++r) s += gqk[r * ES_LDQ + tid];
        sl_l[ES_SL_BQ + tid] = s;
      }
    }
    __syncthreads();
    ES_STAMP(76 + 14 * (3 - layer));
  }
  if (live) *reinterpret_cast<float4*>(g_x0 + (size_t)(n0 + row) * ES_D + 4 * q) = *reinterpret_cast<const float4*>(gx + row * ES_LDX + 4 * q);
}

extern "C" long long msde_escore_mol_slab_floats(void) { return ES_SLAB; }

// Backward of msde_escore_mol_fwd (same arguments; `saved` written by it).  rowptr_s / perm_s: the by-source view of the edges
// (slot -> by-target edge id).  g_out [N,3].  Writes g_x0 [N,32], g_edge_attr [E, ld_gea] (every row: rows behind the last
// molecule's edges are zero-filled) and B slabs of msde_escore_mol_slab_floats() floats with the weight gradients of each
// molecule (layout: csrc/escore_mol.h), to be summed over the B workgroups.
extern "C" int msde_escore_mol_bwd(const void* const* params, const float* x0, const float* edge_attr, int ld_ea,
                                   const float* basis, const int* mol_ptr, int B, const int* rowptr, const int* src,
                                   const int* dst, const int* rowptr_s, const int* perm_s, int N, int E, int hidden, int heads,
                                   int hidden_coff, int n_max, float p_att, float p_ffn, unsigned long long seed0,
                                   const unsigned long long* seed_dev, float eps1, float eps2, const float* saved,
                                   const float* g_out, float* g_x0, float* g_edge_attr, int ld_gea, float* slabs, void* stream) {
  if (!params || !x0 || !edge_attr || !basis || !mol_ptr || !rowptr || !src || !dst || !rowptr_s || !perm_s || !saved || !g_out ||
      !g_x0 || !g_edge_attr || !slabs || N < 0 || B < 0 || E < 0)
    return MSDE_EINVAL;
  if (hidden != ES_D || heads != 8 || hidden_coff != ES_HC || n_max > ES_NMAX) return MSDE_EUNSUP;
  if (ld_ea < ES_D || ld_ea % 4 || ld_gea < ES_D || ld_gea % 4 || (reinterpret_cast<uintptr_t>(edge_attr) & 15) ||
      (reinterpret_cast<uintptr_t>(x0) & 15) || (reinterpret_cast<uintptr_t>(g_edge_attr) & 15) ||
      (reinterpret_cast<uintptr_t>(g_x0) & 15) || (reinterpret_cast<uintptr_t>(saved) & 15) || (reinterpret_cast<uintptr_t>(slabs) & 15))
    return MSDE_EINVAL;
  if (p_att < 0.f || p_att >= 1.f || p_ffn < 0.f || p_ffn >= 1.f) return MSDE_EINVAL;
  EsW W{reinterpret_cast<const float* const*>(params)};
  if (N == 0 || B == 0) return 0;
  MSDE_LAUNCH(escore_mol_bwd_kernel, dim3(B), dim3(256), 0, as_stream(stream), W, x0, edge_attr, ld_ea, basis, mol_ptr, B,
              rowptr, src, dst, rowptr_s, perm_s, N, E, p_att, p_ffn, seed0, seed_dev, eps1, eps2, saved, g_out, g_x0,
              g_edge_attr, ld_gea, slabs);
  MSDE_CHECK_LAUNCH();
  return 0;
}
