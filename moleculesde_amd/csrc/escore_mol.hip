// escore_mol.hip — the whole EquivariantScoreNetwork of the 2D->3D model, ONE WORKGROUP PER MOLECULE
// (equivariant_scorenetwork.py:13-40 GATLayer, :121-169 forward; SDE_model_2D_to_3D.py:386-391, :393-445 get_score).
//
// Every extended edge joins two atoms of one molecule (<= 32 atoms, <= 992 edges, hidden size 32), so a molecule's four
// GAT layers, two basis MLPs and the frame mix never need data of another workgroup.  The kernel is a LATENCY problem (four
// waves walk ~30 dependent phases), so every phase is laid out to need no cross-lane traffic (ds_bpermute shuffles were
// 60 % of the first version's 105 us for ten 14-atom molecules) and no global round trip (indices and edge features are
// staged in LDS once, the weights of layer l + 1 are requested while layer l computes):
//     per layer   qkvs = x Wqkvs^T + b                [n, 128]   fp32 MFMA 16x16x4, A from LDS, B (weights) in registers
//                 ee   = edge_attr Wedge^T            [E_m, 32]  fp32 MFMA, result in LDS
//                 attention: one LANE per (target, head) walking the target's in-edges -- scores, softmax statistics and
//                            the weighted sum stay in the lane's registers; dropout from the counter mask of the operator path
//                 tail: y1 = x + LN1(att) on 8 lanes per atom row (row sums by DPP), the two 32 x 32 feed-forward products
//                       on MFMA (one 16 x 16 tile per wave), out = y1 + LN2(.) [+ SiLU]
//     per block   Z^T = W1 [h_src + h_dst | edge_attr]^T + b1  [128, E_m]  fp32 MFMA, TRANSPOSED so that the 3-wide head
//                 coff^T = W2 SiLU(Z^T) takes the accumulator tile as its B operand (no lane reduction); a wave takes every
//                 fourth 16-edge tile with all 128 hidden rows (W1 resident in registers); frame mix per edge, mean over the
//                 in-edges of each atom in edge order (fixed order: bit-reproducible)
// The operator path (msde_edge_attention_*, msde_gat_tail_*, msde_mlp_head_mix_*, ~18 launches forward) stays as the
// cross-check and for shapes this kernel does not take (hidden != 32, heads != 8, basis-MLP width != 128, > 32 atoms).
// Dropout masks are functions of (seed, GLOBAL edge / element index) exactly as in those kernels, so both paths draw the
// same masks from the same seed.
//
// Training: the forward stores, per layer and atom, the rows the backward kernel does not rebuild (attention output, y1, h0,
// x2, the layer output, the softmax statistics max / 1/sum per head: ES_SV floats); everything per-edge is recomputed there.
#include "escore_mol.h"

// Register-resident weights of one GAT layer (prefetched one layer ahead: the loads of layer l + 1 are in flight while
// layer l computes -- with one wave per SIMD nothing else hides a global round trip).
struct EsLayerRegs {
  float wq[2][8], bq[2];        // B fragments of the wave's two column tiles of Wqkvs, their bias entries
  float we[8];                  // B fragment of the wave's column tile of Wedge
  float w0[8], w3[8];           // B fragments of the wave's column tile of the feed-forward weights
  float prm;                    // this thread's entry of [ln1_g | ln1_b | b0 | b3 | ln2_g | ln2_b] (threads < 192)
};

__device__ __forceinline__ void es_load_layer(const EsW& W, int layer, int tid, EsLayerRegs& R) {
  const int wave = tid >> 6, lane = tid & 63, c = lane & 15, g = lane >> 4;
#pragma unroll
  for (int cti = 0; cti < 2; ++cti) {
    const int colb = 16 * cti + c;               // the wave's two column tiles are exactly block `wave` of q|k|v|skip
    es_ld8(W.Wq(layer, wave) + (size_t)colb * ES_D + 8 * g, R.wq[cti]);
    R.bq[cti] = W.bq(layer, wave)[colb];
  }
  const size_t frag = (size_t)(16 * (wave & 1) + c) * ES_D + 8 * g;
  es_ld8(W.Wedge(layer) + frag, R.we);
  es_ld8(W.W0(layer) + frag, R.w0);
  es_ld8(W.W3(layer) + frag, R.w3);
  R.prm = 0.f;
  if (tid < 6 * ES_D) {
    const int f = tid >> 5, k = tid & 31;
    const float* src = f == 0 ? W.ln1g(layer) : f == 1 ? W.ln1b(layer) : f == 2 ? W.b0(layer) : f == 3 ? W.b3(layer)
                       : f == 4 ? W.ln2g(layer) : W.ln2b(layer);
    R.prm = src[k];
  }
}


#ifdef ES_TIMING
extern "C" int msde_escore_debug_stamps(long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(es_stamps), sizeof(long long) * 128);
}
#endif

// Inputs of the edge features of get_score: per-edge SE(3) frame, distance and frame coordinates, their Gaussian-Fourier
// features, input_mlp / coff_mlp / project and edge_attr = input_mlp(.) * edge_2D + project(.)
// (SDE_model_2D_to_3D.py:342-369,423-432; the ~7 launches of hip.edge_geometry_stacked / frame_mlp / mul_add).
struct EsGeo {
  const float* pos;             // [N, 3] perturbed coordinates
  const float* e2d;             // [E, ld] edge_2D_emb of the node pairs (coordinate independent, computed once per representation)
  int ld_e2d, has_dist;         // has_dist = 0: SDEModel2Dto3D_01 (no distance branch: edge_attr = edge_2D + frame)
};


// ---- get_score in two launches (inference) ---------------------------------------------------------------------------------
// Everything of the score network that depends on an EDGE alone is hoisted out of the per-molecule latency chain into a wide
// launch (one wave per 16 edges, any number of workgroups -- the sampler's ten molecules keep ten CUs busy, its 1 820 edges
// fill 114 waves): the edge features (frame, Fourier features, input_mlp / coff_mlp / project), everything TRANSPOSED (lane =
// edge: the features of a lane's own edge are the B operand, each product's accumulator tile is the next one's), then lin_edge of all four GAT layers and the edge half of both basis MLPs' first Linear.  Row e of `pre` (ES_PRE_LD floats):
//   [0, 128)    lin_edge_l(edge_attr), l = 0..3           (what the attention of layer l adds to k_j and v_j)
//   [128, 384)  W1_m[:, 32:] edge_attr, m = 0, 1          (the edge part of the basis MLP's hidden layer, no bias)
//   [384, 393)  the three frame vectors
// The per-molecule kernel (PRE) then only copies / adds these rows.
#define ES_PRE_LD 400
#define ES_PRE_WLD 36          // row stride of the LDS copies of Wedge (4 x 32 rows) and W1[:, 32:] (2 x 128 rows)

__global__ void __launch_bounds__(256)
escore_edge_pre_kernel(EsW W, EsGeo geo, const int* __restrict__ src, const int* __restrict__ dst, int E,
                       float* __restrict__ pre) {
  // every weight matrix goes through LDS once per workgroup (coalesced 16-byte / 8-byte loads; a lane's MFMA fragments are
  // rows c of 16-row blocks: read straight from global they are ~200 strided loads per lane, 7 us of the first version's 20)
  constexpr int S_IN = 68, S_CM = 132, S_P0 = 68, S_P1 = 36;
  constexpr int O_IN = (4 * ES_D + 2 * ES_HC) * ES_PRE_WLD, O_CM = O_IN + ES_D * S_IN, O_P0 = O_CM + ES_D * S_CM,
                O_P1 = O_P0 + ES_D * S_P0, O_END = O_P1 + ES_D * S_P1;
  __shared__ __attribute__((aligned(16))) float wl[O_END];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 15, g = lane >> 4;
  ES_STAMP(50);
  const float* Wd = W.p[76]; const float* Wc = W.p[77];
  const float* Win = W.p[78]; const float* bin = W.p[79]; const float* Wcm = W.p[80]; const float* bcm = W.p[81];
  const float* Wp0 = W.p[82]; const float* bp0 = W.p[83]; const float* Wp1 = W.p[84]; const float* bp1 = W.p[85];
  {
    // Wedge_l rows 32 l + o, then W1_m[:, 32:] rows 128 + 128 m + o: 384 rows of 32 floats
    // (one loop per matrix: its address is uniform -- a per-row choice of the matrix would load the pointer table per lane,
    // a dependent round trip in front of every row)
    const int q = tid & 7, r8 = tid >> 3;
    float4 stg[4 + 2 * 4];
#pragma unroll
    for (int l = 0; l < 4; ++l) stg[l] = *reinterpret_cast<const float4*>(W.Wedge(l) + (size_t)r8 * ES_D + 4 * q);
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int u = 0; u < 4; ++u)
        stg[4 + 4 * m + u] = *reinterpret_cast<const float4*>(W.bW1(m) + (size_t)(r8 + 32 * u) * (2 * ES_D) + ES_D + 4 * q);
#pragma unroll
    for (int l = 0; l < 4; ++l) *reinterpret_cast<float4*>(wl + (32 * l + r8) * ES_PRE_WLD + 4 * q) = stg[l];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int u = 0; u < 4; ++u)
        *reinterpret_cast<float4*>(wl + (4 * ES_D + ES_HC * m + r8 + 32 * u) * ES_PRE_WLD + 4 * q) = stg[4 + 4 * m + u];
    for (int t = tid; t < ES_D * 16; t += 256)       // input_mlp weight [32][64]
      *reinterpret_cast<float4*>(wl + O_IN + (t >> 4) * S_IN + 4 * (t & 15)) = *reinterpret_cast<const float4*>(Win + (size_t)(t >> 4) * 64 + 4 * (t & 15));
    for (int t = tid; t < ES_D * 32; t += 256)       // coff_mlp weight [32][128]
      *reinterpret_cast<float4*>(wl + O_CM + (t >> 5) * S_CM + 4 * (t & 31)) = *reinterpret_cast<const float4*>(Wcm + (size_t)(t >> 5) * 128 + 4 * (t & 31));
    for (int t = tid; t < ES_D * 32; t += 256)       // project[0] weight [32][66]: columns 2 .. 65 (rows are 8-byte aligned there)
      *reinterpret_cast<float2*>(wl + O_P0 + (t >> 5) * S_P0 + 2 * (t & 31)) = *reinterpret_cast<const float2*>(Wp0 + (size_t)(t >> 5) * 66 + 2 + 2 * (t & 31));
    for (int t = tid; t < ES_D * 8; t += 256)        // project[1] weight [32][32]
      *reinterpret_cast<float4*>(wl + O_P1 + (t >> 3) * S_P1 + 4 * (t & 7)) = *reinterpret_cast<const float4*>(Wp1 + (size_t)(t >> 3) * 32 + 4 * (t & 7));
  }
  float wdv[16], wcv0[16], wcv1[16];
  es_ld16(Wd + 16 * (g & 1), wdv);
  es_ld16(Wc, wcv0);
  es_ld16(Wc + 16, wcv1);
  const int ntile = (E + 15) >> 4;
  const int rt = (int)blockIdx.x * 4 + wave;
  const int e = 16 * rt + c;
  const bool on = rt < ntile && e < E;
  const int r_ = on ? max(src[e], 0) : 0, q_ = on ? max(dst[e], 0) : 0;
  const float prx = geo.pos[3 * r_], pry = geo.pos[3 * r_ + 1], prz = geo.pos[3 * r_ + 2];
  const float pcx = geo.pos[3 * q_], pcy = geo.pos[3 * q_ + 1], pcz = geo.pos[3 * q_ + 2];
  float4 e2v[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
    e2v[nt] = on ? *reinterpret_cast<const float4*>(geo.e2d + (size_t)e * geo.ld_e2d + 16 * nt + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
  ES_STAMP(51);
  __syncthreads();                                     // wl staged (every wave arrives, also those without a tile)
  ES_STAMP(52);
  if (rt >= ntile) return;
  // coord2basis and the frame coordinates of both endpoints: as edge_geometry_fwd_kernel (csrc/sde2d3d.hip)
  float dx = prx - pcx, dy = pry - pcy, dz = prz - pcz;
  float cx = pry * pcz - prz * pcy, cy = prz * pcx - prx * pcz, cz = prx * pcy - pry * pcx;
  const float dist = sqrtf(dx * dx + dy * dy + dz * dz);
  const float nrm = dist + 1e-6f;
  dx /= nrm; dy /= nrm; dz /= nrm;
  const float cn = sqrtf(cx * cx + cy * cy + cz * cz) + 1e-6f;
  cx /= cn; cy /= cn; cz /= cn;
  const float vx = dy * cz - dz * cy, vy = dz * cx - dx * cz, vz = dx * cy - dy * cx;
  const float ci0 = dx * prx + dy * pry + dz * prz, ci1 = fabsf(cx * prx + cy * pry + cz * prz), ci2 = vx * prx + vy * pry + vz * prz;
  const float cj0 = dx * pcx + dy * pcy + dz * pcz, cj1 = fabsf(cx * pcx + cy * pcy + cz * pcz), cj2 = vx * pcx + vy * pcy + vz * pcz;
  const float ni = sqrtf(ci0 * ci0 + ci1 * ci1 + ci2 * ci2), nj = sqrtf(cj0 * cj0 + cj1 * cj1 + cj2 * cj2);
  const float pcos = (ci0 * cj0 + ci1 * cj1 + ci2 * cj2) / (ni + 1e-6f) / (nj + 1e-6f);
  const float psin = sqrtf(1.f - pcos * pcos);
  float* prow = pre + (size_t)(on ? e : 0) * ES_PRE_LD;
  if (g == 0 && on) {
    float* b = prow + 384;
    b[0] = dx; b[1] = dy; b[2] = dz; b[3] = cx; b[4] = cy; b[5] = cz; b[6] = vx; b[7] = vy; b[8] = vz;
  }
  // Gaussian-Fourier features as B operands on the transcendental unit (argument in revolutions; cos x = sin(x + 1/4))
  const float qshift = (g & 1) ? 0.25f : 0.f;
  auto four = [&](float x, float w) -> float { return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(fmaf(x, w, qshift))); };
  const int nrow = 4 * g;                              // accumulator rows of this lane: outputs 16 nt + 4 g + r
  es_f4 accI[2], accEi[2], accEj[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const float4 b0 = *reinterpret_cast<const float4*>(bin + 16 * nt + nrow), b1 = *reinterpret_cast<const float4*>(bcm + 16 * nt + nrow);
    accI[nt] = es_f4{b0.x, b0.y, b0.z, b0.w};
    accEi[nt] = es_f4{b1.x, b1.y, b1.z, b1.w};
    accEj[nt] = accEi[nt];
  }
  if (geo.has_dist) {
    // feat_d[k], k = 16 g + t: sin(dist Wd[k]) for k < 32, cos(dist Wd[k - 32]) above
    const float dshift = g >= 2 ? 0.25f : 0.f;
    float wi[2][16];
    es_ld16(wl + O_IN + c * S_IN + 16 * g, wi[0]);
    es_ld16(wl + O_IN + (16 + c) * S_IN + 16 * g, wi[1]);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const float f = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(fmaf(dist, wdv[t], dshift)));
      accI[0] = es_mfma(wi[0][t], f, accI[0]);
      accI[1] = es_mfma(wi[1][t], f, accI[1]);
    }
  }
  {
    // feat_i[k], k = 32 g + t: [sin(ci0 Wc) | cos(ci0 Wc) | sin(ci2 Wc) | cos(ci2 Wc)]; feat_j likewise
    const float xi = g < 2 ? ci0 : ci2, xj = g < 2 ? cj0 : cj2;
    float wcmA[2][16], wcmB[2][16];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      es_ld16(wl + O_CM + (16 * nt + c) * S_CM + 32 * g, wcmA[nt]);
      es_ld16(wl + O_CM + (16 * nt + c) * S_CM + 32 * g + 16, wcmB[nt]);
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const float fi = four(xi, wcv0[t]), fj = four(xj, wcv0[t]);
      accEi[0] = es_mfma(wcmA[0][t], fi, accEi[0]); accEi[1] = es_mfma(wcmA[1][t], fi, accEi[1]);
      accEj[0] = es_mfma(wcmA[0][t], fj, accEj[0]); accEj[1] = es_mfma(wcmA[1][t], fj, accEj[1]);
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const float fi = four(xi, wcv1[t]), fj = four(xj, wcv1[t]);
      accEi[0] = es_mfma(wcmB[0][t], fi, accEi[0]); accEi[1] = es_mfma(wcmB[1][t], fi, accEi[1]);
      accEj[0] = es_mfma(wcmB[0][t], fj, accEj[0]); accEj[1] = es_mfma(wcmB[1][t], fj, accEj[1]);
    }
  }
  es_f4 accH[2], accF[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const float4 b0 = *reinterpret_cast<const float4*>(bp0 + 16 * nt + nrow);
    float hb[4] = {b0.x, b0.y, b0.z, b0.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float* wr = Wp0 + (size_t)(16 * nt + nrow + r) * 66;
      hb[r] += wr[0] * psin + wr[1] * pcos;
    }
    accH[nt] = es_f4{hb[0], hb[1], hb[2], hb[3]};
  }
  float wp0[2][16], wp1[2][8];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {                   // q4 = 2 e_ + ot: column block 32 e_ + 16 ot + 4 g of project[0]
      const float4 v = *reinterpret_cast<const float4*>(wl + O_P0 + (16 * nt + c) * S_P0 + 16 * q4 + 4 * g);
      wp0[nt][4 * q4] = v.x; wp0[nt][4 * q4 + 1] = v.y; wp0[nt][4 * q4 + 2] = v.z; wp0[nt][4 * q4 + 3] = v.w;
    }
#pragma unroll
    for (int n2 = 0; n2 < 2; ++n2) {
      const float4 v = *reinterpret_cast<const float4*>(wl + O_P1 + (16 * nt + c) * S_P1 + 16 * n2 + 4 * g);
      wp1[nt][4 * n2] = v.x; wp1[nt][4 * n2 + 1] = v.y; wp1[nt][4 * n2 + 2] = v.z; wp1[nt][4 * n2 + 3] = v.w;
    }
  }
#pragma unroll
  for (int ot = 0; ot < 2; ++ot)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        accH[nt] = es_mfma(wp0[nt][ot * 4 + r], accEi[ot][r], accH[nt]);
        accH[nt] = es_mfma(wp0[nt][(2 + ot) * 4 + r], accEj[ot][r], accH[nt]);
      }
    }
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const float4 b0 = *reinterpret_cast<const float4*>(bp1 + 16 * nt + nrow);
    accF[nt] = es_f4{b0.x, b0.y, b0.z, b0.w};
  }
#pragma unroll
  for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float hz = accH[n2][r];
      const float hv = hz * es_sigmoid(hz);
      accF[0] = es_mfma(wp1[0][n2 * 4 + r], hv, accF[0]);
      accF[1] = es_mfma(wp1[1][n2 * 4 + r], hv, accF[1]);
    }
  ES_STAMP(53);
  // edge_attr^T tile: feature 16 nt + 4 g + r of edge c -- the B operand of everything below (k <-> (nt, r), group g)
  es_f4 eaT[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const float e2[4] = {e2v[nt].x, e2v[nt].y, e2v[nt].z, e2v[nt].w};
#pragma unroll
    for (int r = 0; r < 4; ++r) eaT[nt][r] = geo.has_dist ? fmaf(accI[nt][r], e2[r], accF[nt][r]) : e2[r] + accF[nt][r];
  }
  // 24 output tiles of 16 features: 8 of lin_edge (4 layers x 2), 16 of the basis MLPs' edge halves; weights from LDS
#pragma unroll 2
  for (int ot = 0; ot < 24; ++ot) {
    const float* wr = wl + (16 * ot + c) * ES_PRE_WLD + 4 * g;
    const float4 w0 = *reinterpret_cast<const float4*>(wr), w1 = *reinterpret_cast<const float4*>(wr + 16);
    es_f4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = es_mfma(w0.x, eaT[0][0], acc); acc = es_mfma(w0.y, eaT[0][1], acc);
    acc = es_mfma(w0.z, eaT[0][2], acc); acc = es_mfma(w0.w, eaT[0][3], acc);
    acc = es_mfma(w1.x, eaT[1][0], acc); acc = es_mfma(w1.y, eaT[1][1], acc);
    acc = es_mfma(w1.z, eaT[1][2], acc); acc = es_mfma(w1.w, eaT[1][3], acc);
    if (on) *reinterpret_cast<float4*>(prow + 16 * ot + nrow) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  }
  ES_STAMP(54);
}

template <bool TRAIN, bool PRE = false>
__global__ void __launch_bounds__(256)
escore_mol_fwd_kernel(EsW W, const float* __restrict__ x0, const float* __restrict__ ea, int ld_ea,
                      const float* __restrict__ basis, const int* __restrict__ mol_ptr, int B, const int* __restrict__ rowptr,
                      const int* __restrict__ src, const int* __restrict__ dst, int N, float p_att, float p_ffn,
                      unsigned long long seed0, const unsigned long long* __restrict__ seed_dev, float eps1, float eps2,
                      float* __restrict__ out, float* __restrict__ sv) {
  __shared__ __attribute__((aligned(16))) float xs[ES_NMAX * ES_LDX];       // layer input; inside the tail: SiLU(h0) rows
  __shared__ __attribute__((aligned(16))) float att[ES_NMAX * ES_LDX];      // attention output; inside the tail: y1 rows
  __shared__ __attribute__((aligned(16))) float qk[ES_NMAX * ES_LDQ];       // q|k|v|skip; inside the tail: x2 rows
  __shared__ __attribute__((aligned(16))) float eal[ES_EAL * ES_LDX];       // edge features of the molecule (Em <= ES_EAL)
  __shared__ __attribute__((aligned(16))) float ee[ES_ECH * ES_LDX];        // basis phase: per-edge mixed vectors [Em][3]
  __shared__ __attribute__((aligned(16))) float hw[ES_HC + 3 * ES_HC];      // basis phase: b1 | W2
  __shared__ float prm[6 * ES_D];
  __shared__ float gacc[ES_NMAX * 3];
  __shared__ int rp[ES_NMAX + 1];
  __shared__ unsigned char sl[ES_EMAX + 16], dl[ES_EMAX + 16];              // molecule-local source / target of every edge
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  ES_STAMP(0);
  {
    // rows behind the last molecule (capacity padding): finite outputs, one slice per workgroup
    const int nt = mol_ptr[B];
    for (int t = nt * 3 + (int)blockIdx.x * 256 + tid; t < N * 3; t += B * 256) out[t] = 0.f;
  }
  const int n0 = mol_ptr[blockIdx.x], n = min(mol_ptr[blockIdx.x + 1] - n0, ES_NMAX);
  if (n <= 0) return;
  EsLayerRegs R;
  es_load_layer(W, 0, tid, R);
  const int e0 = rowptr[n0], Em = min(rowptr[n0 + n] - e0, ES_EMAX);
  const bool ea_lds = Em <= ES_EAL;
  const unsigned long long sdev = seed_dev ? seed_dev[0] * 0x100000001B3ull : 0ull;
  for (int t = tid; t <= n; t += 256) rp[t] = rowptr[n0 + t] - e0;
  for (int t = tid; t < Em; t += 256) {
    sl[t] = (unsigned char)(src[e0 + t] - n0);
    dl[t] = (unsigned char)(dst[e0 + t] - n0);
  }
  {
    const int q = tid & 7;
    for (int row = tid >> 3; row < ES_NMAX; row += 32) {   // xs <- node_attr rows (zero beyond n)
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < n) v = *reinterpret_cast<const float4*>(x0 + (size_t)(n0 + row) * ES_D + 4 * q);
      *reinterpret_cast<float4*>(xs + row * ES_LDX + 4 * q) = v;
      *reinterpret_cast<float4*>(att + row * ES_LDX + 4 * q) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (ea_lds && !PRE)
      for (int row = tid >> 3; row < Em; row += 32)
        *reinterpret_cast<float4*>(eal + row * ES_LDX + 4 * q) =
            *reinterpret_cast<const float4*>(ea + ((size_t)e0 + row) * ld_ea + 4 * q);
  }
  if (tid < ES_NMAX * 3) gacc[tid] = 0.f;
  const int c = lane & 15, g = lane >> 4;
  int rp_first_chunk = 0;                              // PRE: edges of the first attention chunk (targets 0 .. es_chunk_end)
#pragma unroll 1
  for (int layer = 0; layer < ES_LAYERS; ++layer) {
    const int mi = layer >> 1, ci = layer & 1;
    if (tid < 6 * ES_D) prm[tid] = R.prm;              // (the previous layer's tail finished behind a barrier)
    float wq[2][8], bq[2], we[8], w0[8], w3[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) { wq[0][t] = R.wq[0][t]; wq[1][t] = R.wq[1][t]; we[t] = R.we[t]; w0[t] = R.w0[t]; w3[t] = R.w3[t]; }
    bq[0] = R.bq[0]; bq[1] = R.bq[1];
    // PRE: lin_edge(edge_attr) of this layer was computed by escore_edge_pre_kernel -- the rows of the first chunk are requested
    // here and land in LDS behind the q|k|v|skip product
    __syncthreads();                                   // xs (first layer: + the staged inputs) visible
    float4 pf[PRE ? ES_ECH / 32 : 1];
    if (PRE) {
      if (layer == 0) rp_first_chunk = rp[es_chunk_end(rp, 0, n)];
#pragma unroll
      for (int u = 0; u < ES_ECH / 32; ++u) {
        const int el = (tid >> 3) + 32 * u;
        pf[u] = el < rp_first_chunk ? *reinterpret_cast<const float4*>(ea + ((size_t)e0 + el) * ld_ea + ES_D * layer + 4 * (tid & 7))
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    ES_STAMP(1 + 8 * layer);
    // qkvs = xs Wqkvs^T + b (rows >= n hold the bias: finite, never read); the wave's two column tiles are independent chains
    {
      const int ntile = (n + 15) >> 4;
      for (int rt = 0; rt < ntile; ++rt) {
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
    if (layer + 1 < ES_LAYERS) es_load_layer(W, layer + 1, tid, R);      // in flight during the rest of this layer
    ES_STAMP(2 + 8 * layer);
    // basis-MLP weights of this block: requested at the top of the block's second layer so that they arrive under it.  The first Linear is split,
    //   [h_src + h_dst | edge_attr] W1^T = P[src] + P[dst] + edge_attr W1[:, 32:]^T,   P = h W1[:, :32]^T  (per ATOM),
    // which halves the per-edge product (K = 32) -- b1w: the edge half as A fragments, wp: the atom half as B fragments
    float b1w[8][8], wp[2][8], pb1 = 0.f, pw2a = 0.f, pw2b = 0.f;
    if (ci == 1) {
      if (!PRE) {
#pragma unroll
        for (int ht = 0; ht < 8; ++ht) es_ld8(W.bW1(mi) + (size_t)(16 * ht + c) * (2 * ES_D) + ES_D + 8 * g, b1w[ht]);
      }
#pragma unroll
      for (int cti = 0; cti < 2; ++cti) es_ld8(W.bW1(mi) + (size_t)(32 * wave + 16 * cti + c) * (2 * ES_D) + 8 * g, wp[cti]);
      if (tid < ES_HC) pb1 = W.bb1(mi)[tid];
      pw2a = W.bW2(mi)[tid];
      if (tid < ES_HC) pw2b = W.bW2(mi)[256 + tid];
    }
    const unsigned long long seed_l = seed0 + (unsigned long long)(mi * 4 + ci);
    const unsigned long long seed_att = seed_l + sdev, seed_ffn = (seed_l ^ 0x46464Eull) + sdev;
    const float keep_att = p_att > 0.f ? 1.f / (1.f - p_att) : 1.f;
    for (int t0 = 0; t0 < n;) {
      const int t1 = es_chunk_end(rp, t0, n);
      const int ce0 = rp[t0], cn = rp[t1] - ce0;
      // ee[0 .. cn) = edge_attr[ce0 .. ce0 + cn) Wedge^T: wave -> column tile wave & 1, row tiles wave >> 1, + 2, ...
      // (two tiles per trip: independent MFMA chains)
      if (PRE) {
        if (t0 == 0) {
#pragma unroll
          for (int u = 0; u < ES_ECH / 32; ++u) {
            const int el = (tid >> 3) + 32 * u;
            if (el < cn) *reinterpret_cast<float4*>(ee + el * ES_LDX + 4 * (tid & 7)) = pf[u];
          }
        } else {
          for (int el = tid >> 3; el < cn; el += 32)
            *reinterpret_cast<float4*>(ee + el * ES_LDX + 4 * (tid & 7)) =
                *reinterpret_cast<const float4*>(ea + ((size_t)e0 + ce0 + el) * ld_ea + ES_D * layer + 4 * (tid & 7));
        }
      } else {
        const int ct = wave & 1, ntile = (cn + 15) >> 4;
        for (int rt = wave >> 1; rt < ntile; rt += 4) {
          const int el0 = 16 * rt + c, el1 = el0 + 32;
          float a0[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, a1[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          if (ea_lds) {
            if (el0 < cn) es_ld8(eal + (ce0 + el0) * ES_LDX + 8 * g, a0);
            if (el1 < cn) es_ld8(eal + (ce0 + el1) * ES_LDX + 8 * g, a1);
          } else {
            if (el0 < cn) es_ld8(ea + ((size_t)e0 + ce0 + el0) * ld_ea + 8 * g, a0);
            if (el1 < cn) es_ld8(ea + ((size_t)e0 + ce0 + el1) * ld_ea + 8 * g, a1);
          }
          es_f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int t = 0; t < 8; ++t) { acc0 = es_mfma(a0[t], we[t], acc0); acc1 = es_mfma(a1[t], we[t], acc1); }
          float* o = ee + (16 * rt + 4 * g) * ES_LDX + 16 * ct + c;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r * ES_LDX] = acc0[r];
          if (rt + 2 < ntile) {
#pragma unroll
            for (int r = 0; r < 4; ++r) o[(32 + r) * ES_LDX] = acc1[r];
          }
        }
      }
      __syncthreads();
      ES_STAMP(3 + 8 * layer);
      // attention: lane = (target, head); two walks over the target's in-edges (maximum; exponentials, their sum and the
      // weighted sum) -- scores are recomputed, not stored, and nothing crosses lanes
      if (!TRAIN && n <= 16 && t0 == 0 && t1 == n) {
        // inference, molecules of <= 16 atoms whose edges fit ONE chunk: TWO lanes per (target, head) -- lane half z takes the in-edges of
        // parity z, the halves meet by DPP (row_ror:8 inside the 16-lane row of a target) -- all 256 lanes busy instead of 8 n,
        // half the serial walk.  (The sum over the edges is formed as (even edges) + (odd edges).)
        const int i = tid >> 4, z = (tid >> 3) & 1, h = tid & 7;
        const bool live_i = i < n;
        const int ii = live_i ? i : 0;
        const float4 q4 = *reinterpret_cast<const float4*>(qk + ii * ES_LDQ + h * 4);
        const int s0 = rp[ii], s1 = live_i ? rp[ii + 1] : rp[ii];
        const float* kb = qk + ES_D + h * 4;
        const float* eb = ee + h * 4 - ce0 * ES_LDX;
        float m = -INFINITY;
#pragma unroll 4
        for (int e = s0 + z; e < s1; e += 2) {
          const float4 k4 = *reinterpret_cast<const float4*>(kb + sl[e] * ES_LDQ);
          const float4 e4 = *reinterpret_cast<const float4*>(eb + e * ES_LDX);
          m = fmaxf(m, es_dot4(q4, k4, e4) * 0.5f);
        }
        m = fmaxf(m, es_dpp<0x128>(m));                  // row_ror:8: the other half's maximum
        float sum = 0.f;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
        for (int e = s0 + z; e < s1; e += 2) {
          const int j = sl[e];
          const float4 k4 = *reinterpret_cast<const float4*>(kb + j * ES_LDQ);
          const float4 v4 = *reinterpret_cast<const float4*>(kb + ES_D + j * ES_LDQ);
          const float4 e4 = *reinterpret_cast<const float4*>(eb + e * ES_LDX);
          float p = es_exp(es_dot4(q4, k4, e4) * 0.5f - m);
          sum += p;
          if (p_att > 0.f) p = (msde_uniform(seed_att, (unsigned long long)(e0 + e) * 8 + h) >= p_att) ? p * keep_att : 0.f;
          acc.x = fmaf(p, v4.x + e4.x, acc.x); acc.y = fmaf(p, v4.y + e4.y, acc.y);
          acc.z = fmaf(p, v4.z + e4.z, acc.z); acc.w = fmaf(p, v4.w + e4.w, acc.w);
        }
        // even half + odd half, in that order on both lanes
        const float so = es_dpp<0x128>(sum), ax = es_dpp<0x128>(acc.x), ay = es_dpp<0x128>(acc.y), az = es_dpp<0x128>(acc.z),
                    aw = es_dpp<0x128>(acc.w);
        if (z == 0 && live_i) {
          const float inv = 1.f / ((sum + so) + 1e-16f);
          const float4 s4 = *reinterpret_cast<const float4*>(qk + i * ES_LDQ + 3 * ES_D + h * 4);   // + lin_skip(x_i)
          *reinterpret_cast<float4*>(att + i * ES_LDX + h * 4) =
              make_float4(fmaf(acc.x + ax, inv, s4.x), fmaf(acc.y + ay, inv, s4.y), fmaf(acc.z + az, inv, s4.z), fmaf(acc.w + aw, inv, s4.w));
        }
      } else {
        const int i = tid >> 3, h = tid & 7;
        if (i >= t0 && i < t1) {
          const float4 q4 = *reinterpret_cast<const float4*>(qk + i * ES_LDQ + h * 4);
          const int s0 = rp[i], s1 = rp[i + 1];
          const float* kb = qk + ES_D + h * 4;
          const float* eb = ee + h * 4 - ce0 * ES_LDX;
          float m = -INFINITY;
#pragma unroll 4
          for (int e = s0; e < s1; ++e) {
            const float4 k4 = *reinterpret_cast<const float4*>(kb + sl[e] * ES_LDQ);
            const float4 e4 = *reinterpret_cast<const float4*>(eb + e * ES_LDX);
            m = fmaxf(m, es_dot4(q4, k4, e4) * 0.5f);
          }
          float sum = 0.f;
          float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
          for (int e = s0; e < s1; ++e) {
            const int j = sl[e];
            const float4 k4 = *reinterpret_cast<const float4*>(kb + j * ES_LDQ);
            const float4 v4 = *reinterpret_cast<const float4*>(kb + ES_D + j * ES_LDQ);
            const float4 e4 = *reinterpret_cast<const float4*>(eb + e * ES_LDX);
            float p = es_exp(es_dot4(q4, k4, e4) * 0.5f - m);
            sum += p;
            if (p_att > 0.f) p = (msde_uniform(seed_att, (unsigned long long)(e0 + e) * 8 + h) >= p_att) ? p * keep_att : 0.f;
            acc.x = fmaf(p, v4.x + e4.x, acc.x); acc.y = fmaf(p, v4.y + e4.y, acc.y);
            acc.z = fmaf(p, v4.z + e4.z, acc.z); acc.w = fmaf(p, v4.w + e4.w, acc.w);
          }
          const float inv = 1.f / (sum + 1e-16f);
          const float4 s4 = *reinterpret_cast<const float4*>(qk + i * ES_LDQ + 3 * ES_D + h * 4);   // + lin_skip(x_i)
          *reinterpret_cast<float4*>(att + i * ES_LDX + h * 4) =
              make_float4(fmaf(acc.x, inv, s4.x), fmaf(acc.y, inv, s4.y), fmaf(acc.z, inv, s4.z), fmaf(acc.w, inv, s4.w));
          if (TRAIN) {
            float* st = sv + ((size_t)layer * N + n0 + i) * ES_SV + 160;
            st[h] = m; st[8 + h] = inv;
          }
        }
      }
      __syncthreads();
      ES_STAMP(4 + 8 * layer);
      t0 = t1;
    }
    // ---- tail -------------------------------------------------------------------------------------------------------------
    const int row = tid >> 3, q = tid & 7;
    const bool live = row < n;
    float* svr = TRAIN ? sv + ((size_t)layer * N + n0 + (live ? row : 0)) * ES_SV : nullptr;
    {
      // T1: y1 = x + LN1(att), in place over the attention rows (8 lanes per row, 4 columns per lane)
      const float4 a4 = *reinterpret_cast<const float4*>(att + row * ES_LDX + 4 * q);
      const float4 r4 = *reinterpret_cast<const float4*>(xs + row * ES_LDX + 4 * q);
      float v[4] = {a4.x, a4.y, a4.z, a4.w}, y1[4] = {r4.x, r4.y, r4.z, r4.w};
      float mu, rs;
      es_layernorm(v, eps1, mu, rs);
#pragma unroll
      for (int k = 0; k < 4; ++k) y1[k] += fmaf((v[k] - mu) * rs, prm[q * 4 + k], prm[ES_D + q * 4 + k]);
      *reinterpret_cast<float4*>(att + row * ES_LDX + 4 * q) = make_float4(y1[0], y1[1], y1[2], y1[3]);
      if (TRAIN && live) {
        *reinterpret_cast<float4*>(svr + 4 * q) = a4;
        *reinterpret_cast<float4*>(svr + 32 + 4 * q) = make_float4(y1[0], y1[1], y1[2], y1[3]);
      }
    }
    __syncthreads();
    const int trt = wave >> 1, tct = wave & 1;         // the wave's 16 x 16 tile of the two feed-forward products
    const int tcol = 16 * tct + c;
    {
      // T2: h0 = y1 W0^T + b0; a = Dropout(SiLU(h0)) -> xs region
      float a[8];
      es_ld8(att + (16 * trt + c) * ES_LDX + 8 * g, a);
      es_f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 8; ++t) acc = es_mfma(a[t], w0[t], acc);
      const float b0c = prm[2 * ES_D + tcol];
      const float scale = p_ffn > 0.f ? 1.f / (1.f - p_ffn) : 1.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rw = 16 * trt + 4 * g + r;
        const float h0 = acc[r] + b0c;
        if (TRAIN && rw < n) sv[((size_t)layer * N + n0 + rw) * ES_SV + 64 + tcol] = h0;
        float s = h0 * es_sigmoid(h0);
        if (p_ffn > 0.f) s = msde_uniform(seed_ffn, (unsigned long long)(n0 + rw) * ES_D + tcol) >= p_ffn ? s * scale : 0.f;
        xs[rw * ES_LDX + tcol] = s;
      }
    }
    __syncthreads();
    {
      // T3: x2 = a W3^T + b3 -> qk region (row stride ES_LDX)
      float a[8];
      es_ld8(xs + (16 * trt + c) * ES_LDX + 8 * g, a);
      es_f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 8; ++t) acc = es_mfma(a[t], w3[t], acc);
      const float b3c = prm[3 * ES_D + tcol];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rw = 16 * trt + 4 * g + r;
        const float x2 = acc[r] + b3c;
        if (TRAIN && rw < n) sv[((size_t)layer * N + n0 + rw) * ES_SV + 96 + tcol] = x2;
        qk[rw * ES_LDX + tcol] = x2;
      }
    }
    __syncthreads();
    {
      // T4: out = y1 + LN2(x2) [-> SiLU between the two convolutions of a block, :142] -> xs (rows >= n: zero)
      const float4 x4 = *reinterpret_cast<const float4*>(qk + row * ES_LDX + 4 * q);
      const float4 y4 = *reinterpret_cast<const float4*>(att + row * ES_LDX + 4 * q);
      float v[4] = {x4.x, x4.y, x4.z, x4.w}, y1[4] = {y4.x, y4.y, y4.z, y4.w};
      float mu, rs;
      es_layernorm(v, eps2, mu, rs);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float o = y1[k] + fmaf((v[k] - mu) * rs, prm[4 * ES_D + q * 4 + k], prm[5 * ES_D + q * 4 + k]);
        v[k] = live ? (ci == 0 ? o * es_sigmoid(o) : o) : 0.f;
      }
      *reinterpret_cast<float4*>(xs + row * ES_LDX + 4 * q) = make_float4(v[0], v[1], v[2], v[3]);
      if (TRAIN && live) *reinterpret_cast<float4*>(svr + 128 + 4 * q) = make_float4(v[0], v[1], v[2], v[3]);
    }
    if (ci == 1) {
      if (tid < ES_HC) hw[tid] = pb1;
      hw[ES_HC + tid] = pw2a;
      if (tid < ES_HC) hw[ES_HC + 256 + tid] = pw2b;
    }
    __syncthreads();
    ES_STAMP(5 + 8 * layer);
    if (ci == 1) {
      // basis MLP of block mi on every edge + frame mix + mean over the in-edges of the target (:150-166), transposed:
      //   P[atom][hidden] = h W1[:, :32]^T + b1 / 2                              (qk region; as the q|k|v|skip product)
      //   Z^T[16 ht + 4 g + r][edge c] = P[src] + P[dst] + sum_k W1[.][32 + k] edge_attr[edge][k]   (8 hidden tiles, 64 MFMAs)
      //   coff^T[j][edge c] = sum_hidden W2[j][hidden] SiLU(Z^T)[hidden][edge]  (the Z^T tiles ARE the B operands: 32 MFMAs;
      //   hidden index 16 ht + 4 g + r <-> MFMA (ht, r), k-group g on both operands)
      {
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
      }
      __syncthreads();
      const int ntile = (Em + 15) >> 4;
      float* mix = ee;                                   // [Em][3]
      const float b2j[3] = {W.bb2(mi)[0], W.bb2(mi)[1], W.bb2(mi)[2]};
      for (int rt = wave; rt < ntile; rt += 4) {
        const int el = 16 * rt + c;
        const bool on = el < Em;
        float bs[9];
        if (g == 0) {                                    // lanes that will hold the edge's three coefficients
          const float* bp = PRE ? ea + ((size_t)e0 + (on ? el : 0)) * ld_ea + 384 : basis + 9 * ((size_t)e0 + (on ? el : 0));
#pragma unroll
          for (int k = 0; k < 9; ++k) bs[k] = bp[k];
        }
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const float* pj = qk + (on ? sl[el] : 0) * ES_LDQ + 4 * g;
        const float* pi = qk + (on ? dl[el] : 0) * ES_LDQ + 4 * g;
        if (on && !PRE) {
          if (ea_lds) es_ld8(eal + el * ES_LDX + 8 * g, a);
          else es_ld8(ea + ((size_t)e0 + el) * ld_ea + 8 * g, a);
        }
        es_f4 acc[8];
        if (PRE) {
          // the edge half of the hidden layer comes precomputed: hidden 16 ht + 4 g + r of edge c = one float4 of its row
          const float* zr = ea + ((size_t)e0 + (on ? el : 0)) * ld_ea + 4 * ES_D + ES_HC * mi + 4 * g;
          float4 z4[8];
#pragma unroll
          for (int ht = 0; ht < 8; ++ht) z4[ht] = *reinterpret_cast<const float4*>(zr + 16 * ht);
#pragma unroll
          for (int ht = 0; ht < 8; ++ht) {
            const float4 u = *reinterpret_cast<const float4*>(pj + 16 * ht), w = *reinterpret_cast<const float4*>(pi + 16 * ht);
            acc[ht] = es_f4{(u.x + w.x) + z4[ht].x, (u.y + w.y) + z4[ht].y, (u.z + w.z) + z4[ht].z, (u.w + w.w) + z4[ht].w};
          }
        } else {
#pragma unroll
          for (int ht = 0; ht < 8; ++ht) {
            const float4 u = *reinterpret_cast<const float4*>(pj + 16 * ht), w = *reinterpret_cast<const float4*>(pi + 16 * ht);
            acc[ht] = es_f4{u.x + w.x, u.y + w.y, u.z + w.z, u.w + w.w};
          }
#pragma unroll
          for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int ht = 0; ht < 8; ++ht) acc[ht] = es_mfma(b1w[ht][t], a[t], acc[ht]);
        }
        es_f4 cf = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ht = 0; ht < 8; ++ht) {
          float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
          if (c < 3) w = *reinterpret_cast<const float4*>(hw + ES_HC + c * ES_HC + 16 * ht + 4 * g);
          const float wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float z = acc[ht][r];
            cf = es_mfma(wv[r], z * es_sigmoid(z), cf);
          }
        }
        if (g == 0 && on) {                              // rows 0..2 of the coefficient tile: registers 0..2 of lanes g == 0
          const float c0 = cf[0] + b2j[0], c1 = cf[1] + b2j[1], c2 = cf[2] + b2j[2];
          float* mo = mix + el * 3;                      // (c0 b_diff + c1 b_cross) + c2 b_vert, as written in the reference
          mo[0] = (c0 * bs[0] + c1 * bs[3]) + c2 * bs[6];
          mo[1] = (c0 * bs[1] + c1 * bs[4]) + c2 * bs[7];
          mo[2] = (c0 * bs[2] + c1 * bs[5]) + c2 * bs[8];
        }
      }
      __syncthreads();
      ES_STAMP(6 + 8 * layer);
      if (tid < n * 3) {
        const int i = tid / 3, k = tid - 3 * i;
        const int s0 = rp[i], s1 = rp[i + 1];
        float s = 0.f;
        for (int e = s0; e < s1; ++e) s += mix[e * 3 + k];
        gacc[tid] += s * (1.f / (float)max(s1 - s0, 1));
      }
      __syncthreads();
    }
  }
  if (tid < n * 3) out[(size_t)n0 * 3 + tid] = gacc[tid];
  ES_STAMP(40);
}

extern "C" long long msde_escore_mol_saved_floats(int N) { return (long long)ES_LAYERS * (long long)N * ES_SV; }

// params: DEVICE array of ES_NPTR device pointers (4 x 11 layer pointers, 2 x 4 basis-MLP pointers; struct EsW)
extern "C" int msde_escore_mol_fwd(const void* const* params, const float* x0, const float* edge_attr, int ld_ea,
                                   const float* basis, const int* mol_ptr, int B, const int* rowptr, const int* src,
                                   const int* dst, int N, int E, int hidden, int heads, int hidden_coff, int n_max, float p_att,
                                   float p_ffn, unsigned long long seed0, const unsigned long long* seed_dev, float eps1,
                                   float eps2, float* out, float* saved, void* stream) {
  if (!params || !x0 || !edge_attr || !basis || !mol_ptr || !rowptr || !src || !dst || !out || N < 0 || B < 0 || E < 0)
    return MSDE_EINVAL;
  // n_max: the largest molecule of the batch, stated by the caller (the kernel holds a molecule's rows in LDS: a larger one
  // would be silently cut to ES_NMAX atoms)
  if (hidden != ES_D || heads != 8 || hidden_coff != ES_HC || n_max > ES_NMAX) return MSDE_EUNSUP;
  if (ld_ea < ES_D || ld_ea % 4 || (reinterpret_cast<uintptr_t>(edge_attr) & 15) || (reinterpret_cast<uintptr_t>(x0) & 15))
    return MSDE_EINVAL;
  if (p_att < 0.f || p_att >= 1.f || p_ffn < 0.f || p_ffn >= 1.f) return MSDE_EINVAL;
  if (saved && (reinterpret_cast<uintptr_t>(saved) & 15)) return MSDE_EINVAL;
  EsW W{reinterpret_cast<const float* const*>(params)};
  if (N == 0 || B == 0) return 0;
  if (saved)
    MSDE_LAUNCH((escore_mol_fwd_kernel<true, false>), dim3(B), dim3(256), 0, as_stream(stream), W, x0, edge_attr, ld_ea, basis,
                mol_ptr, B, rowptr, src, dst, N, p_att, p_ffn, seed0, seed_dev, eps1, eps2, out, saved);
  else
    MSDE_LAUNCH((escore_mol_fwd_kernel<false, false>), dim3(B), dim3(256), 0, as_stream(stream), W, x0, edge_attr, ld_ea, basis,
                mol_ptr, B, rowptr, src, dst, N, p_att, p_ffn, seed0, seed_dev, eps1, eps2, out, saved);
  MSDE_CHECK_LAUNCH();
  return 0;
}

// get_score up to the division by -std (SDE_model_2D_to_3D.py:393-445) in two launches: escore_edge_pre_kernel fills `scratch`
// (msde_escore_mol_score_scratch_floats(E) floats), the per-molecule kernel (PRE) consumes it; inference: no dropout, nothing
// saved.  params: DEVICE array of 86 pointers = the 76 of msde_escore_mol_fwd + [dist_gaussian_fourier.W (unused when has_dist =
// 0), coff_gaussian_fourier.W, input_mlp weight [32,64] / bias, coff_mlp weight [32,128] / bias, project[0] weight [32,66] / bias,
// project[1] weight [32,32] / bias].
extern "C" long long msde_escore_mol_score_scratch_floats(int E) { return (long long)E * ES_PRE_LD; }

extern "C" int msde_escore_mol_score(const void* const* params, const float* x0, const float* pos, const float* edge_2D,
                                     int ld_e2d, int has_dist, const int* mol_ptr, int B, const int* rowptr, const int* src,
                                     const int* dst, int N, int E, int hidden, int heads, int hidden_coff, int n_max, float eps1,
                                     float eps2, float* scratch, float* out, void* stream) {
  if (!params || !x0 || !pos || !edge_2D || !mol_ptr || !rowptr || !src || !dst || !out || !scratch || N < 0 || B < 0 || E < 0)
    return MSDE_EINVAL;
  if (hidden != ES_D || heads != 8 || hidden_coff != ES_HC || n_max > ES_NMAX) return MSDE_EUNSUP;
  if (ld_e2d < ES_D || ld_e2d % 4 || ((reinterpret_cast<uintptr_t>(edge_2D) | reinterpret_cast<uintptr_t>(x0) | reinterpret_cast<uintptr_t>(scratch)) & 15))
    return MSDE_EINVAL;
  EsW W{reinterpret_cast<const float* const*>(params)};
  if (N == 0 || B == 0) return 0;
  const EsGeo geo{pos, edge_2D, ld_e2d, has_dist};
  if (E > 0)
    MSDE_LAUNCH(escore_edge_pre_kernel, dim3((unsigned)(((E + 15) / 16 + 3) / 4)), dim3(256), 0, as_stream(stream), W, geo, src, dst, E,
                scratch);
  MSDE_LAUNCH((escore_mol_fwd_kernel<false, true>), dim3(B), dim3(256), 0, as_stream(stream), W, x0, (const float*)scratch, ES_PRE_LD,
              (const float*)nullptr, mol_ptr, B, rowptr, src, dst, N, 0.f, 0.f, 0ull, (const unsigned long long*)nullptr, eps1, eps2,
              out, (float*)nullptr);
  MSDE_CHECK_LAUNCH();
  return 0;
}
