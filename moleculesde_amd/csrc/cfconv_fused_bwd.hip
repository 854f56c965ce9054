// cfconv_fused_bwd.hip — weight gradients of the CFConv filter network in ONE kernel (gfx950).
//
// For every radius edge e = (j -> i), with rbf = smearing(d_e), pre1 = W1 rbf + b1, h1 = ssp(pre1),
// filter = (W2 h1 + b2) * C(d_e) and agg_i = sum_e x1_j * filter (schnet.py:141-145,185-195):
//     g_pre2[e] = g_agg[i] * x1[j] * C(d_e)                      (needs NO recomputation of W2 h1)
//     g_W2     += g_pre2[e] (x) h1[e]        g_b2 += g_pre2[e]
//     g_h1[e]   = W2^T g_pre2[e]             g_pre1[e] = g_h1[e] * sigmoid(pre1[e])
//     g_W1     += g_pre1[e] (x) rbf[e]       g_b1 += g_pre1[e]
// The [E,128] intermediates never exist in HBM: each persistent workgroup (one per CU, 4 waves) walks
// its share of 64-edge chunks, recomputes rbf / h1 with fp32 MFMA, and keeps its partial g_W2 (128x128),
// g_W1 (128xG) and bias sums in MFMA accumulators; partial slabs are then summed over workgroups in a
// fixed order by a second kernel (bitwise reproducible, no float atomics).
//
// MFMA operand tricks (v_mfma_f32_32x32x2_f32; C/D map: col = lane&31, row = (r&3)+8(r>>2)+4(lane>>5)):
//   * a tile held in accumulator layout (column on the lane, 16 rows in registers) is used directly as
//     the A operand of a product that sums over its ROW index (edges): k-step s takes register s, and
//     the B operand is read from LDS at the SAME permuted edge row -> no LDS round trip for g_pre2/g_pre1;
//   * W2 sits once in LDS as [128][129]: the odd stride makes both its row-per-lane (W2 h1) and
//     column-per-lane (W2^T g) operand reads bank-conflict free.
#include "msde_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CB_F 128
#define CB_TE 64
#define CB_HS 129

__device__ __forceinline__ int cb_row(int s, int h) { return (s & 3) + 8 * (s >> 2) + 4 * h; }

template <int KK1>
__global__ void __launch_bounds__(256, 1)
cfconv_fused_bwd_w_kernel(const float* __restrict__ g_agg, const float* __restrict__ x1, const float* __restrict__ dist,
                          const int* __restrict__ rowptr, const int* __restrict__ src, const int* __restrict__ dst,
                          const float* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ W2,
                          const float* __restrict__ offset, int N, int G, float coeff, float cutoff, int cpw,
                          float* __restrict__ slabs, int dbg) {
  constexpr int RS = 2 * KK1 + 1;
  extern __shared__ float lds[];
  float* W2s = lds;                            // [128][129]
  float* rbf_t = W2s + CB_F * CB_HS;           // [64][RS]  (+ slack: reads up to column 63 of the last row)
  float* hid_t = rbf_t + CB_TE * RS + 64;      // [64][129] h1
  float* gp_t = hid_t + CB_TE * CB_HS;         // [64][129] g_pre2
  float* c_s = gp_t + CB_TE * CB_HS;           // [64]
  float* d_s = c_s + CB_TE;                    // [64]
  int* src_s = reinterpret_cast<int*>(d_s + CB_TE);
  int* dst_s = src_s + CB_TE;

  const float PI_F = 3.14159265358979323846f;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lcol = lane & 31, lhalf = lane >> 5;
  const int col = wave * 32 + lcol;

  const int E = rowptr[N];
  const int e_begin = min(blockIdx.x * cpw * CB_TE, E);
  const int e_end = min(e_begin + cpw * CB_TE, E);

  // W2 -> LDS (coalesced), W1 slice -> registers
  for (int t = tid; t < CB_F * CB_F; t += 256) W2s[(t >> 7) * CB_HS + (t & 127)] = W2[t];
  float w1r[KK1];
#pragma unroll
  for (int kk = 0; kk < KK1; ++kk) {
    int g = 2 * kk + lhalf;
    w1r[kk] = g < G ? W1[(size_t)col * G + g] : 0.f;
  }
  const float b1c = b1[col];

  f32x16 aW2[4], aW1[2];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) aW2[j][r] = 0.f;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) aW1[j][r] = 0.f;
  float sb1 = 0.f, sb2 = 0.f;

  // per-edge metadata is requested one chunk ahead (the gathers below depend on it: without the prefetch every
  // chunk starts with two back-to-back global round trips)
  float m_d = 0.f;
  int m_s = -1, m_t = -1;
  auto fetch_meta = [&](int ec) {
    if (tid < CB_TE) {
      int e = ec + tid;
      bool ok = e < e_end;
      m_d = ok ? dist[e] : -1.f;
      m_s = ok ? src[e] : -1;
      m_t = ok ? dst[e] : -1;
    }
  };
  fetch_meta(e_begin);
  for (int ec = e_begin; ec < e_end; ec += CB_TE) {
    __syncthreads();
    if (tid < CB_TE) {
      bool ok = m_t >= 0;
      d_s[tid] = ok ? m_d : 0.f;
      c_s[tid] = ok ? 0.5f * (cosf(m_d * PI_F / cutoff) + 1.0f) : 0.f;
      src_s[tid] = m_s;
      dst_s[tid] = m_t;
    }
    __syncthreads();
    if (ec + CB_TE < e_end) fetch_meta(ec + CB_TE);
    // g_pre2 in accumulator layout straight from the gathers: gp[rb][s] = g_agg[dst] * x1[src] * C
    float gp0[16], gp1[16];
    if (dbg & 1) {
#pragma unroll
      for (int s = 0; s < 16; ++s) { gp0[s] = c_s[cb_row(s, lhalf)]; gp1[s] = c_s[32 + cb_row(s, lhalf)]; }
    } else {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      int row = cb_row(s, lhalf);
      int s0 = src_s[row], s1 = src_s[32 + row], t0 = dst_s[row], t1 = dst_s[32 + row];
      float xa = x1[(size_t)(s0 >= 0 ? s0 : 0) * CB_F + col], ga = g_agg[(size_t)(t0 >= 0 ? t0 : 0) * CB_F + col];
      float xb = x1[(size_t)(s1 >= 0 ? s1 : 0) * CB_F + col], gb = g_agg[(size_t)(t1 >= 0 ? t1 : 0) * CB_F + col];
      gp0[s] = ga * xa * c_s[row];          // padding rows: c_s = 0
      gp1[s] = gb * xb * c_s[32 + row];
    }
    }
    // rbf tile
    if (!(dbg & 2))
    for (int idx = tid; idx < CB_TE * 2 * KK1; idx += 256) {
      int r = idx / (2 * KK1), g = idx % (2 * KK1);
      float v = 0.f;
      if (ec + r < e_end && g < G) {
        float diff = d_s[r] - offset[g];
        v = __expf(coeff * (diff * diff));
      }
      rbf_t[r * RS + g] = v;
    }
    __syncthreads();

    // ---- recompute pre1 = rbf W1^T (+ b1): h1 -> LDS, sigmoid(pre1) stays in registers
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    if (!(dbg & 64)) {
#pragma unroll
    for (int kk = 0; kk < KK1; ++kk) {
      float a0 = rbf_t[lcol * RS + 2 * kk + lhalf];
      float a1 = rbf_t[(32 + lcol) * RS + 2 * kk + lhalf];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, w1r[kk], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, w1r[kk], acc1, 0, 0, 0);
    }
    }
    float sg0[16], sg1[16];
    if (dbg & 4) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        int row = cb_row(i, lhalf);
        hid_t[row * CB_HS + col] = acc0[i];
        hid_t[(32 + row) * CB_HS + col] = acc1[i];
        sg0[i] = acc0[i]; sg1[i] = acc1[i];
        gp_t[row * CB_HS + col] = gp0[i];
        gp_t[(32 + row) * CB_HS + col] = gp1[i];
      }
    } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      int row = cb_row(i, lhalf);
      float p0 = acc0[i] + b1c, p1 = acc1[i] + b1c;
      float e0 = __expf(-fabsf(p0)), e1 = __expf(-fabsf(p1));
      hid_t[row * CB_HS + col] = fmaxf(p0, 0.f) + __logf(1.f + e0) - 0.69314718246459961f;
      hid_t[(32 + row) * CB_HS + col] = fmaxf(p1, 0.f) + __logf(1.f + e1) - 0.69314718246459961f;
      float r0 = 1.f / (1.f + e0), r1 = 1.f / (1.f + e1);
      sg0[i] = p0 >= 0.f ? r0 : e0 * r0;       // sigmoid = d softplus / dx
      sg1[i] = p1 >= 0.f ? r1 : e1 * r1;
      gp_t[row * CB_HS + col] = gp0[i];        // g_pre2 tile for the W2^T product (row-per-lane reads)
      gp_t[(32 + row) * CB_HS + col] = gp1[i];
      sb2 += gp0[i] + gp1[i];
    }
    }
    __syncthreads();

    // ---- g_W2[f][k] += sum_e g_pre2[e][f] h1[e][k]: A = g_pre2 registers, B = h1 rows (same permuted e)
    if (!(dbg & 8)) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      int row = cb_row(s, lhalf);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float bA = hid_t[row * CB_HS + 32 * j + lcol];
        float bB = hid_t[(32 + row) * CB_HS + 32 * j + lcol];
        aW2[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(gp0[s], bA, aW2[j], 0, 0, 0);
        aW2[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(gp1[s], bB, aW2[j], 0, 0, 0);
      }
    }
    }

    // ---- g_h1[e][k] = sum_f g_pre2[e][f] W2[f][k]
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    if (!(dbg & 16))
#pragma unroll
    for (int kk = 0; kk < CB_F / 2; ++kk) {
      float a0 = gp_t[lcol * CB_HS + 2 * kk + lhalf];
      float a1 = gp_t[(32 + lcol) * CB_HS + 2 * kk + lhalf];
      float b = W2s[(2 * kk + lhalf) * CB_HS + col];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc1, 0, 0, 0);
    }
    // g_pre1 = g_h1 * sigmoid(pre1)  (same lanes / registers as pre1)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      sg0[i] *= acc0[i];
      sg1[i] *= acc1[i];
      sb1 += sg0[i] + sg1[i];
    }
    // ---- g_W1[k][g] += sum_e g_pre1[e][k] rbf[e][g]
    if (!(dbg & 32))
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      int row = cb_row(s, lhalf);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float bA = rbf_t[row * RS + 32 * j + lcol];            // g >= 2*KK1 reads slack: columns discarded
        float bB = rbf_t[(32 + row) * RS + 32 * j + lcol];
        aW1[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(sg0[s], bA, aW1[j], 0, 0, 0);
        aW1[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(sg1[s], bB, aW1[j], 0, 0, 0);
      }
    }
  }

  // ---- write this workgroup's slab: [128*128 gW2][128*G gW1][128 gb1][128 gb2]
  const size_t slab_sz = (size_t)CB_F * CB_F + (size_t)CB_F * G + 2 * CB_F;
  float* slab = slabs + (size_t)blockIdx.x * slab_sz;
  if (dbg & 128) { if (aW2[0][0] + aW1[0][0] + sb1 + sb2 == 12345.678f) slab[0] = 1.f; return; }
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int f = wave * 32 + cb_row(r, lhalf);
      slab[(size_t)f * CB_F + 32 * j + lcol] = aW2[j][r];
    }
  float* sW1 = slab + (size_t)CB_F * CB_F;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int k = wave * 32 + cb_row(r, lhalf);
      int g = 32 * j + lcol;
      if (g < G) sW1[(size_t)k * G + g] = aW1[j][r];
    }
  // bias sums: combine the two lane halves (rows) of each column
  sb1 += __shfl_xor(sb1, 32, 64);
  sb2 += __shfl_xor(sb2, 32, 64);
  if (lhalf == 0) {
    slab[(size_t)CB_F * CB_F + (size_t)CB_F * G + col] = sb1;
    slab[(size_t)CB_F * CB_F + (size_t)CB_F * G + CB_F + col] = sb2;
  }
}

// ---- software-pipelined variant -------------------------------------------------------------------------------
// Phase timing of the kernel above (tools/bench_cfconv_bwd.py, MI355X, E = 49090, 256 workgroups, 75.6 us): the
// MFMA blocks cost 35 us and everything else 40.6 us (prologue 12.5, gathers 10, smearing 6, softplus/sigmoid 6.5, slab
// write 4-8), with NO overlap between the two: one wave per SIMD, and every phase separated by a barrier.  This variant
// keeps the arithmetic and the summation order, but
//   * W2 reaches LDS by 16-byte loads and W1 is staged through LDS (coalesced) instead of 26 strided loads per lane;
//   * chunk c+1's gathers are issued, and its smearing tile is computed into a SECOND tile buffer, inside the
//     g_W2 / g_h1 / g_W1 MFMA block of chunk c (MFMAs execute asynchronously: independent VALU / memory instructions
//     issue in their shadow); per-edge metadata runs two chunks ahead;
//   * two barriers per chunk instead of four.
// The only un-overlapped part of a chunk is recomputing pre1 (52 MFMAs) and the softplus / sigmoid that depends on it.
// SYM (csrc/cfconv_pair.hip): the rows are UNORDERED atom pairs {src, dst}; the filter gradient of a pair is the sum over
// both directions, g_pre2 = (g_agg[dst] x1[src] + g_agg[src] x1[dst]) C(d), formed before the weight-gradient products --
// half the rows, half the matrix-core work.  A negative distance marks a pair beyond the cutoff (C = 0).
// (body: workgroup `wg` of one filter network's launch; the kernels below bind it to blockIdx.x of a one-layer launch or to
// (layer, workgroup) of the all-layers launch)
template <int KK1, int DBG, bool W2R = false, bool SYM = false>
__device__ __forceinline__ void
cfconv_bwd_w_pipe_body(const float* __restrict__ g_agg, const float* __restrict__ x1, const float* __restrict__ dist,
                       const int* __restrict__ ecount, const int* __restrict__ src, const int* __restrict__ dst,
                       const float* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ W2,
                       const float* __restrict__ offset, int N, int G, float coeff, float cutoff, int cpw,
                       float* __restrict__ slabs, const int wg) {
  constexpr int dbg = DBG;                     // phase knock-outs for tools/bench_cfconv_bwd.py (0 in the product)
  constexpr int RS = 2 * KK1 + 1;
  constexpr int RBF_SZ = CB_TE * RS + 64;      // + slack: the gW1 product reads up to column 63 of the last row
  extern __shared__ float lds[];
  // W2R: W2 as 64 registers per lane (B operand of the W2^T product) instead of a [128][129] LDS image: the workgroup
  // then needs 96 KB of LDS instead of 158 KB and leaves room on its CU for the other stream's LDS-using kernels
  float* W2s = lds;                            // [128][129]
  float* rbf_t = W2R ? lds : W2s + CB_F * CB_HS;   // [2][64][RS]
  float* hid_t = rbf_t + 2 * RBF_SZ;           // [64][129] h1
  float* gp_t = hid_t + CB_TE * CB_HS;         // [64][129] g_pre2
  float* c_s = gp_t + CB_TE * CB_HS;           // [2][64]
  float* d_s = c_s + 2 * CB_TE;                // [2][64]
  int* src_s = reinterpret_cast<int*>(d_s + 2 * CB_TE);   // [2][64]
  int* dst_s = src_s + 2 * CB_TE;              // [2][64]

  const float PI_F = 3.14159265358979323846f;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lcol = lane & 31, lhalf = lane >> 5;
  const int col = wave * 32 + lcol;

  const int E = ecount[0];                     // rows: radius edges (rowptr[N]) or pairs
  const int e_begin = min(wg * cpw * CB_TE, E);
  const int e_end = min(e_begin + cpw * CB_TE, E);
  const int nchunks = (e_end - e_begin + CB_TE - 1) / CB_TE;

  // ---- prologue: metadata of chunk 0, W2 -> LDS (16-byte loads), W1 slice -> registers through LDS
  float m_d = 0.f;
  int m_s = -1, m_t = -1;
  auto fetch_meta = [&](int c) {               // registers of threads 0..63; lands while other work runs
    if (tid < CB_TE) {
      const int e = e_begin + c * CB_TE + tid;
      const bool ok = c < nchunks && e < e_end;
      m_d = ok ? dist[e] : -1.f;
      m_s = ok ? src[e] : -1;
      m_t = ok ? dst[e] : -1;
    }
  };
  auto store_meta = [&](int c) {
    if (tid < CB_TE) {
      const int o = (c & 1) * CB_TE + tid;
      const bool ok = m_t >= 0;
      d_s[o] = ok ? m_d : 0.f;
      c_s[o] = (ok && m_d >= 0.f) ? 0.5f * (cosf(m_d * PI_F / cutoff) + 1.0f) : 0.f;
      src_s[o] = max(m_s, 0) * (CB_F * 4);     // BYTE offset of the gathered row: one add per load in the gathers
      dst_s[o] = max(m_t, 0) * (CB_F * 4);
    }
  };
  fetch_meta(0);
  float w2r[W2R ? CB_F / 2 : 1];
  if (W2R) {
#pragma unroll
    for (int kk = 0; kk < (W2R ? CB_F / 2 : 1); ++kk) w2r[kk] = W2[(size_t)(2 * kk + lhalf) * CB_F + col];
  }
  {
    const float4* W2v = reinterpret_cast<const float4*>(W2);
    float4 w[CB_F * CB_F / 4 / 256];
    if (!W2R) {
#pragma unroll
      for (int i = 0; i < CB_F * CB_F / 4 / 256; ++i) w[i] = W2v[tid + 256 * i];
    }
    float* stage = hid_t;                      // W1 [128][G] (<= 32 KB) fits in hid_t
    // fixed trip counts: every load is issued before the first store (a run-time bound serialises 26 round trips)
    float w1v[CB_F * 2 * KK1 / 256];
#pragma unroll
    for (int i = 0; i < CB_F * 2 * KK1 / 256; ++i) w1v[i] = tid + 256 * i < CB_F * G ? W1[tid + 256 * i] : 0.f;
#pragma unroll
    for (int i = 0; i < CB_F * 2 * KK1 / 256; ++i)
      if (tid + 256 * i < CB_F * G) stage[tid + 256 * i] = w1v[i];
    if (!W2R) {
#pragma unroll
      for (int i = 0; i < CB_F * CB_F / 4 / 256; ++i) {
        const int t = (tid + 256 * i) * 4;     // row t >> 7, columns (t & 127) .. +3 (odd row stride: scalar stores)
        float* q = &W2s[(t >> 7) * CB_HS + (t & 127)];
        q[0] = w[i].x; q[1] = w[i].y; q[2] = w[i].z; q[3] = w[i].w;
      }
    }
  }
  store_meta(0);
  __syncthreads();
  float w1r[KK1];
#pragma unroll
  for (int kk = 0; kk < KK1; ++kk) {
    const int g = 2 * kk + lhalf;
    w1r[kk] = g < G ? hid_t[col * G + g] : 0.f;
  }
  const float b1c = b1[col];
  fetch_meta(1);

  f32x16 aW2[4], aW1[2];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) aW2[j][r] = 0.f;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) aW1[j][r] = 0.f;
  float sb1 = 0.f, sb2 = 0.f;

  // gathers of one chunk: x1[src] and g_agg[dst] for this lane's column and its 2 x 16 edge rows
  float nx[32], ng[32];
  float nx2[SYM ? 32 : 1], ng2[SYM ? 32 : 1];   // SYM: the reverse direction x1[dst], g_agg[src]
  auto issue_gathers = [&](int c) {
    if (dbg & 1) return;
    const int* ss = src_s + (c & 1) * CB_TE;
    const int* ts = dst_s + (c & 1) * CB_TE;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int row = cb_row(s, lhalf);
      // 32-bit byte offsets (N * 512 < 2^31): scalar base + one VGPR, one v_add per load
      auto at = [&](const float* base, int row_off) {
        return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + (unsigned)(row_off + 4 * col));
      };
      nx[s] = at(x1, ss[row]);
      ng[s] = at(g_agg, ts[row]);
      nx[16 + s] = at(x1, ss[32 + row]);
      ng[16 + s] = at(g_agg, ts[32 + row]);
      if (SYM) {
        nx2[SYM ? s : 0] = at(x1, ts[row]);
        ng2[SYM ? s : 0] = at(g_agg, ss[row]);
        nx2[SYM ? 16 + s : 0] = at(x1, ts[32 + row]);
        ng2[SYM ? 16 + s : 0] = at(g_agg, ss[32 + row]);
      }
    }
  };
  // Smearing tile of chunk c, 4 rows per call (`it` in [0, 16)): lane = column g, wave + 4 it = row, so the centre is a
  // register, the distance a broadcast LDS read, and nothing here branches (the calls sit between MFMAs and must not split
  // the block the scheduler interleaves).  Rows past e_end and column G..2 KK1 - 1 hold finite values that only meet
  // zeros (g_pre2 = 0 there, W1 columns >= G are zero, gW1 columns >= G are not stored); lanes >= 2 KK1 repeat lane
  // 2 KK1 - 1 (same value, same address).
  const float coeff2 = coeff * 1.4426950408889634f;
  const int rbf_g = min(lane, 2 * KK1 - 1);
  const float rbf_mu = rbf_g < G ? offset[rbf_g] : 0.f;
  auto rbf_elem = [&](int c, int it) {
    if (dbg & 2) return;
    const int r = wave + 4 * it;
    const float diff = d_s[(c & 1) * CB_TE + r] - rbf_mu;
    rbf_t[(c & 1) * RBF_SZ + r * RS + rbf_g] = __builtin_amdgcn_exp2f(coeff2 * (diff * diff));   // raw v_exp_f32
  };
  // Per-lane LDS bases: every access below is base + COMPILE-TIME offset (ds_read / ds_write immediates).  Written as
  // row * stride + col the compiler materialises one address VGPR per access, hoists them all out of the chunk loop and
  // spills (240 B of scratch, ~40 % of the VALU instructions were address arithmetic).
  constexpr auto RW = [](int i) constexpr { return (i & 3) + 8 * (i >> 2); };   // cb_row without the lane-half term
  float* const hw = hid_t + 4 * lhalf * CB_HS + col;         // h1 stores            + RW(i) * HS (+ 32 HS)
  float* const gw = gp_t + 4 * lhalf * CB_HS + col;          // g_pre2 stores
  const float* const hr = hid_t + 4 * lhalf * CB_HS + lcol;  // h1 rows, B of gW2    + RW(s) * HS + 32 j (+ 32 HS)
  const float* const gr = gp_t + lcol * CB_HS + lhalf;       // g_pre2, A of g_h1    + 2 kk (+ 32 HS)
  const float* const wr = W2s + lhalf * CB_HS + col;         // W2, B of g_h1        + 2 kk * HS
  const int rb_a = lcol * RS + lhalf;                        // smearing, A of pre1  + 2 kk (+ 32 RS)
  const int rb_b = 4 * lhalf * RS + lcol;                    // smearing, B of gW1   + RW(s) * RS + 32 j (+ 32 RS)
  float gp0[16], gp1[16], sg0[16], sg1[16];
  auto finish_gathers = [&](int c) {           // g_pre2 in accumulator layout: g_agg[dst] * x1[src] * C (padding: C = 0)
    const float* cs = c_s + (c & 1) * CB_TE;
    if (dbg & 1) {
#pragma unroll
      for (int s = 0; s < 16; ++s) { gp0[s] = cs[cb_row(s, lhalf)]; gp1[s] = cs[32 + cb_row(s, lhalf)]; }
      return;
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int row = cb_row(s, lhalf);
      if (SYM) {
        gp0[s] = fmaf(ng[s], nx[s], ng2[SYM ? s : 0] * nx2[SYM ? s : 0]) * cs[row];
        gp1[s] = fmaf(ng[16 + s], nx[16 + s], ng2[SYM ? 16 + s : 0] * nx2[SYM ? 16 + s : 0]) * cs[32 + row];
      } else {
        gp0[s] = ng[s] * nx[s] * cs[row];
        gp1[s] = ng[16 + s] * nx[16 + s] * cs[32 + row];
      }
    }
  };
  // recompute pre1 = rbf W1^T + b1 of chunk c: h1 and g_pre2 -> LDS, sigmoid(pre1) -> registers
  auto hidden = [&](int c) {
    const float* rb = rbf_t + (c & 1) * RBF_SZ + rb_a;
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    // LDS operands are requested one k-step ahead of the MFMAs that use them (an MFMA occupies the pipe for 64 cycles, an
    // LDS read takes longer: read-then-use leaves the pipe idle every step)
    float pa[2][2];
    pa[0][0] = rb[0];
    pa[0][1] = rb[32 * RS];
    if (!(dbg & 64))
#pragma unroll
    for (int kk = 0; kk < KK1; ++kk) {
      if (kk + 1 < KK1) {
        pa[(kk + 1) & 1][0] = rb[2 * (kk + 1)];
        pa[(kk + 1) & 1][1] = rb[32 * RS + 2 * (kk + 1)];
      }
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[kk & 1][0], w1r[kk], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[kk & 1][1], w1r[kk], acc1, 0, 0, 0);
    }
    if (dbg & 4) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = cb_row(i, lhalf);
        hw[RW(i) * CB_HS] = acc0[i];
        hw[(32 + RW(i)) * CB_HS] = acc1[i];
        sg0[i] = acc0[i]; sg1[i] = acc1[i];
        gw[RW(i) * CB_HS] = gp0[i];
        gw[(32 + RW(i)) * CB_HS] = gp1[i];
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = cb_row(i, lhalf);
      // raw v_exp_f32 / v_log_f32 / v_rcp_f32 (1 ulp): the fp32 MFMA leaves NO issue shadow for vector instructions
      // (profiles/r02_probe_fp32_mfma_fillers.txt), so every VALU instruction here is paid in full: ~12 per element
      // instead of the ~30 of the range-checked __expf / __logf / division
      const float p0 = acc0[i] + b1c, p1 = acc1[i] + b1c;
      const float e0 = __builtin_amdgcn_exp2f(-1.4426950408889634f * fabsf(p0));
      const float e1 = __builtin_amdgcn_exp2f(-1.4426950408889634f * fabsf(p1));
      const float u0 = 1.f + e0, u1 = 1.f + e1;
      hw[RW(i) * CB_HS] = fmaf(__builtin_amdgcn_logf(u0), 0.69314718055994531f, fmaxf(p0, 0.f) - 0.69314718055994531f);
      hw[(32 + RW(i)) * CB_HS] = fmaf(__builtin_amdgcn_logf(u1), 0.69314718055994531f, fmaxf(p1, 0.f) - 0.69314718055994531f);
      const float r0 = __builtin_amdgcn_rcpf(u0), r1 = __builtin_amdgcn_rcpf(u1);
      sg0[i] = p0 >= 0.f ? r0 : e0 * r0;       // sigmoid = d softplus / dx
      sg1[i] = p1 >= 0.f ? r1 : e1 * r1;
      gw[RW(i) * CB_HS] = gp0[i];              // g_pre2 tile for the W2^T product (row-per-lane reads)
      gw[(32 + RW(i)) * CB_HS] = gp1[i];
      sb2 += gp0[i] + gp1[i];
    }
  };

  if (nchunks > 0) {
    issue_gathers(0);
#pragma unroll
    for (int it = 0; it < 16; ++it) rbf_elem(0, it);
    finish_gathers(0);
    __syncthreads();                           // smearing tile 0 visible; W1 staging (hid_t) no longer read
    hidden(0);
    store_meta(1);
    fetch_meta(2);
    __syncthreads();
  }

  for (int c = 0; c < nchunks; ++c) {
    const float* rb = rbf_t + (c & 1) * RBF_SZ + rb_b;
    // past the last chunk the look-ahead work below runs on stale (valid) metadata and its results are never read:
    // cheaper than branches, which would split the block the MFMA / VALU interleaving lives in
    issue_gathers(c + 1);
    // ---- g_W2[f][k] += sum_e g_pre2[e][f] h1[e][k]: A = g_pre2 registers, B = h1 rows (same permuted e);
    //      the next chunk's smearing tile is computed in the shadow of these MFMAs
    float hb[2][8];
    auto read_h = [&](int s, float (&q)[8]) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        q[2 * j] = hr[RW(s) * CB_HS + 32 * j];
        q[2 * j + 1] = hr[(32 + RW(s)) * CB_HS + 32 * j];
      }
    };
    read_h(0, hb[0]);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      if (s + 1 < 16) read_h(s + 1, hb[(s + 1) & 1]);
      if (!(dbg & 8))
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        aW2[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(gp0[s], hb[s & 1][2 * j], aW2[j], 0, 0, 0);
        aW2[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(gp1[s], hb[s & 1][2 * j + 1], aW2[j], 0, 0, 0);
      }
      rbf_elem(c + 1, s);
    }
    finish_gathers(c + 1);                     // g_pre2 of chunk c is dead: its registers take chunk c+1's
    // ---- g_h1[e][k] = sum_f g_pre2[e][f] W2[f][k]
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    float ga[2][3];
    auto read_g = [&](int kk, float (&q)[3]) {
      q[0] = gr[2 * kk];
      q[1] = gr[32 * CB_HS + 2 * kk];
      q[2] = W2R ? w2r[W2R ? kk : 0] : wr[2 * kk * CB_HS];
    };
    read_g(0, ga[0]);
    if (!(dbg & 16))
#pragma unroll
    for (int kk = 0; kk < CB_F / 2; ++kk) {
      if (kk + 1 < CB_F / 2) read_g(kk + 1, ga[(kk + 1) & 1]);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[kk & 1][0], ga[kk & 1][2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[kk & 1][1], ga[kk & 1][2], acc1, 0, 0, 0);
    }
    // g_pre1 = g_h1 * sigmoid(pre1)  (same lanes / registers as pre1)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      sg0[i] *= acc0[i];
      sg1[i] *= acc1[i];
      sb1 += sg0[i] + sg1[i];
    }
    // ---- g_W1[k][g] += sum_e g_pre1[e][k] rbf[e][g]
    float rq[2][4];
    auto read_r = [&](int s, float (&q)[4]) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        q[2 * j] = rb[RW(s) * RS + 32 * j];                       // g >= 2*KK1 reads slack: columns discarded
        q[2 * j + 1] = rb[(32 + RW(s)) * RS + 32 * j];
      }
    };
    read_r(0, rq[0]);
    if (!(dbg & 32))
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      if (s + 1 < 16) read_r(s + 1, rq[(s + 1) & 1]);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        aW1[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(sg0[s], rq[s & 1][2 * j], aW1[j], 0, 0, 0);
        aW1[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(sg1[s], rq[s & 1][2 * j + 1], aW1[j], 0, 0, 0);
      }
    }
    if (c + 1 < nchunks) {
      __syncthreads();                         // everyone is done with h1 / g_pre2 of chunk c; smearing tile c+1 visible
      hidden(c + 1);
      store_meta(c + 2);
      fetch_meta(c + 3);
      __syncthreads();
    }
  }

  // ---- write this workgroup's slab: [128*128 gW2][128*G gW1][128 gb1][128 gb2]
  const size_t slab_sz = (size_t)CB_F * CB_F + (size_t)CB_F * G + 2 * CB_F;
  float* slab = slabs + (size_t)wg * slab_sz;
  if (dbg & 128) { if (aW2[0][0] + aW1[0][0] + sb1 + sb2 == 12345.678f) slab[0] = 1.f; return; }
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int f = wave * 32 + cb_row(r, lhalf);
      slab[(size_t)f * CB_F + 32 * j + lcol] = aW2[j][r];
    }
  float* sW1 = slab + (size_t)CB_F * CB_F;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int k = wave * 32 + cb_row(r, lhalf);
      int g = 32 * j + lcol;
      if (g < G) sW1[(size_t)k * G + g] = aW1[j][r];
    }
  sb1 += __shfl_xor(sb1, 32, 64);
  sb2 += __shfl_xor(sb2, 32, 64);
  if (lhalf == 0) {
    slab[(size_t)CB_F * CB_F + (size_t)CB_F * G + col] = sb1;
    slab[(size_t)CB_F * CB_F + (size_t)CB_F * G + CB_F + col] = sb2;
  }
}

template <int KK1, int DBG, bool W2R = false, bool SYM = false>
__global__ void __launch_bounds__(256, 1)
cfconv_fused_bwd_w_pipe_kernel(const float* __restrict__ g_agg, const float* __restrict__ x1, const float* __restrict__ dist,
                               const int* __restrict__ ecount, const int* __restrict__ src, const int* __restrict__ dst,
                               const float* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ W2,
                               const float* __restrict__ offset, int N, int G, float coeff, float cutoff, int cpw,
                               float* __restrict__ slabs) {
  cfconv_bwd_w_pipe_body<KK1, DBG, W2R, SYM>(g_agg, x1, dist, ecount, src, dst, W1, b1, W2, offset, N, G, coeff, cutoff, cpw, slabs,
                                             (int)blockIdx.x);
}

// Pair form for SEVERAL interaction blocks in one launch: the filter-network weight gradients of a block feed nothing in the
// backward chain (they are parameter gradients: slabs for the batched reduction), so the blocks' launches -- 383 chunks of 64
// pairs each at bs 256, 2.2 per workgroup at the width the step gives them: every workgroup rounds up to 3 -- are collected
// and run as ONE launch of L x 383 chunks once the chain has produced the last block's gradient: 13 per workgroup, rounded to
// 14.  Workgroup b serves layer b / wpl with the pair chunks (b % wpl) * cpw ...; per-layer slabs [wpl][slab].
struct cb_multi_ptrs {
  const float* g[MSDE_CFCONV_MAX_LAYERS];
  const float* x1[MSDE_CFCONV_MAX_LAYERS];
  const float* W1[MSDE_CFCONV_MAX_LAYERS];
  const float* b1[MSDE_CFCONV_MAX_LAYERS];
  const float* W2[MSDE_CFCONV_MAX_LAYERS];
  float* slabs[MSDE_CFCONV_MAX_LAYERS];
};

template <int KK1>
__global__ void __launch_bounds__(256, 1)
cfconv_pair_bwd_w_multi_kernel(const cb_multi_ptrs m, const float* __restrict__ pd, const int* __restrict__ count,
                               const int* __restrict__ pi, const int* __restrict__ pj, const float* __restrict__ offset, int N,
                               int G, float coeff, float cutoff, int cpw, int wpl) {
  const int l = blockIdx.x / wpl, wg = blockIdx.x - l * wpl;
  cfconv_bwd_w_pipe_body<KK1, 0, true, true>(m.g[l], m.x1[l], pd, count, pj, pi, m.W1[l], m.b1[l], m.W2[l], offset, N, G, coeff,
                                             cutoff, cpw, m.slabs[l], wg);
}

__global__ void cfconv_reduce_slabs_kernel(const float* __restrict__ slabs, int nslab, size_t slab_sz, int G,
                                           float* __restrict__ gW2, float* __restrict__ gW1, float* __restrict__ gb1,
                                           float* __restrict__ gb2) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < slab_sz; i += (size_t)gridDim.x * blockDim.x) {
    float acc = 0.f;
    int z = 0;
    for (; z + 4 <= nslab; z += 4) {
      float a0 = slabs[(size_t)z * slab_sz + i], a1 = slabs[(size_t)(z + 1) * slab_sz + i];
      float a2 = slabs[(size_t)(z + 2) * slab_sz + i], a3 = slabs[(size_t)(z + 3) * slab_sz + i];
      acc = (((acc + a0) + a1) + a2) + a3;
    }
    for (; z < nslab; ++z) acc += slabs[(size_t)z * slab_sz + i];
    size_t n2 = (size_t)CB_F * CB_F, n1 = (size_t)CB_F * G;
    if (i < n2) gW2[i] = acc;
    else if (i < n2 + n1) gW1[i - n2] = acc;
    else if (i < n2 + n1 + CB_F) gb1[i - n2 - n1] = acc;
    else gb2[i - n2 - n1 - CB_F] = acc;
  }
}

// max_wgs <= 0: one persistent workgroup per CU (each takes 147 KB of LDS, i.e. the whole CU as far as other
// LDS-using kernels are concerned).  A caller that runs this kernel BESIDE latency-critical work on another stream
// passes a smaller number: the kernel takes longer but leaves whole CUs to the other stream (measured on the
// pretrain step: 128 workgroups instead of 256 = +6 % step throughput).
static inline bool cb_pipe() { return true; }   // the software-pipelined kernel (round 2: 57 vs 76 us); the round-1 kernel below is no longer reachable
static inline void cb_geometry(int E_cap, int max_wgs, int* nwg, int* cpw) {
  const int te = CB_TE;
  int chunks = (E_cap + te - 1) / te;
  int cap = max_wgs > 0 ? max_wgs : msde_num_cus();
  int w = chunks < cap ? chunks : cap;
  if (w < 1) w = 1;
  *cpw = (chunks + w - 1) / w;
  if (*cpw < 1) *cpw = 1;
  *nwg = (chunks + *cpw - 1) / *cpw;
  if (*nwg < 1) *nwg = 1;
}

extern "C" int msde_cfconv_fused_bwd_w_slabs(int E_cap, int max_workgroups) {
  int nwg, cpw;
  cb_geometry(E_cap, max_workgroups, &nwg, &cpw);
  return nwg;
}

extern "C" long long msde_cfconv_fused_bwd_w_workspace_floats(int E_cap, int G, int max_workgroups) {
  int nwg, cpw;
  cb_geometry(E_cap, max_workgroups, &nwg, &cpw);
  return (long long)nwg * ((long long)CB_F * CB_F + (long long)CB_F * G + 2 * CB_F);
}

extern "C" int msde_cfconv_fused_bwd_w(const float* g_agg, const float* x1, const float* dist, const int* rowptr,
                                       const int* src, const int* dst, const float* W1, const float* b1,
                                       const float* W2, const float* offset, int N, int F, int G, int E_cap,
                                       float coeff, float cutoff, int max_workgroups, float* gW1, float* gb1, float* gW2,
                                       float* gb2, float* workspace, void* stream) {
  // gW1 == gb1 == gW2 == gb2 == NULL: leave the per-workgroup slabs in `workspace` for a batched reduction
  const bool no_reduce = !gW1 && !gb1 && !gW2 && !gb2;
  if (N < 0 || E_cap < 0 || !g_agg || !x1 || !dist || !rowptr || !src || !dst || !W1 || !b1 || !W2 || !offset ||
      !workspace || (!no_reduce && (!gW1 || !gb1 || !gW2 || !gb2)))
    return MSDE_EINVAL;
  if (F != CB_F || G <= 0 || G > 64) return MSDE_EUNSUP;
  hipStream_t st = as_stream(stream);
  int nwg, cpw;
  cb_geometry(E_cap, max_workgroups, &nwg, &cpw);
  int kk1 = (G + 1) / 2;
  const int dbg = 0;    // (phase-skipping diagnostics of round 2: compile-time only now)
  auto lds_bytes = [](int KK1) {
    return (size_t)(CB_F * CB_HS + CB_TE * (2 * KK1 + 1) + 64 + 2 * CB_TE * CB_HS + 4 * CB_TE) * sizeof(float);
  };
#define CB_LAUNCH(KK)                                                                                                 \
  {                                                                                                                   \
    static bool attr_done = false;                                                                                    \
    if (!attr_done) {                                                                                                 \
      hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void*>(&cfconv_fused_bwd_w_kernel<KK>),              \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes(KK));            \
      if (ae != hipSuccess) return (int)ae;                                                                           \
      attr_done = true;                                                                                               \
    }                                                                                                                 \
  }                                                                                                                   \
  MSDE_LAUNCH(cfconv_fused_bwd_w_kernel<KK>, dim3(nwg), dim3(256), lds_bytes(KK), st, g_agg, x1, dist, rowptr, src, dst, \
              W1, b1, W2, offset, N, G, coeff, cutoff, cpw, workspace, dbg)
  // W2 in registers (96 KB of LDS per workgroup) by default: the same speed alone (56.6 vs 57.4 us) and the step no longer
  // loses 1.8 % when the kernel runs at full width beside the main chain (0.7 %); MSDE_CFBWD_W2REG=0: W2 in LDS (158 KB)
  constexpr int w2reg = 1;
  auto ldsp_bytes = [](int KK1) {
    return (size_t)((w2reg ? 0 : CB_F * CB_HS) + 2 * (CB_TE * (2 * KK1 + 1) + 64) + 2 * CB_TE * CB_HS + 8 * CB_TE + 64) *
           sizeof(float);
  };
#define CBP_LAUNCH(KK)                                                                                                \
  {                                                                                                                   \
    static bool attr_done = false;                                                                                    \
    if (!attr_done) {                                                                                                 \
      hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void*>(&cfconv_fused_bwd_w_pipe_kernel<KK, 0>),         \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsp_bytes(KK));           \
      if (ae != hipSuccess) return (int)ae;                                                                           \
      ae = hipFuncSetAttribute(reinterpret_cast<const void*>(&cfconv_fused_bwd_w_pipe_kernel<KK, 0, true>),            \
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsp_bytes(KK));                      \
      if (ae != hipSuccess) return (int)ae;                                                                           \
      attr_done = true;                                                                                               \
    }                                                                                                                 \
  }                                                                                                                   \
  if (w2reg)                                                                                                          \
    MSDE_LAUNCH((cfconv_fused_bwd_w_pipe_kernel<KK, 0, true>), dim3(nwg), dim3(256), ldsp_bytes(KK), st, g_agg, x1, dist,  \
                rowptr + N, src, dst, W1, b1, W2, offset, N, G, coeff, cutoff, cpw, workspace);                       \
  else                                                                                                                \
    MSDE_LAUNCH((cfconv_fused_bwd_w_pipe_kernel<KK, 0>), dim3(nwg), dim3(256), ldsp_bytes(KK), st, g_agg, x1, dist, rowptr + N, \
                src, dst, W1, b1, W2, offset, N, G, coeff, cutoff, cpw, workspace)
#ifdef MSDE_CF_DIAG          // phase knock-outs of the pipelined kernel as separate instantiations (no run-time branches)
#define CBP_DIAG(D_)                                                                                                  \
  case D_:                                                                                                            \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cfconv_fused_bwd_w_pipe_kernel<26, D_>),                 \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsp_bytes(26));                       \
    MSDE_LAUNCH((cfconv_fused_bwd_w_pipe_kernel<26, D_>), dim3(nwg), dim3(256), ldsp_bytes(26), st, g_agg, x1, dist,    \
                rowptr + N, src, dst, W1, b1, W2, offset, N, G, coeff, cutoff, cpw, workspace);                       \
    MSDE_CHECK_LAUNCH();                                                                                              \
    return 0;
  if (cb_pipe() && kk1 == 26 && dbg) {
    switch (dbg) {
      CBP_DIAG(1) CBP_DIAG(2) CBP_DIAG(4) CBP_DIAG(7) CBP_DIAG(8) CBP_DIAG(16) CBP_DIAG(32) CBP_DIAG(64) CBP_DIAG(120)
      CBP_DIAG(127) CBP_DIAG(128) CBP_DIAG(255)
      default: break;
    }
  }
#undef CBP_DIAG
#endif
  if (cb_pipe() && kk1 <= 26 && !dbg) {
    if (kk1 == 26) { CBP_LAUNCH(26); }
    else if (kk1 == 25) { CBP_LAUNCH(25); }
    else { CBP_LAUNCH(24); }
  } else {
    if (kk1 == 26) { CB_LAUNCH(26); }
    else if (kk1 == 25) { CB_LAUNCH(25); }
    else { CB_LAUNCH(32); }
  }
#undef CB_LAUNCH
#undef CBP_LAUNCH
  MSDE_CHECK_LAUNCH();
  if (no_reduce) return 0;
  size_t slab_sz = (size_t)CB_F * CB_F + (size_t)CB_F * G + 2 * CB_F;
  // outputs laid out like a slab ([gW2 | gW1 | gb1 | gb2] in one buffer): generic 16-lane parallel reduction
  if (gW1 == gW2 + (size_t)CB_F * CB_F && gb1 == gW1 + (size_t)CB_F * G && gb2 == gb1 + CB_F)
    return msde_reduce_slabs(workspace, nwg, slab_sz, gW2, nullptr, 0, nullptr, st);
  int blocks = (int)((slab_sz + 255) / 256);
  MSDE_LAUNCH(cfconv_reduce_slabs_kernel, dim3(blocks), dim3(256), 0, st, (const float*)workspace, nwg, slab_sz, G, gW2,
              gW1, gb1, gb2);
  MSDE_CHECK_LAUNCH();
  return 0;
}

// Pair form (csrc/cfconv_pair.hip): rows = unordered pairs (pi, pj, pd), *count of them valid; slabs / workspace sizes from
// msde_cfconv_fused_bwd_w_slabs / _workspace_floats with E_cap = P_cap.  W2 as registers, pipelined kernel only.
extern "C" int msde_cfconv_pair_bwd_w(const float* g_agg, const float* x1, const float* pd, const int* count, const int* pi,
                                      const int* pj, const float* W1, const float* b1, const float* W2,
                                      const float* offset, int N, int F, int G, int P_cap, float coeff, float cutoff,
                                      int max_workgroups, float* gW1, float* gb1, float* gW2, float* gb2,
                                      float* workspace, void* stream) {
  const bool no_reduce = !gW1 && !gb1 && !gW2 && !gb2;
  if (N < 0 || P_cap < 0 || !g_agg || !x1 || !pd || !count || !pi || !pj || !W1 || !b1 || !W2 || !offset || !workspace ||
      (!no_reduce && (!gW1 || !gb1 || !gW2 || !gb2)))
    return MSDE_EINVAL;
  const int kk1 = (G + 1) / 2;
  if (F != CB_F || G <= 0 || kk1 > 26) return MSDE_EUNSUP;
  hipStream_t st = as_stream(stream);
  int nwg, cpw;
  cb_geometry(P_cap, max_workgroups, &nwg, &cpw);
  auto ldsp_bytes = [](int KK1) {
    return (size_t)(2 * (CB_TE * (2 * KK1 + 1) + 64) + 2 * CB_TE * CB_HS + 8 * CB_TE + 64) * sizeof(float);
  };
#define CBS_LAUNCH(KK)                                                                                                \
  {                                                                                                                   \
    static bool attr_done = false;                                                                                    \
    if (!attr_done) {                                                                                                 \
      hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void*>(&cfconv_fused_bwd_w_pipe_kernel<KK, 0, true, true>), \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsp_bytes(KK));           \
      if (ae != hipSuccess) return (int)ae;                                                                           \
      attr_done = true;                                                                                               \
    }                                                                                                                 \
  }                                                                                                                   \
  MSDE_LAUNCH((cfconv_fused_bwd_w_pipe_kernel<KK, 0, true, true>), dim3(nwg), dim3(256), ldsp_bytes(KK), st, g_agg, x1, pd, \
              count, pj, pi, W1, b1, W2, offset, N, G, coeff, cutoff, cpw, workspace)
  if (kk1 == 26) { CBS_LAUNCH(26); }
  else if (kk1 == 25) { CBS_LAUNCH(25); }
  else { CBS_LAUNCH(24); }
#undef CBS_LAUNCH
  MSDE_CHECK_LAUNCH();
  if (no_reduce) return 0;
  const size_t slab_sz = (size_t)CB_F * CB_F + (size_t)CB_F * G + 2 * CB_F;
  if (gW1 == gW2 + (size_t)CB_F * CB_F && gb1 == gW1 + (size_t)CB_F * G && gb2 == gb1 + CB_F)
    return msde_reduce_slabs(workspace, nwg, slab_sz, gW2, nullptr, 0, nullptr, st);
  const int blocks = (int)((slab_sz + 255) / 256);
  MSDE_LAUNCH(cfconv_reduce_slabs_kernel, dim3(blocks), dim3(256), 0, st, (const float*)workspace, nwg, slab_sz, G, gW2,
              gW1, gb1, gb2);
  MSDE_CHECK_LAUNCH();
  return 0;
}

// ---- all-layers pair form (cfconv_pair_bwd_w_multi_kernel above) -----------------------------------------------------------
// geometry: L layers share `max_workgroups` persistent workgroups (0: one per CU): every layer gets the same number wpl, each
// workgroup cpw chunks of 64 pairs of ITS layer
static inline void cb_multi_geometry(int P_cap, int L, int max_wgs, int* wpl, int* cpw) {
  const int chunks = (P_cap + CB_TE - 1) / CB_TE;
  const int total = max_wgs > 0 ? max_wgs : msde_num_cus();
  int w = total / (L > 0 ? L : 1);
  if (w < 1) w = 1;
  if (w > chunks) w = chunks;
  if (w < 1) w = 1;
  *cpw = (chunks + w - 1) / w;
  if (*cpw < 1) *cpw = 1;
  *wpl = (chunks + *cpw - 1) / *cpw;
  if (*wpl < 1) *wpl = 1;
}

extern "C" int msde_cfconv_pair_bwd_w_multi_slabs(int P_cap, int L, int max_workgroups) {
  int wpl, cpw;
  cb_multi_geometry(P_cap, L, max_workgroups, &wpl, &cpw);
  return wpl;
}

extern "C" int msde_cfconv_pair_bwd_w_multi(const float* const* g_agg, const float* const* x1, const float* pd, const int* count,
                                            const int* pi, const int* pj, const float* const* W1, const float* const* b1,
                                            const float* const* W2, const float* offset, int L, int N, int F, int G, int P_cap,
                                            float coeff, float cutoff, int max_workgroups, float* const* slabs, void* stream) {
  if (N < 0 || P_cap < 0 || L < 0 || !g_agg || !x1 || !pd || !count || !pi || !pj || !W1 || !b1 || !W2 || !offset || !slabs)
    return MSDE_EINVAL;
  const int kk1 = (G + 1) / 2;
  if (F != CB_F || G <= 0 || kk1 > 26 || L > MSDE_CFCONV_MAX_LAYERS) return MSDE_EUNSUP;
  if (L == 0) return 0;
  cb_multi_ptrs m;
  for (int l = 0; l < L; ++l) {
    if (!g_agg[l] || !x1[l] || !W1[l] || !b1[l] || !W2[l] || !slabs[l]) return MSDE_EINVAL;
    m.g[l] = g_agg[l]; m.x1[l] = x1[l]; m.W1[l] = W1[l]; m.b1[l] = b1[l]; m.W2[l] = W2[l]; m.slabs[l] = slabs[l];
  }
  int wpl, cpw;
  cb_multi_geometry(P_cap, L, max_workgroups, &wpl, &cpw);
  auto ldsp_bytes = [](int KK1) {
    return (size_t)(2 * (CB_TE * (2 * KK1 + 1) + 64) + 2 * CB_TE * CB_HS + 8 * CB_TE + 64) * sizeof(float);
  };
  hipStream_t st = as_stream(stream);
#define CBM_LAUNCH(KK)                                                                                                \
  {                                                                                                                   \
    static bool attr_done = false;                                                                                    \
    if (!attr_done) {                                                                                                 \
      hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void*>(&cfconv_pair_bwd_w_multi_kernel<KK>),         \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsp_bytes(KK));           \
      if (ae != hipSuccess) return (int)ae;                                                                           \
      attr_done = true;                                                                                               \
    }                                                                                                                 \
  }                                                                                                                   \
  MSDE_LAUNCH(cfconv_pair_bwd_w_multi_kernel<KK>, dim3(wpl * L), dim3(256), ldsp_bytes(KK), st, m, pd, count, pi, pj, offset, N, \
              G, coeff, cutoff, cpw, wpl)
  if (kk1 == 26) { CBM_LAUNCH(26); }
  else if (kk1 == 25) { CBM_LAUNCH(25); }
  else { CBM_LAUNCH(24); }
#undef CBM_LAUNCH
  MSDE_CHECK_LAUNCH();
  return 0;
}
