// cfconv_pair.hip — CFConv on UNORDERED atom pairs (gfx950).
//
// SchNet's radius graph (schnet.py:91-93: radius_graph(pos, r = cutoff, batch), 32-neighbour cap) is symmetric whenever
// the cap cannot bind (molecules of at most 33 atoms), and the continuous filter of an edge,
//     Wf_ij = (W2 ssp(W1 rbf(d_ij) + b1) + b2) * C(d_ij)                      (schnet.py:141-145,185-195)
// depends on the distance only: Wf_ij == Wf_ji bit for bit.  The filter network -- 99 % of CFConv's arithmetic -- therefore
// runs once per unordered pair {i, j}: half the matrix-core work of the per-edge kernels in cfconv_fused.hip, forward and
// backward.  The pairs of a molecule are ALL i < j of its atoms in row-major order (pair p of molecule m, local atoms
// a < b:  pair_ptr[m] + a n - a (a + 1) / 2 + (b - a - 1)); a pair beyond the cutoff carries distance -1 and the filter row
// 0, which contributes exactly nothing -- so no radius graph, no CSR transposition and no neighbour lists are built at
// all, the index of a pair follows from the two atom numbers.
//
//   msde_pair_build             pair_ptr, (i, j), distances                 (once per forward; replaces radius_graph)
//   msde_cfconv_pair_filter     Wf [P, 128] on the matrix cores             (fp32 MFMA 32x32x2, 32 pairs per block)
//   msde_cfconv_pair_aggregate  out_i = sum_{j != i} x_j * Wf_{ij}          (forward with x = x1; input gradient with
//                               x = g_agg: the pair set and Wf are symmetric) -- fixed order, no atomics, no memset
//   msde_cfconv_pair_bwd_w      filter-network weight gradients: g_pre2_{ij} = (g_i x1_j + g_j x1_i) C(d_ij) summed BEFORE
//                               the two weight-gradient products (csrc/cfconv_fused_bwd.hip, SYM instantiation)
#include "msde_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CP_F 128
#define CP_TE 32
#define CP_HS 129             // odd row stride: ds_read_b32 / ds_write_b32 bank = (a / 4) % 32 per 32-lane half, so a lane-per-row access
                              // (bank = row + const) and a lane-per-column access (bank = column + const) are both conflict free

// ---- pair list ------------------------------------------------------------------------------------------------------
// one workgroup per molecule: base = pairs of all earlier molecules (B <= a few thousand: summed on the spot), then the
// molecule's pairs, row-major over a < b
__global__ void __launch_bounds__(256)
pair_build_kernel(const float* __restrict__ pos, const int* __restrict__ mol_ptr, int B, float r2,
                  int* __restrict__ pair_ptr, int* __restrict__ pi, int* __restrict__ pj, float* __restrict__ pd, int P_cap,
                  int* __restrict__ err) {
  __shared__ int part[256];
  const int m = blockIdx.x, tid = threadIdx.x;
  int s = 0;
  for (int q = tid; q < m; q += 256) {
    const int n = mol_ptr[q + 1] - mol_ptr[q];
    s += n * (n - 1) / 2;
  }
  part[tid] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) part[tid] += part[tid + o];
    __syncthreads();
  }
  const int base = part[0];
  const int a0 = mol_ptr[m], n = mol_ptr[m + 1] - a0;
  const int cnt = n * (n - 1) / 2;
  if (tid == 0) {
    pair_ptr[m] = base;
    if (m == B - 1) pair_ptr[B] = min(base + cnt, P_cap);
    if (base + cnt > P_cap && err) atomicExch(err, 1);     // the list is truncated: the batch is not a valid one
  }
  for (int idx = tid; idx < cnt; idx += 256) {
    // row a of the strict upper triangle starts at a n - a (a + 1) / 2: invert with a float root, then fix up
    const float tn = 2.f * n - 1.f;
    int a = (int)((tn - sqrtf(fmaxf(tn * tn - 8.f * idx, 0.f))) * 0.5f);
    a = max(0, min(a, n - 2));
    while (a > 0 && a * n - a * (a + 1) / 2 > idx) --a;
    while (a < n - 2 && (a + 1) * n - (a + 1) * (a + 2) / 2 <= idx) ++a;
    const int b = idx - (a * n - a * (a + 1) / 2) + a + 1;
    const int i = a0 + a, j = a0 + b, p = base + idx;
    if (p < P_cap) {
      const float dx = pos[3 * i] - pos[3 * j], dy = pos[3 * i + 1] - pos[3 * j + 1], dz = pos[3 * i + 2] - pos[3 * j + 2];
      const float d2 = dx * dx + dy * dy + dz * dz;
      pi[p] = i;
      pj[p] = j;
      pd[p] = d2 < r2 ? sqrtf(d2) : -1.f;          // strict <, as torch_cluster.radius; -1: no edge
    }
  }
}

extern "C" int msde_pair_build(const float* pos, const int* mol_ptr, int B, float r2, int* pair_ptr, int* pi, int* pj,
                               float* pd, int P_cap, int* err, void* stream) {
  if (B < 0 || P_cap < 0 || !pos || !mol_ptr || !pair_ptr || !pi || !pj || !pd) return MSDE_EINVAL;
  if (B == 0) return msde_zero_words(pair_ptr, 1, as_stream(stream));
  MSDE_LAUNCH(pair_build_kernel, dim3(B), dim3(256), 0, as_stream(stream), pos, mol_ptr, B, r2, pair_ptr, pi, pj, pd, P_cap,
              err);
  MSDE_CHECK_LAUNCH();
  return 0;
}

// ---- filter rows on the matrix cores ------------------------------------------------------------------------------
// The pair list is cut into blocks of 32 pairs (one 32-row MFMA block); a workgroup (4 waves) takes `cpw` consecutive
// blocks.  Wave w owns filter columns [32 w, 32 w + 32): its slices of W1 (G x 32) and W2 (128 x 32) stay in VGPRs as B
// operands for the whole kernel; the A operands (smearing tile, then the softplus'd hidden tile) are shared through LDS
// with odd row strides.  The structure is the forward kernel of cfconv_fused.hip without its gathers, messages and
// segmented sum: per block 26 + 64 MFMAs, two barriers, the next block's distances requested right after the first.
__device__ __forceinline__ float cp_ssp(float x) {
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * fabsf(x));
  return fmaf(__builtin_amdgcn_logf(1.f + e), 0.69314718055994531f, fmaxf(x, 0.f) - 0.69314718055994531f);
}

template <int KK1>
__global__ void __launch_bounds__(256, 2)
cfconv_pair_filter_kernel(const float* __restrict__ pd, const int* __restrict__ count, const float* __restrict__ W1,
                          const float* __restrict__ b1, const float* __restrict__ W2, const float* __restrict__ b2,
                          const float* __restrict__ offset, int G, float coeff, float cutoff, int cpw,
                          float* __restrict__ Wf) {
  constexpr int RS = 2 * KK1 + 1;
  extern __shared__ float lds[];
  float* rbf_t = lds;                          // [32][RS]
  float* hid_t = rbf_t + CP_TE * RS;           // [32][129]
  float* c_s = hid_t + CP_TE * CP_HS;          // [2][32] cutoff per pair row (0: padding / beyond the cutoff)

  const float PI_F = 3.14159265358979323846f;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lcol = lane & 31, lhalf = lane >> 5;
  const int col = wave * 32 + lcol;
  const int P = count[0];
  const int p_begin = blockIdx.x * cpw * CP_TE;
  if (p_begin >= P) return;
  const int p_end = min(p_begin + cpw * CP_TE, P);

  // weights -> registers (B operands) from the nn.Linear layouts W1 [F][G], W2 [F][F], staged through LDS
  float w1r[KK1], w2r[CP_F / 2];
  {
    float* stage = hid_t;
    const bool vec = (reinterpret_cast<uintptr_t>(W2) & 15) == 0;
    float4 v2[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i4 = tid + 256 * j;
        const float* srcp = W2 + (size_t)(32 * r + (i4 >> 5)) * CP_F + 4 * (i4 & 31);
        v2[r][j] = vec ? *reinterpret_cast<const float4*>(srcp) : make_float4(srcp[0], srcp[1], srcp[2], srcp[3]);
      }
    constexpr int W1J = 64 * 64 / 256;
    float v1[2][W1J];
    const int half_n = 64 * G;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int j = 0; j < W1J; ++j) {
        const int i = tid + 256 * j;
        v1[r][j] = i < half_n ? W1[(size_t)r * half_n + i] : 0.f;
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i4 = tid + 256 * j;
        // (a staged row is stored column-permuted, column 4 c + k at position 32 k + c: the 32 lanes of a half then cover 32
        // banks in each of the four store instructions -- at position 4 c + k they covered 8, a 4-way conflict on every staging
        // store, which was the 30 % SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of round 4's counters; the loop itself is conflict free)
        float* dd = stage + (i4 >> 5) * CP_HS + (i4 & 31);
        dd[0] = v2[r][j].x; dd[32] = v2[r][j].y; dd[64] = v2[r][j].z; dd[96] = v2[r][j].w;
      }
      __syncthreads();
      if (wave == r) {
#pragma unroll
        for (int kk = 0; kk < CP_F / 2; ++kk)          // column 2 kk + lhalf = 4 c + k sits at 32 k + c
          w2r[kk] = stage[lcol * CP_HS + 32 * lhalf + 64 * (kk & 1) + (kk >> 1)];
      }
      __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
      for (int j = 0; j < W1J; ++j) {
        const int i = tid + 256 * j;
        if (i < half_n) stage[i] = v1[r][j];
      }
      __syncthreads();
      if ((wave >> 1) == r) {
        const int lrow = (wave & 1) * 32 + lcol;
#pragma unroll
        for (int kk = 0; kk < KK1; ++kk) {
          const int g = 2 * kk + lhalf;
          w1r[kk] = g < G ? stage[lrow * G + g] : 0.f;
        }
      }
      __syncthreads();
    }
  }
  const float b1c = b1[col], b2c = b2[col];

  constexpr int NRW = CP_TE / 4;
  float rv[NRW];
  float m_d = -1.f;
  const int rbf_g = min(lane, 2 * KK1 - 1);
  const float rbf_mu = rbf_g < G ? offset[rbf_g] : 0.f;
  const float coeff2 = coeff * 1.4426950408889634f;
  auto produce_load = [&](int pc) {
    const int ce = min(pc + CP_TE, p_end);
#pragma unroll
    for (int j = 0; j < NRW; ++j) {
      const int r = wave + 4 * j;
      rv[j] = pc + r < ce ? pd[pc + r] : -1.f;
    }
    if (tid < CP_TE) m_d = pc + tid < ce ? pd[pc + tid] : -1.f;
  };
  auto produce_math = [&]() {
#pragma unroll
    for (int j = 0; j < NRW; ++j) {
      const float diff = rv[j] - rbf_mu;
      const float v = __builtin_amdgcn_exp2f(coeff2 * (diff * diff));
      rv[j] = (rv[j] >= 0.f && rbf_g < G) ? v : 0.f;
    }
  };
  produce_load(p_begin);
  produce_math();
  constexpr auto RW = [](int i) constexpr { return (i & 3) + 8 * (i >> 2); };
  float* const hw = hid_t + 4 * lhalf * CP_HS + col;
  const float* const ra = rbf_t + lcol * RS + lhalf;
  const float* const ha = hid_t + lcol * CP_HS + lhalf;

  int buf = 0;
  for (int pc = p_begin; pc < p_end; pc += CP_TE, buf ^= 1) {
    const int ce = min(pc + CP_TE, p_end);
#pragma unroll
    for (int j = 0; j < NRW; ++j)
      if (lane < 2 * KK1) rbf_t[(wave + 4 * j) * RS + rbf_g] = rv[j];
    if (tid < CP_TE) c_s[buf * CP_TE + tid] = m_d >= 0.f ? 0.5f * (__cosf(m_d * (PI_F / cutoff)) + 1.0f) : 0.f;
    __syncthreads();   // B1
    const float* cb = c_s + buf * CP_TE + 4 * lhalf;
    const bool more = pc + CP_TE < p_end;
    if (more) produce_load(pc + CP_TE);

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int kk = 0; kk < KK1; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[2 * kk], w1r[kk], acc, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 16; ++i) hw[RW(i) * CP_HS] = cp_ssp(acc[i] + b1c);
    __syncthreads();   // B2
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int kk = 0; kk < CP_F / 2; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ha[2 * kk], w2r[kk], acc, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = RW(i) + 4 * lhalf;
      if (pc + row < ce) Wf[(size_t)(pc + row) * CP_F + col] = (acc[i] + b2c) * cb[RW(i)];
    }
    if (more) produce_math();
  }
}

extern "C" int msde_cfconv_pair_filter(const float* pd, const int* count, const float* W1, const float* b1, const float* W2,
                                       const float* b2, const float* offset, int F, int G, int P_cap, float coeff,
                                       float cutoff, int blocks_per_wg, float* Wf, void* stream) {
  if (P_cap < 0 || !pd || !count || !W1 || !b1 || !W2 || !b2 || !offset || !Wf) return MSDE_EINVAL;
  if (F != CP_F || G <= 0 || G > 64) return MSDE_EUNSUP;
  if (P_cap == 0) return 0;
  const int kk1 = (G + 1) / 2;
  const int blocks = (P_cap + CP_TE - 1) / CP_TE;
  if (blocks_per_wg <= 0) {
    const int resident = msde_num_cus() * 2;
    blocks_per_wg = (blocks + resident - 1) / resident;
  }
  const int grid = (blocks + blocks_per_wg - 1) / blocks_per_wg;
  auto lds_bytes = [](int KK1) { return (size_t)(CP_TE * (2 * KK1 + 1) + CP_TE * CP_HS + 2 * CP_TE) * sizeof(float); };
  hipStream_t st = as_stream(stream);
#define CP_LAUNCH(KK)                                                                                                  \
  MSDE_LAUNCH(cfconv_pair_filter_kernel<KK>, dim3(grid), dim3(256), lds_bytes(KK), st, pd, count, W1, b1, W2, b2, offset, G, \
              coeff, cutoff, blocks_per_wg, Wf)
  if (kk1 == 26) { CP_LAUNCH(26); }
  else if (kk1 == 25) { CP_LAUNCH(25); }
  else { CP_LAUNCH(32); }
#undef CP_LAUNCH
  MSDE_CHECK_LAUNCH();
  return 0;
}

// ---- aggregation ------------------------------------------------------------------------------------------------------
// out[i] = sum over the other atoms j of i's molecule, ascending, of x[j] * Wf[pair(i, j)].  One wave per target: lane half
// h takes the neighbours of parity h, lane & 31 owns four of the 128 columns; two rows per array in flight per half; the
// two halves are added at the end (fixed order: bitwise reproducible).  Atoms past the last molecule (capacity padding)
// get zeros.
__global__ void __launch_bounds__(256)
cfconv_pair_aggregate_kernel(const float* __restrict__ x, const float* __restrict__ Wf, const int* __restrict__ batch,
                             const int* __restrict__ mol_ptr, const int* __restrict__ pair_ptr, int N, int B,
                             float* __restrict__ out) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= N) return;
  const int lane = threadIdx.x & 63, h = lane >> 5, q = lane & 31;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const int m = batch[i];
  if (m >= 0 && m < B) {
    const int a0 = mol_ptr[m], n = mol_ptr[m + 1] - a0, base = pair_ptr[m];
    const int a = i - a0;
    const int cnt = pair_ptr[B];                // pairs actually listed (msde_pair_build truncates at its capacity)
    const float4* __restrict__ X = reinterpret_cast<const float4*>(x);
    const float4* __restrict__ W = reinterpret_cast<const float4*>(Wf);
    auto pid = [&](int b) {                     // pair of local atoms a and b != a
      const int lo = min(a, b), hi = max(a, b);
      return base + lo * n - lo * (lo + 1) / 2 + (hi - lo - 1);
    };
    // neighbour k = 0 .. n - 2 is local atom b = k + (k >= a); this half takes k = h, h + 2, ...
    int k = h;
    for (; k + 2 < n - 1; k += 4) {
      const int b0 = k + (k >= a), b1 = k + 2 + (k + 2 >= a);
      const int p0 = pid(b0), p1 = pid(b1);
      const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 x0 = X[(size_t)(a0 + b0) * 32 + q], w0 = p0 < cnt ? W[(size_t)p0 * 32 + q] : zero;
      const float4 x1 = X[(size_t)(a0 + b1) * 32 + q], w1 = p1 < cnt ? W[(size_t)p1 * 32 + q] : zero;
      acc.x = fmaf(x0.x, w0.x, acc.x); acc.y = fmaf(x0.y, w0.y, acc.y); acc.z = fmaf(x0.z, w0.z, acc.z); acc.w = fmaf(x0.w, w0.w, acc.w);
      acc.x = fmaf(x1.x, w1.x, acc.x); acc.y = fmaf(x1.y, w1.y, acc.y); acc.z = fmaf(x1.z, w1.z, acc.z); acc.w = fmaf(x1.w, w1.w, acc.w);
    }
    for (; k < n - 1; k += 2) {
      const int b0 = k + (k >= a);
      const int p0 = pid(b0);
      const float4 x0 = X[(size_t)(a0 + b0) * 32 + q], w0 = p0 < cnt ? W[(size_t)p0 * 32 + q] : make_float4(0.f, 0.f, 0.f, 0.f);
      acc.x = fmaf(x0.x, w0.x, acc.x); acc.y = fmaf(x0.y, w0.y, acc.y); acc.z = fmaf(x0.z, w0.z, acc.z); acc.w = fmaf(x0.w, w0.w, acc.w);
    }
  }
  acc.x += __shfl_xor(acc.x, 32, 64);
  acc.y += __shfl_xor(acc.y, 32, 64);
  acc.z += __shfl_xor(acc.z, 32, 64);
  acc.w += __shfl_xor(acc.w, 32, 64);
  if (h == 0) reinterpret_cast<float4*>(out)[(size_t)i * 32 + q] = acc;
}

extern "C" int msde_cfconv_pair_aggregate(const float* x, const float* Wf, const int* batch, const int* mol_ptr,
                                          const int* pair_ptr, int N, int B, int F, float* out, void* stream) {
  if (N < 0 || B < 0 || !x || !Wf || !batch || !mol_ptr || !pair_ptr || !out) return MSDE_EINVAL;
  if (F != CP_F) return MSDE_EUNSUP;
  if (N == 0) return 0;
  MSDE_LAUNCH(cfconv_pair_aggregate_kernel, dim3((N + 3) / 4), dim3(256), 0, as_stream(stream), x, Wf, batch, mol_ptr,
              pair_ptr, N, B, out);
  MSDE_CHECK_LAUNCH();
  return 0;
}
