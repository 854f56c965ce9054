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

// body shared by the one-layer launch and the all-layers launch below: the pair blocks [p_begin, p_end) of ONE filter network
template <int KK1>
__device__ __forceinline__ void
cp_filter_body(const float* __restrict__ pd, const int P, const float* __restrict__ W1, const float* __restrict__ b1,
               const float* __restrict__ W2, const float* __restrict__ b2, const float* __restrict__ offset, int G,
               float coeff, float cutoff, const int p_begin, int cpw, float* __restrict__ Wf) {
  constexpr int RS = 2 * KK1 + 1;
  extern __shared__ float lds[];
  float* rbf_t = lds;                          // [32][RS]
  float* hid_t = rbf_t + CP_TE * RS;           // [32][129]
  float* c_s = hid_t + CP_TE * CP_HS;          // [2][32] cutoff per pair row (0: padding / beyond the cutoff)

  const float PI_F = 3.14159265358979323846f;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lcol = lane & 31, lhalf = lane >> 5;
  const int col = wave * 32 + lcol;
  if (p_begin >= P) return;
  const int p_end = min(p_begin + cpw * CP_TE, P);

  // weights -> registers (B operands), straight from the nn.Linear layouts W1 [F][G], W2 [F][F].  The 32x32x2 MFMA sums its two
  // k lanes (lane halves), so ANY pairing of reduction indices with (k step, lane half) gives the same product as long as the A
  // operand uses the same pairing: lane half h takes the CONTIGUOUS reduction range [64 h, 64 h + 64) of layer 2 and
  // [KK1 h, KK1 h + KK1) of layer 1.  A lane's B fragments are then 64 (26) consecutive floats of ONE weight row: sixteen
  // 16-byte loads (26 scalar ones: a W1 row is G = 51 floats, not 16-byte aligned) and no LDS staging -- the round-3 prologue moved
  // the same 90 KB per workgroup global -> registers -> LDS -> registers behind twelve barriers (23.9 -> 21.4 us per launch).
  float w1r[KK1], w2r[CP_F / 2];
  {
    const float* w2p = W2 + (size_t)col * CP_F + 64 * lhalf;
    if ((reinterpret_cast<uintptr_t>(W2) & 15) == 0) {
#pragma unroll
      for (int j = 0; j < CP_F / 8; ++j) {
        const float4 v = reinterpret_cast<const float4*>(w2p)[j];
        w2r[4 * j] = v.x; w2r[4 * j + 1] = v.y; w2r[4 * j + 2] = v.z; w2r[4 * j + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < CP_F / 2; ++kk) w2r[kk] = w2p[kk];
    }
    const int g0 = KK1 * lhalf;
    const float* w1p = W1 + (size_t)col * G + g0;
#pragma unroll
    for (int kk = 0; kk < KK1; ++kk) w1r[kk] = g0 + kk < G ? w1p[kk] : 0.f;
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
  const float* const ra = rbf_t + lcol * RS + KK1 * lhalf;      // A operands: row = pair, this half's reduction range (see above)
  const float* const ha = hid_t + lcol * CP_HS + 64 * lhalf;

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
    for (int kk = 0; kk < KK1; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[kk], w1r[kk], acc, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 16; ++i) hw[RW(i) * CP_HS] = cp_ssp(acc[i] + b1c);
    __syncthreads();   // B2
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int kk = 0; kk < CP_F / 2; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ha[kk], w2r[kk], acc, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = RW(i) + 4 * lhalf;
      if (pc + row < ce) Wf[(size_t)(pc + row) * CP_F + col] = (acc[i] + b2c) * cb[RW(i)];
    }
    if (more) produce_math();
  }
}

template <int KK1>
__global__ void __launch_bounds__(256, 2)
cfconv_pair_filter_kernel(const float* __restrict__ pd, const int* __restrict__ count, const float* __restrict__ W1,
                          const float* __restrict__ b1, const float* __restrict__ W2, const float* __restrict__ b2,
                          const float* __restrict__ offset, int G, float coeff, float cutoff, int cpw,
                          float* __restrict__ Wf) {
  cp_filter_body<KK1>(pd, count[0], W1, b1, W2, b2, offset, G, coeff, cutoff, blockIdx.x * cpw * CP_TE, cpw, Wf);
}

// The filter rows of ALL interaction blocks in one launch.  Wf_l(d) depends on the pair distances and on block l's filter
// network only -- not on the atom features -- so nothing in the layer chain has to wait for it (schnet.py:141-145 computes
// W = self.mlp(edge_attr) * C inside every block; the inputs edge_attr / edge_weight are the same tensors for all blocks,
// schnet.py:96-104).  Workgroup b serves layer b / wpl, pair blocks (b % wpl) * cpw ...: L x more blocks per launch than a
// layer of its own, so the split over the CUs is even (one layer: 766 blocks, 2 or 4 per CU) and the SchNet forward chain is
// four launches per block (lin1, aggregate, lin2 + ssp, lin + residual) instead of five.
struct cp_multi_ptrs {
  const float* W1[MSDE_CFCONV_MAX_LAYERS];
  const float* b1[MSDE_CFCONV_MAX_LAYERS];
  const float* W2[MSDE_CFCONV_MAX_LAYERS];
  const float* b2[MSDE_CFCONV_MAX_LAYERS];
  float* Wf[MSDE_CFCONV_MAX_LAYERS];
};

template <int KK1>
__global__ void __launch_bounds__(256, 2)
cfconv_pair_filter_multi_kernel(const float* __restrict__ pd, const int* __restrict__ count, const cp_multi_ptrs m,
                                const float* __restrict__ offset, int G, float coeff, float cutoff, int cpw, int wpl) {
  const int l = blockIdx.x / wpl, r = blockIdx.x - l * wpl;
  cp_filter_body<KK1>(pd, count[0], m.W1[l], m.b1[l], m.W2[l], m.b2[l], offset, G, coeff, cutoff, r * cpw * CP_TE, cpw, m.Wf[l]);
}

extern "C" int msde_cfconv_pair_filter_multi(const float* pd, const int* count, const float* const* W1, const float* const* b1,
                                             const float* const* W2, const float* const* b2, const float* offset, int L, int F,
                                             int G, int P_cap, float coeff, float cutoff, int blocks_per_wg, float* const* Wf,
                                             void* stream) {
  if (P_cap < 0 || L < 0 || !pd || !count || !W1 || !b1 || !W2 || !b2 || !offset || !Wf) return MSDE_EINVAL;
  if (F != CP_F || G <= 0 || G > 64 || L > MSDE_CFCONV_MAX_LAYERS) return MSDE_EUNSUP;
  if (P_cap == 0 || L == 0) return 0;
  cp_multi_ptrs m;
  for (int l = 0; l < L; ++l) {
    if (!W1[l] || !b1[l] || !W2[l] || !b2[l] || !Wf[l]) return MSDE_EINVAL;
    m.W1[l] = W1[l]; m.b1[l] = b1[l]; m.W2[l] = W2[l]; m.b2[l] = b2[l]; m.Wf[l] = Wf[l];
  }
  const int kk1 = (G + 1) / 2;
  const int blocks = (P_cap + CP_TE - 1) / CP_TE;
  if (blocks_per_wg <= 0) {
    // ~ three rounds of two workgroups per CU over the whole launch: short enough workgroups that the dispatcher evens the
    // CUs out, long enough (>= 3 blocks) that the 90 KB weight prologue of a workgroup stays a small part of it
    const long total = (long)blocks * L, slots = 6L * msde_num_cus();
    blocks_per_wg = (int)((total + slots - 1) / slots);
    if (blocks_per_wg < 3) blocks_per_wg = 3;
  }
  const int wpl = (blocks + blocks_per_wg - 1) / blocks_per_wg;
  auto lds_bytes = [](int KK1) { return (size_t)(CP_TE * (2 * KK1 + 1) + CP_TE * CP_HS + 2 * CP_TE) * sizeof(float); };
  hipStream_t st = as_stream(stream);
#define CPM_LAUNCH(KK)                                                                                                 \
  MSDE_LAUNCH(cfconv_pair_filter_multi_kernel<KK>, dim3(wpl * L), dim3(256), lds_bytes(KK), st, pd, count, m, offset, G, coeff, \
              cutoff, blocks_per_wg, wpl)
  if (kk1 == 26) { CPM_LAUNCH(26); }
  else if (kk1 == 25) { CPM_LAUNCH(25); }
  else { CPM_LAUNCH(32); }
#undef CPM_LAUNCH
  MSDE_CHECK_LAUNCH();
  return 0;
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
