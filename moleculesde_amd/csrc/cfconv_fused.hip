// cfconv_fused.hip — fused CFConv forward for gfx950 (schnet.py:141-145,185-195):
//   rbf(d) -> Linear(G,F) -> ShiftedSoftplus -> Linear(F,F) -> * C(d) -> * x1[src] -> segmented sum
// in ONE kernel with fp32 MFMA (v_mfma_f32_32x32x2_f32: exact fp32 fma chain in k order).
//
// Mapping (F = 128):
//   * the by-target CSR edge list is cut into chunks of 32 edges (one 32-row MFMA block); a workgroup
//     (4 waves) owns `chunks_per_wg` consecutive chunks -- perfectly balanced, independent of node degree.
//     By default the grid is ONE resident wave of persistent workgroups (2 per CU, 243 VGPRs per lane), so the
//     weights are fetched once per workgroup and two workgroups per CU interleave: one's MFMAs run under the
//     other's softplus / gather / store phases;
//   * wave w owns filter columns [32w, 32w+32): its slices of W1 (G x 32) and W2 (128 x 32) live in
//     VGPRs for the whole kernel as MFMA B operands (lane l: B[k = 2kk + (l>>5)][col = l&31]), so
//     weights come from L2 once per workgroup and never touch LDS;
//   * the A operands (rbf tile, then the softplus'd hidden tile) are shared by the 4 waves through
//     LDS, row-major with ODD row strides (2*KK1+1, 129): the row-per-lane A reads and the
//     column-per-lane epilogue writes are bank-conflict free;
//   * gathered x1[src] values and the NEXT chunk's distances / edge ids are requested right after the chunk's
//     first barrier and consumed after its last MFMA; no load depends on another load; two barriers per chunk;
//   * the per-target segmented sum also runs on the matrix cores: agg = S . msg with the 0/1 selection
//     matrix S[t][e] = (dst_e == t); the message tile in accumulator layout is the B operand as it stands
//     (no LDS round trip).  Targets whose edges lie entirely inside the chunk are written with one plain
//     store; the first / last target of a chunk may continue in a neighbouring chunk and are combined
//     with atomicAdd into a zero-initialised output -- with degree <= 32 (radius graph cap, schnet.py:91) a
//     target spans at most two chunks, and two addends on top of zero commute exactly, so the result is
//     bitwise reproducible (a general CSR with higher degrees is still summed correctly, in atomic order);
//   * optionally the filter rows Wf[e] = (W2 h1 + b2) * C(d) are written out for the backward pass.
// History (MI355X, E = 49090, bs256 batch): 64-edge chunks, one workgroup per CU or spilling at two: 54 us;
// this layout 42-45 us.  tools/fused_phases.py (-DCF_TIMING=1 build) prints the per-phase cycle stamps.
#include "msde_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CF_F 128
#define CF_TE 32          // edges per chunk (one 32-row MFMA block)
#define CF_HS 129         // hidden tile row stride (floats)


__device__ __forceinline__ float ssp_fast(float x) {
  // softplus(x) - ln2 = max(x,0) + log(1 + exp(-|x|)) - ln2.  Hardware exp/log (v_exp_f32/v_log_f32):
  // absolute error ~1e-7, the same as the fp32 rounding of the reference's result; identical to torch's
  // threshold-20 branch for x > 20 (exp(-20) vanishes against 1).
  // raw v_exp_f32 / v_log_f32 (1 ulp): the fp32 MFMA leaves no issue shadow for vector instructions
  // (profiles/r02_probe_fp32_mfma_fillers.txt), so the range-checked __expf / __logf sequences are paid in full
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * fabsf(x));
  return fmaf(__builtin_amdgcn_logf(1.f + e), 0.69314718055994531f, fmaxf(x, 0.f) - 0.69314718055994531f);
}

__device__ __forceinline__ int cf_row(int s, int h) { return (s & 3) + 8 * (s >> 2) + 4 * h; }

#ifndef CF_TIMING
#define CF_TIMING 0          // 1: debug build -- per-phase s_memtime stamps of wave 0 go to Wf_out (tools/fused_phases.py)
#endif
#if CF_TIMING
#define CF_STAMP(k)                                                                         \
  do {                                                                                      \
    __builtin_amdgcn_s_waitcnt(0);                                                          \
    if (tid == 0 && stamp_n < 60) stamps[stamp_n++] = (long long)__builtin_readcyclecounter(); \
  } while (0)
#else
#define CF_STAMP(k)
#endif
#ifndef CF_OCC
#define CF_OCC 2             // waves per SIMD promised to the compiler
#endif

template <int KK1>
__global__ void __launch_bounds__(256, CF_OCC)
cfconv_fused_fwd_kernel(const float* __restrict__ x1, const float* __restrict__ dist, const int* __restrict__ rowptr,
                        const int* __restrict__ src, const int* __restrict__ dst, const float* __restrict__ W1,
                        const float* __restrict__ b1, const float* __restrict__ W2, const float* __restrict__ b2,
                        const float* __restrict__ offset, int N, int G, float coeff, float cutoff, int cpw,
                        float* __restrict__ agg, float* __restrict__ Wf_out) {
  constexpr int RS = 2 * KK1 + 1;           // rbf tile row stride (odd)
  extern __shared__ float lds[];
  float* rbf_t = lds;                          // [32][RS]      rbf tile (A of GEMM1)
  float* hid_t = rbf_t + CF_TE * RS;           // [32][129]     hidden tile (A of GEMM2)
  float* c_s = hid_t + CF_TE * CF_HS;          // [2][32] cutoff per edge row (0 for padding rows)
  int* src_s = reinterpret_cast<int*>(c_s + 2 * CF_TE);   // [2][32]
  int* dst_s = src_s + 2 * CF_TE;              // [2][32] target node, -1 for padding rows
  int* flag_s = dst_s + 2 * CF_TE;             // [2][4]: first target partial?, last target partial?, last target
  float* off_s = reinterpret_cast<float*>(flag_s + 8);    // [64] Gaussian centres

  const float PI_F = 3.14159265358979323846f;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lcol = lane & 31, lhalf = lane >> 5;
  const int col = wave * 32 + lcol;  // this lane's filter column

#if CF_TIMING
  long long* stamps = reinterpret_cast<long long*>(Wf_out) + (size_t)blockIdx.x * 64;
  int stamp_n = 0;
  Wf_out = nullptr;
  CF_STAMP(0);
  if (tid == 0) stamps[62] = (long long)wall_clock64();
#endif
  const int E = rowptr[N];
  const int e_begin = blockIdx.x * cpw * CF_TE;
  if (e_begin >= E) return;
  const int e_end = min(e_begin + cpw * CF_TE, E);

  // weights -> registers (B operands) from the nn.Linear layouts W1 [F][G], W2 [F][F] (row = output column), with
  // no transposed copy per step: the workgroup reads them with coalesced loads, stages 32 (W2) / 64 (W1) rows at
  // a time in the still unused tile region of LDS (odd row strides), and the owning waves pick their operands
  // lane (col, half) <- W[col][2kk + half].  All global loads are requested before the first staging round.
  float w1r[KK1], w2r[CF_F / 2];
  {
    float* stage = hid_t;                         // [32][129] floats (>= 64 x G for W1's two rounds)
    const bool vec = (reinterpret_cast<uintptr_t>(W2) & 15) == 0;
    float4 v2[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int i4 = tid + 256 * j;                   // float4 index inside the 32-row slab: row i4/32, column 4*(i4%32)
        const float* srcp = W2 + (size_t)(32 * r + (i4 >> 5)) * CF_F + 4 * (i4 & 31);
        v2[r][j] = vec ? *reinterpret_cast<const float4*>(srcp) : make_float4(srcp[0], srcp[1], srcp[2], srcp[3]);
      }
    constexpr int W1N = 64 * 64;                  // upper bound of one W1 half (64 rows x G <= 64)
    constexpr int W1J = W1N / 256;
    float v1[2][W1J];
    const int half_n = 64 * G;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int j = 0; j < W1J; ++j) {
        int i = tid + 256 * j;
        v1[r][j] = i < half_n ? W1[(size_t)r * half_n + i] : 0.f;
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int i4 = tid + 256 * j;
        float* d = stage + (i4 >> 5) * CF_HS + 4 * (i4 & 31);
        d[0] = v2[r][j].x; d[1] = v2[r][j].y; d[2] = v2[r][j].z; d[3] = v2[r][j].w;
      }
      __syncthreads();
      if (wave == r) {
#pragma unroll
        for (int kk = 0; kk < CF_F / 2; ++kk) w2r[kk] = stage[lcol * CF_HS + 2 * kk + lhalf];
      }
      __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
      for (int j = 0; j < W1J; ++j) {
        int i = tid + 256 * j;
        if (i < half_n) stage[i] = v1[r][j];      // rows of G floats, natural layout (G odd: conflict-free reads)
      }
      __syncthreads();
      if ((wave >> 1) == r) {
        const int lrow = (wave & 1) * 32 + lcol;
#pragma unroll
        for (int kk = 0; kk < KK1; ++kk) {
          int g = 2 * kk + lhalf;
          w1r[kk] = g < G ? stage[lrow * G + g] : 0.f;
        }
      }
      __syncthreads();
    }
  }
  const float b1c = b1[col], b2c = b2[col];

  // ---- one-chunk-ahead production of everything that depends only on the edge list.  The global loads are
  // REQUESTED right after the chunk's first barrier and CONSUMED (exp / cos) after its last MFMA, so their
  // latency lies under the GEMMs; no load depends on another load (the "does my first / last target continue
  // in the neighbouring chunk" flags come from dst[ec-1] and dst[ce], not from rowptr[dst[..]]).
  // smearing tile: lane = column g (lanes >= 2 KK1 repeat the last column: same value, same address), row = wave + 4 j:
  // the centre is a register, the distance a wave-uniform load, no index arithmetic
  constexpr int NRW = CF_TE / 4;
  float rv[NRW];                // distances, then rbf values
  float m_d = 0.f;
  int m_s = -1, m_t = -1, m_prev = -2, m_next = -2;
  const int rbf_g = min(lane, 2 * KK1 - 1);
  const float rbf_mu = rbf_g < G ? offset[rbf_g] : 0.f;
  const float coeff2 = coeff * 1.4426950408889634f;
  auto produce_load = [&](int ec) {
    const int ce = min(ec + CF_TE, e_end);
#pragma unroll
    for (int j = 0; j < NRW; ++j) {
      const int r = wave + 4 * j;
      rv[j] = ec + r < ce ? dist[ec + r] : -1.f;
    }
    if (tid < CF_TE) {
      int e = ec + tid;
      m_s = -1; m_t = -1; m_d = 0.f;
      if (e < ce) { m_d = dist[e]; m_s = src[e]; m_t = dst[e]; }
      if (tid == 0) {
        m_prev = ec > 0 ? dst[ec - 1] : -2;
        m_next = ce < E ? dst[ce] : -2;
      }
    }
  };
  auto produce_math = [&]() {
#pragma unroll
    for (int j = 0; j < NRW; ++j) {
      const float diff = rv[j] - rbf_mu;
      const float v = __builtin_amdgcn_exp2f(coeff2 * (diff * diff));
      rv[j] = (rv[j] >= 0.f && rbf_g < G) ? v : 0.f;      // padding rows and columns >= G are zero (A operand of GEMM1)
    }
  };
  produce_load(e_begin);
  produce_math();
  // per-lane LDS bases: every tile access below is base + compile-time offset (ds immediates) instead of one address
  // VGPR per access (the same change took 240 B of scratch and a third of the VALU work out of the backward kernel)
  constexpr auto RW = [](int i) constexpr { return (i & 3) + 8 * (i >> 2); };     // cf_row without the lane-half term
  float* const hw = hid_t + 4 * lhalf * CF_HS + col;          // hidden tile stores     + RW(i) * HS
  const float* const ra = rbf_t + lcol * RS + lhalf;          // A of GEMM1             + 2 kk
  const float* const ha = hid_t + lcol * CF_HS + lhalf;       // A of GEMM2             + 2 kk
  CF_STAMP(1);

  int buf = 0;
  for (int ec = e_begin; ec < e_end; ec += CF_TE, buf ^= 1) {
    const int ce = min(ec + CF_TE, e_end);
    // ---- P0: publish this chunk's rbf tile and metadata
#pragma unroll
    for (int j = 0; j < NRW; ++j) rbf_t[(wave + 4 * j) * RS + rbf_g] = rv[j];
    const int nxt = __shfl(m_next, 0);
    if (tid < CF_TE) {
      c_s[buf * CF_TE + tid] = m_t >= 0 ? 0.5f * (__cosf(m_d * (PI_F / cutoff)) + 1.0f) : 0.f;
      src_s[buf * CF_TE + tid] = max(m_s, 0) * (CF_F * 4);     // BYTE offset of the gathered x1 row
      dst_s[buf * CF_TE + tid] = m_t;
      if (tid == 0) flag_s[buf * 4] = (m_prev == m_t);
      if (tid == ce - ec - 1) { flag_s[buf * 4 + 1] = (nxt == m_t); flag_s[buf * 4 + 2] = m_t; }
    }
    __syncthreads();   // B1
    CF_STAMP(2);

    const float* cb = c_s + buf * CF_TE + 4 * lhalf;            // + RW(i): this lane half's rows
    const int* sb = src_s + buf * CF_TE + 4 * lhalf;
    const int* db = dst_s + buf * CF_TE;
    const int* dbh = db + 4 * lhalf;
    // gathered x1 rows for the epilogue: request now, consume after the GEMMs
    float xg[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      xg[i] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(x1) + (unsigned)(sb[RW(i)] + 4 * col));
    }
    const bool more = ec + CF_TE < e_end;
    if (more) produce_load(ec + CF_TE);

    // ---- GEMM1: [32 x 2KK1] . [2KK1 x 32] per wave
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int kk = 0; kk < KK1; ++kk)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[2 * kk], w1r[kk], acc, 0, 0, 0);
    CF_STAMP(3);
    // epilogue 1: bias + shifted softplus -> hidden tile
#pragma unroll
    for (int i = 0; i < 16; ++i) hw[RW(i) * CF_HS] = ssp_fast(acc[i] + b1c);
    __syncthreads();   // B2
    CF_STAMP(4);

    // ---- GEMM2: [32 x 128] . [128 x 32] per wave
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int kk = 0; kk < CF_F / 2; ++kk)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ha[2 * kk], w2r[kk], acc, 0, 0, 0);

    CF_STAMP(5);
    // epilogue 2 (registers only): filter = (acc + b2) * C ; message = x1[src] * filter
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      int row = cf_row(i, lhalf);
      float f0 = (acc[i] + b2c) * cb[RW(i)];        // padding rows: C = 0 -> message 0
      if (Wf_out && ec + row < ce) Wf_out[(size_t)(ec + row) * CF_F + col] = f0;
      acc[i] = xg[i] * f0;
    }

    CF_STAMP(6);
    // ---- segmented sum on the matrix cores: agg[t][f] += sum_e S[t][e] msg[e][f] with the 0/1 selection
    // matrix S[t][e] = (dst_e == t).  The message tile in accumulator layout is the B operand as it stands
    // (k-step s <-> register s, i.e. edge row cf_row(s, lane half)); A is built from the target ids.
    const int t0 = db[0];
    const int ntl = flag_s[buf * 4 + 2] - t0 + 1;    // local targets 0 .. ntl-1 (one tile unless the chunk
    const bool part0 = flag_s[buf * 4] != 0, part1 = flag_s[buf * 4 + 1] != 0;   // straddles isolated nodes)
    for (int tile = 0; tile * 32 < ntl; ++tile) {
      f32x16 d;
#pragma unroll
      for (int i = 0; i < 16; ++i) d[i] = 0.f;
      const int want = t0 + tile * 32 + lcol;
#pragma unroll
      for (int s2 = 0; s2 < 16; ++s2) {
        float sa = dbh[RW(s2)] == want ? 1.f : 0.f;
        d = __builtin_amdgcn_mfma_f32_32x32x2f32(sa, acc[s2], d, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        int tl = tile * 32 + cf_row(i, lhalf);
        if (tl < ntl) {
          bool partial = (tl == 0 && part0) || (tl == ntl - 1 && part1);
          float* o = &agg[(size_t)(t0 + tl) * CF_F + col];
          if (partial) atomicAdd(o, d[i]);           // <= 2 addends per target on top of zero: order-free
          else *o = d[i];
        }
      }
    }
    CF_STAMP(7);
    if (more) produce_math();
    CF_STAMP(8);
  }
#if CF_TIMING
  if (tid == 0) stamps[63] = (long long)wall_clock64();
#endif
}

extern "C" int msde_cfconv_fused_fwd(const float* x1, const float* dist, const int* rowptr, const int* src,
                                     const int* dst, const float* W1, const float* b1, const float* W2,
                                     const float* b2, const float* offset, int N, int F, int G, int E_cap,
                                     float coeff, float cutoff, int chunks_per_wg, float* agg, float* Wf_out,
                                     void* stream) {
  if (N < 0 || E_cap < 0 || !x1 || !dist || !rowptr || !src || !dst || !W1 || !b1 || !W2 || !b2 || !offset || !agg)
    return MSDE_EINVAL;
  if (F != CF_F || G <= 0 || G > 64) return MSDE_EUNSUP;
  if (N == 0) return 0;
  hipStream_t st = as_stream(stream);
  int ze = msde_zero_words(agg, (size_t)N * CF_F, st);   // atomics target + isolated nodes (a kernel, see msde_common.h)
  if (ze) return ze;
  if (E_cap == 0) return 0;
  int kk1 = (G + 1) / 2;
  int chunks = (E_cap + CF_TE - 1) / CF_TE;
  if (chunks_per_wg <= 0) {   // auto: one resident wave of workgroups (weights are loaded once per workgroup)
    int resident = msde_num_cus() * CF_OCC;
    chunks_per_wg = (chunks + resident - 1) / resident;
  }
  int grid = (chunks + chunks_per_wg - 1) / chunks_per_wg;
  auto lds_bytes = [](int KK1) {
    return (size_t)(CF_TE * (2 * KK1 + 1) + CF_TE * CF_HS + 6 * CF_TE + 8 + 64) * sizeof(float);
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
  MSDE_LAUNCH(cfconv_fused_fwd_kernel<KK>, dim3(grid), dim3(256), lds_bytes(KK), st, x1, dist, rowptr, src, dst, W1, \
              b1, W2, b2, offset, N, G, coeff, cutoff, chunks_per_wg, agg, Wf_out)
  if (kk1 == 26) { CF_LAUNCH(26); }
  else if (kk1 == 25) { CF_LAUNCH(25); }
  else { CF_LAUNCH(32); }
#undef CF_LAUNCH
  MSDE_CHECK_LAUNCH();
  return 0;
}
