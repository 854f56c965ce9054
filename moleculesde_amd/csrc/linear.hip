// linear.hip — fp32 Linear layer (forward, input gradient, weight/bias gradient) on the gfx950
// matrix cores.  Uses v_mfma_f32_32x32x2_f32: fp32 in, fp32 accumulate, bit-for-bit an fmaf chain in k
// order -- no reduced-precision shortcut, so results match an fp32 CPU GEMM to rounding.
//
// Why not the vendor GEMM: the shapes on the MoleculeSDE path are skinny (M = 3.6 k nodes or 35-49 k
// edges, N and K in 3..600).  One kernel template covers the three products of a Linear layer:
//     forward   Y[M,N]  = X[M,K]  . W[N,K]^T (+ bias)         A row-major,  B "n-major"  (NT)
//     grad-in   gX[M,K] = gY[M,N] . W[N,K]                    A row-major,  B k-major    (NN)
//     grad-w    gW[N,K] = gY[M,N]^T . X[M,K]  (+ column sums) A k-major,    B k-major    (TN, split over M)
// Tiling: 256 threads = 4 waves as 2 x 2; each wave owns TM x TN MFMA tiles of 32 x 32; block tile
// (64 TM) x (64 TN); K step 32.  LDS images are k-major with row stride BM+1 / BN+1 (== 1 mod 32):
// the MFMA operand reads (32 consecutive rows per half-wave at one k) and the transposing staging
// writes are both bank-conflict free.  Next tile's global loads are issued before the current tile's
// MFMAs (register prefetch).  The split-M weight gradient writes per-split slabs that a second
// kernel sums in a fixed order: bitwise reproducible, no float atomics.
// Round 6: inside the grouped weight-gradient launch a layer with a 32-wide (or narrower) side runs the same tile body with
// another wave layout (template parameters WGM / KS: 32 x 128, 128 x 32, or 32 x 32 with the K tile's rows split over the
// waves) instead of leaving two or three of the four waves idle -- see gemm_f32_mfma_body and msde_linear_bwd_w_describe_ld.
#include "msde_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define LG_BK 32
#ifndef MSDE_WGRAD_NARROW
#define MSDE_WGRAD_NARROW 7      // bit mask of the narrow tile shapes in use (1: 32 x 128, 2: 128 x 32, 4: 32 x 32 with split K tiles); 0: 64 x 64 only
#endif
#ifndef LG_SKIP_DEAD
#define LG_SKIP_DEAD 1
#endif
#ifndef LG_PRIO
#define LG_PRIO 2
#endif
#ifndef LG_ST
#define LG_ST 1
#endif

// Operand loads go through GLOBAL-address-space pointers.  The grouped kernel reads its operand pointers from a table
// in memory, so to the compiler they are generic pointers and plain dereferences become FLAT loads -- which count on
// the LDS counter (lgkmcnt) as well as vmcnt: every wait for an LDS operand read then also waits for the global
// prefetch just issued, and the software pipeline collapses (measured: MFMA time + everything-else time, no overlap).
__device__ __forceinline__ float4 lg_ldg4(const void* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef float lg_v4f __attribute__((ext_vector_type(4)));
  const lg_v4f v = *(const __attribute__((address_space(1))) lg_v4f*)(p);
  return make_float4(v.x, v.y, v.z, v.w);
#else
  return *reinterpret_cast<const float4*>(p);
#endif
}
__device__ __forceinline__ float lg_ldg1(const float* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  return *(const __attribute__((address_space(1))) float*)(p);
#else
  return *p;
#endif
}

// -DLG_PROBE (tools only, tools/bench_grouped_wgrad.py probe): s_memtime stamps around the phases of the interior K loop, summed
// over the loop by wave 0 of every workgroup and added into lg_probe[] (read back with msde_debug_lg_probe)
#ifdef LG_PROBE
__device__ unsigned long long lg_probe[8];
#define LG_T(x) __builtin_amdgcn_sched_barrier(0); const unsigned long long x = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0)
extern "C" int msde_debug_lg_probe(unsigned long long* host8, int reset) {
  if (host8 && hipMemcpyFromSymbol(host8, HIP_SYMBOL(lg_probe), sizeof(lg_probe)) != hipSuccess) return MSDE_EINVAL;
  if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(lg_probe), z, sizeof(z)) != hipSuccess) return MSDE_EINVAL; }
  return 0;
}
#else
#define LG_T(x)
#endif

// WGM: waves along the M side of the tile (2: the 2 x 2 wave grid of the 64 TM x 64 TN tile; 1 / 4 with TM = TN = 1: the four waves
// side by side, a 32 x 128 / 128 x 32 tile).  KS = 4 (TM = TN = 1, WGM = 1): a 32 x 32 tile whose four waves all work on the SAME
// outputs, each on its quarter (32 rows) of a 128-row K tile; the four partial accumulators meet in LDS (wave order: fixed).
// These are the weight-gradient shapes of products with a 32-wide (or narrower) side, where the 64 x 64 tile would leave half
// (two narrow sides: three quarters) of its waves idle beside operand columns nobody reads (msde_linear_bwd_w_describe_ld).
template <int TM, int TN, bool A_KM, bool B_KM, bool VEC, int WGM = 2, int KS = 1>
__device__ __forceinline__ void
gemm_f32_mfma_body(const float* __restrict__ A, const float* __restrict__ B, const float* __restrict__ bias,
                   float* __restrict__ C, float* __restrict__ colsum_ws, int M, int N, int K, int lda, int ldb,
                   int ldc, int k_per_split, const int bid_x, const int bid_y, const int bid_z, float* __restrict__ As,
                   float* __restrict__ Bs, const bool edge_fast = false) {
  constexpr int WGN = KS == 4 ? 1 : 4 / WGM;
  constexpr int BM = 32 * (KS == 4 ? 1 : WGM) * TM, BN = 32 * WGN * TN;
  constexpr int BK = LG_BK * KS;            // rows of a K tile
  static_assert(KS == 1 || (KS == 4 && TM == 1 && TN == 1 && WGM == 1 && A_KM && B_KM), "k-split tiles: 32 x 32, k-major operands");
  static_assert(WGM == 2 || (TM == 1 && TN == 1 && A_KM && B_KM), "side-by-side wave layouts: weight-gradient operands only");
  // k-major LDS images.  An operand that is k-major in memory is copied with aligned 16-B stores (row
  // stride BM+4); one that is row-major is transposed on the way in with scalar stores, for which the
  // stride BM+1 (== 1 mod 32) is the conflict-free one.  The MFMA operand reads (32 consecutive floats
  // of one k row per half-wave) are conflict free for any stride.
  // k-major images of 64-wide tiles use the row stride 64 exactly: rows are whole multiples of 64 dwords apart, so the
  // operand reads of a K tile become `ds_read2st64_b32 base offset0:k offset1:k+2` off ONE per-lane base register
  // (with stride 68 every read pair needed a v_add for its address -- 16 vector instructions per K tile that the fp32
  // MFMAs cannot hide, 4.17).  Conflict-free all the same: a half-wave reads 32 consecutive floats of one row, and a
  // 16-byte store instruction is served 8 lanes (= 32 consecutive floats) at a time.
  // (As / Bs: the caller's LDS, LG_BK * LDA_S and LG_BK * LDB_S floats -- the grouped kernel shares one buffer among its tile shapes)
  constexpr int LDA_S = A_KM ? (KS == 4 ? 32 : BM <= 64 ? 64 : (WGM == 2 ? BM + 4 : BM)) : BM + 1;
  constexpr int LDB_S = B_KM ? (KS == 4 ? 32 : BN <= 64 ? 64 : (WGM == 2 ? BN + 4 : BN)) : BN + 1;

  int tid_ = threadIdx.x;
  // (opaque to the optimiser: inside the grouped kernel's tile loop the per-thread constants of EVERY tile shape -- staging offsets,
  // LDS addresses -- would otherwise be hoisted in front of the loop and stay live together: 240 VGPRs instead of the widest shape's)
  asm volatile("" : "+v"(tid_));
  const int tid = tid_, lane = tid & 63, wave = tid >> 6;
  const int lcol = lane & 31, lhalf = lane >> 5;
  const int wm = KS == 4 ? 0 : wave / WGN, wn = KS == 4 ? 0 : wave % WGN;
  const int kq = KS == 4 ? wave * LG_BK : 0;        // first row of this wave's share of the K tile
  const int m0 = bid_y * BM, n0 = bid_x * BN;
  const int kb = bid_z * k_per_split;
  const int ke = max(min(K, kb + k_per_split), kb);      // K may be a device-side row bound below this split: no tiles

  // staging registers: each thread moves (BM*BK/4)/256 float4 of A and (BN*BK/4)/256 of B per tile (2 TM and 2 TN on the 2 x 2 grid)
  constexpr int NA = BM * BK / 1024, NB = BN * BK / 1024;
  static_assert(NA >= 1 && NB >= 1, "tile too small for 256 threads");
  constexpr int ST = LG_ST;  // register prefetch depth: loads are issued ST tiles ahead of their MFMAs
  float4 rsa[ST][NA], rsb[ST][NB];

  // Branch-free tile loads: out-of-range rows / k are redirected to a valid address and zeroed by a
  // select, so the compiler emits back-to-back loads with ONE counted wait (a per-element branch makes
  // hipcc wait vmcnt(0) after every load: 6 serialized L2 round trips per tile).
  auto ld4 = [&](const float* __restrict__ base, int row, int nrows, int ld, int col, int ncols, bool vec) -> float4 {
    bool rv = row < nrows;
    int rc = rv ? row : 0;
    const float* src = base + (size_t)rc * ld;
    float4 v;
    if (vec) {                                   // ncols % 4 == 0: a float4 is all-in or all-out
      bool cv = col < ncols;
      v = lg_ldg4(src + (cv ? col : 0));
      if (!(rv && cv)) v = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      float x0 = lg_ldg1(src + (col < ncols ? col : 0)), x1 = lg_ldg1(src + (col + 1 < ncols ? col + 1 : 0));
      float x2 = lg_ldg1(src + (col + 2 < ncols ? col + 2 : 0)), x3 = lg_ldg1(src + (col + 3 < ncols ? col + 3 : 0));
      v.x = (rv && col < ncols) ? x0 : 0.f;
      v.y = (rv && col + 1 < ncols) ? x1 : 0.f;
      v.z = (rv && col + 2 < ncols) ? x2 : 0.f;
      v.w = (rv && col + 3 < ncols) ? x3 : 0.f;
    }
    return v;
  };

  auto load_tile = [&](float4 (&ra)[NA], float4 (&rb)[NB], int k0) {
#pragma unroll
    for (int p = 0; p < NA; ++p) {
      int idx = p * 256 + tid;
      if (!A_KM) {                       // A[m][k], k contiguous: 8 float4 per row of the tile
        int r = idx >> 3, kq = (idx & 7) * 4;
        ra[p] = ld4(A, m0 + r, M, lda, k0 + kq, ke, VEC);
      } else {                           // A[k][m], m contiguous: BM/4 float4 per k-row
        int kr = idx / (BM / 4), mq = (idx % (BM / 4)) * 4;
        ra[p] = ld4(A, k0 + kr, ke, lda, m0 + mq, M, VEC);
      }
    }
#pragma unroll
    for (int p = 0; p < NB; ++p) {
      int idx = p * 256 + tid;
      if (!B_KM) {                       // B[n][k], k contiguous
        int r = idx >> 3, kq = (idx & 7) * 4;
        rb[p] = ld4(B, n0 + r, N, ldb, k0 + kq, ke, VEC);
      } else {                           // B[k][n], n contiguous
        int kr = idx / (BN / 4), nq = (idx % (BN / 4)) * 4;
        rb[p] = ld4(B, k0 + kr, ke, ldb, n0 + nq, N, VEC);
      }
    }
  };

  // Interior tiles of the k-major x k-major product (the weight-gradient shape: both operands are row blocks of
  // activations): no row / column clamps, no zeroing selects, and every thread's address is a CONSTANT 32-bit byte
  // offset from a wave-uniform tile base (scalar arithmetic, `saddr + voffset` loads).  The general path costs ~12
  // vector instructions per float4, and on this chip vector instructions take issue slots from the fp32 MFMAs
  // (DESIGN 4.17): ~50 per K tile against 16 MFMAs.
  // EDGE tiles of the output (m0 + BM > M or n0 + BN > N) take the same path: a float4 whose columns lie beyond the operand's
  // width (a whole float4: widths are multiples of 4 here) is read from the operand's LAST float4 instead -- valid memory,
  // wrong values, and they only ever reach accumulator rows / columns >= M / N, which the epilogue does not store (and the
  // column sums, which it does not store either).  Rows of the reduction are what must be exact, and full K tiles are.
  constexpr bool FAST = A_KM && B_KM && VEC;
  const bool interior = FAST && (edge_fast ? (M >= 4 && N >= 4) : (m0 + BM <= M && n0 + BN <= N));
  unsigned oa[NA], ob[NB];
  if (FAST) {
#pragma unroll
    for (int p = 0; p < NA; ++p) {
      const int idx = p * 256 + tid, kr = idx / (BM / 4), mq = min((idx % (BM / 4)) * 4, M - 4 - m0);
      oa[p] = (unsigned)(((long long)kr * lda + mq) * 4);
    }
#pragma unroll
    for (int p = 0; p < NB; ++p) {
      const int idx = p * 256 + tid, kr = idx / (BN / 4), nq = min((idx % (BN / 4)) * 4, N - 4 - n0);
      ob[p] = (unsigned)(((long long)kr * ldb + nq) * 4);
    }
  }
  auto load_tile_fast = [&](float4 (&ra)[NA], float4 (&rb)[NB], int k0) {
    const char* __restrict__ ab = reinterpret_cast<const char*>(A + (size_t)k0 * lda + m0);
    const char* __restrict__ bb = reinterpret_cast<const char*>(B + (size_t)k0 * ldb + n0);
#pragma unroll
    for (int p = 0; p < NA; ++p) ra[p] = lg_ldg4(ab + oa[p]);
#pragma unroll
    for (int p = 0; p < NB; ++p) rb[p] = lg_ldg4(bb + ob[p]);
  };

  auto store_tile = [&](const float4 (&ra)[NA], const float4 (&rb)[NB]) {
#pragma unroll
    for (int p = 0; p < NA; ++p) {
      int idx = p * 256 + tid;
      if (!A_KM) {
        int r = idx >> 3, kq = (idx & 7) * 4;
        As[(kq + 0) * LDA_S + r] = ra[p].x;
        As[(kq + 1) * LDA_S + r] = ra[p].y;
        As[(kq + 2) * LDA_S + r] = ra[p].z;
        As[(kq + 3) * LDA_S + r] = ra[p].w;
      } else {
        int kr = idx / (BM / 4), mq = (idx % (BM / 4)) * 4;
        *reinterpret_cast<float4*>(&As[kr * LDA_S + mq]) = ra[p];
      }
    }
#pragma unroll
    for (int p = 0; p < NB; ++p) {
      int idx = p * 256 + tid;
      if (!B_KM) {
        int r = idx >> 3, kq = (idx & 7) * 4;
        Bs[(kq + 0) * LDB_S + r] = rb[p].x;
        Bs[(kq + 1) * LDB_S + r] = rb[p].y;
        Bs[(kq + 2) * LDB_S + r] = rb[p].z;
        Bs[(kq + 3) * LDB_S + r] = rb[p].w;
      } else {
        int kr = idx / (BN / 4), nq = (idx % (BN / 4)) * 4;
        *reinterpret_cast<float4*>(&Bs[kr * LDB_S + nq]) = rb[p];
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float csum = 0.f;  // column sum of A over k (weight-gradient mode: bias gradient), thread tid < BM

  const bool quadrant_live = !LG_SKIP_DEAD || ((m0 + wm * 32 * TM < M) && (n0 + wn * 32 * TN < N));
  bool sub_live[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
      sub_live[i][j] = (m0 + (wm * TM + i) * 32 < M) && (n0 + (wn * TN + j) * 32 < N);
#ifdef LG_PROBE
  unsigned long long pt4 = 0, pacc[6] = {0, 0, 0, 0, 0, 0};
#endif
  auto compute_tile = [&]() {
    if (TM == 1 && TN == 1) {
      // a wave whose 32 x 32 quadrant lies entirely outside the product (narrow layers: N or K <= 32 fill one or two
      // quadrants of the 64 x 64 tile) only helps with the staging: its SIMD is free for other workgroups' MFMAs
      if (quadrant_live) {
      // all operand reads of the K tile are requested first, then the 16 MFMAs retire them in order behind counted
      // waits.  Left to itself the compiler emits {2 reads, wait for both, 2 MFMAs} x 8: one LDS round trip exposed
      // per 128 cycles of MFMA work (the scheduling barrier keeps it from sinking the reads back to their uses).
      float af[LG_BK / 2], bf[LG_BK / 2];
#pragma unroll
      for (int kk = 0; kk < LG_BK / 2; ++kk) {
        af[kk] = As[(kq + 2 * kk + lhalf) * LDA_S + wm * 32 + lcol];
        bf[kk] = Bs[(kq + 2 * kk + lhalf) * LDB_S + wn * 32 + lcol];
      }
      __builtin_amdgcn_sched_barrier(0);
#ifdef LG_PROBE
      pt4 = __builtin_readcyclecounter();
      __builtin_amdgcn_sched_barrier(0);
#endif
      __builtin_amdgcn_s_setprio(LG_PRIO);          // the wave in its MFMA phase wins issue slots from waves staging tiles
#pragma unroll
      for (int kk = 0; kk < LG_BK / 2; ++kk)
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk], bf[kk], acc[0][0], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      }
    } else {
    // (32 x 32 sub-tiles that lie entirely outside the product are skipped: a 300-wide dimension covers 10 of them, not 12)
    __builtin_amdgcn_s_setprio(LG_PRIO);
#pragma unroll
    for (int kk = 0; kk < LG_BK / 2; ++kk) {
      float af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = As[(2 * kk + lhalf) * LDA_S + (wm * TM + i) * 32 + lcol];
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = Bs[(2 * kk + lhalf) * LDB_S + (wn * TN + j) * 32 + lcol];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          if (!LG_SKIP_DEAD || sub_live[i][j])
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    }
    if (colsum_ws != nullptr && bid_x == 0) {
      if (A_KM && (BM == 64 || WGM != 2)) {
        // all four waves share the column sums (BM = 64: wave q takes k rows 8q .. 8q+7 of column tid & 63): 8 additions per
        // thread and K tile instead of 32 on wave 0 alone, whose MFMAs they would hold up (4.17); in general 256 / BM groups of
        // threads, group q on rows q BK / groups .. of column tid % BM
        constexpr int CG = 256 / BM;
        const int c = tid % BM, q = tid / BM;
#pragma unroll
        for (int kr = 0; kr < BK / CG; ++kr) csum += As[(q * (BK / CG) + kr) * LDA_S + c];
      } else if (tid < BM) {
#pragma unroll 8
        for (int kr = 0; kr < LG_BK; ++kr) csum += As[kr * LDA_S + tid];
      }
    }
  };

  // Software pipeline: the loads of tile t+ST are issued while tile t is multiplied.  Every load is
  // unconditional (tiles past the end are masked to zero by ld4), so the steady-state loop is branch
  // free and the compiler can retire each stage with a counted vmcnt instead of vmcnt(0).
  const int ntiles = (ke - kb + BK - 1) / BK;
  if (ntiles > 0) {
#pragma unroll
    for (int s = 0; s < ST; ++s) load_tile(rsa[s], rsb[s], kb + s * BK);
    int t = 0;
    if (FAST && interior) {
      // steady state of an interior tile: every tile requested here (t + s + ST) lies entirely inside [kb, ke), so the
      // loop body is straight-line code with clamp-free loads -- with a second (general) path inside the loop the
      // compiler's wait counters go to vmcnt(0) at the LDS stores, i.e. they wait for the loads just issued
      const int nfull = (ke - kb) / BK;
      for (; t + 2 * ST <= nfull; t += ST) {
#pragma unroll
        for (int s = 0; s < ST; ++s) {
          LG_T(t0);
          __syncthreads();
          LG_T(t1);
          store_tile(rsa[s], rsb[s]);
          LG_T(t2);
          __syncthreads();
          LG_T(t3);
          load_tile_fast(rsa[s], rsb[s], kb + (t + s + ST) * BK);
          compute_tile();
          LG_T(t5);
#ifdef LG_PROBE
          pacc[0] += t1 - t0; pacc[1] += t2 - t1; pacc[2] += t3 - t2; pacc[3] += pt4 - t3; pacc[4] += t5 - pt4; pacc[5] += 1;
#endif
        }
      }
    }
    for (; t + ST <= ntiles; t += ST) {
#pragma unroll
      for (int s = 0; s < ST; ++s) {          // static stage index: the staging arrays stay in VGPRs
        __syncthreads();                      // previous tile fully consumed
        store_tile(rsa[s], rsb[s]);
        __syncthreads();
        load_tile(rsa[s], rsb[s], kb + (t + s + ST) * BK);
        compute_tile();
      }
    }
    const int rem = ntiles - t;               // 0 .. ST-1 tiles left, already in flight
#pragma unroll
    for (int s = 0; s < ST - 1; ++s) {
      if (s < rem) {
        __syncthreads();
        store_tile(rsa[s], rsb[s]);
        __syncthreads();
        compute_tile();
      }
    }
  }

#ifdef LG_PROBE
  if (tid == 0 && quadrant_live)
    for (int i = 0; i < 6; ++i) atomicAdd(&lg_probe[i], pacc[i]);
#endif
  // epilogue.  C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  float* Cz = C + (size_t)bid_z * (size_t)M * ldc;  // split slabs (bid_z == 0 when unsplit)
  if (KS == 4) {
    // four partial accumulators of the same 32 x 32 outputs: red[wave][r][lane] in the A tile buffer (16 KB), then thread
    // (lane, wave q) sums r = 4 q .. 4 q + 3 over the waves in wave order
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) As[(wave * 16 + r) * 64 + lane] = acc[0][0][r];
    __syncthreads();
    const int gn = n0 + lcol;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = 4 * wave + j;
      const float v = ((As[r * 64 + lane] + As[(16 + r) * 64 + lane]) + As[(32 + r) * 64 + lane]) + As[(48 + r) * 64 + lane];
      const int gm = m0 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
      if (gm < M && gn < N) Cz[(size_t)gm * ldc + gn] = v;
    }
  } else {
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      int gn = n0 + (wn * TN + j) * 32 + lcol;
      float bv = (bias != nullptr && gn < N) ? bias[gn] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int gm = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
        if (gm < M && gn < N) Cz[(size_t)gm * ldc + gn] = acc[i][j][r] + bv;
      }
    }
  }
  if (colsum_ws != nullptr && bid_x == 0) {
    if (A_KM && (BM == 64 || WGM != 2)) {   // the row-group partials of a column (four quarters when BM = 64), added in group order
      constexpr int CG = 256 / BM;
      __syncthreads();
      Bs[tid] = csum;
      __syncthreads();
      if (tid < BM) {
        csum = Bs[tid];
#pragma unroll
        for (int q = 1; q < CG; ++q) csum += Bs[q * BM + tid];
      }
    }
    if (tid < BM && m0 + tid < M) colsum_ws[(size_t)bid_z * M + m0 + tid] = csum;
  }
}

template <int TM, int TN, bool A_KM, bool B_KM, bool VEC>
__global__ void __launch_bounds__(256)
gemm_f32_mfma_kernel(const float* __restrict__ A, const float* __restrict__ B, const float* __restrict__ bias,
                     float* __restrict__ C, float* __restrict__ colsum_ws, int M, int N, int K, int lda, int ldb,
                     int ldc, int k_per_split, const int* __restrict__ Kdev) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  constexpr int LDA_S = A_KM ? (BM == 64 ? 64 : BM + 4) : BM + 1, LDB_S = B_KM ? (BN == 64 ? 64 : BN + 4) : BN + 1;
  __shared__ __attribute__((aligned(16))) float As[LG_BK * LDA_S];
  __shared__ __attribute__((aligned(16))) float Bs[LG_BK * LDB_S];
  gemm_f32_mfma_body<TM, TN, A_KM, B_KM, VEC>(A, B, bias, C, colsum_ws, M, N, msde_true_rows(K, Kdev), lda, ldb, ldc,
                                              k_per_split, blockIdx.x, blockIdx.y, blockIdx.z, As, Bs);
}

// Grouped weight gradients: ONE launch runs the split-M GEMMs of many layers.  probs[p] = 16 int64:
// {gY, X, slabs, colsum partials (0: no bias), M, N, K, splits, k_per_split, tiles_x, tiles_y, vec, ldg, ldx, 0, 0}
// (ldg / ldx: row strides of gY / X -- operands may be column blocks of wider buffers);
// prefix[p] = workgroups before problem p.  A workgroup finds its problem by binary search and then runs the very
// same tile body as the per-layer kernel (64 x 64 tiles), so results are bit-identical to it.
__device__ __forceinline__ void
gemm_grouped_wgrad_body(const long long* __restrict__ probs, const int* __restrict__ prefix, int count, int total,
                          int xcd_order) {
  // one LDS buffer for every tile shape: As = lds (<= 4096 floats), Bs = lds + 4096 (<= 4096 floats)
  __shared__ __attribute__((aligned(16))) float lds[8192];
  // grid == total: one tile per workgroup.  grid < total (msde_linear_bwd_w_grouped_ex with a width limit): each
  // workgroup walks tiles blockIdx.x, + gridDim.x, ...: the launch then occupies at most gridDim.x workgroup slots, so it
  // can run BESIDE a latency-critical chain on another stream without taking every CU (same results, tile by tile).
  for (int blk = blockIdx.x; blk < total; blk += gridDim.x) {
    int lo = 0, hi = count;
    while (hi - lo > 1) {
      int mid = (lo + hi) >> 1;
      if (prefix[mid] <= blk) lo = mid; else hi = mid;
    }
    const long long* e = probs + (size_t)lo * MSDE_WGRAD_ROW;
    const float* gY = reinterpret_cast<const float*>(e[0]);
    const float* X = reinterpret_cast<const float*>(e[1]);
    float* slabs = reinterpret_cast<float*>(e[2]);
    float* cs = reinterpret_cast<float*>(e[3]);
    const int M = (int)e[4], N = (int)e[5], K = (int)e[6], kps = (int)e[8], tx = (int)e[9], ty = (int)e[10];
    const int ldg = (int)e[12], ldx = (int)e[13];
    const int Mt = msde_true_rows(M, reinterpret_cast<const int*>(e[14]));      // valid rows of gY / X (row bound)
    // XCD-aware tile order inside a problem: consecutive workgroups run on different XCDs (b % 8), so in natural order
    // every XCD streams every operand block of every layer from the fabric.  The problem's n tiles (x fastest, then y,
    // then split) are cut into 8 contiguous runs and residue class c = local % 8 (one XCD) takes run c: the tiles an
    // XCD works on share gY / X row blocks in ITS L2.  Every problem is still spread evenly over the 8 XCDs.
    int local = blk - prefix[lo];
    if (xcd_order) {
      const int np = prefix[lo + 1] - prefix[lo], q = np >> 3, r = np & 7, c = local & 7;
      local = c * q + min(c, r) + (local >> 3);
    }
    const int bx = local % tx, by = (local / tx) % ty, bz = local / (tx * ty);
    // product C[N][K] = gY^T X: "M" of the product = N, "N" = K, reduction = M (see msde_linear_bwd_w)
    // e[15]: bit 0 = edge tiles on the fast path, bits 1.. = tile shape (msde_linear_bwd_w_describe_ld)
    const int shape = (int)(e[15] >> 1);
    float* As = lds;
    float* Bs = lds + 4096;
    if ((MSDE_WGRAD_NARROW & 1) && e[11] && shape == 1)          // 32 x 128: the four waves side by side along K
      gemm_f32_mfma_body<1, 1, true, true, true, 1, 1>(gY, X, nullptr, slabs, cs, N, K, Mt, ldg, ldx, K, kps, bx, by, bz, As, Bs, true);
    else if ((MSDE_WGRAD_NARROW & 2) && e[11] && shape == 2)     // 128 x 32
      gemm_f32_mfma_body<1, 1, true, true, true, 4, 1>(gY, X, nullptr, slabs, cs, N, K, Mt, ldg, ldx, K, kps, bx, by, bz, As, Bs, true);
    else if ((MSDE_WGRAD_NARROW & 4) && e[11] && shape == 3)     // 32 x 32, the 128 rows of a K tile split over the waves
      gemm_f32_mfma_body<1, 1, true, true, true, 1, 4>(gY, X, nullptr, slabs, cs, N, K, Mt, ldg, ldx, K, kps, bx, by, bz, As, Bs, true);
    else if (e[11])
      gemm_f32_mfma_body<1, 1, true, true, true>(gY, X, nullptr, slabs, cs, N, K, Mt, ldg, ldx, K, kps, bx, by, bz, As, Bs,
                                                 (e[15] & 1) != 0);
    else
      gemm_f32_mfma_body<1, 1, true, true, false>(gY, X, nullptr, slabs, cs, N, K, Mt, ldg, ldx, K, kps, bx, by, bz, As, Bs);
    __syncthreads();                 // the next tile reuses the LDS stages
  }
}

__global__ void __launch_bounds__(256)
gemm_grouped_wgrad_kernel(const long long* __restrict__ probs, const int* __restrict__ prefix, int count, int total,
                          int xcd_order) {
  gemm_grouped_wgrad_body(probs, prefix, count, total, xcd_order);
}
// the same with the register budget of FOUR workgroups per CU (128 VGPRs; the default build takes 140: three)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
gemm_grouped_wgrad_occ4_kernel(const long long* __restrict__ probs, const int* __restrict__ prefix, int count, int total,
                               int xcd_order) {
  gemm_grouped_wgrad_body(probs, prefix, count, total, xcd_order);
}

// out[i] = sum_z slabs[z][i] for the weight slabs (n entries) and, in the same launch, the bias-gradient
// column sums (nb entries; cs == nullptr when the layer has no bias).  16 split lanes per output: lane l
// sums the splits z = l, l+16, ... and the 16 partials are added in lane order -> fixed summation order.
#define RS_LANES 16
__global__ void __launch_bounds__(256)
reduce_slabs_kernel(const float* __restrict__ slabs, int splits, size_t n, float* __restrict__ out,
                    const float* __restrict__ cs, size_t nb, float* __restrict__ outb) {
  __shared__ float part[RS_LANES][16];
  const int ox = threadIdx.x & 15, ly = threadIdx.x >> 4;
  size_t total = n + (cs ? nb : 0);
  for (size_t base = (size_t)blockIdx.x * 16; base < total; base += (size_t)gridDim.x * 16) {
    size_t i = base + ox;
    float acc = 0.f;
    if (i < total) {
      const float* src = i < n ? slabs + i : cs + (i - n);
      size_t stride = i < n ? n : nb;
      for (int z = ly; z < splits; z += RS_LANES) acc += src[(size_t)z * stride];
    }
    part[ly][ox] = acc;
    __syncthreads();
    if (ly == 0 && i < total) {
      float r = part[0][ox];
#pragma unroll
      for (int l = 1; l < RS_LANES; ++l) r += part[l][ox];
      if (i < n) out[i] = r; else outb[i - n] = r;
    }
    __syncthreads();
  }
}

// out[i] = sum_z slabs[z*n + i] (z in index order, 16 interleaved lanes) and, when cs != nullptr, the same
// for a second family outb[j] = sum_z cs[z*nb + j].  Shared by the other translation units (msde_common.h).
int msde_reduce_slabs(const float* slabs, int splits, size_t n, float* out, const float* cs, size_t nb, float* outb,
                      hipStream_t st) {
  size_t total = n + (cs ? nb : 0);
  int blocks = (int)((total + 15) / 16);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  MSDE_LAUNCH(reduce_slabs_kernel, dim3(blocks), dim3(256), 0, st, slabs, splits, n, out, cs, nb, outb);
  MSDE_CHECK_LAUNCH();
  return 0;
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int TM, int TN, bool A_KM, bool B_KM>
static void launch_cfg(bool vec, dim3 grid, hipStream_t st, const float* A, const float* B, const float* bias, float* C,
                       float* colsum_ws, int M, int N, int K, int lda, int ldb, int ldc, int k_per_split, const int* kdev) {
  // (kdev: weight gradients reduce over the rows of both operands and honour a row bound on that extent)
  if (vec)
    MSDE_LAUNCH((gemm_f32_mfma_kernel<TM, TN, A_KM, B_KM, true>), grid, dim3(256), 0, st, A, B, bias, C, colsum_ws, M, N,
                K, lda, ldb, ldc, k_per_split, kdev);
  else
    MSDE_LAUNCH((gemm_f32_mfma_kernel<TM, TN, A_KM, B_KM, false>), grid, dim3(256), 0, st, A, B, bias, C, colsum_ws, M,
                N, K, lda, ldb, ldc, k_per_split, kdev);
}

// tile: rows (product M) and columns (product N) each 64 or 128 wide; `big` prefers 128-wide tiles
// whenever the dimension exceeds 64 (weight gradient: parallelism comes from the M split instead).
template <bool A_KM, bool B_KM>
static int launch_gemm(const float* A, const float* B, const float* bias, float* C, float* colsum_ws, int M, int N,
                       int K, int lda, int ldb, int ldc, int splits, int k_per_split, bool big, hipStream_t st,
                       const int* kdev = nullptr) {
  // vector path: 16-B aligned bases, leading dimensions % 4, and the contiguous extent of each operand % 4
  bool vec = aligned16(A) && aligned16(B) && (lda % 4 == 0) && (ldb % 4 == 0) && (k_per_split % 4 == 0);
  vec = vec && ((A_KM ? M : K) % 4 == 0) && ((B_KM ? N : K) % 4 == 0);
  int tm, tn;
  if (big) {
    tm = M > 64 ? 2 : 1;
    tn = N > 64 ? 2 : 1;
  } else {
    long t22 = (long)((M + 127) / 128) * ((N + 127) / 128);
    long t21 = (long)((M + 127) / 128) * ((N + 63) / 64);
    if (t22 >= 384) { tm = 2; tn = 2; }
    else if (t21 >= 256) { tm = 2; tn = 1; }
    else { tm = 1; tn = 1; }
  }
  dim3 grid((N + 64 * tn - 1) / (64 * tn), (M + 64 * tm - 1) / (64 * tm), splits);
#define LG_ARGS vec, grid, st, A, B, bias, C, colsum_ws, M, N, K, lda, ldb, ldc, k_per_split, kdev
  if (tm == 2 && tn == 2) launch_cfg<2, 2, A_KM, B_KM>(LG_ARGS);
  else if (tm == 2 && tn == 1) launch_cfg<2, 1, A_KM, B_KM>(LG_ARGS);
  else if (tm == 1 && tn == 2) launch_cfg<1, 2, A_KM, B_KM>(LG_ARGS);
  else launch_cfg<1, 1, A_KM, B_KM>(LG_ARGS);
#undef LG_ARGS
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

// ---- zero fill as a kernel (msde_common.h: the library never records memset nodes) --------------------------------
__global__ void __launch_bounds__(256) zero_words_kernel(uint32_t* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0u;
}
__global__ void __launch_bounds__(256) zero_words4_kernel(uint4* __restrict__ p, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p[i] = make_uint4(0u, 0u, 0u, 0u);
}

int msde_zero_words(void* p, size_t n_words, hipStream_t st) {
  if (n_words == 0) return 0;
  if (!p || (reinterpret_cast<uintptr_t>(p) & 3)) return MSDE_EINVAL;
  if ((reinterpret_cast<uintptr_t>(p) & 15) == 0 && n_words % 4 == 0) {
    size_t n4 = n_words / 4, wg = (n4 + 255) / 256;
    MSDE_LAUNCH(zero_words4_kernel, dim3((unsigned)(wg < 4096 ? wg : 4096)), dim3(256), 0, st, reinterpret_cast<uint4*>(p), n4);
  } else {
    size_t wg = (n_words + 255) / 256;
    MSDE_LAUNCH(zero_words_kernel, dim3((unsigned)(wg < 4096 ? wg : 4096)), dim3(256), 0, st, reinterpret_cast<uint32_t*>(p), n_words);
  }
  MSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int msde_linear_fwd(const float* X, const float* W, const float* bias, int M, int N, int K, float* Y,
                               void* stream) {
  if (M < 0 || N <= 0 || K <= 0 || !X || !W || !Y) return MSDE_EINVAL;
  if (M == 0) return 0;
  return launch_gemm<false, false>(X, W, bias, Y, nullptr, M, N, K, K, K, N, 1, (K + 31) / 32 * 32, false,
                                    as_stream(stream));
}

extern "C" int msde_linear_bwd_x(const float* gY, const float* W, int M, int N, int K, float* gX, void* stream) {
  if (M < 0 || N <= 0 || K <= 0 || !gY || !W || !gX) return MSDE_EINVAL;
  if (M == 0) return 0;
  // gX[M,K] = gY[M,N] . W[N,K]: reduce over N; B = W is k-major ([N][K], K contiguous)
  return launch_gemm<false, true>(gY, W, nullptr, gX, nullptr, M, K, N, N, K, K, 1, (N + 31) / 32 * 32, false,
                                   as_stream(stream));
}

// split policy of the weight gradient: 64 x 64 output tiles (300-wide layers pad to 320, not 384), then enough
// splits of the reduction (M) that ~512 workgroups (two per CU) are in flight, at least 64 rows per split, at
// most 512 splits.  MI355X, hipGraph-timed: 3588x300x300 23 us (vendor mm + colsum 39), 49090x128x128 32 us
// (vendor 225).  MSDE_WGRAD_TILE / MSDE_WGRAD_WGS are tuning knobs for tools/bench_wgrad.py.
static inline bool wgrad_big(int M, int N, int K) {
  return false;   // measured (tools/bench_wgrad.py): 64-wide tiles + ~512 workgroups win on every step shape (128-wide: 2.78 vs 2.72 ms)
}
static inline void wgrad_split_for(int M, int N, int K, int target, int* splits, int* k_per_split);
static inline void wgrad_split(int M, int N, int K, int* splits, int* k_per_split) {
  const int target = 512;
  wgrad_split_for(M, N, K, target, splits, k_per_split);
}
// the batched paths (msde_linear_bwd_w_partial / _describe + _grouped) share one launch among all layers, so a
// layer needs far fewer workgroups of its own: fewer, longer splits = less prologue / slab traffic per FLOP
#ifndef MSDE_WGRAD_TARGET
#define MSDE_WGRAD_TARGET 64
#endif
#ifndef MSDE_WGRAD_MIN_ROWS
#define MSDE_WGRAD_MIN_ROWS 256
#endif
#ifndef MSDE_WGRAD_TARGET_KS
#define MSDE_WGRAD_TARGET_KS 64
#endif
static inline void wgrad_split_batched(int M, int N, int K, int* splits, int* k_per_split) {
  // (a product that is narrow on both sides runs on the k-split tile, four times the rows per unit of time: fewer, longer splits)
  const int target = (N <= 32 && K <= 32 && (MSDE_WGRAD_NARROW & 4)) ? MSDE_WGRAD_TARGET_KS : MSDE_WGRAD_TARGET;
  wgrad_split_for(M, N, K, target, splits, k_per_split);
}
static inline void wgrad_split_for(int M, int N, int K, int target, int* splits, int* k_per_split) {
  bool big = wgrad_big(M, N, K);
  int tw_n = (big && N > 64) ? 128 : 64, tw_k = (big && K > 64) ? 128 : 64;
  long tiles = (long)((N + tw_n - 1) / tw_n) * ((K + tw_k - 1) / tw_k);
  long want = (target + tiles - 1) / tiles;
  // at least 256 rows (8 K tiles) per split: a 64-row split spends its time in the pipeline prologue and the 16 KB
  // slab write, and the grouped launch has thousands of workgroups anyway (the dense head alone adds ~150 problems)
  long maxs = (M + MSDE_WGRAD_MIN_ROWS - 1) / MSDE_WGRAD_MIN_ROWS;
  if (maxs < 1) maxs = 1;
  if (want > maxs) want = maxs;
  if (want > 512) want = 512;
  if (want < 1) want = 1;
  int kps = (int)(((M + want - 1) / want + LG_BK - 1) / LG_BK * LG_BK);
  if (kps < LG_BK) kps = LG_BK;
  *k_per_split = kps;
  *splits = M > 0 ? (M + kps - 1) / kps : 1;
}

// Split plan of one problem of the grouped launch, a function of the SHAPE only (the caller sizes the slabs from it before
// the operands' addresses are known).  Round 4 tried 128 x 128 tiles here for the node-level layers (2 x 2 sub-tiles per wave,
// dead 32 x 32 sub-tiles skipped; half the operand traffic per FLOP and half the barriers): 2.78 against 2.73 ms per step --
// the body then needs 216 VGPRs, two workgroups per CU, and the 64 x 64 tiles hide their load -> store -> barrier phases
// behind six other workgroups instead.  Round 5: 128 x 64 tiles (25 % fewer operand bytes per FLOP, 176 VGPRs) for layers with
// >= 128 outputs: 2.48-2.51 vs 2.50-2.51 ms, --full 3.39-3.40 vs 3.38-3.40 -- nothing; the launch is not bound by operand traffic.
static inline bool wgrad_group_plan(int M, int N, int K, int* splits, int* k_per_split) {
  wgrad_split_batched(M, N, K, splits, k_per_split);
  return false;
}

extern "C" long long msde_linear_bwd_w_workspace_bytes(int M, int N, int K) {
  int splits, kps;
  wgrad_split(M, N, K, &splits, &kps);
  return (long long)splits * ((long long)N * K + N) * (long long)sizeof(float);
}

// Batched form of the slab reduction: the weight-gradient GEMMs of a whole backward pass only write their
// slabs (msde_linear_bwd_w_partial) and ONE launch sums them all.  rows[r] = {slab address, splits, entries n, output
// address, row_len, slab_ld, out_ld, split_stride} as eight int64 (see below); prefix[r] = number of chunks before row r
// (prefix[count] = grid size).  A chunk is 256 entries summed by 4 split lanes (lane ly takes the splits ly, ly+4, ...),
// or -- rows with >= MSDE_REDUCE_LONG splits (per-workgroup slabs of the fused kernels: 256-512 of them) -- 64 entries
// summed by 16 split lanes: the chain of dependent loads per thread is 4x shorter, which is what bounds those rows.
// Fixed summation order (per lane in split order, lanes combined in lane order).
__global__ void __launch_bounds__(256)
reduce_slabs_multi_kernel(const long long* __restrict__ rows, const int* __restrict__ prefix, int count) {
  __shared__ float4 part[256];
  int lo = 0, hi = count;                  // last row with prefix[row] <= blockIdx.x
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (prefix[mid] <= (int)blockIdx.x) lo = mid; else hi = mid;
  }
  const long long* e = rows + (size_t)lo * 8;
  const float* slabs = reinterpret_cast<const float*>(e[0]);
  const int splits = (int)e[1];
  const size_t n = (size_t)e[2];
  float* out = reinterpret_cast<float*>(e[3]);
  // a row may be a 2-D block: n = nrows * row_len entries, entry (r, c) at slabs[z * split_stride + r * slab_ld + c] and
  // out[r * out_ld + c] (a weight gradient whose columns are a block of a wider parameter); flat rows: row_len = n
  const size_t row_len = (size_t)e[4], slab_ld = (size_t)e[5], out_ld = (size_t)e[6], sstride = (size_t)e[7];
  const int LY = splits >= MSDE_REDUCE_LONG ? 16 : 4, OXN = 256 / LY;
  const int ox = threadIdx.x % OXN, ly = threadIdx.x / OXN;
  const size_t i = (size_t)(blockIdx.x - prefix[lo]) * (4 * OXN) + 4 * ox;
  const bool flat = row_len == n;
  const bool vec = (n % 4 == 0) && ((reinterpret_cast<uintptr_t>(slabs) | reinterpret_cast<uintptr_t>(out)) & 15) == 0 &&
                   (flat ? sstride % 4 == 0 : ((row_len | slab_ld | out_ld | sstride) % 4 == 0));
  // offsets of the (up to) four entries this thread owns
  size_t so[4], oo[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const size_t j = i + q;
    if (flat) { so[q] = j; oo[q] = j; }
    else { const size_t r = j / row_len, c = j - r * row_len; so[q] = r * slab_ld + c; oo[q] = r * out_ld + c; }
  }
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n) {
    if (vec) {
      const float* base = slabs + so[0];
      int z = ly;
      for (; z + 3 * LY < splits; z += 4 * LY) {       // four independent 16-byte loads in flight
        float4 a = *reinterpret_cast<const float4*>(base + (size_t)z * sstride);
        float4 b = *reinterpret_cast<const float4*>(base + (size_t)(z + LY) * sstride);
        float4 c = *reinterpret_cast<const float4*>(base + (size_t)(z + 2 * LY) * sstride);
        float4 d = *reinterpret_cast<const float4*>(base + (size_t)(z + 3 * LY) * sstride);
        acc = vadd(vadd(vadd(vadd(acc, a), b), c), d);
      }
      for (; z < splits; z += LY) acc = vadd(acc, *reinterpret_cast<const float4*>(base + (size_t)z * sstride));
    } else {
      const bool h1 = i + 1 < n, h2 = i + 2 < n, h3 = i + 3 < n;
      int z = ly;
      for (; z + 3 * LY < splits; z += 4 * LY) {       // same order of additions as the tail loop, 4 rows in flight
        const float* p0 = slabs + (size_t)z * sstride;
        const float* p1 = slabs + (size_t)(z + LY) * sstride;
        const float* p2 = slabs + (size_t)(z + 2 * LY) * sstride;
        const float* p3 = slabs + (size_t)(z + 3 * LY) * sstride;
        const float a0 = p0[so[0]], a1 = p1[so[0]], a2 = p2[so[0]], a3 = p3[so[0]];
        const float b0 = h1 ? p0[so[1]] : 0.f, b1 = h1 ? p1[so[1]] : 0.f, b2 = h1 ? p2[so[1]] : 0.f, b3 = h1 ? p3[so[1]] : 0.f;
        const float c0 = h2 ? p0[so[2]] : 0.f, c1 = h2 ? p1[so[2]] : 0.f, c2 = h2 ? p2[so[2]] : 0.f, c3 = h2 ? p3[so[2]] : 0.f;
        const float d0 = h3 ? p0[so[3]] : 0.f, d1 = h3 ? p1[so[3]] : 0.f, d2 = h3 ? p2[so[3]] : 0.f, d3 = h3 ? p3[so[3]] : 0.f;
        acc.x = (((acc.x + a0) + a1) + a2) + a3;
        acc.y = (((acc.y + b0) + b1) + b2) + b3;
        acc.z = (((acc.z + c0) + c1) + c2) + c3;
        acc.w = (((acc.w + d0) + d1) + d2) + d3;
      }
      for (; z < splits; z += LY) {
        const float* p = slabs + (size_t)z * sstride;
        acc.x += p[so[0]];
        if (h1) acc.y += p[so[1]];
        if (h2) acc.z += p[so[2]];
        if (h3) acc.w += p[so[3]];
      }
    }
  }
  part[ly * OXN + ox] = acc;
  __syncthreads();
  if (ly == 0 && i < n) {
    float4 r = part[ox];
    for (int q = 1; q < LY; ++q) r = vadd(r, part[q * OXN + ox]);
    if (vec) {
      *reinterpret_cast<float4*>(out + oo[0]) = r;
    } else {
      out[oo[0]] = r.x;
      if (i + 1 < n) out[oo[1]] = r.y;
      if (i + 2 < n) out[oo[2]] = r.z;
      if (i + 3 < n) out[oo[3]] = r.w;
    }
  }
}

extern "C" long long msde_reduce_slabs_chunks(long long n, int splits) {
  const long long per = splits >= MSDE_REDUCE_LONG ? 64 : 256;
  return (n + per - 1) / per;
}

extern "C" int msde_reduce_slabs_multi(const long long* rows, const int* prefix, int count, int total_chunks,
                                       void* stream) {
  if (count < 0 || total_chunks < 0 || (count > 0 && (!rows || !prefix))) return MSDE_EINVAL;
  if (count == 0 || total_chunks == 0) return 0;
  MSDE_LAUNCH(reduce_slabs_multi_kernel, dim3(total_chunks), dim3(256), 0, as_stream(stream), rows, prefix, count);
  MSDE_CHECK_LAUNCH();
  return 0;
}

// Fills one row of the grouped-GEMM problem table (host memory, 12 int64) and returns the number of workgroups
// the problem needs (<= 0: error).  want_bias: the bias partials follow the weight slabs in `slabs`.
extern "C" int msde_linear_bwd_w_describe(const float* gY, const float* X, int M, int N, int K, int want_bias,
                                          float* slabs, const int* rows_dev, long long* row) {
  return msde_linear_bwd_w_describe_ld(gY, N, X, K, M, N, K, want_bias, slabs, rows_dev, row);
}

extern "C" int msde_linear_bwd_w_describe_ld(const float* gY, int ldg, const float* X, int ldx, int M, int N, int K,
                                             int want_bias, float* slabs, const int* rows_dev, long long* row) {
  if (M <= 0 || N <= 0 || K <= 0 || !gY || !X || !slabs || !row || ldg < N || ldx < K) return MSDE_EINVAL;
  if (wgrad_big(M, N, K)) return MSDE_EUNSUP;
  bool vec = aligned16(gY) && aligned16(X) && (N % 4 == 0) && (K % 4 == 0) && (ldg % 4 == 0) && (ldx % 4 == 0);
  int splits, kps;
  wgrad_group_plan(M, N, K, &splits, &kps);
  vec = vec && (kps % 4 == 0);
  // tile shape (gemm_grouped_wgrad_body): outputs are N x K.  A 32-wide (or narrower) side leaves half of the 64 x 64 tile's waves
  // idle -- 32 x 128 / 128 x 32 tiles put the four waves side by side along the wide dimension, and a product that is narrow on
  // BOTH sides lets them split the rows of a 128-row K tile (wgrad_narrow_body).  Same splits as the 64 x 64 plan.
  int shape = 0;
  if (vec) {
    if (N <= 32 && K <= 32) shape = (MSDE_WGRAD_NARROW & 4) ? 3 : 0;
    else if (N <= 32 && K > 64) shape = (MSDE_WGRAD_NARROW & 1) ? 1 : 0;
    else if (K <= 32 && N > 64) shape = (MSDE_WGRAD_NARROW & 2) ? 2 : 0;
  }
  const int bm = (shape == 1 || shape == 3) ? 32 : shape == 2 ? 128 : 64, bn = (shape == 2 || shape == 3) ? 32 : shape == 1 ? 128 : 64;
  int tx = (K + bn - 1) / bn, ty = (N + bm - 1) / bm;
  // (round 4, tools/ab_multi.sh on one box: edge tiles on the fast path 2.70-2.73 vs 2.69-2.72 ms, the 128-VGPR build
  // MSDE_WGRAD_OCC4 2.70-2.71, register prefetch two / three tiles deep (-DLG_ST) 2.72 vs 2.71: none of them moves the step)
  const int relax = 0;
  row[12] = ldg; row[13] = ldx; row[14] = reinterpret_cast<long long>(rows_dev); row[15] = relax | (shape << 1);
  row[0] = reinterpret_cast<long long>(gY);
  row[1] = reinterpret_cast<long long>(X);
  row[2] = reinterpret_cast<long long>(slabs);
  row[3] = want_bias ? reinterpret_cast<long long>(slabs + (size_t)splits * N * K) : 0;
  row[4] = M; row[5] = N; row[6] = K; row[7] = splits; row[8] = kps; row[9] = tx; row[10] = ty; row[11] = vec ? 1 : 0;
  return tx * ty * splits;
}

extern "C" int msde_linear_bwd_w_grouped_ex(const long long* probs, const int* prefix, int count, int total_blocks,
                                            int max_workgroups, void* stream) {
  if (count < 0 || total_blocks < 0 || (count > 0 && (!probs || !prefix))) return MSDE_EINVAL;
  if (count == 0 || total_blocks == 0) return 0;
  const int xcd = 1;
  const int grid = max_workgroups > 0 && max_workgroups < total_blocks ? max_workgroups : total_blocks;
  const int occ4 = 0;
  if (occ4)
    MSDE_LAUNCH(gemm_grouped_wgrad_occ4_kernel, dim3(grid), dim3(256), 0, as_stream(stream), probs, prefix, count, total_blocks,
                grid == total_blocks ? xcd : 0);
  else
    MSDE_LAUNCH(gemm_grouped_wgrad_kernel, dim3(grid), dim3(256), 0, as_stream(stream), probs, prefix, count, total_blocks,
                grid == total_blocks ? xcd : 0);
  MSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int msde_linear_bwd_w_grouped(const long long* probs, const int* prefix, int count, int total_blocks,
                                         void* stream) {
  return msde_linear_bwd_w_grouped_ex(probs, prefix, count, total_blocks, 0, stream);
}

extern "C" int msde_linear_bwd_w_splits(int M, int N, int K) {
  int splits, kps;
  wgrad_group_plan(M, N, K, &splits, &kps);
  return splits;
}

// GEMM half of msde_linear_bwd_w: slabs [splits][N*K] followed (when want_bias) by the bias partials
// [splits][N]; sum them with msde_reduce_slabs_multi (or msde_linear_bwd_w does both).
extern "C" int msde_linear_bwd_w_partial(const float* gY, const float* X, int M, int N, int K, int want_bias,
                                         float* slabs, const int* rows_dev, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || !gY || !X || !slabs) return MSDE_EINVAL;
  int splits, k_per_split;
  wgrad_group_plan(M, N, K, &splits, &k_per_split);       // (same splits as the grouped launch: the caller sized the slabs from them)
  float* cs = want_bias ? slabs + (size_t)splits * N * K : nullptr;
  return launch_gemm<true, true>(gY, X, nullptr, slabs, cs, N, K, M, N, K, K, splits, k_per_split, wgrad_big(M, N, K),
                                 as_stream(stream), rows_dev);
}

extern "C" int msde_linear_bwd_w(const float* gY, const float* X, int M, int N, int K, float* gW, float* gb,
                                 float* workspace, const int* rows_dev, void* stream) {
  if (M < 0 || N <= 0 || K <= 0 || !gY || !X || !gW || !workspace) return MSDE_EINVAL;
  hipStream_t st = as_stream(stream);
  int splits, k_per_split;
  wgrad_split(M, N, K, &splits, &k_per_split);
  float* slabs = workspace;
  float* cs = gb ? workspace + (size_t)splits * N * K : nullptr;
  if (M == 0) {
    int e = msde_zero_words(gW, (size_t)N * K, st);
    if (e == 0 && gb) e = msde_zero_words(gb, (size_t)N, st);
    return e;
  }
  // C[N,K] = A^T B with A = gY [M][N] (k-major, "M" of the product = N), B = X [M][K] (k-major)
  if (splits == 1)      // one split (few rows: a 21-atom MD17 step launches ~90 of these): the only slab IS the result
    return launch_gemm<true, true>(gY, X, nullptr, gW, gb, N, K, M, N, K, K, 1, k_per_split, wgrad_big(M, N, K), st, rows_dev);
  int rc = launch_gemm<true, true>(gY, X, nullptr, slabs, cs, N, K, M, N, K, K, splits, k_per_split, wgrad_big(M, N, K), st, rows_dev);
  if (rc != 0) return rc;
  return msde_reduce_slabs(slabs, splits, (size_t)N * K, gW, cs, (size_t)N, gb, st);
}
