// gemm_rs.h — device building blocks of the row-strip fp32 matrix-core GEMMs (gfx950, v_mfma_f32_16x16x4_f32).
//
// The node-level products of the hot path are skinny: M = 3.6 k atoms against N, K in {128, 300, 600}.  A 64 x 64 output
// tiling gives 285 workgroups of very different use for 256 CUs; what fits the chip is one workgroup per STRIP of 16 or
// 32 rows that owns all (or half) of the output columns:
//   * the strip of A (16 RT rows x all of K) is staged in LDS ONCE -- after that the K loop has no barrier at all;
//   * every wave owns its own 16-column tiles, so the B operand (the weights) is never shared between waves and goes
//     straight from global memory (L2 resident) to registers.  The weights are read in the [K][N] layout: for one k the
//     16 column lanes of a wave read 16 (x 1, 2 or 4) consecutive floats -- coalesced runs of 64 to 256 B.  (Measured:
//     the [N][K] layout with 16-B pieces along k per lane -- 64 different 128-B lines per load -- is bound by the
//     address path: 22 us against 16 us for 3588 x 300 x 300; forward products therefore read a transposed copy of
//     the weight, see hip.wt_cache);
//   * the MFMA sums over its 4 k lanes, so lane group g may own k = 32 kb + 8 g + j in step j as long as A agrees:
//     A fragments are two 16-B pieces (8 consecutive k) read from the LDS strip (row stride = 4 mod 32 floats);
//   * 16 x 16 tiles quantise N = 300 into 19 tiles (1 % padding; 32-wide tiles: 6 %, 64-wide: 7 %).
// Because the strip lives in LDS, whatever produces it can be fused in front (BatchNorm apply + ReLU on load, the
// BatchNorm backward formula, a neighbour gather) and whatever consumes the output strip can be chained behind
// (the next Linear of an MLP) without a trip through memory.
#pragma once
#include "msde_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define RS_KB 32                                  // k per block
#define RS_PAD 4                                  // LDS row stride = Kpad + 4 floats (4 mod 32: see rs_lds_ld)
#ifndef RS_STAGES
#define RS_STAGES 3                               // register stages of the operand pipeline (rsa_mma)
#endif

__device__ __forceinline__ f32x4 rs_mfma(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__host__ __device__ __forceinline__ int rs_kpad(int K) { return (K + RS_KB - 1) / RS_KB * RS_KB; }
__host__ __device__ __forceinline__ int rs_lds_ld(int K) { return rs_kpad(K) + RS_PAD; }
__device__ __forceinline__ float rs_f4(const float4& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }

// Column tiles of one wave.  The wave owns the T * 16 consecutive columns starting at wcol; they are cut into segments of
// W in {4, 2, 1} tiles (T = 5: 4 + 1, 4: 4, 3: 2 + 1, 2: 2, 1: 1).  Inside a segment starting at column c0 the tiles are
// INTERLEAVED: tile t holds the columns c0 + W i + t (i = lane & 15), so that lane i's W values for one k are W
// consecutive floats of a weight row [k][.] -- one 16-B (8-B, 4-B) load per lane, whole 256-B (128-B, 64-B) runs per
// 16 lanes -- and its W results for one row are consecutive floats of the output row (vector stores).
template <int T> struct RsSeg;
template <> struct RsSeg<1> { static constexpr int NSEG = 1; static constexpr int W[2] = {1, 0}; };
template <> struct RsSeg<2> { static constexpr int NSEG = 1; static constexpr int W[2] = {2, 0}; };
template <> struct RsSeg<3> { static constexpr int NSEG = 2; static constexpr int W[2] = {2, 1}; };
template <> struct RsSeg<4> { static constexpr int NSEG = 1; static constexpr int W[2] = {4, 0}; };
template <> struct RsSeg<5> { static constexpr int NSEG = 2; static constexpr int W[2] = {4, 1}; };
// tile c of the wave: segment, position in it, first column of its segment relative to wcol
template <int T> __host__ __device__ constexpr int rs_tile_w(int c) { return c < RsSeg<T>::W[0] ? RsSeg<T>::W[0] : RsSeg<T>::W[1]; }
template <int T> __host__ __device__ constexpr int rs_tile_t(int c) { return c < RsSeg<T>::W[0] ? c : c - RsSeg<T>::W[0]; }
template <int T> __host__ __device__ constexpr int rs_tile_c0(int c) { return c < RsSeg<T>::W[0] ? 0 : 16 * RsSeg<T>::W[0]; }
// column of lane-column i of tile c
template <int T> __device__ __forceinline__ int rs_col(int wcol, int c, int i) {
  return wcol + rs_tile_c0<T>(c) + rs_tile_w<T>(c) * i + rs_tile_t<T>(c);
}

// acc[c][r] += strip[r-th 16 rows] . B[:, columns of tile c] for one wave.
//   As      LDS strip [16 RT][ld] (zero for k in [K, Kpad)), ld = rs_lds_ld(K)
//   B       [K][N] row-major, row stride ldb (the weight of an input-gradient product as stored, W[out][in]; the
//           TRANSPOSED weight for a forward product); N % 4 == 0, ldb % 4 == 0, 16-B aligned
//   wcol    first column of the wave; columns >= N read whatever follows in memory or zeros (never stored)
// Per k block of 32: lane group g owns k = 32 kb + 8 g + j in step j -- 8 loads per segment, 8 T RT MFMAs.
typedef unsigned int rs_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int rs_u32x2 __attribute__((ext_vector_type(2)));

// rot: the k blocks are visited in the order rot, rot + 1, ..., wrapping around (rot < blocks): workgroups that start
// together then read DIFFERENT weight rows at any moment instead of all hitting the same L2 channel.
// SYNC: the strip is still being staged by the workgroup; the barrier that publishes it is taken here, AFTER the first
// weight requests are in flight.
template <int RT, int T, bool SYNC = false>
__device__ __forceinline__ void rsa_mma(const float* __restrict__ As, int ld, const float* __restrict__ B, int ldb, int N,
                                        int K, int wcol, f32x4 (&acc)[T][RT], int rot = 0) {
  const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
  const int nkb = rs_kpad(K) / RS_KB;
  auto blk = [&](int i) { const int q = i + rot; return q >= nkb ? q - nkb : q; };
  constexpr int NSEG = RsSeg<T>::NSEG, W0 = RsSeg<T>::W[0], W1 = RsSeg<T>::W[1];
  // The weights are read through a buffer descriptor: per-lane 32-bit byte offset (constant for the whole loop) + a
  // scalar offset that moves with k -- no vector address arithmetic in the loop -- and rows past K (the K tail) or past
  // the last row's N columns come back as zeros from the hardware bounds check, so no block is special.
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(B), 0, (int)(((size_t)(K - 1) * (size_t)ldb + (size_t)N) * 4), 0x00020000);
  const unsigned off0 = ((unsigned)(8 * g) * (unsigned)ldb + (unsigned)(wcol + W0 * n)) * 4u;
  const unsigned off1 = ((unsigned)(8 * g) * (unsigned)ldb + (unsigned)(wcol + 16 * W0 + W1 * n)) * 4u;
  const unsigned rowb = (unsigned)ldb * 4u;          // bytes per k row (uniform)
  auto ld_seg = [&](unsigned voff, unsigned soff, float (&dst)[T], int first, int w) {
    if (w == 4) {
      const rs_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
      dst[first] = __uint_as_float(v.x); dst[first + 1] = __uint_as_float(v.y);
      dst[first + 2] = __uint_as_float(v.z); dst[first + 3] = __uint_as_float(v.w);
    } else if (w == 2) {
      const rs_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, soff, 0);
      dst[first] = __uint_as_float(v.x); dst[first + 1] = __uint_as_float(v.y);
    } else {
      dst[first] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, soff, 0));
    }
  };
  auto loadB = [&](int kb, float (&b)[8][T]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const unsigned soff = (unsigned)(kb * RS_KB + j) * rowb;
      ld_seg(off0, soff, b[j], 0, W0);
      if (NSEG > 1) ld_seg(off1, soff, b[j], W0, W1);
    }
  };
  const float* __restrict__ arow = As + n * ld + 8 * g;
  auto readA = [&](int kb, float4 (&a)[RT][2]) {
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      a[r][0] = *reinterpret_cast<const float4*>(arow + r * 16 * ld + kb * RS_KB);
      a[r][1] = *reinterpret_cast<const float4*>(arow + r * 16 * ld + kb * RS_KB + 4);
    }
  };
  auto mfmas = [&](const float4 (&a)[RT][2], const float (&b)[8][T]) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int c = 0; c < T; ++c)
#pragma unroll
        for (int r = 0; r < RT; ++r) acc[c][r] = rs_mfma(rs_f4(a[r][j >> 2], j & 3), b[j][c], acc[c][r]);
  };
  // Software pipeline with NS register stages: the operands of block kb + NS - 1 are requested BEFORE the MFMAs of block
  // kb, so a request has NS - 1 blocks of matrix work (8 T RT MFMAs each) to come back.  The scheduling barriers keep
  // hipcc from sinking the requests down to their first use (it did: every load sat behind a vmcnt(0) next to its
  // MFMA).  Blocks past the last one are clamped to the last (a few redundant requests at the very end).
  constexpr int NS = RS_STAGES;
  float bq[NS][8][T];
  float4 aq[NS][RT][2];
  const int last = nkb - 1;
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) loadB(blk(min(s, last)), bq[s]);
  if (SYNC) {
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
  }
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) readA(blk(min(s, last)), aq[s]);
  int kb = 0;
  for (; kb + NS <= nkb; kb += NS) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int nx = min(kb + s + NS - 1, last);
      loadB(blk(nx), bq[(s + NS - 1) % NS]);
      readA(blk(nx), aq[(s + NS - 1) % NS]);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(aq[s], bq[s]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // the remaining 0 .. NS - 1 blocks are already in flight: stage s holds block kb + s
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) {
    if (kb + s < nkb) {
      __builtin_amdgcn_sched_barrier(0);
      mfmas(aq[s], bq[s]);
    }
  }
}

// Stage a strip of `rows` x K floats of a row-major matrix into LDS (zero beyond M / K) through `xf(v, k, row)`:
// all 256 threads, 16-B pieces, coalesced along k.  K % 4 == 0, lda % 4 == 0, A 16-B aligned.
template <typename XF>
__device__ __forceinline__ void rs_stage_strip(float* __restrict__ As, int ld, const float* __restrict__ A, int lda, int m0,
                                               int rows, int M, int K, const XF& xf) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int kq = rs_kpad(K) / 4;
  for (int r = wave; r < rows; r += 4) {
    const int gm = m0 + r;
    for (int q = lane; q < kq; q += 64) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gm < M && 4 * q < K) v = xf(*reinterpret_cast<const float4*>(A + (size_t)gm * lda + 4 * q), 4 * q, gm);
      *reinterpret_cast<float4*>(As + r * ld + 4 * q) = v;
    }
  }
}
