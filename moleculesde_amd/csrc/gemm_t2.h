// gemm_t2.hip — 2-D tiled fp32 matrix-core GEMM for the node-level products (gfx950, v_mfma_f32_16x16x4_f32): the forward
// and input-gradient products of the plain nn.Linear layers of the encoders at M = a few thousand atoms.  Reference call
// sites: Geom3D/models/molecule_gnn_model.py:17,28-29,176-182 (GIN MLP + BatchNorm), Geom3D/models/schnet.py:141-148,
// 163-167 (lin1 / lin2 / lin), Geom3D/models/MoleculeSDE/SDE_model_2D_to_3D.py:264-271 -- torch.addmm / torch.mm there.
//
// Why a second kernel beside the row strips of gemm_rs.hip: a 16-row strip streams the WHOLE weight matrix through its CU
// for 16 rows of output (8 FLOP per byte of L2 -> L1 traffic; matrix pipe 27 % busy in the step, round 3).  Here a
// workgroup owns a BM x BN output tile (BM = 64 or 128 rows, BN = N / splits columns), both operands go through LDS in
// K tiles of 32, and every weight byte a CU fetches is used for BM rows:
//   * staging is LDS-DMA (`buffer_load_dwordx4 ... lds`): no VGPRs, no ds_write, no vector instruction in the K loop --
//     which matters doubly for fp32 MFMA, whose issue slots vector instructions do not hide behind.  One piece =
//     8 rows x 128 B (full cache lines of the k-contiguous operand rows) = 1 KiB of LDS.  The 16-B chunks of a row are
//     XOR-swizzled by ((row >> 1) & 7) on the SOURCE side (LDS-DMA writes lane-linear), the same involution on the read
//     side: every ds_read_b128 fragment read is conflict-free;
//   * both operands are read k-CONTIGUOUS: A [M][K] as stored, B as [N][K] -- nn.Linear's weight as stored for a forward
//     product, its transposed copy for an input-gradient product (the opposite of gemm_rs.hip);
//   * the MFMA sums over its 4 k lanes, so lane group q may own k = 16 s + 4 q + j in step j of sub-tile s as long as A
//     and B agree: one 16-B read per 16 x 16 fragment per 16 k;
//   * wave w owns rows 16 w .. 16 w + 15 of the tile and all BN columns (RN accumulator tiles): A rows are private to a
//     wave, B fragments are shared through LDS;
//   * ring of NBUF stages, one barrier per K tile placed in the MIDDLE of the tile's matrix work (the fragments of the
//     second half are already in registers), LDS-DMA requests counted with s_waitcnt vmcnt(N) and never drained in the loop;
//   * optionally (LOADER) a fifth / ninth wave issues every LDS-DMA request, so the computing waves issue MFMAs and LDS
//     reads only.
// Epilogue (bias, activation / derivative, residual, BatchNorm partial statistics per 16-row strip) = gemm_rs_epi.h.
#pragma once
#include "gemm_rs_epi.h"
#include <type_traits>
#include <map>
#include <mutex>
#ifndef T2_NBUF
#define T2_NBUF 4          // stages of the LDS ring (3: measured, tools/ab_libs.sh)
#endif

typedef int t2_i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int t2_u32x4 __attribute__((ext_vector_type(4)));
#define T2_OOB 0x80000000u      // per-lane byte offset beyond any operand (< 2 GiB): the range check returns zeros

__device__ __forceinline__ t2_i32x4 t2_rsrc(const void* p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  t2_i32x4 r;
  r.x = (int)(unsigned)(a & 0xFFFFFFFFull);
  r.y = (int)(unsigned)((a >> 32) & 0xFFFFull);
  r.z = (int)bytes;
  r.w = 0x00020000;
  return r;
}

// One LDS-DMA piece: lane l fetches 16 B at byte offset voff (+ soff) of the buffer and the wave's 1 KiB lands at LDS byte
// address lds_addr + 16 l.  Hidden from the compiler's wait bookkeeping on purpose (it would drain vmcnt before every
// ds_read): completion is counted by hand.  M0 is written and read inside the one statement; the compiler has no use for M0 in
// these kernels (no other LDS-DMA, no indexed register moves: checked in the .s: `m0` appears only in these statements).
__device__ __forceinline__ void t2_glds(unsigned voff, t2_i32x4 rsrc, unsigned soff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds"
               :
               : "v"(voff), "s"(rsrc), "s"(lds_addr), "s"(soff)
               : "memory");
}
// (vmcnt is a 6-bit field: a larger allowance is clamped to 63, a stronger wait than asked for, never a weaker one)
// acc += a x b with the accumulator tied in place.  (Written through the builtin, hipcc rotated the RN accumulators through
// each other across the loop back edge: 4 RN + 4 v_accvgpr_mov per trip, a tenth of the loop's issue time.)  The operands come
// straight from ds_read results (the compiler's own waits cover them); the wait states between the last MFMA and the first
// read of an accumulator are supplied by t2_mfma_drain().
__device__ __forceinline__ void t2_mfma(float a, float b, f32x4& c) {
  asm("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
// hipcc does not know these statements are MFMAs, so it pads none of their hazards (cdna_hip_programming.md 5.7):
//  * accumulator written by an MFMA -> ANY other reader (a register copy the compiler places at a control-flow merge, the
//    epilogue): 16 wait states, and the statement below names the accumulators as read-write operands so that such a
//    reader cannot be scheduled in front of it.  Called at the end of every K tile (2 % of a tile's issue time);
//  * a register the compiler has just written (it does place v_accvgpr_mov copies of single accumulators BETWEEN these
//    statements at merges: seen, with wrong element 0 of every accumulator but the first as the result) -> MFMA operand: every
//    statement opens with s_nop 1 (2 wait states, inside the previous MFMA's 32 issue cycles).
template <int RN> __device__ __forceinline__ void t2_mfma_drain(f32x4 (&acc)[RN][1]) {
  asm volatile("s_nop 15" ::: "memory");
#pragma unroll
  for (int t = 0; t < RN; ++t) asm volatile("" : "+a"(acc[t][0]));
}

template <int N> __device__ __forceinline__ void t2_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N > 63 ? 63 : N) : "memory"); }
__device__ __forceinline__ void t2_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_barrier" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// Column of lane-column i of accumulator tile c (relative to the tile's first column): tiles are grouped 4 + 4 + .. + 2 + 1
// and INTERLEAVED inside a group of W (column = group base + W i + position), so that a lane's W results of one row are W
// consecutive floats (vector stores, vector bias / residual loads) -- the layout rs_epi_segment expects.
template <int RN> __host__ __device__ constexpr int t2_col(int c, int i) {
  constexpr int n4 = RN / 4, rem = RN % 4;
  if (c < 4 * n4) return 64 * (c / 4) + 4 * i + (c % 4);
  const int cc = c - 4 * n4;
  if (rem >= 2 && cc < 2) return 64 * n4 + 2 * i + cc;
  return 16 * (RN - 1) + i;
}

template <int RN>
__device__ __forceinline__ void t2_epilogue(const msde_rs_desc& d, f32x4 (&acc)[RN][1], int n0, int m0, int strip) {
  const int n = threadIdx.x & 15;
  constexpr int n4 = RN / 4, rem = RN % 4;
  if constexpr (n4 >= 1) rs_epi_segment<1, RN, 4, 0>(d, acc, n0 + 4 * n, m0, strip, 16);
  if constexpr (n4 >= 2) rs_epi_segment<1, RN, 4, 4>(d, acc, n0 + 64 + 4 * n, m0, strip, 16);
  if constexpr (n4 >= 3) rs_epi_segment<1, RN, 4, 8>(d, acc, n0 + 128 + 4 * n, m0, strip, 16);
  if constexpr (rem >= 2) rs_epi_segment<1, RN, 2, 4 * n4>(d, acc, n0 + 64 * n4 + 2 * n, m0, strip, 16);
  if constexpr (rem & 1) rs_epi_segment<1, RN, 1, RN - 1>(d, acc, n0 + 16 * (RN - 1) + n, m0, strip, 16);
}

// Eight waves per workgroup: wave w computes rows 16 (w & 3) .. + 15 of the tile against ALL BN columns for the k-half
// h = w >> 2 of every K tile (k = 32 tile + 16 h + 0..15), so each SIMD holds two waves that run the same MFMA blocks on
// different halves -- when one waits (barrier, fragment reads, request issue) the other keeps the matrix pipe busy.
// (Measured with one wave per SIMD: the K loop ran at 71-81 % of its MFMA time; two 4-wave workgroups per CU at 98 %.)
// The two partial accumulators of a row block are added through LDS after the loop (fixed order: deterministic).
// AXF (msde_rs_desc.axf): a transform applied to the A FRAGMENTS in registers, right before their MFMAs -- the rows of a
// fragment are private to the wave, so nothing is transformed twice inside a workgroup (the column splits of a row block
// repeat it; K tile t of the transformed A is written back, for the weight gradient, by the split t % splits).  The
// per-column vectors of the transform sit zero-padded in LDS behind the ring; MSDE_RS_AXF_BNBWD stages its second operand
// (z) like A.  Vector instructions do not hide behind fp32 MFMAs: the transform costs its 8 (affine) to 20 (BatchNorm
// backward) instructions per K tile and wave on top, against 4 RN MFMAs.
template <int RN, int AXF = 0, int ABL = 0>
__global__ void __launch_bounds__(512)
gemm_t2_kernel(const msde_rs_desc d) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char t2_smem[];
  static_assert(RN >= 1 && RN <= 12, "geometry");
  constexpr int NBUF = T2_NBUF;
  constexpr int BM = 64, BN = 16 * RN;
  constexpr int NA = AXF == MSDE_RS_AXF_BNBWD ? 2 : 1;          // row operands staged per stage: A (and z)
  constexpr int AP = BM / 8, BP = BN / 8, P = NA * AP + BP;     // 1 KiB pieces of the A (, z) / B tile of one stage
  constexpr int ST = (NA * BM + BN) * 128;                      // bytes per stage
  constexpr int BOFF = NA * AP * 1024;                          // B tile inside a stage
  constexpr int NV = AXF == MSDE_RS_AXF_BNBWD ? 5 : (AXF == MSDE_RS_AXF_AFFINE ? 2 : 0);   // per-column vectors in LDS
  constexpr int PW = (P + 7) / 8;                               // pieces per wave and stage (the last waves: one less)
  constexpr int STS = ST + 1024;                                // stage stride: + 1 KiB where the dummy pieces land
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int wr = wave & 3, h = wave >> 2;
  const int M = d.M, N = d.N, K = d.K;
#ifdef T2_TIMING      // diagnostic build (tools/t2_phases.py): per-workgroup cycle stamps into the buffer passed as xf4
  long long* t2_dbg = reinterpret_cast<long long*>(const_cast<float*>(d.xf4)) + (size_t)blockIdx.x * 8;
#define T2_STAMP(i_) do { if (threadIdx.x == 0 && d.xf4) { t2_dbg[i_] = (long long)__builtin_amdgcn_s_memtime(); } } while (0)
  if (threadIdx.x == 0 && d.xf4) t2_dbg[6] = (long long)__builtin_amdgcn_s_memrealtime();
#else
#define T2_STAMP(i_) do { } while (0)
#endif
  // ablation bits (timing builds only, results are wrong): 1 no barrier, 2 no request wait, 4 no requests, 8 no fragment
  // reads, 16 no MFMAs
  constexpr int abl = ABL;
  T2_STAMP(0);
  // workgroup -> tile.  Workgroups b and b + 8 share an XCD (round-robin dispatch: speed only): consecutive tiles of the
  // linear order -- the column splits of one row block, then the next row block -- go to ONE XCD, so a row block of A is
  // fetched into one L2.  Bijective for any grid size.
  int lin;
  {
    const int b = blockIdx.x, nwg = gridDim.x, x = b & 7, j = b >> 3, q = nwg >> 3, r = nwg & 7;
    lin = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
  }
  const int S = d.splits;
  const int rowblk = lin / S, split = lin - rowblk * S;
  const int m0 = rowblk * BM, n0 = split * BN;
  const int nt = (K + 31) >> 5;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)t2_smem;

  // ---- this wave's LDS-DMA pieces: piece p = wave + 8 i of a stage (p < 8: rows 8 p .. of the A tile, else of the B tile)
  unsigned vo[PW], vt[PW];               // per-lane byte offsets: K tiles 0 .. nt - 2 / the last tile (chunks at k >= K: zeros)
  {
    const int ktail = (nt - 1) * 32;
#pragma unroll
    for (int i = 0; i < PW; ++i) {
      const int p = wave + 8 * i;
      const int c = (lane & 7) ^ (((p & 1) << 2) | (lane >> 4)); // source chunk that lands in slot lane & 7 of its row
      unsigned o = T2_OOB;
      if (i < NA) {
        const int row = m0 + 8 * (p - 8 * i) + (lane >> 3);
        if (row < M) o = ((unsigned)row * (unsigned)(i == 0 ? d.lda : d.lda2) + 4u * (unsigned)c) * 4u;
      } else if (p < P) {
        const int rho = 8 * (p - NA * AP) + (lane >> 3);
        const int n = n0 + t2_col<RN>(rho >> 4, rho & 15);
        if (n < N) o = ((unsigned)n * (unsigned)d.ldb + 4u * (unsigned)c) * 4u;
      }
      vo[i] = o;
      vt[i] = (ktail + 4 * c < K) ? o : T2_OOB;
    }
  }
  const t2_i32x4 rsA = t2_rsrc(d.A, (unsigned)(((size_t)(M - 1) * (size_t)d.lda + (size_t)K) * 4));
  const t2_i32x4 rsB = t2_rsrc(d.B, (unsigned)(((size_t)(N - 1) * (size_t)d.ldb + (size_t)K) * 4));
  const t2_i32x4 rsZ = t2_rsrc(NA == 2 ? d.A2 : d.A, (unsigned)(((size_t)(M - 1) * (size_t)(NA == 2 ? d.lda2 : d.lda) + (size_t)K) * 4));
  // per-column vectors of the transform -> LDS, zero beyond K (a fragment chunk at k >= K is zero and must stay zero)
  constexpr unsigned XF = NBUF * STS;
  const int KP = nt * 32;
  if (NV > 0) {
    const float* vsrc[5] = {d.xf0, d.xf1, d.xf2, d.xf3, d.xf4};
    float* xl = reinterpret_cast<float*>(t2_smem + XF);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      // BatchNorm backward without a ReLU gate (xf3 == NULL): gate vectors (0, 1) -> z * 0 + 1 > 0 always
      const float dflt = (v == 4) ? 1.f : 0.f;
      for (int k = 4 * threadIdx.x; k < KP; k += 4 * 512) {
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < K) x = vsrc[v] ? *reinterpret_cast<const float4*>(vsrc[v] + k) : make_float4(dflt, dflt, dflt, dflt);
        *reinterpret_cast<float4*>(xl + (size_t)v * KP + k) = x;
      }
    }
  }
  // bias of this lane's columns, requested before anything else (an epilogue that starts with a dependent load pays a
  // memory round trip per column segment: 4.7 k of 41 k cycles at N = 300, K = 600)
  float bv[RN];
#pragma unroll
  for (int t = 0; t < RN; ++t) {
    const int col = n0 + t2_col<RN>(t, lane & 15);
    bv[t] = (d.bias && h == 0 && col < N) ? d.bias[col] : 0.f;
  }
  // Piece i of K tile `tile` into stage `stage`.  GEN = false: an interior tile (offsets vo).  GEN = true: decided at run
  // time -- the last tile uses vt (chunks at k >= K read as zeros), tiles past the end are requested out of range into the
  // stage's spare KiB (a request, no memory traffic), so that every tile_block issues the same number of requests and one
  // counted wait serves the whole loop.  Waves with wave >= NFULL own one piece less (FULLP - 1): nothing is padded.
  constexpr int NFULL = P - 8 * (PW - 1);                        // waves 0 .. NFULL - 1 own PW pieces, the others PW - 1
  const bool fullw = wave < NFULL;
  auto issue1 = [&](auto gen_, int tile, int stage, int i) __attribute__((always_inline)) {
    constexpr bool GEN = decltype(gen_)::value;
    const bool live = !GEN || tile < nt, last = GEN && tile == nt - 1;
    unsigned v = vo[i];
    if (GEN) v = live ? (last ? vt[i] : vo[i]) : T2_OOB;
    // piece p = wave + 8 i lands at byte 1024 p of its stage
    const unsigned la = lds0 + (unsigned)stage * (unsigned)STS + (live ? (unsigned)wave * 1024u + 8192u * (unsigned)i : (unsigned)ST);
    const unsigned so = live ? (unsigned)tile * 128u : 0u;
    if (i == 0) t2_glds(v, rsA, so, la);
    else if (i < NA) t2_glds(v, rsZ, so, la);
    else t2_glds(v, rsB, so, la);
  };
  using GENERIC = std::true_type;
  using INTERIOR = std::false_type;
#pragma unroll
  for (int s = 0; s < NBUF - 1; ++s)
#pragma unroll
    for (int i = 0; i < PW; ++i)
      if (i < PW - 1 || fullw) issue1(GENERIC{}, s, s, i);

  const int r = lane & 15, q = lane >> 4;
  const unsigned lo = (unsigned)(r * 128 + ((q ^ ((r >> 1) & 7)) << 4)) ^ (unsigned)(h * 64);
  f32x4 acc[RN][1];
#pragma unroll
  for (int t = 0; t < RN; ++t) acc[t][0] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (fullw) t2_wait_vm<(NBUF - 2) * PW>(); else t2_wait_vm<(NBUF - 2) * (PW - 1)>();
  // (the transform's vectors were written to LDS with ordinary stores: they must have landed before the barrier that
  // publishes them -- the asm barrier below is opaque to the compiler, which therefore adds no such wait itself)
  if (NV > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  t2_barrier();                                                  // tile 0 has landed for every wave
  T2_STAMP(1);
  float4 fa[2], fb[2][RN];                                       // fragments of this wave's k-half: two sets (tile parity)
  float4 fz[2], fx[2][NV > 0 ? NV : 1];                          // z fragment and the transform's vectors for the same k
  const unsigned xlo = XF + (unsigned)(q * 16 + h * 64);         // this lane's 4 k of a tile: 32 tile + 16 h + 4 q ..
  auto rd_aux = [&](int tile, int stage, float4& z, float4 (&x)[NV > 0 ? NV : 1]) __attribute__((always_inline)) {
    if (NA == 2) z = *reinterpret_cast<const float4*>(t2_smem + stage * STS + lo + AP * 1024 + wr * 2048);
#pragma unroll
    for (int v = 0; v < NV; ++v) x[v] = *reinterpret_cast<const float4*>(t2_smem + xlo + ((size_t)v * KP + 32 * tile) * 4);
  };
  {
    const unsigned char* base = t2_smem + lo;
    fa[0] = *reinterpret_cast<const float4*>(base + wr * 2048);
#pragma unroll
    for (int t = 0; t < RN; ++t) fb[0][t] = *reinterpret_cast<const float4*>(base + BOFF + t * 2048);
    rd_aux(0, 0, fz[0], fx[0]);
  }
  // the transform of one A fragment (4 consecutive k of one row per lane); the transformed K tile `tile` is written back by
  // the column split tile % splits
  const float lbound = (d.flags & MSDE_RS_AXF_RELU) ? 0.f : -3.0e38f;
  const int arow = m0 + 16 * wr + r;
  // A_out through a buffer descriptor (zero records when there is no A_out: every store is dropped)
  const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(
      d.A_out ? d.A_out : const_cast<float*>(d.A), 0,
      (AXF != MSDE_RS_AXF_NONE && d.A_out) ? (int)(((size_t)(M - 1) * (size_t)d.lda_out + (size_t)K) * 4) : 0, 0x00020000);
  const unsigned aout_row = arow < M ? (unsigned)arow * (unsigned)d.lda_out * 4u : T2_OOB;
  auto xform = [&](float4& a, const float4& z, const float4 (&x)[NV > 0 ? NV : 1], int tile, bool mine) __attribute__((always_inline)) {
    if (AXF == MSDE_RS_AXF_AFFINE) {
      a.x = fmaxf(fmaf(a.x, x[0].x, x[1].x), lbound); a.y = fmaxf(fmaf(a.y, x[0].y, x[1].y), lbound);
      a.z = fmaxf(fmaf(a.z, x[0].z, x[1].z), lbound); a.w = fmaxf(fmaf(a.w, x[0].w, x[1].w), lbound);
    } else if (AXF == MSDE_RS_AXF_BNBWD) {
      const float gx = fmaf(z.x, x[3].x, x[4].x) > 0.f ? a.x : 0.f, gy = fmaf(z.y, x[3].y, x[4].y) > 0.f ? a.y : 0.f;
      const float gz = fmaf(z.z, x[3].z, x[4].z) > 0.f ? a.z : 0.f, gw = fmaf(z.w, x[3].w, x[4].w) > 0.f ? a.w : 0.f;
      a.x = fmaf(x[0].x, gx, fmaf(x[1].x, z.x, x[2].x)); a.y = fmaf(x[0].y, gy, fmaf(x[1].y, z.y, x[2].y));
      a.z = fmaf(x[0].z, gz, fmaf(x[1].z, z.z, x[2].z)); a.w = fmaf(x[0].w, gw, fmaf(x[1].w, z.w, x[2].w));
    }
    if (AXF != MSDE_RS_AXF_NONE) {
      // write-back WITHOUT control flow (a branch here splits every K-tile block into basic blocks, and hipcc then shuffles
      // the accumulators between them: 1.5 register copies per MFMA measured): a buffer store whose per-lane offset is out
      // of range for the lanes / tiles that do not write (the range check drops them)
      const int k = 32 * tile + 16 * h + 4 * q;
      const unsigned off = (mine && k < K) ? aout_row + 4u * (unsigned)k : T2_OOB;
      t2_u32x4 v;
      v.x = __float_as_uint(a.x); v.y = __float_as_uint(a.y); v.z = __float_as_uint(a.z); v.w = __float_as_uint(a.w);
      __builtin_amdgcn_raw_buffer_store_b128(v, rsO, off, 0, 0);
    }
  };
  const bool skip_last = h == 1 && K - (nt - 1) * 32 <= 16;     // this wave's half of the last tile lies beyond K
  // One K tile: 4 RN MFMAs with, BETWEEN them (a wave issues in order: a block of other instructions in front of its MFMAs
  // leaves the matrix pipe idle for their issue time), the RN + 1 fragment reads of the next tile and this wave's requests for
  // the tile NBUF - 1 ahead.  The scheduling barriers pin that order.  The two waves of a SIMD (w and w + 4: the k-halves of
  // one row block) run the same block between the same barriers; so that they do not reach their expensive instructions
  // together (an LDS-DMA request costs its wave ~70 issue cycles, measured: 230 cycles per tile were exposed), the k-half 0
  // wave reads its fragments first and issues its requests in the second half of the block, the k-half 1 wave the other way
  // round (HALF).
  int tmod = 0;                                                  // tile mod splits (whose turn it is to write A_out)
  auto tile_block = [&](auto work_, auto gen_, auto half_, auto cur_, int t, int stage) __attribute__((always_inline)) {
    constexpr int cur = decltype(cur_)::value;                   // fragment set of tile t (compile time: registers, not scratch)
    float4& a = fa[cur];
    const float4 (&b)[RN] = fb[cur];
    float4& an = fa[cur ^ 1];
    float4 (&bn)[RN] = fb[cur ^ 1];
    constexpr bool work = decltype(work_)::value;
    constexpr int HALF = decltype(half_)::value;
    constexpr int TOT = 4 * RN;
    const int nstage = stage + 1 == NBUF ? 0 : stage + 1;
    const int istage = stage == 0 ? NBUF - 1 : stage - 1;        // stage of tile t - 1 = of tile t + NBUF - 1
    const unsigned char* base = t2_smem + nstage * STS + lo;
    __builtin_amdgcn_sched_barrier(0);
    if (!(abl & 2)) { if (fullw) t2_wait_vm<(NBUF - 3) * PW>(); else t2_wait_vm<(NBUF - 3) * (PW - 1)>(); }
    if (!(abl & 1)) t2_barrier();                                // everybody's pieces of tile t + 1 are there; tile t - 1 is free
    if (AXF != MSDE_RS_AXF_NONE) {
      xform(a, fz[cur], fx[cur], t, tmod == split);
      tmod = tmod + 1 == S ? 0 : tmod + 1;
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int s = 0; s < TOT; ++s) {
      const int j = s / RN, c = s % RN;
      if (work && !(abl & 16)) t2_mfma(rs_f4(a, j), rs_f4(b[c], j), acc[c][0]);
      __builtin_amdgcn_sched_barrier(0);
      // slot -> what follows this MFMA.  HALF 0: reads behind MFMAs 0 .. RN, requests spread over the rest; HALF 1: requests
      // spread over MFMAs 0 .. TOT - RN - 2, reads behind the last RN + 1.
      const int rs = HALF == 0 ? s : s - (TOT - (RN + 1));       // read index (0: A fragment, 1 .. RN: B fragments)
      const int u = HALF == 0 ? s - (RN + 1) : s;                // request slot index
      constexpr int SLOTS = TOT - (RN + 1);                      // request piece i goes behind request slot (i SLOTS) / PW: spread
      if (rs >= 0 && rs <= RN) {                                 // evenly, several per slot when there are more pieces than slots
        if (!(abl & 8)) {
          if (rs == 0) {
            an = *reinterpret_cast<const float4*>(base + wr * 2048);
            rd_aux(t + 1, nstage, fz[cur ^ 1], fx[cur ^ 1]);
          } else bn[rs - 1] = *reinterpret_cast<const float4*>(base + BOFF + (rs - 1) * 2048);
        }
      } else if (u >= 0 && u < SLOTS && !(abl & 4)) {
#pragma unroll
        for (int i = 0; i < PW; ++i)
          if ((i * SLOTS) / PW == u && (i < PW - 1 || fullw)) issue1(gen_, t + NBUF - 1, istage, i);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (work) t2_mfma_drain<RN>(acc);
  };
  auto run_tiles = [&](auto half_) __attribute__((always_inline)) {
    using Y = std::true_type;
    using NO = std::false_type;
    int stage = 0;
    int t = 0;
    for (; t + 2 + NBUF - 1 < nt; t += 2) {                      // both tiles of the trip issue interior tiles
      tile_block(Y{}, INTERIOR{}, half_, std::integral_constant<int, 0>{}, t, stage);
      stage = stage + 1 == NBUF ? 0 : stage + 1;
      tile_block(Y{}, INTERIOR{}, half_, std::integral_constant<int, 1>{}, t + 1, stage);
      stage = stage + 1 == NBUF ? 0 : stage + 1;
    }
    for (; t + 2 < nt; t += 2) {
      tile_block(Y{}, GENERIC{}, half_, std::integral_constant<int, 0>{}, t, stage);
      stage = stage + 1 == NBUF ? 0 : stage + 1;
      tile_block(Y{}, GENERIC{}, half_, std::integral_constant<int, 1>{}, t + 1, stage);
      stage = stage + 1 == NBUF ? 0 : stage + 1;
    }
    if (t + 2 == nt) {                                           // two tiles left
      tile_block(Y{}, GENERIC{}, half_, std::integral_constant<int, 0>{}, t, stage);
      stage = stage + 1 == NBUF ? 0 : stage + 1;
      if (skip_last) tile_block(NO{}, GENERIC{}, half_, std::integral_constant<int, 1>{}, t + 1, stage);
      else tile_block(Y{}, GENERIC{}, half_, std::integral_constant<int, 1>{}, t + 1, stage);
    } else {                                                     // one
      if (skip_last) tile_block(NO{}, GENERIC{}, half_, std::integral_constant<int, 0>{}, t, stage);
      else tile_block(Y{}, GENERIC{}, half_, std::integral_constant<int, 0>{}, t, stage);
    }
  };
  if (h == 0) run_tiles(std::integral_constant<int, 0>{});
  else run_tiles(std::integral_constant<int, 1>{});
  T2_STAMP(2);
  // The epilogue's share of the descriptor is read from the kernel-argument segment HERE, right behind the K loop (its barriers
  // are memory clobbers, so these scalar loads cannot be hoisted above it; their latency passes under the exchange below): held in scalar registers from the kernel entry, its ~50 words pushed
  // the K loop over the scalar register file (request operands of the LDS-DMA statements spilled: no longer uniform).
  msde_rs_desc de;
  {
    typedef const __attribute__((address_space(4))) msde_rs_desc* kargp;
    kargp kp = (kargp)__builtin_amdgcn_kernarg_segment_ptr();
    de.bias = nullptr;                                           // (added above)
    de.C = kp->C; de.Z = kp->Z; de.R = kp->R; de.Res = kp->Res; de.stats = kp->stats; de.stats_z = kp->stats_z;
    de.stats_mean = kp->stats_mean; de.m_valid = kp->m_valid;
    de.M = kp->M; de.N = kp->N; de.K = kp->K;
    de.ldc = kp->ldc; de.ldz = kp->ldz; de.ldr = kp->ldr; de.ldres = kp->ldres; de.ld_sz = kp->ld_sz;
    de.act = kp->act; de.epi = kp->epi; de.flags = kp->flags; de.stats_mode = kp->stats_mode;
  }
  t2_wait_vm<0>();
  t2_barrier();                                                  // every wave is done with the ring: it becomes the exchange area
  // partial sums of the k-half 1 waves -> their k-half 0 partners (same rows), through LDS
  {
    unsigned char* xb = t2_smem + (size_t)wr * RN * 1024 + lane * 16;
    if (h == 1) {
#pragma unroll
      for (int t = 0; t < RN; ++t)
        *reinterpret_cast<float4*>(xb + t * 1024) = make_float4(acc[t][0][0], acc[t][0][1], acc[t][0][2], acc[t][0][3]);
    }
    __syncthreads();
    if (h == 1) return;
#pragma unroll
    for (int t = 0; t < RN; ++t) {
      const float4 o = *reinterpret_cast<const float4*>(xb + t * 1024);
      acc[t][0][0] += o.x + bv[t]; acc[t][0][1] += o.y + bv[t]; acc[t][0][2] += o.z + bv[t]; acc[t][0][3] += o.w + bv[t];
    }
  }
  t2_epilogue<RN>(de, acc, n0, m0 + 16 * wr, rowblk * 4 + wr);
#ifdef T2_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  T2_STAMP(3);
  if (threadIdx.x == 0 && d.xf4) t2_dbg[7] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
}


// ---- launch of gemm_t2_kernel<RN, AXF> for a run-time RN; one explicit instantiation per AXF and translation unit (the three
// compile in parallel: gemm_t2.hip, gemm_t2_a1.hip, gemm_t2_a2.hip)
template <typename KERN>
static int t2_launch(KERN kern, dim3 grid, dim3 block, size_t lds, hipStream_t st, const msde_rs_desc& d) {
  if (lds > 64 * 1024) {
    static std::map<const void*, size_t> granted;
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    size_t& gr = granted[reinterpret_cast<const void*>(kern)];
    if (lds > gr) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return (int)e;
      gr = lds;
    }
  }
  MSDE_LAUNCH(kern, grid, block, lds, st, d);
  MSDE_CHECK_LAUNCH();
  return 0;
}

// bytes of dynamic LDS of gemm_t2_kernel<rn, axf> for a reduction length K
static inline size_t t2_lds_bytes(int rn, int axf, int K) {
  const int na = axf == MSDE_RS_AXF_BNBWD ? 2 : 1, nv = axf == MSDE_RS_AXF_BNBWD ? 5 : (axf == MSDE_RS_AXF_AFFINE ? 2 : 0);
  return (size_t)T2_NBUF * ((size_t)(64 * na + 16 * rn) * 128 + 1024) + (size_t)nv * (size_t)((K + 31) / 32 * 32) * 4;
}

template <int AXF>
int t2_launch_rn(int rn, dim3 grid, size_t lds, hipStream_t st, const msde_rs_desc& d) {
#define T2_RN(RN_) case RN_: return t2_launch(gemm_t2_kernel<RN_, AXF>, grid, dim3(512), lds, st, d);
  switch (rn) {
#ifdef T2_PROBE
    T2_RN(T2_PROBE)
#else
    T2_RN(1) T2_RN(2) T2_RN(3) T2_RN(4) T2_RN(5) T2_RN(6) T2_RN(8) T2_RN(10)
    case 12: if constexpr (AXF == MSDE_RS_AXF_NONE) return t2_launch(gemm_t2_kernel<12, AXF>, grid, dim3(512), lds, st, d); else return MSDE_EUNSUP;
#endif
    default: return MSDE_EUNSUP;
  }
#undef T2_RN
}
