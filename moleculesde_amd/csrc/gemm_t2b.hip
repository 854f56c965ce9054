// gemm_t2b.hip — EXPERIMENT, off by default (MSDE_BF16X3=1 in moleculesde_amd/hip.py): the node-level fp32 products of
// gemm_t2.h with both operands split into three bf16 terms and multiplied on the bf16 matrix pipe (v_mfma_f32_16x16x32_bf16,
// 16 x the rate of the fp32 shape), fp32 accumulate.  SURVEY §0.4 / §7.3.1 name this as the only route to the north star's
// 0.40-of-HBM forward figure (the fp32 FLOP floor of that forward, 172 us, lies above its 157 us budget); the headline stays
// exact fp32 -- this kernel exists to MEASURE what the split buys and what it costs in accuracy.
//
// Numerics.  a = a_hi + a_mid + a_lo EXACTLY, each term a bf16 (8 significant bits), by truncation: hi = bits(a) & 0xFFFF0000,
// mid = bits(a - hi) & 0xFFFF0000, lo = a - hi - mid (24 = 8 + 8 + 8 bits; the subtractions are exact).  Weights are split once
// per optimiser step into three bf16 planes (msde_transpose_multi modes 2 / 3, beside the transposed copies); activations in
// registers, per fragment.  A product keeps the six terms hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi: what is dropped is
// <= 3 * 2^-24 relative per product -- the size of fp32 rounding itself -- and the bf16 MFMA multiplies exactly and adds in
// fp32, so the result differs from the fp32 kernel by reassociation-level error only (tests: tolerances unchanged).
//
// FINDING (round 4, MI355X, ROCm 7.2): run on one HIP stream while OTHER kernels run on a second one, this kernel's own results stay
// exact and bitwise reproducible, but the kernels beside it sometimes are not: from identical inputs, captured-graph replays of
// the two-stream step returned a different value in ONE register of 16 CONSECUTIVE LANES of one wave of a co-resident kernel
// (seen in msde_dense_edge_layer_fwd, a plain VALU kernel, and in msde_gemm_rs) in roughly one replay in six.  Bisection
// (tools/bf16x3_repro.py): the same kernel with its six bf16 matrix instructions removed (-DT2B_NO_MFMA) or replaced by six
// v_mfma_f32_16x16x4_f32 (-DT2B_F32_MFMA) never does it; neither does the fp32 kernel of gemm_t2.h; NOR DOES THIS KERNEL WITH EACH
// K = 32 INSTRUCTION WRITTEN AS TWO v_mfma_f32_16x16x16_bf16 (-DT2B_K16: the gfx90a-era shape; same sums, tests green, 0 of 8
// trials against 6 of 8) -- the effect belongs to v_mfma_f32_16x16x32_bf16, the K-doubled shape new in gfx950, as hipcc 7.2
// emits it here; 32 idle issue cycles behind every one of them (-DT2B_NOPS) change nothing; allocating AGPRs
// (-DT2B_TOUCH_AGPR) does not help; no out-of-bounds global write (tools/t2b_guard.py), no stray LDS write
// (tools/t2b_canary.py), no kernel of the step reads LDS it did not write (tools/lds_poison_step.py).  Root cause not
// established; the switch therefore forces the single-stream step (pretrain.Trainer), where results are reproducible.
//
// Tiling as gemm_t2.h (64-row x 16 RN-column tiles, K tiles of 32, LDS-DMA staging, counted waits, 8 waves) with these
// differences: one K tile = ONE MFMA k-step per term, so the two waves of a SIMD split the COLUMN tiles of their row block
// instead of the k-halves (no exchange at the end); the B stage holds three 64-byte-per-row planes per column tile; the
// A fragment (8 consecutive k per lane = two 16-byte reads) is split in registers while the previous tile's MFMAs run.
#include "gemm_t2.h"

typedef __bf16 t2b_bf16x8 __attribute__((ext_vector_type(8)));

// segment s of the column tiles (groups 4 + .. + 4 [+ 2] [+ 1], as t2_epilogue) belongs to wave group s % 2
template <int RN> __host__ __device__ constexpr int t2b_owner(int c) {
  constexpr int n4 = RN / 4, rem = RN % 4;
  if (c < 4 * n4) return (c / 4) & 1;
  if (rem >= 2 && c < 4 * n4 + 2) return n4 & 1;
  return (n4 + (rem >= 2 ? 1 : 0)) & 1;
}

template <int RN, int HH>
__device__ __forceinline__ void t2b_epilogue(const msde_rs_desc& d, f32x4 (&acc)[RN][1], int n0, int m0, int strip) {
  const int n = threadIdx.x & 15;
  constexpr int n4 = RN / 4, rem = RN % 4;
  if constexpr (n4 >= 1 && t2b_owner<RN>(0) == HH) rs_epi_segment<1, RN, 4, 0>(d, acc, n0 + 4 * n, m0, strip, 16);
  if constexpr (n4 >= 2 && t2b_owner<RN>(4) == HH) rs_epi_segment<1, RN, 4, 4>(d, acc, n0 + 64 + 4 * n, m0, strip, 16);
  if constexpr (n4 >= 3 && t2b_owner<RN>(8) == HH) rs_epi_segment<1, RN, 4, 8>(d, acc, n0 + 128 + 4 * n, m0, strip, 16);
  if constexpr (rem >= 2 && t2b_owner<RN>(4 * n4) == HH) rs_epi_segment<1, RN, 2, 4 * n4>(d, acc, n0 + 64 * n4 + 2 * n, m0, strip, 16);
  if constexpr ((rem & 1) && t2b_owner<RN>(RN - 1) == HH) rs_epi_segment<1, RN, 1, RN - 1>(d, acc, n0 + 16 * (RN - 1) + n, m0, strip, 16);
}

struct t2b_split { unsigned hi[4], mid[4], lo[4]; };     // 8 bf16 each, packed two per dword (element 2 p in the low half)

__device__ __forceinline__ void t2b_split8(const float4& a0, const float4& a1, t2b_split& s) {
  const float a[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const unsigned u0 = __float_as_uint(a[2 * p]), u1 = __float_as_uint(a[2 * p + 1]);
    const unsigned h0 = u0 & 0xFFFF0000u, h1 = u1 & 0xFFFF0000u;
    const float r0 = a[2 * p] - __uint_as_float(h0), r1 = a[2 * p + 1] - __uint_as_float(h1);
    const unsigned m0 = __float_as_uint(r0) & 0xFFFF0000u, m1 = __float_as_uint(r1) & 0xFFFF0000u;
    const float q0 = r0 - __uint_as_float(m0), q1 = r1 - __uint_as_float(m1);
    s.hi[p] = (h0 >> 16) | h1;
    s.mid[p] = (m0 >> 16) | m1;
    s.lo[p] = (__float_as_uint(q0) >> 16) | (__float_as_uint(q1) & 0xFFFF0000u);
  }
}
__device__ __forceinline__ t2b_bf16x8 t2b_vec(const unsigned (&w)[4]) {
  const t2_u32x4 v = {w[0], w[1], w[2], w[3]};
  return __builtin_bit_cast(t2b_bf16x8, v);
}
__device__ __forceinline__ t2b_bf16x8 t2b_vec(const t2_u32x4& v) { return __builtin_bit_cast(t2b_bf16x8, v); }

template <int RN, int NBUF>
__global__ void __launch_bounds__(512)
gemm_t2b_kernel(const msde_rs_desc d) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char t2_smem[];
  static_assert(RN >= 1 && RN <= 12, "geometry");
  constexpr int BM = 64, BN = 16 * RN;
  constexpr int AP = 8, BP = 3 * RN, P = AP + BP;               // 1 KiB pieces of a stage: A rows, then (tile, plane) blocks
  constexpr int ST = AP * 1024 + BP * 1024;
  constexpr int STS = ST;
  constexpr int PW = (P + 7) / 8, NFULL = P - 8 * (PW - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int wr = wave & 3, hh = wave >> 2;
#ifdef T2B_TOUCH_AGPR
  asm volatile("v_accvgpr_write_b32 a0, 0\n\tv_accvgpr_write_b32 a7, 0" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");
#endif
  const int M = d.M, N = d.N, K = d.K;
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
  const unsigned plane_bytes = (unsigned)N * (unsigned)d.ldb * 2u;      // planes [3][N][ldb] bf16, rows zero-padded to ldb >= 32 nt

  unsigned vo[PW], vt[PW];
  const bool fullw = wave < NFULL;
  {
    const int ktail = (nt - 1) * 32;
#pragma unroll
    for (int i = 0; i < PW; ++i) {
      const int p = wave + 8 * i;
      unsigned o = T2_OOB, ot = T2_OOB;
      if (i == 0) {                                              // A: rows 8 p .., 16-byte chunks swizzled by (row >> 1) & 7
        const int c = (lane & 7) ^ (((p & 1) << 2) | (lane >> 4));
        const int row = m0 + 8 * p + (lane >> 3);
        if (row < M) o = ((unsigned)row * (unsigned)d.lda + 4u * (unsigned)c) * 4u;
        ot = (ktail + 4 * c < K) ? o : T2_OOB;
      } else if (p < P) {                                        // B: block b = 3 tile + plane: 16 rows x 64 B
        const int b = p - AP, tile = b / 3, plane = b - 3 * tile;
        const int row16 = lane >> 2;
        const int c = (lane & 3) ^ ((4 - (row16 >> 2)) & 3);     // chunk (8 k) that lands in slot lane & 3 of its row
        const int n = n0 + t2_col<RN>(tile, row16);
        if (n < N) o = (unsigned)plane * plane_bytes + ((unsigned)n * (unsigned)d.ldb + 8u * (unsigned)c) * 2u;
        ot = o;
      }
      vo[i] = o;
      vt[i] = ot;
    }
  }
  const t2_i32x4 rsA = t2_rsrc(d.A, (unsigned)(((size_t)(M - 1) * (size_t)d.lda + (size_t)K) * 4));
  const t2_i32x4 rsB = t2_rsrc(d.B, 3u * plane_bytes);
  float bv[RN];
#pragma unroll
  for (int t = 0; t < RN; ++t) {
    const int col = n0 + t2_col<RN>(t, lane & 15);
    bv[t] = (d.bias && col < N) ? d.bias[col] : 0.f;
  }
  // A request is issued only for a tile that exists: no dummy pieces (gemm_t2.h keeps its request count constant with
  // out-of-range requests into a spare KiB; here the tail of the kernel is short -- no exchange between the k-halves -- and a
  // workgroup could end with such a write still on its way into LDS that the next workgroup on the CU already owns).
  auto issue1 = [&](int tile, int stage, int i) __attribute__((always_inline)) {
    if (tile >= nt) return;
    const unsigned v = (tile == nt - 1) ? vt[i] : vo[i];
    const unsigned la = lds0 + (unsigned)stage * (unsigned)STS + (unsigned)wave * 1024u + 8192u * (unsigned)i;
    // A: byte offset of K tile = 128 tile; B planes: 64 tile
    if (i == 0) t2_glds(v, rsA, (unsigned)tile * 128u, la);
    else t2_glds(v, rsB, (unsigned)tile * 64u, la);
  };
#pragma unroll
  for (int s = 0; s < NBUF - 1; ++s)
#pragma unroll
    for (int i = 0; i < PW; ++i)
      if (i < PW - 1 || fullw) issue1(s, s, i);

  const int r = lane & 15, q = lane >> 4;
  const unsigned aoff = (unsigned)((wr * 16 + r) * 128 + (((2 * q) ^ ((r >> 1) & 7)) << 4));   // chunk 2 q; chunk 2 q + 1 at ^ 16
  const unsigned boff = (unsigned)(AP * 1024 + ((4 * r + (q ^ ((4 - (r >> 2)) & 3))) << 4));
  f32x4 acc[RN][1];
#pragma unroll
  for (int t = 0; t < RN; ++t) acc[t][0] = f32x4{0.f, 0.f, 0.f, 0.f};
  // tile 0 has landed when at most the requests of the tiles behind it are out (all NBUF - 2 of them exist, or everything)
  if (nt >= NBUF - 1) { if (fullw) t2_wait_vm<(NBUF - 2) * PW>(); else t2_wait_vm<(NBUF - 2) * (PW - 1)>(); }
  else t2_wait_vm<0>();
  t2_barrier();

  auto run = [&](auto hh_) __attribute__((always_inline)) {
    constexpr int HH = decltype(hh_)::value;
    float4 araw[2][2];                                           // raw A fragment of the NEXT tile (two sets by parity)
    t2_u32x4 fb[2][RN][3];                                       // B fragments: [set][tile][plane] (owned tiles only are touched)
    t2b_split as;                                                // split A fragment of the CURRENT tile
    auto rd_a = [&](int stage, float4 (&a)[2]) __attribute__((always_inline)) {
      const unsigned char* base = t2_smem + stage * STS;
      a[0] = *reinterpret_cast<const float4*>(base + aoff);
      a[1] = *reinterpret_cast<const float4*>(base + (aoff ^ 16u));
    };
    auto rd_b = [&](int stage, int c, int pl, t2_u32x4& b) __attribute__((always_inline)) {
      b = *reinterpret_cast<const t2_u32x4*>(t2_smem + stage * STS + boff + (3 * c + pl) * 1024);
    };
    {
      float4 a0[2];
      rd_a(0, a0);
#pragma unroll
      for (int c = 0; c < RN; ++c)
        if (t2b_owner<RN>(c) == HH)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) rd_b(0, c, pl, fb[0][c][pl]);
      t2b_split8(a0[0], a0[1], as);
    }
    auto tile_block = [&](auto cur_, int t, int stage) __attribute__((always_inline)) {
      constexpr int cur = decltype(cur_)::value;
      const int nstage = stage + 1 == NBUF ? 0 : stage + 1;
      const int istage = stage == 0 ? NBUF - 1 : stage - 1;
      __builtin_amdgcn_sched_barrier(0);
      // tile t + 1 has landed when at most the requests of tiles t + 2 .. t + NBUF - 2 are out; near the end: everything
      if (t + NBUF - 2 < nt) { if (fullw) t2_wait_vm<(NBUF - 3) * PW>(); else t2_wait_vm<(NBUF - 3) * (PW - 1)>(); }
      else t2_wait_vm<0>();
      t2_barrier();
      rd_a(nstage, araw[cur]);                                   // next tile's A fragment: split at the end of this block
      const t2b_bf16x8 ah = t2b_vec(as.hi), am = t2b_vec(as.mid), al = t2b_vec(as.lo);
      int piece = 0;
#pragma unroll
      for (int c = 0; c < RN; ++c) {
        if (t2b_owner<RN>(c) != HH) continue;
        const t2b_bf16x8 bh = t2b_vec(fb[cur][c][0]), bm = t2b_vec(fb[cur][c][1]), bl = t2b_vec(fb[cur][c][2]);
        f32x4 x = acc[c][0];
#if defined(T2B_F32_MFMA)
        {   // (debugging: the same number of matrix instructions, of the fp32 shape)
          const t2_u32x4 ua = __builtin_bit_cast(t2_u32x4, ah), ub = __builtin_bit_cast(t2_u32x4, bh);
          const t2_u32x4 uc = __builtin_bit_cast(t2_u32x4, am), ud = __builtin_bit_cast(t2_u32x4, bm);
          x = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, ua[0]), __builtin_bit_cast(float, ub[0]), x, 0, 0, 0);
          x = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, ua[1]), __builtin_bit_cast(float, ub[1]), x, 0, 0, 0);
          rd_b(nstage, c, 0, fb[cur ^ 1][c][0]);
          x = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, ua[2]), __builtin_bit_cast(float, ub[2]), x, 0, 0, 0);
          x = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, ua[3]), __builtin_bit_cast(float, ub[3]), x, 0, 0, 0);
          rd_b(nstage, c, 1, fb[cur ^ 1][c][1]);
          x = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, uc[0]), __builtin_bit_cast(float, ud[0]), x, 0, 0, 0);
          x = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, uc[1]), __builtin_bit_cast(float, ud[1]), x, 0, 0, 0);
        }
#elif defined(T2B_NO_MFMA)
        x[0] += __builtin_bit_cast(float, __builtin_bit_cast(t2_u32x4, al)[0] ^ __builtin_bit_cast(t2_u32x4, bh)[1]) * 1e-30f;
        x[1] += __builtin_bit_cast(float, __builtin_bit_cast(t2_u32x4, am)[0] ^ __builtin_bit_cast(t2_u32x4, bm)[1]) * 1e-30f;
        x[2] += __builtin_bit_cast(float, __builtin_bit_cast(t2_u32x4, ah)[0] ^ __builtin_bit_cast(t2_u32x4, bl)[1]) * 1e-30f;
        rd_b(nstage, c, 0, fb[cur ^ 1][c][0]);
        rd_b(nstage, c, 1, fb[cur ^ 1][c][1]);
#elif defined(T2B_K16)
        {   // (debugging the co-residency finding: each K = 32 instruction as two of the gfx90a-era K = 16 bf16 shape -- the lane's
            //  eight k values split 4 + 4 on both operands alike, so the sum is the same)
          typedef short t2b_s4 __attribute__((ext_vector_type(4)));
          auto lo4 = [](const t2b_bf16x8& v) { const t2_u32x4 u = __builtin_bit_cast(t2_u32x4, v); typedef unsigned u2 __attribute__((ext_vector_type(2))); const u2 w = {u[0], u[1]}; return __builtin_bit_cast(t2b_s4, w); };
          auto hi4 = [](const t2b_bf16x8& v) { const t2_u32x4 u = __builtin_bit_cast(t2_u32x4, v); typedef unsigned u2 __attribute__((ext_vector_type(2))); const u2 w = {u[2], u[3]}; return __builtin_bit_cast(t2b_s4, w); };
#define T2B_M16(A_, B_) x = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(lo4(A_), lo4(B_), x, 0, 0, 0); x = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(hi4(A_), hi4(B_), x, 0, 0, 0)
          T2B_M16(al, bh); T2B_M16(ah, bl);
          rd_b(nstage, c, 0, fb[cur ^ 1][c][0]);
          T2B_M16(am, bm); T2B_M16(am, bh);
          rd_b(nstage, c, 1, fb[cur ^ 1][c][1]);
          T2B_M16(ah, bm); T2B_M16(ah, bh);
#undef T2B_M16
        }
#else
#ifdef T2B_NOPS       // (debugging the co-residency finding: idle issue cycles behind every bf16 matrix instruction)
#define T2B_GAP() asm volatile("s_nop 15\n\ts_nop 15" ::: "memory")
#else
#define T2B_GAP()
#endif
        x = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, x, 0, 0, 0); T2B_GAP();
        x = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, x, 0, 0, 0); T2B_GAP();
        rd_b(nstage, c, 0, fb[cur ^ 1][c][0]);
        x = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, x, 0, 0, 0); T2B_GAP();
        x = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, x, 0, 0, 0); T2B_GAP();
        rd_b(nstage, c, 1, fb[cur ^ 1][c][1]);
        x = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, x, 0, 0, 0); T2B_GAP();
        x = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, x, 0, 0, 0); T2B_GAP();
#undef T2B_GAP
#endif
        rd_b(nstage, c, 2, fb[cur ^ 1][c][2]);
        acc[c][0] = x;
        if (piece < PW) {                                        // one request per owned tile (any left go out behind the last)
          if (piece < PW - 1 || fullw) issue1(t + NBUF - 1, istage, piece);
          ++piece;
        }
      }
#pragma unroll
      for (int i = 0; i < PW; ++i)
        if (i >= piece && (i < PW - 1 || fullw)) issue1(t + NBUF - 1, istage, i);
      t2b_split8(araw[cur][0], araw[cur][1], as);
    };
    int stage = 0, t = 0;
    for (; t + 1 < nt; t += 2) {
      tile_block(std::integral_constant<int, 0>{}, t, stage);
      stage = stage + 1 == NBUF ? 0 : stage + 1;
      tile_block(std::integral_constant<int, 1>{}, t + 1, stage);
      stage = stage + 1 == NBUF ? 0 : stage + 1;
    }
    if (t < nt) tile_block(std::integral_constant<int, 0>{}, t, stage);
    t2_wait_vm<0>();
    msde_rs_desc de;
    {
      typedef const __attribute__((address_space(4))) msde_rs_desc* kargp;
      kargp kp = (kargp)__builtin_amdgcn_kernarg_segment_ptr();
      de.bias = nullptr;
      de.C = kp->C; de.Z = kp->Z; de.R = kp->R; de.Res = kp->Res; de.stats = kp->stats; de.stats_z = kp->stats_z;
      de.stats_mean = kp->stats_mean; de.m_valid = kp->m_valid;
      de.M = kp->M; de.N = kp->N; de.K = kp->K;
      de.ldc = kp->ldc; de.ldz = kp->ldz; de.ldr = kp->ldr; de.ldres = kp->ldres; de.ld_sz = kp->ld_sz;
      de.act = kp->act; de.epi = kp->epi; de.flags = kp->flags; de.stats_mode = kp->stats_mode;
    }
#pragma unroll
    for (int c = 0; c < RN; ++c)
      if (t2b_owner<RN>(c) == HH) { acc[c][0][0] += bv[c]; acc[c][0][1] += bv[c]; acc[c][0][2] += bv[c]; acc[c][0][3] += bv[c]; }
    t2b_epilogue<RN, HH>(de, acc, n0, m0 + 16 * wr, rowblk * 4 + wr);
  };
  if (hh == 0) run(std::integral_constant<int, 0>{});
  else run(std::integral_constant<int, 1>{});
}

// ---- host side ------------------------------------------------------------------------------------------------------
static inline bool t2b_al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int RN>
static int t2b_go(dim3 grid, hipStream_t st, const msde_rs_desc& d) {
  constexpr int NBUF = RN >= 9 ? 3 : 4;
  const size_t lds = (size_t)NBUF * ((size_t)(8 + 3 * RN) * 1024);
  return t2_launch(gemm_t2b_kernel<RN, NBUF>, grid, dim3(512), lds, st, d);
}

// Same descriptor as msde_gemm_t2 (MSDE_RS_AXF_NONE only) except the weight operand: B = three bf16 planes [3][N][ldb] of the
// [N][K] weight (msde_transpose_multi modes 2 / 3), ldb = row length in bf16 elements, a multiple of 32 >= K, zero beyond K.
extern "C" int msde_gemm_t2b(const msde_rs_desc* desc, void* stream) {
  if (!desc) return MSDE_EINVAL;
  msde_rs_desc d = *desc;
  if (d.M < 0 || d.N <= 0 || d.K <= 0 || !d.A || !d.B || !d.C) return MSDE_EINVAL;
  if (d.M == 0) return 0;
  if (d.axf != MSDE_RS_AXF_NONE || d.A_out) return MSDE_EUNSUP;
  if (d.K % 4 || d.lda % 4 || d.ldb % 32 || d.ldb < (d.K + 31) / 32 * 32 || !t2b_al16(d.A) || !t2b_al16(d.B)) return MSDE_EUNSUP;
  if ((size_t)d.M * (size_t)d.lda * 4 >= (1ull << 31) || (size_t)3 * d.N * (size_t)d.ldb * 2 >= (1ull << 31)) return MSDE_EUNSUP;
  if (d.epi == MSDE_EPI_DACT && d.act != MSDE_ACT_NONE && !d.R) return MSDE_EINVAL;
  if (d.M < 512 || d.N % 4 || d.N < 32) return MSDE_EUNSUP;
  const int ntiles = (d.N + 15) / 16, cus = msde_num_cus(), rowblks = (d.M + 63) / 64;
  int S = d.splits > 0 ? d.splits : (cus + rowblks / 2) / rowblks;
  if (S < 1) S = 1;
  if (S > ntiles) S = ntiles;
  int rn = (ntiles + S - 1) / S;
  while (rn > 12) { ++S; rn = (ntiles + S - 1) / S; }
  static const int ok[] = {1, 2, 3, 4, 5, 6, 8, 10, 12};
  for (int v : ok) if (v >= rn) { rn = v; break; }
  S = (ntiles + rn - 1) / rn;
  d.splits = S;
  d.rt = 1;
  d.flags &= ~MSDE_RS_VEC_STORE;
  auto rows_ok = [](const void* p, int ldx) { return !p || (ldx % 4 == 0 && t2b_al16(p)); };
  if (rows_ok(d.C, d.ldc) && rows_ok(d.Res, d.ldres) && rows_ok(d.R, d.ldr) && rows_ok(d.Z, d.ldz) &&
      rows_ok(d.stats_z, d.ld_sz) && rows_ok(d.stats_mean, 0) && rows_ok(d.stats, 0) && d.N % 4 == 0)
    d.flags |= MSDE_RS_VEC_STORE;
  const dim3 grid(rowblks * S);
  hipStream_t st = as_stream(stream);
#define T2B_RN(RN_) case RN_: return t2b_go<RN_>(grid, st, d);
  switch (rn) {
    T2B_RN(1) T2B_RN(2) T2B_RN(3) T2B_RN(4) T2B_RN(5) T2B_RN(6) T2B_RN(8) T2B_RN(10) T2B_RN(12)
    default: return MSDE_EUNSUP;
  }
#undef T2B_RN
}
