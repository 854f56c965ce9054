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
#include "gemm_rs_epi.h"
#include <type_traits>

typedef int t2_i32x4 __attribute__((ext_vector_type(4)));
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
template <int RN, int ABL = 0>
__global__ void __launch_bounds__(512)
gemm_t2_kernel(const msde_rs_desc d) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char t2_smem[];
  static_assert(RN >= 1 && RN <= 12, "geometry");
  constexpr int NBUF = 4;
  constexpr int BM = 64, BN = 16 * RN;
  constexpr int AP = BM / 8, BP = BN / 8, P = AP + BP;          // 1 KiB pieces of the A / B tile of one stage
  constexpr int ST = (BM + BN) * 128;                           // bytes per stage
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
  unsigned pl[PW];                       // LDS byte offset of the piece inside a stage (wave-uniform)
  {
    const int ktail = (nt - 1) * 32;
#pragma unroll
    for (int i = 0; i < PW; ++i) {
      const int p = wave + 8 * i;
      const int c = (lane & 7) ^ (((p & 1) << 2) | (lane >> 4)); // source chunk that lands in slot lane & 7 of its row
      unsigned o = T2_OOB;
      if (i == 0) {
        const int row = m0 + 8 * p + (lane >> 3);
        if (row < M) o = ((unsigned)row * (unsigned)d.lda + 4u * (unsigned)c) * 4u;
      } else if (p < P) {
        const int rho = 8 * (p - AP) + (lane >> 3);
        const int n = n0 + t2_col<RN>(rho >> 4, rho & 15);
        if (n < N) o = ((unsigned)n * (unsigned)d.ldb + 4u * (unsigned)c) * 4u;
      }
      vo[i] = o;
      vt[i] = (ktail + 4 * c < K) ? o : T2_OOB;
      pl[i] = p < P ? (unsigned)p * 1024u : (unsigned)ST;
    }
  }
  const t2_i32x4 rsA = t2_rsrc(d.A, (unsigned)(((size_t)(M - 1) * (size_t)d.lda + (size_t)K) * 4));
  const t2_i32x4 rsB = t2_rsrc(d.B, (unsigned)(((size_t)(N - 1) * (size_t)d.ldb + (size_t)K) * 4));
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
  auto issue1 = [&](auto gen_, int tile, int stage, int i) {
    constexpr bool GEN = decltype(gen_)::value;
    const bool live = !GEN || tile < nt, last = GEN && tile == nt - 1;
    unsigned v = vo[i];
    if (GEN) v = live ? (last ? vt[i] : vo[i]) : T2_OOB;
    const unsigned la = lds0 + (unsigned)stage * (unsigned)STS + (live ? pl[i] : (unsigned)ST);
    const unsigned so = live ? (unsigned)tile * 128u : 0u;
    if (i == 0) t2_glds(v, rsA, so, la);
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
  t2_barrier();                                                  // tile 0 has landed for every wave
  T2_STAMP(1);
  float4 fa[2], fb[2][RN];                                       // fragments of this wave's k-half: two sets (tile parity)
  {
    const unsigned char* base = t2_smem + lo;
    fa[0] = *reinterpret_cast<const float4*>(base + wr * 2048);
#pragma unroll
    for (int t = 0; t < RN; ++t) fb[0][t] = *reinterpret_cast<const float4*>(base + AP * 1024 + t * 2048);
  }
  const bool skip_last = h == 1 && K - (nt - 1) * 32 <= 16;     // this wave's half of the last tile lies beyond K
  // One K tile: 4 RN MFMAs with, BETWEEN them (a wave issues in order: a block of other instructions in front of its MFMAs
  // leaves the matrix pipe idle for their issue time), the RN + 1 fragment reads of the next tile and this wave's requests for
  // the tile NBUF - 1 ahead.  The scheduling barriers pin that order.  The two waves of a SIMD (w and w + 4: the k-halves of
  // one row block) run the same block between the same barriers; so that they do not reach their expensive instructions
  // together (an LDS-DMA request costs its wave ~70 issue cycles, measured: 230 cycles per tile were exposed), the k-half 0
  // wave reads its fragments first and issues its requests in the second half of the block, the k-half 1 wave the other way
  // round (HALF).
  auto tile_block = [&](auto work_, auto gen_, auto half_, int t, int stage, const float4& a, const float4 (&b)[RN], float4& an,
                        float4 (&bn)[RN]) {
    constexpr bool work = decltype(work_)::value;
    constexpr int HALF = decltype(half_)::value;
    constexpr int TOT = 4 * RN;
    const int nstage = stage + 1 == NBUF ? 0 : stage + 1;
    const int istage = stage == 0 ? NBUF - 1 : stage - 1;        // stage of tile t - 1 = of tile t + NBUF - 1
    const unsigned char* base = t2_smem + nstage * STS + lo;
    __builtin_amdgcn_sched_barrier(0);
    if (!(abl & 2)) { if (fullw) t2_wait_vm<(NBUF - 3) * PW>(); else t2_wait_vm<(NBUF - 3) * (PW - 1)>(); }
    if (!(abl & 1)) t2_barrier();                                // everybody's pieces of tile t + 1 are there; tile t - 1 is free
#pragma unroll
    for (int s = 0; s < TOT; ++s) {
      const int j = s / RN, c = s % RN;
      if (work && !(abl & 16)) t2_mfma(rs_f4(a, j), rs_f4(b[c], j), acc[c][0]);
      __builtin_amdgcn_sched_barrier(0);
      // slot -> what follows this MFMA.  HALF 0: reads behind MFMAs 0 .. RN, requests spread over the rest; HALF 1: requests
      // spread over MFMAs 0 .. TOT - RN - 2, reads behind the last RN + 1.
      const int rs = HALF == 0 ? s : s - (TOT - (RN + 1));       // read index (0: A fragment, 1 .. RN: B fragments)
      const int u = HALF == 0 ? s - (RN + 1) : s;                // request slot index
      constexpr int SLOTS = TOT - (RN + 1);
      constexpr int GAP = SLOTS / PW > 0 ? SLOTS / PW : 1;
      if (rs >= 0 && rs <= RN) {
        if (!(abl & 8)) {
          if (rs == 0) an = *reinterpret_cast<const float4*>(base + wr * 2048);
          else bn[rs - 1] = *reinterpret_cast<const float4*>(base + AP * 1024 + (rs - 1) * 2048);
        }
      } else if (u >= 0 && u < SLOTS && u % GAP == 0 && u / GAP < PW && !(abl & 4)) {
        const int i = u / GAP;
        if (i < PW - 1 || fullw) issue1(gen_, t + NBUF - 1, istage, i);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (work) t2_mfma_drain<RN>(acc);
  };
  auto run_tiles = [&](auto half_) {
    using Y = std::true_type;
    using NO = std::false_type;
    int stage = 0;
    int t = 0;
    for (; t + 2 + NBUF - 1 < nt; t += 2) {                      // both tiles of the trip issue interior tiles
      tile_block(Y{}, INTERIOR{}, half_, t, stage, fa[0], fb[0], fa[1], fb[1]);
      stage = stage + 1 == NBUF ? 0 : stage + 1;
      tile_block(Y{}, INTERIOR{}, half_, t + 1, stage, fa[1], fb[1], fa[0], fb[0]);
      stage = stage + 1 == NBUF ? 0 : stage + 1;
    }
    for (; t + 2 < nt; t += 2) {
      tile_block(Y{}, GENERIC{}, half_, t, stage, fa[0], fb[0], fa[1], fb[1]);
      stage = stage + 1 == NBUF ? 0 : stage + 1;
      tile_block(Y{}, GENERIC{}, half_, t + 1, stage, fa[1], fb[1], fa[0], fb[0]);
      stage = stage + 1 == NBUF ? 0 : stage + 1;
    }
    if (t + 2 == nt) {                                           // two tiles left
      tile_block(Y{}, GENERIC{}, half_, t, stage, fa[0], fb[0], fa[1], fb[1]);
      stage = stage + 1 == NBUF ? 0 : stage + 1;
      if (skip_last) tile_block(NO{}, GENERIC{}, half_, t + 1, stage, fa[1], fb[1], fa[0], fb[0]);
      else tile_block(Y{}, GENERIC{}, half_, t + 1, stage, fa[1], fb[1], fa[0], fb[0]);
    } else {                                                     // one
      if (skip_last) tile_block(NO{}, GENERIC{}, half_, t, stage, fa[0], fb[0], fa[1], fb[1]);
      else tile_block(Y{}, GENERIC{}, half_, t, stage, fa[0], fb[0], fa[1], fb[1]);
    }
  };
  if (h == 0) run_tiles(std::integral_constant<int, 0>{});
  else run_tiles(std::integral_constant<int, 1>{});
  T2_STAMP(2);
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
  msde_rs_desc de = d;
  de.bias = nullptr;                                             // (added above)
  t2_epilogue<RN>(de, acc, n0, m0 + 16 * wr, rowblk * 4 + wr);
#ifdef T2_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  T2_STAMP(3);
  if (threadIdx.x == 0 && d.xf4) t2_dbg[7] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
}

// ---- host side ------------------------------------------------------------------------------------------------------
static inline bool t2_al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

struct t2_cfg { int rn, splits; };

// geometry for (M, N, K): 64-row tiles x as many column splits as give every CU about one workgroup; `splits` != 0 forces
// the split count (measurements).
static const int T2_RN_OK[] = {1, 2, 3, 4, 5, 6, 8, 10, 12};
static bool t2_pick(int M, int N, int K, int splits, t2_cfg* c) {
  if (M < 512 || N % 4 || K % 4 || N < 32) return false;
  const int ntiles = (N + 15) / 16;
  const int cus = msde_num_cus();
  const int rowblks = (M + 63) / 64;
  int S = splits > 0 ? splits : (cus + rowblks / 2) / rowblks;
  if (S < 1) S = 1;
  if (S > ntiles) S = ntiles;
  int rn = (ntiles + S - 1) / S;
  for (int ok : T2_RN_OK) if (ok >= rn) { rn = ok; break; }
  if (rn > 12) rn = 12;
  S = (ntiles + rn - 1) / rn;
  c->rn = rn; c->splits = S;
  return true;
}

extern "C" int msde_gemm_t2_supported(int M, int N, int K) {
  t2_cfg c;
  return t2_pick(M, N, K, 0, &c) ? 1 : 0;
}

extern "C" int msde_gemm_t2_geometry(int M, int N, int K, int* strips, int* strip_rows) {
  t2_cfg c;
  if (!strips || !strip_rows || !t2_pick(M, N, K, 0, &c)) return MSDE_EINVAL;
  *strip_rows = 16;
  *strips = ((M + 63) / 64) * 4;
  return 0;
}

#include <map>
#include <mutex>
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

template <int RN>
static int t2_go(dim3 grid, hipStream_t st, const msde_rs_desc& d) {
  const size_t lds = (size_t)4 * ((size_t)(64 + 16 * RN) * 128 + 1024);
#ifdef T2_TIMING
  if (RN == 5 || RN == 10) {
    constexpr int R = (RN == 5 || RN == 10) ? RN : 5;
#define T2_ABL(A_) case A_: return t2_launch(gemm_t2_kernel<R, A_>, grid, dim3(512), lds, st, d);
    switch (d.rt) {
      T2_ABL(1) T2_ABL(2) T2_ABL(3) T2_ABL(4) T2_ABL(7) T2_ABL(8) T2_ABL(15) T2_ABL(16) T2_ABL(24) T2_ABL(31) T2_ABL(12)
      default: break;
    }
#undef T2_ABL
  }
#endif
  return t2_launch(gemm_t2_kernel<RN>, grid, dim3(512), lds, st, d);
}

// C[M,N] = epilogue(A[M,K] . B[N,K]^T + bias): B is read as [N][K] (row stride ldb).  Same descriptor and epilogue
// semantics as msde_gemm_rs (include/msde_hip.h), MSDE_RS_AXF_NONE only; MSDE_EUNSUP for shapes this kernel does not take.
extern "C" int msde_gemm_t2(const msde_rs_desc* desc, void* stream) {
  if (!desc) return MSDE_EINVAL;
  msde_rs_desc d = *desc;
  if (d.M < 0 || d.N <= 0 || d.K <= 0 || !d.A || !d.B || !d.C) return MSDE_EINVAL;
  if (d.M == 0) return 0;
  if (d.flags & MSDE_GEMM_B_KMAJOR) return MSDE_EUNSUP;
  if (d.axf != MSDE_RS_AXF_NONE || d.A_out) return MSDE_EUNSUP;
  if (d.K % 4 || d.lda % 4 || d.ldb % 4 || !t2_al16(d.A) || !t2_al16(d.B)) return MSDE_EUNSUP;
  if ((size_t)d.M * (size_t)d.lda * 4 >= (1ull << 31) || (size_t)d.N * (size_t)d.ldb * 4 >= (1ull << 31)) return MSDE_EUNSUP;
  if (d.epi == MSDE_EPI_DACT && d.act != MSDE_ACT_NONE && !d.R) return MSDE_EINVAL;
  if (d.stats && d.stats_mode == MSDE_RS_STATS_BNBWD && (!d.stats_z || !d.stats_mean)) return MSDE_EINVAL;
  t2_cfg c;
  if (!t2_pick(d.M, d.N, d.K, d.splits, &c)) return MSDE_EUNSUP;
  d.splits = c.splits;
#ifndef T2_TIMING
  d.rt = 1;
#endif
  d.flags &= ~MSDE_RS_VEC_STORE;
  auto rows_ok = [](const void* p, int ldx) { return !p || (ldx % 4 == 0 && t2_al16(p)); };
  if (rows_ok(d.C, d.ldc) && rows_ok(d.Res, d.ldres) && rows_ok(d.R, d.ldr) && rows_ok(d.Z, d.ldz) &&
      rows_ok(d.stats_z, d.ld_sz) && rows_ok(d.stats_mean, 0) && rows_ok(d.stats, 0) && d.N % 4 == 0)
    d.flags |= MSDE_RS_VEC_STORE;
  const dim3 grid(((d.M + 63) / 64) * c.splits);
  hipStream_t st = as_stream(stream);
#define T2_RN(RN_) case RN_: return t2_go<RN_>(grid, st, d);
  switch (c.rn) {
#ifdef T2_PROBE
    T2_RN(T2_PROBE)
#else
    T2_RN(1) T2_RN(2) T2_RN(3) T2_RN(4) T2_RN(5) T2_RN(6) T2_RN(8) T2_RN(10) T2_RN(12)
#endif
    default: return MSDE_EUNSUP;
  }
#undef T2_RN
}
