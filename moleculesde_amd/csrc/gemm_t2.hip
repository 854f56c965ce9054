// gemm_t2.hip — host side of the 2-D tiled fp32 matrix-core GEMM for the node-level products (kernel and tiling argument:
// gemm_t2.h).  Reference call sites: Geom3D/models/molecule_gnn_model.py:17,28-29,176-182 (GIN MLP + BatchNorm),
// Geom3D/models/schnet.py:141-148,163-167 (lin1 / lin2 / lin), Geom3D/models/MoleculeSDE/SDE_model_2D_to_3D.py:264-271 --
// torch.addmm / torch.mm (+ F.batch_norm, F.relu) there.
#include "gemm_t2.h"

template int t2_launch_rn<MSDE_RS_AXF_NONE>(int, dim3, size_t, hipStream_t, const msde_rs_desc&);
extern template int t2_launch_rn<MSDE_RS_AXF_AFFINE>(int, dim3, size_t, hipStream_t, const msde_rs_desc&);
extern template int t2_launch_rn<MSDE_RS_AXF_BNBWD>(int, dim3, size_t, hipStream_t, const msde_rs_desc&);

static inline bool t2_al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

struct t2_cfg { int rn, splits; };

// geometry for (M, N, K): 64-row tiles x as many column splits as give every CU about one workgroup; `splits` != 0 forces
// the split count (measurements).
static const int T2_RN_OK[] = {1, 2, 3, 4, 5, 6, 8, 10, 12};   // (12: the plain kernel only -- with a transform it runs out of registers)
static bool t2_pick(int M, int N, int K, int splits, int axf, t2_cfg* c) {
  if (M < 512 || N % 4 || K % 4 || N < 32) return false;
  const int ntiles = (N + 15) / 16;
  const int cus = msde_num_cus();
  const int rowblks = (M + 63) / 64;
  const int rn_max = axf == MSDE_RS_AXF_NONE ? 12 : 10;
  int S = splits > 0 ? splits : (cus + rowblks / 2) / rowblks;
  if (S < 1) S = 1;
  if (S > ntiles) S = ntiles;
  int rn = (ntiles + S - 1) / S;
  while (rn > rn_max) { ++S; rn = (ntiles + S - 1) / S; }
  for (int ok : T2_RN_OK) if (ok >= rn) { rn = ok; break; }
  S = (ntiles + rn - 1) / rn;
  // Wide outputs (N = 600: 10 column tiles per workgroup) on a node-level operand: HALF the tile width and two workgroups per
  // CU instead (their rings fit the LDS twice when only one A operand is staged) -- one workgroup's prologue and epilogue
  // then overlap the other's K loop (3588 x 600 x 300 alone: 22.5 -> 19.6 us; narrower outputs lose, tools/bench_gemm_t2.py
  // --sweep).  MSDE_T2_SPLIT2=0 keeps one workgroup per CU.
  const int split2 = 2;
  bool two_per_cu = false;
  const int rn_cap = 0;
  if (rn_cap > 0 && splits <= 0 && rowblks <= cus && rn > rn_cap) {
    int rn2 = rn_cap;
    for (int ok : T2_RN_OK) if (ok >= rn2) { rn2 = ok; break; }
    const int S2 = (ntiles + rn2 - 1) / rn2;
    if (rowblks * S2 <= 2 * cus) { rn = rn2; S = S2; two_per_cu = true; }
  }
  if (split2 && splits <= 0 && rowblks <= cus && rn >= 8 && (axf != MSDE_RS_AXF_BNBWD || split2 == 2)) {
    // the narrowest tile >= half the width whose workgroups still number <= 2 per CU
    for (int ok : T2_RN_OK) {
      if (ok < (rn + 1) / 2 || ok >= rn) continue;
      const int S2 = (ntiles + ok - 1) / ok;
      if (rowblks * S2 <= 2 * cus) { rn = ok; S = S2; two_per_cu = true; break; }
    }
  }
  // a node-level operand (fewer row blocks than CUs) must fit the chip in ONE round of workgroups: a second, mostly empty
  // round costs more than the row strips of msde_gemm_rs do (N = 728 with a transform: 57 x 5 workgroups)
  if (splits <= 0 && rowblks <= cus && !two_per_cu && rowblks * S > cus + cus / 16) return false;
  c->rn = rn; c->splits = S;
  return true;
}

extern "C" int msde_gemm_t2_supported(int M, int N, int K, int axf) {
  t2_cfg c;
  return t2_pick(M, N, K, 0, axf, &c) ? 1 : 0;
}

extern "C" int msde_gemm_t2_geometry(int M, int N, int K, int* strips, int* strip_rows) {
  t2_cfg c;
  if (!strips || !strip_rows || !t2_pick(M, N, K, 0, MSDE_RS_AXF_BNBWD, &c)) return MSDE_EINVAL;
  *strip_rows = 16;
  *strips = ((M + 63) / 64) * 4;
  return 0;
}

// C[M,N] = epilogue(xf(A)[M,K] . B[N,K]^T + bias): B is read as [N][K] (row stride ldb).  Same descriptor, transform,
// epilogue and statistics semantics as msde_gemm_rs (include/msde_hip.h); MSDE_EUNSUP for shapes this kernel does not take.
extern "C" int msde_gemm_t2(const msde_rs_desc* desc, void* stream) {
  if (!desc) return MSDE_EINVAL;
  msde_rs_desc d = *desc;
  if (d.M < 0 || d.N <= 0 || d.K <= 0 || !d.A || !d.B || !d.C) return MSDE_EINVAL;
  if (d.M == 0) return 0;
  if (d.flags & MSDE_GEMM_B_KMAJOR) return MSDE_EUNSUP;
  if (d.K % 4 || d.lda % 4 || d.ldb % 4 || !t2_al16(d.A) || !t2_al16(d.B)) return MSDE_EUNSUP;
  if ((size_t)d.M * (size_t)d.lda * 4 >= (1ull << 31) || (size_t)d.N * (size_t)d.ldb * 4 >= (1ull << 31)) return MSDE_EUNSUP;
  if (d.epi == MSDE_EPI_DACT && d.act != MSDE_ACT_NONE && !d.R) return MSDE_EINVAL;
  if (d.stats && d.stats_mode == MSDE_RS_STATS_BNBWD && (!d.stats_z || !d.stats_mean)) return MSDE_EINVAL;
  if (d.axf != MSDE_RS_AXF_NONE && d.axf != MSDE_RS_AXF_AFFINE && d.axf != MSDE_RS_AXF_BNBWD) return MSDE_EINVAL;
  if (d.axf == MSDE_RS_AXF_NONE && d.A_out) return MSDE_EINVAL;
  if (d.axf != MSDE_RS_AXF_NONE && d.K > 768) return MSDE_EUNSUP;
  if (d.axf == MSDE_RS_AXF_AFFINE && (!d.xf0 || !d.xf1 || !t2_al16(d.xf0) || !t2_al16(d.xf1))) return MSDE_EINVAL;
  if (d.axf == MSDE_RS_AXF_BNBWD) {
    if (!d.A2 || !d.xf0 || !d.xf1 || !d.xf2 || d.lda2 % 4 || !t2_al16(d.A2)) return MSDE_EINVAL;
    if ((d.xf3 != nullptr) != (d.xf4 != nullptr)) return MSDE_EINVAL;
    if (!t2_al16(d.xf0) || !t2_al16(d.xf1) || !t2_al16(d.xf2) || !t2_al16(d.xf3) || !t2_al16(d.xf4)) return MSDE_EINVAL;
    if ((size_t)d.M * (size_t)d.lda2 * 4 >= (1ull << 31)) return MSDE_EUNSUP;
  }
  if (d.A_out && (d.lda_out % 4 || !t2_al16(d.A_out))) return MSDE_EINVAL;
  t2_cfg c;
  if (!t2_pick(d.M, d.N, d.K, d.splits, d.axf, &c)) return MSDE_EUNSUP;
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
  const size_t lds = t2_lds_bytes(c.rn, d.axf, d.K);
  if (lds > 160 * 1024) return MSDE_EUNSUP;
#ifdef T2_TIMING
  if (d.axf == MSDE_RS_AXF_NONE && (c.rn == 5 || c.rn == 10) && d.rt) {
#define T2_ABL(A_) case A_: return c.rn == 5 ? t2_launch(gemm_t2_kernel<5, 0, A_>, grid, dim3(512), lds, st, d) \
                                             : t2_launch(gemm_t2_kernel<10, 0, A_>, grid, dim3(512), lds, st, d);
    switch (d.rt) {
      T2_ABL(1) T2_ABL(2) T2_ABL(3) T2_ABL(4) T2_ABL(7) T2_ABL(8) T2_ABL(15) T2_ABL(16) T2_ABL(24) T2_ABL(31) T2_ABL(12)
      default: break;
    }
#undef T2_ABL
  }
#endif
  if (d.axf == MSDE_RS_AXF_AFFINE) return t2_launch_rn<MSDE_RS_AXF_AFFINE>(c.rn, grid, lds, st, d);
  if (d.axf == MSDE_RS_AXF_BNBWD) return t2_launch_rn<MSDE_RS_AXF_BNBWD>(c.rn, grid, lds, st, d);
  return t2_launch_rn<MSDE_RS_AXF_NONE>(c.rn, grid, lds, st, d);
}
