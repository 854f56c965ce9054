// gemm_t2_a1.hip — the MSDE_RS_AXF_AFFINE instantiations of gemm_t2_kernel (gemm_t2.h), a translation unit of their own so that
// the three families compile in parallel.
#include "gemm_t2.h"

template int t2_launch_rn<MSDE_RS_AXF_AFFINE>(int, dim3, size_t, hipStream_t, const msde_rs_desc&);
