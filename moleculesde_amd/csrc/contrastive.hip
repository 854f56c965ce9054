// contrastive.hip — node-level EBM-NCE contrastive loss between the 2D and 3D node representations,
// both directions, in two launches forward and one backward (examples/util.py:52-68,76-79:
// do_CL('EBM_node_dot_prod') + dual_CL; ~50 eager ops in the reference).
//   p_i  = <X_i, Y_i> / T                 (positive pair, shared by both directions)
//   n1_i = <X_i, Y_perm1(i)> / T          (negatives of direction X->Y)
//   n2_i = <Y_i, X_perm2(i)> / T          (negatives of direction Y->X)
//   loss = [ mean BCE(p,1) + mean BCE(n1,0) + mean BCE(p,1) + mean BCE(n2,0) ] / 2
//   acc  = [ (#{p>0} + #{n1<0}) / 2N + (#{p>0} + #{n2<0}) / 2N ] / 2
#include "msde_common.h"

__device__ __forceinline__ float bce_logits(float x, float z) {
  // torch.nn.BCEWithLogitsLoss element: max(x,0) - x*z + log(1 + exp(-|x|))
  return fmaxf(x, 0.f) - x * z + log1pf(expf(-fabsf(x)));
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

template <int V>
__global__ void cl_rows_kernel(const float* __restrict__ X, const float* __restrict__ Y, const int* __restrict__ perm1,
                               const int* __restrict__ perm2, int N, const int* __restrict__ Ndev, int cols, int tpr,
                               float invT, float* __restrict__ rows, int* __restrict__ inv1, int* __restrict__ inv2) {
  N = msde_true_rows(N, Ndev);
  using T = typename VecT<V>::type;
  int rpb = blockDim.x / tpr;
  int i = blockIdx.x * rpb + threadIdx.x / tpr;
  int lane = threadIdx.x % tpr;
  if (i >= N) return;
  int j1 = perm1[i], j2 = perm2[i];
  const T* x = reinterpret_cast<const T*>(X) + (size_t)i * cols;
  const T* y = reinterpret_cast<const T*>(Y) + (size_t)i * cols;
  const T* y1 = reinterpret_cast<const T*>(Y) + (size_t)j1 * cols;
  const T* x2 = reinterpret_cast<const T*>(X) + (size_t)j2 * cols;
  float p = 0.f, n1 = 0.f, n2 = 0.f;
  for (int c = lane; c < cols; c += tpr) {
    T xv = x[c], yv = y[c];
    p += vhsum(vmul(xv, yv));
    n1 += vhsum(vmul(xv, y1[c]));
    n2 += vhsum(vmul(yv, x2[c]));
  }
  p = group_sum(p, tpr); n1 = group_sum(n1, tpr); n2 = group_sum(n2, tpr);
  if (lane == 0) {
    rows[3 * (size_t)i] = p * invT;
    rows[3 * (size_t)i + 1] = n1 * invT;
    rows[3 * (size_t)i + 2] = n2 * invT;
    inv1[j1] = i;      // permutations: unique writes
    inv2[j2] = i;
  }
}

__global__ void __launch_bounds__(1024) cl_reduce_kernel(const float* __restrict__ rows, int N,
                                                        const int* __restrict__ Ndev, float* __restrict__ out) {
  __shared__ float s_l[16], s_a[16];
  N = msde_true_rows(N, Ndev);
  float l = 0.f, a = 0.f;
  for (int i = threadIdx.x; i < N; i += 1024) {
    float p = rows[3 * (size_t)i], n1 = rows[3 * (size_t)i + 1], n2 = rows[3 * (size_t)i + 2];
    l += 2.f * bce_logits(p, 1.f) + bce_logits(n1, 0.f) + bce_logits(n2, 0.f);
    a += 2.f * (p > 0.f ? 1.f : 0.f) + (n1 < 0.f ? 1.f : 0.f) + (n2 < 0.f ? 1.f : 0.f);
  }
  l = group_sum(l, 64); a = group_sum(a, 64);
  if ((threadIdx.x & 63) == 0) { s_l[threadIdx.x >> 6] = l; s_a[threadIdx.x >> 6] = a; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float L = 0.f, A = 0.f;
    for (int w = 0; w < 16; ++w) { L += s_l[w]; A += s_a[w]; }
    out[0] = L / (2.f * (float)N);        // (l1 + l2) / 2 with each mean over N
    out[1] = A / (4.f * (float)N);        // (acc1 + acc2) / 2 with each over 2N
  }
}

template <int V>
__global__ void cl_bwd_kernel(const float* __restrict__ X, const float* __restrict__ Y, const int* __restrict__ perm1,
                              const int* __restrict__ perm2, const int* __restrict__ inv1, const int* __restrict__ inv2,
                              const float* __restrict__ rows, const float* __restrict__ g_loss, int N,
                              const int* __restrict__ Ndev, int cols, int tpr, float invT, float* __restrict__ gX,
                              float* __restrict__ gY) {
  using T = typename VecT<V>::type;
  int rpb = blockDim.x / tpr;
  int i = blockIdx.x * rpb + threadIdx.x / tpr;
  int lane = threadIdx.x % tpr;
  if (i >= N) return;
  const int Ncap = N;
  N = msde_true_rows(N, Ndev);
  if (i >= N) {                         // rows past the valid ones carry a ZERO gradient (never stale memory)
    if (i < Ncap) {
      T* zx = reinterpret_cast<T*>(gX) + (size_t)i * cols;
      T* zy = reinterpret_cast<T*>(gY) + (size_t)i * cols;
      for (int c = lane; c < cols; c += tpr) { zx[c] = vzero<V>(); zy[c] = vzero<V>(); }
    }
    return;
  }
  const float g = g_loss[0] * invT / (float)N;
  int j1 = perm1[i], j2 = perm2[i], k1 = inv1[i], k2 = inv2[i];
  float dp = -g * sigmoidf_(-rows[3 * (size_t)i]);                 // d/dp [2 * BCE(p,1)] / 2N
  float dn1 = 0.5f * g * sigmoidf_(rows[3 * (size_t)i + 1]);
  float dn2 = 0.5f * g * sigmoidf_(rows[3 * (size_t)i + 2]);
  float dn1k = 0.5f * g * sigmoidf_(rows[3 * (size_t)k1 + 1]);     // row k1 used Y_i as its negative
  float dn2k = 0.5f * g * sigmoidf_(rows[3 * (size_t)k2 + 2]);     // row k2 used X_i as its negative
  const T* Xv = reinterpret_cast<const T*>(X);
  const T* Yv = reinterpret_cast<const T*>(Y);
  T* gXv = reinterpret_cast<T*>(gX) + (size_t)i * cols;
  T* gYv = reinterpret_cast<T*>(gY) + (size_t)i * cols;
  for (int c = lane; c < cols; c += tpr) {
    T xi = Xv[(size_t)i * cols + c], yi = Yv[(size_t)i * cols + c];
    T gx = vscale(yi, dp);
    gx = vadd(gx, vscale(Yv[(size_t)j1 * cols + c], dn1));
    gx = vadd(gx, vscale(Yv[(size_t)k2 * cols + c], dn2k));
    T gy = vscale(xi, dp);
    gy = vadd(gy, vscale(Xv[(size_t)j2 * cols + c], dn2));
    gy = vadd(gy, vscale(Xv[(size_t)k1 * cols + c], dn1k));
    gXv[c] = gx;
    gYv[c] = gy;
  }
}

extern "C" int msde_cl_ebm_fwd(const float* X, const float* Y, const int* perm1, const int* perm2, int N, int D,
                               float invT, float* rows, int* inv1, int* inv2, float* out, const int* rows_dev,
                               void* stream) {
  if (N <= 0 || D <= 0 || !X || !Y || !perm1 || !perm2 || !rows || !inv1 || !inv2 || !out) return MSDE_EINVAL;
  const int* ndev = rows_dev;
  LAUNCH_ROWS(cl_rows_kernel, N, D, X, Y, perm1, perm2, N, ndev, cols, tpr, invT, rows, inv1, inv2);
  MSDE_CHECK_LAUNCH();
  MSDE_LAUNCH(cl_reduce_kernel, dim3(1), dim3(1024), 0, as_stream(stream), (const float*)rows, N, ndev, out);
  MSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int msde_cl_ebm_bwd(const float* X, const float* Y, const int* perm1, const int* perm2, const int* inv1,
                               const int* inv2, const float* rows, const float* g_loss, int N, int D, float invT,
                               float* gX, float* gY, const int* rows_dev, void* stream) {
  if (N <= 0 || D <= 0 || !X || !Y || !perm1 || !perm2 || !inv1 || !inv2 || !rows || !g_loss || !gX || !gY)
    return MSDE_EINVAL;
  LAUNCH_ROWS(cl_bwd_kernel, N, D, X, Y, perm1, perm2, inv1, inv2, rows, g_loss, N, rows_dev, cols, tpr, invT,
              gX, gY);
  MSDE_CHECK_LAUNCH();
  return 0;
}
