// sde2d3d.hip — kernels of the 2D->3D equivariant score network (SDE_model_2D_to_3D.py,
// equivariant_scorenetwork.py): per-edge SE(3) frame + Fourier features, edge-featured multi-head
// attention with per-target softmax (PyG TransformerConv), frame mix + mean scatter.
#include "msde_common.h"

#define MSDE_EPS 1e-6f

// one thread per (edge, fourier channel c); every thread rebuilds the (cheap) frame in registers
__global__ void edge_geometry_fwd_kernel(const float* __restrict__ pos, const int* __restrict__ src,
                                         const int* __restrict__ dst, int E, const float* __restrict__ Wd,
                                         const float* __restrict__ Wc, int C, float* __restrict__ feat_d,
                                         float* __restrict__ feat_i, float* __restrict__ feat_j,
                                         int feat_ld, float* __restrict__ angle, int angle_ld, int angle_zero_off,
                                         float* __restrict__ basis) {
  const float PI_F = 3.14159265358979323846f;
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)E * C) return;
  int e = (int)(t / C), c = (int)(t % C);
  int r = src[e], q = dst[e];  // row = edge_index[0] (source), col = edge_index[1] (target)
  r = max(r, 0); q = max(q, 0); // padded edge slots (src = dst = -1 past the true edge count): any finite geometry
  float prx = pos[3 * r], pry = pos[3 * r + 1], prz = pos[3 * r + 2];
  float pcx = pos[3 * q], pcy = pos[3 * q + 1], pcz = pos[3 * q + 2];
  // coord2basis (SDE_model_2D_to_3D.py:35-47)
  float dx = prx - pcx, dy = pry - pcy, dz = prz - pcz;
  float radial = dx * dx + dy * dy + dz * dz;
  float cx = pry * pcz - prz * pcy, cy = prz * pcx - prx * pcz, cz = prx * pcy - pry * pcx;
  float dist = sqrtf(radial);
  float norm = dist + MSDE_EPS;
  dx /= norm; dy /= norm; dz /= norm;
  float cnorm = sqrtf(cx * cx + cy * cy + cz * cz) + MSDE_EPS;
  cx /= cnorm; cy /= cnorm; cz /= cnorm;
  float vx = dy * cz - dz * cy, vy = dz * cx - dx * cz, vz = dx * cy - dy * cx;
  // frame coordinates of both endpoints (:357-360)
  float ci0 = dx * prx + dy * pry + dz * prz;
  float ci1 = fabsf(cx * prx + cy * pry + cz * prz);
  float ci2 = vx * prx + vy * pry + vz * prz;
  float cj0 = dx * pcx + dy * pcy + dz * pcz;
  float cj1 = fabsf(cx * pcx + cy * pcy + cz * pcz);
  float cj2 = vx * pcx + vy * pcy + vz * pcz;
  if (c == 0) {
    float mul = ci0 * cj0 + ci1 * cj1 + ci2 * cj2;
    float ni = sqrtf(ci0 * ci0 + ci1 * ci1 + ci2 * ci2);
    float nj = sqrtf(cj0 * cj0 + cj1 * cj1 + cj2 * cj2);
    float pcos = mul / (ni + MSDE_EPS) / (nj + MSDE_EPS);
    float psin = sqrtf(1.f - pcos * pcos);
    if (angle_ld >= 4) {
      *reinterpret_cast<float4*>(angle + (size_t)angle_ld * e) = make_float4(psin, pcos, 0.f, 0.f);
      if (angle_zero_off) *reinterpret_cast<float4*>(angle + (size_t)angle_ld * e + angle_zero_off) = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      angle[2 * (size_t)e] = psin;
      angle[2 * (size_t)e + 1] = pcos;
    }
    float* b = basis + 9 * (size_t)e;
    b[0] = dx; b[1] = dy; b[2] = dz; b[3] = cx; b[4] = cy; b[5] = cz; b[6] = vx; b[7] = vy; b[8] = vz;
  }
  // GaussianFourierProjection (:57-66): x * W * 2 * pi, in that order
  float wd = Wd[c], wc = Wc[c];
  float a = ((dist * wd) * 2.f) * PI_F;
  float* fd = feat_d + (size_t)e * 2 * C;
  fd[c] = sinf(a);
  fd[C + c] = cosf(a);
  float* fi = feat_i + (size_t)e * feat_ld;
  float* fj = feat_j + (size_t)e * feat_ld;
  float a0 = ((ci0 * wc) * 2.f) * PI_F, a2 = ((ci2 * wc) * 2.f) * PI_F;
  fi[c] = sinf(a0); fi[C + c] = cosf(a0); fi[2 * C + c] = sinf(a2); fi[3 * C + c] = cosf(a2);
  float b0 = ((cj0 * wc) * 2.f) * PI_F, b2 = ((cj2 * wc) * 2.f) * PI_F;
  fj[c] = sinf(b0); fj[C + c] = cosf(b0); fj[2 * C + c] = sinf(b2); fj[3 * C + c] = cosf(b2);
}

extern "C" int msde_edge_geometry_fwd_ld(const float* pos, const int* src, const int* dst, int E, const float* Wd,
                                         const float* Wc, int C, float* feat_d, float* feat_i, float* feat_j, int feat_ld,
                                         float* angle, int angle_ld, int angle_zero_off, float* basis, void* stream) {
  if (E < 0 || C <= 0 || !pos || !src || !dst || !Wd || !Wc || !feat_d || !feat_i || !feat_j || !angle || !basis)
    return MSDE_EINVAL;
  if (feat_ld < 4 * C) return MSDE_EINVAL;
  if (angle_ld != 2 && (angle_ld < 4 || angle_ld % 4 || angle_zero_off % 4 || angle_zero_off < 0 ||
                        (reinterpret_cast<uintptr_t>(angle) & 15)))
    return MSDE_EINVAL;
  if (E == 0) return 0;
  size_t total = (size_t)E * C;
  MSDE_LAUNCH(edge_geometry_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream),
                     pos, src, dst, E, Wd, Wc, C, feat_d, feat_i, feat_j, feat_ld, angle, angle_ld, angle_zero_off, basis);
  MSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int msde_edge_geometry_fwd(const float* pos, const int* src, const int* dst, int E, const float* Wd,
                                      const float* Wc, int C, float* feat_d, float* feat_i, float* feat_j,
                                      float* angle, float* basis, void* stream) {
  return msde_edge_geometry_fwd_ld(pos, src, dst, E, Wd, Wc, C, feat_d, feat_i, feat_j, 4 * C, angle, 2, 0, basis, stream);
}

// ------------------------------------------------------------------------------------------------
// edge attention: one thread per (target node, head); CH channels per head kept in registers.
// ------------------------------------------------------------------------------------------------
#define EA_UB 4     // edges per batch

template <int CH>
__global__ void edge_attention_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                          const float* __restrict__ v, const float* __restrict__ skip, int ld,
                                          const float* __restrict__ ee, int ld_ee,
                                          const int* __restrict__ rowptr, const int* __restrict__ src, int N, int H,
                                          float p_drop, unsigned long long seed,
                                          const unsigned long long* __restrict__ seed_dev,
                                          float* __restrict__ alpha, float* __restrict__ out) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N * H) return;
  if (seed_dev) seed += seed_dev[0] * 0x100000001B3ull;
  int i = t / H, h = t % H;
  const int D = H * CH;
  float qv[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) qv[c] = q[(size_t)i * ld + h * CH + c];
  const float scale = 1.f / sqrtf((float)CH);
  int s0 = rowptr[i], s1 = rowptr[i + 1];
  // Every pass walks the in-edges in batches of EA_UB: indices, then all rows of the batch in flight (the
  // index -> row -> arithmetic chain per edge is what bounds this kernel), arithmetic in edge order.
  float m = -INFINITY;
  for (int e = s0; e < s1; e += EA_UB) {
    float kv[EA_UB][CH], ev[EA_UB][CH];
#pragma unroll
    for (int u = 0; u < EA_UB; ++u) {
      const int eu = min(e + u, s1 - 1);
      const float* kr = k + (size_t)src[eu] * ld + h * CH;
      const float* er = ee + (size_t)eu * ld_ee + h * CH;
#pragma unroll
      for (int c = 0; c < CH; ++c) { kv[u][c] = kr[c]; ev[u][c] = er[c]; }
    }
#pragma unroll
    for (int u = 0; u < EA_UB; ++u) {
      if (e + u < s1) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) s += qv[c] * (kv[u][c] + ev[u][c]);
        s *= scale;
        alpha[(size_t)(e + u) * H + h] = s;
        m = fmaxf(m, s);
      }
    }
  }
  float sum = 0.f;
  for (int e = s0; e < s1; e += EA_UB) {
    float sv[EA_UB];
#pragma unroll
    for (int u = 0; u < EA_UB; ++u) sv[u] = alpha[(size_t)min(e + u, s1 - 1) * H + h];
#pragma unroll
    for (int u = 0; u < EA_UB; ++u) {
      if (e + u < s1) {
        float p = expf(sv[u] - m);
        alpha[(size_t)(e + u) * H + h] = p;
        sum += p;
      }
    }
  }
  float inv = 1.f / (sum + 1e-16f);
  float keep_scale = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  float acc[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) acc[c] = 0.f;
  for (int e = s0; e < s1; e += EA_UB) {
    float vv[EA_UB][CH], ev[EA_UB][CH], av[EA_UB];
#pragma unroll
    for (int u = 0; u < EA_UB; ++u) {
      const int eu = min(e + u, s1 - 1);
      const float* vr = v + (size_t)src[eu] * ld + h * CH;
      const float* er = ee + (size_t)eu * ld_ee + h * CH;
      av[u] = alpha[(size_t)eu * H + h];
#pragma unroll
      for (int c = 0; c < CH; ++c) { vv[u][c] = vr[c]; ev[u][c] = er[c]; }
    }
#pragma unroll
    for (int u = 0; u < EA_UB; ++u) {
      if (e + u < s1) {
        float a = av[u] * inv;
        alpha[(size_t)(e + u) * H + h] = a;
        if (p_drop > 0.f)
          a = (msde_uniform(seed, (unsigned long long)(e + u) * H + h) >= p_drop) ? a * keep_scale : 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = fmaf(a, vv[u][c] + ev[u][c], acc[c]);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < CH; ++c)
    out[(size_t)i * D + h * CH + c] = acc[c] + (skip ? skip[(size_t)i * ld + h * CH + c] : 0.f);   // + lin_skip(x_i)
}

template <int CH>
__global__ void edge_attention_bwd_kernel(const float* __restrict__ g_out, const float* __restrict__ q,
                                          const float* __restrict__ k, const float* __restrict__ v, int ld,
                                          float* __restrict__ g_skip, int ldg,
                                          const float* __restrict__ ee, int ld_ee, const float* __restrict__ alpha,
                                          const int* __restrict__ rowptr, const int* __restrict__ src, int N, int H,
                                          float p_drop, unsigned long long seed,
                                          const unsigned long long* __restrict__ seed_dev, float* __restrict__ g_q,
                                          float* __restrict__ g_ee, float* __restrict__ g_kpe,
                                          float* __restrict__ g_vpe, int ld_kv) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N * H) return;
  if (seed_dev) seed += seed_dev[0] * 0x100000001B3ull;
  int i = t / H, h = t % H;
  const int D = H * CH;
  float qv[CH], go[CH], gq[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    qv[c] = q[(size_t)i * ld + h * CH + c];
    go[c] = g_out[(size_t)i * D + h * CH + c];
    gq[c] = 0.f;
    if (g_skip) g_skip[(size_t)i * ldg + h * CH + c] = go[c];      // d(out)/d(skip) = 1
  }
  const float scale = 1.f / sqrtf((float)CH);
  float keep_scale = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  int s0 = rowptr[i], s1 = rowptr[i + 1];
  // pass 1: g_vpe, and dsum = sum_e alpha_e * g_alpha_e   (batches of EA_UB edges, see the forward kernel)
  float dsum = 0.f;
  for (int e = s0; e < s1; e += EA_UB) {
    float vv[EA_UB][CH], ev[EA_UB][CH], av[EA_UB];
#pragma unroll
    for (int u = 0; u < EA_UB; ++u) {
      const int eu = min(e + u, s1 - 1);
      const float* vr = v + (size_t)src[eu] * ld + h * CH;
      const float* er = ee + (size_t)eu * ld_ee + h * CH;
      av[u] = alpha[(size_t)eu * H + h];
#pragma unroll
      for (int c = 0; c < CH; ++c) { vv[u][c] = vr[c]; ev[u][c] = er[c]; }
    }
#pragma unroll
    for (int u = 0; u < EA_UB; ++u) {
      if (e + u < s1) {
        const float a = av[u];
        float ms = 1.f;
        if (p_drop > 0.f) ms = (msde_uniform(seed, (unsigned long long)(e + u) * H + h) >= p_drop) ? keep_scale : 0.f;
        float ga = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          ga = fmaf(go[c], vv[u][c] + ev[u][c], ga);
          g_vpe[(size_t)(e + u) * ld_kv + h * CH + c] = go[c] * (a * ms);
        }
        dsum = fmaf(a, ga * ms, dsum);
      }
    }
  }
  // pass 2: softmax backward, g_q, g_kpe, g_ee
  for (int e = s0; e < s1; e += EA_UB) {
    float vv[EA_UB][CH], kv[EA_UB][CH], ev[EA_UB][CH], av[EA_UB];
#pragma unroll
    for (int u = 0; u < EA_UB; ++u) {
      const int eu = min(e + u, s1 - 1);
      const int j = src[eu];
      const float* vr = v + (size_t)j * ld + h * CH;
      const float* kr = k + (size_t)j * ld + h * CH;
      const float* er = ee + (size_t)eu * ld_ee + h * CH;
      av[u] = alpha[(size_t)eu * H + h];
#pragma unroll
      for (int c = 0; c < CH; ++c) { vv[u][c] = vr[c]; kv[u][c] = kr[c]; ev[u][c] = er[c]; }
    }
#pragma unroll
    for (int u = 0; u < EA_UB; ++u) {
      if (e + u < s1) {
        const float a = av[u];
        float ms = 1.f;
        if (p_drop > 0.f) ms = (msde_uniform(seed, (unsigned long long)(e + u) * H + h) >= p_drop) ? keep_scale : 0.f;
        float ga = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) ga = fmaf(go[c], vv[u][c] + ev[u][c], ga);
        float gs = a * (ga * ms - dsum) * scale;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          gq[c] = fmaf(gs, kv[u][c] + ev[u][c], gq[c]);
          float gk = gs * qv[c];
          g_kpe[(size_t)(e + u) * ld_kv + h * CH + c] = gk;
          g_ee[(size_t)(e + u) * ld_ee + h * CH + c] = gk + go[c] * (a * ms);
        }
      }
    }
  }
#pragma unroll
  for (int c = 0; c < CH; ++c) g_q[(size_t)i * ldg + h * CH + c] = gq[c];
}

// ---- wave-per-target variants for the production shape (8 heads x 4 channels) -----------------------------------
// The kernels above give one THREAD a (target, head) pair and walk its ~10 in-edges serially: N * H = 28.7 k threads are
// 448 waves for 1024 SIMDs, and every edge is an index -> row -> arithmetic chain (17 us for 17.4 MB = 0.13 of HBM).
// Here one WAVE owns a target: lane = 8 * l + h works on head h of in-edge l, l + 8, ...: the 8 head lanes of an edge
// read one contiguous 128-byte row piece of k / v / edge features as float4, 8 edges are in flight per wave, and the
// softmax statistics and the weighted sums are combined across the 8 edge lanes with three xor shuffles.  Same three
// passes, same use of `alpha` as the score scratch; any in-degree.
__device__ __forceinline__ float ea_red_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 8, 64)); v = fmaxf(v, __shfl_xor(v, 16, 64)); return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float ea_red_sum(float v) {
  v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64);
}
__device__ __forceinline__ float ea_dot4(float4 a, float4 b, float4 c) {      // a . (b + c), channel order
  return ((a.x * (b.x + c.x) + a.y * (b.y + c.y)) + a.z * (b.z + c.z)) + a.w * (b.w + c.w);
}

__global__ void __launch_bounds__(256)
edge_attention_fwd_wave_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                               const float* __restrict__ skip, int ld, const float* __restrict__ ee, int ld_ee,
                               const int* __restrict__ rowptr, const int* __restrict__ src, int N, float p_drop,
                               unsigned long long seed, const unsigned long long* __restrict__ seed_dev,
                               float* __restrict__ alpha, float* __restrict__ out) {
  constexpr int H = 8, CH = 4, D = 32;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= N) return;
  if (seed_dev) seed += seed_dev[0] * 0x100000001B3ull;
  const int lane = threadIdx.x & 63, h = lane & 7, l = lane >> 3;
  const float4 q4 = *reinterpret_cast<const float4*>(q + (size_t)i * ld + h * CH);
  const float scale = 0.5f;                                  // 1 / sqrt(CH)
  const int s0 = rowptr[i], s1 = rowptr[i + 1];
  float m = -INFINITY;
  for (int e = s0 + l; e < s1; e += 8) {
    const float4 k4 = *reinterpret_cast<const float4*>(k + (size_t)src[e] * ld + h * CH);
    const float4 e4 = *reinterpret_cast<const float4*>(ee + (size_t)e * ld_ee + h * CH);
    const float sc = ea_dot4(q4, k4, e4) * scale;
    alpha[(size_t)e * H + h] = sc;
    m = fmaxf(m, sc);
  }
  m = ea_red_max(m);
  float sum = 0.f;
  for (int e = s0 + l; e < s1; e += 8) {
    const float p = expf(alpha[(size_t)e * H + h] - m);
    alpha[(size_t)e * H + h] = p;
    sum += p;
  }
  sum = ea_red_sum(sum);
  const float inv = 1.f / (sum + 1e-16f);
  const float keep_scale = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int e = s0 + l; e < s1; e += 8) {
    const float4 v4 = *reinterpret_cast<const float4*>(v + (size_t)src[e] * ld + h * CH);
    const float4 e4 = *reinterpret_cast<const float4*>(ee + (size_t)e * ld_ee + h * CH);
    float a = alpha[(size_t)e * H + h] * inv;
    alpha[(size_t)e * H + h] = a;
    if (p_drop > 0.f) a = (msde_uniform(seed, (unsigned long long)e * H + h) >= p_drop) ? a * keep_scale : 0.f;
    acc.x = fmaf(a, v4.x + e4.x, acc.x); acc.y = fmaf(a, v4.y + e4.y, acc.y);
    acc.z = fmaf(a, v4.z + e4.z, acc.z); acc.w = fmaf(a, v4.w + e4.w, acc.w);
  }
  acc.x = ea_red_sum(acc.x); acc.y = ea_red_sum(acc.y); acc.z = ea_red_sum(acc.z); acc.w = ea_red_sum(acc.w);
  if (l == 0) {
    if (skip) {
      const float4 s4 = *reinterpret_cast<const float4*>(skip + (size_t)i * ld + h * CH);   // + lin_skip(x_i)
      acc.x += s4.x; acc.y += s4.y; acc.z += s4.z; acc.w += s4.w;
    }
    *reinterpret_cast<float4*>(out + (size_t)i * D + h * CH) = acc;
  }
}

__global__ void __launch_bounds__(256)
edge_attention_bwd_wave_kernel(const float* __restrict__ g_out, const float* __restrict__ q, const float* __restrict__ k,
                               const float* __restrict__ v, int ld, float* __restrict__ g_skip, int ldg,
                               const float* __restrict__ ee, int ld_ee, const float* __restrict__ alpha,
                               const int* __restrict__ rowptr, const int* __restrict__ src, int N, float p_drop,
                               unsigned long long seed, const unsigned long long* __restrict__ seed_dev,
                               float* __restrict__ g_q, float* __restrict__ g_ee, float* __restrict__ g_kpe,
                               float* __restrict__ g_vpe, int ld_kv) {
  constexpr int H = 8, CH = 4, D = 32;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= N) return;
  if (seed_dev) seed += seed_dev[0] * 0x100000001B3ull;
  const int lane = threadIdx.x & 63, h = lane & 7, l = lane >> 3;
  const float4 q4 = *reinterpret_cast<const float4*>(q + (size_t)i * ld + h * CH);
  const float4 go = *reinterpret_cast<const float4*>(g_out + (size_t)i * D + h * CH);
  if (g_skip && l == 0) *reinterpret_cast<float4*>(g_skip + (size_t)i * ldg + h * CH) = go;   // d(out)/d(skip) = 1
  const float scale = 0.5f;
  const float keep_scale = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  const int s0 = rowptr[i], s1 = rowptr[i + 1];
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  // pass 1: g_vpe, and dsum = sum_e alpha_e * g_alpha_e
  float dsum = 0.f;
  for (int e = s0 + l; e < s1; e += 8) {
    const float4 v4 = *reinterpret_cast<const float4*>(v + (size_t)src[e] * ld + h * CH);
    const float4 e4 = *reinterpret_cast<const float4*>(ee + (size_t)e * ld_ee + h * CH);
    const float a = alpha[(size_t)e * H + h];
    float ms = 1.f;
    if (p_drop > 0.f) ms = (msde_uniform(seed, (unsigned long long)e * H + h) >= p_drop) ? keep_scale : 0.f;
    const float ga = ea_dot4(go, v4, e4), am = a * ms;
    *reinterpret_cast<float4*>(g_vpe + (size_t)e * ld_kv + h * CH) = make_float4(go.x * am, go.y * am, go.z * am, go.w * am);
    dsum = fmaf(a, ga * ms, dsum);
  }
  dsum = ea_red_sum(dsum);
  // pass 2: softmax backward, g_q, g_kpe, g_ee
  float4 gq = zero;
  for (int e = s0 + l; e < s1; e += 8) {
    const int j = src[e];
    const float4 v4 = *reinterpret_cast<const float4*>(v + (size_t)j * ld + h * CH);
    const float4 k4 = *reinterpret_cast<const float4*>(k + (size_t)j * ld + h * CH);
    const float4 e4 = *reinterpret_cast<const float4*>(ee + (size_t)e * ld_ee + h * CH);
    const float a = alpha[(size_t)e * H + h];
    float ms = 1.f;
    if (p_drop > 0.f) ms = (msde_uniform(seed, (unsigned long long)e * H + h) >= p_drop) ? keep_scale : 0.f;
    const float ga = ea_dot4(go, v4, e4);
    const float gs = a * (ga * ms - dsum) * scale, am = a * ms;
    gq.x = fmaf(gs, k4.x + e4.x, gq.x); gq.y = fmaf(gs, k4.y + e4.y, gq.y);
    gq.z = fmaf(gs, k4.z + e4.z, gq.z); gq.w = fmaf(gs, k4.w + e4.w, gq.w);
    const float4 gk = make_float4(gs * q4.x, gs * q4.y, gs * q4.z, gs * q4.w);
    *reinterpret_cast<float4*>(g_kpe + (size_t)e * ld_kv + h * CH) = gk;
    *reinterpret_cast<float4*>(g_ee + (size_t)e * ld_ee + h * CH) =
        make_float4(gk.x + go.x * am, gk.y + go.y * am, gk.z + go.z * am, gk.w + go.w * am);
  }
  gq.x = ea_red_sum(gq.x); gq.y = ea_red_sum(gq.y); gq.z = ea_red_sum(gq.z); gq.w = ea_red_sum(gq.w);
  if (l == 0) *reinterpret_cast<float4*>(g_q + (size_t)i * ldg + h * CH) = gq;
}

static inline bool ea_al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline bool ea_wave_path() { return true; }   // wave-per-target kernels for 8 heads x 4 channels; the thread-per-(target, head) kernels take every other shape

extern "C" int msde_edge_attention_fwd(const float* q, const float* k, const float* v, const float* skip, int ld,
                                       const float* ee, int ld_ee,
                                       const int* rowptr, const int* src, int N, int H, int Ch, float p_drop,
                                       unsigned long long seed, const unsigned long long* seed_dev, float* alpha,
                                       float* out, void* stream) {
  if (N < 0 || H <= 0 || Ch <= 0 || !q || !k || !v || !ee || !rowptr || !src || !alpha || !out) return MSDE_EINVAL;
  if (ld_ee == 0) ld_ee = H * Ch;
  if (p_drop < 0.f || p_drop >= 1.f || ld < H * Ch || ld_ee < H * Ch) return MSDE_EINVAL;
  if (N == 0) return 0;
  if (ea_wave_path() && H == 8 && Ch == 4 && ld % 4 == 0 && ld_ee % 4 == 0 && ea_al16(q) && ea_al16(k) && ea_al16(v) &&
      ea_al16(ee) && ea_al16(out) && (!skip || ea_al16(skip))) {
    MSDE_LAUNCH(edge_attention_fwd_wave_kernel, dim3((N + 3) / 4), dim3(256), 0, as_stream(stream), q, k, v, skip, ld, ee,
                ld_ee, rowptr, src, N, p_drop, seed, seed_dev, alpha, out);
    MSDE_CHECK_LAUNCH();
    return 0;
  }
  dim3 grid((N * H + 255) / 256), block(256);
  switch (Ch) {
    case 1: MSDE_LAUNCH(edge_attention_fwd_kernel<1>, grid, block, 0, as_stream(stream), q, k, v, skip, ld, ee, ld_ee, rowptr, src, N, H, p_drop, seed, seed_dev, alpha, out); break;
    case 2: MSDE_LAUNCH(edge_attention_fwd_kernel<2>, grid, block, 0, as_stream(stream), q, k, v, skip, ld, ee, ld_ee, rowptr, src, N, H, p_drop, seed, seed_dev, alpha, out); break;
    case 4: MSDE_LAUNCH(edge_attention_fwd_kernel<4>, grid, block, 0, as_stream(stream), q, k, v, skip, ld, ee, ld_ee, rowptr, src, N, H, p_drop, seed, seed_dev, alpha, out); break;
    case 8: MSDE_LAUNCH(edge_attention_fwd_kernel<8>, grid, block, 0, as_stream(stream), q, k, v, skip, ld, ee, ld_ee, rowptr, src, N, H, p_drop, seed, seed_dev, alpha, out); break;
    default: return MSDE_EUNSUP;
  }
  MSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int msde_edge_attention_bwd(const float* g_out, const float* q, const float* k, const float* v, int ld,
                                       float* g_skip, int ldg,
                                       const float* ee, int ld_ee, const float* alpha, const int* rowptr, const int* src,
                                       int N, int H, int Ch, float p_drop, unsigned long long seed,
                                       const unsigned long long* seed_dev, float* g_q, float* g_ee, float* g_kpe,
                                       float* g_vpe, int ld_kv, void* stream) {
  if (N < 0 || H <= 0 || Ch <= 0 || !g_out || !q || !k || !v || !ee || !alpha || !rowptr || !src || !g_q || !g_ee ||
      !g_kpe || !g_vpe)
    return MSDE_EINVAL;
  if (ld_ee == 0) ld_ee = H * Ch;
  if (ld_kv == 0) ld_kv = H * Ch;
  if (p_drop < 0.f || p_drop >= 1.f || ld_ee < H * Ch || ld_kv < H * Ch) return MSDE_EINVAL;
  if (N == 0) return 0;
  if (ea_wave_path() && H == 8 && Ch == 4 && ld % 4 == 0 && ld_ee % 4 == 0 && ld_kv % 4 == 0 && ldg % 4 == 0 &&
      ea_al16(g_out) && ea_al16(q) && ea_al16(k) && ea_al16(v) && ea_al16(ee) && ea_al16(g_q) && ea_al16(g_ee) &&
      ea_al16(g_kpe) && ea_al16(g_vpe) && (!g_skip || ea_al16(g_skip))) {
    MSDE_LAUNCH(edge_attention_bwd_wave_kernel, dim3((N + 3) / 4), dim3(256), 0, as_stream(stream), g_out, q, k, v, ld,
                g_skip, ldg, ee, ld_ee, alpha, rowptr, src, N, p_drop, seed, seed_dev, g_q, g_ee, g_kpe, g_vpe, ld_kv);
    MSDE_CHECK_LAUNCH();
    return 0;
  }
  dim3 grid((N * H + 255) / 256), block(256);
  switch (Ch) {
    case 1: MSDE_LAUNCH(edge_attention_bwd_kernel<1>, grid, block, 0, as_stream(stream), g_out, q, k, v, ld, g_skip, ldg, ee, ld_ee, alpha, rowptr, src, N, H, p_drop, seed, seed_dev, g_q, g_ee, g_kpe, g_vpe, ld_kv); break;
    case 2: MSDE_LAUNCH(edge_attention_bwd_kernel<2>, grid, block, 0, as_stream(stream), g_out, q, k, v, ld, g_skip, ldg, ee, ld_ee, alpha, rowptr, src, N, H, p_drop, seed, seed_dev, g_q, g_ee, g_kpe, g_vpe, ld_kv); break;
    case 4: MSDE_LAUNCH(edge_attention_bwd_kernel<4>, grid, block, 0, as_stream(stream), g_out, q, k, v, ld, g_skip, ldg, ee, ld_ee, alpha, rowptr, src, N, H, p_drop, seed, seed_dev, g_q, g_ee, g_kpe, g_vpe, ld_kv); break;
    case 8: MSDE_LAUNCH(edge_attention_bwd_kernel<8>, grid, block, 0, as_stream(stream), g_out, q, k, v, ld, g_skip, ldg, ee, ld_ee, alpha, rowptr, src, N, H, p_drop, seed, seed_dev, g_q, g_ee, g_kpe, g_vpe, ld_kv); break;
    default: return MSDE_EUNSUP;
  }
  MSDE_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// frame mix + mean scatter (equivariant_scorenetwork.py:159-164)
// ------------------------------------------------------------------------------------------------
// 8 lanes per target node: lane l takes in-edges l, l + 8, ... (one thread per (node, component) walking its edges one
// after the other was 44 workgroups of serial index -> row chains); partial sums meet in three xor shuffles.
#define FM_LPN 8
__global__ void frame_mix_mean_fwd_kernel(const float* __restrict__ coff, const float* __restrict__ basis,
                                          const int* __restrict__ rowptr, int N, const float* __restrict__ base,
                                          float* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = t / FM_LPN, l = t % FM_LPN;
  float ax = 0.f, ay = 0.f, az = 0.f;
  int s0 = 0, s1 = 0;
  if (i < N) {
    s0 = rowptr[i]; s1 = rowptr[i + 1];
    for (int e = s0 + l; e < s1; e += FM_LPN) {
      const float* c = coff + 3 * (size_t)e;
      const float* b = basis + 9 * (size_t)e;
      const float c0 = c[0], c1 = c[1], c2 = c[2];
      // (c0*diff + c1*cross) + c2*vert, as written in the reference
      ax += (c0 * b[0] + c1 * b[3]) + c2 * b[6];
      ay += (c0 * b[1] + c1 * b[4]) + c2 * b[7];
      az += (c0 * b[2] + c1 * b[5]) + c2 * b[8];
    }
  }
#pragma unroll
  for (int o = 1; o < FM_LPN; o <<= 1) { ax += __shfl_xor(ax, o); ay += __shfl_xor(ay, o); az += __shfl_xor(az, o); }
  if (i < N && l == 0) {
    const float inv = 1.f / (float)max(s1 - s0, 1);
    // base: the running sum of the earlier score layers (equivariant_scorenetwork.py:166 `gradient += ...`)
    const float bx = base ? base[3 * i] : 0.f, by = base ? base[3 * i + 1] : 0.f, bz = base ? base[3 * i + 2] : 0.f;
    out[3 * i] = bx + ax * inv; out[3 * i + 1] = by + ay * inv; out[3 * i + 2] = bz + az * inv;
  }
}

__global__ void frame_mix_mean_bwd_kernel(const float* __restrict__ g_out, const float* __restrict__ basis,
                                          const int* __restrict__ rowptr, int N, int E_cap,
                                          float* __restrict__ g_coff) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = t / FM_LPN, l = t % FM_LPN;
  if (i < N) {
    int s0 = rowptr[i], s1 = rowptr[i + 1];
    float inv = 1.f / (float)max(s1 - s0, 1);
    float gx = g_out[3 * i] * inv, gy = g_out[3 * i + 1] * inv, gz = g_out[3 * i + 2] * inv;
    for (int e = s0 + l; e < s1; e += FM_LPN) {
      const float* b = basis + 9 * (size_t)e;
      float* g = g_coff + 3 * (size_t)e;
      g[0] = gx * b[0] + gy * b[1] + gz * b[2];
      g[1] = gx * b[3] + gy * b[4] + gz * b[5];
      g[2] = gx * b[6] + gy * b[7] + gz * b[8];
    }
  }
  int E = rowptr[N];
  for (int e = E + blockIdx.x * blockDim.x + threadIdx.x; e < E_cap; e += gridDim.x * blockDim.x) {
    g_coff[3 * (size_t)e] = 0.f; g_coff[3 * (size_t)e + 1] = 0.f; g_coff[3 * (size_t)e + 2] = 0.f;
  }
}

extern "C" int msde_frame_mix_mean_add_fwd(const float* coff, const float* basis, const int* rowptr, int N,
                                           const float* base, float* out, void* stream) {
  if (N < 0 || !coff || !basis || !rowptr || !out) return MSDE_EINVAL;
  if (N == 0) return 0;
  MSDE_LAUNCH(frame_mix_mean_fwd_kernel, dim3((N * FM_LPN + 255) / 256), dim3(256), 0, as_stream(stream), coff,
                     basis, rowptr, N, base, out);
  MSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int msde_frame_mix_mean_fwd(const float* coff, const float* basis, const int* rowptr, int N,
                                       float* out, void* stream) {
  return msde_frame_mix_mean_add_fwd(coff, basis, rowptr, N, nullptr, out, stream);
}

extern "C" int msde_frame_mix_mean_bwd(const float* g_out, const float* basis, const int* rowptr, int N, int E_cap,
                                       float* g_coff, void* stream) {
  if (N < 0 || !g_out || !basis || !rowptr || !g_coff) return MSDE_EINVAL;
  if (N == 0) return 0;
  MSDE_LAUNCH(frame_mix_mean_bwd_kernel, dim3((N * FM_LPN + 255) / 256), dim3(256), 0, as_stream(stream), g_out, basis,
                     rowptr, N, E_cap, g_coff);
  MSDE_CHECK_LAUNCH();
  return 0;
}
