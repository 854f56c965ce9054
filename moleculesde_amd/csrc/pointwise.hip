// pointwise.hip — the small element-wise stages of the pretrain step that are not worth a GEMM epilogue but,
// left to the framework, cost one launch per arithmetic operator (the step is launch/latency bound at
// 3.6 k atoms): ShiftedSoftplus, SiLU(+dropout), a*b+c, and the VE position loss with its two reductions.
#include "msde_common.h"

#define PW_LOG2 0.69314718246459961f

// y = softplus(x) - log 2 with torch's threshold-20 rule (schnet.py:199-206 ShiftedSoftplus); fp32 like the
// reference: log1p(exp(x)) evaluated as max(x,0) + log1p(exp(-|x|)) (same value, no overflow)
__device__ __forceinline__ float pw_ssp(float x) {
  float sp = x > 20.f ? x : fmaxf(x, 0.f) + log1pf(expf(-fabsf(x)));
  return sp - PW_LOG2;
}
__device__ __forceinline__ float pw_sigmoid(float x) {
  float e = expf(-fabsf(x));
  float r = 1.f / (1.f + e);
  return x >= 0.f ? r : e * r;
}

__global__ void __launch_bounds__(256) ssp_fwd_kernel(const float* __restrict__ x, long long n, float* __restrict__ y) {
  long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < n) {
    float4 v = *reinterpret_cast<const float4*>(x + i);
    *reinterpret_cast<float4*>(y + i) = make_float4(pw_ssp(v.x), pw_ssp(v.y), pw_ssp(v.z), pw_ssp(v.w));
  } else {
    for (; i < n; ++i) y[i] = pw_ssp(x[i]);
  }
}

// g_x = g * sigmoid(x)   (softplus_backward with beta 1; x > 20 -> 1 to fp32 rounding either way)
__global__ void __launch_bounds__(256)
ssp_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x, long long n, float* __restrict__ gx) {
  long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < n) {
    float4 v = *reinterpret_cast<const float4*>(x + i), d = *reinterpret_cast<const float4*>(g + i);
    *reinterpret_cast<float4*>(gx + i) = make_float4(d.x * (v.x > 20.f ? 1.f : pw_sigmoid(v.x)), d.y * (v.y > 20.f ? 1.f : pw_sigmoid(v.y)),
                                                     d.z * (v.z > 20.f ? 1.f : pw_sigmoid(v.z)), d.w * (v.w > 20.f ? 1.f : pw_sigmoid(v.w)));
  } else {
    for (; i < n; ++i) gx[i] = g[i] * (x[i] > 20.f ? 1.f : pw_sigmoid(x[i]));
  }
}

// y = silu(x) * keep / (1 - p): nn.SiLU followed by nn.Dropout(p) (equivariant_scorenetwork.py:27-31); the keep
// mask is a counter-based function of (seed, element index) and is regenerated in the backward.
__global__ void __launch_bounds__(256)
silu_dropout_fwd_kernel(const float* __restrict__ x, long long n, float p, unsigned long long seed,
                        const unsigned long long* __restrict__ seed_dev, float* __restrict__ y) {
  if (seed_dev) seed += seed_dev[0] * 0x100000001B3ull;
  const float scale = p > 0.f ? 1.f / (1.f - p) : 1.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float v = x[i];
    float s = v * pw_sigmoid(v);
    if (p > 0.f) s = msde_uniform(seed, (unsigned long long)i) >= p ? s * scale : 0.f;
    y[i] = s;
  }
}

__global__ void __launch_bounds__(256)
silu_dropout_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x, long long n, float p,
                        unsigned long long seed, const unsigned long long* __restrict__ seed_dev,
                        float* __restrict__ gx) {
  if (seed_dev) seed += seed_dev[0] * 0x100000001B3ull;
  const float scale = p > 0.f ? 1.f / (1.f - p) : 1.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float v = x[i], d = g[i];
    float sg = pw_sigmoid(v);
    float ds = sg * (1.f + v * (1.f - sg));          // d silu / dx
    if (p > 0.f) d = msde_uniform(seed, (unsigned long long)i) >= p ? d * scale : 0.f;
    gx[i] = d * ds;
  }
}

// out = a * b + c   (edge_attr = invariant * edge_2D + frame_invariant, SDE_model_2D_to_3D.py:393)
__global__ void __launch_bounds__(256)
mul_add_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c, long long n,
                   float* __restrict__ out) {
#pragma clang fp contract(off)   // separate roundings for the product and the sum, like the reference's two operators
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    out[i] = a[i] * b[i] + c[i];
}
__global__ void __launch_bounds__(256)
mul_add_bwd_kernel(const float* __restrict__ g, const float* __restrict__ a, const float* __restrict__ b, long long n,
                   float* __restrict__ ga, float* __restrict__ gb) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float d = g[i];
    if (ga) ga[i] = d * b[i];
    if (gb) gb[i] = d * a[i];
  }
}

// VE position loss (SDE_model_2D_to_3D.py:425-432): per atom l_i = sum_k (score - noise)^2 [* std_i^power],
// scatter_mean over molecules, mean over molecules.  One wave per molecule, then one block over molecules; both
// reductions run in a fixed order.
__global__ void __launch_bounds__(256)
ve_pos_loss_mol_kernel(const float* __restrict__ scores, const float* __restrict__ noise, const float* __restrict__ std,
                       float power, const int* __restrict__ mol_ptr, int B, float* __restrict__ mol_val) {
  int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (b >= B) return;
  int i0 = mol_ptr[b], i1 = mol_ptr[b + 1];
  float acc = 0.f;
  for (int i = i0 + lane; i < i1; i += 64) {
    float dx = scores[3 * i] - noise[3 * i], dy = scores[3 * i + 1] - noise[3 * i + 1], dz = scores[3 * i + 2] - noise[3 * i + 2];
    float w = std ? powf(std[i], power) : 1.f;
    acc += (dx * dx * w + dy * dy * w) + dz * dz * w;
  }
  acc = group_sum(acc, 64);
  if (lane == 0) mol_val[b] = i1 > i0 ? acc / (float)(i1 - i0) : 0.f;
}

__global__ void __launch_bounds__(256) ve_pos_loss_final_kernel(const float* __restrict__ mol_val, int B, float* __restrict__ loss) {
  __shared__ float red[256];
  float acc = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) acc += mol_val[b];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = red[0] / (float)B;
}

__global__ void __launch_bounds__(256)
ve_pos_loss_bwd_kernel(const float* __restrict__ scores, const float* __restrict__ noise, const float* __restrict__ std,
                       float power, const int* __restrict__ mol_ptr, const int* __restrict__ batch, int N, int B,
                       const float* __restrict__ g_loss, float* __restrict__ g_scores) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  int b = batch[i];
  float cnt = (float)(mol_ptr[b + 1] - mol_ptr[b]);
  float w = std ? powf(std[i], power) : 1.f;
  // padded atoms (molecule index B, an empty molecule: capacity buckets) take part in no loss: zero gradient
  float k = (b < B && cnt > 0.f) ? g_loss[0] * 2.f * w / (cnt * (float)B) : 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) g_scores[3 * i + c] = k * (scores[3 * i + c] - noise[3 * i + c]);
}

// VE perturbation of the coordinates (SDE_model_2D_to_3D.py:401-412): per molecule b the time step is
// ts = draws[b] for b < H = B/2+1 and T - draws[b-H] - 1 after that (the antithetic half), t = ts/T*(1-eps)+eps,
// std = sigma_min (sigma_max/sigma_min)^t, pos_perturbed = pos + std * noise.  One thread per atom.
__global__ void __launch_bounds__(256)
ve_perturb_kernel(const float* __restrict__ pos, const float* __restrict__ noise, const long long* __restrict__ draws,
                  const int* __restrict__ batch, int N, int B, int T, float eps, float sigma_min, float sigma_ratio,
                  float* __restrict__ pos_out, float* __restrict__ std_out) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  int b = batch[i];
  int H = B / 2 + 1;
  long long ts = b < H ? draws[b] : (long long)T - draws[b - H] - 1;
  float t = (float)ts / (float)T;
  t = t * (1.0f - eps) + eps;
  float sd = sigma_min * powf(sigma_ratio, t);
  std_out[i] = sd;
#pragma unroll
  for (int c = 0; c < 3; ++c) pos_out[3 * i + c] = pos[3 * i + c] + sd * noise[3 * i + c];
}

extern "C" int msde_ve_perturb(const float* pos, const float* noise, const long long* draws, const int* batch, int N,
                               int B, int T, float eps, float sigma_min, float sigma_max, float* pos_out,
                               float* std_out, void* stream) {
  if (N < 0 || B <= 0 || T <= 0 || !pos || !noise || !draws || !batch || !pos_out || !std_out || sigma_min <= 0.f)
    return MSDE_EINVAL;
  if (N == 0) return 0;
  MSDE_LAUNCH(ve_perturb_kernel, dim3((N + 255) / 256), dim3(256), 0, as_stream(stream), pos, noise, draws, batch, N, B,
              T, eps, sigma_min, sigma_max / sigma_min, pos_out, std_out);
  MSDE_CHECK_LAUNCH();
  return 0;
}

// The same with the draws made in the kernel (counter-based generator: seed, device step counter, index): the position
// noise N(0,1) per coordinate (written to noise_out: the loss needs it) and one uniform integer time step in [0, T) per
// antithetic pair of molecules -- replaces torch.randn_like + torch.randint + msde_ve_perturb (3 launches -> 1).
__global__ void __launch_bounds__(256)
ve_perturb_rng_kernel(const float* __restrict__ pos, const int* __restrict__ batch, int N, int B, int T, float eps,
                      float sigma_min, float sigma_ratio, unsigned long long seed,
                      const unsigned long long* __restrict__ seed_dev, float* __restrict__ noise_out,
                      float* __restrict__ pos_out, float* __restrict__ std_out) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  if (seed_dev) seed += seed_dev[0] * 0x100000001B3ull;
  int b = batch[i];
  int H = B / 2 + 1;
  const float u = msde_uniform(seed ^ 0x7157EEDC0FFEEull, (unsigned long long)(b < H ? b : b - H));
  long long d0 = (long long)(u * (float)T);
  if (d0 > T - 1) d0 = T - 1;
  long long ts = b < H ? d0 : (long long)T - d0 - 1;
  float t = (float)ts / (float)T;
  t = t * (1.0f - eps) + eps;
  float sd = sigma_min * powf(sigma_ratio, t);
  std_out[i] = sd;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float z = msde_randn(seed, 3ull * (unsigned long long)i + c);
    noise_out[3 * i + c] = z;
    pos_out[3 * i + c] = pos[3 * i + c] + sd * z;
  }
}

extern "C" int msde_ve_perturb_rng(const float* pos, const int* batch, int N, int B, int T, float eps, float sigma_min,
                                   float sigma_max, unsigned long long seed, const unsigned long long* seed_dev,
                                   float* noise_out, float* pos_out, float* std_out, void* stream) {
  if (N < 0 || B <= 0 || T <= 0 || !pos || !batch || !noise_out || !pos_out || !std_out || sigma_min <= 0.f) return MSDE_EINVAL;
  if (N == 0) return 0;
  MSDE_LAUNCH(ve_perturb_rng_kernel, dim3((N + 255) / 256), dim3(256), 0, as_stream(stream), pos, batch, N, B, T, eps,
              sigma_min, sigma_max / sigma_min, seed, seed_dev, noise_out, pos_out, std_out);
  MSDE_CHECK_LAUNCH();
  return 0;
}

// Random permutation of 0..n-1 (torch.randperm for the contrastive negatives, examples/util.py:55) for n <= 4096:
// out[i] = rank of key_i among n i.i.d. 64-bit keys (52 counter-based random bits: seed, device step counter,
// index | 12 index bits, so keys are distinct).  The ranks of i.i.d. keys are a uniform random permutation.
// Every workgroup keeps all keys in LDS and ranks 64 of them by counting (four lanes per key, broadcast LDS
// reads): n^2 compares spread over n/64 workgroups -- one short launch instead of key generation + a multi-pass sort.
#define RP_MAX 8192
#define RP_BLOCK 256    // 64 keys per workgroup (four lanes each): every workgroup regenerates all n keys, so fewer,
                        // larger workgroups keep that redundant hashing small
__global__ void __launch_bounds__(RP_BLOCK)
randperm_kernel(int n, const int* __restrict__ ndev, unsigned long long seed,
                const unsigned long long* __restrict__ seed_dev, int* __restrict__ out) {
  __shared__ unsigned long long key[RP_MAX];
  const int ncap = n;
  n = msde_true_rows(n, ndev);          // permutation of the valid rows; entries past them map to themselves
  if (seed_dev) seed += seed_dev[0] * 0x100000001B3ull;
  seed += 0xD1B54A32D192ED03ull * blockIdx.y;       // blockIdx.y: which of the `count` independent permutations
  out += (size_t)blockIdx.y * ncap;
  for (int i = threadIdx.x; i < n; i += RP_BLOCK) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * ((unsigned long long)i + 1ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    key[i] = (z & ~0xFFFull) | (unsigned long long)i;
  }
  __syncthreads();
  // four lanes per key, each counting a quarter of the keys
  const int i = blockIdx.x * (RP_BLOCK / 4) + (threadIdx.x >> 2), part = threadIdx.x & 3;
  const int quarter = (n + 3) / 4;
  const int j0 = part * quarter, j1 = min(j0 + quarter, n);
  const unsigned long long mine = key[max(min(i, n - 1), 0)];
  int r0 = 0, r1 = 0, r2 = 0, r3 = 0;
  int j = j0;
  for (; j + 3 < j1; j += 4) {
    r0 += key[j] < mine; r1 += key[j + 1] < mine; r2 += key[j + 2] < mine; r3 += key[j + 3] < mine;
  }
  for (; j < j1; ++j) r0 += key[j] < mine;
  int r = (r0 + r1) + (r2 + r3);
  r += __shfl_xor(r, 1);
  r += __shfl_xor(r, 2);
  if (part == 0 && i < n) out[i] = r;
  else if (part == 0 && i < ncap) out[i] = i;
}

extern "C" int msde_randperm(int n, int count, unsigned long long seed, const unsigned long long* seed_dev, int* out,
                             const int* rows_dev, void* stream) {
  if (n < 0 || count < 0 || !out) return MSDE_EINVAL;
  if (n > RP_MAX) return MSDE_EUNSUP;
  if (n == 0 || count == 0) return 0;
  MSDE_LAUNCH(randperm_kernel, dim3((n + RP_BLOCK / 4 - 1) / (RP_BLOCK / 4), count), dim3(RP_BLOCK), 0, as_stream(stream), n,
              rows_dev, seed, seed_dev, out);
  MSDE_CHECK_LAUNCH();
  return 0;
}

static inline int pw_blocks(long long n, int per_thread) {
  long long b = (n + 256LL * per_thread - 1) / (256LL * per_thread);
  if (b > 4096 && per_thread == 1) b = 4096;
  return b < 1 ? 1 : (int)b;
}

extern "C" int msde_ssp_fwd(const float* x, long long n, float* y, void* stream) {
  if (n < 0 || !x || !y) return MSDE_EINVAL;
  if (n == 0) return 0;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) return MSDE_EINVAL;
  MSDE_LAUNCH(ssp_fwd_kernel, dim3(pw_blocks(n, 4)), dim3(256), 0, as_stream(stream), x, n, y);
  MSDE_CHECK_LAUNCH();
  return 0;
}
extern "C" int msde_ssp_bwd(const float* g, const float* x, long long n, float* gx, void* stream) {
  if (n < 0 || !g || !x || !gx) return MSDE_EINVAL;
  if (n == 0) return 0;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(gx)) & 15) return MSDE_EINVAL;
  MSDE_LAUNCH(ssp_bwd_kernel, dim3(pw_blocks(n, 4)), dim3(256), 0, as_stream(stream), g, x, n, gx);
  MSDE_CHECK_LAUNCH();
  return 0;
}
extern "C" int msde_silu_dropout_fwd(const float* x, long long n, float p, unsigned long long seed,
                                     const unsigned long long* seed_dev, float* y, void* stream) {
  if (n < 0 || !x || !y || p < 0.f || p >= 1.f) return MSDE_EINVAL;
  if (n == 0) return 0;
  MSDE_LAUNCH(silu_dropout_fwd_kernel, dim3(pw_blocks(n, 1)), dim3(256), 0, as_stream(stream), x, n, p, seed, seed_dev, y);
  MSDE_CHECK_LAUNCH();
  return 0;
}
extern "C" int msde_silu_dropout_bwd(const float* g, const float* x, long long n, float p, unsigned long long seed,
                                     const unsigned long long* seed_dev, float* gx, void* stream) {
  if (n < 0 || !g || !x || !gx || p < 0.f || p >= 1.f) return MSDE_EINVAL;
  if (n == 0) return 0;
  MSDE_LAUNCH(silu_dropout_bwd_kernel, dim3(pw_blocks(n, 1)), dim3(256), 0, as_stream(stream), g, x, n, p, seed, seed_dev,
              gx);
  MSDE_CHECK_LAUNCH();
  return 0;
}
extern "C" int msde_mul_add_fwd(const float* a, const float* b, const float* c, long long n, float* out, void* stream) {
  if (n < 0 || !a || !b || !c || !out) return MSDE_EINVAL;
  if (n == 0) return 0;
  MSDE_LAUNCH(mul_add_fwd_kernel, dim3(pw_blocks(n, 1)), dim3(256), 0, as_stream(stream), a, b, c, n, out);
  MSDE_CHECK_LAUNCH();
  return 0;
}
extern "C" int msde_mul_add_bwd(const float* g, const float* a, const float* b, long long n, float* ga, float* gb,
                                void* stream) {
  if (n < 0 || !g || !a || !b) return MSDE_EINVAL;
  if (n == 0 || (!ga && !gb)) return 0;
  MSDE_LAUNCH(mul_add_bwd_kernel, dim3(pw_blocks(n, 1)), dim3(256), 0, as_stream(stream), g, a, b, n, ga, gb);
  MSDE_CHECK_LAUNCH();
  return 0;
}
extern "C" int msde_ve_pos_loss_fwd(const float* scores, const float* noise, const float* std, float anneal_power,
                                    const int* mol_ptr, int N, int B, float* mol_ws, float* loss, void* stream) {
  if (N < 0 || B <= 0 || !scores || !noise || !mol_ptr || !mol_ws || !loss) return MSDE_EINVAL;
  const float* sd = anneal_power != 0.f ? std : nullptr;
  if (anneal_power != 0.f && !std) return MSDE_EINVAL;
  MSDE_LAUNCH(ve_pos_loss_mol_kernel, dim3((B + 3) / 4), dim3(256), 0, as_stream(stream), scores, noise, sd, anneal_power,
              mol_ptr, B, mol_ws);
  MSDE_CHECK_LAUNCH();
  MSDE_LAUNCH(ve_pos_loss_final_kernel, dim3(1), dim3(256), 0, as_stream(stream), (const float*)mol_ws, B, loss);
  MSDE_CHECK_LAUNCH();
  return 0;
}
extern "C" int msde_ve_pos_loss_bwd(const float* scores, const float* noise, const float* std, float anneal_power,
                                    const int* mol_ptr, const int* batch, int N, int B, const float* g_loss,
                                    float* g_scores, void* stream) {
  if (N < 0 || B <= 0 || !scores || !noise || !mol_ptr || !batch || !g_loss || !g_scores) return MSDE_EINVAL;
  if (N == 0) return 0;
  const float* sd = anneal_power != 0.f ? std : nullptr;
  if (anneal_power != 0.f && !std) return MSDE_EINVAL;
  MSDE_LAUNCH(ve_pos_loss_bwd_kernel, dim3((N + 255) / 256), dim3(256), 0, as_stream(stream), scores, noise, sd,
              anneal_power, mol_ptr, batch, N, B, g_loss, g_scores);
  MSDE_CHECK_LAUNCH();
  return 0;
}

// ---- the weighted sum of the step's loss terms as ONE launch each way (pretrain_MoleculeSDE.py:139-152: loss =
// sum_i c_i * l_i over up to four scalar terms).  As torch operators the combination and its backward were 9 launches of
// one thread each on the critical path between the forward and the backward pass.
__global__ void combine_losses_kernel(const float* a, const float* b, const float* c, const float* d, float ca, float cb,
                                      float cc, float cd, float* out) {
  float s = 0.f;
  if (a) s = fmaf(ca, *a, s);
  if (b) s = fmaf(cb, *b, s);
  if (c) s = fmaf(cc, *c, s);
  if (d) s = fmaf(cd, *d, s);
  *out = s;
}
__global__ void combine_losses_bwd_kernel(const float* g, float ca, float cb, float cc, float cd, float* out4) {
  const float v = *g;
  out4[0] = v * ca; out4[1] = v * cb; out4[2] = v * cc; out4[3] = v * cd;
}
extern "C" int msde_combine_losses(const float* a, const float* b, const float* c, const float* d, float ca, float cb,
                                   float cc, float cd, float* out, void* stream) {
  if (!out) return MSDE_EINVAL;
  MSDE_LAUNCH(combine_losses_kernel, dim3(1), dim3(1), 0, as_stream(stream), a, b, c, d, ca, cb, cc, cd, out);
  MSDE_CHECK_LAUNCH();
  return 0;
}
// the same launch with two riders: seeds4[i] = c_i (the backward's result for a unit upstream gradient -- the trainer's case --
// so that the backward pass needs no launch of its own), and up to five running sums log_dst[k] += *log_src[k] (the
// per-term logs of pretrain_MoleculeSDE.py:158-163, one more launch at the end of every step otherwise)
struct msde_combine_ex_args { const float* src[5]; float* dst[5]; };
__global__ void combine_losses_ex_kernel(const float* a, const float* b, const float* c, const float* d, float ca, float cb,
                                         float cc, float cd, float* out, float* seeds4, msde_combine_ex_args lg) {
  float s = 0.f;
  if (a) s = fmaf(ca, *a, s);
  if (b) s = fmaf(cb, *b, s);
  if (c) s = fmaf(cc, *c, s);
  if (d) s = fmaf(cd, *d, s);
  *out = s;
  if (seeds4) { seeds4[0] = ca; seeds4[1] = cb; seeds4[2] = cc; seeds4[3] = cd; }
#pragma unroll
  for (int k = 0; k < 5; ++k)
    if (lg.src[k] && lg.dst[k]) *lg.dst[k] += *lg.src[k];
}
extern "C" int msde_combine_losses_ex(const float* a, const float* b, const float* c, const float* d, float ca, float cb,
                                      float cc, float cd, float* out, float* seeds4, const float* const* log_src,
                                      float* const* log_dst, int n_log, void* stream) {
  if (!out || n_log < 0 || n_log > 5 || (n_log > 0 && (!log_src || !log_dst))) return MSDE_EINVAL;
  msde_combine_ex_args lg;
  for (int k = 0; k < 5; ++k) { lg.src[k] = k < n_log ? log_src[k] : nullptr; lg.dst[k] = k < n_log ? log_dst[k] : nullptr; }
  MSDE_LAUNCH(combine_losses_ex_kernel, dim3(1), dim3(1), 0, as_stream(stream), a, b, c, d, ca, cb, cc, cd, out, seeds4, lg);
  MSDE_CHECK_LAUNCH();
  return 0;
}
extern "C" int msde_combine_losses_bwd(const float* g, float ca, float cb, float cc, float cd, float* out4, void* stream) {
  if (!g || !out4) return MSDE_EINVAL;
  MSDE_LAUNCH(combine_losses_bwd_kernel, dim3(1), dim3(1), 0, as_stream(stream), g, ca, cb, cc, cd, out4);
  MSDE_CHECK_LAUNCH();
  return 0;
}

// ---- narrow output layer of an MLP over rows: out[e][j] = b[j] + sum_c silu(Z[e][c]) W[j][c] with J <= 4 outputs
// (basis_mlp of equivariant_scorenetwork.py:142-146: Linear(2D, H) -> SiLU -> Linear(H, 3) on every edge).  As a GEMM the
// J = 3 product wastes 61 of 64 tile columns forward and runs a K = 3 product backward (12 + 35 us at 35 k edges); here
// lpr lanes hold one row as float4 pieces: the activation is applied while reading the PRE-activation (the activated
// tensor is never stored), the backward writes d/dZ (activation derivative included) and accumulates the layer's own
// weight / bias gradient in registers -> one slab per workgroup (summed by msde_reduce_slabs[_multi], fixed order).
#define MH_MAXJ 4
#define MH_MAXWG 256
#define FMX_LPN 8        // lanes per node of the mean over in-edges (= FM_LPN of sde2d3d.hip: same order of additions)
// slab of one workgroup: [gW (J x H) | gb (J)] padded to whole float4 (the batched reduction then takes its vector path)
__host__ __device__ __forceinline__ size_t mh_slab_floats(int J, int H) { return ((size_t)J * H + J + 3) & ~(size_t)3; }
__device__ __forceinline__ float mh_sigmoid(float z) { return 1.f / (1.f + __expf(-z)); }

__global__ void __launch_bounds__(256)
mlp_head_fwd_kernel(const float4* __restrict__ Z, int ldz4, const float4* __restrict__ W, const float* __restrict__ b, int E,
                    int H4, int J, int lpr, float* __restrict__ out) {
  const int rpb = 256 / lpr, group = threadIdx.x / lpr, lane = threadIdx.x % lpr;
  for (int e = blockIdx.x * rpb + group; e < E; e += gridDim.x * rpb) {
    float p[MH_MAXJ] = {0.f, 0.f, 0.f, 0.f};
    for (int c = lane; c < H4; c += lpr) {
      const float4 z = Z[(size_t)e * ldz4 + c];
      const float4 a = make_float4(z.x * mh_sigmoid(z.x), z.y * mh_sigmoid(z.y), z.z * mh_sigmoid(z.z), z.w * mh_sigmoid(z.w));
#pragma unroll
      for (int j = 0; j < MH_MAXJ; ++j)
        if (j < J) {
          const float4 w = W[j * H4 + c];
          p[j] += (a.x * w.x + a.y * w.y) + (a.z * w.z + a.w * w.w);
        }
    }
#pragma unroll
    for (int j = 0; j < MH_MAXJ; ++j)
      if (j < J) {
        const float s = group_sum(p[j], lpr);
        if (lane == 0) out[(size_t)e * J + j] = s + (b ? b[j] : 0.f);
      }
  }
}

// H4 <= lpr here (one float4 column piece per lane): the per-lane weight-gradient accumulators stay in registers.
// MIX (J = 3): the head's output went straight into the frame mix + mean over the in-edges of its target node
// (mlp_head_mix_fwd_kernel), so its gradient is formed here from the NODE gradient: g[e][j] = (gnode[dst_e] / deg) . basis[e][j].
template <bool MIX>
__global__ void __launch_bounds__(256)
mlp_head_bwd_kernel(const float4* __restrict__ Z, int ldz4, const float4* __restrict__ W, const float* __restrict__ g, int E,
                    const int* __restrict__ Edev, int H4, int J, int lpr, float4* __restrict__ gZ, float* __restrict__ slabs,
                    const float* __restrict__ basis, const int* __restrict__ dst, const int* __restrict__ rowptr) {
  __shared__ float4 red[256 * MH_MAXJ];
  __shared__ float redb[64 * MH_MAXJ];
  const int Et = msde_true_rows(E, Edev);      // padded rows: zero gradient, no contribution to the weight gradient
  const int rpb = 256 / lpr, group = threadIdx.x / lpr, lane = threadIdx.x % lpr;
  const bool on = lane < H4;
  float4 w[MH_MAXJ], dw[MH_MAXJ];
  float db[MH_MAXJ];
#pragma unroll
  for (int j = 0; j < MH_MAXJ; ++j) {
    w[j] = (j < J && on) ? W[j * H4 + lane] : vzero4();
    dw[j] = vzero4();
    db[j] = 0.f;
  }
  for (int e = blockIdx.x * rpb + group; e < E; e += gridDim.x * rpb) {
    if (!on) continue;
    float4 t = vzero4();
    if (e < Et) {
      const float4 z = Z[(size_t)e * ldz4 + lane];
      const float4 s = make_float4(mh_sigmoid(z.x), mh_sigmoid(z.y), mh_sigmoid(z.z), mh_sigmoid(z.w));
      const float4 a = make_float4(z.x * s.x, z.y * s.y, z.z * s.z, z.w * s.w);
      float ge[MH_MAXJ] = {0.f, 0.f, 0.f, 0.f};
      if (MIX) {
        const int i = dst[e];
        const float inv = 1.f / (float)max(rowptr[i + 1] - rowptr[i], 1);
        const float gx = g[3 * i] * inv, gy = g[3 * i + 1] * inv, gz = g[3 * i + 2] * inv;
        const float* b = basis + 9 * (size_t)e;
        ge[0] = gx * b[0] + gy * b[1] + gz * b[2];
        ge[1] = gx * b[3] + gy * b[4] + gz * b[5];
        ge[2] = gx * b[6] + gy * b[7] + gz * b[8];
      } else {
#pragma unroll
        for (int j = 0; j < MH_MAXJ; ++j)
          if (j < J) ge[j] = g[(size_t)e * J + j];
      }
#pragma unroll
      for (int j = 0; j < MH_MAXJ; ++j)
        if (j < J) {
          const float gj = ge[j];
          t = make_float4(fmaf(gj, w[j].x, t.x), fmaf(gj, w[j].y, t.y), fmaf(gj, w[j].z, t.z), fmaf(gj, w[j].w, t.w));
          dw[j] = make_float4(fmaf(gj, a.x, dw[j].x), fmaf(gj, a.y, dw[j].y), fmaf(gj, a.z, dw[j].z), fmaf(gj, a.w, dw[j].w));
          db[j] += gj;
        }
      // d silu(z)/dz = s (1 + z (1 - s))
      t = make_float4(t.x * s.x * (1.f + z.x * (1.f - s.x)), t.y * s.y * (1.f + z.y * (1.f - s.y)),
                      t.z * s.z * (1.f + z.z * (1.f - s.z)), t.w * s.w * (1.f + z.w * (1.f - s.w)));
    }
    gZ[(size_t)e * H4 + lane] = t;
  }
  // row groups of the workgroup, summed in group order
#pragma unroll
  for (int j = 0; j < MH_MAXJ; ++j) {
    red[(group * MH_MAXJ + j) * lpr + lane] = dw[j];
    if (lane == 0) redb[group * MH_MAXJ + j] = db[j];
  }
  __syncthreads();
  const int H = 4 * H4;
  float* slab = slabs + (size_t)blockIdx.x * mh_slab_floats(J, H);
  if (group == 0 && on) {
    for (int j = 0; j < J; ++j) {
      float4 s = vzero4();
      for (int q = 0; q < rpb; ++q) s = vadd(s, red[(q * MH_MAXJ + j) * lpr + lane]);
      float* o = slab + (size_t)j * H + 4 * lane;          // J * H + J floats per slab: rows need not be 16-B aligned
      o[0] = s.x; o[1] = s.y; o[2] = s.z; o[3] = s.w;
    }
  }
  if (threadIdx.x < MH_MAXJ && J * H + (int)threadIdx.x < (int)mh_slab_floats(J, H)) {     // bias sums, zeros in the padding
    float s = 0.f;
    if ((int)threadIdx.x < J)
      for (int q = 0; q < rpb; ++q) s += redb[q * MH_MAXJ + threadIdx.x];
    slab[(size_t)J * H + threadIdx.x] = s;
  }
}

static inline bool mh_ok(int H, int J, const void* Z, int ldz, const void* W) {
  return H > 0 && H % 4 == 0 && H <= 256 && J >= 1 && J <= MH_MAXJ && ldz % 4 == 0 && ldz >= H &&
         ((reinterpret_cast<uintptr_t>(Z) | reinterpret_cast<uintptr_t>(W)) & 15) == 0;
}
static inline int mh_grid(int E, int lpr) {
  // workgroups of the backward kernel (= slabs of its weight gradient).  Each row group walks its rows one dependent load
  // chain at a time, so the kernel is bound by rows per group, not by bytes: 256 workgroups (17 rows per group at 35 k
  // edges) 20 us, 1024 (4-5 rows) -- see MSDE_MH_MAXWG
  const int maxwg = 1024;
  const int rpb = 256 / lpr;
  int nb = (E + rpb - 1) / rpb;
  if (nb > maxwg) nb = maxwg;
  return nb < 1 ? 1 : nb;
}

extern "C" int msde_mlp_head_fwd(const float* Z, int ldz, const float* W, const float* b, int E, int H, int J, float* out,
                                 void* stream) {
  if (E < 0 || !Z || !W || !out) return MSDE_EINVAL;
  if (!mh_ok(H, J, Z, ldz, W)) return MSDE_EUNSUP;
  if (E == 0) return 0;
  const int lpr = pick_tpr(H / 4), rpb = 256 / lpr;
  int nb = (E + rpb - 1) / rpb;
  if (nb > 8 * MH_MAXWG) nb = 8 * MH_MAXWG;
  MSDE_LAUNCH(mlp_head_fwd_kernel, dim3(nb), dim3(256), 0, as_stream(stream), reinterpret_cast<const float4*>(Z), ldz / 4,
              reinterpret_cast<const float4*>(W), b, E, H / 4, J, lpr, out);
  MSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int msde_mlp_head_bwd_slabs(int E, int H) { return mh_grid(E, pick_tpr(H / 4)); }

extern "C" int msde_mlp_head_bwd(const float* Z, int ldz, const float* W, const float* g, int E, int H, int J, float* gZ,
                                 float* gWb, float* workspace, const int* rows_dev, void* stream) {
  if (E < 0 || !Z || !W || !g || !gZ || !workspace) return MSDE_EINVAL;
  if (!mh_ok(H, J, Z, ldz, W) || (reinterpret_cast<uintptr_t>(gZ) & 15)) return MSDE_EUNSUP;
  hipStream_t st = as_stream(stream);
  const size_t n = mh_slab_floats(J, H);
  if (E == 0) {
    if (gWb) { int e = msde_zero_words(gWb, n, st); if (e) return e; }
    return 0;
  }
  const int lpr = pick_tpr(H / 4), nb = mh_grid(E, lpr);
  MSDE_LAUNCH(mlp_head_bwd_kernel<false>, dim3(nb), dim3(256), 0, st, reinterpret_cast<const float4*>(Z), ldz / 4,
              reinterpret_cast<const float4*>(W), g, E, rows_dev, H / 4, J, lpr, reinterpret_cast<float4*>(gZ),
              workspace, (const float*)nullptr, (const int*)nullptr, (const int*)nullptr);
  MSDE_CHECK_LAUNCH();
  if (!gWb) return 0;                 // slabs stay in `workspace` for a batched reduction
  return msde_reduce_slabs(workspace, nb, n, gWb, nullptr, 0, nullptr, st);
}

// ---- the same head fused with what consumes it in the 2D->3D score network (equivariant_scorenetwork.py:142-166): the three
// outputs per edge are the coefficients of the edge's frame vectors, mixed and averaged over the in-edges of the target node
// and added to the running sum of the earlier score layers.  Forward: one workgroup per MHX_NPW consecutive nodes -- their
// in-edges are contiguous (by-target order); the per-edge mixed vectors go through `mix` [E, 3] (written and read by the
// same workgroup), then 8 lanes per node add them in the order of frame_mix_mean_fwd_kernel.  coff itself is never stored:
// the backward (mlp_head_bwd_kernel<true>) needs only the node gradient and the frame.
#define MHX_NPW 2
__global__ void __launch_bounds__(256)
mlp_head_mix_fwd_kernel(const float4* __restrict__ Z, int ldz4, const float4* __restrict__ W, const float* __restrict__ b,
                        int H4, int lpr, const float* __restrict__ basis, const int* __restrict__ rowptr, int N,
                        const float* __restrict__ base, float* __restrict__ mix, float* __restrict__ out) {
  const int rpb = 256 / lpr, group = threadIdx.x / lpr, lane = threadIdx.x % lpr;
  const int n0 = blockIdx.x * MHX_NPW, n1 = min(n0 + MHX_NPW, N);
  const int e0 = rowptr[n0], e1 = rowptr[n1];
  const float b0 = b ? b[0] : 0.f, b1 = b ? b[1] : 0.f, b2 = b ? b[2] : 0.f;
  // (H4 <= lpr: one column piece per lane -- its three weight rows stay in registers for all edges of the workgroup)
  const bool one = H4 <= lpr;
  float4 w0 = vzero4(), w1 = vzero4(), w2 = vzero4();
  if (one && lane < H4) { w0 = W[lane]; w1 = W[H4 + lane]; w2 = W[2 * H4 + lane]; }
  for (int e = e0 + group; e < e1; e += rpb) {
    float p[3] = {0.f, 0.f, 0.f};
    if (one) {
      if (lane < H4) {
        const float4 z = Z[(size_t)e * ldz4 + lane];
        const float4 a = make_float4(z.x * mh_sigmoid(z.x), z.y * mh_sigmoid(z.y), z.z * mh_sigmoid(z.z), z.w * mh_sigmoid(z.w));
        p[0] = (a.x * w0.x + a.y * w0.y) + (a.z * w0.z + a.w * w0.w);
        p[1] = (a.x * w1.x + a.y * w1.y) + (a.z * w1.z + a.w * w1.w);
        p[2] = (a.x * w2.x + a.y * w2.y) + (a.z * w2.z + a.w * w2.w);
      }
    } else {
      for (int c = lane; c < H4; c += lpr) {
        const float4 z = Z[(size_t)e * ldz4 + c];
        const float4 a = make_float4(z.x * mh_sigmoid(z.x), z.y * mh_sigmoid(z.y), z.z * mh_sigmoid(z.z), z.w * mh_sigmoid(z.w));
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const float4 w = W[j * H4 + c];
          p[j] += (a.x * w.x + a.y * w.y) + (a.z * w.z + a.w * w.w);
        }
      }
    }
    const float c0 = group_sum(p[0], lpr) + b0, c1 = group_sum(p[1], lpr) + b1, c2 = group_sum(p[2], lpr) + b2;
    if (lane == 0) {
      const float* bs = basis + 9 * (size_t)e;
      float* m = mix + 3 * (size_t)e;
      m[0] = (c0 * bs[0] + c1 * bs[3]) + c2 * bs[6];
      m[1] = (c0 * bs[1] + c1 * bs[4]) + c2 * bs[7];
      m[2] = (c0 * bs[2] + c1 * bs[5]) + c2 * bs[8];
    }
  }
  __syncthreads();
  const int t = threadIdx.x;
  if (t < MHX_NPW * FMX_LPN) {
    const int i = n0 + t / FMX_LPN, l = t % FMX_LPN;
    float ax = 0.f, ay = 0.f, az = 0.f;
    int s0 = 0, s1 = 0;
    if (i < n1) {
      s0 = rowptr[i]; s1 = rowptr[i + 1];
      for (int e = s0 + l; e < s1; e += FMX_LPN) {
        const float* m = mix + 3 * (size_t)e;
        ax += m[0]; ay += m[1]; az += m[2];
      }
    }
#pragma unroll
    for (int o = 1; o < FMX_LPN; o <<= 1) { ax += __shfl_xor(ax, o); ay += __shfl_xor(ay, o); az += __shfl_xor(az, o); }
    if (i < n1 && l == 0) {
      const float inv = 1.f / (float)max(s1 - s0, 1);
      const float bx = base ? base[3 * i] : 0.f, by = base ? base[3 * i + 1] : 0.f, bz = base ? base[3 * i + 2] : 0.f;
      out[3 * i] = bx + ax * inv; out[3 * i + 1] = by + ay * inv; out[3 * i + 2] = bz + az * inv;
    }
  }
}

extern "C" int msde_mlp_head_mix_fwd(const float* Z, int ldz, const float* W, const float* b, int H, const float* basis,
                                     const int* rowptr, int N, const float* base, float* mix, float* out, void* stream) {
  if (N < 0 || !Z || !W || !basis || !rowptr || !mix || !out) return MSDE_EINVAL;
  if (!mh_ok(H, 3, Z, ldz, W)) return MSDE_EUNSUP;
  if (N == 0) return 0;
  const int lpr = pick_tpr(H / 4);
  MSDE_LAUNCH(mlp_head_mix_fwd_kernel, dim3((N + MHX_NPW - 1) / MHX_NPW), dim3(256), 0, as_stream(stream),
              reinterpret_cast<const float4*>(Z), ldz / 4, reinterpret_cast<const float4*>(W), b, H / 4, lpr, basis, rowptr, N,
              base, mix, out);
  MSDE_CHECK_LAUNCH();
  return 0;
}

// gnode [N, 3]: gradient of the mixed output; dst [E]: target node of each edge (by-target order, = the CSR of rowptr)
extern "C" int msde_mlp_head_mix_bwd(const float* Z, int ldz, const float* W, const float* gnode, const float* basis,
                                     const int* dst, const int* rowptr, int E, int H, float* gZ, float* workspace,
                                     const int* rows_dev, void* stream) {
  if (E < 0 || !Z || !W || !gnode || !basis || !dst || !rowptr || !gZ || !workspace) return MSDE_EINVAL;
  if (!mh_ok(H, 3, Z, ldz, W) || (reinterpret_cast<uintptr_t>(gZ) & 15)) return MSDE_EUNSUP;
  if (E == 0) return 0;
  const int lpr = pick_tpr(H / 4), nb = mh_grid(E, lpr);
  MSDE_LAUNCH(mlp_head_bwd_kernel<true>, dim3(nb), dim3(256), 0, as_stream(stream), reinterpret_cast<const float4*>(Z),
              ldz / 4, reinterpret_cast<const float4*>(W), gnode, E, rows_dev, H / 4, 3, lpr, reinterpret_cast<float4*>(gZ),
              workspace, basis, dst, rowptr);
  MSDE_CHECK_LAUNCH();
  return 0;
}

// ---- predictor-corrector sampler arithmetic (pretrain_MoleculeSDE_inference_2D_to_3D_VE_VP.py:163-168 ReverseDiffusionPredictor,
// :191-212 LangevinCorrector) for ONE diffusion time shared by every atom, as the sampler uses it: ~25 element-wise / reduction
// operators per half iteration become one single-workgroup kernel each (n atoms x 3 coordinates; the loop is launch bound).
// par = {std(t), G(t), alpha(t), fa(t)} on the DEVICE (a row of a table the caller computed once for all time steps with the
// SDE's own formulas): score = -out / std; drift of the discretised forward SDE f(x) = (fa - 1) x (VE: fa = 1).
__device__ __forceinline__ float pc_block_sum(float v, float* red) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int k = 0; k < nw; ++k) t += red[k];            // every thread adds the wave partials in wave order
  return t;
}

// par_all [S][4] = {std(t), G(t), alpha(t), 1 + f(x)/x} for every time step, row = the device iteration counter: the corrector
// (ONE workgroup) reads step[0] as its row and then advances it; the predictor of the same iteration reads step[0] - 1.  With
// step == NULL both use row 0 of `par_all` (the caller refreshes it).  noise == NULL: N(0,1) draws from the counter generator,
// (seed, row, element) -> the same draw wherever it is asked for (the corrector needs its noise twice).
__device__ __forceinline__ float pc_draw(const float* __restrict__ noise, unsigned long long seed, long long row, int stream_id, int i, int n3) {
  return noise ? noise[i] : msde_randn(seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(2 * row + stream_id), (unsigned long long)i);
}

__global__ void __launch_bounds__(1024)
pc_corrector_kernel(const float* __restrict__ out, const float* __restrict__ pos, const float* __restrict__ noise,
                    const float* __restrict__ par_all, long long* __restrict__ step, unsigned long long seed, int n, float snr,
                    float scale_eps, float* __restrict__ x, float* __restrict__ x_mean) {
  __shared__ float red[16];
  const long long row = step ? step[0] : 0;
  const float* par = par_all + 4 * row;
  const float inv_std = 1.f / par[0], alpha = par[2];
  float gs = 0.f, ns = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float g0 = -out[3 * i] * inv_std, g1 = -out[3 * i + 1] * inv_std, g2 = -out[3 * i + 2] * inv_std;
    const float z0 = pc_draw(noise, seed, row, 0, 3 * i, 3 * n), z1 = pc_draw(noise, seed, row, 0, 3 * i + 1, 3 * n),
                z2 = pc_draw(noise, seed, row, 0, 3 * i + 2, 3 * n);
    gs += sqrtf(g0 * g0 + g1 * g1 + g2 * g2);
    ns += sqrtf(z0 * z0 + z1 * z1 + z2 * z2);
  }
  const float gn = pc_block_sum(gs, red) / (float)n;
  const float nn = pc_block_sum(ns, red) / (float)n;
  const float r = snr * nn / gn;
  const float stp = r * r * 2.f * alpha;
  const float ns2 = sqrtf(stp * 2.f) * scale_eps;
  for (int i = threadIdx.x; i < 3 * n; i += blockDim.x) {
    const float g = -out[i] * inv_std;
    const float m = fmaf(stp, g, pos[i]);
    x_mean[i] = m;
    x[i] = fmaf(ns2, pc_draw(noise, seed, row, 0, i, 3 * n), m);
  }
  if (step && threadIdx.x == 0) step[0] = row + 1;      // (every thread read `row` before the block sums' barriers)
}

__global__ void __launch_bounds__(1024)
pc_predictor_kernel(const float* __restrict__ out, const float* __restrict__ pos, const float* __restrict__ noise,
                    const float* __restrict__ par_all, const long long* __restrict__ step, unsigned long long seed, int n,
                    float* __restrict__ x, float* __restrict__ x_mean) {
  const long long row = step ? step[0] - 1 : 0;
  const float* par = par_all + 4 * row;
  const float inv_std = 1.f / par[0], G = par[1], fa = par[3];
  const float g2 = G * G;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 3 * n; i += gridDim.x * blockDim.x) {
    const float score = -out[i] * inv_std;
    const float p = pos[i];
    const float m = p - ((fa - 1.f) * p - g2 * score);        // x - (f - G^2 score)
    x_mean[i] = m;
    x[i] = fmaf(G, pc_draw(noise, seed, row, 1, i, 3 * n), m);
  }
}

extern "C" int msde_pc_corrector(const float* out, const float* pos, const float* noise, const float* par, long long* step,
                                 unsigned long long seed, int n, float snr, float scale_eps, float* x, float* x_mean, void* stream) {
  if (n <= 0 || !out || !pos || !par || !x || !x_mean) return MSDE_EINVAL;
  MSDE_LAUNCH(pc_corrector_kernel, dim3(1), dim3(n >= 512 ? 1024 : 256), 0, as_stream(stream), out, pos, noise, par, step, seed, n,
              snr, scale_eps, x, x_mean);
  MSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int msde_pc_predictor(const float* out, const float* pos, const float* noise, const float* par, const long long* step,
                                 unsigned long long seed, int n, float* x, float* x_mean, void* stream) {
  if (n <= 0 || !out || !pos || !par || !x || !x_mean) return MSDE_EINVAL;
  const int blocks = (3 * n + 1023) / 1024;
  MSDE_LAUNCH(pc_predictor_kernel, dim3(blocks > 256 ? 256 : blocks), dim3(1024), 0, as_stream(stream), out, pos, noise, par, step,
              seed, n, x, x_mean);
  MSDE_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------ MD17 losses
// loss = ce mean_b |E_b - y_b| + cf mean_{i,k} |F_ik - f_ik| with F = fsign * dE (finetune_MD17.py:68-74: fsign = -1, the forces are
// minus the coordinate gradient) and, in the same launch, d loss / d E and d loss / d (dE) -- the seeds of the backward pass
// (torch's |.| differentiates to sign(.), sign(0) = 0).  One workgroup, fixed-order sums: bit-reproducible.
__global__ void __launch_bounds__(1024)
l1_energy_force_loss_kernel(const float* __restrict__ E, const float* __restrict__ y, int B, const float* __restrict__ dE,
                            const float* __restrict__ f, int n3, float fsign, float ce, float cf, float* __restrict__ loss,
                            float* __restrict__ gE, float* __restrict__ gdE) {
  __shared__ float part[2][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float se = 0.f, sf = 0.f;
  const float we = ce / (float)max(B, 1), wf = cf / (float)max(n3, 1);
  for (int i = tid; i < B; i += 1024) {
    const float r = E[i] - y[i];
    se += fabsf(r);
    gE[i] = we * (r > 0.f ? 1.f : (r < 0.f ? -1.f : 0.f));
  }
  for (int i = tid; i < n3; i += 1024) {
    const float r = fsign * dE[i] - f[i];
    sf += fabsf(r);
    gdE[i] = wf * fsign * (r > 0.f ? 1.f : (r < 0.f ? -1.f : 0.f));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { se += __shfl_down(se, o, 64); sf += __shfl_down(sf, o, 64); }
  if (lane == 0) { part[0][wave] = se; part[1][wave] = sf; }
  __syncthreads();
  if (tid == 0) {
    float a = 0.f, b = 0.f;
    for (int w = 0; w < 16; ++w) { a += part[0][w]; b += part[1][w]; }
    loss[0] = we * a + wf * b;
  }
}

extern "C" int msde_l1_energy_force_loss(const float* E, const float* y, int B, const float* dE, const float* f, int n3, float fsign,
                                         float ce, float cf, float* loss, float* gE, float* gdE, void* stream) {
  if (B <= 0 || n3 <= 0 || !E || !y || !dE || !f || !loss || !gE || !gdE) return MSDE_EINVAL;
  MSDE_LAUNCH(l1_energy_force_loss_kernel, dim3(1), dim3(1024), 0, as_stream(stream), E, y, B, dE, f, n3, fsign, ce, cf, loss, gE, gdE);
  MSDE_CHECK_LAUNCH();
  return 0;
}
