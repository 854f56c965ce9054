// plan.hip — batch construction on the GPU: everything index-shaped that the kernels of a pretrain step
// need, built from the RAW collated arrays of a mini-batch, into buffers of fixed CAPACITY (so that one captured hipGraph
// serves every batch; see msde_set_row_bound).
//
// Replaces, per batch: PyG's collate bookkeeping (App. A.8), `extend_graph` (Geom3D/datasets/dataset_3D.py:12-35:
// A ∪ A², then (·) ∪ (·)², self loops removed = all ordered pairs within <= 4 bonds -- a per-sample torch_sparse.spspmm
// on the CPU in the reference), and the host-side plan of round 1 (moleculesde_amd/plan.py: CSR by target + by-source
// view for bonds and extended edges, pre-offset OGB feature codes, per-table-row atom lists for the embedding backward).
// All outputs are bit-identical to plan.py's (tests/test_gpu_plan.py).
//
// Input (device, int32): x_raw [N_cap][K] atom feature codes, bond_src / bond_dst [Eb_cap] (batch-global atom indices,
// loader order, both directions listed), bond_attr [Eb_cap][3], mol_atoms [B], mol_bonds [B] (atoms / directed bonds
// per molecule; a molecule's atoms and bonds are contiguous).  n <= 32 atoms and <= 1024 directed bonds per molecule.
#include "msde_common.h"

#define PL_NMAX 32
#define PL_EMAX 1024

// inclusive block scan of up to 1024 x 4 ints (four values per thread, one pass of shuffles and barriers); returns the exclusive
// prefixes, totals in *total
__device__ __forceinline__ int4 pl_block_exscan4(int4 v, int4* sh4 /* [1024/64 + 1] */, int4* total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int4 x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int ya = __shfl_up(x.x, o, 64), yb = __shfl_up(x.y, o, 64), yc = __shfl_up(x.z, o, 64), yd = __shfl_up(x.w, o, 64);
    if (lane >= o) { x.x += ya; x.y += yb; x.z += yc; x.w += yd; }
  }
  if (lane == 63) sh4[w] = x;
  __syncthreads();
  if (threadIdx.x == 0) {
    int4 acc = make_int4(0, 0, 0, 0);
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) {
      const int4 t = sh4[k];
      sh4[k] = acc;
      acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
    }
    sh4[blockDim.x >> 6] = acc;
  }
  __syncthreads();
  const int4 b = sh4[w];
  *total = sh4[blockDim.x >> 6];
  return make_int4(b.x + x.x - v.x, b.y + x.y - v.y, b.z + x.z - v.z, b.w + x.w - v.w);
}

// sizes[]: 0 N, 1 E_b, 2 E_e, 3 P = sum n^2, 4 n_max, 5 sum n*min(n-1, max_nbr) (radius-graph edge bound), 6 2*E_e (valid
// rows of tensors holding two rows per extended edge)
// *err: cleared here, set (by this kernel or the per-molecule ones) when a molecule exceeds PL_NMAX / PL_EMAX or the batch
// exceeds a capacity.  The counts are SANITISED before anything is derived from them: a molecule beyond the limits counts as
// empty, and the batch is CUT at the first molecule with which ANY running total -- atoms (N_cap), bonds (Eb_cap), atom
// pairs sum n^2 (P_cap: the rows of the dense head's [P, *] arrays) or the radius-edge bound (Er_cap: the pair / edge
// buffers of the CFConv) -- would pass its capacity: that molecule and every later one count as empty.  All prefix sums are
// monotone, so the molecules that stay are a prefix of the batch and keep their offsets; mol_ptr / bond_ptr / pair_ptr stay
// monotone and inside the buffers whatever the raw blob holds, and every kernel that walks them (here and in the step) is
// memory safe.  The flagged batch is not a valid update, which the caller learns from *err (bucket.Bucket.poll_overflow).
__global__ void __launch_bounds__(1024)
plan_scan_kernel(const int* __restrict__ mol_atoms, const int* __restrict__ mol_bonds, int B, int max_nbr, int N_cap,
                 int Eb_cap, int P_cap, int Er_cap, int* __restrict__ mol_ptr /* [B+2] */, int* __restrict__ bond_ptr /* [B+1] */,
                 int* __restrict__ pair_ptr /* [B+1] */, int* __restrict__ sizes, int* __restrict__ err) {
  __shared__ int4 sh4[20];
  __shared__ int smax, sbad, scut;
  const int t = threadIdx.x;
  int n = t < B ? mol_atoms[t] : 0, m = t < B ? mol_bonds[t] : 0;
  if (t == 0) { smax = 0; sbad = 0; scut = B; }
  __syncthreads();
  if (n < 0 || n > PL_NMAX || m < 0 || m > PL_EMAX) { n = 0; m = 0; atomicExch(&sbad, 1); }
  const int nr = n * min(max(n - 1, 0), max_nbr);
  // the four running totals (atoms, bonds, atom pairs, radius-edge bound) in ONE block scan; its totals are the batch's when
  // nothing is cut below (round 5: four scans + four more for the totals were 12.8 us at the head of every step)
  int4 tot4;
  const int4 ex4 = pl_block_exscan4(make_int4(n, m, n * n, nr), sh4, &tot4);
  const int ex_n = ex4.x, ex_m = ex4.y, ex_p = ex4.z, ex_r = ex4.w;
  if (t < B && (ex_n + n > N_cap || ex_m + m > Eb_cap || ex_p + n * n > P_cap || ex_r + nr > Er_cap)) {
    atomicMin(&scut, t);
    atomicExch(&sbad, 1);
  }
  __syncthreads();
  const int cut = scut;
  if (t == cut || (t == 0 && cut == B)) {          // totals = the offsets of the first molecule that was cut (or of the end)
    const bool all = cut == B;
    // (for cut == B thread 0 holds only ITS offsets: the totals of the whole batch come from one more scan below)
    if (!all) {
      mol_ptr[B] = ex_n; mol_ptr[B + 1] = ex_n; sizes[0] = ex_n;
      bond_ptr[B] = ex_m; sizes[1] = ex_m;
      pair_ptr[B] = ex_p; sizes[3] = ex_p;
      sizes[5] = ex_r;
    }
  }
  if (t >= cut) { n = 0; m = 0; }
  // offsets: kept molecules keep theirs; cut ones collapse onto the cut molecule's (an empty range at the end of the valid rows)
  __shared__ int cn, cm, cp;
  if (t == cut) { cn = ex_n; cm = ex_m; cp = ex_p; }
  __syncthreads();
  if (t < B) {
    mol_ptr[t] = t < cut ? ex_n : cn;
    bond_ptr[t] = t < cut ? ex_m : cm;
    pair_ptr[t] = t < cut ? ex_p : cp;
  }
  if (cut == B) {                                  // nothing cut: the block totals
    const int tn = tot4.x, tm = tot4.y, tp = tot4.z, tr = tot4.w;
    if (t == 0) {
      mol_ptr[B] = tn; mol_ptr[B + 1] = tn; sizes[0] = tn;
      bond_ptr[B] = tm; sizes[1] = tm;
      pair_ptr[B] = tp; sizes[3] = tp;
      sizes[5] = tr;
    }
  }
  atomicMax(&smax, n);
  __syncthreads();
  if (t == 0) { sizes[4] = smax; *err = sbad; }
}


// One workgroup per molecule: atom arrays, bond CSR (by target, ties in loader order = torch.argsort(stable)) with its
// by-source view, bond feature codes, and the <= 4-bond neighbourhood rows (bit masks) + their count.
__global__ void __launch_bounds__(256)
plan_molecule_kernel(const int* __restrict__ x_raw, int K, const int* __restrict__ atom_off,
                     const int* __restrict__ bond_src, const int* __restrict__ bond_dst,
                     const int* __restrict__ bond_attr, const int* __restrict__ bond_off,
                     const int* __restrict__ mol_ptr, const int* __restrict__ bond_ptr,
                     int* __restrict__ batch_i32, int* __restrict__ atom_codes, int* __restrict__ z_codes,
                     int* __restrict__ rowptr, int* __restrict__ src, int* __restrict__ dst, int* __restrict__ rowptr_s,
                     int* __restrict__ perm_s, int* __restrict__ bond_codes, float* __restrict__ bond_type,
                     unsigned* __restrict__ ext_rows, int* __restrict__ ext_cnt, int* __restrict__ err, int B, int N_cap,
                     int Eb_cap) {
  __shared__ short ls[PL_EMAX], ld[PL_EMAX];      // local source / target of the loader-order bonds
  __shared__ short cs[PL_EMAX];                   // local source of the canonical-order bonds
  __shared__ unsigned A[PL_NMAX], P[PL_NMAX];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int a0 = mol_ptr[b], n = mol_ptr[b + 1] - a0;
  const int e0 = bond_ptr[b], m = bond_ptr[b + 1] - e0;
  if (n > PL_NMAX || m > PL_EMAX || n < 0 || m < 0 || a0 + n > N_cap || e0 + m > Eb_cap) {
    // unsupported molecule or a batch beyond the capacities: flag it and leave the rows that exist INERT (atoms of the empty
    // molecule B with code 0, no bonds, edge slots -1) -- never the previous batch's contents, never a write past a buffer
    if (tid == 0) { atomicExch(err, 1); ext_cnt[b] = 0; }
    for (int i = a0 + tid; i < min(a0 + max(n, 0), N_cap); i += 256) {
      batch_i32[i] = B; z_codes[i] = 0; ext_rows[i] = 0u;
      rowptr[i] = min(e0, Eb_cap); rowptr_s[i] = min(e0, Eb_cap);
      for (int k = 0; k < K; ++k) atom_codes[(size_t)i * K + k] = 0;
    }
    for (int e = e0 + tid; e < min(e0 + max(m, 0), Eb_cap); e += 256) {
      src[e] = -1; dst[e] = -1; perm_s[e] = e; bond_type[e] = 0.f;
      bond_codes[3 * e] = 0; bond_codes[3 * e + 1] = 0; bond_codes[3 * e + 2] = 0;
    }
    return;
  }
  for (int e = tid; e < n * K; e += 256) {
    const int i = e / K, k = e - i * K;
    atom_codes[(size_t)(a0 + i) * K + k] = x_raw[(size_t)(a0 + i) * K + k] + atom_off[k];
  }
  if (tid < n) { batch_i32[a0 + tid] = b; z_codes[a0 + tid] = x_raw[(size_t)(a0 + tid) * K]; A[tid] = 0u; }
  for (int e = tid; e < m; e += 256) {
    int s_ = bond_src[e0 + e] - a0, d_ = bond_dst[e0 + e] - a0;
    if (s_ < 0 || s_ >= n || d_ < 0 || d_ >= n) {   // an endpoint outside the molecule (inconsistent blob): keep the index valid
      atomicExch(err, 1);
      s_ = min(max(s_, 0), max(n - 1, 0));
      d_ = min(max(d_, 0), max(n - 1, 0));
    }
    ls[e] = (short)s_;
    ld[e] = (short)d_;
  }
  __syncthreads();
  // canonical order: stable sort by target
  for (int e = tid; e < m; e += 256) {
    const int d = ld[e];
    int r = 0;
    for (int f = 0; f < m; ++f) r += (ld[f] < d) || (ld[f] == d && f < e);
    const int k = e0 + r;
    src[k] = a0 + ls[e];
    dst[k] = a0 + d;
    cs[r] = ls[e];
    const int a = bond_attr[3 * (e0 + e)];
    bond_codes[3 * k] = a + bond_off[0];
    bond_codes[3 * k + 1] = bond_attr[3 * (e0 + e) + 1] + bond_off[1];
    bond_codes[3 * k + 2] = bond_attr[3 * (e0 + e) + 2] + bond_off[2];
    bond_type[k] = (float)a;
    if (ls[e] >= 0 && ls[e] < n && d >= 0 && d < n) atomicOr(&A[ls[e]], 1u << d);
  }
  if (tid < n) {
    int ct = 0, cs_ = 0;
    for (int f = 0; f < m; ++f) { ct += ld[f] < tid; cs_ += ls[f] < tid; }
    rowptr[a0 + tid] = e0 + ct;
    rowptr_s[a0 + tid] = e0 + cs_;
  }
  __syncthreads();
  // by-source view: slot -> canonical edge, stable in the canonical order
  for (int r = tid; r < m; r += 256) {
    const int s = cs[r];
    int slot = 0;
    for (int f = 0; f < m; ++f) slot += (cs[f] < s) || (cs[f] == s && f < r);
    perm_s[e0 + slot] = e0 + r;
  }
  // extend_graph: two rounds of  A <- A | (A·A minus the diagonal)   (dataset_3D.py:12-35)
  for (int round = 0; round < 2; ++round) {
    if (tid < n) {
      unsigned row = A[tid], acc = 0u;
      while (row) { const int k = __ffs(row) - 1; row &= row - 1; acc |= A[k]; }
      P[tid] = acc & ~(1u << tid);
    }
    __syncthreads();
    if (tid < n) A[tid] |= P[tid];
    __syncthreads();
  }
  if (tid < n) ext_rows[a0 + tid] = A[tid] & ~(1u << tid);       // "no self loops" (a diagonal can only come from A itself)
  if (tid == 0) {
    int c = 0;
    for (int i = 0; i < n; ++i) c += __popc(A[i] & ~(1u << i));
    ext_cnt[b] = c;
  }
}

// extended edges (row = source r, col = target c, loader order = row major): CSR by target + by-source view
__global__ void __launch_bounds__(64)
plan_ext_kernel(const unsigned* __restrict__ ext_rows, const int* __restrict__ mol_ptr, const int* __restrict__ ext_cnt, int B,
                int* __restrict__ ext_ptr, int* __restrict__ total_out, int* __restrict__ rowptr, int* __restrict__ src,
                int* __restrict__ dst, int* __restrict__ rowptr_s, int* __restrict__ perm_s, int N_cap, int Ee_cap,
                int* __restrict__ err) {
  __shared__ unsigned R[PL_NMAX], Cc[PL_NMAX];
  __shared__ int rp[PL_NMAX + 1], rs[PL_NMAX + 1];
  const int b = blockIdx.x, t = threadIdx.x;
  // offset of the molecule's extended edges = the counts of the molecules in front of it, summed by this (one-wave) workgroup
  // itself (B <= 1024 counts: 16 loads per lane at most) instead of by a scan launch between plan_molecule and this kernel
  int x0 = 0;
  for (int k = t; k < b; k += 64) x0 += ext_cnt[k];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x0 += __shfl_xor(x0, o, 64);
  const int x1 = x0 + ext_cnt[b];
  if (t == 0) {
    ext_ptr[b] = x0;
    if (b == B - 1) { ext_ptr[B] = x1; *total_out = x1; }
  }
  const int a0 = mol_ptr[b], n = mol_ptr[b + 1] - a0;
  if (n > PL_NMAX || n < 0 || a0 + n > N_cap || x1 > Ee_cap) {
    // (flagged molecules have no extended edges: only their row pointers are set, inside the buffer)
    if (x1 > Ee_cap && t == 0) atomicExch(err, 1);
    if (n > 0 && t < min(n, PL_NMAX) && a0 + t < N_cap) { rowptr[a0 + t] = min(x0, Ee_cap); rowptr_s[a0 + t] = min(x0, Ee_cap); }
    return;
  }
  if (t < n) R[t] = ext_rows[a0 + t];
  __syncthreads();
  if (t < n) {
    unsigned col = 0u;
    for (int r = 0; r < n; ++r) col |= ((R[r] >> t) & 1u) << r;
    Cc[t] = col;
  }
  __syncthreads();
  if (t == 0) {
    int a = x0, s = x0;
    for (int i = 0; i < n; ++i) { rp[i] = a; rs[i] = s; a += __popc(Cc[i]); s += __popc(R[i]); }
    rp[n] = a; rs[n] = s;
  }
  __syncthreads();
  if (t < n) {
    rowptr[a0 + t] = rp[t];
    rowptr_s[a0 + t] = rs[t];
    unsigned col = Cc[t];
    int k = rp[t];
    while (col) { const int r = __ffs(col) - 1; col &= col - 1; src[k] = a0 + r; dst[k] = a0 + t; ++k; }
    unsigned row = R[t];
    int slot = rs[t];
    while (row) {
      const int c = __ffs(row) - 1; row &= row - 1;
      perm_s[slot++] = rp[c] + __popc(Cc[c] & ((1u << t) - 1u));
    }
  }
}

// padded tails: atoms [N, N_cap) belong to the empty molecule B with code 0; row pointers past N hold the edge totals;
// edge slots past the totals hold src = dst = -1, perm_s = own index
__global__ void __launch_bounds__(256)
plan_tail_kernel(int* __restrict__ sizes, int B, int N_cap, int Eb_cap, int Ee_cap, int K, int* __restrict__ batch_i32,
                 int* __restrict__ atom_codes, int* __restrict__ z_codes, int* __restrict__ b_rowptr, int* __restrict__ b_src,
                 int* __restrict__ b_dst, int* __restrict__ b_rowptr_s, int* __restrict__ b_perm_s,
                 int* __restrict__ bond_codes, float* __restrict__ bond_type, int* __restrict__ e_rowptr,
                 int* __restrict__ e_src, int* __restrict__ e_dst, int* __restrict__ e_rowptr_s, int* __restrict__ e_perm_s) {
  const int N = min(sizes[0], N_cap), Eb = min(sizes[1], Eb_cap), Ee = min(sizes[2], Ee_cap);
  const int g = blockIdx.x * 256 + threadIdx.x, G = gridDim.x * 256;
  if (g == 0) sizes[6] = 2 * Ee;
  for (int i = N + g; i <= N_cap; i += G) {
    b_rowptr[i] = Eb; b_rowptr_s[i] = Eb; e_rowptr[i] = Ee; e_rowptr_s[i] = Ee;
    if (i < N_cap) {
      batch_i32[i] = B;
      z_codes[i] = 0;
      for (int k = 0; k < K; ++k) atom_codes[(size_t)i * K + k] = 0;
    }
  }
  for (int e = Eb + g; e < Eb_cap; e += G) {
    b_src[e] = -1; b_dst[e] = -1; b_perm_s[e] = e; bond_type[e] = 0.f;
    bond_codes[3 * e] = 0; bond_codes[3 * e + 1] = 0; bond_codes[3 * e + 2] = 0;
  }
  for (int e = Ee + g; e < Ee_cap; e += G) { e_src[e] = -1; e_dst[e] = -1; e_perm_s[e] = e; }
}

// per-table-row item lists of the embedding backward (plan.py::_row_lists): items = flat entries f = i*K + k of
// codes [N][K] (N from the device), list of row r = items with code r in ascending f, stored as the atom index i.
// (both kernels: one workgroup per table row; a lane takes 4 consecutive entries per 16-byte load and keeps PL_U of them in
// flight per trip -- the first versions walked the N * K entries 256 at a time with one 4-byte load per lane and trip, 126
// dependent trips (count: 20 us) with three barriers each in the fill (61 us at the head of the second stream))
#define PL_U 4
__device__ __forceinline__ int4 pl_load4(const int* __restrict__ codes, int f, int total, int vec) {   // f % 4 == 0; vec: codes 16-B aligned
  if (vec && f + 3 < total) return *reinterpret_cast<const int4*>(codes + f);
  return make_int4(f < total ? codes[f] : -1, f + 1 < total ? codes[f + 1] : -1, f + 2 < total ? codes[f + 2] : -1,
                   f + 3 < total ? codes[f + 3] : -1);
}

__global__ void __launch_bounds__(256)
plan_lists_count_kernel(const int* __restrict__ codes, const int* __restrict__ n_dev, int K, int* __restrict__ cnt, int vec) {
  __shared__ int red[4];
  const int r = blockIdx.x, total = n_dev[0] * K;
  int c = 0;
  for (int f0 = 4 * threadIdx.x; f0 < total; f0 += 1024 * PL_U) {
    int4 v[PL_U];
#pragma unroll
    for (int u = 0; u < PL_U; ++u) v[u] = pl_load4(codes, f0 + 1024 * u, total, vec);
#pragma unroll
    for (int u = 0; u < PL_U; ++u) c += (v[u].x == r) + (v[u].y == r) + (v[u].z == r) + (v[u].w == r);
  }
  c = (int)group_sum((float)c, 64);      // exact: counts < 2^24
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) cnt[r] = (red[0] + red[1]) + (red[2] + red[3]);
}

// wave w takes the w-th contiguous quarter of the entries (a multiple of 256): it counts its hits, the four counts meet once
// in LDS, and it then writes its part of the row's list on its own -- per block of 256 entries four ballots (one per
// component of the lanes' int4) give every hit its rank in ascending f; no barrier in the loop
__global__ void __launch_bounds__(256)
plan_lists_fill_kernel(const int* __restrict__ codes, const int* __restrict__ n_dev, int K, const int* __restrict__ cnt, int R,
                       int* __restrict__ ptr, int* __restrict__ items, int vec) {
  __shared__ int wcnt[4];
  __shared__ int wpre[4];
  const int r = blockIdx.x, total = n_dev[0] * K;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int per = ((total + 1023) / 1024) * 256;
  const int lo = w * per, hi = min(lo + per, total);
  int c = 0;
  for (int f0 = lo + 4 * lane; f0 < hi; f0 += 256 * PL_U) {
    int4 v[PL_U];
#pragma unroll
    for (int u = 0; u < PL_U; ++u) v[u] = pl_load4(codes, f0 + 256 * u, hi, vec);
#pragma unroll
    for (int u = 0; u < PL_U; ++u) c += (v[u].x == r) + (v[u].y == r) + (v[u].z == r) + (v[u].w == r);
  }
  c = (int)group_sum((float)c, 64);
  // start of the row's list = the counts of the rows in front of it (R <= 1024), summed here instead of by a scan launch
  // between the count and the fill; the row's own pointer (and the closing one) is written for the embedding backward
  int pre = 0;
  for (int k = threadIdx.x; k < r; k += 256) pre += cnt[k];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) pre += __shfl_xor(pre, o, 64);
  if (lane == 0) { wcnt[w] = c; wpre[w] = pre; }
  __syncthreads();
  int off = (wpre[0] + wpre[1]) + (wpre[2] + wpre[3]);
  if (threadIdx.x == 0) {
    ptr[r] = off;
    if (r == R - 1) ptr[R] = off + ((wcnt[0] + wcnt[1]) + (wcnt[2] + wcnt[3]));
  }
  for (int k = 0; k < w; ++k) off += wcnt[k];
  const unsigned long long below = (1ull << lane) - 1ull;
  for (int b0 = lo; b0 < hi; b0 += 256 * PL_U) {                  // (uniform trip count: every lane takes part in the ballots)
    int4 v[PL_U];
#pragma unroll
    for (int u = 0; u < PL_U; ++u) v[u] = pl_load4(codes, b0 + 256 * u + 4 * lane, hi, vec);
#pragma unroll
    for (int u = 0; u < PL_U; ++u) {
      const int f = b0 + 256 * u + 4 * lane;
      const bool h0 = v[u].x == r, h1 = v[u].y == r, h2 = v[u].z == r, h3 = v[u].w == r;
      const unsigned long long m0 = __ballot(h0), m1 = __ballot(h1), m2 = __ballot(h2), m3 = __ballot(h3);
      // hits of lower lanes (all four components) come first, then this lane's own components in order
      int pos = off + __popcll(m0 & below) + __popcll(m1 & below) + __popcll(m2 & below) + __popcll(m3 & below);
      if (h0) items[pos++] = f / K;
      if (h1) items[pos++] = (f + 1) / K;
      if (h2) items[pos++] = (f + 2) / K;
      if (h3) items[pos++] = (f + 3) / K;
      off += __popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3);
    }
  }
}

extern "C" int msde_plan_build(const int* x_raw, int K, const int* atom_off, const int* bond_src, const int* bond_dst,
                               const int* bond_attr, const int* bond_off, const int* mol_atoms, const int* mol_bonds,
                               int B, int N_cap, int Eb_cap, int Ee_cap, int P_cap, int Er_cap, int max_nbr, int* mol_ptr,
                               int* bond_ptr, int* pair_ptr, int* sizes, int* batch_i32, int* atom_codes, int* z_codes, int* b_rowptr,
                               int* b_src, int* b_dst, int* b_rowptr_s, int* b_perm_s, int* bond_codes, float* bond_type,
                               unsigned* ext_rows, int* ext_cnt, int* ext_ptr, int* e_rowptr, int* e_src, int* e_dst,
                               int* e_rowptr_s, int* e_perm_s, int* err, void* stream) {
  if (B <= 0 || B > 1024 || K <= 0 || N_cap <= 0 || P_cap < 0 || Er_cap < 0 || !x_raw || !atom_off || !bond_src || !bond_dst || !bond_attr ||
      !bond_off || !mol_atoms || !mol_bonds || !mol_ptr || !bond_ptr || !pair_ptr || !sizes || !batch_i32 || !atom_codes ||
      !z_codes || !b_rowptr || !b_src || !b_dst || !b_rowptr_s || !b_perm_s || !bond_codes || !bond_type || !ext_rows ||
      !ext_cnt || !ext_ptr || !e_rowptr || !e_src || !e_dst || !e_rowptr_s || !e_perm_s || !err)
    return B > 1024 ? MSDE_EUNSUP : MSDE_EINVAL;
  hipStream_t st = as_stream(stream);
  MSDE_LAUNCH(plan_scan_kernel, dim3(1), dim3(1024), 0, st, mol_atoms, mol_bonds, B, max_nbr, N_cap, Eb_cap, P_cap, Er_cap, mol_ptr,
              bond_ptr, pair_ptr, sizes, err);
  MSDE_CHECK_LAUNCH();
  MSDE_LAUNCH(plan_molecule_kernel, dim3(B), dim3(256), 0, st, x_raw, K, atom_off, bond_src, bond_dst, bond_attr, bond_off,
              (const int*)mol_ptr, (const int*)bond_ptr, batch_i32, atom_codes, z_codes, b_rowptr, b_src, b_dst, b_rowptr_s,
              b_perm_s, bond_codes, bond_type, ext_rows, ext_cnt, err, B, N_cap, Eb_cap);
  MSDE_CHECK_LAUNCH();
  MSDE_LAUNCH(plan_ext_kernel, dim3(B), dim3(64), 0, st, (const unsigned*)ext_rows, (const int*)mol_ptr,
              (const int*)ext_cnt, B, ext_ptr, sizes + 2, e_rowptr, e_src, e_dst, e_rowptr_s, e_perm_s, N_cap, Ee_cap, err);
  MSDE_CHECK_LAUNCH();
  int tail = (N_cap + Eb_cap + Ee_cap + 255) / 256;
  if (tail > 512) tail = 512;
  MSDE_LAUNCH(plan_tail_kernel, dim3(tail), dim3(256), 0, st, sizes, B, N_cap, Eb_cap, Ee_cap, K, batch_i32,
              atom_codes, z_codes, b_rowptr, b_src, b_dst, b_rowptr_s, b_perm_s, bond_codes, bond_type, e_rowptr, e_src, e_dst,
              e_rowptr_s, e_perm_s);
  MSDE_CHECK_LAUNCH();
  return 0;
}

extern "C" int msde_plan_row_lists(const int* codes, const int* n_dev, int K, int R, int* cnt, int* list_ptr, int* items,
                                   void* stream) {
  if (K <= 0 || R <= 0 || R > 1024 || !codes || !n_dev || !cnt || !list_ptr || !items) return R > 1024 ? MSDE_EUNSUP : MSDE_EINVAL;
  hipStream_t st = as_stream(stream);
  const int vec = (reinterpret_cast<uintptr_t>(codes) & 15) == 0;
  MSDE_LAUNCH(plan_lists_count_kernel, dim3(R), dim3(256), 0, st, codes, n_dev, K, cnt, vec);
  MSDE_CHECK_LAUNCH();
  MSDE_LAUNCH(plan_lists_fill_kernel, dim3(R), dim3(256), 0, st, codes, n_dev, K, (const int*)cnt, R, list_ptr, items, vec);
  MSDE_CHECK_LAUNCH();
  return 0;
}
