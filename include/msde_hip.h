/* msde_hip.h — C ABI of libmsde_hip.so, the MI355X (gfx950) kernels behind the MoleculeSDE
 * pretrain hot path.
 *
 * The reference (chao1224/MoleculeSDE) has NO FFI/plugin layer of its own: its hot path reaches
 * native code only through third-party wheels (PyG / torch-scatter / torch-cluster).  Each entry
 * point below therefore cites the *reference call site* (path relative to the reference root) whose
 * native work it replaces.  INTEGRATION.md shows the ctypes stub a reference maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success, a positive hipError_t on a HIP failure, or a negative
 *     MSDE_E* code on an argument error; nothing throws, nothing synchronises, nothing allocates;
 *   - the caller owns every buffer (device pointers), all fp32 row-major contiguous, indices int32;
 *   - kernels are enqueued on `stream` (a hipStream_t passed as void*), safe for hipGraph capture;
 *   - graphs are CSR sorted by TARGET node: rowptr[N+1], src[E] ("canonical edge order"); the
 *     transposed view is rowptr_s[N+1] + perm_s[E] (by-source slot -> canonical edge id);
 *   - E may be an upper bound: kernels that walk rowptr never touch the padded tail.
 */
#ifndef MSDE_HIP_H
#define MSDE_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define MSDE_EINVAL (-1) /* bad size / null pointer */
#define MSDE_EUNSUP (-2) /* unsupported shape for this kernel */

/* library identification: returns the ABI version (monotonic integer) */
int msde_abi_version(void);
/* name of the code-object target the library was built for ("gfx950") */
const char* msde_target_arch(void);

/* ------------------------------------------------------------------ graph construction ----- */
/* torch_cluster.radius via PyG radius_graph — Geom3D/models/schnet.py:91.
 * count: deg[i] = #{j in molecule(i), j != i, |p_i-p_j|^2 < r2}, capped at max_nbr (index order). */
int msde_radius_count(const float* pos, const int* batch, const int* mol_ptr, int N, float r2,
                      int max_nbr, int* deg, void* stream);
/* exclusive scan: out[0]=0, out[i+1]=out[i]+in[i]  (n <= 2^24; single workgroup, LDS scan) */
int msde_exclusive_scan_i32(const int* in, int* out, int n, void* stream);
/* fill: src[rowptr[i]+k] = k-th neighbour of i (index order), dst[..]=i, dist[..]=|p_i-p_j|
 * (schnet.py:92-93).  Slots in [rowptr[N], E_cap) get src=dst=N_sentinel(-1), dist=0. */
int msde_radius_fill(const float* pos, const int* batch, const int* mol_ptr, int N, float r2,
                     int max_nbr, const int* rowptr, int* src, int* dst, float* dist, int E_cap,
                     void* stream);

/* Row bounds: one captured hipGraph for batches of different sizes.  Tensors are allocated with row CAPACITIES (atoms,
 * bonds, extended edges, ... padded); the TRUE row count of a batch stays on the device.  Every entry point that REDUCES
 * over rows (BatchNorm statistics, weight / bias gradients, column sums, the contrastive loss and its permutation,
 * LayerNorm-parameter and embedding / bond-table gradients, the statistics epilogues of msde_gemm_rs / msde_gemm_t2) takes
 * the device address of that count as an explicit argument (`rows_dev`, `m_valid`, `n_dev` ...; NULL = all rows) and stops
 * at the valid rows; row-wise kernels keep processing all rows (padded rows hold finite values and zero gradients) and CSR
 * walks never reach padded edges.  The library keeps NO table and no other global mutable state: two buckets with equal
 * capacities, or two threads, cannot interfere.  The reference has no counterpart: PyG's collate rebuilds exact-size
 * tensors on the host for every batch (Geom3D/datasets/dataset_3D.py:114-122, App. A.8). */

/* ------------------------------------------------------------------ batch construction on the GPU --- */
/* Everything index-shaped a pretrain step needs, from the RAW collated arrays of a mini-batch, into buffers of fixed
 * capacity (csrc/plan.hip).  Replaces PyG's collate bookkeeping (App. A.8), `extend_graph`
 * (Geom3D/datasets/dataset_3D.py:12-35: all ordered pairs within <= 4 bonds; a per-sample CPU spspmm in the reference)
 * and the per-batch host plan.  Inputs (int32, device): x_raw [N_cap][K] atom feature codes, bond_src / bond_dst
 * [Eb_cap] (batch-global atom indices, loader order), bond_attr [Eb_cap][3], mol_atoms / mol_bonds [B] (per-molecule
 * counts; a molecule's atoms and bonds are contiguous; <= 32 atoms, <= 1024 directed bonds each; B <= 1024),
 * atom_off [K] / bond_off [3] (offsets of the concatenated OGB embedding tables).  Outputs: mol_ptr [B+2] (entry B+1
 * repeats N: padded atoms belong to the empty molecule B), bond_ptr / pair_ptr [B+1], sizes [8] = {N, E_b, E_e,
 * sum n^2, n_max, radius-edge bound sum n*min(n-1, max_nbr), -, -}, batch_i32 / z_codes [N_cap], atom_codes [N_cap][K],
 * the bond graph as CSR by target (b_rowptr [N_cap+1], b_src, b_dst [Eb_cap]; canonical order = stable sort by target)
 * with its by-source view (b_rowptr_s, b_perm_s) and canonical-order bond_codes [Eb_cap][3] / bond_type [Eb_cap], the
 * extended graph the same way (e_*, Ee_cap), and scratch ext_rows [N_cap] / ext_cnt [B] / ext_ptr [B+1].  Padded tails:
 * see "Row bounds" above.  *err is set to 1 if a molecule exceeds the limits or the batch one of the capacities N_cap,
 * Eb_cap, Ee_cap, P_cap (atom pairs, sum n^2: rows of the dense head's pair arrays) or Er_cap (radius-edge bound sum
 * n * min(n - 1, max_nbr): pair / edge buffers of the CFConv); the batch is then cut in front of the first molecule that
 * does not fit (it and all later molecules count as empty), so every offset stays inside its buffer. */
int msde_plan_build(const int* x_raw, int K, const int* atom_off, const int* bond_src, const int* bond_dst,
                    const int* bond_attr, const int* bond_off, const int* mol_atoms, const int* mol_bonds,
                    int B, int N_cap, int Eb_cap, int Ee_cap, int P_cap, int Er_cap, int max_nbr, int* mol_ptr,
                    int* bond_ptr, int* pair_ptr, int* sizes, int* batch_i32, int* atom_codes, int* z_codes, int* b_rowptr,
                    int* b_src, int* b_dst, int* b_rowptr_s, int* b_perm_s, int* bond_codes, float* bond_type,
                    unsigned* ext_rows, int* ext_cnt, int* ext_ptr, int* e_rowptr, int* e_src, int* e_dst,
                    int* e_rowptr_s, int* e_perm_s, int* err, void* stream);
/* Per-table-row atom lists of the embedding backward (msde_embedding_sum_bwd): for codes [*n_dev][K] with values in
 * [0, R): list_ptr [R+1], items = atom index of every entry with that code, in ascending (atom, column) order; cnt [R]
 * is scratch. */
int msde_plan_row_lists(const int* codes, const int* n_dev, int K, int R, int* cnt, int* list_ptr, int* items,
                        void* stream);

/* ------------------------------------------------------------------ twice-differentiable force path -------- */
/* Non-GEMM members of the closed operator set of the MD17 energy/force path (moleculesde_amd/dd.py; replaces the ATen
 * operators autograd differentiates twice in examples/finetune_MD17.py:47-78 over Geom3D/models/schnet.py:85-125).
 * kind 0: shifted softplus (schnet.py:213-216), 1: cosine cutoff with p0 = cutoff (schnet.py:186; 0 for x >= p0,
 * painn_utils.py:150-154), 2: reciprocal, 3: SiLU (painn.py activation), 4: sqrt(x + p0) (painn.py:104);
 * order = derivative order 0..2.  mask (may be NULL): entries < 0 give 0. */
int msde_dd_unary(const float* x, const int* mask, long long n, int kind, int order, float p0, float* y, void* stream);
/* y = g * (g2 ? g2 : 1) * f^(order)(x) (same kinds): the backward of msde_dd_unary, and of itself, in one launch
 * (moleculesde_amd/dd.py: _UnaryMul; autograd's chain rule for Softplus / cos in schnet.py:186,213-216). */
int msde_dd_unary_mul(const float* g, const float* g2, const float* x, const int* mask, long long n, int kind, int order,
                      float p0, float* y, void* stream);
/* Gaussian smearing exp(coeff (d - mu_g)^2) (schnet.py:205-207) and its 1st / 2nd derivative in d: y [E][G];
 * rows with src[e] < 0 (src may be NULL) are zero. */
int msde_dd_rbf(const float* d, const int* src, const float* mu, int E, int G, float coeff, int order, float* y,
                void* stream);
/* op 0: y = alpha a b, 1: y = a + b, 2: y = alpha a (b unused) */
int msde_dd_binary(const float* a, const float* b, long long n, int op, float alpha, float* y, void* stream);
/* y = srcs[0] + ... + srcs[n-1] elementwise (n <= 8 device pointers in a HOST array, `count` floats each, summed in index
 * order): the gradient autograd accumulates for a tensor with n consumers (finetune_MD17.py:68,76 -- the smeared distances and
 * the cutoff feed all six interaction blocks, schnet.py:185-195) as one launch instead of n - 1 additions. */
int msde_dd_sum_n(const float* const* srcs, int n, long long count, float* y, void* stream);
/* The same for [rows][cols] operands with row strides lds[k] (host array; cols % 4 == 0, rows 16-byte aligned; y contiguous): a
 * consumer's gradient that is a column block of a wider buffer -- the edge half of the basis MLP's input gradient
 * (equivariant_scorenetwork.py:154-157: cat([h_row + h_col, edge_attr])) -- is summed where it lies. */
int msde_dd_sum_rows_n(const float* const* srcs, const int* lds, int n, int rows, int cols, float* y, void* stream);
/* y[e][k] = M[e][k] s[e]   and   y[e] = sum_k a[e][k] b[e][k] (fixed lane order) */
int msde_dd_mul_rows(const float* M, const float* s, int E, int K, float* y, void* stream);
int msde_dd_row_dot(const float* a, const float* b, int E, int K, float* y, void* stream);
/* y[e] = pos[src_e] - pos[dst_e] (schnet.py:98-99; zero on padded slots) and its adjoint over the by-target CSR
 * (rowptr) and its by-source view (rowptr_s, perm_s): y[i] = sum_{src_e = i} g[e] - sum_{dst_e = i} g[e]. */
int msde_dd_edge_diff(const float* pos, const int* src, const int* dst, int E, float* y, void* stream);
int msde_dd_edge_scatter(const float* g, const int* rowptr, const int* rowptr_s, const int* perm_s, int N, float* y,
                         void* stream);
/* y[e] = |v[e]| over 3 coordinates; 1 where src[e] < 0 */
int msde_dd_row_norm(const float* v, const int* src, int E, float* y, void* stream);
/* adjoint of the per-molecule readout (schnet.py:122): y[i] = g[batch[i]] (/ atoms of that molecule if mean) */
int msde_dd_seg_expand(const float* g, const int* batch, const int* mol_ptr, int N, int K, int mean, float* y,
                       void* stream);
/* per-edge 3-vectors [E][3] -> component-major [3][E] (to_soa = 1) and back (0): adjoint pair */
int msde_dd_transpose3(const float* x, int E, int to_soa, float* y, void* stream);
/* y[e] = (a[e], b[e], c[e]); a NULL component reads as zeros */
int msde_dd_merge3(const float* a, const float* b, const float* c, int E, float* y, void* stream);
/* adjoint of msde_colsum: y[m][k] = b[k] */
int msde_dd_broadcast_rows(const float* b, int M, int K, float* y, void* stream);

/* loss = ca *a + cb *b + cc *c + cd *d over device scalars (NULL terms are skipped) and its backward
 * out4[i] = *g * c_i: the loss composition of examples/pretrain_MoleculeSDE.py:139-152 as one launch each way. */
int msde_combine_losses(const float* a, const float* b, const float* c, const float* d, float ca, float cb, float cc,
                        float cd, float* out, void* stream);
int msde_combine_losses_bwd(const float* g, float ca, float cb, float cc, float cd, float* out4, void* stream);
/* msde_combine_losses with two riders in the same launch: seeds4[i] = c_i (= the backward's out4 for a unit upstream
 * gradient: no backward launch then), and log_dst[k][0] += log_src[k][0] for k < n_log <= 5 (HOST arrays of device
 * pointers): the per-term running sums of pretrain_MoleculeSDE.py:158-163. */
int msde_combine_losses_ex(const float* a, const float* b, const float* c, const float* d, float ca, float cb, float cc,
                           float cd, float* out, float* seeds4, const float* const* log_src, float* const* log_dst,
                           int n_log, void* stream);

/* Diagnostics: store the 100 MHz real-time counter into *slot, in stream order (capturable). */
int msde_debug_stamp(long long* slot, void* stream);

/* ------------------------------------------------------------------ generic row ops -------- */
/* torch_scatter.scatter(reduce=sum) over CSR rows: out[i] = sum_{s in [rowptr[i],rowptr[i+1])}
 * rows[perm ? perm[s] : s]; used for every backward "gather by source/target".  D % 4 == 0 or any. */
int msde_segment_sum_rows(const float* rows, int ldi /* row stride of rows (0 = D): a column block of a
                                                          wider gradient can be summed without a copy */,
                          const int* rowptr, const int* perm, int N, int D,
                          float scale_by_inv_count, float* out, int ldo /* row stride of out (0 = D) */,
                          void* stream);
/* the same with a SECOND CSR view (rowptr2, perm2; NULL = none) of the same rows summed into the same output row: the
 * gradient of out[e] = x[src_e] + x[dst_e] (by-source plus by-target segments) in one pass. */
int msde_segment_sum_rows2(const float* rows, int ldi, const int* rowptr, const int* perm, const int* rowptr2,
                           const int* perm2, int N, int D, float scale_by_inv_count, float* out, int ldo,
                           void* stream);
/* out[e] = A[src[e]] + B[dst[e]] for e < E; rows with src<0 are zero-filled.  A and B have row stride ld
 * floats (0 = D): they may be column blocks of one [N, 2D] GEMM result.
 * SDE_model_2D_to_3D.py:346-347 (factored cat+Linear), equivariant_scorenetwork.py:154-155 */
int msde_pair_gather_add(const float* A, const float* B, int ld, const int* src, const int* dst, int E,
                         int D, float* out, void* stream);
/* edge_2D_emb of the 2D->3D model fused around the gather (SDE_model_2D_to_3D.py:35-40,264-271; hip._PairBnReluLinear):
 * msde_pair_gather_add_stats = msde_pair_gather_add that also writes the BatchNorm statistics of its result per strip of
 * msde_pair_strip() edges, stats[strip][2][D] in the MSDE_RS_STATS_BNFWD format over the valid edges (e < *e_valid, or E) -- finish with
 * msde_bn_fin_fwd(stats, ceil(E / strip), strip, ...).  msde_segment_sum_rows_bn: out[i] = p * sum G[e] + w * sum Z[e] + n_i * u
 * over node i's CSR segment (n_i edges; perm = edge ids or NULL): the segment sum of the BatchNorm input gradient
 * p G + w Z + u (vectors of msde_bn_fin_bwd) without materialising it.  D % 4 == 0, 16-byte aligned; else MSDE_EUNSUP. */
int msde_pair_gather_add_stats(const float* A, const float* B, int ld, const int* src, const int* dst, int E, int D,
                               const int* e_valid, float* out, float* stats, void* stream);
int msde_segment_sum_rows_bn(const float* G, const float* Z, int ld, const int* rowptr, const int* perm, int N, int D,
                             const float* p, const float* w, const float* u, float* out, int ldo, void* stream);
/* The backward half of the same fusion.  msde_pair_bn_dgrad_stats: GA[e] = (g[e] W) gated where scale z + shift <= 0
 * (g [E, H], H = 16 or 32, row stride ldg; W [H][D] as nn.Linear stores the second Linear's weight; z = Z[e], the gathered
 * sums) together with the strip sums (sum GA, sum GA (z - mean)) of MSDE_RS_STATS_BNBWD over strips of msde_pair_strip()
 * edges -> msde_bn_fin_bwd.  msde_pair_bn_scatter: gAB [N, 2 D] = gradient of AB through dz = p GA + w z + u, one launch
 * for both column blocks (by-source and by-target segments; sum z over a segment is rebuilt from AB's own rows). */
int msde_pair_strip(void);
int msde_pair_bn_dgrad_stats(const float* g, int ldg, const float* W, const float* Z, const float* scale, const float* shift,
                             const float* mean, int E, int H, int D, const int* e_valid, float* GA, float* stats,
                             void* stream);
int msde_pair_bn_scatter(const float* GA, const float* AB, int D, const int* src, const int* dst, const int* rowptr_s,
                         const int* perm_s, const int* rowptr, int N, const float* p, const float* w, const float* u,
                         float* gAB, void* stream);
/* out[e] = [X[src[e]] + X[dst[e]] | C[e]] ([E, D + D2], rows with src<0 zero): the gather writes the concatenation
 * cat([h_row + h_col, edge_attr]) that the basis MLP reads (equivariant_scorenetwork.py:154-157).  Row strides ldx,
 * ldc in floats; D, D2, ldx, ldc multiples of 4 and 16-byte aligned bases (else MSDE_EUNSUP). */
int msde_pair_gather_cat(const float* X, int ldx, const float* C, int ldc, const int* src, const int* dst, int E,
                         int D, int D2, float* out, void* stream);
/* out[e] = X[idx[e]] (row gather), idx<0 -> zeros */
int msde_gather_rows(const float* X, const int* idx, int E, int D, float* out, void* stream);

/* By-source (transposed) view of the radius graph for the atomic-free input gradient of CFConv
 * (the adjoint of schnet.py:141-145's scatter): rowptr_s[N+1], perm_s[E_cap] = edge positions grouped by
 * source, canonical order inside a group (a stable counting sort; padded slots keep their own index).
 * Uses that edges stay inside a molecule and rows list sources in ascending order.  deg_s: N ints scratch. */
int msde_radius_transpose(const int* batch, const int* mol_ptr, const int* rowptr, const int* src, int N,
                          int E_cap, int* deg_s, int* rowptr_s, int* perm_s, void* stream);
/* The same view in ONE launch, one workgroup per molecule, for batches whose molecules have at most 64 atoms (n_max: the
 * host's bound; larger: MSDE_EUNSUP -- use msde_radius_transpose).  Identical rowptr_s / perm_s.  mol_ptr [B+1]. */
int msde_radius_transpose_mol(const int* mol_ptr, int B, int n_max, const int* rowptr, const int* src, int N, int E_cap,
                              int* rowptr_s, int* perm_s, void* stream);

/* ------------------------------------------------------------------ embeddings ------------- */
/* ogb AtomEncoder/BondEncoder, nn.Embedding — molecule_gnn_model.py:171, schnet.py:89.
 * codes[i*K+k] already offset into the concatenated table tab[R,D]. */
int msde_embedding_sum_fwd(const float* tab, const int* codes, int N, int K, int D, float* out,
                           void* stream);
/* g_tab[r] = sum over list(r) of g[node]; lists given as CSR (list_ptr[R+1], list_nodes[..]).
 * Every element of g_tab is written (no zero-init needed).  A row's list is cut into at most SPLIT slices
 * whose partial sums are combined in slice order: deterministic, no atomics.  workspace:
 * msde_embedding_sum_bwd_workspace_floats(R, D, split) floats. */
long long msde_embedding_sum_bwd_workspace_floats(int R, int D, int split);
int msde_embedding_sum_bwd(const float* g, const int* list_ptr, const int* list_nodes, int R,
                           int D, int split, float* g_tab, float* workspace, void* stream);

/* ------------------------------------------------------------------ GIN -------------------- */
/* GINConv.forward/message — molecule_gnn_model.py:22-29:
 * out[i] = (1+eps)*x[i] + sum_{e in in(i)} relu(x[src[e]] + sum_k tab[codes[e*3+k]]) */
int msde_gin_aggregate_fwd(const float* x, const float* tab, const int* codes, const float* eps,
                           const int* rowptr, const int* src, int N, int D, float* out,
                           void* stream);
/* g_x[j] = (1+eps)*g[j] + sum_{e in out(j)} g[dst[e]] * [x[j]+emb_e > 0]   (by-source CSR) */
int msde_gin_aggregate_bwd_x(const float* g, const float* x, const float* tab, const int* codes,
                             const float* eps, const int* rowptr_s, const int* perm_s,
                             const int* dst, int N, int D, float* g_x, void* stream);
/* g_tab[code] = sum_e g[dst[e]] * [x[src[e]] + emb_e > 0] over the E edges (canonical by-target order, codes
 * [E,3]); g_eps[0] = sum_i g[i].x[i].  Both outputs are fully written (no zero-init); per-workgroup partial
 * tables are summed in a fixed order (deterministic, no atomics).  2*R*D*4 bytes must fit LDS (<= 64 KiB).
 * workspace: msde_gin_aggregate_bwd_tab_workspace_floats(N, E, D, R) floats. */
long long msde_gin_aggregate_bwd_tab_workspace_floats(int N, int E, int D, int R);
/* msde_gin_aggregate_fwd on the layer input h = max(z scale[c] + shift[c], 0 if relu): the outer BatchNorm (+ ReLU) of the
 * previous GIN layer (molecule_gnn_model.py:176-182; scale / shift from msde_bn_fin_fwd) applied on the fly to the gathered
 * rows; h itself is written to h_out.  D % 4 == 0. */
int msde_gin_aggregate_bn_fwd(const float* z, const float* scale, const float* shift, int relu, const float* tab,
                              const int* codes, const float* eps, const int* rowptr, const int* src, int N, int D,
                              float* h_out, float* out, void* stream);
/* msde_gin_aggregate_bwd_x that also emits the BatchNorm-backward partial sums of its result for the previous layer's outer
 * BatchNorm: stats [ceil(N/16)][2][D] = per 16-row strip (sum g', sum g' (zprev - mean)), g' = g_x gated by x > 0 when
 * relu; rows >= *m_valid contribute nothing (input of msde_bn_fin_bwd).  D % 4 == 0. */
int msde_gin_aggregate_bwd_x_stats(const float* g, const float* x, const float* tab, const int* codes, const float* eps,
                                   const int* rowptr_s, const int* perm_s, const int* dst, int N, const int* m_valid,
                                   int D, const float* zprev, const float* mean, int relu, float* g_x, float* stats,
                                   void* stream);
/* With g_tab == g_eps == NULL the call only leaves msde_gin_aggregate_bwd_tab_slabs(N,E) partial tables
 * [slabs][R*D] followed by [slabs] eps partials in `workspace` (for msde_reduce_slabs_multi). */
int msde_gin_aggregate_bwd_tab_slabs(int N, int E);
int msde_gin_aggregate_bwd_tab(const float* g, const float* x, const float* tab, const int* codes,
                               const int* src, const int* dst, int N, int E, int D, int R,
                               float* g_tab, float* g_eps, float* workspace, const int* n_dev, const int* e_dev,
                               void* stream);

/* the partial tables of `layers` <= 8 GIN layers over the SAME graph in ONE launch (g / x / tab / workspace: HOST arrays of
 * device pointers, each workspace as for msde_gin_aggregate_bwd_tab with g_tab == g_eps == NULL): independent leaf work that
 * would otherwise be `layers` launches in a row, each draining before the next starts. */
int msde_gin_aggregate_bwd_tab_multi(const float* const* g, const float* const* x, const float* const* tab,
                                     float* const* workspace, int layers, const int* codes, const int* src,
                                     const int* dst, int N, int E, int D, int R, const int* n_dev, const int* e_dev,
                                     void* stream);

/* ------------------------------------------------------------------ SchNet ----------------- */
/* GaussianSmearing + cosine cutoff — schnet.py:186,205-207: rbf[e,g]=exp(coeff*(d-offset[g])^2),
 * C[e]=0.5(cos(d*pi/cutoff)+1); padded rows (e >= E_dev[0]) are zero. */
int msde_rbf_cutoff_fwd(const float* dist, const int* E_dev, int E_cap, int G,
                        const float* offset, float coeff, float cutoff, float* rbf, float* C,
                        void* stream);
/* CFConv.message + aggregate — schnet.py:190,194-195: agg[i] = sum_e x1[src[e]] * Wf[e] * C[e]
 * (C may be NULL = 1 in all three aggregate kernels) */
int msde_cfconv_aggregate_fwd(const float* x1, const float* Wf, const float* C, const int* rowptr,
                              const int* src, int N, int F, float* agg, void* stream);
/* g_Wf[e] = g_agg[dst[e]] * x1[src[e]] * C[e]   (rows e >= rowptr[N] zero-filled up to E_cap) */
int msde_cfconv_aggregate_bwd_w(const float* g_agg, const float* x1, const float* C,
                                const int* rowptr, const int* src, int N, int F, int E_cap,
                                float* g_Wf, void* stream);
/* g_x1[j] = sum_{e in out(j)} g_agg[dst[e]] * Wf[e] * C[e]   (C may be NULL: Wf already holds Wf*C) */
int msde_cfconv_aggregate_bwd_x(const float* g_agg, const float* Wf, const float* C,
                                const int* rowptr_s, const int* perm_s, const int* dst, int N,
                                int F, float* g_x1, void* stream);
/* Fused CFConv forward: RBF + filter MLP (Linear(G,F) -> ShiftedSoftplus -> Linear(F,F)) + cutoff +
 * gather + segmented sum in one fp32-MFMA kernel — schnet.py:141-145,185-195.  F must be 128, G <= 64.
 * Weights in the nn.Linear layout of the reference's `mlp` (W1 [F,G], W2 [F,F], b1 [F], b2 [F]): no
 * transposed copies.  E_cap bounds the edge count (true count = rowptr[N]).
 * Wf_out (may be NULL) receives the filter rows (W2 h1 + b2) * C(d) [E_cap,F] for the backward. */
int msde_cfconv_fused_fwd(const float* x1, const float* dist, const int* rowptr, const int* src,
                          const int* dst, const float* W1, const float* b1, const float* W2,
                          const float* b2, const float* offset, int N, int F, int G, int E_cap,
                          float coeff, float cutoff,
                          int chunks_per_wg /* 32-edge chunks per persistent workgroup; <= 0: two workgroups
                                               per CU.  A caller running this kernel beside latency-critical
                                               work on another stream asks for fewer, longer workgroups. */,
                          float* agg, float* Wf_out, void* stream);
/* Weight gradients of the filter network for the fused CFConv, recomputing rbf/h1 on chip:
 * gW1 [F,G], gb1 [F], gW2 [F,F], gb2 [F] from g_agg [N,F], x1 [N,F], dist.  Per-workgroup slabs in
 * `workspace` (msde_cfconv_fused_bwd_w_workspace_floats floats) are summed in a fixed order.
 * max_workgroups <= 0: one persistent workgroup per CU (each holds 147 KB of LDS); smaller values leave whole
 * CUs to work running concurrently on other streams (the kernel itself then takes longer). */
long long msde_cfconv_fused_bwd_w_workspace_floats(int E_cap, int G, int max_workgroups);
int msde_cfconv_fused_bwd_w(const float* g_agg, const float* x1, const float* dist,
                            const int* rowptr, const int* src, const int* dst, const float* W1,
                            const float* b1, const float* W2, const float* offset, int N, int F,
                            int G, int E_cap, float coeff, float cutoff, int max_workgroups,
                            float* gW1, float* gb1, float* gW2, float* gb2, float* workspace,
                            void* stream);
/* With gW1 == gb1 == gW2 == gb2 == NULL the call only leaves msde_cfconv_fused_bwd_w_slabs(E_cap, max_workgroups) slabs of
 * F*F + F*G + 2F floats ([gW2 | gW1 | gb1 | gb2]) in `workspace` (for msde_reduce_slabs_multi). */
int msde_cfconv_fused_bwd_w_slabs(int E_cap, int max_workgroups);

/* ------------------------------------------------------------------ 2D->3D score net ------- */
/* coord2basis / get_perturb_distance / GaussianFourierProjection / pseudo-angle —
 * SDE_model_2D_to_3D.py:35-66,342-369.  Outputs per edge (canonical order):
 * feat_d [E,2C] = [sin,cos](2pi d Wd); feat_i/feat_j [E,4C] = Fourier of frame coords 0 and 2 of
 * r_i (source) / r_j (target); angle [E,2] = (pseudo_sin, pseudo_cos); basis [E,9]. */
int msde_edge_geometry_fwd(const float* pos, const int* src, const int* dst, int E,
                           const float* Wd, const float* Wc, int C, float* feat_d, float* feat_i,
                           float* feat_j, float* angle, float* basis, void* stream);
/* the same with strided outputs: feat_i / feat_j rows at stride feat_ld (>= 4C: e.g. the two halves of one [E, 2, 4C]
 * buffer, so that a layer shared by both is ONE product over 2E rows); angle_ld >= 4: (pseudo_sin, pseudo_cos, 0, 0) is
 * written to angle[e * angle_ld ..+3] and, when angle_zero_off > 0, four zeros to angle[e * angle_ld + angle_zero_off ..+3]
 * -- column blocks of the buffer the `project` MLP reads (SDE_model_2D_to_3D.py:369-370), so no torch.cat. */
int msde_edge_geometry_fwd_ld(const float* pos, const int* src, const int* dst, int E,
                              const float* Wd, const float* Wc, int C, float* feat_d, float* feat_i,
                              float* feat_j, int feat_ld, float* angle, int angle_ld, int angle_zero_off, float* basis,
                              void* stream);
/* PyG TransformerConv message+softmax+aggregate — equivariant_scorenetwork.py:18-24,35:
 * s[e,h] = q[dst]·(k[src]+ee[e]) / sqrt(Ch); alpha = softmax over in-edges of dst;
 * out[i] = sum_e dropout(alpha)[e,h] * (v[src]+ee[e]).  H*Ch == D <= 64.  alpha [E,H] is saved.
 * The dropout mask is a pure function of (seed + seed_dev[0]*prime, edge, head); seed_dev (may be
 * NULL) lets a captured hipGraph draw a fresh mask on every replay. */
int msde_edge_attention_fwd(const float* q, const float* k, const float* v,
                            const float* skip /* may be NULL, else out += skip (lin_skip(x_i)) */,
                            int ld /* row stride of q,k,v,skip: they may be column blocks of one
                                      fused projection [N, 4D] */,
                            const float* ee, int ld_ee /* row stride of ee (0 = D): it may be a column block
                                                          of one projection shared by several layers */,
                            const int* rowptr, const int* src, int N, int H, int Ch,
                            float p_drop, unsigned long long seed,
                            const unsigned long long* seed_dev, float* alpha, float* out,
                            void* stream);
/* backward of the above: writes g_q [N,D], g_ee [E,D] (row stride ld_ee, like ee) and the per-edge grads
 * g_kpe/g_vpe [E,D] (to be segment-summed by source into g_k / g_v with msde_segment_sum_rows). */
int msde_edge_attention_bwd(const float* g_out, const float* q, const float* k, const float* v,
                            int ld, float* g_skip /* may be NULL, else receives g_out */,
                            int ldg /* row stride of g_q and g_skip */,
                            const float* ee, int ld_ee, const float* alpha, const int* rowptr,
                            const int* src, int N, int H, int Ch, float p_drop,
                            unsigned long long seed, const unsigned long long* seed_dev,
                            float* g_q, float* g_ee, float* g_kpe, float* g_vpe,
                            int ld_kv /* row stride of g_kpe and g_vpe (0 = D): as the two halves of one
                                         [E, 2D] buffer they are segment-summed by ONE launch */,
                            void* stream);
/* basis mix + EquiLayer mean — equivariant_scorenetwork.py:159-164:
 * out[i] = mean_{e in in(i)} (c0*b_diff + c1*b_cross + c2*b_vert) */
int msde_frame_mix_mean_fwd(const float* coff, const float* basis, const int* rowptr, int N,
                            float* out, void* stream);
/* the same with the running sum over score layers folded in (equivariant_scorenetwork.py:166 `gradient += ...`):
 * out[i] = base[i] + mean(...); base == NULL: plain form */
int msde_frame_mix_mean_add_fwd(const float* coff, const float* basis, const int* rowptr, int N,
                                const float* base, float* out, void* stream);
int msde_frame_mix_mean_bwd(const float* g_out, const float* basis, const int* rowptr, int N,
                            int E_cap, float* g_coff, void* stream);

/* ------------------------------------------------------------------ dense layers ----------- */
/* torch.nn.functional.linear on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, exact fp32) — every
 * nn.Linear on the path (schnet.py:141-150,173-174; molecule_gnn_model.py:17; layers/common.py:21).
 * Y[M,N] = X[M,K] . W[N,K]^T + bias[N] (bias may be NULL). */
int msde_linear_fwd(const float* X, const float* W, const float* bias, int M, int N, int K, float* Y,
                    void* stream);
/* gX[M,K] = gY[M,N] . W[N,K] */
int msde_linear_bwd_x(const float* gY, const float* W, int M, int N, int K, float* gX,
                      void* stream);
/* gW[N,K] = gY^T . X and gb[N] = column sums of gY (gb may be NULL); split over M into slabs in
 * `workspace` (msde_linear_bwd_w_workspace_bytes) that are summed in a fixed order: reproducible. */
long long msde_linear_bwd_w_workspace_bytes(int M, int N, int K);
int msde_linear_bwd_w(const float* gY, const float* X, int M, int N, int K, float* gW, float* gb,
                      float* workspace, const int* rows_dev, void* stream);

/* ------------------------------------------------------------------ general fused GEMM ------ */
/* The dense products of the score networks with everything a library GEMM cannot fuse (csrc/gemm_ex.hip):
 *     C[M,N] (+)= alpha * rowscale[m] * epilogue( A[M,K1] . B + A2[M,K2] . B2 + bias[n] )
 * Replaces, per call site: nn.Linear on a concatenated input (torch.cat + F.linear:
 * SDE_model_3D_to_2D_node_adj_dense.py:156 embedding_3D + embedding_X; invariant_scorenetwork_dense.py:123-127
 * cat(x_list) -> final MLP), Linear + tanh/SiLU/ELU (layers/common.py:26-38), the input gradients of those through
 * the activation, the `flags` row mask (mask_x, SDE_model_3D_to_2D_node_adj_dense.py:543-548), and the per-channel
 * (block-diagonal) products of EdgeLayer (edge_network_dense.py:55-64) as `groups` problems in one launch.
 *   B layout: [N][K] with row stride ldb (nn.Linear weight), or with MSDE_GEMM_B_KMAJOR [K][N] (input gradients;
 *   NodeNetwork_dense weights, stored [in, out]: node_network_dense.py:32).
 *   epi = MSDE_EPI_ACT : v = acc + bias; Z (optional) receives v; columns [act_lo, act_hi) get act(v)
 *   epi = MSDE_EPI_DACT: v = (acc + bias) * act'(R[m,n]) on those columns; R = the forward's saved OUTPUT for
 *                        tanh / ELU / ReLU and its saved PRE-ACTIVATION for SiLU / shifted softplus
 *   groups > 1: problem g uses A + g*a_gs (also A2), B + g*b_gs (also B2), bias + g*bias_gs, C + g*c_gs (also Z),
 *   R + g*r_gs (all in floats); M, N, K, leading dimensions are shared. */
#define MSDE_ACT_NONE 0
#define MSDE_ACT_TANH 1
#define MSDE_ACT_SILU 2
#define MSDE_ACT_ELU 3
#define MSDE_ACT_SSP 4  /* shifted softplus, schnet.py:213-216 */
#define MSDE_ACT_RELU 5
#define MSDE_ACT_SSPO 6 /* msde_gemm_rs, MSDE_EPI_DACT only: derivative of the shifted softplus from its OUTPUT */
#define MSDE_EPI_ACT 0
#define MSDE_EPI_DACT 1
#define MSDE_GEMM_B_KMAJOR 1
#define MSDE_GEMM_ACCUMULATE 2
typedef struct msde_gemm_desc {
  const float* A;        /* [M, K1], row stride lda */
  const float* A2;       /* optional second K segment [M, K2], row stride lda2 (NULL: none) */
  const float* B;        /* weights of segment 1 */
  const float* B2;       /* weights of segment 2 */
  const float* bias;     /* [N] or NULL */
  const float* bias2;    /* optional second bias (two Linear layers summed into one product) */
  float* C;              /* [M, N], row stride ldc */
  float* Z;              /* optional pre-activation output, row stride ldz (MSDE_EPI_ACT only) */
  const float* R;        /* MSDE_EPI_DACT: saved forward tensor, row stride ldr */
  const float* rowscale; /* optional [M] multiplier (row mask) */
  long long a_gs, b_gs, bias_gs, c_gs, r_gs; /* per-group strides in floats */
  long long b_kblk_stride; /* with b_kblk_log2 > 0 ([N][K] layout only): k is cut into blocks of 2^b_kblk_log2 and block q
                              of row n starts at B + q * b_kblk_stride + n * ldb */
  int M, N, K1, K2;
  int lda, lda2, ldb, ldb2, ldc, ldz, ldr;
  int act, act_lo, act_hi, epi, flags, groups, b_kblk_log2;
  float alpha;           /* scales the result (1.0f for none) */
} msde_gemm_desc;
int msde_gemm_ex(const msde_gemm_desc* desc, void* stream);

/* Row-strip fp32 matrix-core GEMM (csrc/gemm_rs.hip): C[M,N] = epilogue( xf(A)[M,K] . B + bias ) for the plain nn.Linear
 * products of the encoders -- forward and input gradient of Geom3D/models/molecule_gnn_model.py:17 (GIN MLP),
 * Geom3D/models/schnet.py:141-148,163-167 (lin1 / lin2 / lin), SDE_model_2D_to_3D.py:264-271 (node_emb, edge_2D_emb,
 * input_mlp, coff_mlp) -- torch.addmm / torch.mm in the reference, with the BatchNorm1d / ReLU around them
 * (molecule_gnn_model.py:17,176-182; SDE_model_2D_to_3D.py:265) folded into the product:
 *   B layout     [K][N] with row stride ldb (MSDE_GEMM_B_KMAJOR must be set in flags): an input-gradient product takes
 *                nn.Linear's weight [out][in] as stored, a forward product its transposed copy (the [N][K] layout with
 *                per-lane pieces along k is bound by the address path on gfx950; MSDE_EUNSUP);
 *   axf          transform applied to A while it is loaded:
 *                MSDE_RS_AXF_AFFINE  a = A[m,k] * xf0[k] + xf1[k], then max(.,0) with MSDE_RS_AXF_RELU (BatchNorm apply);
 *                MSDE_RS_AXF_BNBWD   a = xf0[k] g + xf1[k] z + xf2[k] with g = A[m,k] (gated to 0 where z xf3[k] + xf4[k] <= 0
 *                                    when xf3 != NULL) and z = A2[m,k]: the BatchNorm input gradient, vectors from
 *                                    msde_bn_fin_bwd;
 *                A_out (optional) receives the transformed A (row stride lda_out): the weight gradient's operand;
 *   epilogue     v = acc + bias; Z (optional) receives v; epi = MSDE_EPI_ACT: v = act(v); MSDE_EPI_DACT: v = v * act'(R[m,n])
 *                (R as in msde_gemm_desc); then v += Res[m,n] (optional), v += C[m,n] with MSDE_GEMM_ACCUMULATE; C = v;
 *   stats        optional [strips][2][N] per-strip column statistics of the stored C over the VALID rows (*m_valid, or M):
 *                MSDE_RS_STATS_BNFWD  (mean, sum of squared deviations from it) -> msde_bn_fin_fwd;
 *                MSDE_RS_STATS_BNBWD  (sum v, sum v * (stats_z[m,n] - stats_mean[n]))   -> msde_bn_fin_bwd;
 *                strips and rows per strip from msde_gemm_rs_geometry(M, N, K).
 * K % 4 == 0, 16-byte aligned A / A2 / A_out / xf vectors with leading dimensions % 4 == 0, and for N > 64 also N % 4 == 0
 * with 16-byte aligned weight rows; otherwise MSDE_EUNSUP (callers use msde_gemm_ex).  rt: 0 = chosen by the library. */
#define MSDE_RS_AXF_NONE 0
#define MSDE_RS_AXF_AFFINE 1
#define MSDE_RS_AXF_BNBWD 2
#define MSDE_RS_AXF_RELU 4 /* flags bit */
#define MSDE_RS_VEC_STORE 8 /* flags bit, set by the library: C (and Res) rows take 8- / 16-byte stores */
#define MSDE_RS_STATS_BNFWD 1
#define MSDE_RS_STATS_BNBWD 2
typedef struct msde_rs_desc {
  const float* A;
  const float* A2;
  const float* B;
  const float* bias;
  float* C;
  float* Z;
  const float* R;
  const float* Res;
  float* A_out;
  const float *xf0, *xf1, *xf2, *xf3, *xf4;
  float* stats;
  const float* stats_z;
  const float* stats_mean;
  const int* m_valid;
  int M, N, K;
  int lda, lda2, ldb, ldc, ldz, ldr, ldres, lda_out, ld_sz;
  int act, epi, flags, axf, stats_mode;
  int rt, splits;
} msde_rs_desc;
int msde_gemm_rs(const msde_rs_desc* desc, void* stream);
int msde_gemm_rs_geometry(int M, int N, int K, int* strips, int* strip_rows);
/* 2-D tiled variant of the same product (csrc/gemm_t2.hip): 64-row x (N / splits)-column output tiles, BOTH operands staged
 * through LDS in K tiles of 32 by LDS-DMA, so a weight byte fetched by a CU serves 64 rows instead of 16.  Same descriptor,
 * epilogue and statistics semantics as msde_gemm_rs with two differences: B is read k-CONTIGUOUS, as [N][K] with row stride
 * ldb (MSDE_GEMM_B_KMAJOR must NOT be set: nn.Linear's weight as stored for a forward product, its transposed copy for an
 * input-gradient product), and the statistics strips are always 16 rows (msde_gemm_t2_geometry).  axf: MSDE_RS_AXF_NONE,
 * MSDE_RS_AXF_AFFINE and MSDE_RS_AXF_BNBWD as in msde_gemm_rs (applied to the A fragments in registers).  msde_gemm_t2_supported:
 * 1 when the library would run (M, N, K) with transform `axf` on this kernel (M >= 512, N >= 32, N % 4 == K % 4 == 0, and
 * a tiling that fills the chip in one round of workgroups), else 0 -- callers pick the weight copy to pass accordingly.  Operands below 2 GiB, 16-byte aligned rows; otherwise MSDE_EUNSUP. */
int msde_gemm_t2(const msde_rs_desc* desc, void* stream);
int msde_gemm_t2_supported(int M, int N, int K, int axf);
int msde_gemm_t2_geometry(int M, int N, int K, int* strips, int* strip_rows);
/* Finish the fused BatchNorm statistics (one small launch): forward -> scale = gamma rstd, shift = beta - mean scale (what
 * MSDE_RS_AXF_AFFINE of the consuming product applies), save_mean / save_rstd for the backward, running buffers updated
 * with `momentum` (unbiased variance), exactly as msde_bn_fwd.  Backward -> the three vectors of MSDE_RS_AXF_BNBWD and
 * dgamma / dbeta (may be NULL).  Partials are merged in a fixed order (bitwise reproducible). */
int msde_bn_fin_fwd(const float* stats, int strips, int strip_rows, int M, const int* m_valid, int C,
                    const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                    float* running_var, float* scale, float* shift, float* save_mean, float* save_rstd,
                    void* stream);
int msde_bn_fin_bwd(const float* stats, int strips, int M, const int* m_valid, int C, const float* gamma,
                    const float* mean, const float* rstd, float* p, float* w, float* u, float* dgamma,
                    float* dbeta, void* stream);

/* The BatchNorm input gradient as a pass of its own -- exactly what MSDE_RS_AXF_BNBWD applies to the A fragments:
 * out[m,c] = p[c] g' + w[c] Z[m,c] + u[c], g' = G[m,c] gated to 0 where Z[m,c] xf3[c] + xf4[c] <= 0 (xf3, xf4 both or neither);
 * rows behind *m_valid are written as 0.  The GIN layer's BatchNorm backward (molecule_gnn_model.py:17,176-182) in front of a
 * PLAIN product.  C % 4 == 0, leading dimensions % 4 == 0. */
int msde_bn_bwd_cols(const float* G, int ldg, const float* Z, int ldz, const float* p, const float* w, const float* u,
                     const float* xf3, const float* xf4, int M, const int* m_valid, int C, float* out, int ldo,
                     void* stream);
/* msde_bn_fin_bwd + msde_bn_bwd_cols as ONE launch (the finish no longer sits on the GIN backward chain as a launch of its own):
 * every workgroup sums the strip partials `stats` [strips][2][C] of its 64 columns itself (fixed order), forms p | w | u as
 * msde_bn_fin_bwd does and applies them to its rows; dgamma / dbeta (NULL: not wanted) are written once.  Same operand
 * conventions as the two entry points it replaces. */
int msde_bn_bwd_fin_cols(const float* stats, int strips, const float* gamma, const float* mean, const float* rstd,
                         const float* G, int ldg, const float* Z, int ldz, const float* xf3, const float* xf4, int M,
                         const int* m_valid, int C, float* out, int ldo, float* dgamma, float* dbeta, void* stream);
/* Y = max(X * scale[c] + shift[c], 0 if relu) over rows (the BatchNorm apply for a tensor with several consumers: the GIN
 * layer output, molecule_gnn_model.py:176-182); C % 4 == 0. */
int msde_affine_cols(const float* X, int M, int C, const float* scale, const float* shift, int relu, float* Y,
                     void* stream);
/* BatchNorm-backward partial sums for msde_bn_fin_bwd of a gradient G [M,C] that is not the result of msde_gemm_rs:
 * stats [ceil(M/64)][2][C] = per 64-row strip (sum g', sum g' (Z - mean)), g' = G gated by Y > 0 when Y != NULL (the
 * fused ReLU; Y = the BatchNorm + ReLU output); rows >= *m_valid contribute nothing.  C % 4 == 0. */
int msde_bn_bwd_colstats(const float* G, const float* Z, const float* Y, const float* mean, int M, const int* m_valid,
                         int C, float* stats, void* stream);
/* Re-laid-out copies of n fp32 blocks in one launch: the transposed weight copies the forward products of msde_gemm_rs read
 * and the stacked / permuted operands of fused layers (refreshed once per optimiser step).  table: n rows of 8 x int64
 * {src, dst, rows, cols, src_ld, dst_ld, mode, plane_stride} (device); mode 0: dst[c * dst_ld + r] = src[r * src_ld + c]
 * (transpose of the rows x cols block), mode 1: dst[r * dst_ld + c] = src[r * src_ld + c] (copy); word 7 unused;
 * prefix [n+1]: first 32 x 32 tile of each block, prefix[n] = total_tiles. */
int msde_transpose_multi(const long long* table, const int* prefix, int n, int total_tiles, void* stream);
/* the same for one block, no tables */
int msde_relayout(const float* src, int src_ld, float* dst, int dst_ld, int rows, int cols, int mode, void* stream);
/* the same for one matrix, no tables: dst [cols][rows] = src [rows][cols]^T */
int msde_transpose(const float* src, float* dst, int rows, int cols, void* stream);

/* Narrow output layer of an MLP over rows (basis_mlp, equivariant_scorenetwork.py:142-146: Linear -> SiLU -> Linear(H, 3)):
 * out[e][j] = b[j] + sum_c silu(Z[e][c]) W[j][c] on the PRE-activation Z [E, H] (row stride ldz), J <= 4, H % 4 == 0,
 * H <= 256, 16-byte aligned Z / W / gZ (else MSDE_EUNSUP).  Backward: gZ [E, H] = d/dZ (SiLU' included; rows past the
 * row bound of E are zero) and the layer's own gradient [gW (J x H) | gb (J)] = gWb, summed in fixed order from
 * msde_mlp_head_bwd_slabs(E, H) workgroup slabs in `workspace` (gWb == NULL: the slabs stay there for
 * msde_reduce_slabs_multi).  Slabs and gWb hold J*H + J floats rounded up to a multiple of 4 (zero padding). */
int msde_mlp_head_fwd(const float* Z, int ldz, const float* W, const float* b, int E, int H, int J, float* out,
                      void* stream);
int msde_mlp_head_bwd_slabs(int E, int H);
int msde_mlp_head_bwd(const float* Z, int ldz, const float* W, const float* g, int E, int H, int J, float* gZ, float* gWb,
                      float* workspace, const int* rows_dev, void* stream);
/* The head fused with the frame mix + mean that consumes it in the 2D->3D score network (equivariant_scorenetwork.py:142-166):
 * out[i] = base[i] + mean over in-edges e of i of sum_j coff[e][j] basis[e][j][:], coff[e] = b + W silu(Z[e]) (3 outputs);
 * rowptr = by-target CSR of the edges (in-edges of a node contiguous), mix [E, 3] = scratch.  Backward: as msde_mlp_head_bwd
 * with the head's gradient formed from the NODE gradient gnode [N, 3] (dst [E] = target node of each edge); the slabs of the
 * head's own weight / bias gradient stay in `workspace` (msde_mlp_head_bwd_slabs(E, H) of them) for a batched reduction. */
int msde_mlp_head_mix_fwd(const float* Z, int ldz, const float* W, const float* b, int H, const float* basis,
                          const int* rowptr, int N, const float* base, float* mix, float* out, void* stream);
int msde_mlp_head_mix_bwd(const float* Z, int ldz, const float* W, const float* gnode, const float* basis, const int* dst,
                          const int* rowptr, int E, int H, float* gZ, float* workspace, const int* rows_dev, void* stream);

/* ------------------------------------------------------------------ CFConv on unordered atom pairs -- */
/* SchNet's interaction graph (schnet.py:91-93: radius_graph over the molecule, 32-neighbour cap) is symmetric whenever the
 * cap cannot bind (<= 33 atoms per molecule) and the continuous filter depends on the distance only (schnet.py:141-145,
 * 185-195): the filter network runs once per UNORDERED pair (csrc/cfconv_pair.hip).  Pairs of molecule m, local atoms
 * a < b of its n atoms: row pair_ptr[m] + a n - a (a + 1) / 2 + (b - a - 1); pair_ptr[B] = number of pairs.
 * msde_pair_build: pair_ptr [B+1], pi / pj [P_cap] (batch-global atom indices, pi < pj), pd [P_cap] = |p_i - p_j|, or -1
 * when d^2 >= r2 (no edge: its filter row is zero).  P_cap >= sum n (n - 1) / 2; a batch with more pairs is truncated
 * (pair_ptr[B] = P_cap; the aggregation treats the missing pairs as absent) and *err (may be NULL) is set to 1. */
int msde_pair_build(const float* pos, const int* mol_ptr, int B, float r2, int* pair_ptr, int* pi, int* pj, float* pd,
                    int P_cap, int* err, void* stream);
/* Wf [P_cap, F] = (W2 ssp(W1 rbf(pd) + b1) + b2) * C(pd) for the first *count pairs (count = pair_ptr + B); F = 128,
 * G <= 64; W1 [F][G], W2 [F][F] as nn.Linear stores them.  blocks_per_wg <= 0: chosen by the library. */
int msde_cfconv_pair_filter(const float* pd, const int* count, const float* W1, const float* b1, const float* W2,
                            const float* b2, const float* offset, int F, int G, int P_cap, float coeff, float cutoff,
                            int blocks_per_wg, float* Wf, void* stream);
/* The same for the filter networks of ALL L <= MSDE_CFCONV_MAX_LAYERS interaction blocks in ONE launch: the inputs of
 * `W = self.mlp(edge_attr) * C` (schnet.py:141-145) are the same smeared distances for every block (schnet.py:96-104), so no
 * block's filter rows depend on the layer chain.  W1 / b1 / W2 / b2 / Wf: HOST arrays of L device pointers (copied into the
 * launch's arguments).  Results equal msde_cfconv_pair_filter's bit for bit. */
#define MSDE_CFCONV_MAX_LAYERS 8
int msde_cfconv_pair_filter_multi(const float* pd, const int* count, const float* const* W1, const float* const* b1,
                                  const float* const* W2, const float* const* b2, const float* offset, int L, int F, int G,
                                  int P_cap, float coeff, float cutoff, int blocks_per_wg, float* const* Wf, void* stream);
/* out[i] = sum_{j != i in i's molecule} x[j] * Wf[pair(i, j)], ascending j (fixed order, no atomics): the CFConv message
 * aggregation with x = x1, and -- the pair set and Wf being symmetric -- its input gradient with x = g_agg. */
int msde_cfconv_pair_aggregate(const float* x, const float* Wf, const int* batch, const int* mol_ptr, const int* pair_ptr,
                               int N, int B, int F, float* out, void* stream);
/* Filter-network weight gradients over pairs: as msde_cfconv_fused_bwd_w with the rows (pi, pj, pd) and the two directions
 * of a pair summed before the weight-gradient products; slab / workspace sizes from msde_cfconv_fused_bwd_w_slabs /
 * _workspace_floats(P_cap, ...). */
int msde_cfconv_pair_bwd_w(const float* g_agg, const float* x1, const float* pd, const int* count, const int* pi,
                           const int* pj, const float* W1, const float* b1, const float* W2, const float* offset, int N,
                           int F, int G, int P_cap, float coeff, float cutoff, int max_workgroups, float* gW1, float* gb1,
                           float* gW2, float* gb2, float* workspace, void* stream);
/* The same for L <= MSDE_CFCONV_MAX_LAYERS interaction blocks in ONE launch (slabs only): the filter-network gradients of a
 * block are parameter gradients, nothing in the backward chain of schnet.py:185-195 reads them, so the caller collects the
 * blocks' (g_agg, x1) pairs while the chain runs and launches them together -- L x as many 64-pair chunks over the same
 * workgroups, i.e. an even split (one block alone: 2.2 chunks per workgroup, rounded up to 3).  g_agg / x1 / W1 / b1 / W2 /
 * slabs: HOST arrays of L device pointers; slabs[l]: msde_cfconv_pair_bwd_w_multi_slabs(P_cap, L, max_workgroups) slabs of
 * F*F + F*G + 2F floats for block l (the whole launch runs L x that many workgroups, <= max_workgroups; 0: one per CU).
 * Slab contents equal those of L msde_cfconv_pair_bwd_w calls with the same per-block workgroup count. */
int msde_cfconv_pair_bwd_w_multi_slabs(int P_cap, int L, int max_workgroups);
int msde_cfconv_pair_bwd_w_multi(const float* const* g_agg, const float* const* x1, const float* pd, const int* count,
                                 const int* pi, const int* pj, const float* const* W1, const float* const* b1,
                                 const float* const* W2, const float* offset, int L, int N, int F, int G, int P_cap,
                                 float coeff, float cutoff, int max_workgroups, float* const* slabs, void* stream);

/* ------------------------------------------------------------------ 3D->2D dense score head -- */
/* SDEModel3Dto2D_node_adj_dense.forward (SDE_model_3D_to_2D_node_adj_dense.py:101-179) with its
 * EdgeScoreNetwork_dense / NodeScoreNetwork_dense (invariant_scorenetwork_dense.py:74-93,118-131) on RAGGED data
 * (csrc/dense_head.hip): atoms in batch order, atom pairs of molecule b at rows pair_ptr[b] + i*n_b + j of every
 * [P, *] array, P = sum_b n_b^2, n_b <= 32.  The pair channel buffer AC is [P, MSDE_DENSE_AC_LD]: columns 0-1 the
 * perturbed adjacency and its square (pow_tensor, :28-37), then the outputs of the 4 EdgeNetwork_dense layers
 * (8, 8, 8, 4 channels) -- exactly the concatenation the final pair MLP reads (:81-84).  Atom-class arrays
 * ([N, MSDE_DENSE_XP_LD], 119 classes) are padded to a 16-byte row. */
#define MSDE_DENSE_AC_LD 32
#define MSDE_DENSE_XP_LD 120
/* to_dense_adj + node_flags + gen_noise + perturbation of adjacency and one-hot classes (:112-152, 523-548).
 * Time: t_in[b] if given, else the antithetic integer draws (:112-114; draws == NULL: drawn on the device).  sde_vp = 0: VESDE(p0 = sigma_min, p1 =
 * sigma_max); 1: VPSDE(p0 = beta_0, p1 = beta_1) (SDE_dense.py).  Noise: noise_adj [B, Nm_pad, Nm_pad] and noise_x
 * [B, Nm_pad, ncls] (the reference's padded randn draws, replay mode), or both NULL: counter-based N(0,1) from
 * (seed + *seed_dev).  Outputs: AC columns 0-1 (and zeros in the two pad columns), z_adj [P], flags [N],
 * mean_std [B][2], px / z_x [N, MSDE_DENSE_XP_LD]. */
int msde_dense_prepare(const int* rowptr, const int* src, const float* bond_val, const int* z_atom,
                       const int* mol_ptr, const int* pair_ptr, const long long* draws, const float* t_in,
                       int B, int T, float eps, int sde_vp, float p0, float p1, const float* noise_adj,
                       const float* noise_x, int Nm_pad, unsigned long long seed,
                       const unsigned long long* seed_dev, int ncls, int n_max, float* AC, float* z_adj,
                       float* flags, float* mean_std, float* px, float* z_x, void* stream);
/* One EdgeNetwork_dense layer minus its node-level GEMMs (edge_network_dense.py:105-128): inputs QK [N, 64 C]
 * (func_q outputs of the C channels, then func_k's), XV [N, 16 C] (x W_c of the per-channel GCNs), the C input channels
 * AC[:, in_off ..]; outputs x_out [N,16], AC[:, out_off .. out_off+CO) and the tensors the backward reuses (IN [P,2C],
 * H1, H2 [P,16], xcat [N,16C], Hmc [N,16]).  (C, CO) in {(2,8), (8,8), (8,4)}. */
typedef struct msde_edge_layer_params {
  const float* bv;                                 /* [C][16]  func_v biases */
  const float *mW0, *mb0, *mW1, *mb1, *mW2, *mb2;  /* pair MLP: [16][2C], [16][16], [CO][16] */
  const float *cW0, *cb0, *cW1, *cb1;              /* channel MLP (multi_channel): [16][16C], [16][16] */
} msde_edge_layer_params;
int msde_dense_edge_layer_fwd(const float* QK, const float* XV, float* AC, int in_off, int out_off, int C,
                              int CO, const float* flags, const int* mol_ptr, const int* pair_ptr,
                              const msde_edge_layer_params* params, int B, int n_max, float* x_out,
                              float* IN, float* H1, float* H2, float* xcat, float* Hmc, void* stream);
/* Its backward: reads gAC[:, out block] (and g_xout, NULL when x_out is unused), accumulates into gAC[:, in block]
 * when need_gadj, writes gQK [N,64C], gXV [N,16C] and the operand pairs of the weight-gradient GEMMs (GO, GH2, GH1
 * [P, .]; GY, GHm [N,16]; GV [N,16C]). */
int msde_dense_edge_layer_bwd(const float* QK, const float* XV, const float* AC, float* gAC, int in_off,
                              int out_off, int C, int CO, const float* flags, const int* mol_ptr,
                              const int* pair_ptr, const msde_edge_layer_params* params, int B, int n_max,
                              const float* x_out, const float* g_xout, const float* IN, const float* H1,
                              const float* H2, const float* xcat, const float* Hmc, int need_gadj, float* gQK,
                              float* gXV, float* GO, float* GH2, float* GH1, float* GY, float* GHm, float* GV,
                              void* stream);
/* The 4 dense-GCN + tanh layers of NodeScoreNetwork_dense (invariant_scorenetwork_dense.py:118-122) after the first
 * product: XW0 = x W_0 [N,16]; Wl = W_1..3 [3][16 in][16 out]; bl [4][16]; XS [N, ldxs] receives [x_1|x_2|x_3|x_4]. */
int msde_dense_node_gcn_fwd(const float* XW0, const float* AC, const int* mol_ptr, const int* pair_ptr,
                            const float* Wl, const float* bl, int B, int n_max, float* XS, int ldxs,
                            void* stream);
/* backward: GP [N,64] = gradients of the 4 pre-activations (bias gradients = column sums), MM [N,64] = An^T GP per
 * layer (block 0 = gradient of XW0; x_l^T MM_l = weight gradients of layers 1..3). */
int msde_dense_node_gcn_bwd(const float* gXS, int ldg, const float* XS, int ldxs, const float* AC,
                            const int* mol_ptr, const int* pair_ptr, const float* Wl, int B, int n_max,
                            float* GP, float* MM, void* stream);
/* Last Linear(F2, 1) of the pair MLP + diagonal / flag masks + score = -net / std + both losses (:86-94,157-179).
 * out[0] = loss_x, out[1] = loss_adj; scale_* = 1/(B Nmax ncls), 1/(B Nmax^2) (reduce_mean) or 0.5/B;
 * residuals res_adj [P], res_x [N, MSDE_DENSE_XP_LD] are kept for the backward; part = [B * MSDE_DENSE_LOSS_SPLITS][2]
 * workspace (a molecule's pairs / class rows are shared by MSDE_DENSE_LOSS_SPLITS workgroups).  F2 <= 64. */
#define MSDE_DENSE_LOSS_SPLITS 4
int msde_dense_loss_fwd(const float* G2, int F2, const float* w2, const float* b2, const float* OUT,
                        const float* z_adj, const float* z_x, const float* flags, const float* mean_std,
                        const int* mol_ptr, const int* pair_ptr, int B, int ncls, float anneal_power,
                        float scale_x, float scale_adj, const int* nmax_dev /* non-NULL: reduce_mean scales from this
                        device-side N_max instead of scale_x / scale_adj */, float* res_adj, float* res_x, float* part,
                        float* out, void* stream);
/* g_lx / g_la: device scalars dL/dloss_x, dL/dloss_adj (NULL = 0).  gS [P] = gradient of the pair MLP's scalar output, gZ2 [P,F2] =
 * gradient of the pre-activation of its last hidden layer (SiLU), gOUT [N, MSDE_DENSE_XP_LD] = gradient of the node
 * MLP's output. */
int msde_dense_loss_bwd(const float* g_lx, const float* g_la, const float* res_adj, const float* res_x,
                        const float* Z2, int F2,
                        const float* w2, const float* flags, const float* mean_std, const int* mol_ptr,
                        const int* pair_ptr, int B, int ncls, float anneal_power, float scale_x,
                        float scale_adj, const int* nmax_dev, float* gS, float* gZ2, float* gOUT, void* stream);

/* ------------------------------------------------------------------ contrastive loss ------- */
/* do_CL('EBM_node_dot_prod') in both directions + dual_CL — examples/util.py:52-68,76-79.
 * perm1/perm2: the two negative-sample permutations (torch.randperm).  out[0] = loss, out[1] = accuracy.
 * rows [N,3] (p, n1, n2 logits) and inv1/inv2 [N] (inverse permutations) feed the backward. */
int msde_cl_ebm_fwd(const float* X, const float* Y, const int* perm1, const int* perm2, int N, int D,
                    float invT, float* rows, int* inv1, int* inv2, float* out, const int* rows_dev, void* stream);
int msde_cl_ebm_bwd(const float* X, const float* Y, const int* perm1, const int* perm2,
                    const int* inv1, const int* inv2, const float* rows, const float* g_loss, int N,
                    int D, float invT, float* gX, float* gY, const int* rows_dev, void* stream);

/* ------------------------------------------------------------------ normalisation ---------- */
/* nn.BatchNorm1d in training mode over the rows of X[M,C], optional fused ReLU —
 * molecule_gnn_model.py:17,176-182; SDE_model_2D_to_3D.py:265.  Batch statistics by Welford partials
 * combined in a fixed order; running_mean/var (may be NULL) are updated with `momentum` (unbiased
 * variance), save_mean/save_rstd [C] feed the backward.  workspace: msde_bn_workspace_floats floats. */
int msde_bn_workspace_floats(int M, int C);
int msde_bn_fwd(const float* X, int M, int C, const float* gamma, const float* beta, float eps,
                float momentum, float* running_mean, float* running_var, int relu, float* Y,
                float* save_mean, float* save_rstd, float* workspace, const int* rows_dev, void* stream);
int msde_bn_bwd(const float* dY, const float* X, const float* save_mean, const float* save_rstd,
                const float* gamma, const float* beta, int relu, int M, int C, float* dX,
                float* dgamma, float* dbeta, float* workspace, const int* rows_dev, void* stream);

/* y = res + LayerNorm(x) over rows of D floats (res may be NULL) — GATLayer, equivariant_scorenetwork.py:36,38.
 * mean/rstd [N] are saved for the backward.  D % 4 == 0, D <= 1024. */
int msde_res_layernorm_fwd(const float* x, const float* res, const float* gamma, const float* beta,
                           int N, int D, float eps, float* y, float* mean, float* rstd,
                           void* stream);
/* gx (the residual branch's gradient is g itself), and [ggamma | gbeta] which must be one contiguous
 * buffer of 2*D floats (gbeta == ggamma + D).  workspace: 64 * 2 * D floats. */
int msde_res_layernorm_bwd(const float* g, const float* x, const float* gamma, const float* mean,
                           const float* rstd, int N, int D, float* gx, float* ggamma, float* gbeta,
                           float* workspace, void* stream);

/* out[c] = sum_m X[m,c] (bias gradient of an nn.Linear when the vendor GEMM computes the weight
 * gradient); workspace: msde_bn_workspace_floats(M, C) floats; fixed summation order. */
int msde_colsum(const float* X, int M, int C, float* out, float* workspace, const int* rows_dev, void* stream);

/* Batched weight gradients: msde_linear_bwd_w_partial runs only the split-M GEMM of msde_linear_bwd_w and
 * leaves slabs [splits][N*K] (+ bias partials [splits][N] when want_bias) in `slabs`
 * (msde_linear_bwd_w_workspace_bytes); splits = msde_linear_bwd_w_splits(M,N,K).  msde_reduce_slabs_multi then
 * sums the slabs of MANY layers in one launch: rows[r] = MSDE_REDUCE_ROW (8) int64 {slab address, splits, entries n,
 * output address, row_len, slab_ld, out_ld, split_stride}: entry (i / row_len, i % row_len) of split z is read at
 * slabs[z * split_stride + r * slab_ld + c] and the sum written to out[r * out_ld + c] -- a flat row has row_len =
 * slab_ld = out_ld = split_stride = n; a 2-D row sums a column block of the slabs into a column block of a wider
 * gradient (a layer consumed as part of a stacked / permuted operand).  prefix[r] = chunks before row r
 * (msde_reduce_slabs_chunks), prefix[count] = total_chunks.  Fixed summation order, like msde_linear_bwd_w. */
#define MSDE_REDUCE_ROW 8
int msde_linear_bwd_w_splits(int M, int N, int K);
/* Grouped form: the split-M GEMMs of many layers in ONE launch (same tile code and splits; slabs bit-identical to the
 * per-layer kernel except for layers with N <= 32 AND K <= 32, see below).
 * msde_linear_bwd_w_describe fills one HOST row of the problem table for a layer and returns the
 * number of workgroups it needs; prefix[p] = workgroups before problem p, prefix[count] = total_blocks.
 * The row also names the layer's TILE SHAPE (outputs are N x K): 64 x 64 (four waves, a 32 x 32 quadrant each); N <= 32 < 64 < K:
 * 32 x 128, K <= 32 < 64 < N: 128 x 32 (the four waves side by side along the wide dimension; same summation order per output);
 * N <= 32 and K <= 32: 32 x 32 with the four waves on the same outputs, each summing every fourth 32-row block of the rows,
 * their partial sums added in wave order (fixed, but not the per-layer kernel's order). */
int msde_linear_bwd_w_describe(const float* gY, const float* X, int M, int N, int K, int want_bias,
                               float* slabs, const int* rows_dev, long long* host_row);
/* the same for operands that are column blocks of wider buffers: ldg / ldx = row strides of gY / X (floats).
 * Every table row is MSDE_WGRAD_ROW int64 wide. */
#define MSDE_WGRAD_ROW 16
int msde_linear_bwd_w_describe_ld(const float* gY, int ldg, const float* X, int ldx, int M, int N, int K,
                                  int want_bias, float* slabs, const int* rows_dev, long long* row);
int msde_linear_bwd_w_grouped(const long long* probs, const int* prefix, int count, int total_blocks,
                              void* stream);
/* the same launch limited to `max_workgroups` resident workgroups (0 = one per tile): each walks several tiles, so the
 * GEMMs can run beside a latency-critical kernel chain on another stream without occupying every CU. */
int msde_linear_bwd_w_grouped_ex(const long long* probs, const int* prefix, int count, int total_blocks,
                                 int max_workgroups, void* stream);
int msde_linear_bwd_w_partial(const float* gY, const float* X, int M, int N, int K, int want_bias,
                              float* slabs, const int* rows_dev, void* stream);
int msde_reduce_slabs_multi(const long long* rows, const int* prefix, int count, int total_chunks,
                            void* stream);
/* chunks (= workgroups) a row of n entries x `splits` slabs takes in msde_reduce_slabs_multi's prefix table: 256
 * entries per chunk, 64 for rows of >= MSDE_REDUCE_LONG splits (16 split lanes instead of 4) */
#define MSDE_REDUCE_LONG 64
long long msde_reduce_slabs_chunks(long long n, int splits);

/* ------------------------------------------------------------------ pointwise stages ------- */
/* ShiftedSoftplus (schnet.py:199-206): y = softplus(x) - log 2 (threshold 20); g_x = g * sigmoid(x).
 * 16-byte aligned buffers. */
int msde_ssp_fwd(const float* x, long long n, float* y, void* stream);
int msde_ssp_bwd(const float* g, const float* x, long long n, float* gx, void* stream);
/* nn.SiLU -> nn.Dropout(p) of the GAT feed-forward block (equivariant_scorenetwork.py:27-31), p = 0 for the
 * plain F.silu between layers (:142).  keep = uniform(seed [+ seed_dev[0]*FNV], element index) >= p, kept
 * values scaled by 1/(1-p); the backward regenerates the mask. */
int msde_silu_dropout_fwd(const float* x, long long n, float p, unsigned long long seed,
                          const unsigned long long* seed_dev, float* y, void* stream);
int msde_silu_dropout_bwd(const float* g, const float* x, long long n, float p, unsigned long long seed,
                          const unsigned long long* seed_dev, float* gx, void* stream);
/* out = a*b + c (SDE_model_2D_to_3D.py:393); ga = g*b, gb = g*a (either may be NULL); gc = g. */
int msde_mul_add_fwd(const float* a, const float* b, const float* c, long long n, float* out, void* stream);
int msde_mul_add_bwd(const float* g, const float* a, const float* b, long long n, float* ga, float* gb,
                     void* stream);
/* Predictor-corrector sampler arithmetic for ONE diffusion time shared by all n atoms (pretrain_MoleculeSDE_inference_2D_to_3D_VE_VP.py
 * :191-212 LangevinCorrector.update_fn, :163-168 ReverseDiffusionPredictor.update_fn): `out` is the score network's raw output
 * [n, 3] (score = -out / std); par [S][4] = {std(t), G(t), alpha(t), fa(t)} per time step on the device, fa = 1 + f(x)/x of the
 * discretised forward SDE (VE: 1).  corrector: step = (snr mean|noise| / mean|score|)^2 2 alpha, x_mean = pos + step score,
 * x = x_mean + sqrt(2 step) scale_eps noise (one workgroup, fixed-order reductions); predictor: x_mean = pos - ((fa - 1) pos -
 * G^2 score), x = x_mean + G noise.
 * step (device int64, may be NULL = row 0): the iteration counter -- the corrector uses row step[0] and then advances it, the
 * predictor of the same iteration uses row step[0] - 1: a captured iteration replays without any host work.
 * noise [n, 3], or NULL: N(0,1) draws from the counter generator keyed by (seed, row, corrector / predictor, element). */
int msde_pc_corrector(const float* out, const float* pos, const float* noise, const float* par, long long* step,
                      unsigned long long seed, int n, float snr, float scale_eps, float* x, float* x_mean, void* stream);
int msde_pc_predictor(const float* out, const float* pos, const float* noise, const float* par, const long long* step,
                      unsigned long long seed, int n, float* x, float* x_mean, void* stream);
/* MD17 force fine-tuning losses (finetune_MD17.py:68-74): loss[0] = ce mean_b |E_b - y_b| + cf mean |fsign dE - f| over the n3 = 3 N
 * force components (fsign = -1: forces are minus the coordinate gradient dE), and the seeds of the backward pass in the same
 * launch: gE [B] = d loss / d E, gdE [n3] = d loss / d (dE) (|x|' = sign x, 0 at 0, as torch).  One workgroup, fixed order. */
int msde_l1_energy_force_loss(const float* E, const float* y, int B, const float* dE, const float* f, int n3, float fsign, float ce,
                              float cf, float* loss, float* gE, float* gdE, void* stream);
/* torch.randperm(n) for the contrastive negatives (examples/util.py:55), n <= 4096 (else MSDE_EUNSUP):
 * out[count][n] int32, `count` independent uniform shuffles in one launch (dual_CL draws two) from the
 * counter-based generator (seed [+ seed_dev[0]*FNV], permutation number, index). */
int msde_randperm(int n, int count, unsigned long long seed, const unsigned long long* seed_dev, int* out,
                  const int* rows_dev, void* stream);
/* VE perturbation (SDE_model_2D_to_3D.py:401-412, SDE_sparse.py VESDE.marginal_prob): draws [B/2+1] int64 in
 * [0,T); molecule b uses ts = draws[b] (b < B/2+1) or T - draws[b-(B/2+1)] - 1; t = ts/T*(1-eps)+eps;
 * std_out[i] = sigma_min (sigma_max/sigma_min)^t of atom i's molecule; pos_out = pos + std * noise. */
int msde_ve_perturb(const float* pos, const float* noise, const long long* draws, const int* batch, int N,
                    int B, int T, float eps, float sigma_min, float sigma_max, float* pos_out,
                    float* std_out, void* stream);
/* the same with the draws made in the kernel: noise_out [N,3] ~ N(0,1) and one uniform integer time step per antithetic
 * pair of molecules from the counter-based generator (seed [+ seed_dev[0]*FNV], index); replaces torch.randn_like +
 * torch.randint (SDE_model_2D_to_3D.py:401-404) + msde_ve_perturb. */
int msde_ve_perturb_rng(const float* pos, const int* batch, int N, int B, int T, float eps, float sigma_min,
                        float sigma_max, unsigned long long seed, const unsigned long long* seed_dev,
                        float* noise_out, float* pos_out, float* std_out, void* stream);
/* VE position loss (SDE_model_2D_to_3D.py:425-432): loss[0] = mean_b mean_{i in b} sum_k (scores-noise)^2
 * [* std_i^anneal_power when anneal_power != 0]; mol_ws: B floats.  bwd: g_scores [N,3] from g_loss[0]. */
int msde_ve_pos_loss_fwd(const float* scores, const float* noise, const float* std, float anneal_power,
                         const int* mol_ptr, int N, int B, float* mol_ws, float* loss, void* stream);
int msde_ve_pos_loss_bwd(const float* scores, const float* noise, const float* std, float anneal_power,
                         const int* mol_ptr, const int* batch, int N, int B, const float* g_loss,
                         float* g_scores, void* stream);

/* ------------------------------------------------------------------ GAT layer tail --------- */
/* GATLayer after the attention (equivariant_scorenetwork.py:27-38,142), D = 32:
 *   y1 = res + LayerNorm1(x); a = Dropout_p(SiLU(W0 y1 + b0)); x2 = W3 a + b3; out = y1 + LayerNorm2(x2)
 *   [out = SiLU(out) when silu_out].  Saves y1, h0 = W0 y1 + b0 and x2 ([N,D] each) for the backward.
 * bwd: g_x (to the attention output), g_res (to the layer input), the two (gradient, input) pairs of the
 * feed-forward weight gradients -- (g_x2, a) for W3/b3 and (g_h0, y1) for W0/b0 -- and per-workgroup partial sums
 * ln_part [msde_gat_tail_blocks(N)][4D] = [d ln2_g | d ln2_b | d ln1_g | d ln1_b] to be summed over workgroups. */
int msde_gat_tail_blocks(int N);
int msde_gat_tail_fwd(const float* x, const float* res, const float* ln1_g, const float* ln1_b,
                      const float* W0, const float* b0, const float* W3, const float* b3,
                      const float* ln2_g, const float* ln2_b, int N, int D, float eps1, float eps2,
                      float p_drop, unsigned long long seed, const unsigned long long* seed_dev,
                      int silu_out, float* out, float* y1, float* h0, float* x2, void* stream);
int msde_gat_tail_bwd(const float* g_out, const float* x, const float* y1, const float* h0,
                      const float* x2, const float* ln1_g, const float* W0, const float* W3,
                      const float* ln2_g, const float* ln2_b, int N, int D, float eps1, float eps2,
                      float p_drop, unsigned long long seed, const unsigned long long* seed_dev,
                      int silu_out, float* g_x, float* g_res, float* g_x2, float* a, float* g_h0,
                      float* ln_part, const int* rows_dev, void* stream);

/* ------------------------------------------------------------------ score network, one workgroup per molecule --- */
/* EquivariantScoreNetwork.forward (equivariant_scorenetwork.py:121-169; called from SDE_model_2D_to_3D.py:386-391 and
 * get_score :393-445) for hidden = 32, 8 heads, basis-MLP width 128 and molecules of <= 32 atoms: 4 GATLayers (:13-40),
 * 2 basis MLPs, frame mix and the mean over in-edges, ONE launch, one workgroup per molecule (csrc/escore_mol.hip).
 *   params: DEVICE array of 76 device pointers (the caller keeps it alive and current), nn.Linear layouts: per GAT layer
 *     (x4) [lin_query, lin_key, lin_value, lin_skip weights ([32,32] each), their 4 biases, lin_edge weight ([32,32]), ln1_g,
 *     ln1_b, W0, b0, W3, b3, ln2_g, ln2_b], then per basis MLP (x2) [W1 ([128,64]), b1, W2 ([3,128]), b2]
 *   x0 [N,32] node features, edge_attr [E,ld_ea] (by-target edge order), basis [E,9], mol_ptr [B+1] atom ranges,
 *   rowptr [N+1] / src [E] / dst [E] the by-target CSR of the (extended) edges.
 *   Dropout: attention weights (p_att) and feed-forward (p_ffn) masks are the counter masks of msde_edge_attention_fwd /
 *   msde_gat_tail_fwd with per-layer seeds seed0 + 4*block + conv (feed-forward: ^ 0x46464E), + seed_dev[0] * 0x100000001B3.
 *   out [N,3]: the score ("gradient"); rows behind mol_ptr[B] are zero-filled.
 *   n_max: the largest molecule of the batch (atoms); > 32 -> MSDE_EUNSUP (a molecule's rows live in LDS; nothing is cut silently).
 *   saved (or NULL): what msde_escore_mol_bwd needs, msde_escore_mol_saved_floats(N) floats (per layer and atom: attention
 *   output, y1, h0, x2, layer output, softmax max and 1/sum per head; everything per-edge is recomputed in the backward). */
long long msde_escore_mol_saved_floats(int N);
int msde_escore_mol_fwd(const void* const* params, const float* x0, const float* edge_attr, int ld_ea,
                        const float* basis, const int* mol_ptr, int B, const int* rowptr, const int* src,
                        const int* dst, int N, int E, int hidden, int heads, int hidden_coff, int n_max, float p_att,
                        float p_ffn, unsigned long long seed0, const unsigned long long* seed_dev, float eps1,
                        float eps2, float* out, float* saved, void* stream);

/* get_score of SDEModel2Dto3D_01/_02 up to the division by -std (SDE_model_2D_to_3D.py:393-445; _01: :200-249): the
 * coordinate-dependent edge features (frame, Gaussian-Fourier features, input_mlp, coff_mlp, project,
 * edge_attr = input_mlp(.) * edge_2D + project(.)) are built by the library from `pos` [N,3] and the coordinate-INDEPENDENT
 * rows edge_2D [E, ld_e2d] (edge_2D_emb of the node pairs, computed once per representation), then the score network runs as in
 * msde_escore_mol_fwd in inference mode.  params: DEVICE array of 86 pointers = the 76 above, then dist_gaussian_fourier.W [32]
 * (any valid pointer when has_dist = 0), coff_gaussian_fourier.W [32], input_mlp weight [32,64] / bias, coff_mlp weight [32,128]
 * / bias, project[0] weight [32,66] / bias, project[1] weight [32,32] / bias.  has_dist = 0: the _01 model (no distance
 * branch).  n_max: the largest molecule of the batch.
 * scratch: msde_escore_mol_score_scratch_floats(E) floats, 16-byte aligned.  TWO launches: everything that depends on an edge
 * alone (edge features, lin_edge of the four layers, the edge half of both basis MLPs' first Linear) in a wide launch, one wave
 * per 16 edges, into `scratch`; then one workgroup per molecule.  Molecules of <= 32 atoms (MSDE_EUNSUP above). */
long long msde_escore_mol_score_scratch_floats(int E);
int msde_escore_mol_score(const void* const* params, const float* x0, const float* pos, const float* edge_2D, int ld_e2d,
                          int has_dist, const int* mol_ptr, int B, const int* rowptr, const int* src, const int* dst, int N, int E,
                          int hidden, int heads, int hidden_coff, int n_max, float eps1, float eps2, float* scratch, float* out,
                          void* stream);

/* Backward of msde_escore_mol_fwd (same arguments; `saved` written by it).  rowptr_s [N+1] / perm_s [E]: the by-source view
 * of the edges (slot -> by-target edge id).  g_out [N,3] -> g_x0 [N,32], g_edge_attr [E,ld_gea] (all rows written; rows behind
 * the last molecule are zero) and B slabs of msde_escore_mol_slab_floats() floats: the weight gradients of every molecule in
 * the order of `params` (4 x [4 x 32x32, 4 x 32, 32x32, 32, 32, 32x32, 32, 32x32, 32, 32, 32], 2 x [128x64, 128, 3x128, 3 + 1 pad]),
 * to be summed over the B slabs (fixed order) by the caller -- the backward of equivariant_scorenetwork.py:121-169. */
long long msde_escore_mol_slab_floats(void);
int msde_escore_mol_bwd(const void* const* params, const float* x0, const float* edge_attr, int ld_ea,
                        const float* basis, const int* mol_ptr, int B, const int* rowptr, const int* src,
                        const int* dst, const int* rowptr_s, const int* perm_s, int N, int E, int hidden, int heads,
                        int hidden_coff, int n_max, float p_att, float p_ffn, unsigned long long seed0,
                        const unsigned long long* seed_dev, float eps1, float eps2, const float* saved,
                        const float* g_out, float* g_x0, float* g_edge_attr, int ld_gea, float* slabs, void* stream);

/* ------------------------------------------------------------------ optimiser -------------- */
/* torch.optim.Adam step over a flat parameter buffer with per-element lr via segment table —
 * examples/pretrain_MoleculeSDE.py:331-337,156.  seg_end[S] (exclusive ends), seg_lr[S].
 * step_dev[0] holds the 1-based step count (device int, incremented by the kernel's caller). */
int msde_adam_flat(float* p, const float* g, float* m, float* v, long long n, const int* step_dev,
                   const long long* seg_end, const float* seg_lr, int S, float beta1, float beta2,
                   float eps, float weight_decay, float grad_scale, void* stream);

/* Chunk-table variants (no flattening copy): table[c] = {gradient chunk address (0 = no gradient: zeros),
 * chunk offset in the flat buffers, element count <= msde_chunk_elems()} as three int64 per chunk; chunks never
 * straddle tensors.  msde_gather_chunks copies the gradients into the flat buffer (the message of the
 * data-parallel all-reduce); msde_adam_chunks is msde_adam_flat reading the gradients through the table. */
int msde_chunk_elems(void);
/* step_counter[0] += 1 (int64: seeds of the in-kernel noise / dropout masks of a replayed step) and optimiser_step[0] += 1
 * (int32: the `step` of torch.optim.Adam, pretrain_MoleculeSDE.py:156) in one launch; either pointer may be NULL. */
int msde_step_counters(long long* step_counter, int* optimiser_step, void* stream);
int msde_gather_chunks(const long long* table, int n_chunks, float* flat, void* stream);
int msde_adam_chunks(float* p, const long long* table, int n_chunks, float* m, float* v,
                     const int* step_dev, const long long* seg_end, const float* seg_lr, int S,
                     float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                     void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MSDE_HIP_H */
