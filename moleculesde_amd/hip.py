"""torch-facing wrappers over the C ABI (include/msde_hip.h).

PyTorch is used here as plumbing only: device buffers, the current HIP stream and the autograd
tape.  Every function launches hand-written gfx950 kernels from libmsde_hip.so on torch's current
stream; tensors must live on a HIP device -- there is no CPU path (a CPU tensor raises).
"""
import ctypes

import weakref as _weakref

import torch

from . import _lib


from ._torchabi import _stream, _p, _f32, _i32  # noqa: E402


# ------------------------------------------------------------------------------------------------
# row bounds (include/msde_hip.h, "Row bounds"): the C library takes the device row count of an operand as an explicit
# argument of every reducing kernel and keeps no table.  The operators of this module see plain tensors, so while a step on
# a capacity bucket is being LAUNCHED (eagerly, or recorded into a hipGraph) the bucket's {row capacity: count tensor} map
# is in scope here -- `with hip.row_bounds(bucket.bounds_map()):`, entered by Trainer.step / Trainer.capture for the batch
# they are given and restored on exit -- and each operator hands the count of its operand's capacity to its kernel.  A
# replayed graph needs nothing in scope: the pointers are part of the recorded launches.  (The scope is process wide, not
# thread local: the backward functions, which look bounds up too, run on the autograd engine's thread.)
# ------------------------------------------------------------------------------------------------
import contextlib as _contextlib

_BOUNDS = {}              # the map in scope: {row capacity: int32 device tensor holding the valid row count}


@_contextlib.contextmanager
def row_bounds(bounds):
    """Scope in which kernels reducing over rows of a tensor with exactly `capacity` rows stop at the valid rows."""
    global _BOUNDS
    for t in bounds.values():
        assert t.dtype == torch.int32 and t.is_cuda and t.numel() >= 1
    prev, _BOUNDS = _BOUNDS, {int(c): t for c, t in bounds.items()}
    try:
        yield
    finally:
        _BOUNDS = prev


def clear_row_bounds():
    """(tests) leave every scope: exact-size tensors from here on."""
    global _BOUNDS
    _BOUNDS = {}


# ------------------------------------------------------------------------------------------------
# graph plans
# ------------------------------------------------------------------------------------------------
class CsrPlan:
    """CSR by target (canonical edge order) + transposed view by source.

    rowptr [N+1], src [E], dst [E]  : canonical order (sorted by target, ties in input order)
    rowptr_s [N+1], perm_s [E]      : by-source slot -> canonical edge id
    perm_t [E] (int64)              : canonical edge id -> position in the caller's edge_index
    E may be an upper bound (radius graph): slots >= rowptr[N] are padding (src = dst = -1).
    """

    __slots__ = ("N", "E", "rowptr", "src", "dst", "rowptr_s", "perm_s", "perm_t", "E_dev")

    def to(self, device):
        for k in ("rowptr", "src", "dst", "rowptr_s", "perm_s", "perm_t", "E_dev"):
            v = getattr(self, k, None)
            if isinstance(v, torch.Tensor):
                setattr(self, k, v.to(device))
        return self


def build_csr(edge_index, num_nodes):
    """Plan for a static edge set (bonds / extended edges).  Device agnostic torch ops: runs on the
    host at batch-collation time (preferred) or on the device if handed a device batch."""
    src, dst = edge_index[0].long(), edge_index[1].long()
    N = int(num_nodes)
    dev = edge_index.device
    perm_t = torch.argsort(dst, stable=True)
    src_c, dst_c = src[perm_t], dst[perm_t]
    ar = torch.arange(N + 1, device=dev)
    rowptr = torch.searchsorted(dst_c, ar).to(torch.int32)
    perm_s = torch.argsort(src_c, stable=True)
    rowptr_s = torch.searchsorted(src_c[perm_s], ar).to(torch.int32)
    p = CsrPlan()
    p.N, p.E = N, int(src.numel())
    p.rowptr, p.src, p.dst = rowptr, src_c.to(torch.int32), dst_c.to(torch.int32)
    p.rowptr_s, p.perm_s, p.perm_t = rowptr_s, perm_s.to(torch.int32), perm_t
    p.E_dev = None
    return p


RADIUS_TRANSPOSE_MOL_NMAX = 64      # msde_radius_transpose_mol: molecules of at most this many atoms (the host's bound n_max)


def radius_plan(pos, batch_i32, mol_ptr_i32, cutoff, E_cap, max_nbr=32, n_max=None):
    """Radius graph as a by-target CSR emitted directly on the device (schnet.py:91-93).
    Returns (plan, dist[E_cap]).  No host synchronisation: E_cap is the host-side upper bound
    sum_m n_m * min(n_m - 1, max_nbr); the true edge count stays on the device (plan.E_dev)."""
    pos = _f32(pos.detach())
    N = pos.size(0)
    dev = pos.device
    deg = torch.empty(N, dtype=torch.int32, device=dev)
    rowptr = torch.empty(N + 1, dtype=torch.int32, device=dev)
    src = torch.empty(E_cap, dtype=torch.int32, device=dev)
    dst = torch.empty(E_cap, dtype=torch.int32, device=dev)
    dist = torch.empty(E_cap, dtype=torch.float32, device=dev)
    st = _stream()
    r2 = float(cutoff) * float(cutoff)
    _lib.call("msde_radius_count", _p(pos), _p(batch_i32), _p(mol_ptr_i32), N, r2, max_nbr, _p(deg), st)
    _lib.call("msde_exclusive_scan_i32", _p(deg), _p(rowptr), N, st)
    _lib.call("msde_radius_fill", _p(pos), _p(batch_i32), _p(mol_ptr_i32), N, r2, max_nbr, _p(rowptr), _p(src),
              _p(dst), _p(dist), E_cap, st)
    # transposed view (by source): stable counting sort on the device, no torch.sort
    rowptr_s = torch.empty(N + 1, dtype=torch.int32, device=dev)
    perm_s = torch.empty(E_cap, dtype=torch.int32, device=dev)
    B = int(mol_ptr_i32.numel()) - 1
    if n_max is not None and 0 < int(n_max) <= RADIUS_TRANSPOSE_MOL_NMAX and B > 0:      # one launch, one workgroup per molecule
        _lib.call("msde_radius_transpose_mol", _p(mol_ptr_i32), B, int(n_max), _p(rowptr), _p(src), N, E_cap, _p(rowptr_s),
                  _p(perm_s), st)
    else:
        _lib.call("msde_radius_transpose", _p(batch_i32), _p(mol_ptr_i32), _p(rowptr), _p(src), N, E_cap, _p(deg),
                  _p(rowptr_s), _p(perm_s), st)
    p = CsrPlan()
    p.N, p.E = N, E_cap
    p.rowptr, p.src, p.dst = rowptr, src, dst
    p.rowptr_s, p.perm_s, p.perm_t = rowptr_s, perm_s, None
    p.E_dev = rowptr[N:]
    return p, dist


# ------------------------------------------------------------------------------------------------
# plain (non-differentiable) launches
# ------------------------------------------------------------------------------------------------
def segment_sum_rows(rows, rowptr, perm, N, mean=False, out=None, ldo=0):
    """out[i] = sum of rows over CSR row i; `out`/`ldo` let the result land in a column block of a wider
    buffer (row stride ldo floats)."""
    ldi = 0
    if rows.dtype == torch.float32 and rows.dim() == 2 and rows.stride(1) == 1 and rows.stride(0) > rows.size(1):
        ldi = rows.stride(0)            # a column block of a wider tensor (e.g. a slice of a concat's gradient)
    else:
        rows = _f32(rows)
    D = rows.size(1)
    if out is None:
        out = torch.empty(N, D, dtype=torch.float32, device=rows.device)
    _lib.call("msde_segment_sum_rows", _p(rows), ldi, _p(rowptr), _p(perm), N, D, 1.0 if mean else 0.0, _p(out),
              int(ldo), _stream())
    return out


def gather_rows(X, idx):
    X = _f32(X)
    E, D = idx.numel(), X.size(1)
    out = torch.empty(E, D, dtype=torch.float32, device=X.device)
    _lib.call("msde_gather_rows", _p(X), _p(idx), E, D, _p(out), _stream())
    return out


def rbf_cutoff(dist, E_dev, offset, coeff, cutoff):
    """GaussianSmearing + cosine cutoff (schnet.py:186,205-207) -> rbf [E,G], C [E]."""
    dist = _f32(dist)
    E = dist.numel()
    G = offset.numel()
    rbf = torch.empty(E, G, dtype=torch.float32, device=dist.device)
    C = torch.empty(E, dtype=torch.float32, device=dist.device)
    _lib.call("msde_rbf_cutoff_fwd", _p(dist), _p(E_dev), E, G, _p(_f32(offset)), float(coeff),
              float(cutoff), _p(rbf), _p(C), _stream())
    return rbf, C


FUSED_CHUNKS_PER_WG = 0   # 0 = auto: one resident wave of persistent workgroups (msde_cfconv_fused_fwd)
# Geometry of the two wide CFConv kernels when they run BESIDE latency-critical work on another stream (the trainer
# runs SchNet next to the GIN -> 2D->3D chain): at full width they hold most of every CU's LDS / registers and the
# other stream's kernels wait for a free CU.  Fewer, longer workgroups make the kernels themselves slower but the
# step faster (MI355X, bs 256: 71.8k -> 78.7k molecules/s).  None = full width (standalone use, the roofline runs).
CFCONV_FWD_WGS = None     # workgroups of the fused forward (full width: 2 per CU)
CFCONV_BWD_WGS = None     # workgroups of the fused weight-gradient kernel (full width: 1 per CU)


def _fwd_chunks_per_wg(E_cap):
    if CFCONV_FWD_WGS is None:
        return FUSED_CHUNKS_PER_WG
    chunks = (E_cap + 31) // 32
    return max(1, -(-chunks // int(CFCONV_FWD_WGS)))


def cfconv_fused_forward(x1, dist, plan, W1, b1, W2, b2, offset, coeff, cutoff, chunks_per_wg=None, want_filter=False):
    """Fused CFConv forward (no autograd): see csrc/cfconv_fused.hip.  Returns agg (and the filter rows
    Wf = (W2 h1 + b2) * C when want_filter)."""
    x1 = _f32(x1)
    N, Fd = x1.shape
    G = W1.size(1)
    agg = torch.empty(N, Fd, dtype=torch.float32, device=x1.device)
    Wf = torch.empty(plan.E, Fd, dtype=torch.float32, device=x1.device) if want_filter else None
    cpw = _fwd_chunks_per_wg(plan.E) if chunks_per_wg is None else chunks_per_wg
    _lib.call("msde_cfconv_fused_fwd", _p(x1), _p(_f32(dist)), _p(plan.rowptr), _p(plan.src), _p(plan.dst),
              _p(_f32(W1)), _p(_f32(b1)), _p(_f32(W2)), _p(_f32(b2)), _p(_f32(offset)), N, Fd, G, plan.E,
              float(coeff), float(cutoff), int(cpw), _p(agg), _p(Wf), _stream())
    return (agg, Wf) if want_filter else agg


_CF_WS = {}


def _cf_workspace(E_cap, G, device, max_wgs=0):
    n = int(_lib.load().msde_cfconv_fused_bwd_w_workspace_floats(E_cap, G, max_wgs))
    device = _ws_key(device)
    ws = _CF_WS.get(device)
    if ws is None or ws.numel() < n:
        _retire([ws])
        ws = torch.empty(n, dtype=torch.float32, device=device[0])
        _CF_WS[device] = ws
    return ws


class _CFConvFused(torch.autograd.Function):
    """Whole CFConv (smearing, filter MLP, cutoff, gather, segmented sum) as one forward kernel; the
    backward is two kernels (+ one slab reduce): g_x1 by a by-source gather over the saved filter rows,
    and all four filter-network weight gradients by the recomputing MFMA kernel of cfconv_fused_bwd.hip."""

    @staticmethod
    def forward(ctx, x1, W1, b1, W2, b2, dist, plan, offset, coeff, cutoff):
        x1, W1, b1, W2, b2 = _f32(x1), _f32(W1), _f32(b1), _f32(W2), _f32(b2)
        agg, Wf = cfconv_fused_forward(x1, dist, plan, W1, b1, W2, b2, offset, coeff, cutoff, want_filter=True)
        ctx.save_for_backward(x1, W1, b1, W2, Wf, dist, offset)
        ctx.plan, ctx.coeff, ctx.cutoff = plan, float(coeff), float(cutoff)
        return agg

    @staticmethod
    def backward(ctx, g):
        x1, W1, b1, W2, Wf, dist, offset = ctx.saved_tensors
        plan = ctx.plan
        g = _f32(g)
        N, Fd = x1.shape
        G = W1.size(1)
        st = _stream()
        g_x1 = None
        if ctx.needs_input_grad[0]:
            g_x1 = torch.empty_like(x1)
            _lib.call("msde_cfconv_aggregate_bwd_x", _p(g), _p(Wf), _p(None), _p(plan.rowptr_s), _p(plan.perm_s),
                      _p(plan.dst), N, Fd, _p(g_x1), st)
        # one buffer in the slab order [gW2 | gW1 | gb1 | gb2]: the kernel then sums the slabs with the generic
        # parallel reduction
        gall = torch.empty(Fd * Fd + Fd * G + 2 * Fd, dtype=torch.float32, device=g.device)
        gW2 = gall[:Fd * Fd].view(Fd, Fd)
        gW1 = gall[Fd * Fd:Fd * Fd + Fd * G].view(Fd, G)
        gb1 = gall[Fd * Fd + Fd * G:Fd * Fd + Fd * G + Fd]
        gb2 = gall[Fd * Fd + Fd * G + Fd:]
        mw = int(CFCONV_BWD_WGS or 0)
        if _SLABS.active:            # slabs into the arena, summed by the batched reduction of the backward pass
            nslab = int(_lib.load().msde_cfconv_fused_bwd_w_slabs(plan.E, mw))
            ws = _SLABS.alloc(nslab * gall.numel(), g.device)
            _lib.call("msde_cfconv_fused_bwd_w", _p(g), _p(x1), _p(dist), _p(plan.rowptr), _p(plan.src), _p(plan.dst),
                      _p(W1), _p(b1), _p(W2), _p(offset), N, Fd, G, plan.E, ctx.coeff, ctx.cutoff, mw, _p(None), _p(None),
                      _p(None), _p(None), _p(ws), st)
            _SLABS.add(ws.data_ptr(), nslab, gall.numel(), gall, written=True)
        else:
            ws = _cf_workspace(plan.E, G, x1.device, mw)
            _lib.call("msde_cfconv_fused_bwd_w", _p(g), _p(x1), _p(dist), _p(plan.rowptr), _p(plan.src), _p(plan.dst),
                      _p(W1), _p(b1), _p(W2), _p(offset), N, Fd, G, plan.E, ctx.coeff, ctx.cutoff, mw, _p(gW1), _p(gb1),
                      _p(gW2), _p(gb2), _p(ws), st)
        return g_x1, gW1, gb1, gW2, gb2, None, None, None, None, None


def cfconv_fused(x1, W1, b1, W2, b2, dist, plan, offset, coeff, cutoff):
    return _CFConvFused.apply(x1, W1, b1, W2, b2, dist, plan, offset, coeff, cutoff)


# ---- CFConv on unordered atom pairs (csrc/cfconv_pair.hip) -----------------------------------------------------
import os as _os_pair
CFCONV_PAIR = True     # CFConv on unordered pairs (False: the per-edge fused kernels, kept as the cross-check and for > 33 atoms)


class PairPlan:
    """Unordered atom pairs of every molecule (all a < b, row-major) with their distances; see msde_pair_build."""

    __slots__ = ("P", "pair_ptr", "pi", "pj", "pd", "count", "B", "N", "mol_ptr", "batch_i32")


def pair_capacity(pl):
    """Host-side bound on the number of pairs of a batch plan: exact when the plan was built on the host, half the
    radius-edge capacity for a capacity bucket (molecules of <= 33 atoms: every neighbour list is complete)."""
    p2 = getattr(pl, "P2_cap", None)
    return int(p2) if p2 is not None else (int(pl.E_r_cap) + 1) // 2


def pair_plan(pos, pl, cutoff):
    """Pair list + distances of the batch on the device, one launch, no host synchronisation (replaces the radius graph of
    schnet.py:91-93 for molecules the 32-neighbour cap cannot bind on)."""
    pos = _f32(pos.detach())
    dev = pos.device
    pp = PairPlan()
    pp.P, pp.B, pp.N = pair_capacity(pl), int(pl.B), int(pos.size(0))
    pp.mol_ptr, pp.batch_i32 = pl.mol_ptr, pl.batch_i32
    pp.pair_ptr = torch.empty(pp.B + 1, dtype=torch.int32, device=dev)
    n = max(pp.P, 1)
    pp.pi = torch.empty(n, dtype=torch.int32, device=dev)
    pp.pj = torch.empty(n, dtype=torch.int32, device=dev)
    pp.pd = torch.empty(n, dtype=torch.float32, device=dev)
    _lib.call("msde_pair_build", _p(pos), _p(pl.mol_ptr), pp.B, float(cutoff) * float(cutoff), _p(pp.pair_ptr), _p(pp.pi),
              _p(pp.pj), _p(pp.pd), pp.P, _p(getattr(pl, "err_dev", None)), _stream())
    pp.count = pp.pair_ptr[pp.B:]
    return pp


def _pair_blocks_per_wg(P):
    if CFCONV_FWD_WGS is None:
        return 0
    blocks = (P + 31) // 32
    return max(1, -(-blocks // int(CFCONV_FWD_WGS)))


CFCONV_FILTER_MULTI = True     # the filter rows of all interaction blocks in ONE launch at the head of SchNet's forward
CFCONV_MULTI_BLOCKS_PER_WG = 0   # 0: chosen by the library (msde_cfconv_pair_filter_multi)


def cfconv_pair_filters(pp, nets, offset, coeff, cutoff):
    """Filter rows Wf_l [P, 128] of every interaction block l from ONE launch (msde_cfconv_pair_filter_multi): they depend on
    the pair distances and on block l's filter network only, so the layer chain need not wait for them one block at a time.
    nets: [(W1, b1, W2, b2)] per block (schnet.py:141-145).  Returns the list of Wf_l (views of one buffer)."""
    import ctypes
    L = len(nets)
    Fd, G = nets[0][2].size(0), nets[0][0].size(1)
    dev = pp.pd.device
    buf = torch.empty(L, max(pp.P, 1), Fd, dtype=torch.float32, device=dev)
    keep = [[_f32(t) for t in n] for n in nets]
    arr = lambda k: ctypes.cast((ctypes.c_void_p * L)(*[n[k].data_ptr() for n in keep]), ctypes.c_void_p)
    outs = ctypes.cast((ctypes.c_void_p * L)(*[buf[l].data_ptr() for l in range(L)]), ctypes.c_void_p)
    _lib.call("msde_cfconv_pair_filter_multi", _p(pp.pd), _p(pp.count), arr(0), arr(1), arr(2), arr(3), _p(_f32(offset)), L, Fd, G,
              pp.P, float(coeff), float(cutoff), int(CFCONV_MULTI_BLOCKS_PER_WG), outs, _stream())
    return [buf[l] for l in range(L)]


def cfconv_pair_forward(x1, pp, W1, b1, W2, b2, offset, coeff, cutoff, Wf=None):
    """(agg, Wf): filter rows per unordered pair on the matrix cores (unless the caller already has them: Wf from
    cfconv_pair_filters), then the fixed-order aggregation."""
    x1 = _f32(x1)
    N, Fd = x1.shape
    G = W1.size(1)
    st = _stream()
    if Wf is None:
        Wf = torch.empty(max(pp.P, 1), Fd, dtype=torch.float32, device=x1.device)
        _lib.call("msde_cfconv_pair_filter", _p(pp.pd), _p(pp.count), _p(_f32(W1)), _p(_f32(b1)), _p(_f32(W2)), _p(_f32(b2)),
                  _p(_f32(offset)), Fd, G, pp.P, float(coeff), float(cutoff), _pair_blocks_per_wg(pp.P), _p(Wf), st)
    agg = torch.empty(N, Fd, dtype=torch.float32, device=x1.device)
    stamp("cf_agg_start")     # no-ops unless enable_stamps(): bench.py times this launch inside the captured step
    _lib.call("msde_cfconv_pair_aggregate", _p(x1), _p(Wf), _p(pp.batch_i32), _p(pp.mol_ptr), _p(pp.pair_ptr), N, pp.B, Fd,
              _p(agg), st)
    stamp("cf_agg_end")
    return agg, Wf


CFCONV_BWD_GROUP = 6     # interaction blocks per filter-weight-gradient launch inside a parameter-gradient batch (1: a launch per block)


class CfBwdBatch:
    """The filter-network weight gradients of SchNet's interaction blocks as ONE launch per `group` blocks
    (msde_cfconv_pair_bwd_w_multi).  They are parameter gradients -- nothing in the backward chain reads them -- so each
    block's backward only hands its (g_agg, x1, weights, result buffer) over; the launch goes out when `group` blocks have
    reported (the chain runs from the last block to the first) or, at the latest, with the deferred leaf kernels of the
    parameter-gradient batch (a backward pass that stops short of some block).  One object per SchNet forward."""

    def __init__(self, pp, offset, coeff, cutoff, total, group=None):
        self.pp, self.offset, self.coeff, self.cutoff = pp, offset, float(coeff), float(cutoff)
        self.total, self.group = int(total), max(1, min(int(group or CFCONV_BWD_GROUP), 8))
        self.pending, self.seen, self.hooked = [], 0, False

    def usable(self):
        return self.group > 1 and _SLABS.active

    def add(self, g, x1, W1, b1, W2, gall):
        self.pending.append((g, x1, W1, b1, W2, gall))
        self.seen += 1
        if not self.hooked:          # safety net: whatever is still pending when the backward pass ends goes out then
            self.hooked = True
            _SLABS.deferred.append(lambda st_=None: self.flush())
        if len(self.pending) >= self.group or self.seen >= self.total:
            self.flush()

    def flush(self):
        import ctypes
        items, self.pending = self.pending, []
        if not items:
            return
        pp, L = self.pp, len(items)
        N, Fd = items[0][1].shape
        G = items[0][2].size(1)
        mw = int(CFCONV_BWD_WGS or 0)
        nslab = int(_lib.load().msde_cfconv_pair_bwd_w_multi_slabs(pp.P, L, mw))
        ws = [_SLABS.alloc(nslab * it[5].numel(), it[0].device) for it in items]
        arr = lambda ts: ctypes.cast((ctypes.c_void_p * L)(*[t.data_ptr() for t in ts]), ctypes.c_void_p)
        _lib.call("msde_cfconv_pair_bwd_w_multi", arr([it[0] for it in items]), arr([it[1] for it in items]), _p(pp.pd), _p(pp.count),
                  _p(pp.pi), _p(pp.pj), arr([it[2] for it in items]), arr([it[3] for it in items]), arr([it[4] for it in items]),
                  _p(self.offset), L, N, Fd, G, pp.P, self.coeff, self.cutoff, mw, arr(ws), _stream())
        for it, w in zip(items, ws):
            _SLABS.add(w.data_ptr(), nslab, it[5].numel(), it[5], written=True)
        _SLABS.launched.append(items)      # operands stay referenced until the batch is finished


class _CFConvPair(torch.autograd.Function):
    """CFConv (schnet.py:141-145,185-195) on unordered pairs: forward = filter kernel + aggregation; backward = the same
    aggregation applied to the incoming gradient (input gradient) and the pair form of the recomputing weight-gradient
    kernel (both directions of a pair summed before its products)."""

    @staticmethod
    def forward(ctx, x1, W1, b1, W2, b2, pp, offset, coeff, cutoff, Wf_pre=None, bwd_batch=None):
        x1, W1, b1, W2, b2 = _f32(x1), _f32(W1), _f32(b1), _f32(W2), _f32(b2)
        agg, Wf = cfconv_pair_forward(x1, pp, W1, b1, W2, b2, offset, coeff, cutoff, Wf=Wf_pre)
        ctx.save_for_backward(x1, W1, b1, W2, Wf, offset)
        ctx.pp, ctx.coeff, ctx.cutoff, ctx.bwd_batch = pp, float(coeff), float(cutoff), bwd_batch
        return agg

    @staticmethod
    def backward(ctx, g):
        x1, W1, b1, W2, Wf, offset = ctx.saved_tensors
        pp = ctx.pp
        g = _f32(g)
        N, Fd = x1.shape
        G = W1.size(1)
        st = _stream()
        g_x1 = None
        if ctx.needs_input_grad[0]:
            g_x1 = torch.empty_like(x1)
            _lib.call("msde_cfconv_pair_aggregate", _p(g), _p(Wf), _p(pp.batch_i32), _p(pp.mol_ptr), _p(pp.pair_ptr), N, pp.B,
                      Fd, _p(g_x1), st)
        gall = torch.empty(Fd * Fd + Fd * G + 2 * Fd, dtype=torch.float32, device=g.device)
        gW2 = gall[:Fd * Fd].view(Fd, Fd)
        gW1 = gall[Fd * Fd:Fd * Fd + Fd * G].view(Fd, G)
        gb1 = gall[Fd * Fd + Fd * G:Fd * Fd + Fd * G + Fd]
        gb2 = gall[Fd * Fd + Fd * G + Fd:]
        mw = int(CFCONV_BWD_WGS or 0)
        args = (_p(g), _p(x1), _p(pp.pd), _p(pp.count), _p(pp.pi), _p(pp.pj), _p(W1), _p(b1), _p(W2), _p(offset), N, Fd, G,
                pp.P, ctx.coeff, ctx.cutoff, mw)
        if ctx.bwd_batch is not None and ctx.bwd_batch.usable():
            ctx.bwd_batch.add(g, x1, W1, b1, W2, gall)        # one launch for `group` blocks (CfBwdBatch)
        elif _SLABS.active:
            nslab = int(_lib.load().msde_cfconv_fused_bwd_w_slabs(pp.P, mw))
            ws = _SLABS.alloc(nslab * gall.numel(), g.device)
            _lib.call("msde_cfconv_pair_bwd_w", *args, _p(None), _p(None), _p(None), _p(None), _p(ws), st)
            _SLABS.add(ws.data_ptr(), nslab, gall.numel(), gall, written=True)
        else:
            ws = _cf_workspace(pp.P, G, x1.device, mw)
            _lib.call("msde_cfconv_pair_bwd_w", *args, _p(gW1), _p(gb1), _p(gW2), _p(gb2), _p(ws), st)
        return g_x1, gW1, gb1, gW2, gb2, None, None, None, None, None, None


def cfconv_pair(x1, W1, b1, W2, b2, pp, offset, coeff, cutoff, Wf=None, bwd_batch=None):
    """Wf: this block's filter rows when the caller computed them for all blocks at once (cfconv_pair_filters); bwd_batch: the
    CfBwdBatch of this forward pass (the blocks' filter-weight gradients as one launch)."""
    return _CFConvPair.apply(x1, W1, b1, W2, b2, pp, offset, coeff, cutoff, Wf, bwd_batch)


def edge_geometry(pos, plan, Wd, Wc):
    """Per-edge frame + Fourier features (SDE_model_2D_to_3D.py:35-66,342-369); no gradient: the
    perturbed coordinates do not depend on any parameter."""
    pos = _f32(pos.detach())
    E, C = plan.E, Wd.numel()
    dev = pos.device
    feat_d = torch.empty(E, 2 * C, dtype=torch.float32, device=dev)
    feat_i = torch.empty(E, 4 * C, dtype=torch.float32, device=dev)
    feat_j = torch.empty(E, 4 * C, dtype=torch.float32, device=dev)
    angle = torch.empty(E, 2, dtype=torch.float32, device=dev)
    basis = torch.empty(E, 9, dtype=torch.float32, device=dev)
    _lib.call("msde_edge_geometry_fwd", _p(pos), _p(plan.src), _p(plan.dst), E, _p(_f32(Wd.detach())),
              _p(_f32(Wc.detach())), C, _p(feat_d), _p(feat_i), _p(feat_j), _p(angle), _p(basis), _stream())
    return feat_d, feat_i, feat_j, angle, basis


# ------------------------------------------------------------------------------------------------
# differentiable ops
# ------------------------------------------------------------------------------------------------
class _EmbeddingSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tab, codes, list_ptr, list_nodes):
        tab = _f32(tab)
        N, K = codes.shape
        D = tab.size(1)
        out = torch.empty(N, D, dtype=torch.float32, device=tab.device)
        _lib.call("msde_embedding_sum_fwd", _p(tab), _p(codes), N, K, D, _p(out), _stream())
        ctx.save_for_backward(list_ptr, list_nodes)
        ctx.R, ctx.D = tab.size(0), D
        return out

    @staticmethod
    def backward(ctx, g):
        list_ptr, list_nodes = ctx.saved_tensors
        g = _f32(g)
        g_tab = torch.empty(ctx.R, ctx.D, dtype=torch.float32, device=g.device)
        nfl = int(_lib.load().msde_embedding_sum_bwd_workspace_floats(ctx.R, ctx.D, EMB_BWD_SPLIT))
        R, D = ctx.R, ctx.D
        if _SLABS.active:
            # the table gradient is a leaf gradient: queued like the GIN bond-table gradients (own workspace in the arena;
            # only the ADDRESS of g_tab is kept, see _SlabBatch.add)
            ws = _SLABS.alloc(nfl, g.device)
            out_ptr = g_tab.data_ptr()

            def launch(st_=None, g=g, list_ptr=list_ptr, list_nodes=list_nodes, ws=ws):
                _lib.call("msde_embedding_sum_bwd", _p(g), _p(list_ptr), _p(list_nodes), R, D, EMB_BWD_SPLIT,
                          ctypes.c_void_p(out_ptr), _p(ws), st_ if st_ is not None else _stream())
            _SLABS.deferred.append(launch)
        else:
            ws = _scratch(nfl, g.device)
            _lib.call("msde_embedding_sum_bwd", _p(g), _p(list_ptr), _p(list_nodes), R, D, EMB_BWD_SPLIT, _p(g_tab),
                      _p(ws), _stream())
        return g_tab, None, None, None


def embedding_sum(tab, codes, list_ptr, list_nodes):
    """out[i] = sum_k tab[codes[i,k]]; lists = CSR of nodes per table row (for the backward)."""
    return _EmbeddingSum.apply(tab, codes, list_ptr, list_nodes)


# (Round 3 measurement, removed from the product: the aggregation backward emitting the BatchNorm-backward partial sums of its
# result -- msde_gin_aggregate_bwd_x_stats, still in the library and in tests/test_gpu_kernels.py -- was 27.7 us against 13.0 +
# 6.4 us for the separate column-statistics launch; step 2.91 vs 2.86 ms.)


class BnLink:
    """What a fused GIN layer hands to the NEXT layer's aggregation so that its outer BatchNorm (+ ReLU) is applied on the
    fly there instead of by a launch of its own (hip._GinMlpBN with defer_apply): z = the second product, vec = scale |
    shift | mean | rstd."""

    __slots__ = ("z", "vec", "relu")

    def __init__(self, z, vec, relu):
        self.z, self.vec, self.relu = z, vec, bool(relu)


class _GinAggregate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, tab, eps, plan, codes, link=None):
        x, tab, eps = _f32(x), _f32(tab), _f32(eps)
        N, D = x.shape
        out = torch.empty_like(x)
        if link is not None:
            # x is allocated but NOT yet written: this kernel forms it from link.z and writes it
            _lib.call("msde_gin_aggregate_bn_fwd", _p(link.z), _p(link.vec[0]), _p(link.vec[1]), int(link.relu), _p(tab),
                      _p(codes), _p(eps), _p(plan.rowptr), _p(plan.src), N, D, _p(x), _p(out), _stream())
        else:
            _lib.call("msde_gin_aggregate_fwd", _p(x), _p(tab), _p(codes), _p(eps), _p(plan.rowptr), _p(plan.src), N, D,
                      _p(out), _stream())
        ctx.save_for_backward(x, tab, eps, codes)
        ctx.plan, ctx.link = plan, link
        return out

    @staticmethod
    def backward(ctx, g):
        x, tab, eps, codes = ctx.saved_tensors
        plan, link = ctx.plan, ctx.link
        g = _f32(g)
        N, D = x.shape
        R = tab.size(0)
        st = _stream()
        g_x = torch.empty_like(x)
        _lib.call("msde_gin_aggregate_bwd_x", _p(g), _p(x), _p(tab), _p(codes), _p(eps), _p(plan.rowptr_s),
                  _p(plan.perm_s), _p(plan.dst), N, D, _p(g_x), st)
        g_tab = torch.empty_like(tab)
        g_eps = torch.empty(1, dtype=torch.float32, device=x.device)
        nfl = int(_lib.load().msde_gin_aggregate_bwd_tab_workspace_floats(N, plan.E, D, R))
        if _SLABS.active:            # partial tables into the arena, summed by the batched reduction
            nslab = int(_lib.load().msde_gin_aggregate_bwd_tab_slabs(N, plan.E))
            ws = _SLABS.alloc(nfl, x.device)
            E_ = plan.E

            def launch(st_=None, g=g, x=x, tab=tab, codes=codes, plan=plan, ws=ws):
                _lib.call("msde_gin_aggregate_bwd_tab", _p(g), _p(x), _p(tab), _p(codes), _p(plan.src), _p(plan.dst), N,
                          E_, D, R, _p(None), _p(None), _p(ws), _p(bound_tensor(N)), _p(bound_tensor(E_)),
                          st_ if st_ is not None else _stream())
            # a parameter gradient nothing in the backward chain reads: queued (operands kept alive) and launched by
            # run_deferred_leaf_kernels() -- the trainer runs them beside the grouped weight-gradient launch; the
            # layers of one graph go out as ONE launch there (msde_gin_aggregate_bwd_tab_multi)
            launch.gin_tab = (g, x, tab, codes, plan, ws, N, E_, D, R)
            _SLABS.deferred.append(launch)
            _SLABS.add(ws.data_ptr(), nslab, R * D, g_tab)
            _SLABS.add(ws.data_ptr() + 4 * nslab * R * D, nslab, 1, g_eps)
        else:
            ws = _scratch(nfl, x.device)
            _lib.call("msde_gin_aggregate_bwd_tab", _p(g), _p(x), _p(tab), _p(codes), _p(plan.src), _p(plan.dst), N,
                      plan.E, D, R, _p(g_tab), _p(g_eps), _p(ws), _p(bound_tensor(N)), _p(bound_tensor(plan.E)), st)
        return g_x, g_tab, g_eps, None, None, None


def bn_apply_link(h, link):
    """Materialise a deferred BatchNorm output (see BnLink) with a launch of its own."""
    M, D = link.z.shape
    _lib.call("msde_affine_cols", _p(link.z), M, D, _p(link.vec[0]), _p(link.vec[1]), int(link.relu), _p(h), _stream())


def gin_aggregate(x, tab, eps, plan, codes, link=None):
    """(1+eps) x_i + sum_j relu(x_j + bond_emb(e_ji))  (molecule_gnn_model.py:22-29).  link: x is the not yet applied
    BatchNorm output of the previous fused layer (BnLink)."""
    return _GinAggregate.apply(x, tab, eps, plan, codes, link)


class _CFConvAggregate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x1, Wf, C, plan):
        x1, Wf = _f32(x1), _f32(Wf)
        N, Fd = x1.shape
        agg = torch.empty(N, Fd, dtype=torch.float32, device=x1.device)
        _lib.call("msde_cfconv_aggregate_fwd", _p(x1), _p(Wf), _p(C), _p(plan.rowptr), _p(plan.src), N, Fd, _p(agg),
                  _stream())
        ctx.save_for_backward(x1, Wf, C)
        ctx.plan = plan
        return agg

    @staticmethod
    def backward(ctx, g):
        x1, Wf, C = ctx.saved_tensors
        plan = ctx.plan
        g = _f32(g)
        N, Fd = x1.shape
        st = _stream()
        g_x1 = g_Wf = None
        if ctx.needs_input_grad[0]:
            g_x1 = torch.empty_like(x1)
            _lib.call("msde_cfconv_aggregate_bwd_x", _p(g), _p(Wf), _p(C), _p(plan.rowptr_s), _p(plan.perm_s),
                      _p(plan.dst), N, Fd, _p(g_x1), st)
        if ctx.needs_input_grad[1]:
            g_Wf = torch.empty_like(Wf)
            _lib.call("msde_cfconv_aggregate_bwd_w", _p(g), _p(x1), _p(C), _p(plan.rowptr), _p(plan.src), N, Fd,
                      Wf.size(0), _p(g_Wf), st)
        return g_x1, g_Wf, None, None


def cfconv_aggregate(x1, Wf, C, plan):
    """agg_i = sum_j x1_j * (Wf_ij * C_ij)  (schnet.py:187,190,194-195); C carries no gradient here."""
    return _CFConvAggregate.apply(x1, Wf, C, plan)


# ---- twice-differentiable edge primitives (MD17 force path: F = -dE/dpos with create_graph=True, then
# loss(F).backward(); finetune_MD17.py:47-78).  The three bilinear maps below are closed under
# differentiation -- each backward is expressed with the other two -- so autograd can differentiate the
# backward graph again.
class _EdgeAgg(torch.autograd.Function):
    """agg(x, W)[i] = sum_{e: dst_e = i} x[src_e] * W[e]"""

    @staticmethod
    def forward(ctx, x, W, plan):
        x, W = _f32(x), _f32(W)
        N, Fd = x.shape
        out = torch.empty(N, Fd, dtype=torch.float32, device=x.device)
        _lib.call("msde_cfconv_aggregate_fwd", _p(x), _p(W), _p(None), _p(plan.rowptr), _p(plan.src), N, Fd, _p(out),
                  _stream())
        ctx.save_for_backward(x, W)
        ctx.plan = plan
        return out

    @staticmethod
    def backward(ctx, g):
        x, W = ctx.saved_tensors
        return _EdgeAggT.apply(g, W, ctx.plan), _EdgeProd.apply(g, x, ctx.plan), None


class _EdgeAggT(torch.autograd.Function):
    """aggT(g, W)[j] = sum_{e: src_e = j} g[dst_e] * W[e]"""

    @staticmethod
    def forward(ctx, g, W, plan):
        g, W = _f32(g), _f32(W)
        N, Fd = g.shape
        out = torch.empty(N, Fd, dtype=torch.float32, device=g.device)
        _lib.call("msde_cfconv_aggregate_bwd_x", _p(g), _p(W), _p(None), _p(plan.rowptr_s), _p(plan.perm_s), _p(plan.dst),
                  N, Fd, _p(out), _stream())
        ctx.save_for_backward(g, W)
        ctx.plan = plan
        return out

    @staticmethod
    def backward(ctx, u):
        g, W = ctx.saved_tensors
        return _EdgeAgg.apply(u, W, ctx.plan), _EdgeProd.apply(g, u, ctx.plan), None


class _EdgeProd(torch.autograd.Function):
    """prod(g, x)[e] = g[dst_e] * x[src_e]   (padded rows zero)"""

    @staticmethod
    def forward(ctx, g, x, plan):
        g, x = _f32(g), _f32(x)
        N, Fd = x.shape
        out = torch.empty(plan.E, Fd, dtype=torch.float32, device=x.device)
        _lib.call("msde_cfconv_aggregate_bwd_w", _p(g), _p(x), _p(None), _p(plan.rowptr), _p(plan.src), N, Fd, plan.E,
                  _p(out), _stream())
        ctx.save_for_backward(g, x)
        ctx.plan = plan
        return out

    @staticmethod
    def backward(ctx, U):
        g, x = ctx.saved_tensors
        return _EdgeAgg.apply(x, U, ctx.plan), _EdgeAggT.apply(g, U, ctx.plan), None


def edge_aggregate_dd(x, W, plan):
    """Twice-differentiable  agg_i = sum_{e -> i} x[src_e] * W[e]  (W must already include the cutoff)."""
    return _EdgeAgg.apply(x, W, plan)


class _PairGatherAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, A, B, plan):
        A, B = _f32(A), _f32(B)
        E, D = plan.E, A.size(1)
        out = torch.empty(E, D, dtype=torch.float32, device=A.device)
        _lib.call("msde_pair_gather_add", _p(A), _p(B), 0, _p(plan.src), _p(plan.dst), E, D, _p(out), _stream())
        ctx.plan = plan
        return out

    @staticmethod
    def backward(ctx, g):
        plan = ctx.plan                 # g may be a column block of a concat's gradient: summed without a copy
        g_A = segment_sum_rows(g, plan.rowptr_s, plan.perm_s, plan.N) if ctx.needs_input_grad[0] else None
        g_B = segment_sum_rows(g, plan.rowptr, None, plan.N) if ctx.needs_input_grad[1] else None
        return g_A, g_B, None


class _PairGatherAddSame(torch.autograd.Function):
    """out[e] = x[src_e] + x[dst_e]: ONE gradient kernel walks the by-source and the by-target segments of a node
    (two segment sums and autograd's add of the two gradients otherwise)."""

    @staticmethod
    def forward(ctx, x, plan):
        x = _f32(x)
        E, D = plan.E, x.size(1)
        out = torch.empty(E, D, dtype=torch.float32, device=x.device)
        _lib.call("msde_pair_gather_add", _p(x), _p(x), 0, _p(plan.src), _p(plan.dst), E, D, _p(out), _stream())
        ctx.plan = plan
        return out

    @staticmethod
    def backward(ctx, g):
        plan = ctx.plan
        g = g if (g.is_cuda and g.dtype == torch.float32 and g.stride(-1) == 1) else _f32(g)
        D = g.size(1)
        out = torch.empty(plan.N, D, dtype=torch.float32, device=g.device)
        _lib.call("msde_segment_sum_rows2", _p(g), _row_stride(g, D), _p(plan.rowptr_s), _p(plan.perm_s), _p(plan.rowptr),
                  _p(None), plan.N, D, 0.0, _p(out), D, _stream())
        return out, None


def pair_gather_add(A, B, plan):
    """out[e] = A[row_e] + B[col_e]  (row = source, col = target)."""
    if A is B:
        return _PairGatherAddSame.apply(A, plan)
    return _PairGatherAdd.apply(A, B, plan)


class _PairGatherAddCols(torch.autograd.Function):
    """out[e] = AB[src_e, :D] + AB[dst_e, D:] for AB = [A | B] of one GEMM; the backward writes the two segment
    sums straight into the column blocks of g_AB."""

    @staticmethod
    def forward(ctx, AB, plan):
        AB = _f32(AB)
        E, D = plan.E, AB.size(1) // 2
        out = torch.empty(E, D, dtype=torch.float32, device=AB.device)
        _lib.call("msde_pair_gather_add", _p(AB), AB.data_ptr() + 4 * D, 2 * D, _p(plan.src), _p(plan.dst), E, D, _p(out),
                  _stream())
        ctx.plan, ctx.D = plan, D
        return out

    @staticmethod
    def backward(ctx, g):
        plan, D = ctx.plan, ctx.D
        g = _f32(g)
        g_AB = torch.empty(plan.N, 2 * D, dtype=torch.float32, device=g.device)
        segment_sum_rows(g, plan.rowptr_s, plan.perm_s, plan.N, out=g_AB, ldo=2 * D)
        segment_sum_rows(g, plan.rowptr, None, plan.N, out=g_AB[:, D:], ldo=2 * D)
        return g_AB, None


def pair_gather_add_cols(AB, plan):
    return _PairGatherAddCols.apply(AB, plan)


class _PairGatherCat(torch.autograd.Function):
    """[x[src_e] + x[dst_e] | c[e]]: cat([h_row + h_col, edge_attr]) of equivariant_scorenetwork.py:154-157 written by the
    gather kernel itself (no separate concatenation); the backward sums the left column block over both CSRs in one
    kernel and hands the right block on as a view."""

    @staticmethod
    def forward(ctx, x, c, plan):
        x, c = _f32(x), _f32(c)
        E, D, D2 = plan.E, x.size(1), c.size(1)
        out = torch.empty(E, D + D2, dtype=torch.float32, device=x.device)
        _lib.call("msde_pair_gather_cat", _p(x), D, _p(c), D2, _p(plan.src), _p(plan.dst), E, D, D2, _p(out), _stream())
        ctx.plan, ctx.D = plan, D
        return out

    @staticmethod
    def backward(ctx, g):
        plan, D = ctx.plan, ctx.D
        g = g if (g.is_cuda and g.dtype == torch.float32 and g.stride(-1) == 1) else _f32(g)
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty(plan.N, D, dtype=torch.float32, device=g.device)
            _lib.call("msde_segment_sum_rows2", _p(g), _row_stride(g, g.size(1)), _p(plan.rowptr_s), _p(plan.perm_s),
                      _p(plan.rowptr), _p(None), plan.N, D, 0.0, _p(gx), D, _stream())
        return gx, (g[:, D:] if ctx.needs_input_grad[1] else None), None


def pair_gather_cat(x, c, plan):
    """cat([x[row] + x[col], c], -1) for per-edge rows c."""
    return _PairGatherCat.apply(x, c, plan)


class _MlpFused(torch.autograd.Function):
    """Linear -> act -> Linear (-> act -> Linear ...) on csrc/gemm_ex.hip: bias + activation in the GEMM epilogues (the
    pre-activation is stored beside the output), the activation's derivative in the epilogue of the input-gradient GEMM
    of the layer above, weight gradients queued for the grouped split-M launch.  A last layer with <= 4 outputs behind a
    SiLU runs on the row kernels msde_mlp_head_fwd/_bwd (activation applied while reading the pre-activation, own weight
    gradient accumulated in the backward kernel).  Replaces the GEMM / activation / GEMM operator chains of
    layers/common.py:5-40 and equivariant_scorenetwork.py:142-146."""

    @staticmethod
    def forward(ctx, x, act, offload, *params):
        x = x if (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1) else _f32(x)
        n = len(params) // 2
        need = any(ctx.needs_input_grad)
        M = x.size(0)
        Wl = params[2 * n - 2]
        head = (n >= 2 and act == "silu" and Wl.size(0) <= 4 and Wl.size(1) % 4 == 0 and Wl.size(1) <= 256
                and Wl.is_contiguous() and Wl.data_ptr() % 16 == 0)
        h, saved = x, []
        for i in range(n):
            W, b = params[2 * i], params[2 * i + 1]
            last = i == n - 1
            out = torch.empty(M, W.size(0), dtype=torch.float32, device=x.device)
            if last and head:              # h is the PRE-activation of the layer below
                _lib.call("msde_mlp_head_fwd", _p(h), _ld(h), _p(W), _p(b), M, W.size(1), W.size(0), _p(out), _stream())
                saved += [None, None]
            elif head and i == n - 2:      # only the pre-activation is stored: the head kernels apply the SiLU on load
                gemm_fwd(h, W, out, bias=b)
                saved += [h, out]
            else:
                Z = torch.empty_like(out) if (need and not last) else None
                gemm_fwd(h, W, out, bias=b, act=None if last else act, Z=Z)
                saved += [h, Z]
            h = out
        ctx.save_for_backward(*[t for t in saved if t is not None], *params)
        ctx.layout = [t is not None for t in saved]
        ctx.act, ctx.n, ctx.head = act, n, head
        ctx.deferrable = offload and all(t is None or t.is_leaf or getattr(t, "_msde_leaf_like", False) for t in params)
        return h

    @staticmethod
    def backward(ctx, g):
        n, act = ctx.n, ctx.act
        it = iter(ctx.saved_tensors)
        saved = [next(it) if present else None for present in ctx.layout]
        params = list(it)
        g = g if (g.is_cuda and g.dtype == torch.float32 and g.is_contiguous()) else _f32(g)
        grads = [None] * (2 * n)
        top = n - 1
        if ctx.head:
            # last layer on the row kernel: d/dZ of the layer below (SiLU' included) + its own [gW | gb] slabs
            W, b = params[2 * n - 2], params[2 * n - 1]
            Zp = saved[2 * (n - 2) + 1]
            E, H, J = Zp.size(0), W.size(1), W.size(0)
            gz = torch.empty(E, H, dtype=torch.float32, device=g.device)
            gall = torch.empty((J * H + J + 3) & ~3, dtype=torch.float32, device=g.device)   # slab size: whole float4
            nslab = int(_lib.load().msde_mlp_head_bwd_slabs(E, H))
            if _SLABS.active and ctx.deferrable:
                ws = _SLABS.alloc(nslab * gall.numel(), g.device)
                _lib.call("msde_mlp_head_bwd", _p(Zp), _ld(Zp), _p(W), _p(g), E, H, J, _p(gz), _p(None), _p(ws), _p(bound_tensor(E)),
                          _stream())
                _SLABS.add(ws.data_ptr(), nslab, gall.numel(), gall, written=True)
            else:
                ws = torch.empty(nslab * gall.numel(), dtype=torch.float32, device=g.device)
                _lib.call("msde_mlp_head_bwd", _p(Zp), _ld(Zp), _p(W), _p(g), E, H, J, _p(gz), _p(gall), _p(ws), _p(bound_tensor(E)),
                          _stream())
            grads[2 * n - 2] = gall[:J * H].view(J, H)
            grads[2 * n - 1] = gall[J * H:J * H + J] if b is not None else None
            g, top = gz, n - 2
        for i in range(top, -1, -1):           # invariant: g = d/d(pre-activation of layer i)
            W, b = params[2 * i], params[2 * i + 1]
            h_in = saved[2 * i]
            if ctx.needs_input_grad[3 + 2 * i] or (b is not None and ctx.needs_input_grad[4 + 2 * i]):
                grads[2 * i], grads[2 * i + 1] = weight_grad(g, h_in, b is not None, ctx.deferrable)
            if i == 0 and not ctx.needs_input_grad[0]:
                g = None
                break
            gin = torch.empty(g.size(0), W.size(1), dtype=torch.float32, device=g.device)
            # d/d(input of layer i) = g W; for i > 0 that input is act(Z_{i-1}): times act'(Z_{i-1}) in the same epilogue
            gemm_dgrad(g, W, gin, act=act if i > 0 else None, dact_from=saved[2 * i - 1] if i > 0 else None)
            g = gin
        return (g, None, None) + tuple(grads)


def mlp_fused(x, layers, act="silu", offload=True):
    """layers: [(weight, bias), ...] of consecutive nn.Linear; `act` between them (none after the last)."""
    flat = []
    for W, b in layers:
        flat += [W, b]
    return _MlpFused.apply(x, act, offload, *flat)


class _CatParams(torch.autograd.Function):
    """torch.cat(params, 0) that costs nothing when the parameters already lie back to back in memory (FlatAdam
    lays fusion sets out that way): the result aliases them.  Falls back to a real concatenation otherwise."""

    @staticmethod
    def forward(ctx, *ws):
        ctx.rows = [w.size(0) for w in ws]
        w0 = ws[0]
        tail = tuple(w0.shape[1:])
        adjacent = all(w.is_contiguous() and w.dtype == torch.float32 for w in ws)
        if adjacent:
            base = w0.untyped_storage().data_ptr()
            nxt = w0.data_ptr()
            for w in ws:
                if w.untyped_storage().data_ptr() != base or w.data_ptr() != nxt:
                    adjacent = False
                    break
                nxt += 4 * w.numel()
        if not adjacent:
            return torch.cat([w.detach() for w in ws], 0)
        total = sum(ctx.rows)
        size = (total,) + tail
        stride, acc = [], 1
        for d in reversed(size):
            stride.append(acc)
            acc *= d
        return torch.as_strided(w0.detach(), size, tuple(reversed(stride)), w0.storage_offset())

    @staticmethod
    def backward(ctx, g):
        return tuple(g.split(ctx.rows, 0))


_CAT_CACHE = {}      # inference: concatenations of parameters that do NOT lie back to back (no FlatAdam laid them out)


def cat_params(ws):
    ws = list(ws)
    if len(ws) == 1:
        return ws[0]
    if not torch.is_grad_enabled() and all(isinstance(w, torch.nn.Parameter) for w in ws):
        # Without autograd (sampling, evaluation) a real concatenation is made once per parameter version, not per call, and --
        # being the same tensor every time -- its transposed copy is cached by weight_t like a parameter's (2 launches per
        # stacked layer and score-network call otherwise: 16 of the sampler's ~140 launches per iteration).
        key = tuple(id(w) for w in ws)
        ver = tuple(w._version for w in ws) + (weight_epoch(),) + tuple(w.data_ptr() for w in ws)
        hit = _CAT_CACHE.get(key)
        if hit is not None and hit[0] == ver:
            return hit[1]
        if not torch.cuda.is_current_stream_capturing():
            if hit is not None and hit[1].data_ptr() != ws[0].data_ptr():
                # parameters changed: refill the SAME buffer (the transposed copy cached for it stays addressed to it)
                out = hit[1]
                torch.cat([w.detach() for w in ws], 0, out=out)
                _CAT_CACHE[key] = (ver, out, hit[2])
                # re-laid-out copies made FROM this buffer (weight_t keys end with its address) may have been "refreshed" from
                # its old contents by a training step's batched refresh in between: stale whatever their bookkeeping says
                for k_, e_ in _WT.items():
                    if isinstance(k_, tuple) and k_ and k_[-1] == out.data_ptr():
                        e_["versions"] = None
                return out
            out = _CatParams.apply(*ws)
            out._msde_leaf_like = all(w.is_leaf for w in ws)
            out._msde_src = tuple(ws)
            out._msde_volatile = False            # the cached tensor is stable: the copy made from it may be cached too
            refs = [_weakref.ref(w, lambda _r, key=key: _CAT_CACHE.pop(key, None)) for w in ws]
            _CAT_CACHE[key] = (ver, out, refs)
            return out
    out = _CatParams.apply(*ws)
    out._msde_leaf_like = all(w.is_leaf for w in ws)     # its gradient only gets split into views for the leaves
    out._msde_src = tuple(ws)                            # weight_t(): the copy is valid while these are unchanged
    out._msde_volatile = out.data_ptr() != ws[0].data_ptr()   # a real concatenation: a fresh tensor at every call
    return out


class _EdgeAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, ee, plan, heads, p_drop, seed, seed_dev):
        q, k, v, ee = _f32(q), _f32(k), _f32(v), _f32(ee)
        N, D = q.shape
        Ch = D // heads
        alpha = torch.empty(plan.E, heads, dtype=torch.float32, device=q.device)
        out = torch.empty_like(q)
        _lib.call("msde_edge_attention_fwd", _p(q), _p(k), _p(v), _p(None), D, _p(ee), 0, _p(plan.rowptr), _p(plan.src), N,
                  heads, Ch, float(p_drop), int(seed), _p(seed_dev), _p(alpha), _p(out), _stream())
        ctx.save_for_backward(q, k, v, ee, alpha)
        ctx.plan, ctx.heads, ctx.p_drop, ctx.seed, ctx.seed_dev = plan, heads, float(p_drop), int(seed), seed_dev
        return out

    @staticmethod
    def backward(ctx, g):
        q, k, v, ee, alpha = ctx.saved_tensors
        plan, H = ctx.plan, ctx.heads
        g = _f32(g)
        N, D = q.shape
        g_q = torch.empty_like(q)
        g_ee = torch.empty_like(ee)
        g_kpe = torch.empty_like(ee)
        g_vpe = torch.empty_like(ee)
        _lib.call("msde_edge_attention_bwd", _p(g), _p(q), _p(k), _p(v), D, _p(None), D, _p(ee), 0, _p(alpha),
                  _p(plan.rowptr), _p(plan.src), N, H, D // H, ctx.p_drop, ctx.seed, _p(ctx.seed_dev), _p(g_q), _p(g_ee),
                  _p(g_kpe), _p(g_vpe), 0, _stream())
        g_k = segment_sum_rows(g_kpe, plan.rowptr_s, plan.perm_s, N)
        g_v = segment_sum_rows(g_vpe, plan.rowptr_s, plan.perm_s, N)
        return g_q, g_k, g_v, g_ee, None, None, None, None, None


class _EdgeAttentionFused(torch.autograd.Function):
    """TransformerConv with ONE fused projection: qkvs = [q | k | v | skip] as column blocks of a [N, 4D]
    buffer (one GEMM instead of four); out = attention(q, k, v, ee) + skip.  The backward writes g_q and
    g_skip straight into their column blocks of g_qkvs and lets the two by-source segment sums land in
    the k and v blocks, so the projection needs one dgrad + one wgrad.

    `ee` may itself be a column block of a wider projection shared by several layers (all GAT layers project
    the SAME edge features): ee_all [E, L*D], this layer's block starting at column `col`.  The layers' backward
    passes then fill their blocks of ONE gradient buffer (`shared`, a dict owned by the caller); the first to run
    allocates and returns it, the others return None, and autograd hands the completed buffer to the shared
    projection's backward."""

    @staticmethod
    def forward(ctx, qkvs, ee, plan, heads, p_drop, seed, seed_dev, col=0, shared=None):
        qkvs, ee = _f32(qkvs), _f32(ee)
        N, D4 = qkvs.shape
        D = D4 // 4
        Ch = D // heads
        ld_ee = ee.size(1)
        alpha = torch.empty(plan.E, heads, dtype=torch.float32, device=qkvs.device)
        out = torch.empty(N, D, dtype=torch.float32, device=qkvs.device)
        base = qkvs.data_ptr()
        pq, pk, pv, ps = (ctypes.c_void_p(base + 4 * D * j) for j in range(4))
        _lib.call("msde_edge_attention_fwd", pq, pk, pv, ps, D4, ctypes.c_void_p(ee.data_ptr() + 4 * col), ld_ee,
                  _p(plan.rowptr), _p(plan.src), N, heads, Ch, float(p_drop), int(seed), _p(seed_dev), _p(alpha), _p(out),
                  _stream())
        ctx.save_for_backward(qkvs, ee, alpha)
        ctx.plan, ctx.heads, ctx.p_drop, ctx.seed, ctx.seed_dev = plan, heads, float(p_drop), int(seed), seed_dev
        ctx.col, ctx.shared = col, shared
        if shared is not None and col == 0:
            shared.clear()       # a backward pass that aborted midway must not leave a stale gradient buffer behind
        return out

    @staticmethod
    def backward(ctx, g):
        qkvs, ee, alpha = ctx.saved_tensors
        plan, H = ctx.plan, ctx.heads
        g = _f32(g)
        N, D4 = qkvs.shape
        D = D4 // 4
        E, ld_ee = ee.shape
        g_qkvs = torch.empty_like(qkvs)
        first = True
        if ctx.shared is None or ld_ee == D:
            g_ee = torch.empty_like(ee)
        else:
            g_ee = ctx.shared.get("g_ee")
            first = g_ee is None
            if first:
                g_ee = ctx.shared["g_ee"] = torch.empty_like(ee)
                ctx.shared["left"] = ld_ee // D
            ctx.shared["left"] -= 1
            if ctx.shared["left"] == 0:
                del ctx.shared["g_ee"]          # the buffer now lives in the autograd graph only
        g_kv = torch.empty(E, 2 * D, dtype=torch.float32, device=g.device)     # [g_kpe | g_vpe] per edge
        base, gbase = qkvs.data_ptr(), g_qkvs.data_ptr()
        pq, pk, pv = (ctypes.c_void_p(base + 4 * D * j) for j in range(3))
        gq, gk, gv, gs = (ctypes.c_void_p(gbase + 4 * D * j) for j in range(4))
        _lib.call("msde_edge_attention_bwd", _p(g), pq, pk, pv, D4, gs, D4, ctypes.c_void_p(ee.data_ptr() + 4 * ctx.col),
                  ld_ee, _p(alpha), _p(plan.rowptr), _p(plan.src), N, H, D // H, ctx.p_drop, ctx.seed, _p(ctx.seed_dev), gq,
                  ctypes.c_void_p(g_ee.data_ptr() + 4 * ctx.col), _p(g_kv), ctypes.c_void_p(g_kv.data_ptr() + 4 * D), 2 * D,
                  _stream())
        # one by-source segment sum over the 2D columns lands in the adjacent k | v blocks of g_qkvs
        _lib.call("msde_segment_sum_rows", _p(g_kv), 0, _p(plan.rowptr_s), _p(plan.perm_s), N, 2 * D, 0.0, gk, D4, _stream())
        return g_qkvs, (g_ee if first else None), None, None, None, None, None, None, None


def edge_attention_fused(qkvs, ee, plan, heads, p_drop=0.0, seed=0, seed_dev=None, col=0, shared=None):
    """attention(q, k, v, ee[:, col:col+D]) + skip with q, k, v, skip = column blocks of qkvs [N, 4D]."""
    return _EdgeAttentionFused.apply(qkvs, ee, plan, heads, p_drop, seed, seed_dev, col, shared)


def edge_attention(q, k, v, ee, plan, heads, p_drop=0.0, seed=0, seed_dev=None):
    """TransformerConv message + per-target softmax + aggregate (App. A.4).  seed_dev: optional
    device uint64 added to the seed inside the kernel (fresh masks under hipGraph replay)."""
    return _EdgeAttention.apply(q, k, v, ee, plan, heads, p_drop, seed, seed_dev)


class _FrameMixMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, coff, basis, plan, base=None):
        coff = _f32(coff)
        base = _f32(base) if base is not None else None
        out = torch.empty(plan.N, 3, dtype=torch.float32, device=coff.device)
        _lib.call("msde_frame_mix_mean_add_fwd", _p(coff), _p(basis), _p(plan.rowptr), plan.N, _p(base), _p(out), _stream())
        ctx.save_for_backward(basis)
        ctx.plan = plan
        return out

    @staticmethod
    def backward(ctx, g):
        (basis,) = ctx.saved_tensors
        plan = ctx.plan
        g = _f32(g)
        g_coff = torch.empty(plan.E, 3, dtype=torch.float32, device=g.device)
        _lib.call("msde_frame_mix_mean_bwd", _p(g), _p(basis), _p(plan.rowptr), plan.N, plan.E, _p(g_coff), _stream())
        return g_coff, None, None, (g if ctx.needs_input_grad[3] else None)


def frame_mix_mean(coff, basis, plan, base=None):
    """base + mean_{e -> i} (c0 b_diff + c1 b_cross + c2 b_vert)  (equivariant_scorenetwork.py:159-166; base: the sum
    over the earlier score layers, added by the kernel instead of a separate operator)."""
    return _FrameMixMean.apply(coff, basis, plan, base)


class _MlpHeadMix(torch.autograd.Function):
    """basis MLP of the 2D->3D score network and what consumes it, as one autograd node (equivariant_scorenetwork.py:142-166):
    Linear(in, H) -> SiLU -> Linear(H, 3) on every edge, the three outputs mixed with the edge's frame vectors, averaged over
    the in-edges of the target node and added to `base`.  Forward: the first Linear (pre-activation stored) + ONE kernel
    (msde_mlp_head_mix_fwd) instead of head kernel + frame-mix kernel; the per-edge coefficients are never stored.  Backward:
    msde_mlp_head_mix_bwd forms the head's gradient from the node gradient while it reads the pre-activations (instead of a
    frame-mix backward kernel writing it first), then the first Linear's input gradient; weight gradients queued."""

    @staticmethod
    def forward(ctx, x, basis, plan, base, offload, W1, b1, W2, b2):
        x = x if (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1) else _f32(x)
        E, H = x.size(0), W1.size(0)
        dev = x.device
        Z = torch.empty(E, H, dtype=torch.float32, device=dev)
        gemm_fwd(x, W1, Z, bias=b1)
        mix = torch.empty(E, 3, dtype=torch.float32, device=dev)
        out = torch.empty(plan.N, 3, dtype=torch.float32, device=dev)
        base = _f32(base) if base is not None else None
        _lib.call("msde_mlp_head_mix_fwd", _p(Z), _ld(Z), _p(W2), _p(b2), H, _p(basis), _p(plan.rowptr), plan.N, _p(base),
                  _p(mix), _p(out), _stream())
        ctx.save_for_backward(x, Z, basis, W1, W2)
        ctx.plan = plan
        ctx.has_b2 = b2 is not None
        ctx.deferrable = offload and all(t is None or t.is_leaf or getattr(t, "_msde_leaf_like", False) for t in (W1, b1, W2, b2))
        return out

    @staticmethod
    def backward(ctx, g):
        x, Z, basis, W1, W2 = ctx.saved_tensors
        plan = ctx.plan
        g = g if (g.is_cuda and g.dtype == torch.float32 and g.is_contiguous()) else _f32(g)
        E, H = Z.shape
        dev = g.device
        gz = torch.empty(E, H, dtype=torch.float32, device=dev)
        gall = torch.empty((3 * H + 3 + 3) & ~3, dtype=torch.float32, device=dev)      # [gW2 | gb2], whole float4
        nslab = int(_lib.load().msde_mlp_head_bwd_slabs(E, H))
        if _SLABS.active and ctx.deferrable:
            ws = _SLABS.alloc(nslab * gall.numel(), dev)
            _lib.call("msde_mlp_head_mix_bwd", _p(Z), _ld(Z), _p(W2), _p(g), _p(basis), _p(plan.dst), _p(plan.rowptr), E, H,
                      _p(gz), _p(ws), _p(bound_tensor(E)), _stream())
            _SLABS.add(ws.data_ptr(), nslab, gall.numel(), gall, written=True)
        else:       # (no batch open: the two-kernel form, whose head kernel reduces its own slabs)
            g_coff = torch.empty(E, 3, dtype=torch.float32, device=dev)
            _lib.call("msde_frame_mix_mean_bwd", _p(g), _p(basis), _p(plan.rowptr), plan.N, E, _p(g_coff), _stream())
            ws = torch.empty(nslab * gall.numel(), dtype=torch.float32, device=dev)
            _lib.call("msde_mlp_head_bwd", _p(Z), _ld(Z), _p(W2), _p(g_coff), E, H, 3, _p(gz), _p(gall), _p(ws),
                      _p(bound_tensor(E)), _stream())
        gW2 = gall[:3 * H].view(3, H)
        gb2 = gall[3 * H:3 * H + 3] if ctx.has_b2 else None
        gW1 = gb1 = None
        if ctx.needs_input_grad[5] or ctx.needs_input_grad[6]:
            gW1, gb1 = weight_grad(gz, x, True, ctx.deferrable)
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty(E, W1.size(1), dtype=torch.float32, device=dev)
            gemm_dgrad(gz, W1, gx)
        return gx, None, None, (g if ctx.needs_input_grad[3] else None), None, gW1, gb1, gW2, gb2


def mlp_head_mix_ok(x, lin1, lin2):
    H = lin1.weight.size(0)
    return (x.is_cuda and x.dim() == 2 and lin2.weight.size(0) == 3 and lin2.weight.size(1) == H and H % 4 == 0 and H <= 256
            and lin1.bias is not None and lin2.weight.is_contiguous() and lin2.weight.data_ptr() % 16 == 0)


def mlp_head_mix(x, lin1, lin2, basis, plan, base=None, offload=True):
    """base + mean over in-edges of (lin2(silu(lin1(x))) mixed with the frame vectors): _MlpHeadMix."""
    return _MlpHeadMix.apply(x, basis, plan, base, offload, lin1.weight, lin1.bias, lin2.weight, lin2.bias)


class _SegmentMean(torch.autograd.Function):
    """Per-molecule mean/sum of node rows (torch_scatter.scatter over the sorted `batch` vector)."""

    @staticmethod
    def forward(ctx, x, mol_ptr, batch_i32, mean):
        x = _f32(x)
        B = mol_ptr.numel() - 1
        out = segment_sum_rows(x, mol_ptr, None, B, mean=mean)
        ctx.save_for_backward(mol_ptr, batch_i32)
        ctx.mean = mean
        return out

    @staticmethod
    def backward(ctx, g):
        mol_ptr, batch_i32 = ctx.saved_tensors
        g = _f32(g)
        if ctx.mean:
            cnt = (mol_ptr[1:] - mol_ptr[:-1]).clamp(min=1).to(torch.float32)
            g = g / cnt.unsqueeze(1)
        return gather_rows(g, batch_i32), None, None, None


class _GatherRowsGrad(torch.autograd.Function):
    """out[s] = x[idx[s]] (idx < 0 -> zero row) where every row of x appears at most once: the backward is
    the inverse gather g_x[i] = g_out[inv[i]] (ragged -> padded packing, PyG to_dense_batch)."""

    @staticmethod
    def forward(ctx, x, idx, inv):
        ctx.save_for_backward(inv)
        return gather_rows(x, idx)

    @staticmethod
    def backward(ctx, g):
        (inv,) = ctx.saved_tensors
        return gather_rows(_f32(g), inv), None, None


def gather_rows_grad(x, idx, inv):
    return _GatherRowsGrad.apply(x, idx, inv)


def segment_reduce(x, mol_ptr, batch_i32, mean=True):
    return _SegmentMean.apply(x, mol_ptr, batch_i32, mean)


# ------------------------------------------------------------------------------------------------
# dense layers
# ------------------------------------------------------------------------------------------------
# (capture lifetime, workspaces, batched weight gradients / slab reduction: moleculesde_amd/slabs.py; what this module uses of it)
from .slabs import (_retire, EMB_BWD_SPLIT, _scratch, _ws_key, _SlabBatch, _SLABS, run_deferred_leaf_kernels, _row_stride,  # noqa: E402
                    weight_grad, weight_grad_blocks)




# Dispatch policy of the dense layers, from per-shape rocprofv3 timings on MI355X (profiles/):
#   * forward / input gradient: the row-strip MFMA kernels of csrc/gemm_rs.hip (round 3; rounds 1-2 used the vendor
#     fp32 GEMM here, which was faster than csrc/linear.hip and the 64 x 64-tile csrc/gemm_ex.hip on these skinny shapes);
#   * weight + bias gradient: the vendor kernel runs ~100 output tiles over the whole M loop (31-39 us at
#     M = 3588 whatever the layer size, 110-225 us at edge level); the split-M MFMA kernel + fixed-order
#     slab reduce takes 9-33 us (tools/bench_wgrad.py, hipGraph-timed), fuses the bias gradient and is
#     bitwise reproducible -> hand-written kernel for every layer.
# The vendor GEMM is not reachable from the product (tools/bench_gemm*.py time it beside the kernels).
import os as _os

_LINEAR_MODE = "auto"            # set_linear_mode("hip"): the round-1 kernels of csrc/linear.hip everywhere (cross-check tests)


def set_linear_mode(mode):
    global _LINEAR_MODE
    assert mode in ("auto", "hip")
    _LINEAR_MODE = mode




class _PairLinear(torch.autograd.Function):
    """AB [M, 2D] = [h W[:, :D]^T | h W[:, D:]^T + b] for the Linear(2D, D) that the reference applies to cat(h[row], h[col])
    of every edge (SDE_model_2D_to_3D.py:386-388: edge_2D_emb[0]): the product is linear in the two halves, so it is
    formed ONCE PER NODE and the edge-level sum is a gather-add.  The weight is read through a cached stacked copy
    (hip.weight_layout, refreshed with the transposed weights once per optimiser step: no per-step re-layout, no cat);
    backward: two input-gradient products on the weight's column halves as stored, and two queued weight-gradient
    problems whose results the batched reduction writes into the column halves of the [D, 2D] gradient."""

    @staticmethod
    def forward(ctx, h, W, b):
        h = _f32(h)
        M, D = h.shape
        B = weight_layout("pair_linear", (W,), (D, 2 * D), [(W, 0, 0, D, D, 0, 0, True), (W, 0, D, D, D, 0, D, True)])
        bst = weight_layout("pair_bias", (b,), (2 * D,), [(b, 0, 0, 1, D, 0, D, False)])
        AB = torch.empty(M, 2 * D, dtype=torch.float32, device=h.device)
        gemm_rs(h, B, AB, bias=bst, b_kmajor=True, N=2 * D, K=D, fallback=False)
        ctx.save_for_backward(h, W)
        ctx.deferrable = W.is_leaf and b.is_leaf
        return AB

    @staticmethod
    def backward(ctx, g):
        h, W = ctx.saved_tensors
        g = _f32(g)
        M, D = h.shape
        g1, g2 = g[:, :D], g[:, D:]
        gh = None
        if ctx.needs_input_grad[0]:
            gh = torch.empty(M, D, dtype=torch.float32, device=g.device)
            gemm_rs(g1, W, gh, b_kmajor=True, N=D, K=D, fallback=False)
            gemm_rs(g2, W[:, D:], gh, b_kmajor=True, N=D, K=D, accumulate=True, fallback=False)
        gW = torch.empty(D, 2 * D, dtype=torch.float32, device=g.device)
        weight_grad_blocks(g1, h, False, [(0, D, gW, 0)], ctx.deferrable)
        gb = weight_grad_blocks(g2, h, True, [(0, D, gW, D)], ctx.deferrable)
        return gh, gW, gb


def pair_linear_ok(h, lin):
    D = lin.weight.size(0)
    return (h.is_cuda and h.dim() == 2 and lin.weight.size(1) == 2 * D and D % 4 == 0 and 0 < h.size(0) <= RS_MAX_ROWS
            and lin.weight.is_leaf and lin.bias is not None and lin.bias.is_leaf)


def pair_linear(h, lin):
    return _PairLinear.apply(h, lin.weight, lin.bias)


class _FrameMLP(torch.autograd.Function):
    """project(cat([angle, coff_mlp(feat_i), coff_mlp(feat_j)])) of SDE_model_2D_to_3D.py:366-370 without cat, without the
    second application of the shared coff_mlp and without immediate weight-gradient launches:
      * feat [2E, 4H] holds feat_i(e), feat_j(e) in rows 2e, 2e + 1 (msde_edge_geometry_fwd_ld writes them there): coff_mlp
        is ONE product over 2E rows; its result goes to columns 0..H of X [2E, H + 4], whose columns H..H+4 the geometry
        kernel has filled with (sin, cos, 0, 0) of the pseudo-angle (rows 2e) and zeros (rows 2e + 1);
      * X seen as [E, 2H + 8] is then exactly project[0]'s input in another column order: one msde_gemm_ex launch against
        a cached permuted, zero-padded copy of its weight ([W_i | W_angle 0 0 | W_j | 0 0 0 0], hip.weight_layout), SiLU
        in the epilogue;
      * backward: SiLU' in the epilogue of project[1]'s input-gradient product; ONE product gives the gradient of both
        coff_mlp applications ([W_i | W_j] cached); the shared coff_mlp gets ONE queued weight-gradient problem over 2E
        rows, project[0] one whose column blocks the batched reduction writes into the [H, 2H + 2] gradient.
    feat / angle carry no gradient.  Valid rows of the 2E-row tensors: 2 x (valid extended edges), a
    row bound of its own in a capacity bucket."""

    @staticmethod
    def forward(ctx, feat, X, Wc, bc, W1, b1, W2, b2):
        E2, H = feat.size(0), Wc.size(0)
        E = E2 // 2
        dev = feat.device
        gemm_ex(feat, Wc, X[:, :H], bias=bc)
        W1p = weight_layout("frame_project", (W1,), (H, 2 * H + 8),
                            [(W1, 0, 2, H, H, 0, 0, False), (W1, 0, 0, H, 2, 0, H, False), (W1, 0, H + 2, H, H, 0, H + 4, False)])
        Z1 = torch.empty(E, H, dtype=torch.float32, device=dev)
        h1 = torch.empty(E, H, dtype=torch.float32, device=dev)
        gemm_ex(X.view(E, 2 * H + 8), W1p, h1, bias=b1, act="silu", Z=Z1)
        out = torch.empty(E, W2.size(0), dtype=torch.float32, device=dev)
        gemm_ex(h1, W2, out, bias=b2)
        ctx.save_for_backward(feat, X, Z1, h1, Wc, W1, W2)
        ctx.deferrable = all(t.is_leaf for t in (Wc, bc, W1, b1, W2, b2))
        return out

    @staticmethod
    def backward(ctx, g):
        feat, X, Z1, h1, Wc, W1, W2 = ctx.saved_tensors
        g = _f32(g)
        E, H = Z1.shape
        dev = g.device
        gZ1 = torch.empty(E, H, dtype=torch.float32, device=dev)
        gemm_ex(g, W2, gZ1, b_kmajor=True, act="silu", dact_from=Z1)
        W1q = weight_layout("frame_project_ij", (W1,), (H, 2 * H), [(W1, 0, 2, H, 2 * H, 0, 0, False)])     # [W_i | W_j]
        g_embed = torch.empty(2 * E, H, dtype=torch.float32, device=dev)
        gemm_ex(gZ1, W1q, g_embed.view(E, 2 * H), b_kmajor=True)
        gW2, gb2 = weight_grad(g, h1, True, ctx.deferrable)
        gW1 = torch.empty(H, 2 * H + 2, dtype=torch.float32, device=dev)
        gb1 = weight_grad_blocks(gZ1, X.view(E, 2 * H + 8), True,
                                 [(0, H, gW1, 2), (H, 2, gW1, 0), (H + 4, H, gW1, H + 2)], ctx.deferrable)
        gWc, gbc = weight_grad(g_embed, feat, True, ctx.deferrable)
        return None, None, gWc, gbc, gW1, gb1, gW2, gb2


def frame_mlp(feat, X, coff, l1, l2):
    return _FrameMLP.apply(feat, X, coff.weight, coff.bias, l1.weight, l1.bias, l2.weight, l2.bias)


def edge_geometry_stacked(pos, plan, Wd, Wc, H):
    """hip.edge_geometry with the frame features laid out for hip.frame_mlp: returns (feat_d [E, 2C], feat [2E, 4C] with
    feat_i(e) / feat_j(e) in rows 2e / 2e + 1, X [2E, H + 4] with (sin, cos, 0, 0) of the pseudo-angle in columns H.. of
    rows 2e and zeros in those of rows 2e + 1, basis [E, 9])."""
    pos = _f32(pos.detach())
    E, C = plan.E, Wd.numel()
    dev = pos.device
    feat_d = torch.empty(E, 2 * C, dtype=torch.float32, device=dev)
    feat = torch.empty(2 * E, 4 * C, dtype=torch.float32, device=dev)
    X = torch.empty(2 * E, H + 4, dtype=torch.float32, device=dev)
    basis = torch.empty(E, 9, dtype=torch.float32, device=dev)
    _lib.call("msde_edge_geometry_fwd_ld", _p(pos), _p(plan.src), _p(plan.dst), E, _p(_f32(Wd.detach())),
              _p(_f32(Wc.detach())), C, _p(feat_d), _p(feat), ctypes.c_void_p(feat.data_ptr() + 16 * C), 8 * C,
              ctypes.c_void_p(X.data_ptr() + 4 * H), 2 * (H + 4), H + 4, _p(basis), _stream())
    return feat_d, feat, X, basis


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, offload=True):
        shape = x.shape
        x2 = _f32(x.reshape(-1, shape[-1]))
        w = _f32(weight)
        M, K = x2.shape
        N = w.size(0)
        if _LINEAR_MODE == "hip":
            y = torch.empty(M, N, dtype=torch.float32, device=x2.device)
            _lib.call("msde_linear_fwd", _p(x2), _p(w), _p(_f32(bias) if bias is not None else None), M, N, K, _p(y),
                      _stream())
        else:
            y = torch.empty(M, N, dtype=torch.float32, device=x2.device)
            if M > 0:
                gemm_fwd(x2, weight, y, bias=_f32(bias) if bias is not None else None)
        ctx.save_for_backward(x2, w)
        ctx.has_bias = bias is not None
        ctx.in_shape = shape
        # deferred slab reduction (see _SlabBatch) only for gradients nothing reads before the optimiser
        ctx.deferrable = offload and all(t is None or t.is_leaf or getattr(t, "_msde_leaf_like", False)
                                         for t in (weight, bias))
        return y.view(*shape[:-1], N)

    @staticmethod
    def backward(ctx, g, g_res=None):
        x2, w = ctx.saved_tensors
        M, K = x2.shape
        N = w.size(0)
        g2 = _f32(g.reshape(-1, N))
        st = _stream()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            if _LINEAR_MODE == "hip":
                gx = torch.empty(M, K, dtype=torch.float32, device=g2.device)
                _lib.call("msde_linear_bwd_x", _p(g2), _p(w), M, N, K, _p(gx), st)
                if g_res is not None:
                    gx = gx + g_res.reshape(M, K)
            else:
                gx = torch.empty(M, K, dtype=torch.float32, device=g2.device)
                if M > 0:
                    gemm_dgrad(g2, w, gx, res=_f32(g_res.reshape(M, K)) if g_res is not None else None)
            gx = gx.view(ctx.in_shape)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            gw, gb = weight_grad(g2, x2, ctx.has_bias, ctx.deferrable)
        return gx, gw, gb, None


class _LinearFork(torch.autograd.Function):
    """(x, Linear(x)): the input comes back as a second output for a residual / second consumer, so that in the
    backward the gradient arriving on that branch is ACCUMULATED BY THE INPUT-GRADIENT GEMM (C = g_res + g W)
    instead of by a separate element-wise add launched by autograd."""

    @staticmethod
    def forward(ctx, x, weight, bias, offload=True):
        y = _Linear.forward(ctx, x, weight, bias, offload)
        # an alias of x WITHOUT autograd's view relation to it (a tracked view of an input returned from a custom
        # Function makes the engine copy its gradient); as an output of this node it is differentiable all the same
        return x.detach(), y

    @staticmethod
    def backward(ctx, g_res, g):
        if g is None:
            return g_res, None, None, None
        gx, gw, gb, _ = _Linear.backward(ctx, g, g_res)
        return gx, gw, gb, None


def linear_fork(x, weight, bias=None, offload=True):
    """Returns (x_again, F.linear(x, weight, bias)); use x_again for the residual branch."""
    return _LinearFork.apply(x, weight, bias, offload)


def linear(x, weight, bias=None, offload=True):
    """F.linear with the dispatch policy above (library GEMM / csrc/linear.hip MFMA kernels).  offload=False:
    the parameters are used more than once per forward, so autograd adds their gradients and the weight-gradient
    slabs are reduced on the spot instead of in the batched reduction."""
    return _Linear.apply(x, weight, bias, offload)


# ------------------------------------------------------------------------------------------------
# contrastive loss
# ------------------------------------------------------------------------------------------------
class _ContrastiveEBM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, Y, perm1, perm2, T):
        X, Y = _f32(X), _f32(Y)
        N, D = X.shape
        p1, p2 = _i32(perm1), _i32(perm2)
        rows = torch.empty(N, 3, dtype=torch.float32, device=X.device)
        inv1 = torch.empty(N, dtype=torch.int32, device=X.device)
        inv2 = torch.empty(N, dtype=torch.int32, device=X.device)
        out = torch.empty(2, dtype=torch.float32, device=X.device)
        _lib.call("msde_cl_ebm_fwd", _p(X), _p(Y), _p(p1), _p(p2), N, D, 1.0 / float(T), _p(rows), _p(inv1), _p(inv2),
                  _p(out), _p(bound_tensor(N)), _stream())
        ctx.save_for_backward(X, Y, p1, p2, inv1, inv2, rows)
        ctx.invT = 1.0 / float(T)
        loss, acc = out[0], out[1]         # two outputs (a select on ONE output costs a zeros + copy in its backward)
        ctx.mark_non_differentiable(acc)
        ctx.set_materialize_grads(False)   # no zero-filled gradient (one fill launch) for the accuracy output
        return loss, acc

    @staticmethod
    def backward(ctx, g_loss, g_acc):
        if g_loss is None:
            return None, None, None, None, None
        X, Y, p1, p2, inv1, inv2, rows = ctx.saved_tensors
        N, D = X.shape
        gX, gY = torch.empty_like(X), torch.empty_like(Y)
        g = _f32(g_loss).reshape(1)        # d/d(loss); the accuracy carries no gradient
        _lib.call("msde_cl_ebm_bwd", _p(X), _p(Y), _p(p1), _p(p2), _p(inv1), _p(inv2), _p(rows), _p(g), N, D, ctx.invT,
                  _p(gX), _p(gY), _p(bound_tensor(N)), _stream())
        return gX, gY, None, None, None


def contrastive_ebm(X, Y, perm1, perm2, T):
    """dual_CL with 'EBM_node_dot_prod' (examples/util.py:52-68,76-79): returns (loss, accuracy)."""
    loss, acc = _ContrastiveEBM.apply(X, Y, perm1, perm2, T)
    return loss, acc


# ------------------------------------------------------------------------------------------------
# normalisation
# ------------------------------------------------------------------------------------------------
_BN_WS = {}


def _bn_workspace(M, C, device):
    n = int(_lib.load().msde_bn_workspace_floats(M, C))
    device = _ws_key(device)
    ws = _BN_WS.get(device)
    if ws is None or ws.numel() < n:
        _retire([ws])
        ws = torch.empty(max(n, 1 << 16), dtype=torch.float32, device=device[0])
        _BN_WS[device] = ws
    return ws


class _BatchNormTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, momentum, relu):
        x = _f32(x)
        M, C = x.shape
        y = torch.empty_like(x)
        mean = torch.empty(C, dtype=torch.float32, device=x.device)
        rstd = torch.empty(C, dtype=torch.float32, device=x.device)
        ws = _bn_workspace(M, C, x.device)
        _lib.call("msde_bn_fwd", _p(x), M, C, _p(gamma), _p(beta), float(eps), float(momentum), _p(running_mean),
                  _p(running_var), int(relu), _p(y), _p(mean), _p(rstd), _p(ws), _p(bound_tensor(M)), _stream())
        ctx.save_for_backward(x, gamma, beta, mean, rstd)
        ctx.relu = int(relu)
        return y

    @staticmethod
    def backward(ctx, g):
        x, gamma, beta, mean, rstd = ctx.saved_tensors
        g = _f32(g)
        M, C = x.shape
        dx = torch.empty_like(x)
        dgamma = torch.empty(C, dtype=torch.float32, device=x.device) if gamma is not None else None
        dbeta = torch.empty(C, dtype=torch.float32, device=x.device) if beta is not None else None
        ws = _bn_workspace(M, C, x.device)
        _lib.call("msde_bn_bwd", _p(g), _p(x), _p(mean), _p(rstd), _p(gamma), _p(beta), ctx.relu, M, C, _p(dx), _p(dgamma),
                  _p(dbeta), _p(ws), _p(bound_tensor(M)), _stream())
        return dx, dgamma, dbeta, None, None, None, None, None


def batch_norm_train(x, gamma, beta, running_mean, running_var, eps, momentum, relu=False):
    """Training-mode BatchNorm1d over rows (+ optional fused ReLU); updates the running buffers in place."""
    return _BatchNormTrain.apply(x, gamma, beta, running_mean, running_var, eps, momentum, relu)


class _ResLayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, res, gamma, beta, eps):
        x = _f32(x)
        N, D = x.shape
        y = torch.empty_like(x)
        mean = torch.empty(N, dtype=torch.float32, device=x.device)
        rstd = torch.empty(N, dtype=torch.float32, device=x.device)
        _lib.call("msde_res_layernorm_fwd", _p(x), _p(_f32(res) if res is not None else None), _p(gamma), _p(beta), N, D,
                  float(eps), _p(y), _p(mean), _p(rstd), _stream())
        ctx.save_for_backward(x, gamma, mean, rstd)
        ctx.has_res = res is not None
        return y

    @staticmethod
    def backward(ctx, g):
        x, gamma, mean, rstd = ctx.saved_tensors
        g = _f32(g)
        N, D = x.shape
        gx = torch.empty_like(x)
        gab = torch.empty(2 * D, dtype=torch.float32, device=x.device)
        ws = _bn_workspace(64 * 2, D, x.device)           # >= 64 * 2 * D floats
        _lib.call("msde_res_layernorm_bwd", _p(g), _p(x), _p(gamma), _p(mean), _p(rstd), N, D, _p(gx), _p(gab),
                  ctypes.c_void_p(gab.data_ptr() + 4 * D), _p(ws), _stream())
        return gx, (g if ctx.has_res else None), gab[:D], gab[D:], None


def res_layernorm(x, res, gamma, beta, eps=1e-5):
    """res + LayerNorm(x) in one kernel (backward: one kernel + a column-sum finaliser)."""
    return _ResLayerNorm.apply(x, res, gamma, beta, eps)


# ------------------------------------------------------------------------------------------------
# optimiser
# ------------------------------------------------------------------------------------------------
# ------------------------------------------------------------------------------------------------
# pointwise stages (csrc/pointwise.hip)
# ------------------------------------------------------------------------------------------------
class _ShiftedSoftplus(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _f32(x)
        y = torch.empty_like(x)
        _lib.call("msde_ssp_fwd", _p(x), x.numel(), _p(y), _stream())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        g = _f32(g)
        gx = torch.empty_like(x)
        _lib.call("msde_ssp_bwd", _p(g), _p(x), x.numel(), _p(gx), _stream())
        return gx


def shifted_softplus(x):
    """F.softplus(x) - log 2 (schnet.py:199-206) as one kernel."""
    return _ShiftedSoftplus.apply(x)


class _SiluDropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed, seed_dev):
        x = _f32(x)
        y = torch.empty_like(x)
        _lib.call("msde_silu_dropout_fwd", _p(x), x.numel(), float(p), int(seed) & 0xFFFFFFFFFFFFFFFF, _p(seed_dev), _p(y),
                  _stream())
        ctx.save_for_backward(x)
        ctx.p, ctx.seed, ctx.seed_dev = float(p), int(seed) & 0xFFFFFFFFFFFFFFFF, seed_dev
        return y

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        g = _f32(g)
        gx = torch.empty_like(x)
        _lib.call("msde_silu_dropout_bwd", _p(g), _p(x), x.numel(), ctx.p, ctx.seed, _p(ctx.seed_dev), _p(gx), _stream())
        return gx, None, None, None


def silu_dropout(x, p=0.0, seed=0, seed_dev=None):
    """nn.SiLU followed by nn.Dropout(p) (p = 0: plain SiLU) as one kernel; the mask is a function of
    (seed [+ device counter], element index) and is regenerated in the backward."""
    return _SiluDropout.apply(x, p, seed, seed_dev)


class _MulAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, c):
        a, b, c = _f32(a), _f32(b), _f32(c)
        assert a.shape == b.shape == c.shape
        out = torch.empty_like(a)
        _lib.call("msde_mul_add_fwd", _p(a), _p(b), _p(c), a.numel(), _p(out), _stream())
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = _f32(g)
        ga = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        gb = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        _lib.call("msde_mul_add_bwd", _p(g), _p(a), _p(b), a.numel(), _p(ga), _p(gb), _stream())
        return ga, gb, (g if ctx.needs_input_grad[2] else None)


def mul_add(a, b, c):
    """a * b + c for same-shape tensors, one kernel each way."""
    return _MulAdd.apply(a, b, c)


RANDPERM_MAX = 8192


def randperm(n, device, seed, seed_dev=None, count=None):
    """Uniform random permutation(s) of range(n) as int32 (n <= RANDPERM_MAX), one kernel: [n], or [count, n]
    independent ones when `count` is given."""
    out = torch.empty((n,) if count is None else (count, n), dtype=torch.int32, device=device)
    _lib.call("msde_randperm", int(n), 1 if count is None else int(count), int(seed) & 0xFFFFFFFFFFFFFFFF, _p(seed_dev),
              _p(out), _p(bound_tensor(n)), _stream())
    return out


def ve_perturb(pos, noise, draws, batch_i32, B, T, eps, sigma_min, sigma_max):
    """pos + std(t_mol) * noise and std per atom for the VE SDE, from the integer time-step draws (one kernel for
    the ~9 pointwise operators of SDE_model_2D_to_3D.py:401-412).  No gradient (coordinates carry none)."""
    pos, noise = _f32(pos.detach()), _f32(noise)
    N = pos.size(0)
    out = torch.empty_like(pos)
    std = torch.empty(N, dtype=torch.float32, device=pos.device)
    _lib.call("msde_ve_perturb", _p(pos), _p(noise), _p(draws.contiguous()), _p(batch_i32), N, int(B), int(T), float(eps),
              float(sigma_min), float(sigma_max), _p(out), _p(std), _stream())
    return out, std


def ve_perturb_rng(pos, batch_i32, B, T, eps, sigma_min, sigma_max, seed, seed_dev=None):
    """hip.ve_perturb with the draws made in the kernel (counter-based generator; seed_dev: device step counter mixed in
    under hipGraph replay).  Returns (noise [N,3], pos + std * noise, std per atom)."""
    pos = _f32(pos.detach())
    N = pos.size(0)
    noise, out = torch.empty_like(pos), torch.empty_like(pos)
    std = torch.empty(N, dtype=torch.float32, device=pos.device)
    _lib.call("msde_ve_perturb_rng", _p(pos), _p(batch_i32), N, int(B), int(T), float(eps), float(sigma_min), float(sigma_max),
              int(seed) & 0xFFFFFFFFFFFFFFFF, _p(seed_dev), _p(noise), _p(out), _p(std), _stream())
    return noise, out, std


class _VEPosLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scores, noise, std, anneal_power, mol_ptr, batch_i32):
        scores, noise = _f32(scores), _f32(noise)
        std = _f32(std) if std is not None else None
        N, B = scores.size(0), mol_ptr.numel() - 1
        ws = torch.empty(B, dtype=torch.float32, device=scores.device)
        loss = torch.empty(1, dtype=torch.float32, device=scores.device)
        _lib.call("msde_ve_pos_loss_fwd", _p(scores), _p(noise), _p(std), float(anneal_power), _p(mol_ptr), N, B, _p(ws),
                  _p(loss), _stream())
        ctx.save_for_backward(scores, noise, std if std is not None else scores, mol_ptr, batch_i32)
        ctx.power, ctx.has_std = float(anneal_power), std is not None
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        scores, noise, std, mol_ptr, batch_i32 = ctx.saved_tensors
        N, B = scores.size(0), mol_ptr.numel() - 1
        g = _f32(g).reshape(1)
        gs = torch.empty_like(scores)
        _lib.call("msde_ve_pos_loss_bwd", _p(scores), _p(noise), _p(std if ctx.has_std else None), ctx.power, _p(mol_ptr),
                  _p(batch_i32), N, B, _p(g), _p(gs), _stream())
        return gs, None, None, None, None, None


_UNIT_GRADS = {}          # data_ptr -> weakref of a LIVE tensor that holds the constant 1.0 (pretrain.Trainer._one_grad, the
                          # d(loss)/d(loss) it passes to backward()): lets the loss composition skip its backward launch


def register_unit_grad(t):
    """t: a device scalar that holds 1.0 for as long as it lives and is passed to loss.backward(t)."""
    _UNIT_GRADS[t.data_ptr()] = _weakref.ref(t)


def _is_unit_grad(g):
    r = _UNIT_GRADS.get(g.data_ptr())
    if r is None:
        return False
    if r() is None:                      # the registered tensor died: its address may belong to anything now
        _UNIT_GRADS.pop(g.data_ptr(), None)
        return False
    return True


class _CombineLosses(torch.autograd.Function):
    """sum_i c_i * l_i over device scalars: one launch forward (csrc/pointwise.hip), which also leaves the backward's result
    for a unit upstream gradient and adds the logged terms to their running sums; the backward launches only when the
    upstream gradient is something else than the trainer's constant one."""

    @staticmethod
    def forward(ctx, coeffs, logs, *terms):
        assert 1 <= len(terms) <= 4 and len(coeffs) == len(terms)
        ts = [_f32(t).reshape(1) for t in terms]
        c = [float(x) for x in coeffs] + [0.0] * (4 - len(terms))
        dev = ts[0].device
        out = torch.empty(1, dtype=torch.float32, device=dev)
        seeds = torch.empty(4, dtype=torch.float32, device=dev)
        ptrs = [_p(t) for t in ts] + [_p(None)] * (4 - len(ts))
        logs = list(logs or [])[:5]
        n = len(logs)
        src = (ctypes.c_void_p * 5)(*([t.data_ptr() for t, _ in logs] + [None] * (5 - n)))
        dst = (ctypes.c_void_p * 5)(*([d.data_ptr() for _, d in logs] + [None] * (5 - n)))
        _lib.call("msde_combine_losses_ex", *ptrs, *c, _p(out), _p(seeds), ctypes.cast(src, ctypes.c_void_p),
                  ctypes.cast(dst, ctypes.c_void_p), n, _stream())
        ctx.c, ctx.n = c, len(terms)
        ctx.seeds = seeds
        return out[0]

    @staticmethod
    def backward(ctx, g):
        if _is_unit_grad(g):
            out4 = ctx.seeds                 # g == 1: c_i * g was written by the forward launch
        else:
            g = _f32(g).reshape(1)
            out4 = torch.empty(4, dtype=torch.float32, device=g.device)
            _lib.call("msde_combine_losses_bwd", _p(g), *ctx.c, _p(out4), _stream())
        return (None, None) + tuple(out4[i] for i in range(ctx.n))


def combine_losses(coeffs, terms, logs=None):
    """logs: optional [(device scalar, running-sum device scalar)] added in the same launch (<= 5 pairs)."""
    return _CombineLosses.apply(tuple(coeffs), logs, *terms)


def ve_position_loss(scores, noise, std, anneal_power, mol_ptr, batch_i32):
    """mean over molecules of the per-molecule mean of sum_k (scores - noise)^2 [* std^anneal_power]
    (SDE_model_2D_to_3D.py:425-432); gradient flows to `scores` only."""
    return _VEPosLoss.apply(scores, noise, std, anneal_power, mol_ptr, batch_i32)


class _GatTail(torch.autograd.Function):
    """GATLayer after the attention (LayerNorm + residual, feed-forward with SiLU and dropout, LayerNorm +
    residual, optional SiLU) as one kernel each way -- csrc/gat_tail.hip."""

    @staticmethod
    def forward(ctx, x, res, g1, b1, W0, b0, W3, b3, g2, b2, eps1, eps2, p, seed, seed_dev, silu_out):
        x, res = _f32(x), _f32(res)
        N, D = x.shape
        out, y1, h0, x2 = (torch.empty_like(x) for _ in range(4))
        seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        _lib.call("msde_gat_tail_fwd", _p(x), _p(res), _p(g1), _p(b1), _p(W0), _p(b0), _p(W3), _p(b3), _p(g2), _p(b2), N, D,
                  float(eps1), float(eps2), float(p), seed, _p(seed_dev), int(silu_out), _p(out), _p(y1), _p(h0), _p(x2),
                  _stream())
        ctx.save_for_backward(x, y1, h0, x2, g1, W0, W3, g2, b2)
        ctx.cfg = (float(eps1), float(eps2), float(p), seed, seed_dev, int(silu_out))
        ctx.deferrable = all(t.is_leaf for t in (g1, b1, W0, b0, W3, b3, g2, b2))
        return out

    @staticmethod
    def backward(ctx, g):
        x, y1, h0, x2, g1, W0, W3, g2, b2 = ctx.saved_tensors
        eps1, eps2, p, seed, seed_dev, silu_out = ctx.cfg
        g = _f32(g)
        N, D = x.shape
        g_x, g_res, g_x2, a, g_h0 = (torch.empty_like(x) for _ in range(5))
        nblk = int(_lib.load().msde_gat_tail_blocks(N))
        defer = _SLABS.active and ctx.deferrable
        part = _SLABS.alloc(nblk * 4 * D, x.device) if defer else torch.empty(nblk * 4 * D, dtype=torch.float32, device=x.device)
        _lib.call("msde_gat_tail_bwd", _p(g), _p(x), _p(y1), _p(h0), _p(x2), _p(g1), _p(W0), _p(W3), _p(g2), _p(b2), N, D,
                  eps1, eps2, p, seed, _p(seed_dev), silu_out, _p(g_x), _p(g_res), _p(g_x2), _p(a), _p(g_h0), _p(part),
                  _p(bound_tensor(N)), _stream())
        ln = torch.empty(4 * D, dtype=torch.float32, device=x.device)      # [d ln2_g | d ln2_b | d ln1_g | d ln1_b]
        if defer:
            _SLABS.add(part.data_ptr(), nblk, 4 * D, ln)
        else:
            torch.sum(part.view(nblk, 4 * D), dim=0, out=ln)
        gW3, gb3 = weight_grad(g_x2, a, True, ctx.deferrable)
        gW0, gb0 = weight_grad(g_h0, y1, True, ctx.deferrable)
        return (g_x, g_res, ln[2 * D:3 * D], ln[3 * D:], gW0, gb0, gW3, gb3, ln[:D], ln[D:2 * D],
                None, None, None, None, None, None)


def gat_tail(x, res, norm1, ffn0, ffn3, norm2, p, seed, seed_dev=None, silu_out=False):
    """out = y1 + LN2(FFN(y1)), y1 = res + LN1(x), optionally followed by SiLU; modules give the parameters."""
    return _GatTail.apply(x, res, norm1.weight, norm1.bias, ffn0.weight, ffn0.bias, ffn3.weight, ffn3.bias, norm2.weight,
                          norm2.bias, norm1.eps, norm2.eps, p, seed, seed_dev, silu_out)


def chunk_elems():
    return int(_lib.load().msde_chunk_elems())


def gather_chunks(table, n_chunks, flat):
    _lib.call("msde_gather_chunks", _p(table), n_chunks, _p(flat), _stream())


def adam_chunks(p, table, n_chunks, m, v, step_dev, seg_end, seg_lr, beta1, beta2, eps, weight_decay, grad_scale=1.0):
    _lib.call("msde_adam_chunks", _p(p), _p(table), n_chunks, _p(m), _p(v), _p(step_dev), _p(seg_end), _p(seg_lr),
              int(seg_end.numel()), float(beta1), float(beta2), float(eps), float(weight_decay), float(grad_scale), _stream())


def adam_flat(p, g, m, v, step_dev, seg_end, seg_lr, beta1, beta2, eps, weight_decay, grad_scale=1.0):
    _lib.call("msde_adam_flat", _p(p), _p(g), _p(m), _p(v), p.numel(), _p(step_dev), _p(seg_end), _p(seg_lr),
              seg_lr.numel(), float(beta1), float(beta2), float(eps), float(weight_decay), float(grad_scale), _stream())


# ------------------------------------------------------------------------------------------------
# general fused GEMM (csrc/gemm_ex.hip)
# ------------------------------------------------------------------------------------------------
def _ld(t):
    """Row stride of a 2-D fp32 tensor whose rows are contiguous (a column block of a wider buffer is fine)."""
    assert t.dim() == 2 and t.dtype == torch.float32 and t.is_cuda and (t.size(1) == 1 or t.stride(1) == 1), \
        (t.shape, t.stride(), t.dtype)
    return t.stride(0) if t.size(0) > 1 else max(t.stride(0), t.size(1))


def gemm_ex(A, B, out, bias=None, A2=None, B2=None, act=None, act_cols=None, Z=None, dact_from=None, rowscale=None,
            b_kmajor=False, accumulate=False, alpha=1.0, groups=1, group_strides=None, N=None, K=None, K2=None, bias2=None,
            b_kblk=None, _debug_flags=0):
    """out[M,N] (+)= alpha * rowscale * epi(A . B(^T) + A2 . B2(^T) + bias): one launch of msde_gemm_ex, no autograd.
    A, A2, out, Z, dact_from: 2-D fp32 device tensors with unit column stride (views into wider buffers are fine; their
    row strides become the leading dimensions).  B: [N, K] (nn.Linear layout) or, with b_kmajor, [K, N].
    act: None|'tanh'|'silu'|'elu'|'ssp'|'relu' applied on act_cols=(lo, hi) (default: all columns); Z: also store the
    pre-activation; dact_from: instead of applying `act`, multiply by its derivative evaluated on this saved tensor
    (the forward OUTPUT for tanh/elu/relu, the PRE-ACTIVATION for silu/ssp).
    groups > 1: `group_strides` = dict(a=, b=, bias=, c=, r=) in floats; N / K give the per-group problem size (then
    B may be any tensor starting at group 0's weights)."""
    d = _lib.GemmDesc()
    M = A.size(0)
    d.M = M
    d.K1 = int(K if K is not None else A.size(1))
    if N is None:
        N = B.size(1) if b_kmajor else B.size(0)
    d.N = int(N)
    d.A, d.lda = A.data_ptr(), _ld(A)
    d.B = B.data_ptr()
    d.ldb = (B.stride(0) if B.dim() == 2 else (d.N if b_kmajor else d.K1))
    if A2 is not None:
        d.A2, d.lda2 = A2.data_ptr(), _ld(A2)
        d.K2 = int(K2 if K2 is not None else A2.size(1))
        d.B2, d.ldb2 = B2.data_ptr(), (B2.stride(0) if B2.dim() == 2 else (d.N if b_kmajor else d.K2))
    d.bias = bias.data_ptr() if bias is not None else None
    d.bias2 = bias2.data_ptr() if bias2 is not None else None
    if b_kblk is not None:          # (block length = power of two, stride between blocks in floats, row stride)
        d.b_kblk_log2, d.b_kblk_stride, d.ldb = int(b_kblk[0]).bit_length() - 1, int(b_kblk[1]), int(b_kblk[2])
    d.C, d.ldc = out.data_ptr(), _ld(out)
    if Z is not None:
        d.Z, d.ldz = Z.data_ptr(), _ld(Z)
    d.act = _lib.ACT[act]
    if act_cols is not None:
        d.act_lo, d.act_hi = int(act_cols[0]), int(act_cols[1])
    else:
        d.act_lo, d.act_hi = 0, d.N
    d.epi = _lib.EPI_ACT
    if dact_from is not None:
        d.epi = _lib.EPI_DACT
        d.R, d.ldr = dact_from.data_ptr(), _ld(dact_from)
    d.rowscale = rowscale.data_ptr() if rowscale is not None else None
    d.flags = (_lib.GEMM_B_KMAJOR if b_kmajor else 0) | (_lib.GEMM_ACCUMULATE if accumulate else 0) | int(_debug_flags)
    d.groups = int(groups)
    gs = group_strides or {}
    d.a_gs, d.b_gs, d.bias_gs = int(gs.get("a", 0)), int(gs.get("b", 0)), int(gs.get("bias", 0))
    d.c_gs, d.r_gs = int(gs.get("c", 0)), int(gs.get("r", 0))
    d.alpha = float(alpha)
    _lib.call("msde_gemm_ex", ctypes.byref(d), _stream())
    return out


# ------------------------------------------------------------------------------------------------
# row-strip GEMM family (csrc/gemm_rs.hip): the plain nn.Linear products and the BatchNorm fused around them
# ------------------------------------------------------------------------------------------------
# (re-laid-out weight copies: moleculesde_amd/wcache.py; what this module uses of it)
from .wcache import _WT, weight_epoch, weight_t, weight_layout  # noqa: E402


_T2_MODE = "1"     # "0" (tests, measurements): every node-level product on the row strips
_T2_OK = {}
# (Round 4's bf16x3 experiment -- both fp32 operands split into three bf16 terms, six products on v_mfma_f32_16x16x32_bf16 --
# is out of the tree: 2.63 vs 2.71 ms two-stream, and kernels co-resident with it were not reproducible; numbers and the
# bisection in DESIGN.md section 5.000.)


# (round 5: SchNet's 300 <-> 128 products, 0.12 of peak on the 2-D tiles inside the step, on the row strips instead: 2.555 vs 2.512 ms -- not kept)
def t2_ok(M, N, K, axf=None):
    """True when msde_gemm_t2 (csrc/gemm_t2.hip) takes this node-level product with the A transform `axf` (and the switch
    is on).  The statistics geometry of a fused chain must not depend on the transform: callers that write BatchNorm
    partials ask with the heaviest transform of their chain."""
    if _T2_MODE == "0":
        return False
    code = {None: _lib.RS_AXF_NONE, "affine": _lib.RS_AXF_AFFINE, "bnbwd": _lib.RS_AXF_BNBWD}[axf]
    key = (int(M), int(N), int(K), code)
    v = _T2_OK.get(key)
    if v is None:
        v = _T2_OK[key] = bool(_lib.load().msde_gemm_t2_supported(*key))
    return v


def rs_forward_ok(M, N, K, w):
    """Shapes msde_gemm_rs takes for a forward product on weight w [N][K] (the rest goes to msde_gemm_ex)."""
    return K % 4 == 0 and N % 4 == 0 and w.dim() == 2 and w.is_contiguous() and w.dtype == torch.float32


def bound_tensor(rows):
    """Device int32 scalar holding the VALID row count of tensors with `rows` rows (capacity buckets), or None."""
    return _BOUNDS.get(int(rows))


def rs_geometry(M, N, K):
    """(strips, rows per strip) of the statistics partials the node-level product kernel writes for this problem (the 2-D
    tiled kernel when it takes the shape, else the row strips)."""
    a, b = ctypes.c_int(0), ctypes.c_int(0)
    _lib.call("msde_gemm_t2_geometry" if t2_ok(M, N, K, "bnbwd") else "msde_gemm_rs_geometry", int(M), int(N), int(K),
              ctypes.byref(a), ctypes.byref(b))
    return a.value, b.value


def gemm_node(A, W, out, forward, N, K, **kw):
    """A node-level product against the nn.Linear weight W [out][in]: forward (A W^T, N = out, K = in) or input gradient
    (A W, N = in, K = out), with every fusion of gemm_rs (**kw), on whichever kernel takes the shape -- and the weight copy
    that kernel reads: 2-D tiles read the reduction index contiguous ([N][K]: W as stored forward, its transposed copy
    backward), row strips the other one."""
    M = A.size(0)
    # (one kernel family per chain: a product that writes or follows BatchNorm partials is asked with the heaviest transform,
    # so that rs_geometry and every product of the chain agree on the strips)
    fused = kw.get("stats") is not None or kw.get("axf") is not None
    if t2_ok(M, N, K, "bnbwd" if fused else None):
        return gemm_rs(A, W if forward else weight_t(W), out, N=N, K=K, t2=True, **kw)
    return gemm_rs(A, weight_t(W) if forward else W, out, b_kmajor=True, N=N, K=K, fallback=False, **kw)


def gemm_rs(A, B, out, bias=None, act=None, Z=None, dact_from=None, res=None, b_kmajor=False, accumulate=False,
            axf=None, xf=(), relu=False, A2=None, A_out=None, stats=None, stats_mode=None, stats_z=None,
            stats_mean=None, m_valid=None, N=None, K=None, rt=0, splits=0, fallback=True, t2=False):
    """out[M,N] = epilogue(xf(A) . B(^T) + bias) on msde_gemm_rs (see include/msde_hip.h: msde_rs_desc); no autograd.
    Returns `out`.  Shapes the row-strip kernels do not take (K % 4, unaligned operands) go to msde_gemm_ex when
    `fallback` and no fusion beyond bias / activation / derivative / accumulate is asked for; otherwise raises."""
    d = _lib.RsDesc()
    M = A.size(0)
    d.M, d.K = M, int(K if K is not None else A.size(1))
    if N is None:
        N = B.size(1) if b_kmajor else B.size(0)
    d.N = int(N)
    d.A, d.lda = A.data_ptr(), _ld(A)
    d.B, d.ldb = B.data_ptr(), (B.stride(0) if B.dim() == 2 else (d.N if b_kmajor else d.K))
    d.bias = bias.data_ptr() if bias is not None else None
    d.C, d.ldc = out.data_ptr(), _ld(out)
    if Z is not None:
        d.Z, d.ldz = Z.data_ptr(), _ld(Z)
    d.act = _lib.ACT[act]
    d.epi = _lib.EPI_ACT
    if dact_from is not None:
        d.epi, d.R, d.ldr = _lib.EPI_DACT, dact_from.data_ptr(), _ld(dact_from)
    if res is not None:
        d.Res, d.ldres = res.data_ptr(), _ld(res)
    d.flags = (_lib.GEMM_B_KMAJOR if b_kmajor else 0) | (_lib.GEMM_ACCUMULATE if accumulate else 0) | \
        (_lib.RS_AXF_RELU if relu else 0)
    d.axf = {None: _lib.RS_AXF_NONE, "affine": _lib.RS_AXF_AFFINE, "bnbwd": _lib.RS_AXF_BNBWD}[axf]
    for i, v in enumerate(xf):
        setattr(d, "xf%d" % i, v.data_ptr() if v is not None else None)
    if A2 is not None:
        d.A2, d.lda2 = A2.data_ptr(), _ld(A2)
    if A_out is not None:
        d.A_out, d.lda_out = A_out.data_ptr(), _ld(A_out)
    if stats is not None:
        d.stats = stats.data_ptr()
        d.stats_mode = {"bnfwd": _lib.RS_STATS_BNFWD, "bnbwd": _lib.RS_STATS_BNBWD}[stats_mode]
        if stats_z is not None:
            d.stats_z, d.ld_sz = stats_z.data_ptr(), _ld(stats_z)
        if stats_mean is not None:
            d.stats_mean = stats_mean.data_ptr()
    if m_valid is None:
        m_valid = bound_tensor(M)
    d.m_valid = m_valid.data_ptr() if m_valid is not None else None
    d.rt, d.splits = int(rt), int(splits)
    if t2:          # the 2-D tiled kernel (csrc/gemm_t2.hip): B is the [N][K] operand
        fam = STAMPS is not None and STAMPS.get("family")
        if fam:         # bench.py: device timestamps around EVERY launch of the family inside the captured step
            STAMPS["t2_shapes"].append((int(d.M), int(d.N), int(d.K), axf or "none"))
            stamp("t2_start", seq=True)
        _lib.check(_lib.load().msde_gemm_t2(ctypes.byref(d), _stream()), "msde_gemm_t2")
        if fam:
            stamp("t2_end", seq=True)
        return out
    code = _lib.load().msde_gemm_rs(ctypes.byref(d), _stream())
    if code == -2 and fallback and axf is None and res is None and stats is None and A_out is None:
        return gemm_ex(A, B, out, bias=bias, act=act, Z=Z, dact_from=dact_from, b_kmajor=b_kmajor, accumulate=accumulate,
                       N=N, K=K)
    _lib.check(code, "msde_gemm_rs")
    return out


# (edge-level Linear layers, M > RS_MAX_ROWS, on the 2-D tiles too: measured 2.71 vs 2.69 ms -- gemm_ex wins at 35 k x 64..128 -- removed)
RS_MAX_ROWS = 8192     # above this (edge-level operands) msde_gemm_ex's 64 x 64 tiles are faster than 16-row strips (tools/bench_gemm_rs.py)


def gemm_fwd(x, W, out, bias=None, act=None, Z=None, res=None):
    """out = act(x W^T + bias) (+ res) for an nn.Linear weight W [N][K]: the row-strip kernel on the transposed copy of
    W for node-level operands, msde_gemm_ex otherwise.  No autograd."""
    M, K = x.shape
    N = W.size(0)
    if 0 < M <= RS_MAX_ROWS and rs_forward_ok(M, N, K, W) and x.data_ptr() % 16 == 0 and _ld(x) % 4 == 0:
        if t2_ok(M, N, K):        # 2-D tiles read the weight k-contiguous: as stored
            return gemm_rs(x, W, out, bias=bias, act=act, Z=Z, res=res, N=N, K=K, t2=True)
        return gemm_rs(x, weight_t(W), out, bias=bias, act=act, Z=Z, res=res, b_kmajor=True, N=N, K=K, fallback=False)
    if res is not None:
        raise _lib.MsdeHipError("gemm_fwd: a residual needs the row-strip kernel (M <= %d, K %% 4 == N %% 4 == 0)" % RS_MAX_ROWS)
    return gemm_ex(x, W, out, bias=bias, act=act, Z=Z)


def gemm_dgrad(g, W, out, act=None, dact_from=None, res=None):
    """out = (g W) * act'(dact_from) (+ res) for an nn.Linear weight W [N][K] and g [M][N]: the input gradient."""
    M, N = g.shape
    K = W.size(1)
    if (0 < M <= RS_MAX_ROWS and N % 4 == 0 and K % 4 == 0 and W.is_contiguous() and g.data_ptr() % 16 == 0
            and _ld(g) % 4 == 0):
        if t2_ok(M, K, N):        # ... for an input gradient that is the transposed copy [K][N]
            return gemm_rs(g, weight_t(W), out, act=act, dact_from=dact_from, res=res, N=K, K=N, t2=True)
        return gemm_rs(g, W, out, act=act, dact_from=dact_from, res=res, b_kmajor=True, N=K, K=N, fallback=False)
    if act == "sspo":
        raise _lib.MsdeHipError("gemm_dgrad: 'sspo' needs the row-strip kernel")
    if res is not None:
        if dact_from is not None:
            raise _lib.MsdeHipError("gemm_dgrad: residual + activation derivative needs the row-strip kernel")
        out.copy_(res)
        return gemm_ex(g, W, out, b_kmajor=True, accumulate=True)
    return gemm_ex(g, W, out, b_kmajor=True, act=act, dact_from=dact_from)


def _bn_fin_fwd(stats, strips, srows, M, C, gamma, beta, eps, momentum, rm, rv):
    dev = stats.device
    vec = torch.empty(4, C, dtype=torch.float32, device=dev)          # scale | shift | mean | rstd
    _lib.call("msde_bn_fin_fwd", _p(stats), strips, srows, M, _p(bound_tensor(M)), C, _p(gamma), _p(beta), float(eps),
              float(momentum), _p(rm), _p(rv), _p(vec[0]), _p(vec[1]), _p(vec[2]), _p(vec[3]), _stream())
    return vec


def _bn_fin_bwd(stats, strips, M, C, gamma, mean, rstd, need_affine_grads=True):
    dev = stats.device
    vec = torch.empty(3, C, dtype=torch.float32, device=dev)          # p | w | u
    gb = torch.empty(2, C, dtype=torch.float32, device=dev) if need_affine_grads else None
    _lib.call("msde_bn_fin_bwd", _p(stats), strips, M, _p(bound_tensor(M)), C, _p(gamma), _p(mean), _p(rstd), _p(vec[0]),
              _p(vec[1]), _p(vec[2]), _p(gb[0] if gb is not None else None), _p(gb[1] if gb is not None else None), _stream())
    return vec, gb


BN_BWD_FUSED_FIN = True     # ... with the finish of the strip partials inside the pass (msde_bn_bwd_fin_cols): one launch, not two


def _bn_bwd_fin_cols(stats, strips, M, C, gamma, mean, rstd, G, Z, xf3, xf4, out, rows):
    """(dgamma | dbeta) [2, C]; out = the BatchNorm input gradient (see msde_bn_bwd_fin_cols)."""
    gb = torch.empty(2, C, dtype=torch.float32, device=stats.device)
    _lib.call("msde_bn_bwd_fin_cols", _p(stats), strips, _p(gamma), _p(mean), _p(rstd), _p(G), _ld(G), _p(Z), _ld(Z), _p(xf3),
              _p(xf4), M, _p(rows), C, _p(out), _ld(out), _p(gb[0]), _p(gb[1]), _stream())
    return gb


BN_BWD_PASS = True     # GIN backward: BatchNorm input gradient as a streaming pass + plain product: 2.543 vs 2.587 ms (False: on the A fragments)


class _GinMlpBN(torch.autograd.Function):
    """Linear(D, 2D) -> BatchNorm1d -> ReLU -> Linear(2D, D) -> BatchNorm1d (-> ReLU) of a GIN layer in training mode
    (molecule_gnn_model.py:17,28-29,176-182) on the row-strip GEMMs: batch statistics in the epilogue of the producing
    product, finished by one small launch; BatchNorm apply + ReLU in the A load of the consuming product; backward:
    ReLU gate + BatchNorm-backward partial sums in the epilogue of the input-gradient product, the BatchNorm input
    gradient formed in the A load of the next one.  Forward 5 launches, backward 5 (+ the queued weight gradients)."""

    @staticmethod
    def forward(ctx, agg, W1, b1, g1, be1, rm1, rv1, W2, b2, g2, be2, rm2, rv2, eps1, mom1, eps2, mom2, relu_out, link_out):
        agg = _f32(agg)
        M, D = agg.shape
        H = W1.size(0)
        dev = agg.device
        s1, r1 = rs_geometry(M, H, D)
        st1 = torch.empty(s1, 2, H, dtype=torch.float32, device=dev)
        z1 = torch.empty(M, H, dtype=torch.float32, device=dev)
        gemm_node(agg, W1, z1, True, H, D, bias=b1, stats=st1, stats_mode="bnfwd")
        v1 = _bn_fin_fwd(st1, s1, r1, M, H, g1, be1, eps1, mom1, rm1, rv1)
        s2, r2 = rs_geometry(M, D, H)
        st2 = torch.empty(s2, 2, D, dtype=torch.float32, device=dev)
        a1 = torch.empty(M, H, dtype=torch.float32, device=dev)
        z2 = torch.empty(M, D, dtype=torch.float32, device=dev)
        # (round 5: the forward transform as a pass of its own + plain product is neither faster nor slower -- 2.549 vs 2.545 ms,
        # one launch more; it stays on the A fragments.  The BACKWARD transform is the one that pays as a pass: BN_BWD_PASS.)
        stamp("gin_gemm2_start", seq=True)          # no-ops unless enable_stamps(): bench.py times this launch inside the captured step
        gemm_node(z1, W2, z2, True, D, H, bias=b2, axf="affine", xf=(v1[0], v1[1]), relu=True, A_out=a1, stats=st2,
                  stats_mode="bnfwd")
        stamp("gin_gemm2_end", seq=True)
        v2 = _bn_fin_fwd(st2, s2, r2, M, D, g2, be2, eps2, mom2, rm2, rv2)
        h = torch.empty(M, D, dtype=torch.float32, device=dev)
        if link_out is not None:
            # the next layer's aggregation applies this BatchNorm while it gathers (and fills h): no launch here
            link_out.append(BnLink(z2, v2, relu_out))
            ctx.link = link_out[0]
        else:
            _lib.call("msde_affine_cols", _p(z2), M, D, _p(v2[0]), _p(v2[1]), int(relu_out), _p(h), _stream())
            ctx.link = None
        ctx.save_for_backward(agg, z1, a1, z2, h, v1, v2, W1, W2, g1, g2)
        ctx.relu_out = bool(relu_out)
        ctx.deferrable = all(t.is_leaf or getattr(t, "_msde_leaf_like", False) for t in (W1, b1, W2, b2))
        return h

    @staticmethod
    def backward(ctx, g):
        agg, z1, a1, z2, h, v1, v2, W1, W2, g1, g2 = ctx.saved_tensors
        g = _f32(g)
        M, D = agg.shape
        H = W1.size(0)
        dev = g.device
        st = _stream()
        # BatchNorm 2 backward: partial sums of the incoming gradient (gated by the output ReLU), finished, and the input
        # gradient formed while the next product loads its A strip
        sb = (M + 63) // 64
        stb = torch.empty(sb, 2, D, dtype=torch.float32, device=dev)
        _lib.call("msde_bn_bwd_colstats", _p(g), _p(z2), _p(h if ctx.relu_out else None), _p(v2[2]), M,
                  _p(bound_tensor(M)), D, _p(stb), st)
        fused_fin = BN_BWD_PASS and BN_BWD_FUSED_FIN
        if not fused_fin:
            pw2, gb2 = _bn_fin_bwd(stb, sb, M, D, g2, v2[2], v2[3])
        # g_a1 = dz2 W2, gated by the ReLU behind BatchNorm 1 (a1 > 0), with BatchNorm 1's partial sums
        sa, _ = rs_geometry(M, H, D)
        sta = torch.empty(sa, 2, H, dtype=torch.float32, device=dev)
        dz2 = torch.empty(M, D, dtype=torch.float32, device=dev)
        ga1 = torch.empty(M, H, dtype=torch.float32, device=dev)
        rows = bound_tensor(M)
        if BN_BWD_PASS:
            # the BatchNorm input gradient as a streaming pass in front of a PLAIN product (see msde_bn_bwd_cols)
            if fused_fin:
                gb2 = _bn_bwd_fin_cols(stb, sb, M, D, g2, v2[2], v2[3], g, z2, v2[0] if ctx.relu_out else None,
                                       v2[1] if ctx.relu_out else None, dz2, rows)
            else:
                _lib.call("msde_bn_bwd_cols", _p(g), _ld(g), _p(z2), _ld(z2), _p(pw2[0]), _p(pw2[1]), _p(pw2[2]),
                          _p(v2[0] if ctx.relu_out else None), _p(v2[1] if ctx.relu_out else None), M, _p(rows), D, _p(dz2), D, st)
            gemm_node(dz2, W2, ga1, False, H, D, act="relu", dact_from=a1, stats=sta, stats_mode="bnbwd", stats_z=z1,
                      stats_mean=v1[2])
        else:
            gemm_node(g, W2, ga1, False, H, D, axf="bnbwd",
                      xf=(pw2[0], pw2[1], pw2[2]) + ((v2[0], v2[1]) if ctx.relu_out else (None, None)), A2=z2, A_out=dz2,
                      act="relu", dact_from=a1, stats=sta, stats_mode="bnbwd", stats_z=z1, stats_mean=v1[2])
        if not fused_fin:
            pw1, gb1 = _bn_fin_bwd(sta, sa, M, H, g1, v1[2], v1[3])
        dz1 = torch.empty(M, H, dtype=torch.float32, device=dev)
        g_agg = torch.empty(M, D, dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        if BN_BWD_PASS:
            if fused_fin:
                gb1 = _bn_bwd_fin_cols(sta, sa, M, H, g1, v1[2], v1[3], ga1, z1, None, None, dz1, rows)
            else:
                _lib.call("msde_bn_bwd_cols", _p(ga1), _ld(ga1), _p(z1), _ld(z1), _p(pw1[0]), _p(pw1[1]), _p(pw1[2]), _p(None), _p(None),
                          M, _p(rows), H, _p(dz1), H, st)
            if g_agg is not None:
                gemm_node(dz1, W1, g_agg, False, D, H)
        elif g_agg is not None:
            gemm_node(ga1, W1, g_agg, False, D, H, axf="bnbwd", xf=(pw1[0], pw1[1], pw1[2]), A2=z1, A_out=dz1)
        else:       # nothing upstream wants a gradient: only dz1 for the weight gradient (product result discarded)
            scratch = torch.empty(M, D, dtype=torch.float32, device=dev)
            gemm_node(ga1, W1, scratch, False, D, H, axf="bnbwd", xf=(pw1[0], pw1[1], pw1[2]), A2=z1, A_out=dz1)
        gW2, gbias2 = weight_grad(dz2, a1, True, ctx.deferrable)
        gW1, gbias1 = weight_grad(dz1, agg, True, ctx.deferrable)
        return (g_agg, gW1, gbias1, gb1[0], gb1[1], None, None, gW2, gbias2, gb2[0], gb2[1], None, None,
                None, None, None, None, None, None)


def gin_mlp_bn(agg, lin1, bn1, lin2, bn2, relu_out, defer_apply=False):
    """mlp(agg) followed by the layer's outer BatchNorm (+ ReLU) -- see _GinMlpBN.  Modules give the parameters.
    defer_apply: returns (h, link) with h allocated but NOT yet written; the caller must pass (h, link) to the next
    layer's hip.gin_aggregate, whose kernel applies the BatchNorm on the fly and fills h."""
    box = [] if defer_apply else None
    h = _GinMlpBN.apply(agg, lin1.weight, lin1.bias, bn1.weight, bn1.bias, bn1.running_mean, bn1.running_var,
                        lin2.weight, lin2.bias, bn2.weight, bn2.bias, bn2.running_mean, bn2.running_var,
                        bn1.eps, 0.1 if bn1.momentum is None else bn1.momentum, bn2.eps,
                        0.1 if bn2.momentum is None else bn2.momentum, relu_out, box)
    return (h, box[0]) if defer_apply else h


class _PairBnReluLinear(torch.autograd.Function):
    """edge_2D_emb of the 2D->3D model on the edges of the (extended) graph: Linear(cat(h_row, h_col)) -> BatchNorm1d ->
    ReLU -> Linear (SDE_model_2D_to_3D.py:35-40,264-271), given AB = [h W_row^T | h W_col^T + b] from ONE node-level product
    (hip.pair_linear).  Forward 3 launches: gather-add that also writes the BatchNorm strip statistics of its result, their
    finish, the second Linear with BatchNorm apply + ReLU in its A load (the normalised tensor is written once, by that
    product, for the weight gradient).  Backward 4 launches (+ the queued weight gradient): input gradient of the second
    Linear gated by the ReLU with the BatchNorm-backward sums in its epilogue, their finish, and the two segment sums of the
    gradient of AB taken of (g, z) and combined per column -- the BatchNorm input gradient is never materialised.  Replaces
    gather-add, statistics, apply, product / product, partial sums, apply, two segment sums (9 launches, ~250 MB more traffic
    at 35 k edges x 300)."""

    @staticmethod
    def forward(ctx, AB, plan, gamma, beta, rm, rv, eps, momentum, W2, b2):
        AB = _f32(AB)
        E, D = plan.E, AB.size(1) // 2
        H = W2.size(0)
        dev = AB.device
        SR = pair_strip()
        strips = (E + SR - 1) // SR
        pre = torch.empty(E, D, dtype=torch.float32, device=dev)
        st = torch.empty(strips, 2, D, dtype=torch.float32, device=dev)
        _lib.call("msde_pair_gather_add_stats", _p(AB), AB.data_ptr() + 4 * D, 2 * D, _p(plan.src), _p(plan.dst), E, D,
                  _p(bound_tensor(E)), _p(pre), _p(st), _stream())
        v = _bn_fin_fwd(st, strips, SR, E, D, gamma, beta, eps, momentum, rm, rv)
        y = torch.empty(E, D, dtype=torch.float32, device=dev)
        out = torch.empty(E, H, dtype=torch.float32, device=dev)
        gemm_node(pre, W2, out, True, H, D, bias=b2, axf="affine", xf=(v[0], v[1]), relu=True, A_out=y)
        ctx.save_for_backward(pre, y, v, W2, gamma, AB)
        ctx.plan = plan
        ctx.deferrable = all(t.is_leaf or getattr(t, "_msde_leaf_like", False) for t in (W2, b2))
        return out

    @staticmethod
    def backward(ctx, g):
        pre, y, v, W2, gamma, AB = ctx.saved_tensors
        plan = ctx.plan
        g = _f32(g)
        E, D = pre.shape
        H = W2.size(0)
        dev = g.device
        ga = torch.empty(E, D, dtype=torch.float32, device=dev)
        if H in (16, 32) and W2.is_contiguous() and W2.data_ptr() % 16 == 0:
            # K = H is tiny: the product, the ReLU gate and the BatchNorm-backward strip sums in one streaming kernel
            SR = pair_strip()
            sa = (E + SR - 1) // SR
            sta = torch.empty(sa, 2, D, dtype=torch.float32, device=dev)
            _lib.call("msde_pair_bn_dgrad_stats", _p(g), _ld(g), _p(W2), _p(pre), _p(v[0]), _p(v[1]), _p(v[2]), E, H, D,
                      _p(bound_tensor(E)), _p(ga), _p(sta), _stream())
        else:
            sa, _ = rs_geometry(E, D, H)
            sta = torch.empty(sa, 2, D, dtype=torch.float32, device=dev)
            gemm_node(g, W2, ga, False, D, H, act="relu", dact_from=y, stats=sta, stats_mode="bnbwd", stats_z=pre,
                      stats_mean=v[2])
        pw, gb = _bn_fin_bwd(sta, sa, E, D, gamma, v[2], v[3])
        g_AB = None
        if ctx.needs_input_grad[0]:
            g_AB = torch.empty(plan.N, 2 * D, dtype=torch.float32, device=dev)
            _lib.call("msde_pair_bn_scatter", _p(ga), _p(AB), D, _p(plan.src), _p(plan.dst), _p(plan.rowptr_s), _p(plan.perm_s),
                      _p(plan.rowptr), plan.N, _p(pw[0]), _p(pw[1]), _p(pw[2]), _p(g_AB), _stream())
        gW2, gb2 = weight_grad(g, y, True, ctx.deferrable)
        return g_AB, None, gb[0], gb[1], None, None, None, None, gW2, gb2


_PAIR_STRIP = None


def pair_strip():
    global _PAIR_STRIP
    if _PAIR_STRIP is None:
        _PAIR_STRIP = int(_lib.load().msde_pair_strip())
    return _PAIR_STRIP


def pair_bn_relu_linear_ok(AB, bn, lin2):
    D = AB.size(1) // 2
    # D <= 768: the second Linear runs with an A transform, which msde_gemm_t2 / msde_gemm_rs take up to K = 768 (beyond that
    # the launch would fail with MSDE_EUNSUP and the fused autograd node has no fallback: refuse here instead)
    return (AB.is_cuda and AB.dtype == torch.float32 and AB.dim() == 2 and AB.size(1) == 2 * D and D % 4 == 0 and D <= 768
            and AB.is_contiguous() and AB.data_ptr() % 16 == 0
            and lin2.weight.size(1) == D and lin2.weight.size(0) % 4 == 0 and lin2.bias is not None
            and lin2.weight.is_contiguous() and bn.weight.data_ptr() % 16 == 0 and bn.bias.data_ptr() % 16 == 0)


def pair_bn_relu_linear(AB, plan, bn, lin2):
    """lin2(relu(bn(AB[src, :D] + AB[dst, D:]))) for a training-mode BatchNorm1d `bn` and nn.Linear `lin2`: _PairBnReluLinear."""
    return _PairBnReluLinear.apply(AB, plan, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum,
                                   lin2.weight, lin2.bias)


class _SchNetTail(torch.autograd.Function):
    """h + lin(ssp(lin2(agg))) of a SchNet interaction (schnet.py:163-167,97,189; CFConv.lin2, InteractionBlock.act /
    lin, the residual of SchNet.forward) as two products: bias + shifted softplus in the first epilogue, bias + residual in
    the second; backward: the softplus derivative (from its saved output) in the epilogue of lin's input-gradient product."""

    @staticmethod
    def forward(ctx, agg, h, W2, b2, Wl, bl):
        agg, h = _f32(agg), _f32(h)
        M = agg.size(0)
        Hd = W2.size(0)
        a = torch.empty(M, Hd, dtype=torch.float32, device=agg.device)
        gemm_fwd(agg, W2, a, bias=b2, act="ssp")
        out = torch.empty(M, Wl.size(0), dtype=torch.float32, device=agg.device)
        gemm_fwd(a, Wl, out, bias=bl, res=h)
        ctx.save_for_backward(agg, a, W2, Wl)
        ctx.deferrable = all(t.is_leaf for t in (W2, b2, Wl, bl))
        return out

    @staticmethod
    def backward(ctx, g):
        agg, a, W2, Wl = ctx.saved_tensors
        g = _f32(g)
        M = g.size(0)
        gx = torch.empty(M, Wl.size(1), dtype=torch.float32, device=g.device)       # d/d(lin2 output)
        gemm_dgrad(g, Wl, gx, act="sspo", dact_from=a)
        g_agg = None
        if ctx.needs_input_grad[0]:
            g_agg = torch.empty(M, W2.size(1), dtype=torch.float32, device=g.device)
            gemm_dgrad(gx, W2, g_agg)
        gWl, gbl = weight_grad(g, a, True, ctx.deferrable)
        gW2, gb2 = weight_grad(gx, agg, True, ctx.deferrable)
        return g_agg, (g if ctx.needs_input_grad[1] else None), gW2, gb2, gWl, gbl


def schnet_tail(agg, h, lin2, lin):
    return _SchNetTail.apply(agg, h, lin2.weight, lin2.bias, lin.weight, lin.bias)


# ---- diagnostics: device timestamps in stream order (tools/probes/step_timeline.py) ---------------------------
STAMPS = None            # {"buf": int64[256] device tensor, "names": [..]} when enabled


def enable_stamps(device, family=False):
    """family: also stamp around every msde_gemm_t2 launch (`t2_start#k` / `t2_end#k`; shapes in STAMPS["t2_shapes"], in launch
    order since the last "step_start")."""
    global STAMPS
    STAMPS = {"buf": torch.zeros(1024, dtype=torch.int64, device=device), "names": [], "family": bool(family), "t2_shapes": []}


def stamp(name, seq=False):
    """No-op unless enable_stamps() was called: one 1-thread launch on the current stream storing the real-time counter.
    seq: a name that occurs several times per step (one per GIN layer) gets its own slot per occurrence, `name#k`, counted
    from the step's "step_start" stamp."""
    if STAMPS is None:
        return
    if name == "step_start":
        STAMPS["seq"] = {}
        STAMPS["t2_shapes"] = []
    if seq:
        k = STAMPS.setdefault("seq", {}).get(name, 0)
        STAMPS["seq"][name] = k + 1
        name = "%s#%d" % (name, k)
    if name not in STAMPS["names"]:
        STAMPS["names"].append(name)
    i = STAMPS["names"].index(name)
    _lib.call("msde_debug_stamp", ctypes.c_void_p(STAMPS["buf"].data_ptr() + 8 * i), _stream())


def read_stamps():
    t = STAMPS["buf"].cpu().tolist()
    return {n: t[i] for i, n in enumerate(STAMPS["names"])}
