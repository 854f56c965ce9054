"""A CLOSED set of twice-differentiable operators on HIP kernels (MD17 force fine-tuning, examples/finetune_MD17.py:47-78).

The reference obtains forces as F = -dE/dpos with `create_graph=True` and then back-propagates loss(E, F): the energy
network (Geom3D/models/schnet.py:85-125) is differentiated TWICE, by autograd over ATen operators.  Here every operator
on the path positions -> energy is a `torch.autograd.Function` whose forward is a kernel of csrc/ (dd.hip, gemm_ex.hip,
linear.hip's weight-gradient kernel, the cfconv aggregation kernels) and whose backward is written with operators OF THIS
SAME SET.  Autograd therefore only ever records library launches, at any order of differentiation: the set is closed.

    dense        linear (x W^T + b), mm_nn (g W), mm_nt (x W^T), mm_tn (g^T x), colsum <-> broadcast_rows
    pointwise    ssp / cutoff derivatives of order 0..2, recip, mul, add, scale
    per edge     rbf derivatives of order 0..2 (row expanding), mul_rows <-> row_dot, edge_diff <-> edge_scatter, row_norm
    per molecule seg_reduce <-> seg_expand
    message passing   hip._EdgeAgg / _EdgeAggT / _EdgeProd

Third derivatives (the backward of an order-2 pointwise op) are not needed by a loss on energies and forces and raise.
"""
import torch

from . import _lib, hip, slabs

_p, _f32, _stream = hip._p, hip._f32, hip._stream

# counts launches per entry point; tests assert the force path ran here and nowhere else
CALLS = {}


def _call(name, *args):
    CALLS[name] = CALLS.get(name, 0) + 1
    _lib.call(name, *args)


def _new(*shape, like):
    return torch.empty(*shape, dtype=torch.float32, device=like.device)


# ------------------------------------------------------------------------------------------------ pointwise
class _Unary(torch.autograd.Function):
    """kind 0: shifted softplus (schnet.py:213-216); 1: cosine cutoff (schnet.py:186), p0 = cutoff; derivative `order`."""

    @staticmethod
    def forward(ctx, x, kind, order, p0, mask):
        x = _f32(x)
        y = torch.empty_like(x)
        _call("msde_dd_unary", _p(x), _p(mask), x.numel(), kind, order, float(p0), _p(y), _stream())
        ctx.save_for_backward(x)
        ctx.cfg = (kind, order, p0, mask)
        return y

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        kind, order, p0, mask = ctx.cfg
        if order >= 2:
            raise NotImplementedError("third derivative of a pointwise operator of the force path")
        return _UnaryMul.apply(g, None, x, kind, order + 1, p0, mask), None, None, None, None


class _UnaryMul(torch.autograd.Function):
    """y = g * (g2 ? g2 : 1) * f^(order)(x): the chain-rule product of _Unary's backward as ONE launch (msde_dd_unary_mul)
    instead of a derivative launch and one or two product launches.  Closed: its own backward is two more of the same."""

    @staticmethod
    def forward(ctx, g, g2, x, kind, order, p0, mask):
        g, x = _f32(g), _f32(x)
        g2 = _f32(g2) if g2 is not None else None
        assert g.shape == x.shape and (g2 is None or g2.shape == x.shape)
        y = torch.empty_like(x)
        _call("msde_dd_unary_mul", _p(g), _p(g2), _p(x), _p(mask), x.numel(), kind, order, float(p0), _p(y), _stream())
        ctx.save_for_backward(g, x)
        ctx.cfg = (kind, order, p0, mask, g2 is not None)
        return y

    @staticmethod
    def backward(ctx, u):
        g, x = ctx.saved_tensors
        kind, order, p0, mask, three = ctx.cfg
        if three or order >= 2:
            raise NotImplementedError("third derivative of a pointwise operator of the force path")
        ni = ctx.needs_input_grad
        return (_UnaryMul.apply(u, None, x, kind, order, p0, mask) if ni[0] else None), None, \
            (_UnaryMul.apply(u, g, x, kind, order + 1, p0, mask) if ni[2] else None), None, None, None, None


def ssp(x):
    return _Unary.apply(x, 0, 0, 0.0, None)


def silu(x):
    return _Unary.apply(x, 3, 0, 0.0, None)


def sqrt_eps(x, eps):
    """sqrt(x + eps)"""
    return _Unary.apply(x, 4, 0, float(eps), None)


def cosine_cutoff(d, cutoff, src):
    """0.5 (cos(pi d / cutoff) + 1); padded edge slots (src < 0) -> 0 at every order."""
    return _Unary.apply(d, 1, 0, cutoff, src)


class _Recip(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _f32(x)
        y = torch.empty_like(x)
        _call("msde_dd_unary", _p(x), _p(None), x.numel(), 2, 0, 0.0, _p(y), _stream())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        r = _Recip.apply(x)
        return scale(mul(g, mul(r, r)), -1.0)


recip = _Recip.apply


class _Mul(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _f32(a), _f32(b)
        assert a.shape == b.shape
        y = torch.empty_like(a)
        _call("msde_dd_binary", _p(a), _p(b), a.numel(), 0, 1.0, _p(y), _stream())
        ctx.save_for_backward(a, b)
        return y

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        return (mul(g, b) if ctx.needs_input_grad[0] else None), (mul(g, a) if ctx.needs_input_grad[1] else None)


mul = _Mul.apply


class _Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _f32(a), _f32(b)
        assert a.shape == b.shape
        y = torch.empty_like(a)
        _call("msde_dd_binary", _p(a), _p(b), a.numel(), 1, 1.0, _p(y), _stream())
        return y

    @staticmethod
    def backward(ctx, g):
        return g, g


add = _Add.apply


class _SumN(torch.autograd.Function):
    """x_0 + ... + x_{n-1} (n <= 8) in one launch; its backward hands the same gradient to every addend (no launch)."""

    @staticmethod
    def forward(ctx, *xs):
        import ctypes
        xs = [_f32(x) for x in xs]
        y = torch.empty_like(xs[0])
        arr = (ctypes.c_void_p * len(xs))(*[x.data_ptr() for x in xs])
        _call("msde_dd_sum_n", ctypes.cast(arr, ctypes.c_void_p), len(xs), y.numel(), _p(y), _stream())
        ctx.n = len(xs)
        return y

    @staticmethod
    def backward(ctx, g):
        return (g,) * ctx.n


def sum_n(xs):
    xs = list(xs)
    while len(xs) > 8:
        xs = [_SumN.apply(*xs[:8])] + xs[8:]
    return xs[0] if len(xs) == 1 else _SumN.apply(*xs)


class _Fanout(torch.autograd.Function):
    """n aliases of x for n consumers: their gradients come back to ONE node, which sums them with one launch (sum_n) instead of
    the n - 1 additions autograd's accumulation would launch -- in the first differentiation and, since sum_n is a member
    of the closed set, in the second."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.set_materialize_grads(False)
        return tuple(x.detach() for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        gs = [g for g in gs if g is not None]
        if not gs:
            return None, None
        if len(gs) > 1 and not torch.is_grad_enabled() and any(not g.is_contiguous() for g in gs):
            y = _sum_rows_n(gs)           # column blocks of wider gradient buffers, summed where they lie
            if y is not None:
                return y, None
        return sum_n(gs), None


def _sum_rows_n(gs):
    """g_0 + ... + g_{n-1} for 2-D fp32 device tensors with unit column stride and row strides of their own (one launch,
    msde_dd_sum_rows_n); None when the operands do not qualify (the caller falls back to sum_n on contiguous copies)."""
    import ctypes
    g0 = gs[0]
    if len(gs) > 8 or g0.dim() != 2 or g0.size(1) % 4:
        return None
    for g in gs:
        if (not g.is_cuda or g.dtype != torch.float32 or g.shape != g0.shape or g.stride(1) != 1 or g.stride(0) % 4
                or g.stride(0) < g.size(1) or g.data_ptr() % 16):
            return None
    y = torch.empty(g0.shape, dtype=torch.float32, device=g0.device)
    arr = (ctypes.c_void_p * len(gs))(*[g.data_ptr() for g in gs])
    lds = (ctypes.c_int * len(gs))(*[g.stride(0) for g in gs])
    _call("msde_dd_sum_rows_n", ctypes.cast(arr, ctypes.c_void_p), ctypes.cast(lds, ctypes.c_void_p), len(gs), g0.size(0),
          g0.size(1), _p(y), _stream())
    return y


def fanout(x, n):
    return _Fanout.apply(x, n) if n > 1 else (x,)


class _Scale(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, alpha):
        x = _f32(x)
        y = torch.empty_like(x)
        _call("msde_dd_binary", _p(x), _p(None), x.numel(), 2, float(alpha), _p(y), _stream())
        ctx.alpha = alpha
        return y

    @staticmethod
    def backward(ctx, g):
        return scale(g, ctx.alpha), None


scale = _Scale.apply


# ------------------------------------------------------------------------------------------------ per edge
class _Rbf(torch.autograd.Function):
    """Gaussian smearing exp(coeff (d - offset)^2) (schnet.py:205-207), derivative `order` in d: [E] -> [E, G]."""

    @staticmethod
    def forward(ctx, d, src, offset, coeff, order):
        d = _f32(d)
        E, G = d.numel(), offset.numel()
        y = _new(E, G, like=d)
        _call("msde_dd_rbf", _p(d), _p(src), _p(offset), E, G, float(coeff), order, _p(y), _stream())
        ctx.save_for_backward(d)
        ctx.cfg = (src, offset, coeff, order)
        return y

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        src, offset, coeff, order = ctx.cfg
        if order >= 2:
            raise NotImplementedError("third derivative of the Gaussian smearing")
        return row_dot(g, _Rbf.apply(d, src, offset, coeff, order + 1)), None, None, None, None


def rbf(d, src, offset, coeff):
    return _Rbf.apply(d, src, offset, coeff, 0)


class _MulRows(torch.autograd.Function):
    """y[e, :] = M[e, :] * s[e]"""

    @staticmethod
    def forward(ctx, M, s):
        M, s = _f32(M), _f32(s)
        E, K = M.shape
        assert s.numel() == E
        y = torch.empty_like(M)
        _call("msde_dd_mul_rows", _p(M), _p(s), E, K, _p(y), _stream())
        ctx.save_for_backward(M, s)
        return y

    @staticmethod
    def backward(ctx, g):
        M, s = ctx.saved_tensors
        return (mul_rows(g, s) if ctx.needs_input_grad[0] else None), (row_dot(g, M) if ctx.needs_input_grad[1] else None)


mul_rows = _MulRows.apply


class _RowDot(torch.autograd.Function):
    """y[e] = sum_k a[e, k] b[e, k]"""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _f32(a), _f32(b)
        assert a.shape == b.shape
        E, K = a.shape
        y = _new(E, like=a)
        _call("msde_dd_row_dot", _p(a), _p(b), E, K, _p(y), _stream())
        ctx.save_for_backward(a, b)
        return y

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        return (mul_rows(b, g) if ctx.needs_input_grad[0] else None), (mul_rows(a, g) if ctx.needs_input_grad[1] else None)


row_dot = _RowDot.apply


class _Split3(torch.autograd.Function):
    """[E, 3] -> three contiguous [E] component vectors (views of one [3, E] buffer, returned as three OUTPUTS)"""

    @staticmethod
    def forward(ctx, x):
        x = _f32(x)
        E = x.size(0)
        y = _new(3, E, like=x)
        _call("msde_dd_transpose3", _p(x), E, 1, _p(y), _stream())
        return y[0], y[1], y[2]

    @staticmethod
    def backward(ctx, g0, g1, g2):
        return _Merge3.apply(g0, g1, g2)


class _Merge3(torch.autograd.Function):
    """three [E] vectors (None = zeros) -> [E, 3]"""

    @staticmethod
    def forward(ctx, a, b, c):
        ref = next(t for t in (a, b, c) if t is not None)
        E = ref.numel()
        a, b, c = (None if t is None else _f32(t) for t in (a, b, c))
        y = _new(E, 3, like=ref)
        _call("msde_dd_merge3", _p(a), _p(b), _p(c), E, _p(y), _stream())
        ctx.has = (a is not None, b is not None, c is not None)
        return y

    @staticmethod
    def backward(ctx, g):
        g0, g1, g2 = _Split3.apply(g)
        return tuple(t if h else None for t, h in zip((g0, g1, g2), ctx.has))


components = _Split3.apply


class _EdgeDiff(torch.autograd.Function):
    """diff[e] = pos[src_e] - pos[dst_e]  (schnet.py:98-99), zero on padded slots."""

    @staticmethod
    def forward(ctx, pos, plan):
        pos = _f32(pos)
        y = _new(plan.E, 3, like=pos)
        _call("msde_dd_edge_diff", _p(pos), _p(plan.src), _p(plan.dst), plan.E, _p(y), _stream())
        ctx.plan = plan
        return y

    @staticmethod
    def backward(ctx, g):
        return _EdgeScatter.apply(g, ctx.plan), None


class _EdgeScatter(torch.autograd.Function):
    """adjoint of _EdgeDiff: out[i] = sum_{src_e = i} g[e] - sum_{dst_e = i} g[e], fixed summation order"""

    @staticmethod
    def forward(ctx, g, plan):
        g = _f32(g)
        y = _new(plan.N, 3, like=g)
        _call("msde_dd_edge_scatter", _p(g), _p(plan.rowptr), _p(plan.rowptr_s), _p(plan.perm_s), plan.N, _p(y), _stream())
        ctx.plan = plan
        return y

    @staticmethod
    def backward(ctx, u):
        return _EdgeDiff.apply(u, ctx.plan), None


edge_diff = _EdgeDiff.apply


class _RowNorm(torch.autograd.Function):
    """d[e] = |v[e]| over the 3 coordinates (1 on padded slots, where v = 0 has no direction)."""

    @staticmethod
    def forward(ctx, v, plan):
        v = _f32(v)
        y = _new(plan.E, like=v)
        _call("msde_dd_row_norm", _p(v), _p(plan.src), plan.E, _p(y), _stream())
        ctx.save_for_backward(v)
        ctx.plan = plan
        return y

    @staticmethod
    def backward(ctx, g):
        (v,) = ctx.saved_tensors
        d = _RowNorm.apply(v, ctx.plan)
        return mul_rows(v, mul(g, recip(d))), None


row_norm = _RowNorm.apply


# ------------------------------------------------------------------------------------------------ per molecule
class _SegReduce(torch.autograd.Function):
    """per-molecule sum / mean of atom rows (schnet.py:122 scatter)"""

    @staticmethod
    def forward(ctx, h, mol_ptr, batch_i32, mean):
        h = _f32(h)
        CALLS["msde_segment_sum_rows"] = CALLS.get("msde_segment_sum_rows", 0) + 1
        out = hip.segment_sum_rows(h, mol_ptr, None, mol_ptr.numel() - 1, mean=mean)
        ctx.cfg = (mol_ptr, batch_i32, mean)
        return out

    @staticmethod
    def backward(ctx, g):
        return _SegExpand.apply(g, *ctx.cfg), None, None, None


class _SegExpand(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g, mol_ptr, batch_i32, mean):
        g = _f32(g)
        N, K = batch_i32.numel(), g.size(1)
        y = _new(N, K, like=g)
        _call("msde_dd_seg_expand", _p(g), _p(batch_i32), _p(mol_ptr), N, K, int(bool(mean)), _p(y), _stream())
        ctx.cfg = (mol_ptr, batch_i32, mean)
        return y

    @staticmethod
    def backward(ctx, u):
        return _SegReduce.apply(u, *ctx.cfg), None, None, None


seg_reduce = _SegReduce.apply


# ------------------------------------------------------------------------------------------------ dense
PARAM_GRADS = True      # False inside positions_only(): this differentiation is w.r.t. the coordinates alone


class positions_only:
    """`with dd.positions_only(): force = -autograd.grad(energy, positions, create_graph=True)` -- the caller states that
    this differentiation does not ask for parameter gradients (finetune_MD17.py:68).  ctx.needs_input_grad cannot tell: it
    is fixed at forward time (every weight requires grad), so each Linear's backward would also form its weight / bias
    gradient -- a GEMM, a slab reduction and a column sum per layer that the engine then drops."""

    def __enter__(self):
        global PARAM_GRADS
        self._was, PARAM_GRADS = PARAM_GRADS, False

    def __exit__(self, *exc):
        global PARAM_GRADS
        PARAM_GRADS = self._was
        return False


def _gemm(A, B, bias=None, b_kmajor=False):
    M = A.size(0)
    N = B.size(1) if b_kmajor else B.size(0)
    out = _new(M, N, like=A)
    CALLS["msde_gemm_ex"] = CALLS.get("msde_gemm_ex", 0) + 1
    hip.gemm_ex(A, B, out, bias=bias, b_kmajor=b_kmajor)
    return out


class _Linear(torch.autograd.Function):
    """y = x W^T + b"""

    @staticmethod
    def forward(ctx, x, W, b):
        x, W = _f32(x), _f32(W)
        ctx.save_for_backward(x, W)
        ctx.has_bias = b is not None
        ctx.b_key = b.data_ptr() if (b is not None and b.is_leaf) else None
        return _gemm(x, W, bias=_f32(b) if b is not None else None)

    @staticmethod
    def backward(ctx, g):
        x, W = ctx.saved_tensors
        ni = ctx.needs_input_grad
        if not PARAM_GRADS:
            return (mm_nn(g, W) if ni[0] else None), None, None
        if ni[1] and not torch.is_grad_enabled():
            # the LAST differentiation (nothing will differentiate this backward again): weight and bias gradient from the
            # split-M kernel instead of mm_tn + colsum (two members of the closed set, two to three launches) -- queued for
            # the step's ONE grouped launch when the trainer has a parameter-gradient batch open (slabs.weight_grad_leaf).
            # EVERY contribution to one leaf W must take this path (the force path's _MMnn / _MMnt do): a deferred buffer is
            # filled only at finish_param_grad_batch(), so an immediate mm_tn result added to it by AccumulateGrad would be
            # summed with unfilled memory.  A frozen or non-leaf bias only drops the BIAS half of the deferred problem.
            g2 = _f32(g)
            CALLS["msde_linear_bwd_w"] = CALLS.get("msde_linear_bwd_w", 0) + 1
            fused_bias = ctx.has_bias and ni[2] and ctx.b_key is not None
            gW, gb = slabs.weight_grad_leaf(g2, x, fused_bias, W, ctx.b_key if fused_bias else None)
            if ctx.has_bias and ni[2] and not fused_bias:
                gb = colsum(g)              # a bias that is not a leaf parameter: its gradient flows on, formed right here
            return (mm_nn(g, W) if ni[0] else None), gW, gb
        return (mm_nn(g, W) if ni[0] else None), (mm_tn(g, x) if ni[1] else None), \
            (colsum(g) if ctx.has_bias and ni[2] else None)


def linear(x, W, b=None):
    return _Linear.apply(x, W, b)


class _MMnn(torch.autograd.Function):
    """y = g W   (g [M, N], W [N, K])"""

    @staticmethod
    def forward(ctx, g, W):
        g, W = _f32(g), _f32(W)
        ctx.save_for_backward(g, W)
        return _gemm(g, W, b_kmajor=True)

    @staticmethod
    def backward(ctx, u):
        g, W = ctx.saved_tensors
        ni = ctx.needs_input_grad
        if not PARAM_GRADS:
            return (mm_nt(u, W) if ni[0] else None), None
        if ni[1] and not torch.is_grad_enabled() and W.is_leaf:
            # last differentiation, W a parameter: its second contribution (the force path) joins the grouped launch
            CALLS["msde_linear_bwd_w"] = CALLS.get("msde_linear_bwd_w", 0) + 1
            return (mm_nt(u, W) if ni[0] else None), slabs.weight_grad_leaf(g, _f32(u), False, W)[0]
        return (mm_nt(u, W) if ni[0] else None), (mm_tn(g, u) if ni[1] else None)


mm_nn = _MMnn.apply


class _MMnt(torch.autograd.Function):
    """y = x W^T   (x [M, K], W [N, K])"""

    @staticmethod
    def forward(ctx, x, W):
        x, W = _f32(x), _f32(W)
        ctx.save_for_backward(x, W)
        return _gemm(x, W)

    @staticmethod
    def backward(ctx, u):
        x, W = ctx.saved_tensors
        ni = ctx.needs_input_grad
        if ni[1] and PARAM_GRADS and not torch.is_grad_enabled() and W.is_leaf:
            CALLS["msde_linear_bwd_w"] = CALLS.get("msde_linear_bwd_w", 0) + 1
            return (mm_nn(u, W) if ni[0] else None), slabs.weight_grad_leaf(_f32(u), x, False, W)[0]
        return (mm_nn(u, W) if ni[0] else None), (mm_tn(u, x) if ni[1] and PARAM_GRADS else None)


mm_nt = _MMnt.apply


class _MMtn(torch.autograd.Function):
    """y = g^T x   (g [M, N], x [M, K] -> [N, K]): the weight-gradient GEMM (csrc/linear.hip, split over M, fixed order)"""

    @staticmethod
    def forward(ctx, g, x):
        g, x = _f32(g), _f32(x)
        M, N = g.shape
        K = x.size(1)
        assert x.size(0) == M
        y = _new(N, K, like=g)
        ws = slabs._wgrad_workspace(M, N, K, g.device)
        _call("msde_linear_bwd_w", _p(g), _p(x), M, N, K, _p(y), _p(None), _p(ws), _p(hip.bound_tensor(M)), _stream())
        ctx.save_for_backward(g, x)
        return y

    @staticmethod
    def backward(ctx, u):
        g, x = ctx.saved_tensors
        ni = ctx.needs_input_grad
        return (mm_nt(x, u) if ni[0] else None), (mm_nn(g, u) if ni[1] else None)


mm_tn = _MMtn.apply


class _ColSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g):
        g = _f32(g)
        ctx.M = g.size(0)
        CALLS["msde_colsum"] = CALLS.get("msde_colsum", 0) + 1
        return slabs.colsum(g)

    @staticmethod
    def backward(ctx, u):
        return _BroadcastRows.apply(u, ctx.M)


class _BroadcastRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, b, M):
        b = _f32(b)
        y = _new(M, b.numel(), like=b)
        _call("msde_dd_broadcast_rows", _p(b), M, b.numel(), _p(y), _stream())
        return y

    @staticmethod
    def backward(ctx, g):
        return colsum(g), None


colsum = _ColSum.apply


def edge_aggregate(x, W, plan):
    CALLS["msde_cfconv_aggregate"] = CALLS.get("msde_cfconv_aggregate", 0) + 1
    return hip._EdgeAgg.apply(x, W, plan)
