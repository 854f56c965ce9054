"""Capture lifetime, grow-on-demand workspaces and the batched weight-gradient / slab-reduction machinery of a training step
(split out of hip.py in round 5; round 6: callers import this module -- `slabs.begin_param_grad_batch`, `slabs._SLABS`, ... -- hip.py no
longer re-exports it).

  * note_capture / _retire / _KEEP_ALIVE, no_gc: what a captured hipGraph needs to stay valid while eager steps go on;
  * _scratch / _wgrad_workspace / _ws_key: per-(device, stream) workspaces;
  * upload_table / flush_table_uploads: pointer tables of captured launches;
  * _SlabBatch (`_SLABS`): weight-gradient GEMMs queued during a backward pass -> ONE grouped launch + ONE slab reduction
    (include/msde_hip.h: msde_linear_bwd_w_grouped_ex, msde_reduce_slabs_multi), deferred leaf kernels, table slots;
  * weight_grad / weight_grad_leaf / weight_grad_blocks / colsum: the entry points the autograd functions call.
"""
import ctypes

import torch

from . import _lib
from ._torchabi import _stream, _p, _f32


def bound_tensor(rows):
    """hip.bound_tensor (the row-bound map in scope lives in hip.py; looked up at call time)."""
    from . import hip
    return hip.bound_tensor(rows)


def _bn_workspace(M, C, device):
    from . import hip
    return hip._bn_workspace(M, C, device)


WGRAD_HIP_MIN_ROWS = 64          # below this a split over M has nothing to split

_WS = {}          # per-device wgrad slab workspace, grown on demand (stream-ordered reuse)
_WS_BYTES = {}    # (M, N, K) -> workspace bytes


# Grow-on-demand workspaces and captured hipGraphs: a captured graph has the raw addresses of the workspaces it was
# captured with baked in.  Once any capture has happened (note_capture) an outgrown workspace is therefore never
# freed -- it is parked in _KEEP_ALIVE for the life of the process -- so a replay can never write into memory the
# caching allocator has handed to someone else.
_KEEP_ALIVE = []
_CAPTURED = False


def note_capture():
    global _CAPTURED
    _CAPTURED = True


class no_gc:
    """Wrap a hipGraph capture: the interpreter's cyclic garbage collector must not run inside it.  A collection that happens
    to fall into a capture finalises whatever garbage earlier code left behind -- another trainer's captured graphs, tensors
    of their memory pools -- and destroying a graph or releasing pool memory while a capture is under way aborts the process
    (seen in the test suite: `Fatal Python error: Aborted ... Garbage-collecting` inside Trainer.capture).  Collect first,
    then keep the collector off until the capture has ended.  collect=False: only the second half (a full collection costs
    ~20 ms of host time -- a tenth of a 1000-iteration sampling call -- and with the collector off nothing is finalised inside
    the capture either way)."""

    def __init__(self, collect=True):
        self._collect = collect

    def __enter__(self):
        import gc
        if self._collect:
            gc.collect()
        self._was = gc.isenabled()
        gc.disable()
        return self

    def __exit__(self, *exc):
        import gc
        if self._was:
            gc.enable()
        return False


def _retire(bufs):
    if _CAPTURED:
        _KEEP_ALIVE.extend(b for b in bufs if b is not None)


_SCRATCH = {}     # per-(device, stream) scratch of the table-gradient kernels
EMB_BWD_SPLIT = 64   # max slices per table row in msde_embedding_sum_bwd


def _scratch(nfloats, device):
    key = _ws_key(device)
    ws = _SCRATCH.get(key)
    if ws is None or ws.numel() < nfloats:
        _retire([ws])
        ws = torch.empty(max(nfloats, 1 << 20), dtype=torch.float32, device=key[0])
        _SCRATCH[key] = ws
    return ws


def _ws_key(device):
    """Workspaces are per (device, stream): kernels on concurrent streams must not share scratch."""
    return (device, torch.cuda.current_stream().cuda_stream)


def _wgrad_workspace(M, N, K, device):
    key = (M, N, K)
    nbytes = _WS_BYTES.get(key)
    if nbytes is None:
        nbytes = int(_lib.load().msde_linear_bwd_w_workspace_bytes(M, N, K))
        _WS_BYTES[key] = nbytes
    device = _ws_key(device)
    ws = _WS.get(device)
    if ws is None or ws.numel() * 4 < nbytes:
        _retire([ws])
        ws = torch.empty(max(nbytes // 4, 1 << 20), dtype=torch.float32, device=device[0])
        _WS[device] = ws
    return ws


# ---- batched slab reduction -----------------------------------------------------------------------------------
# Every split-M weight gradient is a GEMM that writes per-split slabs plus a reduction over the splits.  The
# reductions are leaves of the backward graph (only the optimiser reads their results), and the step is bound
# by the number of dependent launches: between begin_param_grad_batch() and finish_param_grad_batch() the GEMMs
# write their slabs into one arena and ONE kernel (msde_reduce_slabs_multi) sums the slabs of all layers at the
# end.  Until then the returned gradient tensors are allocated but not yet filled, which is safe exactly when
# nothing but autograd's leaf bookkeeping touches them: parameters used once per forward (`offload`) that are
# leaves or concatenation views of leaves.  Everything else keeps the immediate two-kernel path.
# Pointer tables (slab rows, grouped-GEMM problems, Adam chunks) are built on the host and uploaded.  Under hipGraph
# capture the addresses in them are static, so re-uploading them at every replay (a memcpy node each) is wasted
# work at the tail of the step: while capturing, uploads are only RECORDED here and the trainer performs them once
# after the capture (`flush_table_uploads`); the captured kernels just read the device tables.
_PENDING_UPLOADS = []


def upload_table(dev, host):
    if torch.cuda.is_current_stream_capturing():
        _PENDING_UPLOADS.append((dev, host))
    else:
        dev.copy_(host, non_blocking=True)


def flush_table_uploads():
    """After a capture ends: upload the tables its kernels read (their host images stay untouched afterwards)."""
    for dev, host in _PENDING_UPLOADS:
        dev.copy_(host, non_blocking=True)
    _PENDING_UPLOADS.clear()
    torch.cuda.synchronize()


class _SlabBatch:
    MAX_ROWS = 4096
    EAGER_SLOTS = 3

    def __init__(self):
        self.active = False
        self.arena = None
        self.used = 0
        self.rows = []           # (slab address, splits, n, out address, device)
        self.gemms = []          # queued weight-gradient GEMMs: (gY, X, M, N, K, has_bias, slab) -- inputs kept alive
        self.retired = []        # outgrown arenas still referenced by queued rows
        self.launched = []       # operands of GEMMs already launched in this backward pass (kept alive until finish)
        self.deferred = []       # leaf-only kernels queued by backward functions (run_deferred_leaf_kernels)
        self.leaf_first = {}     # weight_grad_leaf: parameter address -> (address, entries) of its first queued gradient
        self.leaf_more = []      # ... and the later contributions to the same parameter: (first address, tensor, level)
        self.slot = None
        self.slots = []          # slots 0..EAGER_SLOTS-1: the eager ring; one more per captured hipGraph
        self.events = []         # per slot: event recorded behind the last upload of its pinned host images
        self.slot_i = -1
        self.eager_i = 0

    def new_slot(self, device):
        """Pinned host image + device copy of the row / prefix tables (one per captured graph: the upload is a
        memcpy node that re-reads its host image at every replay)."""
        host_rows = torch.zeros(self.MAX_ROWS, 8, dtype=torch.int64).pin_memory()
        host_pre = torch.zeros(self.MAX_ROWS + 64, dtype=torch.int32).pin_memory()
        host_prob = torch.zeros(self.MAX_ROWS, 16, dtype=torch.int64).pin_memory()      # MSDE_WGRAD_ROW
        host_ppre = torch.zeros(self.MAX_ROWS + 1, dtype=torch.int32).pin_memory()
        self.slot = (host_rows, host_pre, torch.zeros(self.MAX_ROWS, 8, dtype=torch.int64, device=device),
                     torch.zeros(self.MAX_ROWS + 64, dtype=torch.int32, device=device),
                     host_prob, host_ppre, torch.zeros(self.MAX_ROWS, 16, dtype=torch.int64, device=device),
                     torch.zeros(self.MAX_ROWS + 1, dtype=torch.int32, device=device))
        self.slots.append(self.slot)
        self.events.append(None)
        self.slot_i = len(self.slots) - 1

    def _ensure_eager_ring(self, device):
        while len(self.slots) < self.EAGER_SLOTS:
            self.new_slot(device)

    def _rotate_eager(self, device):
        """Eager backward pass: take the next slot of the ring and wait for the copies that last read its pinned host
        images (the GPU may be several steps behind the host: rewriting a pinned table it has not fetched yet would
        hand the earlier step the later step's slab / gradient addresses)."""
        if torch.cuda.is_current_stream_capturing():
            return
        if self.slot_i >= self.EAGER_SLOTS and self.slot is not None and self.slot[2].device == device:
            return                   # a capture slot was selected explicitly (new_param_grad_slot)
        self._ensure_eager_ring(device)
        self.eager_i = (self.eager_i + 1) % self.EAGER_SLOTS
        self.slot, self.slot_i = self.slots[self.eager_i], self.eager_i
        ev = self.events[self.slot_i]
        if ev is not None:
            ev.synchronize()

    def begin(self, check_params=None):
        assert not self.rows, "finish_param_grad_batch() was not called after the previous backward"
        if check_params is not None:
            # the deferred reduction hands autograd gradient buffers that are filled only at finish(): that is sound
            # only when AccumulateGrad STEALS them, i.e. when no .grad exists yet (no gradient accumulation)
            for p in check_params:
                if p.grad is not None:
                    raise _lib.MsdeHipError("begin_param_grad_batch(): a parameter already has a .grad; the batched "
                                            "slab reduction needs zero_grad(set_to_none=True) before every backward")
        self.active = True
        self.used = 0
        self.leaf_first, self.leaf_more = {}, []
        self.prob_used = self.pre_used = 0
        self.rows_used = self.rpre_used = 0
        self.rotated = False

    def alloc(self, nfloats, device):
        nfloats = (nfloats + 3) & ~3
        if self.arena is None or self.arena.device != device or self.used + nfloats > self.arena.numel():
            if self.arena is not None:
                self.retired.append(self.arena)
            size = max(1 << 26, 2 * (self.used + nfloats))
            self.arena = torch.empty(size, dtype=torch.float32, device=device)
            self.used = 0
        view = self.arena[self.used:self.used + nfloats]
        self.used += nfloats
        return view

    def add(self, slab_ptr, splits, n, out, written=False, block=None):
        # only the ADDRESS of the output is kept: an extra reference to the gradient tensor would make autograd's
        # AccumulateGrad clone it (it steals the buffer only when it holds the sole reference) -- a copy of the
        # not yet reduced buffer.  The leaf's .grad keeps the memory alive until the optimiser has used it.
        # written: the kernel that fills these slabs is already queued on the CURRENT stream (not a queued GEMM or a
        # deferred kernel), so reduce_written() may sum them on that stream before the end of the backward pass
        # block = (row_len, slab_ld, out_ld, split_stride): the n = nrows * row_len entries are a column block of the
        # slabs and go to a column block of `out` (include/msde_hip.h: msde_reduce_slabs_multi); None: flat
        self.rows.append((slab_ptr, splits, n, out.data_ptr(), out.device,
                          torch.cuda.current_stream().cuda_stream if written else None, block or (n, n, n, n)))

    def queue_gemm(self, gY, X, M, N, K, has_bias, slab):
        # the stream the operands were produced on rides along: a caller may flush one stream's GEMMs on that stream
        # (launch_gemms(only_stream=...)) while the other stream is still in its backward chain
        self.gemms.append((gY, X, M, N, K, has_bias, slab, torch.cuda.current_stream().cuda_stream))

    def launch_gemms(self, max_wgs=0, only_stream=None):
        """One grouped launch, on the current stream, for the weight-gradient GEMMs queued so far.  May be called
        several times per backward pass (each call takes the next rows of the problem table).  max_wgs > 0 limits the
        launch to that many resident workgroups (a flush that runs beside the backward chain on another stream);
        only_stream: take only the GEMMs whose operands were produced on that stream (handle), leave the rest queued."""
        if only_stream is not None:
            mine = [t for t in self.gemms if t[7] == only_stream]
            rest = [t for t in self.gemms if t[7] != only_stream]
            if not mine:
                return
            self.gemms = mine
            try:
                self.launch_gemms(max_wgs)
            finally:
                self.gemms = rest + self.gemms
            return
        if not self.gemms:
            return
        dev = self.gemms[0][0].device
        self._select_slot(dev)
        host_prob, host_ppre, dev_prob, dev_ppre = self.slot[4:]
        lib = _lib.load()
        # longest workgroups first (rows per split = K tiles per workgroup): the launch ends with short workgroups
        # instead of draining a few 100-us ones at partial occupancy
        def rows_per_split(t):
            sp = _SPLITS.get((t[2], t[3], t[4])) or int(lib.msde_linear_bwd_w_splits(t[2], t[3], t[4]))
            return t[2] / max(sp, 1)
        self.gemms.sort(key=rows_per_split, reverse=True)
        r0, q0, ng = self.prob_used, self.pre_used, len(self.gemms)
        assert r0 + ng <= self.MAX_ROWS
        hp2 = host_ppre.numpy()
        total_b = 0
        for r, (gY, X, M, N, K, hb, slab, _st) in enumerate(self.gemms):
            nb = lib.msde_linear_bwd_w_describe_ld(_p(gY), _row_stride(gY, N), _p(X), _row_stride(X, K), M, N, K, hb,
                                                   _p(slab), _p(bound_tensor(M)), ctypes.c_void_p(host_prob[r0 + r].data_ptr()))
            if nb <= 0:
                raise _lib.MsdeHipError(f"msde_linear_bwd_w_describe failed ({nb}) for {M}x{N}x{K}")
            hp2[q0 + r] = total_b
            total_b += nb
        hp2[q0 + ng] = total_b
        upload_table(dev_prob[r0:r0 + ng], host_prob[r0:r0 + ng])
        upload_table(dev_ppre[q0:q0 + ng + 1], host_ppre[q0:q0 + ng + 1])
        self.prob_used, self.pre_used = r0 + ng, q0 + ng + 1
        _lib.call("msde_linear_bwd_w_grouped_ex", ctypes.c_void_p(dev_prob[r0].data_ptr()),
                  ctypes.c_void_p(dev_ppre[q0:].data_ptr()), ng, total_b, int(max_wgs), _stream())
        # the operands stay referenced until finish(): a flush may run on ANOTHER stream than the one that allocated
        # them, and the caching allocator would otherwise hand their memory to the allocating stream's next kernels
        self.launched.extend(self.gemms)
        self.gemms = []

    def run_deferred(self):
        d, self.deferred = self.deferred, []
        groups = {}
        for fn in d:
            t = getattr(fn, "gin_tab", None)
            if t is not None:       # the GIN bond-table gradients of one graph: ONE launch for all layers (below)
                groups.setdefault((t[3].data_ptr(), t[4].src.data_ptr(), t[6], t[7], t[8], t[9]), []).append(t)
            else:
                fn(None)
        for ts in groups.values():
            for i in range(0, len(ts), 8):
                part = ts[i:i + 8]
                n = len(part)
                arr = lambda k: (ctypes.c_void_p * n)(*[t[k].data_ptr() for t in part])
                _, _, _, codes, plan, _, N, E_, D, R = part[0]
                _lib.call("msde_gin_aggregate_bwd_tab_multi", ctypes.cast(arr(0), ctypes.c_void_p), ctypes.cast(arr(1), ctypes.c_void_p),
                          ctypes.cast(arr(2), ctypes.c_void_p), ctypes.cast(arr(5), ctypes.c_void_p), n, _p(codes), _p(plan.src),
                          _p(plan.dst), N, E_, D, R, _p(bound_tensor(N)), _p(bound_tensor(E_)), _stream())
        # the closures hold the kernels' operands: kept until finish(), because they may be launched on ANOTHER stream than
        # the one that allocated the operands (the caching allocator would hand their memory to that stream's next kernels)
        self.launched.extend(d)

    def park(self):
        """Set the GEMMs queued so far aside (returned as an opaque group) instead of launching them: the caller launches
        the group later with launch_group(), e.g. on another stream once that stream is free."""
        g, self.gemms = self.gemms, []
        return g

    def launch_group(self, group, max_wgs=0):
        rest, self.gemms = self.gemms, group
        try:
            self.launch_gemms(max_wgs)
        finally:
            self.gemms = rest

    def partition(self, bucket_of, nb):
        """Data-parallel tail (pretrain.Trainer): cut what finish() would launch -- the queued weight-gradient
        GEMMs and the slab rows still to be summed -- into `nb` parts by the gradient bucket (= model) their RESULT belongs
        to, so that finish_part(b) can run bucket by bucket and bucket b's all-reduce travels while bucket b + 1's weight
        gradients are computed.  bucket_of(address of a result) -> bucket or None (unknown: the last part).  Same kernels on the
        same tiles and rows in another grouping: results are bit-identical to finish()."""
        assert self.active and not self.leaf_more, "partitioned finish: no multi-contribution parameters (MD17 path)"
        self.run_deferred()
        rows, self.rows = self.rows, []
        gemms, self.gemms = self.gemms, []
        parts = [([], []) for _ in range(nb)]
        where = []
        for r in rows:
            b = bucket_of(r[3])
            b = nb - 1 if b is None else b
            parts[b][1].append(r)
            where.append((r[0], b))
        where.sort()
        import bisect
        keys = [w[0] for w in where]
        for t in gemms:
            lo, hi = t[6].data_ptr(), t[6].data_ptr() + 4 * t[6].numel()
            i = bisect.bisect_left(keys, lo)
            b = where[i][1] if i < len(where) and where[i][0] < hi else nb - 1
            parts[b][0].append(t)
        self._parts = parts

    def finish_part(self, b):
        """One grouped launch + one slab reduction for part b of partition() (current stream)."""
        gemms, rows = self._parts[b]
        self._parts[b] = ([], [])
        if rows:
            self._select_slot(rows[0][4])
        if gemms:
            self.gemms = gemms
            self.launch_gemms()
        if rows:
            self._reduce(rows)

    def finish(self):
        self.active = False
        self._parts = None
        self.run_deferred()          # nobody ran them on another stream: here, before their slabs are summed
        rows = self.rows
        if not rows and not self.rotated:
            return
        if rows:
            self._select_slot(rows[0][4])
            self.launch_gemms()      # the (still) queued weight-gradient GEMMs as one grouped launch
            self._reduce(rows)
            # parameters with several queued contributions (weight_grad_leaf): first + later ones, one more pass of the same
            # kernel per level -- a two-"slab" row whose slabs are the two gradient buffers themselves
            level = 1
            while True:
                extra = [(first, t) for first, t, lv in self.leaf_more if lv == level]
                if not extra:
                    break
                rows2 = []
                for first, t in extra:
                    lo, hi = min(first, t.data_ptr()), max(first, t.data_ptr())
                    n = t.numel()
                    rows2.append((lo, 2, n, first, t.device, None, (n, n, n, (hi - lo) // 4)))
                self._reduce(rows2)
                level += 1
            self.leaf_more = []
        if not torch.cuda.is_current_stream_capturing():
            ev = self.events[self.slot_i] or torch.cuda.Event()
            ev.record()
            self.events[self.slot_i] = ev
        self.rows = []
        self.launched = []
        _retire(self.retired)
        self.retired = []

    def _reduce(self, rows):
        """One msde_reduce_slabs_multi launch on the current stream for `rows` (the next rows of the slot's tables)."""
        host_rows, host_pre, dev_rows, dev_pre = self.slot[:4]
        r0, q0, k = self.rows_used, self.rpre_used, len(rows)
        assert r0 + k <= self.MAX_ROWS and q0 + k + 1 <= host_pre.numel(), "slab row tables full"
        hr, hp = host_rows.numpy(), host_pre.numpy()
        total = 0
        for r, row in enumerate(rows):
            hr[r0 + r, 0], hr[r0 + r, 1], hr[r0 + r, 2], hr[r0 + r, 3] = row[0], row[1], row[2], row[3]
            hr[r0 + r, 4], hr[r0 + r, 5], hr[r0 + r, 6], hr[r0 + r, 7] = row[6]
            hp[q0 + r] = total
            total += (row[2] + 63) // 64 if row[1] >= _lib.REDUCE_LONG else (row[2] + 255) // 256
        hp[q0 + k] = total
        upload_table(dev_rows[r0:r0 + k], host_rows[r0:r0 + k])
        upload_table(dev_pre[q0:q0 + k + 1], host_pre[q0:q0 + k + 1])
        self.rows_used, self.rpre_used = r0 + k, q0 + k + 1
        _lib.call("msde_reduce_slabs_multi", ctypes.c_void_p(dev_rows[r0:].data_ptr()),
                  ctypes.c_void_p(dev_pre[q0:].data_ptr()), k, total, _stream())

    def reduce_written(self):
        """Sum, on the CURRENT stream, the slabs whose producing kernels are already queued on it (add(written=True)):
        a stream that finishes its part of the backward early reduces its own slabs while the other stream is still in
        the backward chain, and the final reduction has that much less to read."""
        if not self.active or not self.rows:
            return 0
        me = torch.cuda.current_stream().cuda_stream
        mine = [r for r in self.rows if r[5] is not None and r[5] == me]
        if not mine:
            return 0
        self._select_slot(mine[0][4])
        self._reduce(mine)
        self.rows = [r for r in self.rows if not (r[5] is not None and r[5] == me)]
        return len(mine)

    def _select_slot(self, dev):
        """Once per backward pass: the table slot its uploads go to (eager ring, or the capture's own slot)."""
        if not self.rotated:
            self._rotate_eager(dev)
            self.rotated = True
        if self.slot is None or self.slot[2].device != dev:
            self.new_slot(dev)


_SLABS = _SlabBatch()
_SPLITS = {}


def begin_param_grad_batch(params=None):
    """params: the parameters of the step (optional); each must have .grad None (see _SlabBatch.begin)."""
    _SLABS.begin(params)


def flush_wgrad_gemms(max_wgs=0, only_stream=None):
    """Launch the weight-gradient GEMMs queued so far as one grouped kernel on the current stream (their slabs are
    still summed by finish_param_grad_batch, whose stream must by then be ordered after this one).  only_stream: only
    those whose operands were produced on that stream."""
    if _SLABS.active:
        _SLABS.launch_gemms(max_wgs, only_stream)


def reduce_written_slabs():
    """Sum the slabs written by kernels already queued on the current stream (see _SlabBatch.reduce_written); returns
    the number of gradient tensors reduced.  finish_param_grad_batch()'s stream must be ordered after this one."""
    return _SLABS.reduce_written()


def run_deferred_leaf_kernels():
    """Launch, on the current stream, the leaf-only kernels the backward functions queued (see DEFER_LEAF_KERNELS)."""
    _SLABS.run_deferred()


def have_deferred_leaf_kernels():
    return bool(_SLABS.deferred)


def park_wgrad_gemms():
    """Set the weight-gradient GEMMs queued so far aside; launch them later with launch_wgrad_group (any stream that is
    ordered after their operands)."""
    return _SLABS.park() if _SLABS.active else []


def launch_wgrad_group(group, max_wgs=0):
    if group:
        _SLABS.launch_group(group, max_wgs)


def partition_param_grad_batch(bucket_of, nb):
    """See _SlabBatch.partition: then finish_param_grad_part(b) per bucket, finish_param_grad_batch() at the end."""
    _SLABS.partition(bucket_of, nb)


def finish_param_grad_part(b):
    _SLABS.finish_part(b)


def finish_param_grad_batch():
    """Sum all queued slabs (one launch on the current stream, which must already be ordered after every stream
    that ran part of the backward -- loss.backward() returns in that state)."""
    _SLABS.finish()


def new_param_grad_slot(device):
    """Before a hipGraph capture: give the graph its own table slot (see _SlabBatch.new_slot)."""
    _SLABS._ensure_eager_ring(device)    # slots 0..EAGER_SLOTS-1 stay the eager ring
    _SLABS.new_slot(device)
    note_capture()


def use_eager_param_grad_slot():
    """After a capture: eager steps must not overwrite the host tables a captured graph re-reads."""
    if _SLABS.slots:
        _SLABS.slot, _SLABS.slot_i = _SLABS.slots[_SLABS.eager_i], _SLABS.eager_i


def _row_stride(t, cols):
    """Row stride of a 2-D operand with unit column stride (a column block of a wider buffer is fine)."""
    return int(t.stride(0)) if t.size(0) > 1 else max(int(t.stride(0)), cols)


def weight_grad(g2, x2, has_bias, deferrable=True, out_w=None, out_b=None):
    """gW [N,K] = g2^T x2 and (has_bias) gb [N] = column sums of g2 for g2 [M,N], x2 [M,K]: the hand-written
    split-M kernel -- queued for the grouped launch + batched slab reduction when a parameter-gradient batch is
    open and the results are `deferrable` -- or the per-layer launch of the same kernel with its own slab reduction.
    Under the grouped launch the operands may be column blocks of wider buffers (unit column stride)."""
    M, N = g2.shape
    K = x2.size(1)
    if not (_SLABS.active and deferrable) or M < WGRAD_HIP_MIN_ROWS:
        g2, x2 = g2.contiguous(), x2.contiguous()          # only the grouped kernel takes row strides
    st = _stream()
    # out_w / out_b: contiguous slices of a stacked gradient (several layers' weights consumed as one operand)
    gw = out_w if out_w is not None else torch.empty(N, K, dtype=torch.float32, device=g2.device)
    gb = (out_b if out_b is not None else torch.empty(N, dtype=torch.float32, device=g2.device)) if has_bias else None
    if _SLABS.active and deferrable:
        splits = _SPLITS.get((M, N, K))
        if splits is None:
            splits = _SPLITS[(M, N, K)] = int(_lib.load().msde_linear_bwd_w_splits(M, N, K))
        slab = _SLABS.alloc(splits * (N * K + (N if has_bias else 0)), g2.device)
        _SLABS.queue_gemm(g2, x2, M, N, K, int(has_bias), slab)       # one grouped launch at the end of the backward pass
        _SLABS.add(slab.data_ptr(), splits, N * K, gw)
        if has_bias:
            _SLABS.add(slab.data_ptr() + 4 * splits * N * K, splits, N, gb)
    else:
        ws = _wgrad_workspace(M, N, K, g2.device)
        _lib.call("msde_linear_bwd_w", _p(g2), _p(x2), M, N, K, _p(gw), _p(gb), _p(ws), _p(bound_tensor(M)), st)
    return gw, gb


def weight_grad_leaf(g2, x2, has_bias, W, b_key=None):
    """weight_grad for a LEAF parameter W that may receive several contributions in one backward pass (the twice-
    differentiated force path of finetune_MD17.py:68-78: every weight is used by the energy AND by d(energy)/d(positions)).
    With a parameter-gradient batch open each contribution is a problem of the grouped launch; the FIRST one's result
    tensors are returned (autograd's AccumulateGrad steals them), later ones return (None, None) and are added to the first
    behind the batched slab reduction (_SlabBatch.finish) -- no per-layer GEMM, slab-reduction or add launch.  b_key: the
    bias parameter's address (has_bias).  Without an open batch: the immediate per-layer launch."""
    if not (_SLABS.active and W.is_leaf):
        return weight_grad(g2, x2, has_bias, deferrable=False)
    if not (g2.dtype == torch.float32 and x2.dtype == torch.float32 and g2.stride(-1) == 1 and x2.stride(-1) == 1):
        g2, x2 = _f32(g2), _f32(x2)
    keys = [W.data_ptr()] + ([b_key] if has_bias else [])
    levels = [len([1 for f, _, _ in _SLABS.leaf_more if f == _SLABS.leaf_first[k][0]]) + 1 if k in _SLABS.leaf_first else 0
              for k in keys]
    gw, gb = weight_grad(g2, x2, has_bias, deferrable=True)
    outs = []
    for k, lv, t in zip(keys, levels, (gw, gb)):
        if lv == 0:
            _SLABS.leaf_first[k] = (t.data_ptr(), t.numel())
            outs.append(t)
        else:
            first, n = _SLABS.leaf_first[k]
            assert n == t.numel()
            _SLABS.leaf_more.append((first, t, lv))       # (kept alive here until finish())
            outs.append(None)
    return outs[0], (outs[1] if has_bias else None)


def colsum(x):
    """Column sums of a contiguous [M, C] tensor (fixed summation order)."""
    x = _f32(x)
    M, C = x.shape
    out = torch.empty(C, dtype=torch.float32, device=x.device)
    _lib.call("msde_colsum", _p(x), M, C, _p(out), _p(_bn_workspace(M, C, x.device)), _p(bound_tensor(M)), _stream())
    return out


def colsum_leaf(x):
    """colsum(x) as a PARAMETER gradient nothing in the backward chain reads (a bias gradient whose product has no weight
    operand of its own): inside an open parameter-gradient batch the launch is queued with the deferred leaf kernels (they
    run beside the grouped weight-gradient launch, off the backward chain) and only the result's address goes to autograd."""
    if not _SLABS.active:
        return colsum(x)
    x = _f32(x)
    M, C = x.shape
    out = torch.empty(C, dtype=torch.float32, device=x.device)
    ws = _bn_workspace(M, C, x.device)

    def launch(st_=None, x=x, out=out, ws=ws):
        _lib.call("msde_colsum", _p(x), M, C, _p(out), _p(ws), _p(bound_tensor(M)), st_ if st_ is not None else _stream())
    _SLABS.deferred.append(launch)
    return out


def weight_grad_blocks(g2, x2, has_bias, blocks, deferrable=True):
    """gW [N,K] = g2^T x2 whose COLUMN BLOCKS belong to wider / permuted parameter gradients: blocks = [(k0, kn, out,
    out_col0)] sends columns k0 .. k0+kn to out[:, out_col0 .. +kn] (out [N, *] contiguous).  Returns the bias gradient
    (or None).  Queued like weight_grad: the batched slab reduction writes the blocks in place (2-D reduce rows);
    without an open parameter-gradient batch the product is formed on the spot and copied."""
    M, N = g2.shape
    K = x2.size(1)
    if _SLABS.active and deferrable:
        splits = _SPLITS.get((M, N, K))
        if splits is None:
            splits = _SPLITS[(M, N, K)] = int(_lib.load().msde_linear_bwd_w_splits(M, N, K))
        slab = _SLABS.alloc(splits * (N * K + (N if has_bias else 0)), g2.device)
        _SLABS.queue_gemm(g2, x2, M, N, K, int(has_bias), slab)
        for k0, kn, out, c0 in blocks:
            _SLABS.add(slab.data_ptr() + 4 * k0, splits, N * kn, out[:, c0:], block=(kn, K, out.size(1), N * K))
        gb = None
        if has_bias:
            gb = torch.empty(N, dtype=torch.float32, device=g2.device)
            _SLABS.add(slab.data_ptr() + 4 * splits * N * K, splits, N, gb)
        return gb
    gw, gb = weight_grad(g2, x2, has_bias, False)
    for k0, kn, out, c0 in blocks:
        out[:, c0:c0 + kn].copy_(gw[:, k0:k0 + kn])
    return gb
