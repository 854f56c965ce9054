"""Flat-buffer Adam on the HIP kernel `msde_adam_flat` + gradient flattening for data parallelism.

Mirrors torch.optim.Adam(param_groups, lr, weight_decay) as used by
examples/pretrain_MoleculeSDE.py:331-337: one lr per model (param group), betas (0.9, 0.999),
eps 1e-8.  All trainable parameters are re-pointed into ONE contiguous fp32 buffer ordered
GIN, SchNet, 2D->3D, 3D->2D, so that (a) the optimiser is a single kernel launch and (b) the
data-parallel gradient exchange is a single RCCL all-reduce of one message.

Difference to torch.optim.Adam, by design: a parameter that received no gradient in a step is
treated as having a zero gradient (torch skips it).  Every parameter on the pretrain path
receives a gradient, so the two coincide there.
"""
import torch

from . import hip, slabs, wcache


class FlatAdam:
    EAGER_SLOTS = 3

    def __init__(self, groups, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        """groups: list of {"params": iterable of Parameters, "lr": float}."""
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        self.params, seg_end, seg_lr = [], [], []
        seen = set()
        total = 0
        offsets = []
        for g in groups:
            for p in g["params"]:
                if not p.requires_grad or id(p) in seen:
                    continue
                seen.add(id(p))
                self.params.append(p)
                total = (total + 3) & ~3           # 16-byte aligned start: kernels read parameters with float4 loads
                offsets.append(total)
                total += p.numel()
            total = (total + 3) & ~3
            seg_end.append(total)
            seg_lr.append(float(g["lr"]))
        dev = self.params[0].device
        self.n = total
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)   # alignment gaps stay zero
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.m = torch.zeros(total, dtype=torch.float32, device=dev)
        self.v = torch.zeros(total, dtype=torch.float32, device=dev)
        self.offsets, self.sizes = [], []
        self._chunk = hip.chunk_elems()
        self._grad_keep = []
        with torch.no_grad():
            for p, off in zip(self.params, offsets):
                n = p.numel()
                view = self.flat_p[off:off + n].view_as(p)
                view.copy_(p.data)
                p.data = view                       # parameters now alias the flat buffer
                self.offsets.append(off)
                self.sizes.append(n)
        import numpy as np
        ch = self._chunk
        self._chunks_per_param = [(n + ch - 1) // ch for n in self.sizes]
        self.n_chunks = sum(self._chunks_per_param)
        self._chunk_bytes = np.arange(max(self._chunks_per_param), dtype=np.int64) * (4 * ch)
        self._chunk_off = torch.tensor([o + c * ch for o, nc in zip(self.offsets, self._chunks_per_param) for c in range(nc)],
                                       dtype=torch.int64)
        self._chunk_cnt = torch.tensor([min(ch, n - c * ch) for n, nc in zip(self.sizes, self._chunks_per_param)
                                        for c in range(nc)], dtype=torch.int64)
        # Slots 0..EAGER_SLOTS-1 form the eager ring (a pinned host image is rewritten only after the event recorded
        # behind its last upload has completed, so a GPU that runs several steps behind the host never sees the next
        # step's addresses); every captured hipGraph gets one more slot of its own.
        self._slots, self._events, self._eager_i = [], [], 0
        for _ in range(self.EAGER_SLOTS):
            self.new_table_slot()
        self._slot, self._slot_i = self._slots[0], 0
        # chunk-table rows of every param group (= data-parallel bucket): groups are laid out one after the other
        self._bucket_chunk_rows, r, k = [], 0, 0
        for end in seg_end:
            r0 = r
            while k < len(self.params) and self.offsets[k] < end:
                r += self._chunks_per_param[k]
                k += 1
            self._bucket_chunk_rows.append((r0, r))
        self.seg_end = torch.tensor(seg_end, dtype=torch.int64, device=dev)
        self.seg_lr = torch.tensor(seg_lr, dtype=torch.float32, device=dev)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        # data-parallel buckets = the param groups (one model each), as [begin, end) ranges of the flat buffers
        self.bucket_ranges = [(0 if i == 0 else seg_end[i - 1], seg_end[i]) for i in range(len(seg_end))]
        self._bucket_seg = [(torch.tensor([b - a], dtype=torch.int64, device=dev),
                             torch.tensor([seg_lr[i]], dtype=torch.float32, device=dev))
                            for i, (a, b) in enumerate(self.bucket_ranges)]

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def new_table_slot(self):
        """Fresh pinned host image + device copy of the chunk table.  A captured hipGraph re-reads ITS host
        image at every replay (the upload is a memcpy node), so each capture gets its own slot; call this
        before entering the capture (no allocation happens inside it)."""
        host = torch.empty(self.n_chunks, 3, dtype=torch.int64).pin_memory()
        host[:, 0] = 0
        host[:, 1] = self._chunk_off
        host[:, 2] = self._chunk_cnt
        self._slot = (host, torch.empty(self.n_chunks, 3, dtype=torch.int64, device=self.flat_p.device))
        self._slots.append(self._slot)
        self._events.append(None)
        self._slot_i = len(self._slots) - 1

    def use_eager_slot(self):
        """After a capture: eager steps go back to slot 0 so that they never overwrite the host image a captured
        graph re-reads at every replay."""
        self._slot, self._slot_i = self._slots[self._eager_i], self._eager_i

    def _chunk_table(self):
        """Device table of (gradient chunk address, flat offset, count) for the current .grad tensors."""
        capturing = torch.cuda.is_current_stream_capturing() if self.flat_p.is_cuda else False
        if not capturing and self._slot_i < self.EAGER_SLOTS:
            # eager: next slot of the ring; wait until the copy that last read its pinned image is done
            self._eager_i = (self._eager_i + 1) % self.EAGER_SLOTS
            self._slot, self._slot_i = self._slots[self._eager_i], self._eager_i
            ev = self._events[self._slot_i]
            if ev is not None:
                ev.synchronize()
        host, dev = self._slot
        col = host[:, 0].numpy()
        self._grad_keep = []
        r = 0
        for p, n, nc in zip(self.params, self.sizes, self._chunks_per_param):
            g = p.grad
            if g is not None and (g.dtype != torch.float32 or not g.is_contiguous()):
                g = g.contiguous().float()
                self._grad_keep.append(g)
            if g is None:
                col[r:r + nc] = 0
            else:
                col[r:r + nc] = g.data_ptr() + self._chunk_bytes[:nc]
            r += nc
        slabs.upload_table(dev, host)        # recorded (not captured) while a hipGraph is being captured
        if not capturing and self.flat_p.is_cuda:
            ev = self._events[self._slot_i] or torch.cuda.Event()
            ev.record()
            self._events[self._slot_i] = ev
        return dev, self.n_chunks

    def gather_grads(self):
        """Copy the per-parameter .grad tensors into the flat gradient buffer (one kernel, chunk table);
        parameters without a gradient contribute zeros.  Only the data-parallel path needs the flat copy."""
        table, n = self._chunk_table()
        hip.gather_chunks(table, n, self.flat_g)
        return self.flat_g

    def grad_table(self):
        """The chunk table for the current .grad tensors (built and uploaded once per step): for gather_bucket()."""
        return self._chunk_table()[0]

    def gather_bucket(self, i, table):
        """gather_grads() for bucket i (= param group i) only: its rows of `table` (grad_table()) into its range of the flat
        gradient buffer -- the data-parallel tail flattens, reduces and applies bucket by bucket."""
        r0, r1 = self._bucket_chunk_rows[i]
        if r1 > r0:
            hip.gather_chunks(table[r0:r1], r1 - r0, self.flat_g)

    def grad_bucket_lookup(self):
        """address -> bucket of the parameter whose .grad holds that address (None: no such parameter): how the batched
        weight-gradient machinery learns which bucket a result belongs to (slabs.partition_param_grad_batch)."""
        import bisect
        iv = []
        ends = [e for _, e in self.bucket_ranges]
        for p, off in zip(self.params, self.offsets):
            g = p.grad
            if g is not None:
                b = bisect.bisect_right(ends, off)
                iv.append((g.data_ptr(), g.data_ptr() + 4 * g.numel(), b))
        iv.sort()
        starts = [t[0] for t in iv]

        def look(addr):
            i = bisect.bisect_right(starts, addr) - 1
            return iv[i][2] if i >= 0 and addr < iv[i][1] else None
        return look

    def step(self, grad_scale=1.0):
        """Adam on the flat gradient buffer: assumes gather_grads() (and, under DP, the all-reduce of
        flat_g) already happened."""
        self.step_dev.add_(1)
        wcache.bump_weight_epoch()            # parameters change through raw pointers: transposed weight copies go stale
        hip.adam_flat(self.flat_p, self.flat_g, self.m, self.v, self.step_dev, self.seg_end, self.seg_lr,
                      self.betas[0], self.betas[1], self.eps, self.weight_decay, grad_scale)

    def begin_bucket_step(self):
        """Bucketed form of step(): call once, then step_bucket(i) for every bucket (any order, each once)."""
        self.step_dev.add_(1)
        wcache.bump_weight_epoch()

    def step_bucket(self, i, grad_scale=1.0):
        """Adam on bucket i (= param group i) of the flat buffers only: lets the optimiser start on a bucket whose
        all-reduce has finished while the next bucket is still on the wire."""
        a, b = self.bucket_ranges[i]
        if b <= a:
            return
        seg_end, seg_lr = self._bucket_seg[i]
        hip.adam_flat(self.flat_p[a:b], self.flat_g[a:b], self.m[a:b], self.v[a:b], self.step_dev, seg_end, seg_lr,
                      self.betas[0], self.betas[1], self.eps, self.weight_decay, grad_scale)

    def step_from_grads(self, grad_scale=1.0, bump=True):
        """Single-GPU step: Adam reads each parameter's .grad in place through the chunk table (no flattening).  bump=False:
        the caller already advanced step_dev for this step (pretrain.Trainer: with its own counter, at the head of the step)."""
        table, n = self._chunk_table()
        if bump:
            self.step_dev.add_(1)
        wcache.bump_weight_epoch()
        hip.adam_chunks(self.flat_p, table, n, self.m, self.v, self.step_dev, self.seg_end, self.seg_lr,
                        self.betas[0], self.betas[1], self.eps, self.weight_decay, grad_scale)
