"""Capacity buckets: ONE set of static device buffers (and one captured hipGraph) for every mini-batch.

A `Bucket` owns, for fixed row capacities (atoms N, bonds E_b, extended edges E_e, radius edges E_r, atom pairs P):
  * the RAW input blob of a batch -- what a data loader's collate produces (atom feature codes, coordinates, the bond
    list with its features, atoms / bonds per molecule) -- as one int32 device buffer, so feeding a batch is ONE copy;
  * every plan buffer of moleculesde_amd.plan (CSR views, feature codes, embedding row lists, dense-head pair layout),
    filled ON THE DEVICE from the raw blob by csrc/plan.hip (`build_plan_on_device`), including the extended graph
    (`extend_graph`, Geom3D/datasets/dataset_3D.py:12-35) the reference computes per sample on the CPU;
  * a Batch-like object (`bucket.batch`) the model classes accept unchanged.
True sizes stay on the device (`sizes`): kernels that reduce over rows read them through the row bounds
(msde_set_row_bound), everything else processes the padded rows too (finite values, zero gradients).
"""
import ctypes
import types

import numpy as np
import torch

from . import _lib, hip, plan as _plan
from .batch import Batch

K_ATOM = 9
MAX_NBR = 32


def _round_up(v, q):
    return ((int(v) + q - 1) // q) * q


PLAN_NMAX = 32           # atoms per molecule csrc/plan.hip builds (PL_NMAX: 32-bit neighbourhood masks)


class BucketOverflow(RuntimeError):
    """A batch does not fit the capacities of its bucket (or holds a molecule csrc/plan.hip cannot build)."""


class Caps:
    """Row capacities of a bucket.  Pairwise distinct (the row-bound table is keyed by capacity) and different from every
    other row count a step allocates (B, B * n_max, the `reserved` numbers): a tensor whose row count happened to equal a
    capacity would be clamped to the bucket's valid rows."""

    __slots__ = ("B", "N", "E_b", "E_e", "E_r", "P", "n_max")

    def __init__(self, B, N, E_b, E_e, E_r, P, n_max, reserved=()):
        self.B, self.n_max = int(B), int(n_max)
        if self.n_max > PLAN_NMAX:
            raise BucketOverflow(f"molecules of up to {self.n_max} atoms: the device-side plan builder takes <= {PLAN_NMAX} "
                                 "(use the exact-size path: prepare_batch + Trainer.step)")
        # fine granularity: every padded row is processed by the row-wise kernels, so padding costs time one for one
        vals = [_round_up(N, 64), _round_up(E_b, 64) + 8, _round_up(E_e, 256) + 16, _round_up(E_r, 256) + 24,
                _round_up(P, 256) + 40]
        taken = {self.B, self.B * self.n_max, self.B + 1, self.B + 2} | {int(r) for r in reserved}
        for i, q in enumerate((64, 64, 256, 256, 256)):
            # (2 * E_e is a row count too: tensors with two rows per extended edge)
            while vals[i] in taken or vals[i] in vals[:i] or (i > 2 and vals[i] == 2 * vals[2]) or \
                    (i == 2 and 2 * vals[i] in taken | set(vals[:i])):
                vals[i] += q
        assert len(set(vals + [2 * vals[2]])) == 6 and not (set(vals + [2 * vals[2]]) & taken)
        self.N, self.E_b, self.E_e, self.E_r, self.P = vals

    def fits(self, need):
        return (need["B"] == self.B and need["N"] <= self.N and need["E_b"] <= self.E_b and need["E_e"] <= self.E_e and
                need["E_r"] <= self.E_r and need["P"] <= self.P and need["n_max"] <= self.n_max)

    @staticmethod
    def covering(needs, n_max=None):
        """Smallest capacities that hold every entry of `needs` (dicts from raw_sizes)."""
        mx = lambda k: max(n[k] for n in needs)
        return Caps(needs[0]["B"], mx("N"), mx("E_b"), mx("E_e"), mx("E_r"), mx("P"), n_max or max(mx("n_max"), 1))

    def as_dict(self):
        return {k: getattr(self, k) for k in self.__slots__}


# ------------------------------------------------------------------------------------------------ raw batches
def raw_layout(caps):
    """Offsets (in int32 words) of the fields of the raw blob."""
    o, lay = 0, {}
    for name, n in (("x", caps.N * K_ATOM), ("pos", caps.N * 3), ("bond_src", caps.E_b), ("bond_dst", caps.E_b),
                    ("bond_attr", caps.E_b * 3), ("mol_atoms", caps.B), ("mol_bonds", caps.B)):
        lay[name] = (o, n)
        o += (n + 3) & ~3
    lay["_total"] = o
    return lay


def raw_sizes(b):
    """What a batch needs from a bucket (host side, from the collated Batch)."""
    cnt = torch.bincount(b.batch, minlength=b.num_graphs)
    n = cnt.numpy().astype(np.int64)
    E_e = int(b.extended_edge_index.size(1)) if getattr(b, "extended_edge_index", None) is not None else int((n * (n - 1)).sum())
    return {"B": int(b.num_graphs), "N": int(b.x.size(0)), "E_b": int(b.edge_index.size(1)), "E_e": E_e,
            "E_r": int((n * np.minimum(np.maximum(n - 1, 0), MAX_NBR)).sum()), "P": int((n * n).sum()),
            "n_max": int(n.max()) if len(n) else 0}


def pack_raw(b, caps, pin=False):
    """Raw blob (int32, host) of a collated Batch for a bucket of capacities `caps`: exactly the arrays a loader's
    collate holds (no plan, no extended edges)."""
    need = raw_sizes(b)
    if not caps.fits(need) or need["n_max"] > PLAN_NMAX:
        # an oversize field would silently overwrite the next field of the blob: refuse here, on the host, for free
        raise BucketOverflow(f"batch {need} does not fit the bucket {caps.as_dict()}")
    lay = raw_layout(caps)
    blob = torch.zeros(lay["_total"], dtype=torch.int32)
    if pin:
        blob = blob.pin_memory()
    N, Eb = b.x.size(0), b.edge_index.size(1)
    assert b.x.dim() == 2 and b.x.size(1) == K_ATOM

    def put(name, t):
        o, _ = lay[name]
        flat = t.reshape(-1)
        blob[o:o + flat.numel()] = flat

    put("x", b.x.to(torch.int32))
    o, _ = lay["pos"]
    blob[o:o + 3 * N] = b.positions.float().contiguous().view(-1).view(torch.int32)
    put("bond_src", b.edge_index[0].to(torch.int32))
    put("bond_dst", b.edge_index[1].to(torch.int32))
    put("bond_attr", b.edge_attr.to(torch.int32))
    cnt = torch.bincount(b.batch, minlength=b.num_graphs)
    put("mol_atoms", cnt.to(torch.int32))
    put("mol_bonds", torch.bincount(b.batch[b.edge_index[0]], minlength=b.num_graphs).to(torch.int32) if Eb else
        torch.zeros(b.num_graphs, dtype=torch.int32))
    return blob


# ------------------------------------------------------------------------------------------------ the bucket
class Bucket:
    def __init__(self, caps, device, atom_dims=None, bond_dims=None, node_class=119):
        self.caps, self.device = caps, device
        atom_dims = list(atom_dims or _plan.ATOM_FEATURE_DIMS)
        bond_dims = list(bond_dims or _plan.BOND_FEATURE_DIMS)
        c = caps
        i32 = lambda *s: torch.zeros(*s, dtype=torch.int32, device=device)
        self.layout = raw_layout(c)
        self.raw = i32(self.layout["_total"])
        v = lambda name: self.raw[self.layout[name][0]:self.layout[name][0] + self.layout[name][1]]
        self.x_raw = v("x").view(c.N, K_ATOM)
        self.positions = v("pos").view(torch.float32).view(c.N, 3)
        self.bond_src_raw, self.bond_dst_raw, self.bond_attr_raw = v("bond_src"), v("bond_dst"), v("bond_attr").view(c.E_b, 3)
        self.mol_atoms, self.mol_bonds = v("mol_atoms"), v("mol_bonds")
        self.atom_off = torch.tensor(_plan._offsets(atom_dims), dtype=torch.int32, device=device)
        self.bond_off = torch.tensor(_plan._offsets(bond_dims), dtype=torch.int32, device=device)
        self.sizes, self.err = i32(8), i32(1)
        self.mol_ptr_full, self.bond_ptr, self.pair_ptr = i32(c.B + 2), i32(c.B + 1), i32(c.B + 1)
        self.ext_rows, self.ext_cnt, self.ext_ptr = i32(c.N), i32(c.B), i32(c.B + 1)
        pl = types.SimpleNamespace()
        pl.N, pl.B, pl.N_max, pl.max_nbr, pl.E_r_cap = c.N, c.B, c.n_max, MAX_NBR, c.E_r
        pl.err_dev = self.err             # kernels that can detect a capacity overflow of their own raise the bucket's flag
        pl.mol_ptr = self.mol_ptr_full[:c.B + 1]
        pl.batch_i32 = i32(c.N)
        pl.atom_codes, pl.atom_R = i32(c.N, K_ATOM), sum(atom_dims)
        pl.z_codes, pl.z_list = i32(c.N, 1), None
        pl.atom_list_ptr, pl.atom_list_nodes = i32(pl.atom_R + 1), i32(c.N * K_ATOM)
        self._cnt_scratch = i32(max(pl.atom_R, node_class) + 1)
        self.node_class = node_class
        pl.z_list = (node_class, i32(node_class + 1), i32(c.N))

        def csr(E):
            p = hip.CsrPlan()
            p.N, p.E = c.N, E
            p.rowptr, p.src, p.dst = i32(c.N + 1), i32(E), i32(E)
            p.rowptr_s, p.perm_s, p.perm_t, p.E_dev = i32(c.N + 1), i32(E), None, None
            return p
        pl.bond, pl.ext = csr(c.E_b), csr(c.E_e)
        pl.bond.E_dev, pl.ext.E_dev = self.sizes[1:2], self.sizes[2:3]
        pl.bond_codes, pl.bond_R = i32(c.E_b, 3), sum(bond_dims)
        pl.bond_type = torch.zeros(c.E_b, dtype=torch.float32, device=device)
        pl.N_dev = self.sizes[0:1]
        dn = types.SimpleNamespace(N_max=c.n_max, pair_ptr=self.pair_ptr, P=c.P, nmax_dev=self.sizes[4:5])
        pl.dense = dn
        self.plan = pl
        # the Batch-like object the models see: tensors whose VALUES the kernels never read (identity / dtype only) are
        # static placeholders; coordinates are the live view of the raw blob
        z64 = lambda *s: torch.zeros(*s, dtype=torch.int64, device=device)
        b = Batch(x=z64(c.N, K_ATOM), edge_index=z64(2, c.E_b), edge_attr=z64(c.E_b, 3), positions=self.positions,
                  extended_edge_index=z64(2, 8), batch=z64(c.N))
        b.num_graphs = c.B
        b._msde_plan = pl
        b._bucket = self
        self.batch = b
        from .geom3d import nn as _nn
        _nn.register_plan(b, pl)

    # -- feeding -----------------------------------------------------------------------------------------------
    def load(self, blob):
        """Copy a raw blob (host pinned / pageable, or device) into the bucket: the only per-batch transfer."""
        self.raw.copy_(blob, non_blocking=True)

    def build_plan_on_device(self, side_stream=None):
        """csrc/plan.hip: 8 launches, no host synchronisation (capturable).  The per-table-row atom lists (two thirds of
        the time: ~100 of 150 us) are only read by the embedding BACKWARD kernels: with `side_stream` they are built
        there, behind the CSR build, so the forward of the main chain starts without them; the caller's streams must be
        joined before the backward pass (the trainer's second stream is)."""
        c, pl, p, st = self.caps, self.plan, hip._p, hip._stream()
        _lib.call("msde_plan_build", p(self.x_raw), K_ATOM, p(self.atom_off), p(self.bond_src_raw), p(self.bond_dst_raw),
                  p(self.bond_attr_raw), p(self.bond_off), p(self.mol_atoms), p(self.mol_bonds), c.B, c.N, c.E_b, c.E_e,
                  c.P, c.E_r, MAX_NBR, p(self.mol_ptr_full), p(self.bond_ptr), p(self.pair_ptr), p(self.sizes), p(pl.batch_i32),
                  p(pl.atom_codes), p(pl.z_codes), p(pl.bond.rowptr), p(pl.bond.src), p(pl.bond.dst), p(pl.bond.rowptr_s),
                  p(pl.bond.perm_s), p(pl.bond_codes), p(pl.bond_type), p(self.ext_rows), p(self.ext_cnt), p(self.ext_ptr),
                  p(pl.ext.rowptr), p(pl.ext.src), p(pl.ext.dst), p(pl.ext.rowptr_s), p(pl.ext.perm_s), p(self.err), st)
        # (msde_plan_build clears *err itself at the start of every build)
        def lists(st_):
            _lib.call("msde_plan_row_lists", p(pl.atom_codes), p(pl.N_dev), K_ATOM, pl.atom_R, p(self._cnt_scratch),
                      p(pl.atom_list_ptr), p(pl.atom_list_nodes), st_)
            _lib.call("msde_plan_row_lists", p(pl.z_codes), p(pl.N_dev), 1, self.node_class, p(self._cnt_scratch),
                      p(pl.z_list[1]), p(pl.z_list[2]), st_)
        if side_stream is None:
            lists(st)
        else:
            side_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side_stream):
                lists(hip._stream())

    def poll_overflow(self):
        """Device-side overflow flag WITHOUT a synchronisation of the running step: returns the flag as it was when the
        PREVIOUS poll's copy completed (one call late), and queues the next copy behind the work submitted so far.
        csrc/plan.hip sets the flag when a molecule exceeds its limits or the batch exceeds a capacity (such rows are
        then left inert: no out-of-bounds write)."""
        if not hasattr(self, "_err_host"):
            self._err_host = torch.zeros(1, dtype=torch.int32).pin_memory()
            self._err_event = None
        seen = False
        if self._err_event is not None:
            self._err_event.synchronize()          # a copy queued one step ago: done long before
            seen = bool(int(self._err_host[0]))
        self._err_host.copy_(self.err, non_blocking=True)
        self._err_event = torch.cuda.Event()
        self._err_event.record()
        return seen

    def bounds_map(self):
        """{row capacity: device count} of this bucket's tensors (hip.row_bounds): atoms, bonds, extended edges, atom pairs,
        and the two-rows-per-extended-edge tensors of the 2D->3D model."""
        c = self.caps
        return {c.N: self.sizes[0:1], c.E_b: self.sizes[1:2], c.E_e: self.sizes[2:3], c.P: self.sizes[3:4],
                2 * c.E_e: self.sizes[6:7]}

    def bounds(self):
        """`with bucket.bounds():` -- the scope in which steps on this bucket's batch are launched or captured."""
        return hip.row_bounds(self.bounds_map())

    def check(self):
        """Host-side validation (synchronises): the loaded batch fitted the capacities."""
        s = self.sizes.cpu().tolist()
        c = self.caps
        ok = (int(self.err.cpu()) == 0 and s[0] <= c.N and s[1] <= c.E_b and s[2] <= c.E_e and s[3] <= c.P and
              s[4] <= c.n_max and s[5] <= c.E_r)
        return ok, dict(N=s[0], E_b=s[1], E_e=s[2], P=s[3], n_max=s[4], E_r_bound=s[5])


class BlobFeeder:
    """Pinned-host raw blobs -> the bucket, ONE STEP AHEAD: blob t+1 crosses PCIe on a copy stream into a staging buffer
    while step t runs; `load_next()` then needs only a device-to-device copy of the staged blob on the compute stream.
        feeder.submit(blob_0)
        for t in ...: feeder.submit(blob_{t+1}); feeder.load_next(); trainer.step_graph(bucket.batch)
    Slots are recycled only after the compute stream has copied out of them (events both ways, no host sync)."""

    def __init__(self, bucket, depth=2):
        self.bk = bucket
        self.copy_stream = torch.cuda.Stream(device=bucket.device)
        self.staging = [torch.empty_like(bucket.raw) for _ in range(depth)]
        self.ready = [None] * depth           # recorded on the copy stream when slot i holds a blob
        self.drained = [None] * depth         # recorded on the compute stream when slot i has been copied out
        self.head = self.tail = 0             # next slot to fill / to consume

    def submit(self, blob):
        i = self.head % len(self.staging)
        assert self.head - self.tail < len(self.staging), "BlobFeeder: more submits than slots"
        with torch.cuda.stream(self.copy_stream):
            if self.drained[i] is not None:
                self.copy_stream.wait_event(self.drained[i])
            self.staging[i].copy_(blob, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
            self.ready[i] = ev
        self.head += 1

    def load_next(self):
        assert self.tail < self.head, "BlobFeeder: nothing submitted"
        i = self.tail % len(self.staging)
        cur = torch.cuda.current_stream()
        cur.wait_event(self.ready[i])
        self.bk.raw.copy_(self.staging[i], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(cur)
        self.drained[i] = ev
        self.tail += 1
