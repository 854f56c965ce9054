"""Re-laid-out copies of the parameters that a kernel reads in another arrangement than nn.Linear stores them (transposed
weights for the row strips / 2-D tiles, stacked or permuted layouts for fused layers): one cache for the process, refreshed
lazily by version counter / parameter epoch or by ONE batched launch behind the optimiser step -- also inside a captured
hipGraph (split out of hip.py in round 5; hip.py re-exports every name)."""
import ctypes
import weakref as _weakref

import torch

from . import _lib
from ._torchabi import _stream, _p
from .slabs import _retire, _KEEP_ALIVE, upload_table


def weight_epoch():
    """The parameter epoch (moved by optimisers that update parameters through raw pointers): part of every cache key that must
    not survive a parameter update."""
    return _WT_EPOCH


# ---- transposed weight copies ------------------------------------------------------------------------------------
# The row-strip kernels read their weight operand as [K][N] (csrc/gemm_rs.h): an input-gradient product takes nn.Linear's
# weight [out][in] as stored, a forward product needs its transpose.  One copy per weight is kept here and refreshed
#   * lazily, when the weight's autograd version counter moved (torch optimisers, load_state_dict, init), or when the
#     parameter epoch moved (moleculesde_amd.optim.FlatAdam updates parameters through raw pointers and calls
#     bump_weight_epoch());
#   * by ONE batched launch for all known weights (refresh_weight_t(), called by the trainer right after the optimiser
#     step, inside the captured graph) -- then no forward of the next step launches a transpose.
_WT = {}                 # key -> entry dict(wt, refs, versions, epoch)
_WT_EPOCH = 0
_WT_TABLE = {}           # device -> dict(n, host/dev tables) of the batched refresh


def bump_weight_epoch():
    global _WT_EPOCH
    _WT_EPOCH += 1


def _transpose_into(entries):
    """Tables of one msde_transpose_multi launch for `entries` (same device): one table row per block."""
    dev = entries[0]["wt"].device
    blocks = [b for e in entries for b in e["blocks"]]
    n = len(blocks)
    tab = torch.zeros(n, 8, dtype=torch.int64)
    pre = torch.empty(n + 1, dtype=torch.int32)
    total = 0
    for i, blk in enumerate(blocks):
        src_ptr, dst_ptr, r, c, src_ld, dst_ld, mode = blk[:7]
        tab[i, 0], tab[i, 1], tab[i, 2], tab[i, 3], tab[i, 4], tab[i, 5], tab[i, 6] = src_ptr, dst_ptr, r, c, src_ld, dst_ld, mode
        pre[i] = total
        total += ((r + 31) // 32) * ((c + 31) // 32)          # tiles of a block: 32 x 32 source elements
    pre[n] = total
    return tab, pre, total, dev


def _fill_entry(ent):
    for blk in ent["blocks"]:
        src_ptr, dst_ptr, r, c, src_ld, dst_ld, mode = blk[:7]
        _lib.call("msde_relayout", ctypes.c_void_p(src_ptr), src_ld, ctypes.c_void_p(dst_ptr), dst_ld, r, c, mode, _stream())


def _clear_tables():
    """Forget the eager refresh tables (the set of entries changed).  A captured graph never reads THESE tables -- a capture
    builds its own, owned by the process for good (refresh_weight_t) -- but once anything was captured they are parked
    instead of freed all the same."""
    for t in _WT_TABLE.values():
        _retire([t["tab"], t["pre"]])
    _WT_TABLE.clear()


def _drop_entry(key):
    """Remove a cache entry (its parameter died or moved).  A captured graph may still write the entry's buffer at every
    replay: parked, never handed back to the allocator, once a capture happened."""
    ent = _WT.pop(key, None)
    if ent is not None:
        _retire([ent["wt"]])
    _clear_tables()


def _cached_layout(key, src, make):
    """Entry of the re-laid-out weight cache for the leaf parameters `src`; make() -> (buffer, blocks) on a miss.  The
    buffer is refreshed (one msde_relayout per block on the current stream) when a source's version counter or the
    parameter epoch moved; refresh_weight_t() does it for every entry in one launch."""
    ent = _WT.get(key)
    if ent is None:
        def drop(_r, key=key):
            _drop_entry(key)
        wt, blocks = make()
        ent = {"wt": wt, "blocks": blocks, "refs": [_weakref.ref(p, drop) for p in src], "versions": None, "epoch": -1,
               "src_ptrs": tuple(p.data_ptr() for p in src)}
        _WT[key] = ent
        _clear_tables()
    versions = tuple(p._version for p in src)
    if ent["versions"] != versions or ent["epoch"] != _WT_EPOCH:
        _fill_entry(ent)
        ent["versions"], ent["epoch"] = versions, _WT_EPOCH
    return ent["wt"]


def weight_t(w):
    """[K][N] copy of the 2-D fp32 weight w [N][K] (see above); refreshed when stale.  Only leaf tensors (parameters)
    and free concatenation views of leaves are kept; anything else (a weight computed in the forward) is transposed
    on the spot."""
    src = getattr(w, "_msde_src", None) or (w,)
    # only PARAMETERS are cached (their storage lives as long as the model); any other leaf -- a weight computed under
    # no_grad, a test tensor -- would leave an entry whose buffer a later capture bakes in and whose death frees it
    stable = not getattr(w, "_msde_volatile", False) and all(isinstance(p, torch.nn.Parameter) for p in src)
    if not stable:
        wc = w if w.is_contiguous() else w.contiguous()
        wt = torch.empty(w.size(1), w.size(0), dtype=torch.float32, device=w.device)
        _lib.call("msde_transpose", _p(wc), _p(wt), int(w.size(0)), int(w.size(1)), _stream())
        return wt
    key = (tuple(id(p) for p in src), int(w.size(0)), int(w.size(1)), w.data_ptr())

    def make():
        wt = torch.empty(w.size(1), w.size(0), dtype=torch.float32, device=w.device)
        return wt, [(w.data_ptr(), wt.data_ptr(), int(w.size(0)), int(w.size(1)), int(w.size(1)), int(w.size(0)), 0)]
    return _cached_layout(key, src, make)


def weight_layout(tag, params, shape, blocks):
    """A cached buffer of `shape` (zero-initialised once) assembled from blocks of the leaf parameters `params`:
    blocks = [(param, row0, col0, rows, cols, dst_row0, dst_col0, transpose)] -- the rows x cols block of `param` at
    (row0, col0) is copied (transpose False) to, or written transposed (True) at, (dst_row0, dst_col0) of the buffer.
    Used for operands a fused layer reads in another arrangement than nn.Linear stores them (stacked halves, permuted /
    zero-padded input columns, a bias behind a zero half); refreshed with the transposed weight copies -- once per
    optimiser step, by the same launch."""
    assert all(p.is_leaf and p.dim() in (1, 2) and p.is_contiguous() and p.dtype == torch.float32 for p in params)
    key = (tag, tuple(id(p) for p in params), tuple(shape), tuple(p.data_ptr() for p in params))

    def make():
        buf = torch.zeros(*shape, dtype=torch.float32, device=params[0].device)
        ld = int(shape[-1]) if len(shape) == 2 else int(shape[0])
        out = []
        for (p, r0, c0, rows, cols, dr, dc, tr) in blocks:
            p_ld = int(p.size(1)) if p.dim() == 2 else int(p.size(0))
            out.append((p.data_ptr() + 4 * (r0 * p_ld + c0), buf.data_ptr() + 4 * (dr * ld + dc), int(rows), int(cols), p_ld, ld,
                        0 if tr else 1))
        return buf, out
    return _cached_layout(key, tuple(params), make)


def refresh_weight_t():
    """Re-lay-out every known weight copy with one launch per device on the current stream and mark the copies fresh for
    the current parameter epoch (the trainer calls this right after the optimiser step).  Returns the keys of the entries
    it refreshed.  While a hipGraph is being captured, the table the launch reads and the buffers of the entries it names are
    parked for the life of the process: entries that appear or disappear later build NEW eager tables (a table is never
    edited in place), never touch the memory a captured launch reads or writes."""
    if not _WT:
        return ()
    # a parameter whose storage moved since its copy was made (an optimiser that re-points .data into a flat buffer, .to())
    # has a new entry under its new address: the old one would read freed memory -- dropped here
    moved = [k for k, e in _WT.items()
             if any(r() is None or r().data_ptr() != q for r, q in zip(e["refs"], e["src_ptrs"]))]
    for k in moved:
        _drop_entry(k)
    if not _WT:
        return ()
    capturing = torch.cuda.is_current_stream_capturing()
    by_dev = {}
    for k, e in _WT.items():
        by_dev.setdefault(e["wt"].device, []).append((k, e))
    done = []
    for dev, items in by_dev.items():
        entries = [e for _, e in items]
        t = _WT_TABLE.get(dev)
        if t is None or t["n"] != len(entries):
            tab, pre, total, _ = _transpose_into(entries)
            if capturing:
                # entries first made INSIDE this capture (a layer the warm-up steps did not reach: e.g. a loss term switched
                # on since): a host-to-device copy is not capturable, so the table's upload is recorded and performed once
                # after the capture (flush_table_uploads), like every other pointer table of a captured step
                dtab = torch.empty(tab.shape, dtype=tab.dtype, device=dev)
                dpre = torch.empty(pre.shape, dtype=pre.dtype, device=dev)
                upload_table(dtab, tab)
                upload_table(dpre, pre)
                t = {"n": len(entries), "rows": tab.size(0), "tab": dtab, "pre": dpre, "total": total}
            else:
                t = {"n": len(entries), "rows": tab.size(0), "tab": tab.to(dev), "pre": pre.to(dev), "total": total}
                _WT_TABLE[dev] = t
        if capturing:
            # tables are never edited in place (a changed entry set builds new ones): parking this one and the buffers it
            # names is all a replay needs
            _KEEP_ALIVE.extend([t["tab"], t["pre"]] + [e["wt"] for e in entries])
        _lib.call("msde_transpose_multi", _p(t["tab"]), _p(t["pre"]), t["rows"], t["total"], _stream())
        for k, e in items:
            e["versions"] = tuple(r()._version for r in e["refs"] if r() is not None)
            e["epoch"] = _WT_EPOCH
            done.append(k)
    return tuple(done)


def weight_copies_after_replay(keys):
    """Call after REPLAYING a captured step whose optimiser update runs inside the graph: the parameters changed without
    any Python running, the graph's own refresh launch re-laid-out the entries `keys` (what refresh_weight_t returned at
    capture) -- those are fresh, every other entry (made later, e.g. by an eager step on another batch shape) is stale
    and is refreshed lazily at its next use."""
    bump_weight_epoch()
    for k in keys:
        e = _WT.get(k)
        if e is not None:
            e["versions"] = tuple(r()._version for r in e["refs"] if r() is not None)
            e["epoch"] = _WT_EPOCH


def invalidate_weight_copies():
    """The parameters were rewritten through a path that moves neither the autograd version counters nor the optimiser
    (p.data.copy_, a write into the flat parameter buffer, a broadcast): every re-laid-out copy is stale."""
    bump_weight_epoch()


def sync_weight_copies():
    """Call before REPLAYING a captured graph: a replay runs no Python, so it cannot notice that a parameter was changed
    from outside the optimiser since the copies were refreshed (load_state_dict, an in-place edit: the autograd version
    counters moved).  Host-side comparison of the counters (microseconds); on a mismatch every copy is refreshed with one
    launch on the current stream, in front of the replay.  Returns True if it had to."""
    for e in _WT.values():
        if e["epoch"] != _WT_EPOCH or e["versions"] != tuple(r()._version for r in e["refs"] if r() is not None):
            refresh_weight_t()
            return True
    return False
