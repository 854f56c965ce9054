"""MD17 force fine-tuning step -- the counterpart of the training loop of examples/finetune_MD17.py:34-88.

    energy  = graph_pred_linear(model(x, positions, batch)).squeeze(1)            (:52-60; PaiNN: + radius_edge_index)
    energy  = energy * FORCE_MEAN_TOTAL + ENERGY_MEAN_TOTAL * NUM_ATOM           (:63-66, --energy_force_with_normalization)
    force   = -grad(energy, positions, ones, create_graph=True, retain_graph=True) (:68)
    loss    = MD17_energy_coeff * L1(energy, y) + MD17_force_coeff * L1(force, f) (:74; config.py:35-36: 0.05 / 0.95)
    optimizer.zero_grad(); loss.backward(); optimizer.step()                      (:76-78, Adam)

Every batch of an MD17 task is the same molecule in another conformation (MD17_train_batch_size molecules of NUM_ATOM
atoms), so the shapes never change: `ForceTrainer.capture` records the WHOLE step -- radius graph, energy, the
create_graph differentiation, both losses, the backward pass through the forces and Adam -- into ONE hipGraph whose
inputs (positions, energies, forces) are static device buffers; `step_graph` copies a batch into them and replays.  The
step is ~1500 tiny launches on a 21-atom molecule: launched from the host it is bound by launch latency, replayed it is
bound by the dependent chain on the device.  The encoder runs on the closed twice-differentiable operator set of
moleculesde_amd/dd.py (SchNet._forward_force_path, PaiNN.forward); there is no host synchronisation in the step.
"""
import torch

from . import _lib, dd, hip, slabs, wcache
from .optim import FlatAdam


class ForceTrainer:
    def __init__(self, model, graph_pred_linear=None, lr=5e-4, energy_coeff=0.05, force_coeff=0.95, normalization=None,
                 weight_decay=0.0):
        """model: geom3d.SchNet or geom3d.PaiNN on a HIP device; graph_pred_linear: the nn.Linear(emb_dim, 1) head of
        finetune_MD17.py:276-279 (None: the encoder's readout is the energy).  normalization: None or
        (ENERGY_MEAN_TOTAL, FORCE_MEAN_TOTAL, NUM_ATOM) of finetune_MD17.py:217-230."""
        self.model, self.head = model, graph_pred_linear
        self.energy_coeff, self.force_coeff = float(energy_coeff), float(force_coeff)
        self.normalization = normalization
        params = list(model.parameters()) + (list(graph_pred_linear.parameters()) if graph_pred_linear is not None else [])
        self.device = params[0].device
        if self.device.type != "cuda":
            raise RuntimeError("moleculesde_amd.finetune_md17 runs on the HIP device only (no CPU fallback)")
        self.opt = FlatAdam([{"params": params, "lr": lr}], weight_decay=weight_decay)
        self.is_painn = type(model).__name__ == "PaiNN"
        self._graph = None

    # ---- one step, launched from the host ------------------------------------------------------------------------
    def energy_and_force(self, batch, positions, create_graph=True, negate=True):
        """finetune_MD17.py:49-68 (train) / :107-127 (eval, create_graph=False).  negate=False: returns d energy / d positions
        (= -force) without the sign flip (the training step's loss kernel applies it)."""
        if self.is_painn:
            rep = self.model(batch.x, positions, batch.radius_edge_index, batch.batch)
        else:
            rep = self.model(batch.x, positions, batch.batch)
        if isinstance(self.head, torch.nn.Linear):
            # nn.Linear(emb_dim, 1) (finetune_MD17.py:276-279) on the library's twice-differentiable Linear: its product, both
            # input gradients and the weight gradients (queued into the step's grouped launch) instead of five vendor GEMM
            # launches + a bias-gradient reduction
            energy = dd.linear(rep, self.head.weight, self.head.bias).squeeze(1)
        else:
            energy = (self.head(rep) if self.head is not None else rep).squeeze(1)
        if self.normalization is not None:
            e_mean, f_mean, n_atom = self.normalization
            energy = energy * f_mean + e_mean * n_atom
        with dd.positions_only():        # (no parameter gradients in this differentiation: skips ~30 unused weight-gradient GEMMs)
            dE = torch.autograd.grad(energy, positions, grad_outputs=self._ones_like(energy), create_graph=create_graph,
                                     retain_graph=create_graph)[0]
        return energy, (-dE if negate else dE)

    def _ones_like(self, energy):
        """d energy / d energy, kept (no fill launch per step)."""
        o = getattr(self, "_ones", None)
        if o is None or o.shape != energy.shape or o.device != energy.device:
            o = self._ones = torch.ones_like(energy)
        return o

    def _body(self, batch, positions, y, force_t):
        pos = positions.detach().requires_grad_(True)
        energy, dE = self.energy_and_force(batch, pos, negate=False)      # dE = d energy / d positions = -force
        # both L1 losses AND their derivatives w.r.t. energy and dE from one launch (msde_l1_energy_force_loss): the backward
        # pass starts from those instead of walking sub / abs / mean / mul / add / neg and their backward operators (~20 launches)
        y, force_t = hip._f32(y), hip._f32(force_t)
        e32, d32 = hip._f32(energy.detach()), hip._f32(dE.detach())
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        g_e, g_d = torch.empty_like(e32), torch.empty_like(d32)
        _lib.call("msde_l1_energy_force_loss", hip._p(e32), hip._p(y), e32.numel(), hip._p(d32), hip._p(force_t), d32.numel(), -1.0,
                  self.energy_coeff, self.force_coeff, hip._p(loss), hip._p(g_e), hip._p(g_d), hip._stream())
        loss = loss[0]
        self.opt.zero_grad()
        # the weight gradients of the whole step (two contributions per Linear: energy path and force path) as ONE grouped
        # launch + ONE slab reduction behind the backward pass (slabs.weight_grad_leaf) instead of ~170 per-layer launches
        slabs.begin_param_grad_batch(self.opt.params)
        try:
            torch.autograd.backward([energy, dE], [g_e, g_d])
        finally:
            slabs.finish_param_grad_batch()
        self.opt.step_from_grads()
        self._refreshed = wcache.refresh_weight_t()    # re-laid-out weight copies (if a layer reads one) follow the update
        return loss.detach()

    def step(self, batch, y=None, force=None):
        """batch: prepared (geom3d.prepare_batch) Batch with .positions; y [B], force [N, 3] default to batch.y / batch.force."""
        y = batch.y if y is None else y
        force = batch.force if force is None else force
        return self._body(batch, batch.positions, y.view(-1).float(), force.float())

    # ---- the same step as one hipGraph ---------------------------------------------------------------------------
    def capture(self, batch, y=None, force=None, eager_steps=2):
        """Warm up on `batch`, then capture.  The batch object (its plan: atom types, molecule pointers) is baked in: every
        later batch must hold the same molecules in the same order -- the MD17 loaders do; only positions, energies and
        forces change."""
        y = batch.y if y is None else y
        force = batch.force if force is None else force
        self._batch = batch
        self._pos = batch.positions.detach().clone()
        self._y = y.view(-1).float().clone()
        self._f = force.float().clone()
        for _ in range(eager_steps):
            self._body(batch, self._pos, self._y, self._f)
        self.opt.new_table_slot()
        slabs.new_param_grad_slot(self.device)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with slabs.no_gc(), torch.cuda.graph(g, capture_error_mode="thread_local"):
            self._loss = self._body(batch, self._pos, self._y, self._f)
        slabs.flush_table_uploads()
        self.opt.use_eager_slot()
        slabs.use_eager_param_grad_slot()
        self._graph = g
        self._graph_wt_keys = self._refreshed          # the copies the captured refresh launch re-lays-out at every replay
        return g

    def step_graph(self, positions, y, force):
        """Copy one batch's positions / energies / forces into the captured step's buffers and replay it.  Returns the loss
        (a device scalar, overwritten by the next replay)."""
        self._pos.copy_(positions, non_blocking=True)
        self._y.copy_(y.view(-1), non_blocking=True)
        self._f.copy_(force, non_blocking=True)
        wcache.sync_weight_copies()           # parameters edited from outside since the last step (load_state_dict, ...)
        self._graph.replay()
        wcache.weight_copies_after_replay(self._graph_wt_keys)
        return self._loss
