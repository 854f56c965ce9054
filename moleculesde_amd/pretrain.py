"""Counterpart of examples/pretrain_MoleculeSDE.py:106-175,178-348 on the HIP-backed models.

Same flag names and defaults for everything that shapes the hot path (examples/config.py), same
loss composition  CL*c1 + L_2Dto3D*c2 + 0.5(L_x + L_adj)*c3  (:135-152), same optimiser grouping
(:331-337), same checkpoint dictionary (:78-88).  Differences, all deliberate:
  * per-step loss logging accumulates on the device (the reference forces 3 D2H syncs per step);
  * Adam is one flat HIP kernel; under DP one RCCL all-reduce precedes it;
  * the data here is the synthetic PCQM4Mv2-shaped generator (no dataset exists in the containers).
"""
import argparse
import os
import sys
import time

import torch
import torch.nn.functional as F

from . import dp, slabs, wcache
from .geom3d import GNN, SchNet, SDEModel2Dto3D_01, SDEModel2Dto3D_02, prepare_batch
from .geom3d import nn as _nn
from .optim import FlatAdam


def add_hot_path_flags(parser):
    """Subset of examples/config.py that shapes the pretrain step (same names and defaults)."""
    a = parser.add_argument
    a("--seed", type=int, default=42)
    a("--device", type=int, default=0)
    a("--epochs", type=int, default=100)
    a("--batch_size", type=int, default=128)
    a("--lr", type=float, default=1e-4)
    a("--decay", type=float, default=0)
    a("--gnn_type", type=str, default="GIN")
    a("--num_layer", type=int, default=5)
    a("--emb_dim", type=int, default=300)
    a("--dropout_ratio", type=float, default=0.5)
    a("--JK", type=str, default="last")
    a("--gnn_2d_lr_scale", type=float, default=1)
    a("--gnn_3d_lr_scale", type=float, default=1)
    a("--model_3d", type=str, default="SchNet")
    a("--SchNet_num_filters", type=int, default=128)
    a("--SchNet_num_interactions", type=int, default=6)
    a("--SchNet_num_gaussians", type=int, default=51)
    a("--SchNet_cutoff", type=float, default=10)
    a("--SchNet_readout", type=str, default="mean", choices=["mean", "add"])
    a("--PaiNN_radius_cutoff", type=float, default=5.0)      # config.py (PaiNN block)
    a("--PaiNN_n_interactions", type=int, default=3)
    a("--PaiNN_n_rbf", type=int, default=20)
    a("--PaiNN_readout", type=str, default="add", choices=["mean", "add"])
    a("--CL_similarity_metric", type=str, default="InfoNCE_dot_prod")
    a("--T", type=float, default=0.1)
    a("--normalize", dest="normalize", action="store_true")
    a("--no_normalize", dest="normalize", action="store_false")
    a("--SDE_type_2Dto3D", type=str, default="VE")
    a("--SDE_type_3Dto2D", type=str, default="VE")
    a("--SDE_2Dto3D_model", type=str, default="SDEModel2Dto3D_01")
    a("--SDE_3Dto2D_model", type=str, default="SDEModel3Dto2D_node_adj_dense")
    a("--SDE_coeff_contrastive", type=float, default=1)
    a("--SDE_coeff_contrastive_skip_epochs", type=int, default=0)
    a("--SDE_coeff_generative_2Dto3D", type=float, default=1)
    a("--SDE_coeff_generative_3Dto2D", type=float, default=1)
    a("--use_extend_graph", dest="use_extend_graph", action="store_true")
    a("--no_extend_graph", dest="use_extend_graph", action="store_false")
    parser.set_defaults(use_extend_graph=True)
    a("--noise_on_one_hot", dest="noise_on_one_hot", action="store_true")
    a("--no_noise_on_one_hot", dest="noise_on_one_hot", action="store_false")
    parser.set_defaults(noise_on_one_hot=True)
    a("--SDE_anneal_power", type=float, default=0)
    a("--output_model_dir", type=str, default="")
    a("--verbose", dest="verbose", action="store_true")
    # (not in examples/config.py) the 2D->3D score network under autograd: "ops" = operator by operator (default, the faster
    # pretrain step), "mol" = one launch each way with one workgroup per molecule (moleculesde_amd/escore.py)
    a("--score_kernel", type=str, default="ops", choices=["ops", "mol"])
    return parser


def readme_args(**over):
    """The README pre-training command (README.md:86-93) as a namespace."""
    p = add_hot_path_flags(argparse.ArgumentParser())
    args = p.parse_args([
        "--model_3d=SchNet", "--lr=1e-4", "--batch_size=256", "--gnn_3d_lr_scale=0.1", "--dropout_ratio=0",
        "--emb_dim=300", "--epochs=1", "--SDE_coeff_contrastive=1", "--CL_similarity_metric=EBM_node_dot_prod",
        "--T=0.1", "--normalize", "--SDE_coeff_contrastive_skip_epochs=0", "--SDE_coeff_generative_2Dto3D=1",
        "--SDE_2Dto3D_model=SDEModel2Dto3D_02", "--SDE_type_2Dto3D=VE", "--use_extend_graph",
        "--SDE_coeff_generative_3Dto2D=1", "--SDE_3Dto2D_model=SDEModel3Dto2D_node_adj_dense",
        "--SDE_type_3Dto2D=VE", "--noise_on_one_hot"])
    for k, v in over.items():
        setattr(args, k, v)
    return args


USE_FUSED_CL = True
_STREAMS = {}


def shared_stream(device, name):
    """One HIP stream per (device, role) for the whole process.  HIP maps streams onto a handful of hardware queues: a second
    Trainer with streams of its own (bench.py times configs[2] behind configs[1] in one process) found its plan-ahead stream on
    the queue of another of its streams -- the two-bucket mode ran 3 % slower as the second trainer of a process than as the
    first (3.32 vs 3.22 ms).  Trainers of one process run one after the other, so they can share."""
    key = (str(device), name)
    st = _STREAMS.get(key)
    if st is None:
        st = _STREAMS[key] = torch.cuda.Stream(device=device)
    return st


# Stream layout of the step, fixed by alternating A/B runs on one box (numbers: DESIGN.md rounds 2-4).  What lost is gone from
# the code: weight gradients flushed early or on the second stream, the contrastive loss on the second stream, SchNet started
# behind GIN, a third stream for the coordinate branch, leaf kernels in front of the grouped launch.
SPLIT_HEAD_ROOT = True          # the 3D->2D head's loss as a backward root of its own on the second stream (Trainer.losses)
SIDE_CFCONV_FWD_WGS = 512     # SchNet's CFConv kernels beside the main chain: forward 128: 2.825 ... 512: 2.736, 1024: 2.750 ms
SIDE_CFCONV_BWD_WGS = 224     # weight gradient (round 4, a launch per block) 128: 2.768, 160: 2.757, 192: 2.755, 256: 2.783 ms; round 6 (launches of
                              # three blocks) 160: 2.473, 176: 2.476, 192: 2.471, 224: 2.469 ms -- flat within 0.3 %
SIDE_CFCONV_BWD_GROUP = 3     # SchNet alone on the second stream: its filter-weight gradients as two launches of three blocks
                              # beside the GIN backward (2.476 vs 2.493 ms for one launch of six, 2.500 for six of one; with the
                              # 3D->2D head behind SchNet the second stream is the long pole and ONE launch wins: 3.268 / 3.281 /
                              # 3.318 / 3.430 ms for 6 / 3 / 2 / 1 blocks per launch -- profiles/r06_ab_cfconv_hoisted.txt)
GEOMETRY_ON_SIDE = True       # coordinate-only branch of the 2D->3D model at the head of the second stream (2.86 vs 2.98 ms)
EARLY_SLAB_REDUCE = True      # the second stream sums the CFConv slabs it wrote, in the shadow of the GIN backward
PLAN_LISTS_ON_SIDE = True     # bucket mode: embedding row lists off the main chain
# (Round 6, measured and removed: with the filter-weight gradients batched the second stream's backward CHAIN ends ~320 us before the
# main stream's in configs[1] (stamps: bwd_side_chain_end 1750 vs bwd_main_end 2070 us); launching the weight-gradient problems whose
# operands the second stream produced THERE, as a grouped launch of their own beside the GIN backward, took 100 us off the tail
# (435 -> 333 us) and put 84 us on the GIN backward: 2.476 vs 2.470 ms, --full 3.389 vs 3.268 -- profiles/r06_ab_side_wgrad_flush.txt)
# Data parallel: weight gradients, flattening and all-reduce bucket by bucket, so that a bucket's all-reduce travels behind the
# next bucket's weight-gradient work (Trainer._dp_tail).  Built and parity-green (tests/test_gpu_dp.py runs it), OFF by default:
# the per-bucket pieces cost the compute stream more than the collectives they hide -- 1-rank RCCL, one box, alternating:
# 2.775 vs 2.573 ms per step (three grouped launches + three reductions + three flattening launches as graph pieces of their
# own instead of one of each inside the step's graph, and the leaf kernels no longer run beside the grouped launch), against
# an all-reduce of 14 MB that the ring estimate puts at ~0.2 ms on 8 GPUs (profiles/r06_dp_tail_overlap_ab.txt).
FANOUT_2D_REPR = True       # the GIN output through one fan-out node for its three consumers (losses())
DP_OVERLAP = False

_SDE_RANGES_2D3D = {"VE": ("VE", 0.2, 1.0), "VP": ("VP", 0.2, 1.0), "VE02": ("VE", 0.1, 10.0), "VP02": ("VP", 0.2, 30.0),
                    "VE03": ("VE", 0.1, 1000.0), "VP03": ("VP", 0.2, 1000.0)}


def build_models(args, device):
    """pretrain_MoleculeSDE.py:197-315 (SchNet / SDEModel2Dto3D_02 / SDEModel3Dto2D_node_adj_dense)."""
    node_class = 119
    models = {}
    models["model_2D"] = GNN(args.num_layer, args.emb_dim, JK=args.JK, drop_ratio=args.dropout_ratio,
                             gnn_type=args.gnn_type).to(device)
    if args.model_3d == "SchNet":
        models["model_3D"] = SchNet(hidden_channels=args.emb_dim, num_filters=args.SchNet_num_filters,
                                    num_interactions=args.SchNet_num_interactions,
                                    num_gaussians=args.SchNet_num_gaussians, cutoff=args.SchNet_cutoff,
                                    readout=args.SchNet_readout, node_class=node_class).to(device)
    elif args.model_3d == "PaiNN":                              # pretrain_MoleculeSDE.py:212-221
        from .geom3d import PaiNN
        models["model_3D"] = PaiNN(n_atom_basis=args.emb_dim, n_interactions=args.PaiNN_n_interactions,
                                   n_rbf=args.PaiNN_n_rbf, cutoff=args.PaiNN_radius_cutoff, max_z=node_class, n_out=1,
                                   readout=args.PaiNN_readout).to(device)
    else:
        raise NotImplementedError(f"Model {args.model_3d} not included.")
    cls23 = {"SDEModel2Dto3D_01": SDEModel2Dto3D_01, "SDEModel2Dto3D_02": SDEModel2Dto3D_02}.get(args.SDE_2Dto3D_model)
    if cls23 is None:
        raise NotImplementedError(args.SDE_2Dto3D_model)
    sde_type, bmin, bmax = _SDE_RANGES_2D3D[args.SDE_type_2Dto3D]
    models["SDE_2Dto3D_model"] = cls23(
        emb_dim=args.emb_dim, hidden_dim=32, beta_min=bmin, beta_max=bmax, num_diffusion_timesteps=1000,
        beta_schedule=None, SDE_type=sde_type, use_extend_graph=args.use_extend_graph).to(device)
    if args.SDE_coeff_generative_3Dto2D > 0:
        from .geom3d import sde_3d_to_2d  # noqa: F401  (built in a later milestone)
        models["SDE_3Dto2D_model"] = sde_3d_to_2d.build_from_args(args, node_class).to(device)
    return models


def ordered_parameters(model):
    """model.parameters() with every fusion set (module.fusion_sets(): parameters the kernels consume as one
    concatenated operand) moved together, so that FlatAdam lays them out back to back and hip.cat_params is a
    free view.  Adam is element-wise: the order inside a param group changes nothing else."""
    params = list(model.parameters())
    for mod in model.modules():
        sets = getattr(mod, "fusion_sets", None)
        if sets is None:
            continue
        for group in sets():
            ids = {id(p) for p in group}
            first = next(i for i, p in enumerate(params) if id(p) in ids)
            rest = [p for p in params if id(p) not in ids]
            n_before = sum(1 for p in params[:first] if id(p) not in ids)
            params = rest[:n_before] + list(group) + rest[n_before:]
    return params


def make_optimizer(args, models):
    """pretrain_MoleculeSDE.py:331-337."""
    groups = [{"params": ordered_parameters(models["model_2D"]), "lr": args.lr * args.gnn_2d_lr_scale},
              {"params": ordered_parameters(models["model_3D"]), "lr": args.lr * args.gnn_3d_lr_scale},
              {"params": ordered_parameters(models["SDE_2Dto3D_model"]), "lr": args.lr * args.gnn_2d_lr_scale}]
    if "SDE_3Dto2D_model" in models:
        groups.append({"params": ordered_parameters(models["SDE_3Dto2D_model"]),
                       "lr": args.lr * args.gnn_3d_lr_scale})
    return FlatAdam(groups, betas=(0.9, 0.999), eps=1e-8, weight_decay=args.decay)


def do_CL(X, Y, args, noise, neg_index=None):
    """'EBM_node_dot_prod' branch of examples/util.py:52-68 (the metric of the README command)."""
    if args.CL_similarity_metric != "EBM_node_dot_prod":
        raise NotImplementedError(args.CL_similarity_metric)
    if neg_index is None:
        neg_index = noise.randperm(len(Y), Y.device)
    neg_Y = Y[neg_index]
    pred_pos = torch.sum(X * Y, dim=1) / args.T
    pred_neg = torch.sum(X * neg_Y, dim=1) / args.T
    loss_pos = F.binary_cross_entropy_with_logits(pred_pos, torch.ones_like(pred_pos))
    loss_neg = F.binary_cross_entropy_with_logits(pred_neg, torch.zeros_like(pred_neg))
    acc = (torch.sum(pred_pos > 0).float() + torch.sum(pred_neg < 0).float()) / (len(pred_pos) + len(pred_neg))
    return loss_pos + loss_neg, acc.detach()


def dual_CL(X, Y, args, noise, neg_indices=(None, None)):
    """examples/util.py:76-79; the accuracy stays a device scalar (no sync).  On the HIP device both
    directions run as one fused kernel pair (csrc/contrastive.hip)."""
    if X.is_cuda and args.CL_similarity_metric == "EBM_node_dot_prod" and USE_FUSED_CL:
        n1 = neg_indices[0] if neg_indices[0] is not None else noise.randperm(len(Y), Y.device)
        n2 = neg_indices[1] if neg_indices[1] is not None else noise.randperm(len(X), X.device)
        from . import hip
        return hip.contrastive_ebm(X, Y, n1, n2, args.T)
    l1, a1 = do_CL(X, Y, args, noise, neg_indices[0])
    l2, a2 = do_CL(Y, X, args, noise, neg_indices[1])
    return (l1 + l2) / 2, (a1 + a2) / 2


class Trainer:
    """One object per rank: models, flat Adam, the step function of pretrain_MoleculeSDE.py:125-156."""

    def __init__(self, args, device, noise=None, overlap_streams=True):
        self.args, self.device = args, device
        self.models = build_models(args, device)
        self.opt = make_optimizer(args, self.models)
        dp.broadcast_flat(self.opt.flat_p)
        # Replicas start from rank 0's parameters but must draw DIFFERENT noise (position noise, time steps,
        # contrastive negatives, dropout masks): fold the rank into every random stream, as dp.shard_seed does for
        # the data.  (Parameters are already broadcast, so reseeding torch here does not desynchronise them.)
        r = dp.rank()
        self.dp_enabled = True            # False: ignore the process group (single-replica reference runs in DP tests)
        self.dp_buckets = True            # per-model all-reduce buckets, each followed by its Adam (False: one flat all-reduce)
        self.dp_timing = None             # a list: step_graph() appends the timing events of each DP step (bench.py)
        self._tm_cur = None
        if r:
            torch.manual_seed(torch.initial_seed() + 7919 * r)
        self.noise = noise or _nn.DeviceNoise(seed=0x5EED + 7919 * r)
        self.models["SDE_2Dto3D_model"].score_network._seed_base += 0x10001 * r
        self.models["SDE_2Dto3D_model"].noise = self.noise
        self.models["SDE_2Dto3D_model"].score_network.mol_kernel_train = getattr(args, "score_kernel", "ops") == "mol"
        if "SDE_3Dto2D_model" in self.models:
            self.models["SDE_3Dto2D_model"].noise = self.noise
        self.coeff_cl = args.SDE_coeff_contrastive
        self.log = {k: torch.zeros((), device=device) for k in ("CL", "CL_acc", "2Dto3D", "3Dto2D")}
        self.steps = 0
        for m in self.models.values():
            m.train()
        # hipGraph mode: one captured graph per batch shape (forward + backward + gradient flattening
        # [+ Adam when single-GPU]); a device-side step counter re-seeds the dropout masks per replay
        self.overlap_streams = overlap_streams      # SchNet (and the coordinate branch) on a second HIP stream
        self._side_stream = shared_stream(device, "side") if torch.device(device).type == "cuda" else None
        self._one_grad = torch.ones((), dtype=torch.float32, device=device)
        if self._one_grad.is_cuda:
            from . import hip as _hip0
            _hip0.register_unit_grad(self._one_grad)      # the loss composition's backward needs no launch for it
        self._bn_modules = [mod for m_ in self.models.values() for mod in m_.modules() if isinstance(mod, _nn.BatchNorm1d)]
        # Measured on MI355X (tools/marginal_cost.py, hipGraph replay, bs 256): 1 stream 5.26 ms, SchNet beside the
        # 2D branch 4.55 ms, a third stream for the 2D->3D coordinate branch 4.73 ms, weight gradients on a fourth
        # 5.4 ms -- the step is a chain of small kernels and every extra queue / cross-stream event costs more
        # dispatch latency than it hides.  Two streams it is.
        self._graphs = {}
        self.adam_outside_graph = False   # True reproduces the multi-GPU structure (graph; all-reduce; Adam) on 1 GPU
        self._graph_pool = None
        self._graph_loss = {}
        self._graph_wt_keys = {}
        self._graph_tails = {}            # data-parallel overlap: the per-bucket tail graphs of a captured step
        self.step_counter = torch.zeros(1, dtype=torch.int64, device=device)

    def _side_geometry(self, on):
        """While SchNet runs beside the GIN -> 2D->3D chain its two wide CFConv kernels take only part of the chip
        (see hip.CFCONV_FWD_WGS); restored afterwards so that standalone users get the full-width kernels."""
        from . import hip
        hip.CFCONV_FWD_WGS = SIDE_CFCONV_FWD_WGS if on else None
        hip.CFCONV_BWD_WGS = SIDE_CFCONV_BWD_WGS if on else None
        if on:
            self._bwd_group_was, hip.CFCONV_BWD_GROUP = hip.CFCONV_BWD_GROUP, min(hip.CFCONV_BWD_GROUP, SIDE_CFCONV_BWD_GROUP)
        elif getattr(self, "_bwd_group_was", None) is not None:
            hip.CFCONV_BWD_GROUP, self._bwd_group_was = self._bwd_group_was, None

    def _encode_3d(self, batch):
        """pretrain_MoleculeSDE.py:131-133: SchNet takes (z, pos, batch); PaiNN also the radius graph of the batch
        (dataset_3D_Radius.py:155; built here with the radius kernels when the loader did not attach one)."""
        m3 = self.models["model_3D"]
        if type(m3).__name__ != "PaiNN":
            return m3(batch.x[:, 0], batch.positions, batch.batch, return_latent=True)
        ei = getattr(batch, "radius_edge_index", None)
        if ei is None:
            from . import hip, plan as _pl
            pl = _pl.get_plan(batch)
            rp, _ = hip.radius_plan(batch.positions, pl.batch_i32, pl.mol_ptr, m3.cutoff, pl.E_r_cap, 32, n_max=getattr(pl, "N_max", None))
            keep = rp.src >= 0                                    # host sync: PaiNN batches are not graph-captured
            # PyG radius_graph orientation: row 0 = source, row 1 = target (the 32-neighbour cap applies per TARGET);
            # PaiNN aggregates at row 0 (painn.py:235), i.e. at the source, exactly as the reference does
            ei = torch.stack([rp.src[keep], rp.dst[keep]]).long()
            batch.radius_edge_index = ei
        return m3(batch.x[:, 0], batch.positions, ei, batch.batch, return_latent=True)

    def losses(self, batch, log=False, split_roots=False):
        """Loss composition of pretrain_MoleculeSDE.py:128-152.  The 3D encoder does not depend on the 2D
        branch (GIN -> 2D->3D score model) until the contrastive term, and most kernels of this 256-molecule
        step fill only part of the chip, so SchNet runs on a second HIP stream beside the 2D branch; autograd
        replays each op's backward on the stream of its forward, so the backward overlaps the same way."""
        a, m = self.args, self.models
        terms, coeffs = [], []      # loss = sum_i c_i * term_i, composed by ONE kernel (hip.combine_losses)
        parts = {}
        main = torch.cuda.current_stream()
        want_32 = a.SDE_coeff_generative_3Dto2D > 0
        # the 3D->2D head depends only on the SchNet output: it follows SchNet on the side stream unless the
        # noise source replays the reference's program order (its draws come last there)
        head_on_side = want_32 and self.overlap_streams and not getattr(self.noise, "replay", False)
        # split_roots (the trainer's own steps): with the 3D->2D head behind SchNet on the second stream, its loss stays a
        # backward ROOT OF ITS OWN on that stream -- the main stream joins SchNet's output only (an event recorded in front of
        # the head), composes the 2D->3D and contrastive terms and starts ITS backward without waiting for the head's ~600 us
        # forward; _backward() differentiates both roots in one pass.  Returned loss: (main root, head root).
        split = bool(split_roots and head_on_side and SPLIT_HEAD_ROOT)
        if self.overlap_streams and not head_on_side:
            # SchNet alone on the second stream is the shorter chain: its wide kernels give way to the main chain
            # (with the 3D->2D head behind it the second stream is the critical one and keeps the full width)
            self._side_geometry(True)          # until the end of this step's backward pass

        def head_32(rep):
            return m["SDE_3Dto2D_model"](rep, batch, reduce_mean=a.noise_on_one_hot, continuous=True, train=True,
                                         anneal_power=a.SDE_anneal_power)      # (loss_x, loss_adj); their mean below

        # random draws keep the reference's program order (contrastive permutations, then the 2D->3D noise);
        # both happen before any encoder runs so that the coordinate-only branch of the 2D->3D model can start
        # right away on its own stream (neither encoder draws random numbers on this path)
        negs = (None, None)
        # device noise: the permutation kernel (20 us, read 1.2 ms later by the contrastive loss) runs at the head of the
        # SECOND stream instead of at the head of the main chain; host call order (= the reference's draw order) unchanged
        negs_on_side = self.coeff_cl > 0 and self.overlap_streams and not getattr(self.noise, "replay", False)
        if self.coeff_cl > 0 and not negs_on_side:
            n = batch.x.size(0)
            negs = self.noise.randperm_pair(n, batch.x.device)
        if (a.SDE_coeff_generative_2Dto3D > 0 and GEOMETRY_ON_SIDE and self.overlap_streams
                and not head_on_side and not getattr(self.noise, "replay", False)):
            # (with the 3D->2D head behind SchNet the second stream is the longer one: 3.92 vs 4.03 ms without / with)
            # the coordinate-only branch of the 2D->3D model (noise, perturbation, frame / Fourier features, their MLPs:
            # ~10 launches forward, as many backward) depends on nothing the main chain computes: it runs at the head
            # of the second stream, in the slack SchNet leaves there; the model joins it by event
            side = self._side_stream
            side.wait_stream(main)
            with torch.cuda.stream(side):
                m["SDE_2Dto3D_model"].begin(batch)
                ev = torch.cuda.Event()
                ev.record(side)
            m["SDE_2Dto3D_model"]._pending_event = ev
        l32 = None
        from . import hip as _hip
        stamps = _hip.STAMPS is not None
        _hip.stamp("fwd_start")
        def schnet_on_side():
            side = self._side_stream
            side.wait_stream(main)
            with torch.cuda.stream(side):
                _hip.stamp("schnet_fwd_start")
                _, rep = self._encode_3d(batch)
                _hip.stamp("schnet_fwd_end")
                # (the permutation kernel BEHIND SchNet's forward: the second stream waits for the main chain there anyway, and
                # the contrastive loss reads the negatives ~300 us later; at the head of the stream it delayed SchNet by 19 us)
                ng = self.noise.randperm_pair(batch.x.size(0), batch.x.device) if negs_on_side else None
                if stamps:
                    rep.register_hook(lambda g: _hip.stamp("schnet_bwd_start"))
                ev = None
                if split:
                    ev = torch.cuda.Event()
                    ev.record(side)
                return rep, (head_32(rep) if head_on_side else None), ng, ev
        rep_ready = None
        if self.overlap_streams:
            node_3D_repr, l32, ng, rep_ready = schnet_on_side()
            if ng is not None:
                negs = ng
                for t in negs:
                    if t is not None:
                        t.record_stream(main)
        else:
            _, node_3D_repr = self._encode_3d(batch)
        node_2D_repr = m["model_2D"](batch.x, batch.edge_index, batch.edge_attr)
        _hip.stamp("gin_fwd_end")
        if stamps:
            node_2D_repr.register_hook(lambda g: _hip.stamp("gin_bwd_start"))
        repr_cl = node_2D_repr
        if a.SDE_coeff_generative_2Dto3D > 0:
            repr_23 = node_2D_repr
            if (FANOUT_2D_REPR and self.coeff_cl > 0 and node_2D_repr.is_cuda and torch.is_grad_enabled()
                    and node_2D_repr.requires_grad):
                # three consumers (the contrastive loss, the 2D->3D model's node and edge embeddings): one fan-out node sums
                # their gradients in ONE launch in front of the GIN backward instead of two autograd additions
                from . import dd as _dd
                repr_cl, r_edge, r_node = _dd.fanout(node_2D_repr, 3)
                repr_23 = (r_edge, r_node)
            l23 = m["SDE_2Dto3D_model"](repr_23, batch, anneal_power=a.SDE_anneal_power)["position"]
            terms.append(l23); coeffs.append(a.SDE_coeff_generative_2Dto3D)
            parts["2Dto3D"] = l23.detach()
            _hip.stamp("2d3d_fwd_end")
            if stamps:
                l23.register_hook(lambda g: _hip.stamp("2d3d_bwd_start"))
        if self.overlap_streams:
            if rep_ready is not None:
                main.wait_event(rep_ready)         # SchNet's output; the head queued behind it keeps the second stream
            else:
                main.wait_stream(self._side_stream)
            node_3D_repr.record_stream(main)
            if l32 is not None and not split:
                l32[0].record_stream(main)
                l32[1].record_stream(main)
        if self.coeff_cl > 0:
            cl, acc = dual_CL(repr_cl, node_3D_repr, a, self.noise, negs)
            terms.append(cl); coeffs.append(self.coeff_cl)
            parts["CL"], parts["CL_acc"] = cl.detach(), acc
        loss_head = None
        if want_32 and split:
            with torch.cuda.stream(self._side_stream):
                c32 = 0.5 * a.SDE_coeff_generative_3Dto2D       # (loss_x + loss_adj) / 2 (pretrain_MoleculeSDE.py:148)
                loss_head = _hip.combine_losses([c32, c32], [l32[0], l32[1]])
                with torch.no_grad():
                    parts["3Dto2D"] = _hip.combine_losses([0.5, 0.5], [l32[0].detach(), l32[1].detach()])
            # allocated on the second stream, read on the main one (_total, the loss log): tell the allocator, and let
            # _backward() join the streams explicitly instead of relying on the autograd engine's leaf-stream join
            loss_head.record_stream(main)
            parts["3Dto2D"].record_stream(main)
        elif want_32:
            if l32 is None:
                l32 = head_32(node_3D_repr)
            terms += [l32[0], l32[1]]                      # (loss_x + loss_adj) / 2 (pretrain_MoleculeSDE.py:148)
            coeffs += [0.5 * a.SDE_coeff_generative_3Dto2D] * 2
            with torch.no_grad():
                parts["3Dto2D"] = (_hip.combine_losses([0.5, 0.5], [l32[0].detach(), l32[1].detach()])
                                   if l32[0].is_cuda else (l32[0].detach() + l32[1].detach()) * 0.5)
        if terms and all(t.is_cuda for t in terms) and len(terms) <= 4:
            logs = None
            if log and not want_32 and len(parts) <= 5:
                # the per-term running sums ride in the same launch (else: one more launch at the end of the step)
                logs = [(parts[k].to(torch.float32) if parts[k].dtype != torch.float32 else parts[k], self.log[k]) for k in parts]
                self._logged_by_launch = True
            loss = _hip.combine_losses(coeffs, terms, logs)
        else:
            loss = 0
            for c_, t_ in zip(coeffs, terms):
                loss = loss + t_ * c_
        if loss_head is not None:
            return (loss, loss_head), parts
        return loss, parts

    def _log_parts(self, parts):
        keys = list(parts.keys())
        if getattr(self, "_logged_by_launch", False):
            self._logged_by_launch = False
            return                     # already added by the loss composition's launch (losses(..., log=True))
        torch._foreach_add_([self.log[k] for k in keys], [parts[k].to(torch.float32) for k in keys])   # one launch

    def _backward(self, loss, finish=True):
        """loss.backward() with the weight-gradient slab reductions of all layers batched into one launch.  finish=False
        (data-parallel tail, _dp_tail): the backward chains and the leaf kernels only -- the queued weight-gradient GEMMs and
        the slab reduction are left to the caller, who runs them bucket by bucket (hip.partition_param_grad_batch ...)."""
        from . import hip
        try:
            roots = loss if isinstance(loss, tuple) else (loss,)
            one = self._one_grad if roots[0].is_cuda else None     # preallocated d(loss)/d(loss): no fill launch per step
            slabs.begin_param_grad_batch(self.opt.params)
            try:
                hip.stamp("bwd_start")
                if len(roots) == 1:
                    roots[0].backward(one)
                else:        # (main root, head root on the second stream): one pass of the engine over both
                    torch.autograd.backward(list(roots), [one] * len(roots))
                hip.stamp("bwd_main_end")
                if hip.STAMPS is not None and self.overlap_streams:
                    with torch.cuda.stream(self._side_stream):
                        hip.stamp("bwd_side_chain_end")     # the second stream's backward CHAIN (bwd_side_end: + its leaf kernels)
                if self.overlap_streams and EARLY_SLAB_REDUCE:
                    # the second stream finished its backward (SchNet) long before the main chain (GIN): it sums the
                    # CFConv filter-gradient slabs (6 x 256 slabs, ~125 MB) it wrote, in the shadow of the GIN backward
                    with torch.cuda.stream(self._side_stream):
                        early = slabs.reduce_written_slabs()
                    if early:
                        torch.cuda.current_stream().wait_stream(self._side_stream)
                if self.overlap_streams and slabs.have_deferred_leaf_kernels():
                    # leaf-only kernels of the backward pass (GIN bond-table gradients: 5 x 17 us that nothing downstream
                    # reads) run on the second stream BESIDE the grouped weight-gradient launch instead of inside the
                    # backward chain.  Host order: everything of SchNet's backward is already queued on that stream.
                    main_, side_ = torch.cuda.current_stream(), self._side_stream
                    side_.wait_stream(main_)
                    with torch.cuda.stream(side_):
                        slabs.run_deferred_leaf_kernels()
                    if finish:
                        slabs.flush_wgrad_gemms()
                    main_.wait_stream(side_)
                if hip.STAMPS is not None and self.overlap_streams:
                    with torch.cuda.stream(self._side_stream):
                        hip.stamp("bwd_side_end")
                    torch.cuda.current_stream().wait_stream(self._side_stream)
                elif len(roots) > 1 and self.overlap_streams:
                    # a root of its own on the second stream (the 3D->2D head's loss): the main stream reads its value in
                    # _total() and in the loss log -- ordered behind the second stream here, not by engine internals
                    torch.cuda.current_stream().wait_stream(self._side_stream)
            finally:
                if finish or sys.exc_info()[0] is not None:
                    slabs.finish_param_grad_batch()
                    hip.stamp("wgrad_end")
        finally:
            self._side_geometry(False)

    @staticmethod
    def _total(loss):
        """The step's loss value from its backward roots (after the backward pass: the current stream is ordered behind both)."""
        if isinstance(loss, tuple):
            return loss[0].detach() + loss[1].detach()
        return loss

    def _use_dp(self):
        return self.dp_enabled and (dp.world_size() > 1 or dp.FORCE_COLLECTIVES)

    def _allreduce_and_adam(self):
        """Data-parallel tail of a step: the flat gradient is all-reduced per bucket (= per model, heads first) and
        each bucket's Adam starts as soon as ITS reduction is done, while the next bucket is still on the wire."""
        if not self.dp_buckets:
            scale = dp.allreduce_mean_(self.opt.flat_g)
            self.opt.step(grad_scale=scale)
            self._refresh_weights()
            return
        order, works, scale = dp.allreduce_buckets_async(self.opt.flat_g, self.opt.bucket_ranges)
        self.opt.begin_bucket_step()
        tm = self._tm_cur                 # bench.py (Trainer.dp_timing): events on the compute stream around each piece
        for i, w in zip(order, works):
            if w is not None:
                w.wait()                  # stream wait: the host does not block
            if tm is not None:
                tm.append(self._timing_event())       # bucket i reduced (what the compute stream had to wait for)
            self.opt.step_bucket(i, grad_scale=scale)
            if tm is not None:
                tm.append(self._timing_event())       # bucket i's Adam done
        self._refresh_weights()

    @staticmethod
    def _timing_event():
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def _dp_order(self):
        """Buckets (= models) in the order their gradients are finished and sent: largest message first, so that it has the
        whole rest of the weight-gradient work to travel behind, smallest last (the one whose all-reduce nothing hides)."""
        r = self.opt.bucket_ranges
        return sorted(range(len(r)), key=lambda i: r[i][0] - r[i][1])

    def _dp_finish_bucket(self, i, table):
        """Bucket i's weight gradients (its problems of the grouped launch), their slab reduction, its rows of the flat
        gradient buffer -- on the current stream; the bucket is then ready for its all-reduce."""
        from . import hip
        slabs.finish_param_grad_part(i)
        self.opt.gather_bucket(i, table)

    def _dp_tail(self, graphs=None):
        """Data-parallel tail of a step (buckets overlapped with the backward work).  The step's weight gradients
        are ONE grouped launch at the end of the backward pass; here that launch is cut by model (hip.partition_param_grad_
        batch): bucket k's problems run, are summed and flattened, and its all-reduce is issued (RCCL's own stream) while
        the compute stream goes on with bucket k + 1's problems; the Adam launches follow in the same order, each behind its
        bucket's reduction.  Only the last (smallest) bucket's all-reduce has nothing to hide behind.  graphs: the captured
        per-bucket pieces (replayed instead of launched)."""
        from . import hip
        order = self._dp_order()
        ranges = self.opt.bucket_ranges
        tm = self._tm_cur
        works = []
        if graphs is None:
            slabs.partition_param_grad_batch(self.opt.grad_bucket_lookup(), len(ranges))
            table = self.opt.grad_table()
        for k, i in enumerate(order):
            if graphs is None:
                self._dp_finish_bucket(i, table)
            else:
                graphs[k].replay()
            a, b = ranges[i]
            _, w, scale = dp.allreduce_buckets_async(self.opt.flat_g, [(a, b)], order=[0])
            works.append(w[0])
        if graphs is None:
            slabs.finish_param_grad_batch()
            hip.stamp("wgrad_end")
        if tm is not None:
            tm.append(self._timing_event())           # every bucket's work queued: end of the "graph" part
        self.opt.begin_bucket_step()
        for i, w in zip(order, works):
            if w is not None:
                w.wait()
            if tm is not None:
                tm.append(self._timing_event())
            self.opt.step_bucket(i, grad_scale=1.0 / dp.world_size())
            if tm is not None:
                tm.append(self._timing_event())
        self._refresh_weights()

    def _refresh_weights(self):
        """Right after an optimiser step: ONE launch re-transposes every weight the forward products read as [K][N]
        (wcache.weight_t), so that no forward of the next step has to."""
        from . import hip as _hip
        self._last_refreshed = wcache.refresh_weight_t()

    def _bounds(self, batch):
        """The row-bound scope of a batch: its capacity bucket's counts, or none for an exact-size batch."""
        from . import hip as _hip
        bk = getattr(batch, "_bucket", None)
        return _hip.row_bounds(bk.bounds_map() if bk is not None else {})

    def step(self, batch):
        with self._bounds(batch):         # (kernels reducing over rows stop at the bucket's valid rows)
            overlap = self._use_dp() and DP_OVERLAP and self.dp_buckets
            single = not self._use_dp()
            self._bump_counters(single)   # the device step counter re-seeds dropout / negatives once a capture set it
            loss, parts = self.losses(batch, log=True, split_roots=True)
            self.opt.zero_grad()
            self._backward(loss, finish=not overlap)
            loss = self._total(loss)
            if overlap:
                self._dp_tail()
            elif self._use_dp():
                self.opt.gather_grads()
                self._allreduce_and_adam()
            else:
                self.opt.step_from_grads(bump=False)
                self._refresh_weights()
            self._log_parts(parts)
        self.steps += 1
        return loss.detach(), parts

    def _bump_counters(self, with_optimiser):
        """step_counter += 1 (and the optimiser's step count when this step applies Adam through step_from_grads) in one
        launch at the head of the step instead of one elementwise add here and one in the serial tail in front of Adam."""
        from . import _lib, hip as _hip
        if self.step_counter.is_cuda:
            _lib.call("msde_step_counters", _hip._p(self.step_counter), _hip._p(self.opt.step_dev if with_optimiser else None),
                      _hip._stream())
        else:
            self.step_counter.add_(1)
            if with_optimiser:
                self.opt.step_dev.add_(1)

    # ---- hipGraph path ---------------------------------------------------------------------------
    def _graph_body(self, batch, with_adam, finish=True):
        from . import hip as _hip
        _hip.stamp("step_start")
        self._bump_counters(with_adam)
        loss, parts = self.losses(batch, log=True, split_roots=True)
        self.opt.zero_grad()
        self._backward(loss, finish=finish)
        loss = self._total(loss)
        if with_adam:
            self.opt.step_from_grads(bump=False)
            self._refresh_weights()
        elif finish:
            self.opt.gather_grads()
        self._log_parts(parts)
        from . import hip as _hip
        _hip.stamp("step_end")
        return loss.detach()

    def capture(self, batch, pre=None):
        """Capture one step on `batch` (static shapes / addresses) into a hipGraph.  Call after at least
        one eager step on a batch of the same shape (sizes workspaces, sets kernel attributes).  `pre`: launches
        captured in front of the step (a bucket's device-side plan construction)."""
        key = id(batch)
        if key in self._graphs:
            return self._graphs[key][0]
        with_adam = not self._use_dp() and not self.adam_outside_graph
        sd = self.step_counter.view(torch.int64)
        self.models["SDE_2Dto3D_model"].score_network.seed_dev = sd
        if hasattr(self.noise, "seed_dev"):
            self.noise.seed_dev = sd        # fresh contrastive negatives on every replay
        self.opt.new_table_slot()       # this graph's own (pinned) gradient chunk table
        from . import hip as _hip
        slabs.new_param_grad_slot(batch.x.device)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        if self._graph_pool is None:
            self._graph_pool = torch.cuda.graph_pool_handle()
        # thread_local: other threads of the process (the RCCL watchdog under DP) may issue HIP calls
        # while this thread captures; they must not invalidate the capture
        # (round 4: capturing on a high-priority stream -- main chain over the second stream -- changes nothing: 2.575 vs 2.582 ms)
        overlap = (not with_adam) and self._use_dp() and DP_OVERLAP and self.dp_buckets
        tail_graphs = None
        with slabs.no_gc(), self._bounds(batch), torch.cuda.graph(g, pool=self._graph_pool, capture_error_mode="thread_local"):
            if pre is not None:
                pre()
            loss = self._graph_body(batch, with_adam, finish=not overlap)
        if overlap:
            # the data-parallel tail as one more captured piece per bucket (weight gradients of the bucket, their slab
            # reduction, its rows of the flat gradient buffer): step_graph() issues the bucket's all-reduce behind each piece
            # (_dp_tail).  Same memory pool, replayed in capture order.
            order = self._dp_order()
            slabs.partition_param_grad_batch(self.opt.grad_bucket_lookup(), len(self.opt.bucket_ranges))
            table = self.opt.grad_table()
            tail_graphs = []
            for k, i in enumerate(order):
                gk = torch.cuda.CUDAGraph()
                with slabs.no_gc(), self._bounds(batch), torch.cuda.graph(gk, pool=self._graph_pool, capture_error_mode="thread_local"):
                    self._dp_finish_bucket(i, table)
                    if k == len(order) - 1:
                        slabs.finish_param_grad_batch()
                        _hip.stamp("wgrad_end")
                tail_graphs.append(gk)
        slabs.flush_table_uploads()      # the graph's pointer tables: uploaded once, not at every replay
        self.opt.use_eager_slot()
        slabs.use_eager_param_grad_slot()
        # the captured batch is held strongly: its tensors' addresses are baked into the graph, and a live reference
        # keeps id(batch) from being recycled for a different batch
        self._graphs[key] = (g, batch, with_adam)
        self._graph_tails[key] = tail_graphs
        # the weight copies the captured refresh launch re-lays-out at every replay (wcache.weight_copies_after_replay)
        self._graph_wt_keys[key] = getattr(self, "_last_refreshed", ()) if with_adam else None
        self._graph_loss[key] = loss
        return g

    def step_graph(self, batch):
        """Replay the captured step; under DP the all-reduce and Adam run after the replay."""
        g, held, with_adam = self._graphs[id(batch)]
        assert held is batch
        from . import hip as _hip
        wcache.sync_weight_copies()          # parameters edited from outside since the last step (load_state_dict, ...)
        timing = self.dp_timing is not None and not with_adam and self._use_dp() and self.dp_buckets
        if timing:                         # [replay start, replay end, (bucket reduced, bucket's Adam done) x buckets]
            self._tm_cur = [self._timing_event()]
            self.dp_timing.append(self._tm_cur)
        g.replay()
        tails = self._graph_tails.get(id(batch))
        if timing and tails is None:
            self._tm_cur.append(self._timing_event())
        if self._graph_wt_keys.get(id(batch)) is not None:
            wcache.weight_copies_after_replay(self._graph_wt_keys[id(batch)])
        for bn in self._bn_modules:
            bn.pending_batches += 1        # the captured forward does not run Python: count its BatchNorm calls here
        if not with_adam:
            if self._use_dp():
                try:
                    if tails is not None:
                        self._dp_tail(graphs=tails)
                    else:
                        self._allreduce_and_adam()
                finally:
                    self._tm_cur = None
            else:
                for gk in tails or ():     # (captured under DP, replayed without it: the tail pieces still finish the gradients)
                    gk.replay()
                self.opt.step()
                self._refresh_weights()
        self.steps += 1
        return self._graph_loss[id(batch)]

    # ---- capacity buckets: ONE captured graph for every batch (moleculesde_amd/bucket.py) ------------------------
    def make_bucket(self, caps):
        from .bucket import Bucket
        return Bucket(caps, self.device)

    def capture_bucket(self, bk, warm_blob, eager_steps=2, split_plan=False):
        """Warm up on one raw batch (sizes the workspaces), then capture {device-side plan construction + step} for
        the bucket's capacities.  Afterwards every batch that fits the capacities is `step_bucket(bk, blob)`: one
        copy of its raw blob + one graph replay, no per-batch host work."""
        bk.load(warm_blob)
        bk.build_plan_on_device()
        ok, sizes = bk.check()
        if not ok:
            raise RuntimeError(f"batch does not fit the bucket: {sizes} vs {bk.caps.as_dict()}")
        for _ in range(eager_steps):
            self.step(bk.batch)
        if split_plan:
            # the plan construction as a graph of its own (BucketPipeline runs it for batch t+1 beside the step of batch t)
            torch.cuda.synchronize()
            bk.plan_graph = torch.cuda.CUDAGraph()
            from . import hip as _hipg
            with slabs.no_gc(), torch.cuda.graph(bk.plan_graph, capture_error_mode="thread_local"):
                bk.build_plan_on_device(None)
            return self.capture(bk.batch)
        side = self._side_stream if (self.overlap_streams and PLAN_LISTS_ON_SIDE) else None
        return self.capture(bk.batch, pre=lambda: bk.build_plan_on_device(side))

    def step_bucket(self, bk, blob):
        """One copy of the raw blob + one graph replay.  The device-side overflow flag of the PREVIOUS call is looked at
        here (one step late: no synchronisation of the running step); blobs made by bucket.pack_raw were already checked
        on the host, so the flag only fires for blobs packed some other way."""
        if bk.poll_overflow():
            from .bucket import BucketOverflow
            raise BucketOverflow("the batch of the previous step_bucket() call did not fit the bucket "
                                 f"{bk.caps.as_dict()}: its rows were left inert, that step is not a valid update")
        bk.load(blob)
        return self.step_graph(bk.batch)

    def step_stream(self, bk, batch):
        """One step on a collated HOST batch: through the bucket's captured graph when the batch fits its capacities,
        otherwise the exact-size eager step (host plan) -- the fallback for a batch larger than anything the bucket was
        sized for, or holding a molecule the device-side plan builder cannot take."""
        from . import bucket as BK
        need = BK.raw_sizes(batch)
        if bk.caps.fits(need) and need["n_max"] <= BK.PLAN_NMAX:
            return self.step_bucket(bk, BK.pack_raw(batch, bk.caps))
        out, _ = self.step(prepare_batch(batch.clone(), self.device))
        return out

    def state_dicts(self):
        """Checkpoint dictionary of pretrain_MoleculeSDE.py:78-88."""
        return {k: m.state_dict() for k, m in self.models.items()}

    def save(self, path):
        torch.save(self.state_dicts(), path)


class BucketPipeline:
    """Two buckets of the same capacities used alternately, so that batch construction leaves the step's critical path:
    while the captured step of batch t runs, the raw blob of batch t+1 is copied into the OTHER bucket and its plans
    (CSR views, extended graph, row lists: csrc/plan.hip) are built there by a captured plan graph on a third stream.  The
    step graph of a bucket then starts with the encoders' first kernels instead of ~100 us of plan construction.
        pipe = BucketPipeline(trainer, caps, warm_blob)
        pipe.submit(blob_0)
        for t in ...: pipe.submit(blob_{t+1}); loss = pipe.step()
    Events both ways, no host synchronisation: a bucket is refilled only after the step that read it has finished."""

    def __init__(self, trainer, caps, warm_blob):
        self.tr = trainer
        self.bks = [trainer.make_bucket(caps) for _ in range(2)]
        for bk in self.bks:
            trainer.capture_bucket(bk, warm_blob, split_plan=True)
        self.stream = shared_stream(trainer.device, "plan")
        self.ready = [None, None]       # recorded on the plan stream: bucket i holds a built batch
        self.done = [None, None]        # recorded on the compute stream: the step that read bucket i has finished
        self.head = self.tail = 0       # batches submitted / stepped

    def submit(self, blob):
        assert self.head - self.tail < 2, "BucketPipeline: two batches are already waiting"
        i = self.head & 1
        bk = self.bks[i]
        with torch.cuda.stream(self.stream):
            if self.done[i] is not None:
                self.stream.wait_event(self.done[i])
            bk.load(blob)
            bk.plan_graph.replay()
            ev = torch.cuda.Event()
            ev.record(self.stream)
            self.ready[i] = ev
        self.head += 1

    def step(self):
        assert self.tail < self.head, "BucketPipeline: nothing submitted"
        i = self.tail & 1
        bk = self.bks[i]
        cur = torch.cuda.current_stream()
        cur.wait_event(self.ready[i])
        if bk.poll_overflow():
            from .bucket import BucketOverflow
            raise BucketOverflow(f"a batch did not fit the bucket {bk.caps.as_dict()}: its rows were left inert, that "
                                 "step is not a valid update")
        loss = self.tr.step_graph(bk.batch)
        ev = torch.cuda.Event()
        ev.record(cur)
        self.done[i] = ev
        self.tail += 1
        return loss

    def check(self):
        return all(bk.check()[0] for bk in self.bks)


class StreamRunner:
    """The data path of the training loop (examples/pretrain_MoleculeSDE.py:125-156: `for step, batch in enumerate(loader)`)
    at graph-replay speed: ONE capacity bucket sized from a sample of the stream, ONE captured step graph (device-side
    batch construction + forward + backward + Adam), raw collated batches packed into pinned blobs and copied to the device
    one step ahead (bucket.BlobFeeder), one replay per batch -- no per-batch plan on the host, no per-kernel launch.  A
    batch that does not fit the capacities (or holds a molecule the device-side plan builder does not take) goes through
    the exact-size eager step instead; the per-step losses accumulate on the device (Trainer.log) and are read once per
    epoch.  `replay=False` launches the same bucket step from the host (the eager twin of the replayed step, for tests)."""

    def __init__(self, trainer, sample, n_max=None, replay=True):
        from . import bucket as BK
        self.BK, self.tr, self.replay = BK, trainer, replay
        needs = [BK.raw_sizes(b) for b in sample]
        self.caps = BK.Caps.covering(needs, n_max=n_max)
        self.bk = trainer.make_bucket(self.caps)
        trainer.capture_bucket(self.bk, BK.pack_raw(sample[0], self.caps).to(trainer.device))
        self._captured_coeffs = self._coeffs()
        self.feeder = BK.BlobFeeder(self.bk)
        self.fallbacks = 0

    def _coeffs(self):
        t = self.tr
        return (float(t.coeff_cl),)

    def ensure_captured(self):
        """The loss coefficients are launch arguments baked into the graph: after a change (the contrastive skip epochs of
        pretrain_MoleculeSDE.py:339-344) the step is captured again on the batch the bucket holds."""
        if self._coeffs() != self._captured_coeffs:
            # (the superseded graph is parked, not destroyed: freeing a graph of the shared memory pool while another capture
            # into that pool follows trips an allocator assertion in this torch build; it is a few hundred nodes)
            self._parked = getattr(self, "_parked", []) + [self.tr._graphs.pop(id(self.bk.batch), None)]
            self._parked.append(self.tr._graph_tails.pop(id(self.bk.batch), None))
            self.tr._graph_loss.pop(id(self.bk.batch), None)
            self.tr._graph_wt_keys.pop(id(self.bk.batch), None)
            self.tr.capture(self.bk.batch, pre=lambda: self.bk.build_plan_on_device(
                self.tr._side_stream if (self.tr.overlap_streams and PLAN_LISTS_ON_SIDE) else None))
            self._captured_coeffs = self._coeffs()

    def pack(self, batch, pin=True):
        """Pinned raw blob of a collated host batch, or None when it does not fit the bucket."""
        need = self.BK.raw_sizes(batch)
        if not self.caps.fits(need) or need["n_max"] > self.BK.PLAN_NMAX:
            return None
        return self.BK.pack_raw(batch, self.caps, pin=pin)

    def _step_loaded(self):
        bk = self.bk
        if bk.poll_overflow():
            raise self.BK.BucketOverflow(f"a batch did not fit the bucket {bk.caps.as_dict()}: that step is not a valid update")
        if self.replay:
            return self.tr.step_graph(bk.batch)
        bk.build_plan_on_device(None)
        return self.tr.step(bk.batch)[0]

    def prepare(self, batch, pin=True):
        """(host batch, its pinned raw blob or None when it does not fit the bucket): what run() consumes -- the host batch stays
        beside the blob so that a batch outside the capacities takes the exact-size eager step instead of being re-packed."""
        return (batch, self.pack(batch, pin=pin))

    def run(self, items, steps=None):
        """One pass over `items` -- collated host batches, (batch, blob-or-None) pairs from prepare(), or bare blobs already
        packed with pack(); `steps`: stop after that many, cycling over a list.  Returns the number of steps taken."""
        seq = list(items) if not isinstance(items, (list, tuple)) else items
        n = len(seq) if steps is None else steps
        if n == 0:
            return 0

        def item_of(i):
            it = seq[i % len(seq)]
            if torch.is_tensor(it):
                return None, it                     # a bare blob (the caller vouches that it fits)
            if isinstance(it, tuple) and len(it) == 2 and (it[1] is None or torch.is_tensor(it[1])):
                return it                           # prepare()'s pair
            if it is None:
                raise ValueError("StreamRunner.run: a None item (a batch that did not fit, packed without its host batch): "
                                 "pass prepare(batch) pairs or the host batches themselves")
            return it, self.pack(it)
        nxt = item_of(0)
        if nxt[1] is not None:
            self.feeder.submit(nxt[1])
        for i in range(n):
            cur = nxt
            nxt = item_of(i + 1) if i + 1 < n else (None, None)
            if nxt[1] is not None:
                self.feeder.submit(nxt[1])         # crosses PCIe while step i runs
            if cur[1] is None:                     # does not fit: exact-size eager step on a host-built plan
                self.fallbacks += 1
                self.tr.step(prepare_batch(cur[0].clone(), self.tr.device))
            else:
                self.feeder.load_next()
                self._step_loaded()
        return n


def train_epochs(args, trainer, runner, epoch_items, rank=0, out=print):
    """The epoch loop of examples/pretrain_MoleculeSDE.py:339-348 with its per-epoch report (:159-175) and best / final
    checkpoints (:168-170, 348).  epoch_items(epoch) -> (items, steps) for StreamRunner.run.  Returns the per-epoch means
    [(CL loss, CL acc, 2D->3D loss, 3D->2D loss, seconds, steps)]."""
    original = args.SDE_coeff_contrastive
    optimal, hist = 1e10, []
    for epoch in range(1, args.epochs + 1):
        trainer.coeff_cl = original if epoch > args.SDE_coeff_contrastive_skip_epochs else 0
        runner.ensure_captured()
        for k in trainer.log:
            trainer.log[k].zero_()
        torch.cuda.synchronize()
        t0 = time.time()
        items, steps = epoch_items(epoch)
        n = runner.run(items, steps)
        torch.cuda.synchronize()
        dt = time.time() - t0
        cl, acc, l23, l32 = (float(trainer.log[k]) / max(n, 1) for k in ("CL", "CL_acc", "2Dto3D", "3Dto2D"))
        hist.append((cl, acc, l23, l32, dt, n))
        if rank == 0:
            out("epoch: {}".format(epoch))
            out("CL Loss: {:.5f}\tCL Acc: {:.5f}\t\tSDE 2Dto3D Loss: {:.5f}\tSDE 3Dto2D Loss: {:.5f}".format(cl, acc, l23, l32))
            out("Time: {:.5f}\t({:.3f} ms/step, {} steps, {} fell back to the exact-size eager step)\n".format(
                dt, dt / max(n, 1) * 1e3, n, runner.fallbacks))
            temp = args.SDE_coeff_contrastive * cl + args.SDE_coeff_generative_2Dto3D * l23 + \
                args.SDE_coeff_generative_3Dto2D * l32
            if temp < optimal and args.output_model_dir:
                optimal = temp
                trainer.save(os.path.join(args.output_model_dir, "model_complete.pth"))
    if rank == 0 and args.output_model_dir:
        trainer.save(os.path.join(args.output_model_dir, "model_complete_final.pth"))
    return hist


def main(argv=None):
    """Counterpart of examples/pretrain_MoleculeSDE.py:178-348 on synthetic PCQM4Mv2-shaped data (no dataset in this
    image): the same flags, models, optimiser groups, epoch report and checkpoints; the batches of an epoch stream through
    StreamRunner (one captured graph, raw blobs prefetched) -- the mode bench.py's headline measures.  `--eager` keeps the
    per-kernel host launches on resident batches (the round-1 loop)."""
    from .synthetic import make_batch
    p = add_hot_path_flags(argparse.ArgumentParser(description=__doc__))
    p.add_argument("--steps_per_epoch", type=int, default=200)
    p.add_argument("--synthetic_pool", type=int, default=64, help="distinct synthetic batches an epoch cycles over")
    p.add_argument("--eager", action="store_true", help="launch every kernel from the host (no hipGraph, no bucket)")
    args = p.parse_args(argv)
    rank, world, local = dp.init_from_env("cuda")
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    torch.manual_seed(0)
    trainer = Trainer(args, device)
    cpu_pool = [make_batch(args.batch_size, seed=dp.shard_seed(s, rank)) for s in range(args.synthetic_pool)]
    if args.eager:
        pool = [prepare_batch(b, device) for b in cpu_pool[:4]]

        class _Eager:
            fallbacks = 0
            def ensure_captured(self): pass
            def run(self, items, steps):
                for s in range(steps):
                    trainer.step(items[s % len(items)])
                return steps
        return train_epochs(args, trainer, _Eager(), lambda e: (pool, args.steps_per_epoch), rank)
    runner = StreamRunner(trainer, cpu_pool)
    blobs = [runner.prepare(b) for b in cpu_pool]
    return train_epochs(args, trainer, runner, lambda e: (blobs, args.steps_per_epoch), rank)


if __name__ == "__main__":
    main()
