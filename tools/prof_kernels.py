"""Launches for the rocprofv3 `--pmc` passes (SEPARATE passes per counter set, MI355X_MICROARCH.md), bs-256 synthetic
batch (seed 0).  mode "gemm": the two dominant gemm_t2 instantiations of the step, 20 launches each, one shape each --
  <5,0,0> plain product 3588 x 300 x 300 (SchNet / GIN node-level Linear layers), <5,2,0> the BatchNorm-backward product
  3588 x 600 x 300 of a GIN layer (A = p g + w z + u on the fragments, ReLU gate and statistics in the epilogue);
mode "step": ten eager --full training steps: gemm_grouped_wgrad_kernel, the dense_edge_layer_* kernels of the 3D->2D head
and (with --score_kernel mol) the escore_mol_* kernels are picked out of them by name (tools/pmc_summary.py); round 6: so are the
all-blocks CFConv launches (cfconv_pair_filter_multi_kernel, cfconv_pair_bwd_w_multi_kernel)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moleculesde_amd import hip, pretrain
from moleculesde_amd.geom3d import prepare_batch
from moleculesde_amd.synthetic import make_batch
dev = torch.device("cuda", 0)
torch.manual_seed(0)
mode = sys.argv[1] if len(sys.argv) > 1 else "gemm"
REP = 20
if mode == "gemm":
    with torch.no_grad():
        N, D, H = 3588, 300, 600
        x = torch.randn(N, D, device=dev); W = torch.nn.Parameter(torch.randn(D, D, device=dev) / D ** 0.5); b = torch.randn(D, device=dev)
        out = torch.empty(N, D, device=dev)
        for _ in range(REP):
            hip.gemm_fwd(x, W, out, bias=b)
        torch.cuda.synchronize()
        # BatchNorm-backward product of _GinMlpBN.backward: ga1 = bnbwd(g, z2) W2, gated by relu'(a1), statistics for BatchNorm 1
        g = torch.randn(N, D, device=dev); z2 = torch.randn(N, D, device=dev); a1 = torch.randn(N, H, device=dev)
        z1 = torch.randn(N, H, device=dev); W2 = torch.nn.Parameter(torch.randn(D, H, device=dev) / H ** 0.5)
        pw = [torch.randn(D, device=dev) for _ in range(3)]
        mean1 = torch.randn(H, device=dev)
        sa, _ = hip.rs_geometry(N, H, D)
        sta = torch.empty(sa, 2, H, device=dev); dz2 = torch.empty(N, D, device=dev); ga1 = torch.empty(N, H, device=dev)
        for _ in range(REP):
            hip.gemm_node(g, W2, ga1, False, H, D, axf="bnbwd", xf=(pw[0], pw[1], pw[2], None, None), A2=z2, A_out=dz2,
                          act="relu", dact_from=a1, stats=sta, stats_mode="bnbwd", stats_z=z1, stats_mean=mean1)
        torch.cuda.synchronize()
    print("gemm passes done")
elif mode == "infer":
    # the latency-bound configurations: 30 predictor-corrector iterations of the 2D->3D sampler (BASELINE configs[3]:
    # escore_edge_pre_kernel + escore_mol_fwd_kernel<false, true>) and 5 MD17 force fine-tuning steps (configs[4]: gemm_small_kernel)
    import numpy as np
    import moleculesde_amd.geom3d as G
    from moleculesde_amd import sampler
    from moleculesde_amd.batch import Batch
    from moleculesde_amd.finetune_md17 import ForceTrainer
    from moleculesde_amd.synthetic import make_molecule, make_md17_batch
    mol = make_molecule(np.random.default_rng(0), 14)
    b = G.prepare_batch(Batch.from_data_list([mol] * 10), dev)
    gnn = G.GNN(5, 300, JK="last", drop_ratio=0, gnn_type="GIN").to(dev).eval()
    s23 = G.SDEModel2Dto3D_02(emb_dim=300, hidden_dim=32, beta_min=0.2, beta_max=1.0, num_diffusion_timesteps=1000,
                              beta_schedule=None, SDE_type="VE", use_extend_graph=True).to(dev).eval()
    with torch.no_grad():
        rep = gnn(b.x, b.edge_index, b.edge_attr)
    sampler.position_PC_generation(s23, rep, b, num_steps=30, use_graph=False)
    kw = dict(hidden_channels=300, num_filters=128, num_interactions=6, num_gaussians=51, cutoff=10, readout="mean", node_class=119)
    sch, head = G.SchNet(**kw).to(dev), torch.nn.Linear(300, 1).to(dev)
    mb = G.prepare_batch(make_md17_batch(1, seed=3, n_atoms=21), dev)
    ft = ForceTrainer(sch, head, lr=5e-4, energy_coeff=1.0, force_coeff=1.0)
    et, ftg = torch.randn(1, device=dev), torch.randn(21, 3, device=dev)
    for _ in range(5):
        ft.step(mb, et, ftg)
    torch.cuda.synchronize()
    print("inference passes done")
else:
    tr = pretrain.Trainer(pretrain.readme_args(SDE_coeff_generative_3Dto2D=1, score_kernel="mol"), dev)
    bt = prepare_batch(make_batch(256, seed=0), dev)
    for _ in range(10):
        tr.step(bt)
    torch.cuda.synchronize()
    print("step passes done")
