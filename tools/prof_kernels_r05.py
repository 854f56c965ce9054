"""Launches for the round-5 rocprofv3 `--pmc` passes (SEPARATE passes per counter set, MI355X_MICROARCH.md), bs-256 synthetic
batch (seed 0).  mode "gemm": the two dominant gemm_t2 instantiations of the step, 20 launches each, one shape each --
  <5,0,0> plain product 3588 x 300 x 300 (SchNet / GIN node-level Linear layers), <5,2,0> the BatchNorm-backward product
  3588 x 600 x 300 of a GIN layer (A = p g + w z + u on the fragments, ReLU gate and statistics in the epilogue);
mode "step": ten eager --full training steps: gemm_grouped_wgrad_kernel, the dense_edge_layer_* kernels of the 3D->2D head
and (with --score_kernel mol) the escore_mol_* kernels are picked out of them by name (tools/pmc_summary_r05.py)."""
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
else:
    tr = pretrain.Trainer(pretrain.readme_args(SDE_coeff_generative_3Dto2D=1, score_kernel="mol"), dev)
    bt = prepare_batch(make_batch(256, seed=0), dev)
    for _ in range(10):
        tr.step(bt)
    torch.cuda.synchronize()
    print("step passes done")
