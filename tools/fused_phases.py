"""Per-phase cycle stamps of the fused CFConv forward (needs a -DCF_TIMING=1 build of libmsde_hip.so)."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moleculesde_amd import hip, plan as P, pretrain
from moleculesde_amd.geom3d import prepare_batch
from moleculesde_amd.synthetic import make_batch
dev = torch.device("cuda", 0)
torch.manual_seed(0)
args = pretrain.readme_args(SDE_coeff_generative_3Dto2D=0)
tr = pretrain.Trainer(args, dev)
b = prepare_batch(make_batch(256, seed=0), dev)
sch = tr.models["model_3D"]; pl = P.get_plan(b); blk = sch.interactions[0]; de = sch.distance_expansion
cpw = int(sys.argv[1]) if len(sys.argv) > 1 else 3
with torch.no_grad():
    rplan, dist = hip.radius_plan(b.positions, pl.batch_i32, pl.mol_ptr, sch.cutoff, pl.E_r_cap, 32)
    x1 = torch.randn(b.x.size(0), 128, device=dev)
    for _ in range(3):
        agg, Wf = hip.cfconv_fused_forward(x1, dist, rplan, blk.mlp[0].weight, blk.mlp[0].bias, blk.mlp[2].weight,
                                           blk.mlp[2].bias, de.offset, de.coeff, sch.cutoff, chunks_per_wg=cpw, want_filter=True)
    torch.cuda.synchronize()
    st = Wf.view(torch.int64).cpu().numpy().reshape(-1)
nblk = (pl.E_r_cap + 31) // 32
nblk = (nblk + cpw - 1) // cpw
names = ["start", "prologue", "P0+B1", "gathers+GEMM1", "ssp+B2", "GEMM2", "epi2(x wait)", "segsum+store", "produce_math"]
rows = []
for blkid in range(nblk - 1):
    s = st[blkid * 64: blkid * 64 + 2 + 7 * cpw]
    rows.append(np.diff(s))
rows = np.array(rows)
print("blocks", rows.shape[0], "stamps/unit: s_memtime ticks (100 MHz constant clock on gfx9: 1 tick = 10 ns)" )
print(f"{'prologue':>16}: mean {rows[:,0].mean():8.1f}  max {rows[:,0].max()}")
per = rows[:, 1:].reshape(rows.shape[0], cpw, 7)
for k in range(7):
    print(f"{names[k+2]:>16}: mean {per[:,:,k].mean():8.1f}  max {per[:,:,k].max()}")
print("total per block mean", rows.sum(1).mean(), "max", rows.sum(1).max())
starts = st[np.arange(nblk - 1) * 64]
print("block start spread:", starts.max() - starts.min())
wall = st[np.arange(nblk - 1) * 64 + 63] - st[np.arange(nblk - 1) * 64 + 62]
print("wall_clock64 ticks per block mean", wall.mean(), "-> cycles per wall tick", rows.sum(1).mean() / wall.mean(), "(wall clock is 100 MHz => shader MHz =", 100 * rows.sum(1).mean() / wall.mean(), ")")
