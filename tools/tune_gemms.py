"""Produce moleculesde_amd/tuning/tunableop_gfx950.csv: run eager pretrain steps with PyTorch TunableOp tuning ON
over the synthetic batches bench.py uses (ranks 0..7 x pool of 4, bs 256, without and with the 3D->2D head) so that
every library-GEMM signature of those steps gets its fastest rocBLAS / hipBLASLt solution recorded.  GPU box only;
takes a few minutes.  Usage: python tools/tune_gemms.py [out.csv]"""
import os, sys, shutil, glob
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moleculesde_amd import tuned_gemm, pretrain, dp
from moleculesde_amd.geom3d import prepare_batch
from moleculesde_amd.synthetic import make_batch

out = sys.argv[1] if len(sys.argv) > 1 else tuned_gemm.PATH
scratch = "/tmp/msde_tune_run.csv"
for f in glob.glob("/tmp/msde_tune_run*.csv"):
    os.remove(f)
tuned_gemm.enable(tuning=True, scratch=scratch)
T = torch.cuda.tunable
T.set_max_tuning_duration(50)
dev = torch.device("cuda", 0)
ranks = int(os.environ.get("MSDE_TUNE_RANKS", "8"))
for full in (False, True):
    torch.manual_seed(0)
    args = pretrain.readme_args() if full else pretrain.readme_args(SDE_coeff_generative_3Dto2D=0)
    tr = pretrain.Trainer(args, dev)
    for rank in range(ranks):
        for s in range(4):
            b = prepare_batch(make_batch(256, seed=dp.shard_seed(s, rank)), dev)
            tr.step(b)
        torch.cuda.synchronize()
        print(f"tuned: full={full} rank={rank} entries={len(T.get_results())}", flush=True)
    del tr
T.write_file() if hasattr(T, "write_file") else None
torch.cuda.synchronize()
res = T.get_results()
val = T.get_validators()
os.makedirs(os.path.dirname(out), exist_ok=True)
with open(out, "w") as f:
    for k, v in val:
        f.write(f"Validator,{k},{v}\n")
    for r in res:
        f.write(",".join(str(x) for x in r) + "\n")
print("wrote", out, len(res), "entries")
