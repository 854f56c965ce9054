"""CPU tests (no GPU): host-side logic of the package + the C-ABI library loads and exports every
symbol include/msde_hip.h declares (no compute calls)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_loads_and_exports_every_declared_symbol():
    from moleculesde_amd import _lib, build
    lib_path = build.build(verbose=False)
    assert os.path.exists(lib_path)
    header = open(os.path.join(ROOT, "include", "msde_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)     # comments may sit inside prototypes
    declared = set(re.findall(r"\b(msde_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 20
    lib = ctypes.CDLL(lib_path)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/msde_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    # argument counts in the ctypes table match the header prototypes
    protos = re.findall(r"(?:long long|int|const char\*)\s+(msde_[a-z0-9_]+)\s*\(([^;]*?)\);", header, flags=re.S)
    assert len(protos) == len(declared)
    for name, args in protos:
        args = args.strip()
        n = 0 if args in ("", "void") else len(args.split(","))
        assert n == len(_lib.SIGNATURES[name]), (name, n, len(_lib.SIGNATURES[name]))
    assert _lib.load().msde_abi_version() == 1


def test_product_path_has_no_cpu_fallback():
    """A CPU tensor handed to a kernel wrapper raises instead of silently computing on the host,
    and nothing in the package imports the oracle."""
    from moleculesde_amd import hip, _lib
    with pytest.raises(_lib.MsdeHipError):
        hip._f32(torch.zeros(3))
    pkg = os.path.join(ROOT, "moleculesde_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_library_records_no_memset_nodes():
    """The library zero-fills with a kernel (csrc/msde_common.h: msde_zero_words), never with hipMemsetAsync: the memset NODE in
    front of the per-edge CFConv forward kernel did not take effect in the first replay of a captured step that followed an eager
    step (round 5, tools/replay_growth_debug2.py), and its partial-tile atomics landed on stale data."""
    csrc = os.path.join(ROOT, "moleculesde_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h")):
            code = re.sub(r"//[^\n]*", "", open(os.path.join(csrc, f)).read())
            assert "hipMemset" not in code, f


def test_grouped_wgrad_problem_rows_name_their_tile_shape():
    """msde_linear_bwd_w_describe_ld is host code (it fills one row of the grouped launch's problem table): the tile shape it picks
    for a layer -- outputs are N x K -- and the number of workgroups it reports follow the rule in include/msde_hip.h: 64 x 64;
    N <= 32 < 64 < K: 32 x 128; K <= 32 < 64 < N: 128 x 32; both <= 32: 32 x 32 with the K tile split over the waves; operands off
    the 16-byte vector path keep 64 x 64.  The splits are the shape-only plan of msde_linear_bwd_w_splits in every case."""
    import ctypes
    from moleculesde_amd import _lib
    lib = _lib.load()
    row = (ctypes.c_longlong * 16)()
    base = 1 << 20                       # fake 16-byte aligned device addresses: nothing is dereferenced here
    ptr = lambda off=0: ctypes.c_void_p(base + off)
    ceil = lambda a, b: (a + b - 1) // b
    cases = [(3588, 300, 600, 0), (35186, 32, 300, 1), (70372, 32, 128, 1), (35186, 128, 32, 2), (35186, 32, 72, 1),
             (35186, 32, 64, 0), (35186, 32, 32, 3), (3588, 16, 364, 1), (3588, 16, 16, 3), (52680, 16, 32, 3), (2049, 300, 8, 2),
             (52680, 60, 30, 0), (52680, 1, 60, 0)]
    for M, N, K, shape in cases:
        n = lib.msde_linear_bwd_w_describe_ld(ptr(), N, ptr(1 << 16), K, M, N, K, 1, ptr(1 << 18), None, row)
        splits = lib.msde_linear_bwd_w_splits(M, N, K)
        bm, bn = {0: (64, 64), 1: (32, 128), 2: (128, 32), 3: (32, 32)}[shape]
        assert row[15] >> 1 == shape, (M, N, K, row[15])
        assert (row[9], row[10], row[7]) == (ceil(K, bn), ceil(N, bm), splits), (M, N, K, list(row))
        assert n == ceil(K, bn) * ceil(N, bm) * splits
    # an operand that is not 16-byte aligned leaves the vector path: 64 x 64 whatever the widths
    lib.msde_linear_bwd_w_describe_ld(ptr(4), 32, ptr(1 << 16), 300, 35186, 32, 300, 1, ptr(1 << 18), None, row)
    assert row[15] >> 1 == 0 and row[11] == 0


def test_extend_graph_path_graph_known_answer():
    """dataset_3D.py:12-35 on a 7-node path: pairs within <= 4 bonds, no self loops, sorted."""
    from moleculesde_amd.batch import extend_graph_index
    n = 7
    src = list(range(n - 1)) + list(range(1, n))
    dst = list(range(1, n)) + list(range(n - 1))
    ext = extend_graph_index(torch.tensor([src, dst]), n)
    want = [(i, j) for i in range(n) for j in range(n) if i != j and abs(i - j) <= 4]
    assert list(map(tuple, ext.t().tolist())) == want


def test_extend_graph_matches_reference_algorithm_when_available():
    """The reference's extend_graph sequence (dataset_3D.py:18-34) replayed on the stand-in
    spspmm/coalesce (the dataset module itself imports rdkit and cannot be loaded)."""
    from oracle import ref_loader
    if not ref_loader.available():
        pytest.skip("/root/reference not present")
    ref_loader.verbatim()
    import torch_sparse
    from torch_geometric.utils import remove_self_loops
    from moleculesde_amd.batch import extend_graph_index
    from moleculesde_amd.synthetic import make_molecule
    rng = np.random.default_rng(0)
    for _ in range(5):
        m = make_molecule(rng)
        N = m.x.size(0)
        ei = m.edge_index
        val = ei.new_ones((ei.size(1),), dtype=torch.float)
        idx, v = torch_sparse.spspmm(ei, val, ei, val, N, N, N)
        idx, v = remove_self_loops(idx, v)
        e2, _ = torch_sparse.coalesce(torch.cat([ei, idx], 1), None, N, N)
        val = e2.new_ones((e2.size(1),), dtype=torch.float)
        idx, v = torch_sparse.spspmm(e2, val, e2, val, N, N, N)
        idx, v = remove_self_loops(idx, v)
        e4, _ = torch_sparse.coalesce(torch.cat([e2, idx], 1), None, N, N)
        assert torch.equal(e4, extend_graph_index(ei, N))


def test_collate_rules_and_plan():
    from moleculesde_amd import plan as P
    from moleculesde_amd.synthetic import make_batch, batch_stats
    b = make_batch(16, seed=3)
    st = batch_stats(b)
    assert st["B"] == 16 and st["N"] == b.x.size(0)
    assert int(b.edge_index.max()) < st["N"] and int(b.extended_edge_index.max()) < st["N"]
    assert torch.equal(b.batch, torch.sort(b.batch)[0])
    # edges never cross molecules
    assert torch.equal(b.batch[b.edge_index[0]], b.batch[b.edge_index[1]])
    assert torch.equal(b.batch[b.extended_edge_index[0]], b.batch[b.extended_edge_index[1]])
    pl = P.build_plan(b)
    assert pl.E_r_cap == st["sum_n2"] - st["N"]
    for plan, ei in ((pl.bond, b.edge_index), (pl.ext, b.extended_edge_index)):
        N, E = st["N"], ei.size(1)
        assert plan.rowptr[0] == 0 and plan.rowptr[-1] == E and plan.rowptr_s[-1] == E
        assert torch.equal(plan.dst.long(), torch.sort(ei[1], stable=True)[0])
        assert torch.equal(ei[:, plan.perm_t], torch.stack([plan.src.long(), plan.dst.long()]))
        deg = torch.bincount(ei[1], minlength=N)
        assert torch.equal((plan.rowptr[1:] - plan.rowptr[:-1]).long(), deg)
        assert torch.equal(plan.src.long()[plan.perm_s.long()], torch.sort(ei[0], stable=True)[0])
    # embedding row lists: every (node, column) appears exactly once under its table row
    assert pl.atom_list_nodes.numel() == st["N"] * 9
    for r in range(pl.atom_R):
        nodes = pl.atom_list_nodes[pl.atom_list_ptr[r]:pl.atom_list_ptr[r + 1]].long()
        if nodes.numel():
            assert bool((pl.atom_codes[nodes] == r).any(dim=1).all())
    # bond codes follow the canonical order and the table offsets [0, 5, 11]
    ea = b.edge_attr[pl.bond.perm_t]
    assert torch.equal(pl.bond_codes.long(), ea + torch.tensor([0, 5, 11]))


def test_synthetic_generator_shape_statistics():
    """SURVEY §8d recipe: counts land within a few percent of the survey's probe."""
    from moleculesde_amd.synthetic import make_batch, batch_stats, make_qm9_batch
    st = batch_stats(make_batch(256, seed=0))
    assert st["N_max"] <= 20 and 3300 < st["N"] < 3800
    assert 6800 < st["E_b"] < 7700 and 31000 < st["E_e"] < 38000 and 43000 < st["E_r"] < 53000
    q = make_qm9_batch(32, seed=0)
    assert q.x.dim() == 1 and q.num_graphs == 32 and int(torch.bincount(q.batch).max()) <= 29


def test_cpu_replay_noise_matches_global_seed_program_order():
    from moleculesde_amd.geom3d.nn import CpuReplayNoise
    torch.manual_seed(77)
    a = torch.randn_like(torch.zeros(9, 3))
    b = torch.randint(0, 1000, size=(5,))
    c = torch.randperm(13)
    n = CpuReplayNoise(77)
    assert torch.equal(n.randn_like(torch.zeros(9, 3)), a)
    assert torch.equal(n.randint(1000, (5,), "cpu"), b)
    assert torch.equal(n.randperm(13, "cpu"), c)


def test_sde_definitions_match_oracle():
    from moleculesde_amd.geom3d.sde import VESDE, VPSDE
    from oracle import restate as R
    t = torch.tensor([1e-6, 0.25, 0.5, 0.999, 1.0])
    x = torch.randn(5, 3)
    a, b = VESDE(0.2, 1.0, 1000), R.VESDE(0.2, 1.0, 1000)
    assert torch.allclose(a.marGINal_prob(x, t)[1], b.marGINal_prob(x, t)[1])
    assert torch.allclose(a.discretize(x, t)[1], b.discretize(x, t)[1])
    assert torch.allclose(a.sde(x, t)[1], b.sde(x, t)[1])
    pa, pb = VPSDE(0.1, 20, 1000), R.VPSDE(0.1, 20, 1000)
    for u, v in zip(pa.marGINal_prob(x, t), pb.marGINal_prob(x, t)):
        assert torch.allclose(u, v)


def test_state_dict_surface_matches_reference_inventory():
    import json
    import moleculesde_amd.geom3d as G
    inv = json.load(open(os.path.join(ROOT, "tests", "golden", "inventory.json")))
    m = {"model_2D": G.GNN(5, 300, JK="last", drop_ratio=0, gnn_type="GIN"),
         "model_3D": G.SchNet(hidden_channels=300, num_filters=128, num_interactions=6, num_gaussians=51, cutoff=10,
                              readout="mean", node_class=119),
         "SDE_2Dto3D_model": G.SDEModel2Dto3D_02(emb_dim=300, hidden_dim=32, beta_min=0.2, beta_max=1.0,
                                                 num_diffusion_timesteps=1000, beta_schedule=None, SDE_type="VE",
                                                 use_extend_graph=True),
         "SDE_3Dto2D_model": G.SDEModel3Dto2D_node_adj_dense(
             dim3D=300, c_init=2, c_hid=8, c_final=4, num_heads=4, adim=16, nhid=16, num_layers=4, emb_dim=300,
             num_linears=3, beta_min=0.1, beta_max=1.0, num_diffusion_timesteps=1000, SDE_type="VE", num_class_X=119,
             noise_on_one_hot=True)}
    assert set(m) == set(inv)
    for k, mod in m.items():
        sd = mod.state_dict()
        assert list(sd.keys()) == list(inv[k]["state_dict"].keys()), k
        for n, (shape, dtype) in inv[k]["state_dict"].items():
            assert list(sd[n].shape) == shape and str(sd[n].dtype) == dtype, (k, n)
        assert sum(p.numel() for p in mod.parameters() if p.requires_grad) == inv[k]["trainable"]


def test_sde3d2d_02_surface_and_builder():
    """§8 f3: SDEModel3Dto2D_node_adj_dense_02 has the state-dict keys of the fixture the reference's class produced
    (tests/golden/f3_sde3d2d_02.npz), its score networks take 2 * dim3D features, and --SDE_3Dto2D_model selects it."""
    import numpy as np
    import moleculesde_amd.geom3d as G
    from moleculesde_amd import pretrain
    from moleculesde_amd.geom3d import sde_3d_to_2d as S
    g = np.load(os.path.join(ROOT, "tests", "golden", "f3_sde3d2d_02.npz"))
    m = G.SDEModel3Dto2D_node_adj_dense_02(dim3D=8, c_init=2, c_hid=8, c_final=4, num_heads=4, adim=16, nhid=16,
                                           num_layers=4, emb_dim=8, num_linears=3, beta_min=0.1, beta_max=1.0,
                                           num_diffusion_timesteps=1000, SDE_type="VE", num_class_X=119,
                                           noise_on_one_hot=True)
    assert [k for k, _ in m.named_parameters()] == list(g["param_names"])
    base = G.SDEModel3Dto2D_node_adj_dense(dim3D=8, c_init=2, c_hid=8, c_final=4, num_heads=4, adim=16, nhid=16,
                                           num_layers=4, emb_dim=8, num_linears=3, beta_min=0.1, beta_max=1.0,
                                           num_diffusion_timesteps=1000, SDE_type="VE", num_class_X=119,
                                           noise_on_one_hot=True)
    assert list(m.state_dict().keys()) == list(base.state_dict().keys())
    wide = [k for k in m.state_dict() if m.state_dict()[k].shape != base.state_dict()[k].shape]
    assert wide and all("score_network" in k for k in wide)          # only the networks' input layers grow
    a = pretrain.readme_args(SDE_3Dto2D_model="SDEModel3Dto2D_node_adj_dense_02", emb_dim=16)
    assert isinstance(S.build_from_args(a), G.SDEModel3Dto2D_node_adj_dense_02)


def test_painn_surface_and_builder():
    """§8 f4: PaiNN has the parameter names of the fixture the reference's class produced (tests/golden/f4_painn.npz), the
    reference's buffers, an output head like painn.py:208-216, and --model_3d PaiNN selects it."""
    import numpy as np
    import moleculesde_amd.geom3d as G
    from moleculesde_amd import pretrain
    g = np.load(os.path.join(ROOT, "tests", "golden", "f4_painn.npz"))
    m = G.PaiNN(n_atom_basis=32, n_interactions=3, n_rbf=20, cutoff=3.5, max_z=119, n_out=1, readout="mean")
    assert [k for k, _ in m.named_parameters()] == list(g["param_names"])
    assert {k for k, _ in m.named_buffers()} == {"cutoff_fn.cutoff", "radial_basis.widths", "radial_basis.offsets"}
    assert float(m.embedding.weight[0].abs().max()) == 0.0                # padding_idx = 0
    head = m.create_output_layers()
    assert [tuple(l.weight.shape) for l in head] == [(16, 32), (1, 16)]   # build_mlp: 32 -> 16 -> 1
    a = pretrain.readme_args(model_3d="PaiNN", emb_dim=16, SDE_coeff_generative_3Dto2D=0)
    models = pretrain.build_models(a, torch.device("cpu"))
    assert isinstance(models["model_3D"], G.PaiNN) and models["model_3D"].n_interactions == 3
    with pytest.raises(RuntimeError):                                     # no CPU fallback
        m(torch.zeros(3, dtype=torch.long), torch.zeros(3, 3), torch.zeros(2, 0, dtype=torch.long), torch.zeros(3, dtype=torch.long))


def test_readme_args_and_flag_defaults():
    from moleculesde_amd import pretrain
    a = pretrain.readme_args()
    assert a.batch_size == 256 and a.emb_dim == 300 and a.gnn_3d_lr_scale == 0.1 and a.dropout_ratio == 0
    assert a.SDE_2Dto3D_model == "SDEModel2Dto3D_02" and a.use_extend_graph and a.T == 0.1
    assert a.SchNet_num_gaussians == 51 and a.SchNet_num_filters == 128 and a.SchNet_num_interactions == 6


_DP_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from moleculesde_amd import dp
rank, world, local = dp.init_from_env("cpu")
assert world == 2
torch.manual_seed(0)
flat = torch.arange(10, dtype=torch.float32) * (rank + 1)
dp.broadcast_flat(flat, 0)
assert torch.equal(flat, torch.arange(10, dtype=torch.float32))
g = torch.full((1000,), float(rank + 1))
scale = dp.allreduce_mean_(g)
assert scale == 0.5 and torch.equal(g, torch.full((1000,), 3.0))
# DP semantics: mean of per-rank gradients == gradient of the mean loss over the union of shards
w = torch.nn.Parameter(torch.ones(4))
x = torch.arange(8, dtype=torch.float32).view(2, 4)[rank]
(w * x).sum().backward()
gg = w.grad.clone(); s = dp.allreduce_mean_(gg); gg *= s
full = torch.arange(8, dtype=torch.float32).view(2, 4).mean(0)
assert torch.allclose(gg, full)
assert dp.shard_seed(5, 0) != dp.shard_seed(5, 1)
dp.barrier()
dist.destroy_process_group()
print("OK", rank)
"""


def test_data_parallel_gloo_world_size_2(tmp_path):
    script = tmp_path / "dp_worker.py"
    script.write_text(_DP_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", "29533", str(script), ROOT]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("OK") == 2


def test_bench_gpus_n_starts_its_own_ranks():
    """`python bench.py --gpus 2` (no torchrun environment) must become a launcher of 2 rank processes, not a 1-rank run
    labelled 2.  Runs here without a GPU: `--census_only` stops after the rank count, gloo backend."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["MSDE_DP_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--census_only"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["rccl_ranks_seen"] == 2 and len(j["rank_devices"]) == 2
    # the keys a `--gpus N` line carries beside the contract's own (round 6: configs[2]'s full step timed by the same ranks,
    # the pieces of a DP step, the per-rank spread); bench.main() asserts the same tuple on the line it prints
    assert {"config2_full", "dp_step_parts_us", "ms_per_step_per_rank"} <= set(j["dp_line_keys"])
    assert {"graph_us", "allreduce_us", "adam_us"} <= set(j["dp_part_keys"])
    # without the smoke backend and without 2 devices the launcher refuses instead of mislabelling a 1-GPU run
    if torch.cuda.device_count() < 2:
        env.pop("MSDE_DP_BACKEND")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--census_only"], env=env,
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == 2 and "device(s) visible" in r.stderr
