"""Stand-alone time of the one-workgroup-per-molecule score network (csrc/escore_mol.hip) at the sampler's shape (10 x 14
atoms) and at the training batch (256 molecules), beside the operator-by-operator path; HIP events over back-to-back launches."""
import os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import moleculesde_amd.geom3d as G
from moleculesde_amd import plan as P, escore
from moleculesde_amd.geom3d import sde_2d_to_3d as M
from moleculesde_amd.batch import Batch
from moleculesde_amd.synthetic import make_batch, make_molecule
from moleculesde_amd import slabs  # noqa: E402
dev = torch.device("cuda", 0)
torch.manual_seed(0)


def timeit(fn, iters=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def case(name, cpu_b, train):
    b = G.prepare_batch(cpu_b.clone(), dev)
    pl = P.get_plan(b)
    ep = pl.ext
    net = M.EquivariantScoreNetwork(32, hidden_coff_dim=128).to(dev)
    net.train(train)
    x, ea, bs = torch.randn(ep.N, 32, device=dev), torch.randn(ep.E, 32, device=dev), torch.randn(ep.E, 9, device=dev)
    with torch.no_grad():
        t_mol = timeit(lambda: net(ep, x, ea, bs, pl))
        M.MOL_KERNEL = False
        t_ops = timeit(lambda: net(ep, x, ea, bs, pl), 50)
        M.MOL_KERNEL = True
    g = torch.cuda.CUDAGraph()
    with torch.no_grad():
        net(ep, x, ea, bs, pl)
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(20):
                net(ep, x, ea, bs, pl)
    t_graph = timeit(g.replay, 20) / 20
    print(f"{name}: N={ep.N} E={ep.E} B={pl.B} train={train}: mol kernel {t_mol:.1f} us (20 in a graph: {t_graph:.1f} us each), "
          f"operator path {t_ops:.1f} us (host-launched)")


mol = make_molecule(np.random.default_rng(0), 14)
case("sampler 10x14", Batch.from_data_list([mol] * 10), False)
case("batch 256", make_batch(256, 0), False)
case("batch 256", make_batch(256, 0), True)


def case_bwd(name, cpu_b):
    """forward (training variant) + backward, graph-timed, mol kernels vs operator path"""
    from moleculesde_amd import hip
    b = G.prepare_batch(cpu_b.clone(), dev)
    pl = P.get_plan(b)
    ep = pl.ext
    net = M.EquivariantScoreNetwork(32, hidden_coff_dim=128).to(dev).train()
    x = torch.randn(ep.N, 32, device=dev, requires_grad=True)
    ea = torch.randn(ep.E, 32, device=dev, requires_grad=True)
    bs = torch.randn(ep.E, 9, device=dev)
    w = torch.randn(ep.N, 3, device=dev)
    res = {}
    for mol in (True, False):
        M.MOL_KERNEL_TRAIN = mol

        def step():
            net.zero_grad(set_to_none=True)
            x.grad = None
            ea.grad = None
            slabs.begin_param_grad_batch()
            out = net(ep, x, ea, bs, pl)["gradient"]
            (out * w).sum().backward()
            slabs.finish_param_grad_batch()
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            slabs.new_param_grad_slot(dev)
            with torch.cuda.graph(g, stream=s):
                step()
        res[mol] = timeit(g.replay, 50)
    M.MOL_KERNEL_TRAIN = False
    print(f"{name}: fwd+bwd+weight-gradient reduction in a graph: mol kernels {res[True]:.1f} us, operator path {res[False]:.1f} us")


case_bwd("batch 256", make_batch(256, 0))
case_bwd("10x14", Batch.from_data_list([mol] * 10))
