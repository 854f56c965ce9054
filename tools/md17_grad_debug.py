"""MD17 step: parameter gradients with the weight gradients deferred into the grouped launch vs computed layer by layer."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import moleculesde_amd.geom3d as G
from moleculesde_amd import hip, dd
from moleculesde_amd.synthetic import make_md17_batch
from moleculesde_amd.finetune_md17 import ForceTrainer
from moleculesde_amd import slabs  # noqa: E402
dev = torch.device("cuda", 0)
torch.manual_seed(0)
kw = dict(hidden_channels=300, num_filters=128, num_interactions=6, num_gaussians=51, cutoff=10, readout="mean", node_class=119)
cpu_b = make_md17_batch(2, seed=3, n_atoms=21)
sch, head = G.SchNet(**kw).to(dev), torch.nn.Linear(300, 1).to(dev)
b = G.prepare_batch(cpu_b.clone(), dev)
ft = ForceTrainer(sch, head, lr=5e-4, energy_coeff=1.0, force_coeff=1.0)
et, ftg = torch.randn(2, device=dev), torch.randn(42, 3, device=dev)
names = [n for n, _ in sch.named_parameters()] + ["head." + n for n, _ in head.named_parameters()]
params = list(sch.parameters()) + list(head.parameters())

def grads(defer, skip):
    for p in params:
        p.grad = None
    pos = b.positions.detach().requires_grad_(True)
    if skip:
        energy, force = ft.energy_and_force(b, pos)
    else:
        rep = sch(b.x, pos, b.batch)
        energy = head(rep).squeeze(1)
        force = -torch.autograd.grad(energy, pos, grad_outputs=torch.ones_like(energy), create_graph=True, retain_graph=True)[0]
    loss = (energy - et).abs().mean() + (force - ftg).abs().mean()
    if defer:
        slabs.begin_param_grad_batch(params)
    try:
        loss.backward()
    finally:
        if defer:
            slabs.finish_param_grad_batch()
    torch.cuda.synchronize()
    return [None if p.grad is None else p.grad.clone() for p in params]

ref = grads(False, False)
for label, d, s in (("skip only", False, True), ("defer only", True, False), ("both", True, True)):
    g = grads(d, s)
    worst = []
    for n, a, r in zip(names, g, ref):
        if (a is None) != (r is None):
            worst.append((float("inf"), n, "missing"))
            continue
        if r is None:
            continue
        e = float((a - r).abs().max()) / (float(r.abs().max()) + 1e-30)
        worst.append((e, n, tuple(r.shape)))
    worst.sort(reverse=True)
    print(label, worst[:6])
