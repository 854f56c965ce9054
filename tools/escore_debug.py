import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_escore as T
dev = torch.device("cuda", 0)
B, seed = int(sys.argv[1]), int(sys.argv[2])
cpu_b, pl, ep, net, x, ea, basis = T._case(dev, B, seed)
net.eval()
xd, ed, bd = x.to(dev), ea.to(dev), basis.to(dev)
w = torch.randn(ep.N, 3).to(dev)
o1, gx1, ge1, gp1 = T._grads(net, ep, pl, xd, ed, bd, w, True)
o2, gx2, ge2, gp2 = T._grads(net, ep, pl, xd, ed, bd, w, False)
mp = pl.mol_ptr.cpu().tolist(); rp = ep.rowptr.cpu().tolist()
for m in range(B):
    n0, n1 = mp[m], mp[m + 1]; e0, e1 = rp[n0], rp[n1]
    d = (ge1[e0:e1] - ge2[e0:e1]).abs()
    bad_rows = (d.max(1).values > 1e-4).nonzero().flatten().tolist()
    print(f"mol {m}: n={n1 - n0} Em={e1 - e0} gea err {float(d.max()):.2e} gx err {float((gx1[n0:n1] - gx2[n0:n1]).abs().max()):.2e} bad rows {bad_rows[:12]}{'...' if len(bad_rows) > 12 else ''} ({len(bad_rows)})")
    if bad_rows:
        r = bad_rows[0]
        print("   cols bad in first bad row:", (d[r] > 1e-4).nonzero().flatten().tolist())
for k in gp2:
    e = float((gp1[k] - gp2[k]).abs().max()); s = float(gp2[k].abs().max())
    if e > 1e-4 * max(s, 1e-3):
        print("param", k, "err", e, "scale", s)
