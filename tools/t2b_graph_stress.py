"""Two-branch hipGraph of node-level products (the trainer's structure: GIN on the capture stream, SchNet on a second stream):
are the outputs of the second branch identical at every replay?  KIND = t2b | t2 for the kernels of both branches."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moleculesde_amd import hip

dev = torch.device("cuda", 0)
torch.manual_seed(0)
KIND = sys.argv[1] if len(sys.argv) > 1 else "t2b"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 550
F = int(sys.argv[3]) if len(sys.argv) > 3 else 64


def lin(n, k):
    return torch.nn.Parameter(torch.randn(n, k, device=dev) / k ** 0.5), torch.randn(n, device=dev) * 0.1


side_w = [lin(F, F) for _ in range(12)]
main_w = [lin(2 * F, F) if i % 2 == 0 else lin(F, 2 * F) for i in range(12)]
x_side = torch.randn(M, F, device=dev)
x_main = torch.randn(M, F, device=dev)


def prod(x, w, b, act):
    n, k = w.shape
    out = torch.empty(x.size(0), n, device=dev)
    if KIND == "t2b":
        p, ld = hip.weight_planes(w, False)
        hip.gemm_rs(x, p, out, bias=b, act=act, N=n, K=k, t2b_ld=ld)
    else:
        hip.gemm_rs(x, w.detach(), out, bias=b, act=act, N=n, K=k, t2=True)
    return out


def body(side):
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        h = x_side
        outs = []
        for w, b in side_w:
            h = prod(h, w, b, "ssp")
            outs.append(h)
    g = x_main
    for w, b in main_w:
        g = prod(g, w, b, "relu")
    main.wait_stream(side)
    return outs, g


side = torch.cuda.Stream()
body(side); body(side)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    outs, g = body(side)
graph.replay(); torch.cuda.synchronize()
ref = [o.clone() for o in outs]; gref = g.clone()
bad = 0
for it in range(400):
    graph.replay()
    torch.cuda.synchronize()
    for i, (o, r) in enumerate(zip(outs, ref)):
        if not torch.equal(o, r):
            bad += 1
            if bad <= 8:
                d = (o - r).abs()
                print("replay", it, "side product", i, "differs: max %.3g" % float(d.max()), "elements", int((d > 0).sum()),
                      "rows", sorted(set((d > 0).nonzero()[:, 0].tolist()))[:12], "cols", sorted(set((d > 0).nonzero()[:, 1].tolist()))[:12], flush=True)
            break
    if not torch.equal(g, gref):
        bad += 1
        if bad <= 8:
            print("replay", it, "main chain differs", flush=True)
print(KIND, "M", M, "F", F, "replays with a difference:", bad, "of 400")
