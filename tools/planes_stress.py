"""Do the bf16 planes a captured graph refreshes always reach the product that reads them in the NEXT launch?
graph A: W += D (a parameter update), refresh of every cached copy (one msde_transpose_multi launch);
graph B: the bf16x3 product and the fp32 product on the same activations.  A stale plane shows as an O(1) difference."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moleculesde_amd import hip

dev = torch.device("cuda", 0)
torch.manual_seed(0)
shapes = [(550, 64, 64), (550, 128, 64), (3588, 300, 300), (550, 64, 192), (3588, 128, 300), (700, 300, 128)]
Ws = [torch.nn.Parameter(torch.randn(N, K, device=dev) / K ** 0.5) for (_, N, K) in shapes]
As = [torch.randn(M, K, device=dev) for (M, _, K) in shapes]
Ds = [torch.randn_like(w) * 0.5 for w in Ws]
o3 = [torch.empty(M, N, device=dev) for (M, N, _) in shapes]
o32 = [torch.empty(M, N, device=dev) for (M, N, _) in shapes]
o3t = [torch.empty(M, K, device=dev) for (M, _, K) in shapes]
o32t = [torch.empty(M, K, device=dev) for (M, _, K) in shapes]
Gs = [torch.randn(M, N, device=dev) for (M, N, _) in shapes]


def update():
    with torch.no_grad():
        for w, d in zip(Ws, Ds):
            w.data.add_(d)
            d.neg_()
    hip.bump_weight_epoch()
    hip.refresh_weight_t()


def products():
    for (M, N, K), w, a, g, x3, x32, y3, y32 in zip(shapes, Ws, As, Gs, o3, o32, o3t, o32t):
        p, ld = hip.weight_planes(w, False)
        hip.gemm_rs(a, p, x3, N=N, K=K, t2b_ld=ld)
        hip.gemm_rs(a, w.detach(), x32, N=N, K=K, t2=True)
        if hip.t2_ok(M, K, N):
            pt, ldt = hip.weight_planes(w, True)
            hip.gemm_rs(g, pt, y3, N=K, K=N, t2b_ld=ldt)
            hip.gemm_rs(g, hip.weight_t(w), y32, N=K, K=N, t2=True)
            torch.maximum(ERR[1], (y3 - y32).abs().max(), out=ERR[1])
        torch.maximum(ERR[0], (x3 - x32).abs().max(), out=ERR[0])


ERR = torch.zeros(2, device=dev)
products(); update(); products(); update()
torch.cuda.synchronize()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    gA, gB = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    hip.note_capture()
    with torch.cuda.graph(gA, stream=s):
        update()
    with torch.cuda.graph(gB, stream=s):
        products()
    hip.flush_table_uploads()
    ERR.zero_()
    for it in range(3000):
        gA.replay()
        gB.replay()
    torch.cuda.synchronize()
    print("no host sync between launches: max |bf16x3 - fp32| fwd %.3g dgrad %.3g" % (float(ERR[0]), float(ERR[1])))
