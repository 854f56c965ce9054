"""LDS canaries on a second stream while node-level products run on the first: does a product kernel write LDS it does not own?"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moleculesde_amd import hip
dev = torch.device("cuda", 0)
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "liblds_canary.so"))
lib.lds_canary_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
KIND = sys.argv[1] if len(sys.argv) > 1 else "t2b"
torch.manual_seed(0)
shapes = [(3442, 128, 64), (3442, 32, 64), (3442, 64, 128), (3442, 64, 32), (3588, 300, 300)]
ops = []
for (M, N, K) in shapes:
    A = torch.randn(M, K, device=dev); W = torch.nn.Parameter(torch.randn(N, K, device=dev) / K ** 0.5); b = torch.randn(N, device=dev)
    ops.append((A, W, b, torch.empty(M, N, device=dev), hip.weight_planes(W, False)))
side = torch.cuda.Stream()
bad = torch.zeros(1, dtype=torch.int32, device=dev); first = torch.zeros(64, dtype=torch.int32, device=dev)
SHORT = len(sys.argv) > 2 and sys.argv[2] == "short"      # many short-lived canaries: they move into LDS a product workgroup just left
for lds_kb in (16, 48, 64):
    bad.zero_(); torch.cuda.synchronize()
    with torch.cuda.stream(side):
        if SHORT:
            lib.lds_canary_launch(600000, lds_kb * 1024, 6, bad.data_ptr(), first.data_ptr(), ctypes.c_void_p(side.cuda_stream))
        else:
            lib.lds_canary_launch(1024, lds_kb * 1024, 3000, bad.data_ptr(), first.data_ptr(), ctypes.c_void_p(side.cuda_stream))
    n = 0
    while not side.query():
        for (A, W, b, out, (p, ld)) in ops:
            if KIND == "t2b":
                hip.gemm_rs(A, p, out, bias=b, N=W.size(0), K=W.size(1), t2b_ld=ld)
            elif KIND == "t2":
                hip.gemm_rs(A, W.detach(), out, bias=b, N=W.size(0), K=W.size(1), t2=True)
        n += 1
        if n % 50 == 0:
            torch.cuda.current_stream().synchronize()
    torch.cuda.synchronize()
    print(KIND, "canary LDS %d KB: product rounds beside it %d, corrupted words %d" % (lds_kb, n, int(bad)),
          first[:16].view(4, 4).tolist() if int(bad) else "", flush=True)
