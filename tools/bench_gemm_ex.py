"""Time msde_gemm_ex (csrc/gemm_ex.hip) against the library GEMM and the round-1 kernel (msde_linear_fwd) on the
dense shapes of the step.  HIP events over back-to-back launches; TFLOP/s against the 157.3 fp32 matrix peak."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moleculesde_amd import hip, _lib

dev = torch.device("cuda", 0)
SHAPES = [(3588, 728, 364, False), (3588, 728, 728, False), (3588, 119, 728, False), (3588, 728, 728, True),
          (3588, 300, 300, False), (3588, 600, 300, False), (3588, 300, 600, False), (3588, 128, 300, False),
          (3588, 300, 128, False), (52680, 60, 32, False), (52680, 60, 60, False), (35186, 32, 300, False),
          (49090, 128, 128, False), (3588, 640, 16, False)]


def timeit(fn, reps=20, replays=5):
    """GPU time per call: `reps` back-to-back launches captured in a hipGraph (no host launch overhead), replayed."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(replays):
        g.replay()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / (reps * replays) * 1e3


if __name__ == "__main__":
  for M, N, K, km in SHAPES:
      A = torch.randn(M, K, device=dev)
      W = torch.randn(N, K, device=dev) / K ** 0.5
      Wk = W.t().contiguous()
      b = torch.randn(N, device=dev)
      out = torch.empty(M, N, device=dev)
      fl = 2.0 * M * N * K
      t_ex = timeit(lambda: hip.gemm_ex(A, Wk if km else W, out, bias=b, act="silu", b_kmajor=km))
      t_lib = timeit(lambda: torch.addmm(b, A, W.t(), out=out))
      p, st = hip._p, hip._stream()
      if not km:
          t_old = timeit(lambda: _lib.call("msde_linear_fwd", p(A), p(W), p(b), M, N, K, p(out), st))
      else:
          G = torch.randn(M, N, device=dev)        # dgrad-shaped: gX[M,K'] = gY[M,N] W[N,K'] with K' = N here
          t_old = timeit(lambda: _lib.call("msde_linear_bwd_x", p(A), p(Wk), M, K, N, p(out), st))
      print(f"M={M:6d} N={N:4d} K={K:4d} km={int(km)}  gemm_ex {t_ex:7.1f} us ({fl / t_ex / 1e6:6.1f} TF)   "
            f"library {t_lib:7.1f} us ({fl / t_lib / 1e6:6.1f} TF)   round-1 kernel {t_old:7.1f} us ({fl / t_old / 1e6:6.1f} TF)",
            flush=True)
