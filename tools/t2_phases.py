"""Per-workgroup phase times of msde_gemm_t2 from in-kernel cycle stamps (library built with -DT2_TIMING: the stamps go to
the buffer passed as xf4; in the real kernel no stamp executes).  Prints, per configuration, the median cycles of prologue
(entry -> first tile landed), K loop and epilogue, the in-kernel clock, and the spread of workgroup start / end times."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moleculesde_amd import hip

dev = torch.device("cuda", 0)


def run(M, N, K, rt=0, S=0, reps=30):
    A = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev)
    dbg = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
    for _ in range(reps):      # warm clocks
        hip.gemm_rs(A, W, out, bias=b, t2=True, rt=rt, splits=S)
    dbg.zero_()
    hip.gemm_rs(A, W, out, bias=b, t2=True, rt=rt, splits=S, xf=(None, None, None, None, dbg))
    torch.cuda.synchronize()
    t = dbg.view(-1, 8).cpu()
    t = t[t[:, 0] != 0]
    pro, loop, epi = (t[:, 1] - t[:, 0]).double(), (t[:, 2] - t[:, 1]).double(), (t[:, 3] - t[:, 2]).double()
    real = (t[:, 7] - t[:, 6]).double() / 100.0        # us (100 MHz)
    clk = ((t[:, 3] - t[:, 0]).double() / real).median().item() / 1e3
    start = (t[:, 6] - t[:, 6].min()).double() / 100.0
    end = (t[:, 7] - t[:, 6].min()).double() / 100.0
    print(f"M={M} N={N} K={K} rt={rt} S={S}: wgs {t.size(0)} | cycles median: prologue {pro.median():.0f} loop {loop.median():.0f} "
          f"(max {loop.max():.0f}) epilogue {epi.median():.0f} | clock {clk:.2f} GHz | wg time {real.median():.2f} us "
          f"| start spread {start.max():.2f} us, last end {end.max():.2f} us", flush=True)


if __name__ == "__main__":
    for (M, N, K) in [(3588, 300, 600), (3588, 600, 300)]:
        for abl in (0, 1, 2, 3, 4, 8, 12, 7, 15, 16, 24, 31):
            run(M, N, K, rt=abl, S=4)
