"""Library-GEMM algorithm selection for the forward / input-gradient products (the weight gradients use the
hand-written split-M kernel).

torch's default picks the first heuristic result of hipBLASLt for every shape; PyTorch's TunableOp can instead look
up, per GEMM signature, the fastest of the rocBLAS / hipBLASLt solutions.  `tuning/tunableop_gfx950.csv` holds that
choice for the GEMM signatures of the pretrain step on the synthetic PCQM4Mv2-shaped batches of bench.py (ranks 0-7,
pool of 4, bs 256, with and without the 3D->2D head), produced by tools/tune_gemms.py on an MI355X.  It is only
READ here (no tuning at run time); signatures that are not in the file, or a file written by another library
version (its validator lines are checked by torch), fall back to the default algorithm.  MSDE_TUNED_GEMMS=0 turns
the lookup off.  Measured: 78.5k -> 82.5k molecules/s on the bench step."""
import os

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuning", "tunableop_gfx950.csv")
_state = {"done": False, "ok": False}


def enable(tuning=False, path=PATH, scratch="/tmp/msde_tunableop_scratch.csv"):
    """Idempotent.  tuning=True (tools/tune_gemms.py) also benchmarks unseen signatures on first use."""
    import torch
    if _state["done"] and not tuning:
        return _state["ok"]
    _state["done"] = True
    if os.environ.get("MSDE_TUNED_GEMMS", "1") == "0" or not torch.cuda.is_available():
        return False
    T = torch.cuda.tunable
    T.enable(True)
    T.tuning_enable(bool(tuning))
    T.set_filename(scratch)             # where torch may write at exit: never into the repository
    ok = False
    if os.path.exists(path):
        try:
            ok = bool(T.read_file(path))
        except Exception:
            ok = False
    _state["ok"] = ok
    return ok
