"""Shared helpers for the parity tests: golden loading, state-dict / batch reconstruction."""
import os

import numpy as np
import torch

from moleculesde_amd.batch import Batch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def record_parity(line):
    """Achieved parity numbers of a GPU test run, kept: appended to gpurun_out/parity_numbers.txt (scratch that gpurun merges
    back; the round's copy is committed as profiles/rNN_parity_numbers.txt) and printed."""
    print(line)
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_numbers.txt"), "a") as f:
            f.write(line.rstrip("\n") + "\n")
    except OSError:
        pass


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name)) as z:
        return {k: z[k] for k in z.files}


def sub(g, prefix):
    """Tensors of all entries with the given prefix, prefix stripped."""
    return {k[len(prefix):]: torch.from_numpy(np.asarray(v)) for k, v in g.items() if k.startswith(prefix)}


def batch_from(g, prefix="batch."):
    d = sub(g, prefix)
    ng = int(d.pop("num_graphs"))
    b = Batch(**d)
    b.num_graphs = ng
    return b


def disable_dropout(model):
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if hasattr(m, "dropout") and isinstance(getattr(m, "dropout"), float):
            m.dropout = 0.0
    return model


def assert_close(a, b, rtol, atol, what=""):
    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = err > tol
    assert not bool(bad.any()), (
        f"{what}: {int(bad.sum())}/{a.numel()} mismatches, max abs err {err.max().item():.3e}, "
        f"ref scale {b.abs().max().item():.3e}")


def grads_close(model, ggrads, rtol, atol_scale, what=""):
    """Compare parameter grads to golden grads; atol is scaled by the largest grad magnitude of the
    whole model (gradients that are analytically zero -- bias before BatchNorm, key bias under
    softmax -- are pure rounding noise on both sides)."""
    scale = max(float(np.abs(v).max()) for v in ggrads.values()) if ggrads else 1.0
    for n, p in model.named_parameters():
        if n not in ggrads:
            continue
        assert p.grad is not None, f"{what}: no grad for {n}"
        assert_close(p.grad, ggrads[n], rtol, atol_scale * scale, f"{what}:{n}")
