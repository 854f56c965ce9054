"""Bitwise reproducibility of the captured two-stream pretrain step.

The node-level products of the step run on v_mfma_f32_16x16x4_f32 issued from inline assembly with hand-counted waits and
hand-padded hazards (csrc/gemm_t2.h), beside the kernels of a second stream.  Round 4's bf16x3 experiment showed what a
hazard in such code looks like: one register of 16 consecutive lanes of a CO-RESIDENT kernel changes about one replay in six.
This test is the check that caught it, pointed at the default fp32 step: configs[1] (GIN + SchNet + contrastive + 2D->3D VE,
256 molecules, emb_dim 300), captured as one hipGraph over both streams, replayed 16 times from IDENTICAL state (parameters,
Adam moments, step counters restored before every replay) -- every gradient must be bit-identical across the replays."""
import pytest
import torch
from moleculesde_amd import wcache  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("full", [False, True], ids=["configs1", "configs2_per_gpu"])
def test_captured_two_stream_step_replays_bit_identically(full):
    assert torch.cuda.is_available()
    import moleculesde_amd.geom3d as G
    from moleculesde_amd import hip, pretrain
    from moleculesde_amd.synthetic import make_batch
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    tr = pretrain.Trainer(pretrain.readme_args(SDE_coeff_generative_3Dto2D=1 if full else 0), dev)
    assert tr.overlap_streams and tr._side_stream is not None, "the default step runs on two streams"
    b = G.prepare_batch(make_batch(256, seed=17), dev)
    for _ in range(3):
        tr.step(b)
    tr.capture(b)
    tr.step_graph(b)
    torch.cuda.synchronize()
    state = (tr.opt.flat_p, tr.opt.m, tr.opt.v, tr.opt.step_dev, tr.step_counter)
    snap = [t.clone() for t in state]
    params = [(mk + "." + n, p) for mk in tr.models for n, p in tr.models[mk].named_parameters() if p.requires_grad]
    first, loss0 = None, None
    for it in range(16):
        with torch.no_grad():
            for t, s0 in zip(state, snap):
                t.copy_(s0)
        wcache.invalidate_weight_copies()             # parameters rewritten behind the optimiser's back
        loss = tr.step_graph(b)
        torch.cuda.synchronize()
        grads = {n: p.grad.clone() for n, p in params if p.grad is not None}
        if first is None:
            first, loss0 = grads, loss.clone()
            assert len(first) > 100 and all(torch.isfinite(g).all() for g in first.values())
            continue
        assert torch.equal(loss, loss0), f"replay {it}: loss {float(loss)} vs {float(loss0)}"
        bad = [n for n in first if not torch.equal(first[n], grads[n])]
        assert not bad, f"replay {it}: {len(bad)} gradients differ from replay 0, e.g. {bad[:5]}"


def test_head_loss_as_its_own_backward_root_equals_one_root():
    """configs[2]'s per-GPU step: the 3D->2D head's loss differentiated as a second root on the second stream
    (pretrain.SPLIT_HEAD_ROOT, the main stream starts its backward without waiting for the head's forward) gives the same
    update as the single composed loss of pretrain_MoleculeSDE.py:128-152 -- the same kernels on the same operands, and
    SchNet's output gradient is a sum of two contributions either way: parameters and Adam moments bit-equal after three steps,
    captured and replayed included."""
    assert torch.cuda.is_available()
    import moleculesde_amd.geom3d as G
    from moleculesde_amd import pretrain
    from moleculesde_amd.synthetic import make_batch
    dev = torch.device("cuda", 0)
    b = G.prepare_batch(make_batch(64, seed=23), dev)
    runs = []
    old = pretrain.SPLIT_HEAD_ROOT
    try:
        for split in (True, False):
            pretrain.SPLIT_HEAD_ROOT = split
            torch.manual_seed(0)
            tr = pretrain.Trainer(pretrain.readme_args(SDE_coeff_generative_3Dto2D=1), dev)
            assert tr.overlap_streams
            losses = [tr.step(b)[0].clone() for _ in range(2)]
            tr.capture(b)
            losses.append(tr.step_graph(b).clone())
            torch.cuda.synchronize()
            runs.append((losses, tr.opt.flat_p.clone(), tr.opt.m.clone(), tr.opt.v.clone()))
    finally:
        pretrain.SPLIT_HEAD_ROOT = old
    (l1, p1, m1, v1), (l0, p0, m0, v0) = runs
    for a, c in zip(l1, l0):
        assert abs(float(a) - float(c)) <= 1e-6 * abs(float(c)), (float(a), float(c))      # (total = main + head: one more addition)
    assert torch.equal(p1, p0) and torch.equal(m1, m0) and torch.equal(v1, v0)
