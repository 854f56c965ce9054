"""2D->3D reverse-SDE predictor-corrector sampler (BASELINE.json config 4), as INTENDED by
examples/pretrain_MoleculeSDE_inference_2D_to_3D_VE_VP.py:92-212.  The script itself is broken as
shipped (missing import, undefined args, hard-coded `break` after 11 steps); this
restates the algorithm:  for t in linspace(T, eps, N):  Langevin corrector -> reverse-diffusion
predictor, each calling SDEModel2Dto3D_02.get_score on the replicated molecule batch."""
import os
import torch
from . import _lib
from . import hip as _hip_mod
from . import slabs

FUSED_PC = True     # sampler arithmetic on msde_pc_corrector / msde_pc_predictor (False: operator by operator, the reference path of the tests)


def predictor_update(sde, score_model, representation, data, pos, t, noise=None):
    """ReverseDiffusionPredictor.update_fn (:163-168)."""
    f, G = sde.reverse_discretize(score_model, pos, representation, data, t)
    noise = torch.randn_like(pos) if noise is None else noise
    x_mean = pos - f
    return x_mean + G[:, None] * noise, x_mean


def corrector_update(sde, score_model, representation, data, pos, t, snr, scale_eps, n_steps, noises=None):
    """LangevinCorrector.update_fn (:191-212): alpha = 1 for the VE SDE, alphas[timestep] for VP.  As in the script, the inner
    iterations do not feed `pos` back (App. B.2), so the score is evaluated once per call when
    n_steps == 1 and only the last pass matters otherwise."""
    alpha = sde.corrector_alpha(t) if hasattr(sde, "corrector_alpha") else torch.ones_like(t)     # VP: :198-200
    x = x_mean = pos
    for i in range(n_steps):
        grad = score_model.get_score(representation, data, pos, None, t)
        noise = torch.randn_like(pos) if noises is None else noises[i]
        grad_norm = torch.norm(grad.reshape(grad.shape[0], -1), dim=-1).mean()
        noise_norm = torch.norm(noise.reshape(noise.shape[0], -1), dim=-1).mean()
        step_size = (snr * noise_norm / grad_norm) ** 2 * 2 * alpha
        x_mean = pos + step_size[:, None] * grad
        x = x_mean + torch.sqrt(step_size * 2)[:, None] * noise * scale_eps
    return x, x_mean


@torch.no_grad()
def position_PC_generation(score_model, representation, data, num_steps=1000, snr=0.16, scale_eps=0.7,
                           n_corrector_steps=1, eps=1e-4, denoise=True, pos_init=None, use_graph=True, noise_seed=None,
                           torch_noise=False, iters_per_graph=1):
    """position_PC_generation (:92-138): returns the final coordinates [N, 3].

    The loop is latency bound (every iteration = 1 + n_corrector_steps score-network calls on a graph of
    ~10 x 14 atoms, ~150 small launches), so by default one iteration (corrector + predictor, noise drawn
    inside) is captured into a hipGraph after two eager warm-up iterations and replayed.  On the fused path (one shared
    diffusion time, one corrector step) the replay needs nothing from the host: the iteration counter and the noise live in
    the two update kernels (noise_seed: their seed; None = drawn from torch's generator), and iters_per_graph iterations
    share one graph launch.  torch_noise=True: the noise comes
    from torch.randn_like as in the operator path (parity tests: same generator state => same trajectory)."""
    sde = score_model.sde_pos
    n = representation.size(0)
    dev = representation.device
    pos = (torch.randn(n, 3, device=dev) if pos_init is None else pos_init.clone()).contiguous()
    timesteps = torch.linspace(sde.T, eps, num_steps, device=dev)
    vec_t = torch.ones(n, device=dev)
    x_mean = pos.clone()

    # Fused arithmetic (csrc/pointwise.hip: msde_pc_corrector / msde_pc_predictor): every atom shares the diffusion time, so
    # std(t), G(t), alpha(t) and the drift factor are tabulated ONCE for all time steps with the SDE's own methods, and each
    # half iteration is: score network -> one noise draw -> one kernel.  The operator-by-operator functions above stay the
    # reference path (tests, n_corrector_steps > 1, CPU).
    fused = (FUSED_PC and dev.type == "cuda" and n_corrector_steps == 1 and hasattr(score_model, "get_score_raw")
             and pos.dtype == torch.float32)
    if fused:
        S = num_steps
        ones = torch.ones(S, 1, device=dev)
        std_all = sde.marGINal_prob(ones, timesteps)[1]
        f_all, G_all = sde.discretize(ones, timesteps)
        alpha_all = sde.corrector_alpha(timesteps) if hasattr(sde, "corrector_alpha") else torch.ones_like(timesteps)
        par_all = torch.stack([std_all.float(), G_all.float(), alpha_all.float(), f_all[:, 0].float() + 1.0], 1).contiguous()
        # the iteration counter lives on the device (the corrector advances it) and both kernels draw their own noise from the
        # counter generator: a replayed iteration needs no host work and no random-number operator (noise_seed: one per call)
        it_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        # default seed: ONE draw from the device's torch generator per call (so torch.manual_seed / torch.cuda.manual_seed
        # govern the trajectory as they did when the noise came from torch.randn_like on the device; one host read per
        # trajectory, outside the iteration loop); an explicit noise_seed is taken modulo 2^64
        if noise_seed is not None:
            seed = int(noise_seed) & 0xFFFFFFFFFFFFFFFF
        else:       # (torch_noise: the kernels get their noise from torch.randn_like -- no draw, the generator is left alone)
            seed = 0 if torch_noise else int(torch.randint(0, 2 ** 62, (1,), device=dev).item())
        xc = torch.empty_like(pos)
        xm_c = torch.empty_like(pos)
        stream = _hip_mod._stream
        p_ = _hip_mod._p

    def one_step():
        if fused:
            raw = score_model.get_score_raw(representation, data, pos).contiguous()
            nz = torch.randn_like(pos) if torch_noise else None
            _lib.call("msde_pc_corrector", p_(raw), p_(pos), p_(nz), p_(par_all), p_(it_dev), seed, n, float(snr), float(scale_eps),
                      p_(xc), p_(xm_c), stream())
            raw2 = score_model.get_score_raw(representation, data, xc).contiguous()
            nz2 = torch.randn_like(pos) if torch_noise else None
            _lib.call("msde_pc_predictor", p_(raw2), p_(xc), p_(nz2), p_(par_all), p_(it_dev), seed, n, p_(pos), p_(x_mean), stream())
            return
        p, _ = corrector_update(sde, score_model, representation, data, pos, vec_t, snr, scale_eps, n_corrector_steps)
        p, m = predictor_update(sde, score_model, representation, data, p, vec_t)
        pos.copy_(p)
        x_mean.copy_(m)

    # Replay: on the fused path an iteration reads nothing from the host (counter and noise live on the device), so
    # `iters_per_graph` consecutive iterations can share ONE hipGraph; the tail that does not fill a graph replays a
    # one-iteration graph.  Measured (tools/sampler_slope.py): the loop is bound by the two score-network launches of an
    # iteration, 165-169 us per iteration for 1, 5 and 25 iterations per graph -- so the default stays 1 (cheapest capture).
    graph, graph_k, k_iters = None, None, 1
    i = 0
    while i < num_steps:
        if not fused:
            vec_t.fill_(1.0).mul_(timesteps[i])
        if use_graph and dev.type == "cuda" and graph is None and i == 2:
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with slabs.no_gc(collect=False), torch.cuda.graph(graph):
                one_step()                 # capture only records; the replay below executes iteration i
            k_iters = max(1, min(int(iters_per_graph), num_steps - i)) if fused else 1
            if k_iters > 1:
                graph_k = torch.cuda.CUDAGraph()
                with slabs.no_gc(collect=False), torch.cuda.graph(graph_k):
                    for _ in range(k_iters):
                        one_step()
        if graph_k is not None and num_steps - i >= k_iters:
            graph_k.replay()
            i += k_iters
        elif graph is not None:
            graph.replay()
            i += 1
        else:
            one_step()
            i += 1
    return (x_mean if denoise else pos).clone()
