"""VE / VP SDE definitions used by the score models (marginal std, SMLD/DDPM discretisation,
reverse-time rule).  Restates SDE_sparse.py:64-102,105-162,172-222 and SDE_dense.py (same maths,
trailing-dimension broadcasting differs) as two small closed-form classes."""
import math

import torch


class VESDE:
    def __init__(self, sigma_min=0.01, sigma_max=50, N=1000):
        self.sigma_min, self.sigma_max, self.N = float(sigma_min), float(sigma_max), int(N)
        self.discrete_sigmas = torch.exp(torch.linspace(math.log(self.sigma_min), math.log(self.sigma_max), self.N))
        self._sig_dev = {}

    T = 1

    def _sigmas(self, device):
        s = self._sig_dev.get(device)
        if s is None:
            s = self.discrete_sigmas.to(device)
            self._sig_dev[device] = s
        return s

    def sigma(self, t):
        return self.sigma_min * (self.sigma_max / self.sigma_min) ** t

    def sde(self, x, t):
        diffusion = self.sigma(t) * math.sqrt(2 * (math.log(self.sigma_max) - math.log(self.sigma_min)))
        return torch.zeros_like(x), diffusion

    def marGINal_prob(self, x, t):          # (sic) the reference's spelling is the API
        return x, self.sigma(t)

    def prior_sampling(self, shape):
        return torch.randn(*shape)

    def discretize(self, x, t):
        sig = self._sigmas(t.device)
        timestep = (t * (self.N - 1) / self.T).long()
        sigma = sig[timestep]
        adjacent = torch.where(timestep == 0, torch.zeros_like(t), sig[(timestep - 1).clamp(min=0)])
        return torch.zeros_like(x), torch.sqrt(sigma ** 2 - adjacent ** 2)

    def reverse_discretize(self, score_model, x, representation, data, t, probability_flow=False):
        """RSDE.discretize (SDE_sparse.py:94-100)."""
        f, G = self.discretize(x, t)
        score = score_model.get_score(representation, data, x, None, t)
        rev_f = f - G[:, None] ** 2 * score * (0.5 if probability_flow else 1.0)
        return rev_f, (torch.zeros_like(G) if probability_flow else G)


class VPSDE:
    def __init__(self, beta_min=0.1, beta_max=20, N=1000):
        self.beta_0, self.beta_1, self.N = float(beta_min), float(beta_max), int(N)
        self.discrete_betas = torch.linspace(self.beta_0 / self.N, self.beta_1 / self.N, self.N)
        self.alphas = 1.0 - self.discrete_betas

    T = 1

    def sde(self, x, t):
        beta_t = self.beta_0 + t * (self.beta_1 - self.beta_0)
        return -0.5 * beta_t.view([-1] + [1] * (x.dim() - 1)) * x, torch.sqrt(beta_t)

    def marGINal_prob(self, x, t):
        log_mean_coeff = -0.25 * t ** 2 * (self.beta_1 - self.beta_0) - 0.5 * t * self.beta_0
        mean = torch.exp(log_mean_coeff.view([-1] + [1] * (x.dim() - 1))) * x
        return mean, torch.sqrt(1.0 - torch.exp(2.0 * log_mean_coeff))

    def discretize(self, x, t):
        timestep = (t * (self.N - 1) / self.T).long()
        beta = self.discrete_betas.to(x.device)[timestep]
        alpha = self.alphas.to(x.device)[timestep]
        f = torch.sqrt(alpha).view([-1] + [1] * (x.dim() - 1)) * x - x
        return f, torch.sqrt(beta)

    def reverse_discretize(self, score_model, x, representation, data, t, probability_flow=False):
        """RSDE.discretize (SDE_sparse.py:94-100) on the DDPM discretisation above."""
        f, G = self.discretize(x, t)
        score = score_model.get_score(representation, data, x, None, t)
        rev_f = f - G[:, None] ** 2 * score * (0.5 if probability_flow else 1.0)
        return rev_f, (torch.zeros_like(G) if probability_flow else G)

    def corrector_alpha(self, t):
        """alpha of the Langevin corrector for VP SDEs (pretrain_MoleculeSDE_inference_2D_to_3D_VE_VP.py:198-200)."""
        timestep = (t * (self.N - 1) / self.T).long()
        return self.alphas.to(t.device)[timestep]
