"""CPU restatement of the two diffusers==0.27 schedulers BrepGen samples with.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  *** PARITY UNPINNED ***:
``diffusers`` is a third-party dependency pinned at ==0.27 in the reference's
``requirements.txt:5``; it is not vendored under /root/reference and cannot be
installed offline, so this file restates the published 0.27 algorithm of
``DDPMScheduler`` / ``PNDMScheduler`` and is anchored on the reference's call
sites only:

  constructors   sample.py:101-117 (PNDM; DDPM with clip_sample=True, range 3)
                 trainer.py:285-292 (DDPM, no clipping, training add_noise)
  set_timesteps  sample.py:128,144,191,210,224,269
  step           sample.py:137,153,202,222,236,282
  add_noise      trainer.py:348,399,...

Defaults that matter (0.27): beta_schedule="linear", timestep_spacing="leading",
steps_offset=0, variance_type="fixed_small", thresholding=False,
PNDM skip_prk_steps=False, set_alpha_to_one=False.

All scalar coefficient math is done in float32, in the same operation order as
upstream (0-d float32 tensors there), the tensor math in float32.
"""
import numpy as np
import torch

f32 = np.float32


def make_alphas_cumprod(num_train_timesteps=1000, beta_start=1e-4, beta_end=0.02):
    betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
    return torch.cumprod(1.0 - betas, dim=0).numpy().astype(np.float32)


class OracleDDPM:
    def __init__(self, num_train_timesteps=1000, beta_start=1e-4, beta_end=0.02,
                 clip_sample=True, clip_sample_range=1.0):
        self.T = num_train_timesteps
        self.acp = make_alphas_cumprod(num_train_timesteps, beta_start, beta_end)
        self.clip_sample = clip_sample
        self.clip_sample_range = float(clip_sample_range)
        self.num_inference_steps = None
        self.timesteps = torch.arange(num_train_timesteps - 1, -1, -1, dtype=torch.int64)

    def set_timesteps(self, n):
        self.num_inference_steps = n
        ratio = self.T // n
        ts = (np.arange(0, n) * ratio).round()[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts)

    def coefficients(self, t):
        """float32 scalars of one ancestral step (upstream op order)."""
        t = int(t)
        n = self.num_inference_steps if self.num_inference_steps else self.T
        prev_t = t - self.T // n
        a_t = f32(self.acp[t])
        a_p = f32(self.acp[prev_t]) if prev_t >= 0 else f32(1.0)
        b_t = f32(1.0) - a_t
        b_p = f32(1.0) - a_p
        cur_a = f32(a_t / a_p)
        cur_b = f32(1.0) - cur_a
        c = {
            "sqrt_beta_prod_t": f32(np.sqrt(b_t)),
            "sqrt_alpha_prod_t": f32(np.sqrt(a_t)),
            "x0_coeff": f32(f32(np.sqrt(a_p)) * cur_b / b_t),
            "xt_coeff": f32(f32(np.sqrt(cur_a)) * b_p / b_t),
            "sigma": f32(0.0),
        }
        if t > 0:
            var = f32(b_p / b_t * cur_b)
            var = max(var, f32(1e-20))
            c["sigma"] = f32(np.sqrt(var))
            c["variance"] = var
        return c

    def step(self, eps, t, x, noise=None):
        """eps, x float32 tensors; noise: N(0,1) tensor (required for t>0)."""
        c = self.coefficients(t)
        eps = eps.to(torch.float32)
        x0 = (x - float(c["sqrt_beta_prod_t"]) * eps) / float(c["sqrt_alpha_prod_t"])
        if self.clip_sample:
            x0 = x0.clamp(-self.clip_sample_range, self.clip_sample_range)
        prev = float(c["x0_coeff"]) * x0 + float(c["xt_coeff"]) * x
        if int(t) > 0:
            assert noise is not None, "inject the noise for t>0 (global-RNG draw upstream)"
            prev = prev + float(c["sigma"]) * noise
        return prev

    def add_noise(self, x0, noise, timesteps):
        a = torch.from_numpy(self.acp)[timesteps.long()]
        sa = (a ** 0.5).reshape(-1, *([1] * (x0.dim() - 1)))
        sb = ((1 - a) ** 0.5).reshape(-1, *([1] * (x0.dim() - 1)))
        return sa * x0 + sb * noise


class OraclePNDM:
    def __init__(self, num_train_timesteps=1000, beta_start=1e-4, beta_end=0.02):
        self.T = num_train_timesteps
        self.acp = make_alphas_cumprod(num_train_timesteps, beta_start, beta_end)
        self.final_alpha_cumprod = f32(self.acp[0])       # set_alpha_to_one=False
        self.pndm_order = 4
        self.set_timesteps(num_train_timesteps)

    def set_timesteps(self, n):
        self.num_inference_steps = n
        ratio = self.T // n
        _t = (np.arange(0, n) * ratio).round()
        prk = np.array(_t[-self.pndm_order:]).repeat(2) + np.tile(
            np.array([0, self.T // n // 2]), self.pndm_order)
        self.prk_timesteps = (prk[:-1].repeat(2)[1:-1])[::-1].copy()
        self.plms_timesteps = _t[:-3][::-1].copy()
        self.timesteps = torch.from_numpy(
            np.concatenate([self.prk_timesteps, self.plms_timesteps]).astype(np.int64))
        self.ets = []
        self.counter = 0
        self.cur_model_output = 0
        self.cur_sample = None

    def prev_sample_coeffs(self, t, prev_t):
        a_t = f32(self.acp[int(t)])
        a_p = f32(self.acp[int(prev_t)]) if prev_t >= 0 else self.final_alpha_cumprod
        b_t = f32(1.0) - a_t
        b_p = f32(1.0) - a_p
        sample_coeff = f32(np.sqrt(f32(a_p / a_t)))
        denom = f32(a_t * f32(np.sqrt(b_p))) + f32(np.sqrt(f32(f32(a_t * b_t) * a_p)))
        eps_coeff = f32(f32(a_p - a_t) / denom)
        return sample_coeff, eps_coeff

    def _prev(self, x, t, prev_t, eps):
        sc, ec = self.prev_sample_coeffs(t, prev_t)
        return float(sc) * x - float(ec) * eps

    def step(self, eps, t, x):
        eps = eps.to(torch.float32)
        t = int(t)
        ratio = self.T // self.num_inference_steps
        if self.counter < len(self.prk_timesteps):
            diff = 0 if self.counter % 2 else ratio // 2
            prev_t = t - diff
            t_eff = int(self.prk_timesteps[self.counter // 4 * 4])
            r = self.counter % 4
            if r == 0:
                self.cur_model_output = self.cur_model_output + (1 / 6) * eps
                self.ets.append(eps)
                self.cur_sample = x
            elif r in (1, 2):
                self.cur_model_output = self.cur_model_output + (1 / 3) * eps
            else:
                eps = self.cur_model_output + (1 / 6) * eps
                self.cur_model_output = 0
            out = self._prev(self.cur_sample, t_eff, prev_t, eps)
            self.counter += 1
            return out
        # PLMS
        prev_t = t - ratio
        self.ets = self.ets[-3:]
        self.ets.append(eps)
        e = self.ets
        if len(e) == 1:      # unreachable after PRK; kept for completeness
            comb = e[-1]
        elif len(e) == 2:
            comb = (3 * e[-1] - e[-2]) / 2
        elif len(e) == 3:
            comb = (23 * e[-1] - 16 * e[-2] + 5 * e[-3]) / 12
        else:
            comb = (1 / 24) * (55 * e[-1] - 59 * e[-2] + 37 * e[-3] - 9 * e[-4])
        out = self._prev(x, t, prev_t, comb)
        self.counter += 1
        return out
