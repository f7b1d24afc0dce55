"""Diffusers-style DDPM / PNDM schedulers whose ``step`` is one fused HIP kernel.

Call surface mirrors what BrepGen uses from ``diffusers==0.27`` (constructed at sample.py:101-117 and
trainer.py:285-292; ``set_timesteps`` / ``timesteps`` / ``step(...).prev_sample`` at sample.py:128-153;
``add_noise`` at trainer.py:348): same constructor keywords, same attribute names, same stateful PNDM behaviour.

Host side: the per-step scalar coefficients are computed in float32 numpy in upstream's operation order.
Device side: ``bg_cfg_ddpm_step`` / ``bg_pndm_step`` read eps (optionally the two halves of a classifier-free
guidance batch, combined in-kernel), the sample, the noise / PLMS history, and write prev_sample -- one pass.

Extensions over upstream (all keyword-only, default = upstream behaviour):
  noise=...      inject the ancestral noise (upstream draws it from the global device RNG, sample.py:153)
  guidance=(w)   with model_output of batch 2B: eps = eps[:B]*(1+w) - eps[B:]*w fused into the step
"""
from types import SimpleNamespace

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream
from .utils import randn_tensor

f32 = np.float32


class SchedulerOutput:
    def __init__(self, prev_sample):
        self.prev_sample = prev_sample

    def __iter__(self):             # return_dict=False convention: (prev_sample,)
        yield self.prev_sample


def _alphas_cumprod(num_train_timesteps, beta_start, beta_end, beta_schedule):
    if beta_schedule != "linear":
        raise NotImplementedError("BrepGen only uses beta_schedule='linear'")
    betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
    return betas, torch.cumprod(1.0 - betas, dim=0)


def _split_guidance(model_output, sample, guidance):
    """-> (eps_c ptr-tensor, eps_u ptr-tensor or None, w)."""
    if guidance is None:
        if model_output.shape != sample.shape:
            raise ValueError("model_output and sample shapes differ (pass guidance=w for a 2B CFG batch)")
        return model_output, None, 0.0
    B = sample.shape[0]
    if model_output.shape[0] != 2 * B:
        raise ValueError("guidance needs model_output of batch 2B (cond ; uncond)")
    return model_output[:B], model_output[B:], float(guidance)


def _prep(t):
    if not t.is_cuda:
        raise _lib.BrepgenHipError(f"scheduler.step runs on the MI355X only (tensor on {t.device}); no CPU fallback")
    return t.detach().to(torch.float32).contiguous()


class DDPMScheduler:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 prediction_type="epsilon", clip_sample=True, clip_sample_range=1.0, variance_type="fixed_small",
                 timestep_spacing="leading", steps_offset=0):
        if prediction_type != "epsilon" or variance_type != "fixed_small" or timestep_spacing != "leading":
            raise NotImplementedError("only the configuration BrepGen uses is implemented")
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start,
                                      beta_end=beta_end, beta_schedule=beta_schedule,
                                      prediction_type=prediction_type, clip_sample=clip_sample,
                                      clip_sample_range=clip_sample_range, variance_type=variance_type,
                                      timestep_spacing=timestep_spacing, steps_offset=steps_offset)
        self.betas, self.alphas_cumprod = _alphas_cumprod(num_train_timesteps, beta_start, beta_end, beta_schedule)
        self._acp = self.alphas_cumprod.numpy().astype(np.float32)
        self.num_inference_steps = None
        self._set_schedule(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64), None)

    def _set_schedule(self, ts, device):
        self._host_ts, self._cursor = [int(v) for v in ts], 0
        self.timesteps = torch.from_numpy(ts).to(device)

    def _timestep(self, timestep):
        """The step's timestep as a host integer WITHOUT a device synchronisation.  A caller that follows the reference
        literally iterates `scheduler.timesteps.cuda()` and passes 0-d device tensors: `int()` of one would stall the host on
        everything enqueued so far, at every step.  The schedule is host state (set_timesteps built it), so a device timestep is
        taken from it by position -- the steps of a schedule are consumed in order, as sample.py:126-202 does, and the cursor
        wraps for the next sampling run; a host timestep (int / CPU tensor) is used as given and re-seats the cursor."""
        if torch.is_tensor(timestep) and timestep.device.type != "cpu":
            t = self._host_ts[self._cursor % len(self._host_ts)]
            self._cursor += 1
            return t
        t = int(timestep)
        n = len(self._host_ts)
        c = self._cursor % n
        if self._host_ts[c] == t:
            self._cursor = c + 1
        elif t in self._host_ts:
            self._cursor = self._host_ts.index(t) + 1
        return t

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps, device=None):
        T = self.config.num_train_timesteps
        if num_inference_steps > T:
            raise ValueError("num_inference_steps cannot exceed num_train_timesteps")
        self.num_inference_steps = num_inference_steps
        ratio = T // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        ts += self.config.steps_offset
        self._set_schedule(ts, device)

    def _coefficients(self, t):
        T = self.config.num_train_timesteps
        n = self.num_inference_steps if self.num_inference_steps else T
        prev_t = t - T // n
        a_t = f32(self._acp[t])
        a_p = f32(self._acp[prev_t]) if prev_t >= 0 else f32(1.0)
        b_t, b_p = f32(1.0) - a_t, f32(1.0) - a_p
        cur_a = f32(a_t / a_p)
        cur_b = f32(1.0) - cur_a
        sigma = f32(0.0)
        if t > 0:
            var = max(f32(b_p / b_t * cur_b), f32(1e-20))
            sigma = f32(np.sqrt(var))
        return dict(sqrt_alpha_prod=f32(np.sqrt(a_t)), sqrt_beta_prod=f32(np.sqrt(b_t)),
                    x0_coeff=f32(f32(np.sqrt(a_p)) * cur_b / b_t), xt_coeff=f32(f32(np.sqrt(cur_a)) * b_p / b_t),
                    sigma=sigma)

    def step(self, model_output, timestep, sample, generator=None, return_dict=True, *, noise=None, guidance=None):
        t = self._timestep(timestep)
        x = _prep(sample)
        eps_c, eps_u, w = _split_guidance(_prep(model_output), x, guidance)
        c = self._coefficients(t)
        if t > 0 and noise is None:
            # upstream: randn_tensor(model_output.shape, generator=generator, device=..., dtype=...)
            noise = randn_tensor(tuple(x.shape), generator=generator, device=x.device, dtype=torch.float32)
        z = _prep(noise) if (t > 0 and noise is not None) else None
        out = torch.empty_like(x)
        clip = float(self.config.clip_sample_range) if self.config.clip_sample else 0.0
        check(_lib.load().bg_cfg_ddpm_step(ptr(eps_c), ptr(eps_u), w, ptr(x), ptr(z), ptr(out), x.numel(),
                                           float(c["sqrt_alpha_prod"]), float(c["sqrt_beta_prod"]),
                                           float(c["x0_coeff"]), float(c["xt_coeff"]), float(c["sigma"]), clip,
                                           stream()), "bg_cfg_ddpm_step")
        return SchedulerOutput(out) if return_dict else (out,)

    def add_noise(self, original_samples, noise, timesteps):
        x0, z = _prep(original_samples), _prep(noise)
        # the table follows the samples onto the device once; the per-sample gather stays there (no host round trip
        # when the trainer draws its timesteps on the device, trainer.py:346)
        if getattr(self, "_acp_dev", None) is None or self._acp_dev.device != x0.device:
            self._acp_dev = self.alphas_cumprod.to(x0.device)
        acp = self._acp_dev.index_select(0, timesteps.reshape(-1).to(device=x0.device, dtype=torch.long))
        sa = acp.sqrt().contiguous()                                 # correctly rounded, as the host pow(., 0.5) is
        sb = (1 - acp).sqrt().contiguous()
        B = x0.shape[0]
        if sa.numel() != B:
            raise ValueError("add_noise needs one timestep per sample")
        out = torch.empty_like(x0)
        check(_lib.load().bg_add_noise(ptr(x0), ptr(z), ptr(sa), ptr(sb), ptr(out), B, x0.numel() // B, stream()),
              "bg_add_noise")
        return out

    def __len__(self):
        return self.config.num_train_timesteps


class PNDMScheduler:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 skip_prk_steps=False, set_alpha_to_one=False, prediction_type="epsilon",
                 timestep_spacing="leading", steps_offset=0):
        if skip_prk_steps or prediction_type != "epsilon" or timestep_spacing != "leading":
            raise NotImplementedError("only the configuration BrepGen uses is implemented")
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start,
                                      beta_end=beta_end, beta_schedule=beta_schedule, skip_prk_steps=skip_prk_steps,
                                      set_alpha_to_one=set_alpha_to_one, prediction_type=prediction_type,
                                      timestep_spacing=timestep_spacing, steps_offset=steps_offset)
        self.betas, self.alphas_cumprod = _alphas_cumprod(num_train_timesteps, beta_start, beta_end, beta_schedule)
        self._acp = self.alphas_cumprod.numpy().astype(np.float32)
        self.final_alpha_cumprod = f32(1.0) if set_alpha_to_one else f32(self._acp[0])
        self.pndm_order = 4
        self.num_inference_steps = None
        self._reset()
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))
        self._host_ts = [int(v) for v in self.timesteps]
        self.prk_timesteps = np.array([], dtype=np.int64)

    def _reset(self):
        self.counter = 0
        self.ets = []                 # device tensors: the eps history of PLMS
        self.cur_model_output = None  # PRK running combination (device tensor or None == 0)
        self.cur_sample = None

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps, device=None):
        T = self.config.num_train_timesteps
        self.num_inference_steps = num_inference_steps
        ratio = T // num_inference_steps
        _t = (np.arange(0, num_inference_steps) * ratio).round() + self.config.steps_offset
        prk = np.array(_t[-self.pndm_order:]).repeat(2) + np.tile(
            np.array([0, T // num_inference_steps // 2]), self.pndm_order)
        self.prk_timesteps = (prk[:-1].repeat(2)[1:-1])[::-1].copy()
        self.plms_timesteps = _t[:-3][::-1].copy()
        ts = np.concatenate([self.prk_timesteps, self.plms_timesteps]).astype(np.int64)
        self.timesteps = torch.from_numpy(ts).to(device)
        self._host_ts = [int(v) for v in ts]
        self._reset()

    def _prev_coeffs(self, t, prev_t):
        a_t = f32(self._acp[t])
        a_p = f32(self._acp[prev_t]) if prev_t >= 0 else self.final_alpha_cumprod
        b_t, b_p = f32(1.0) - a_t, f32(1.0) - a_p
        sample_coeff = f32(np.sqrt(f32(a_p / a_t)))
        denom = f32(a_t * f32(np.sqrt(b_p))) + f32(np.sqrt(f32(f32(a_t * b_t) * a_p)))
        return float(sample_coeff), float(f32(f32(a_p - a_t) / denom))

    def step(self, model_output, timestep, sample, return_dict=True, *, guidance=None):
        if self.num_inference_steps is None:
            raise ValueError("call set_timesteps first")
        if torch.is_tensor(timestep) and timestep.device.type != "cpu":
            # no device synchronisation (see DDPMScheduler._timestep): PNDM is stateful anyway -- evaluation number `counter` of the
            # schedule IS timesteps[counter], as upstream's own step_prk / step_plms assume
            t = self._host_ts[self.counter % len(self._host_ts)]
        else:
            t = int(timestep)
        x = _prep(sample)
        eps_c, eps_u, w = _split_guidance(_prep(model_output), x, guidance)
        ratio = self.config.num_train_timesteps // self.num_inference_steps
        lib = _lib.load()
        out = torch.empty_like(x)
        n = x.numel()
        if self.counter < len(self.prk_timesteps):
            # ---- Runge-Kutta warm-up (4 evaluations per step) ----
            r = self.counter % 4
            prev_t = t - (0 if self.counter % 2 else ratio // 2)
            t_eff = int(self.prk_timesteps[self.counter // 4 * 4])
            sc, ec = self._prev_coeffs(t_eff, prev_t)
            if r == 0:
                self.cur_sample = x
            acc = self.cur_model_output
            e_store = acc_out = None
            a_old = a_e = 0.0
            if r == 0:          # cur_out += e/6 ; ets.append(e) ; prev from e
                e_store = torch.empty_like(x)
                acc_out = torch.empty_like(x)
                a_old, a_e, c_e, c_acc = (1.0 if acc is not None else 0.0), 1.0 / 6.0, 1.0, 0.0
            elif r in (1, 2):   # cur_out += e/3 ; prev from e
                acc_out = acc    # in place
                a_old, a_e, c_e, c_acc = 1.0, 1.0 / 3.0, 1.0, 0.0
            else:               # e' = cur_out + e/6 ; cur_out = 0 ; prev from e'
                c_e, c_acc = 1.0 / 6.0, 1.0
            check(lib.bg_pndm_step(ptr(eps_c), ptr(eps_u), w, ptr(self.cur_sample), ptr(e_store), ptr(acc),
                                   ptr(acc_out), a_old, a_e, c_e, c_acc, None, None, None, 0.0, 0.0, 0.0,
                                   sc, ec, ptr(out), n, stream()), "bg_pndm_step[prk]")
            if r == 0:
                self.ets.append(e_store)
                self.cur_model_output = acc_out
            elif r == 3:
                self.cur_model_output = None
        else:
            # ---- linear multistep ----
            prev_t = t - ratio
            hist = self.ets[-3:]
            e_store = torch.empty_like(x)
            k = len(hist)
            if k == 0:
                c_e, ch = 1.0, []
            elif k == 1:
                c_e, ch = 3.0 / 2.0, [-1.0 / 2.0]
            elif k == 2:
                c_e, ch = 23.0 / 12.0, [-16.0 / 12.0, 5.0 / 12.0]
            else:
                c_e, ch = 55.0 / 24.0, [-59.0 / 24.0, 37.0 / 24.0, -9.0 / 24.0]
            hs = list(reversed(hist)) + [None] * (3 - k)          # newest first: ets[-2], ets[-3], ets[-4]
            ch = ch + [0.0] * (3 - k)
            sc, ec = self._prev_coeffs(t, prev_t)
            check(lib.bg_pndm_step(ptr(eps_c), ptr(eps_u), w, ptr(x), ptr(e_store), None, None, 0.0, 0.0,
                                   c_e, 0.0, ptr(hs[0]), ptr(hs[1]), ptr(hs[2]), ch[0], ch[1], ch[2],
                                   sc, ec, ptr(out), n, stream()), "bg_pndm_step[plms]")
            self.ets = hist + [e_store]
        self.counter += 1
        return SchedulerOutput(out) if return_dict else (out,)

    def __len__(self):
        return self.config.num_train_timesteps
