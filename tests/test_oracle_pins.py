"""Independent pins of the restated third-party pieces to their PUBLISHED math (diffusers itself is not installable
offline, so `oracle/schedulers.py` cannot be diffed against it here -- `tests/golden/pin_diffusers.py` does that the
moment diffusers==0.27 is importable):

  DDPM ancestral step   Ho, Jain, Abbeel 2020 (arXiv:2006.11239): x0 from eq. 15, posterior mean / variance eq. 6-7
  PNDM transfer         Liu et al. 2022 (arXiv:2202.09778) eq. 11 (the "transfer part" phi), pseudo RK eq. 13, PLMS eq. 12
everything in float64 from the beta schedule alone, against the float32 restatement over the FULL schedules the cascade
runs (1000 / 50-step DDPM; 209-evaluation PNDM at 200 inference steps).  fp32 scalar arithmetic in upstream's operation
order has cancellation in (1 - alpha_bar) at the smallest timesteps, hence the two tolerance bands.
"""
import numpy as np
import torch

from oracle.schedulers import OracleDDPM, OraclePNDM

BETAS = np.linspace(1e-4, 0.02, 1000, dtype=np.float64)
ACP = np.cumprod(1.0 - BETAS)


def _acp(t):
    return 1.0 if t < 0 else ACP[t]


def _ddpm_fp64(x, eps, z, t, ratio, clip):
    a_t, a_p = _acp(t), _acp(t - ratio)
    alpha_t = a_t / a_p                                  # the step's alpha (== 1 - beta_t when ratio == 1)
    beta_t = 1.0 - alpha_t
    x0 = (x - np.sqrt(1.0 - a_t) * eps) / np.sqrt(a_t)   # eq. 15 solved for x0
    if clip:
        x0 = np.clip(x0, -clip, clip)
    mean = np.sqrt(a_p) * beta_t / (1.0 - a_t) * x0 + np.sqrt(alpha_t) * (1.0 - a_p) / (1.0 - a_t) * x     # eq. 7
    var = (1.0 - a_p) / (1.0 - a_t) * beta_t                                                              # eq. 7
    return mean + (np.sqrt(max(var, 1e-20)) * z if t > 0 else 0.0)


def test_ddpm_step_equals_published_posterior_over_full_schedules():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 11, generator=g) * 1.5
    eps = torch.randn(3, 11, generator=g)
    z = torch.randn(3, 11, generator=g)
    for n in (1000, 50):
        for clip in (0.0, 3.0):
            s = OracleDDPM(clip_sample=clip > 0, clip_sample_range=clip if clip else 1.0)
            s.set_timesteps(n)
            ratio = 1000 // n
            assert s.timesteps.tolist() == list(range(1000 - ratio, -1, -ratio))      # "leading" spacing
            for t in s.timesteps.tolist():
                got = s.step(eps, t, x, noise=z).double().numpy()
                want = _ddpm_fp64(x.double().numpy(), eps.double().numpy(), z.double().numpy(), t, ratio, clip)
                tol = 2e-3 if t < 20 else 2e-4           # fp32 cancellation in 1 - alpha_bar near t = 0
                assert np.abs(got - want).max() <= tol * max(1.0, np.abs(want).max()), (n, clip, t)
                if t == 1000 - ratio:                    # x_T is kept: alpha_bar_T ~ 4e-5 would blow x0 up
                    continue


def test_ddpm_with_true_eps_recovers_x0_at_every_step():
    g = torch.Generator().manual_seed(1)
    x0 = torch.randn(2, 9, generator=g)
    e = torch.randn(2, 9, generator=g)
    s = OracleDDPM(clip_sample=False)
    s.set_timesteps(1000)
    for t in (999, 700, 249, 50, 5, 1):
        a = ACP[t]
        xt = (np.sqrt(a) * x0.double() + np.sqrt(1 - a) * e.double()).float()
        prev = s.step(e, t, xt, noise=torch.zeros_like(xt)).double()
        # posterior mean with the true x0 == mean of q(x_{t-1} | x_t, x0): closed form (eq. 6-7)
        a_p = _acp(t - 1)
        want = np.sqrt(a_p) * BETAS[t] / (1 - a) * x0.double() + np.sqrt(1 - BETAS[t]) * (1 - a_p) / (1 - a) * xt.double()
        assert float((prev - want).abs().max()) < 2e-3 * max(1.0, float(want.abs().max()))


def _transfer_fp64(x, eps, t, prev_t, final_acp):
    a_t = ACP[t]
    a_p = ACP[prev_t] if prev_t >= 0 else final_acp
    # Liu et al. eq. 11
    return (np.sqrt(a_p / a_t) * x
            - (a_p - a_t) / (np.sqrt(a_t) * (np.sqrt((1 - a_p) * a_t) + np.sqrt((1 - a_t) * a_p))) * eps)


def test_pndm_transfer_coefficients_equal_published_formula():
    s = OraclePNDM()
    s.set_timesteps(200)
    for t, p in [(995, 992), (992, 990), (980, 975), (500, 495), (255, 250), (5, 0), (0, -5)]:
        sc, ec = s.prev_sample_coeffs(t, p)
        got = float(sc) * 1.25 - float(ec) * 0.75
        want = _transfer_fp64(1.25, 0.75, t, p, ACP[0])
        assert abs(got - want) < 3e-5 * max(1.0, abs(want)), (t, p)


def test_pndm_constant_eps_run_telescopes_to_the_ddim_endpoint():
    """With a constant eps every PRK / PLMS combination equals eps (weights 1/6+1/3+1/3+1/6 and (55-59+37-9)/24 sum
    to 1) and the transfer formula is the deterministic DDIM map, so the 209-evaluation run must land exactly on
    sqrt(abar_end/abar_start)-scaled closed form -- which it only does if the time-step sequence chains
    (prev_t of one step == t of the next), the PRK bookkeeping re-uses cur_sample, and final_alpha_cumprod = abar_0."""
    s = OraclePNDM()
    s.set_timesteps(200)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(4, 6, generator=g)
    e = torch.randn(4, 6, generator=g) * 0.7
    xs = x.clone()
    for t in s.timesteps:
        xs = s.step(e, t, xs)
    a_s, a_e = ACP[995], ACP[0]                          # first PRK timestep 995; last prev_t = -5 -> abar_0
    x0 = (x.double() - np.sqrt(1 - a_s) * e.double()) / np.sqrt(a_s)
    want = np.sqrt(a_e) * x0 + np.sqrt(1 - a_e) * e.double()
    # x0 is amplified by 1/sqrt(abar_995) ~ 150: compare relative to that scale (fp32 chain of 209 affine maps)
    assert float((xs.double() - want).abs().max()) < 2e-3 * float(want.abs().max())


def test_pndm_prk_is_rk4_and_plms_is_adams_bashforth():
    """Liu et al. eq. 13 (pseudo Runge-Kutta, weights 1/6 1/3 1/3 1/6) and eq. 12 (4th-order pseudo linear multistep,
    55/-59/37/-9 over 24) on a strongly time-dependent eps field, against an fp64 implementation written from the
    paper.  The only inputs taken from the schedule rather than the paper are the INTEGER evaluation times of the
    half steps -- the reference iterates `timesteps` = [995, 992, 992, 990, ...] (sample.py:129) and
    upstream transfers to t - 2 on even evaluations: 995 -> 993, then to 992, 990, 990."""
    field = lambda xx, t: np.sin(0.7 * t) * 0.3 * xx + 0.1 * np.cos(1.0 * t)
    s = OraclePNDM()
    s.set_timesteps(200)
    x = torch.tensor([1.0, -2.0, 0.5])
    xs = x.clone()
    for t in s.timesteps[:15]:                           # 12 PRK evaluations (3 RK steps) + 3 PLMS steps
        xs = s.step(torch.from_numpy(field(xs.double().numpy(), int(t))).float(), t, xs)
    phi = lambda xx, ee, t, p: _transfer_fp64(xx, ee, t, p, ACP[0])
    xx = x.double().numpy()
    ets = []
    for t in (995, 990, 985):                            # pseudo Runge-Kutta steps of size delta = 5
        e1 = field(xx, t)
        x1 = phi(xx, e1, t, t - 2)
        e2 = field(x1, t - 3)
        x2 = phi(xx, e2, t, t - 3)
        e3 = field(x2, t - 3)
        x3 = phi(xx, e3, t, t - 5)
        e4 = field(x3, t - 5)
        xx = phi(xx, (e1 + 2 * e2 + 2 * e3 + e4) / 6, t, t - 5)
        ets.append(e1)
    for t in (980, 975, 970):
        ets.append(field(xx, t))
        h = ets[-4:]
        comb = (55 * h[-1] - 59 * h[-2] + 37 * h[-3] - 9 * h[-4]) / 24
        xx = phi(xx, comb, t, t - 5)
    assert np.abs(xs.double().numpy() - xx).max() < 1e-4 * max(1.0, np.abs(xx).max())
