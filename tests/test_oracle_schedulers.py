"""Self-consistency + known-answer tests of the restated diffusers==0.27 schedulers
(parity unpinned: diffusers is not vendored / installable -- see oracle/schedulers.py)."""
import numpy as np
import torch

from oracle.schedulers import OracleDDPM, OraclePNDM, make_alphas_cumprod


def test_alphas_cumprod_known_answers():
    a = make_alphas_cumprod()
    # SURVEY.md App. B.4
    for t, v in [(0, 0.99989998), (249, 0.52408534), (255, 0.50816011), (500, 0.07779665),
                 (980, 5.9037520e-05), (999, 4.0358304e-05)]:
        assert abs(a[t] - v) <= 2e-7 * max(1.0, v / 1e-4) or abs(a[t] / v - 1) < 1e-6


def test_ddpm_coefficients_t249():
    s = OracleDDPM(clip_sample=True, clip_sample_range=3)
    s.set_timesteps(1000)
    c = s.coefficients(249)
    assert abs(c["x0_coeff"] - 0.0077166818) < 1e-8
    assert abs(c["xt_coeff"] - 0.99188036) < 1e-6
    assert abs(c["variance"] - 0.0050317375) < 1e-8


def test_ddpm_timesteps():
    s = OracleDDPM()
    s.set_timesteps(1000)
    assert s.timesteps[0] == 999 and s.timesteps[-1] == 0 and len(s.timesteps) == 1000
    assert list(s.timesteps[-250:][:2]) == [249, 248]
    s.set_timesteps(50)
    assert list(s.timesteps[:3]) == [980, 960, 940] and s.timesteps[-1] == 0


def test_pndm_timesteps_200():
    s = OraclePNDM()
    s.set_timesteps(200)
    ts = s.timesteps.tolist()
    assert len(ts) == 209
    assert ts[:12] == [995, 992, 992, 990, 990, 987, 987, 985, 985, 982, 982, 980]
    assert ts[12:15] == [980, 975, 970] and ts[-1] == 0
    assert ts[157] == 255          # sample.py:129 runs timesteps[:158]


def test_ddpm_exact_eps_roundtrip():
    """With the true eps and no clipping, the x0 estimate is exact and the last step returns x0."""
    g = torch.Generator().manual_seed(0)
    s = OracleDDPM(clip_sample=False)
    s.set_timesteps(1000)
    x0 = torch.randn(4, 7, 6, generator=g)
    eps = torch.randn(4, 7, 6, generator=g)
    xt = s.add_noise(x0, eps, torch.tensor([0, 0, 0, 0]))
    out = s.step(eps, 0, xt)            # t=0: no noise, prev alpha = 1 -> returns x0_hat
    assert float((out - x0).abs().max()) < 1e-5


def test_ddpm_clip_range():
    s = OracleDDPM(clip_sample=True, clip_sample_range=3)
    s.set_timesteps(1000)
    x = torch.full((2, 3), 50.0)
    out = s.step(torch.zeros(2, 3), 0, x)
    assert float(out.max()) <= 3.0 + 1e-6


def test_pndm_full_run_is_finite_and_reaches_x0_for_consistent_eps():
    """Drive PNDM with the analytically consistent eps of a point mass at x0: must converge to x0."""
    s = OraclePNDM()
    s.set_timesteps(200)
    g = torch.Generator().manual_seed(1)
    x0 = torch.randn(3, 5, generator=g)
    x = torch.randn(3, 5, generator=g)
    acp = torch.from_numpy(make_alphas_cumprod())
    for t in s.timesteps:
        a = acp[int(t)]
        eps = (x - a.sqrt() * x0) / (1 - a).sqrt()
        x = s.step(eps, t, x)
        assert torch.isfinite(x).all()
    # the last step (t=0 -> prev=-5 -> final_alpha_cumprod=acp[0]) is the identity, so the residual is the
    # t=0 noise level sqrt(1-acp[0])=0.01 times |eps|~3.6
    assert float((x - x0).abs().max()) < 5e-2
    assert s.counter == 209


def test_pndm_plms_uses_four_term_formula_after_prk():
    s = OraclePNDM()
    s.set_timesteps(200)
    x = torch.zeros(2)
    hist = []
    for i, t in enumerate(s.timesteps[:13]):
        e = torch.full((2,), float(i + 1))
        hist.append(e)
        out = s.step(e, t, x)
    # 13th call (index 12) is the first PLMS step: ets = [e0, e4, e8, e12]
    comb = (55 * hist[12] - 59 * hist[8] + 37 * hist[4] - 9 * hist[0]) / 24
    sc, ec = s.prev_sample_coeffs(980, 975)
    assert torch.allclose(out, float(sc) * x - float(ec) * comb, atol=1e-6)
