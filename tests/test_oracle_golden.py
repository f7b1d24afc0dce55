"""The oracle restatement vs the golden vectors written by the reference's own classes
(tests/golden/gen_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import denoisers as orc

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
MANIFEST = json.load(open(os.path.join(GOLDEN, "MANIFEST.json")))


def load_case(name):
    meta = MANIFEST["cases"][name]
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    args = [torch.from_numpy(z[k]) if k in z.files else None for k in meta["args"]]
    return meta, args, torch.from_numpy(z["out"])


@pytest.mark.parametrize("name", sorted(MANIFEST["cases"]))
def test_oracle_matches_reference_golden(name):
    meta, args, want = load_case(name)
    sd = orc.make_state_dict(meta.get("weights", "seeded"), meta["net"], meta["weight_seed"], meta["use_cf"])
    with torch.no_grad():
        got = orc.FORWARD[meta["net"]](sd, *args)
    assert got.shape == want.shape
    # fp32 summation-order noise only (recorded 1.3e-6 .. 2.2e-6 at generation time; the hostile-weight cases, whose residual
    # stream reaches |x| ~ 900, 2.6e-6 .. 3.5e-5: MANIFEST.json) -- relative to the output's magnitude
    assert float((got - want).abs().max()) < 1e-5 * max(1.0, meta["out_absmax"])


def test_state_dict_spec_counts():
    # 164 tensors for SurfPosNet without CFG
    assert len(orc.state_dict_spec("SurfPosNet")) == 164
    n = sum(int(np.prod(s)) for s in orc.state_dict_spec("SurfPosNet").values())
    assert abs(n - 49.66e6) < 0.02e6
    n = sum(int(np.prod(s)) for s in orc.state_dict_spec("EdgeZNet").values())
    assert abs(n - 52.10e6) < 0.02e6


def test_masked_keys_do_not_influence_valid_tokens():
    """Padded tokens are excluded as keys (network.py:1196): changing them must not move valid outputs."""
    meta, args, want = load_case("surfz_b3_n60")
    sd = orc.make_state_dict(meta.get("weights", "seeded"), meta["net"], meta["weight_seed"], meta["use_cf"])
    surfZ, t, surfPos, mask, cl = args
    z2 = surfZ.clone()
    z2[mask] = 123.0
    with torch.no_grad():
        got = orc.surfz_forward(sd, z2, t, surfPos, mask, cl)
    assert float((got - want)[~mask].abs().max()) < 1e-5


def test_sincos_cos_first():
    e = orc.sincos_embedding(torch.tensor([0]))
    assert torch.all(e[0, :384] == 1) and torch.all(e[0, 384:] == 0)


@pytest.mark.parametrize("name", sorted(MANIFEST["cases"]))
def test_reference_formulation_matches_reference_golden(name):
    """oracle/ref_formulation.py (stock nn.TransformerEncoder, seq-first -- the like-for-like autocast target and the
    CPU-baseline formulation) loads the reference-keyed weights strictly and reproduces the reference's own outputs."""
    from oracle import ref_formulation as rf
    meta, args, want = load_case(name)
    sd = orc.make_state_dict(meta.get("weights", "seeded"), meta["net"], meta["weight_seed"], meta["use_cf"])
    m = rf.build(meta["net"], sd, meta["use_cf"])
    with torch.no_grad():
        got = m(*args)
    assert got.shape == want.shape
    valid = torch.isfinite(want)
    assert float((got - want)[valid].abs().max()) < 1e-5 * max(1.0, meta["out_absmax"])
