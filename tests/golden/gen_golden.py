#!/usr/bin/env python
"""Generate golden input/output vectors from the REFERENCE's own denoiser classes.

Runs only in the build container (needs /root/reference).  The reference's
``network.py`` imports ``diffusers`` at module import time (network.py:7-14) for
the VAE classes; the four denoisers (network.py:1066-1393) use only ``torch.nn``.
``diffusers`` is not installable offline, so the import is satisfied with empty
stand-in modules -- none of the stand-ins is ever *called* by the denoisers.

For every case: build the reference class, load the oracle's deterministic
synthetic weights (``oracle.denoisers.seeded_state_dict``, or its hostile
``stress_state_dict`` for the ``*_stress_*`` cases -- strict=True, which
also pins the checkpoint key layout), run ``.eval()`` forward in fp32 on CPU,
and store inputs + output in ``tests/golden/<case>.npz``.  The oracle
restatement is checked against the same outputs here and the max-abs diff goes
into ``tests/golden/MANIFEST.json``.

    python tests/golden/gen_golden.py
"""
import importlib
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def _stub_diffusers():
    """Empty stand-ins for the diffusers names network.py imports."""
    class _Any:
        def __init__(self, *a, **k):
            pass

    def _identity_decorator(fn=None, *a, **k):
        return fn

    names = {
        "diffusers": {},
        "diffusers.configuration_utils": {"ConfigMixin": _Any, "register_to_config": _identity_decorator},
        "diffusers.utils": {"BaseOutput": _Any, "is_torch_version": lambda *a, **k: True},
        "diffusers.utils.accelerate_utils": {"apply_forward_hook": _identity_decorator},
        "diffusers.models": {},
        "diffusers.models.attention_processor": {"AttentionProcessor": _Any, "AttnProcessor": _Any,
                                                 "SpatialNorm": _Any},
        "diffusers.models.modeling_utils": {"ModelMixin": torch.nn.Module},
        "diffusers.models.autoencoders": {},
        "diffusers.models.autoencoders.vae": {"Decoder": _Any, "DecoderOutput": _Any,
                                              "DiagonalGaussianDistribution": _Any, "Encoder": _Any},
        "diffusers.models.unets": {},
        "diffusers.models.unets.unet_1d_blocks": {"ResConvBlock": _Any, "SelfAttention1d": _Any,
                                                  "get_down_block": _Any, "get_up_block": _Any,
                                                  "Upsample1d": _Any},
    }
    for mod, attrs in names.items():
        m = types.ModuleType(mod)
        m.__dict__.update(attrs)
        sys.modules[mod] = m


def _rand_mask(g, B, N, min_valid):
    """True = padded; valid entries are left-aligned like sample.py:176-178."""
    m = torch.ones(B, N, dtype=torch.bool)
    for b in range(B):
        nv = int(torch.randint(min_valid, N + 1, (1,), generator=g))
        m[b, :nv] = False
    return m


def cases():
    g = torch.Generator().manual_seed(1234)
    R = lambda *s: torch.randn(*s, generator=g)
    out = []
    # (case name, class name, use_cf, weight seed, kwargs in the reference's argument order[, weights kind])
    out.append(("surfpos_b2_n30", "SurfPosNet", False, 11,
                dict(surfPos=R(2, 30, 6).clamp(-3, 3), timesteps=torch.tensor([995]), class_label=None)))
    out.append(("surfpos_cf_b2_n60", "SurfPosNet", True, 12,
                dict(surfPos=R(4, 60, 6).clamp(-3, 3), timesteps=torch.tensor([249]),
                     class_label=torch.tensor([[6], [6], [0], [0]]))))
    out.append(("surfz_b3_n60", "SurfZNet", False, 13,
                dict(surfZ=R(3, 60, 48), timesteps=torch.tensor([500]), surfPos=R(3, 60, 6).clamp(-3, 3),
                     surf_mask=_rand_mask(g, 3, 60, 8), class_label=None)))
    out.append(("surfz_cf_b2_n17", "SurfZNet", True, 14,
                dict(surfZ=R(2, 17, 48), timesteps=torch.tensor([10]), surfPos=R(2, 17, 6).clamp(-3, 3),
                     surf_mask=_rand_mask(g, 2, 17, 1), class_label=torch.tensor([[3], [0]]))))
    out.append(("edgepos_b2_s6_e5", "EdgePosNet", False, 15,
                dict(edgePos=R(2, 6, 5, 6).clamp(-3, 3), timesteps=torch.tensor([255]),
                     surfPos=R(2, 6, 6).clamp(-3, 3), surfZ=R(2, 6, 48), mask=_rand_mask(g, 2, 6, 2),
                     class_label=None)))
    em = torch.rand(2, 7, 9, generator=g) < 0.4
    em[:, :, 0] = False                                   # first edge always valid (sample.py:261)
    out.append(("edgez_b2_s7_e9", "EdgeZNet", False, 16,
                dict(edge=R(2, 7, 9, 18), timesteps=torch.tensor([980]), edgePos=R(2, 7, 9, 6).clamp(-3, 3),
                     surfPos=R(2, 7, 6).clamp(-3, 3), surfZ=R(2, 7, 48), mask=em, class_label=None)))
    em2 = torch.rand(2, 4, 40, generator=g) < 0.5
    em2[:, :, 0] = False
    out.append(("edgez_cf_b2_s4_e40", "EdgeZNet", True, 17,
                dict(edge=R(2, 4, 40, 18), timesteps=torch.tensor([0]), edgePos=R(2, 4, 40, 6).clamp(-3, 3),
                     surfPos=R(2, 4, 6).clamp(-3, 3), surfZ=R(2, 4, 48), mask=em2,
                     class_label=torch.tensor([[9], [0]]))))
    # HOSTILE weights (oracle.denoisers.stress_state_dict: LayerNorm gains up to 5 / biases ~ 1, heavy-tailed matrices,
    # two outlier channels of ~ 100 x ("outlier") or rows with |mean| ~ 10 x std ("offset")), run through the
    # reference's own classes like every other case: the regime in which a LayerNorm fold cancels instead of normalising
    for kind, ws in (("outlier", 21), ("offset", 22)):
        out.append((f"surfz_stress_{kind}_b3_n60", "SurfZNet", False, ws,
                    dict(surfZ=R(3, 60, 48), timesteps=torch.tensor([500]), surfPos=R(3, 60, 6).clamp(-3, 3),
                         surf_mask=_rand_mask(g, 3, 60, 8), class_label=None), "stress_" + kind))
        em3 = torch.rand(2, 7, 9, generator=g) < 0.4
        em3[:, :, 0] = False
        out.append((f"edgez_stress_{kind}_b2_s7_e9", "EdgeZNet", False, ws + 10,
                    dict(edge=R(2, 7, 9, 18), timesteps=torch.tensor([980]), edgePos=R(2, 7, 9, 6).clamp(-3, 3),
                         surfPos=R(2, 7, 6).clamp(-3, 3), surfZ=R(2, 7, 48), mask=em3, class_label=None),
                    "stress_" + kind))
    return out


def main():
    assert os.path.isdir(REF), "needs the reference checkout"
    _stub_diffusers()
    sys.path.insert(0, REF)
    network = importlib.import_module("network")          # the reference, unmodified
    from oracle import denoisers as orc

    os.makedirs(OUT, exist_ok=True)
    torch.set_grad_enabled(False)
    manifest = {"generator": "tests/golden/gen_golden.py", "torch": torch.__version__,
                "reference": "samxuxiang/BrepGen @ 2024_08_07 network.py (diffusers imports stubbed)",
                "cases": {}}
    for name, cls, use_cf, wseed, kw, *rest in cases():
        weights = rest[0] if rest else "seeded"
        sd = orc.make_state_dict(weights, cls, wseed, use_cf)
        model = getattr(network, cls)(use_cf)
        missing = model.load_state_dict(sd, strict=True)     # pins the key layout
        model.eval()
        ref_kw = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in kw.items()}
        ref = model(**ref_kw)
        mine = orc.FORWARD[cls](sd, *[kw[k] for k in kw])    # same positional order
        diff = float((ref - mine).abs().max())
        arrays = {k: v.numpy() for k, v in kw.items() if v is not None}
        np.savez_compressed(os.path.join(OUT, name + ".npz"), out=ref.numpy(), **arrays)
        manifest["cases"][name] = {"net": cls, "use_cf": use_cf, "weight_seed": wseed, "weights": weights,
                                   "args": list(kw.keys()), "out_absmax": float(ref.abs().max()),
                                   "oracle_vs_reference_maxabs": diff}
        print(f"{name:24s} {cls:11s} out{tuple(ref.shape)} absmax={ref.abs().max():.4f} "
              f"oracle-vs-reference max|d|={diff:.3e}")
        assert diff < 2e-5 * max(1.0, float(ref.abs().max())), "oracle restatement disagrees with the reference"
    with open(os.path.join(OUT, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1)


if __name__ == "__main__":
    main()
