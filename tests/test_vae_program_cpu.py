"""The VAE modules compile themselves into flat bg_vae_op programs; bg_vae_workspace_bytes (host-only: shape inference +
workspace planning, no launch) must accept all four and reject malformed ones.  No GPU needed."""
import pytest
import torch

import brepgen_amd as bga
from brepgen_amd import _lib, vae
from tests import parity_cases as pc

CASES = [("AutoencoderKLFastDecode", "SURF_CFG", (4, 4, 3)), ("AutoencoderKL1DFastDecode", "EDGE_CFG", (1, 4, 3)),
         ("AutoencoderKLFastEncode", "SURF_CFG", (32, 32, 3)), ("AutoencoderKL1DFastEncode", "EDGE_CFG", (1, 32, 3))]


def _program(cls, cfg, dt):
    m = getattr(bga, cls)(**getattr(pc, cfg))
    return m, m._program(vae._Program(), m._pack(dt)).finish()


@pytest.mark.parametrize("cls,cfg,shape", CASES)
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_programs_plan(cls, cfg, shape, dt):
    lib = _lib.load()
    m, pg = _program(cls, cfg, dt)
    assert pg.steps[-1].dst == vae.VAE_OUT and 2 <= pg.n_slots <= 8
    size = lambda n, chunk: lib.bg_vae_workspace_bytes(pg.ops, len(pg.steps), pg.n_slots, *shape, n, chunk)
    one, many = size(1, 1), size(512, 512)
    assert one > 0 and many > one
    assert size(512, 128) < many                               # chunking bounds the workspace
    assert size(512 + 3, 512) == many                          # a short tail never needs more than the full chunk here
    assert size(100, 512) == size(100, 100)                    # chunk > n: one chunk of n
    # slots are written before they are read, and no step writes its own inputs
    written = {0}
    for o in pg.steps:
        assert o.src in written and (o.res < 0 or o.res in written) and o.dst not in (o.src, o.res)
        written.add(o.dst)


def test_malformed_programs_are_refused():
    lib = _lib.load()
    m, pg = _program("AutoencoderKL1DFastDecode", "EDGE_CFG", torch.bfloat16)
    n = len(pg.steps)
    assert lib.bg_vae_workspace_bytes(pg.ops, n, pg.n_slots, 1, 4, 3, 0, 8) == 0          # nothing to do
    assert lib.bg_vae_workspace_bytes(pg.ops, n, 9, 1, 4, 3, 8, 8) == 0 and b"n_slots" in lib.bg_last_error()
    assert lib.bg_vae_workspace_bytes(pg.ops, n - 1, pg.n_slots, 1, 4, 3, 8, 8) == 0 and b"BG_VAE_OUT" in lib.bg_last_error()
    bad = (_lib.VaeOp * n)(*pg.steps)
    bad[5].res = bad[5].dst
    assert lib.bg_vae_workspace_bytes(bad, n, pg.n_slots, 1, 4, 3, 8, 8) == 0 and b"step 5" in lib.bg_last_error()
    bad = (_lib.VaeOp * n)(*pg.steps)
    bad[2].op = 17
    assert lib.bg_vae_workspace_bytes(bad, n, pg.n_slots, 1, 4, 3, 8, 8) == 0 and b"opcode" in lib.bg_last_error()
