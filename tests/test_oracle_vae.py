"""CPU checks of the restated VAE decoders (parity unpinned: diffusers is absent -- see oracle/vae.py)."""
import numpy as np
import torch

import brepgen_amd as bga
from oracle import vae as ov

SURF_CFG = dict(in_channels=3, out_channels=3, down_block_types=["DownEncoderBlock2D"] * 4,
                up_block_types=["UpDecoderBlock2D"] * 4, block_out_channels=[128, 256, 512, 512], layers_per_block=2,
                act_fn="silu", latent_channels=3, norm_num_groups=32, sample_size=512)
EDGE_CFG = dict(in_channels=3, out_channels=3, down_block_types=["DownBlock1D"] * 3, up_block_types=["UpBlock1D"] * 3,
                block_out_channels=[128, 256, 512], layers_per_block=2, act_fn="silu", latent_channels=3,
                norm_num_groups=32, sample_size=512)


def test_decoder_parameter_counts_match_survey():
    # 49.49 M (surface decoder, the SD-VAE decoder size) and 39.12 M (edge decoder)
    n2 = sum(int(np.prod(s)) for s in ov.surf_decoder_spec().values())
    n1 = sum(int(np.prod(s)) for s in ov.edge_decoder_spec().values())
    assert abs(n2 - 49.49e6) < 0.01e6 and abs(n1 - 39.12e6) < 0.01e6


def test_shapes_and_token_layout():
    with torch.no_grad():
        y = ov.surf_decode(ov.seeded_state_dict(ov.surf_decoder_spec(), 1), torch.zeros(2, 3, 4, 4))
        assert y.shape == (2, 3, 32, 32)
        y = ov.edge_decode(ov.seeded_state_dict(ov.edge_decoder_spec(), 2), torch.zeros(5, 3, 4))
        assert y.shape == (5, 3, 32)


def test_cubic_upsample_depthwise_equals_dense_transposed_conv():
    x = torch.randn(2, 7, 8, generator=torch.Generator().manual_seed(0))
    assert float((ov.upsample1d_cubic(x) - ov.upsample1d_cubic_taps(x)).abs().max()) < 1e-6
    # partition of unity: a constant signal stays constant
    c = torch.ones(1, 1, 6)
    assert torch.allclose(ov.upsample1d_cubic(c), torch.ones(1, 1, 12), atol=1e-6)


def test_module_keys_match_checkpoint_layout():
    m2 = bga.AutoencoderKLFastDecode(**SURF_CFG)
    m1 = bga.AutoencoderKL1DFastDecode(**EDGE_CFG)
    assert set(m2.state_dict()) == set(ov.surf_decoder_spec())
    assert set(m1.state_dict()) == set(ov.edge_decoder_spec())
    sd = ov.seeded_state_dict(ov.edge_decoder_spec(), 3)
    sd["encoder.conv_in.weight"] = torch.zeros(1)          # full-VAE checkpoints carry the encoder too (strict=False)
    r = m1.load_state_dict(sd, strict=False)
    assert not r.missing_keys and r.unexpected_keys == ["encoder.conv_in.weight"]


def test_encoder_shapes_and_keys():
    with torch.no_grad():
        y = ov.surf_encode(ov.seeded_state_dict(ov.surf_encoder_spec(), 1), torch.zeros(2, 3, 32, 32))
        assert y.shape == (2, 3, 4, 4)
        y = ov.edge_encode(ov.seeded_state_dict(ov.edge_encoder_spec(), 2), torch.zeros(3, 3, 32))
        assert y.shape == (3, 3, 4)
    assert set(bga.AutoencoderKLFastEncode(**SURF_CFG).state_dict()) == set(ov.surf_encoder_spec())
    assert set(bga.AutoencoderKL1DFastEncode(**EDGE_CFG).state_dict()) == set(ov.edge_encoder_spec())
    # a constant signal stays constant under the cubic down-sampler (kernel sums to 1)
    assert torch.allclose(ov.downsample1d_cubic(torch.ones(1, 1, 8)), torch.ones(1, 1, 4), atol=1e-6)
