"""Host logic of brepgen_amd.pipeline (the eval_config.yaml driver) -- no GPU needed."""
import os

import pytest
import torch
import yaml

from brepgen_amd import _lib, pipeline
from oracle import vae as ov

CONFIG = """
deepcad:
  surfpos_weight: deepcad_ldm_surfpos.pt
  surfz_weight:   deepcad_ldm_surfz.pt
  edgepos_weight: deepcad_ldm_edgepos.pt
  edgez_weight:   deepcad_ldm_edgez.pt
  surfvae_weight: deepcad_vae_surf.pt
  edgevae_weight: deepcad_vae_edge.pt
  save_folder:    samples_deepcad
  batch_size:     16
  z_threshold:    0.2
  bbox_threshold: 0.08
  num_surfaces:   30
  num_edges:      30
  use_cf:         False
  class_label:    []
furniture:
  surfpos_weight: furniture_ldm_surfpos.pt
  surfz_weight:   furniture_ldm_surfz.pt
  edgepos_weight: furniture_ldm_edgepos.pt
  edgez_weight:   furniture_ldm_edgez.pt
  surfvae_weight: furniture_vae_surf.pt
  edgevae_weight: furniture_vae_edge.pt
  save_folder:    samples_furniture
  batch_size:     16
  z_threshold:    0.2
  bbox_threshold: 0.08
  num_surfaces:   60
  num_edges:      40
  use_cf:         True
  class_label:    chair
broken:
  batch_size: 4
"""


@pytest.fixture()
def cfg(tmp_path):
    p = tmp_path / "eval_config.yaml"
    p.write_text(CONFIG)
    return str(p)


def test_config_sections_in_the_reference_format(cfg):
    a = pipeline.load_eval_args(cfg, "deepcad")
    assert a["num_surfaces"] == 30 and a["num_edges"] == 30 and a["use_cf"] is False and a["bbox_threshold"] == 0.08
    assert pipeline.class_id(a) == 0
    f = pipeline.load_eval_args(cfg, "furniture")
    assert f["use_cf"] is True and pipeline.class_id(f) == 6              # 'chair' -> 6, sample.py:21-32
    with pytest.raises(KeyError):
        pipeline.load_eval_args(cfg, "abc")                               # section absent
    with pytest.raises(KeyError):
        pipeline.load_eval_args(cfg, "broken")                            # keys missing
    f["class_label"] = "spaceship"
    with pytest.raises(KeyError):
        pipeline.class_id(f)


def test_class_table_matches_sample_py():
    assert pipeline.TEXT2INT == {"uncond": 0, "bathtub": 1, "bed": 2, "bench": 3, "bookshelf": 4, "cabinet": 5,
                                 "chair": 6, "couch": 7, "lamp": 8, "sofa": 9, "table": 10}


def test_vae_files_hold_the_full_autoencoder_and_load_non_strict(tmp_path):
    """sample.py:83,98 load the full-VAE checkpoints into the decode-only modules with strict=False."""
    full = dict(ov.seeded_state_dict(ov.surf_decoder_spec(), 1))
    n_dec = len(full)
    full.update(ov.seeded_state_dict(ov.surf_encoder_spec(), 2))          # encoder.*, quant_conv.*
    assert len(full) > n_dec
    path = str(tmp_path / "vae_surf.pt")
    torch.save(full, path)
    from brepgen_amd import AutoencoderKLFastDecode
    m = pipeline._load(AutoencoderKLFastDecode(**pipeline.SURF_VAE_CFG), path, False, "cpu")
    assert not m.training
    key = next(k for k in full if k.startswith("decoder.conv_in.weight"))
    assert torch.equal(m.state_dict()[key], full[key])
    del full[key]
    torch.save(full, path)
    with pytest.raises(RuntimeError):                                      # a MISSING decoder tensor is an error
        pipeline._load(AutoencoderKLFastDecode(**pipeline.SURF_VAE_CFG), path, False, "cpu")


def test_driver_refuses_to_run_without_the_gpu(cfg):
    with pytest.raises(_lib.BrepgenHipError):
        pipeline.build(pipeline.load_eval_args(cfg, "deepcad"), device="cpu")


def test_vae_configs_are_the_ones_of_sample_py():
    assert pipeline.SURF_VAE_CFG["block_out_channels"] == [128, 256, 512, 512] and pipeline.SURF_VAE_CFG["latent_channels"] == 3
    assert pipeline.EDGE_VAE_CFG["block_out_channels"] == [128, 256, 512] and pipeline.EDGE_VAE_CFG["up_block_types"] == ["UpBlock1D"] * 3
    assert yaml.safe_load(CONFIG)["deepcad"]["surfvae_weight"] == "deepcad_vae_surf.pt" and os.sep
