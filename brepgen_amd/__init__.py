"""brepgen_amd -- MI355X-native (gfx950) implementation of BrepGen's latent-diffusion denoising hot path.

Drop-in for the reference's call surface on that path only:
    SurfPosNet / SurfZNet / EdgePosNet / EdgeZNet    (network.py)   -> bg_denoiser_fwd
    DDPMScheduler / PNDMScheduler                    (schedulers.py) -> bg_cfg_ddpm_step / bg_pndm_step
    randn_tensor                                     (utils.py)
    AutoencoderKLFastDecode / AutoencoderKL1DFastDecode (vae.py)    -> bg_im2col + GEMM, bg_small_attn, ...
All compute goes through libbrepgen_hip.so (hand-written HIP kernels behind a C ABI, include/brepgen_hip.h).
"""
from .network import EdgePosNet, EdgeZNet, SurfPosNet, SurfZNet  # noqa: F401
from .schedulers import DDPMScheduler, PNDMScheduler  # noqa: F401
from .utils import randn_tensor  # noqa: F401
from .vae import (AutoencoderKL1DFastDecode, AutoencoderKL1DFastEncode, AutoencoderKLFastDecode,  # noqa: F401
                  AutoencoderKLFastEncode)

__all__ = ["SurfPosNet", "SurfZNet", "EdgePosNet", "EdgeZNet", "DDPMScheduler", "PNDMScheduler", "randn_tensor",
           "AutoencoderKLFastDecode", "AutoencoderKL1DFastDecode", "AutoencoderKLFastEncode",
           "AutoencoderKL1DFastEncode"]
