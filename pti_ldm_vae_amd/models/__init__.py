from .autoencoder import VAEModel
from .autoencoderkl import AutoencoderKL
from .patch_discriminator import PatchDiscriminator
from .perceptual import PerceptualLoss, SqueezeLPIPS
from .losses import compute_ar_vae_loss, compute_kl_loss, compute_total_loss, fused_recon_kl_loss
from .regression_head import LatentRegressor, VAELatentRegressor

__all__ = ["AutoencoderKL", "LatentRegressor", "PatchDiscriminator", "PerceptualLoss", "SqueezeLPIPS", "VAELatentRegressor", "VAEModel", "compute_ar_vae_loss",
           "compute_kl_loss", "compute_total_loss", "fused_recon_kl_loss"]
