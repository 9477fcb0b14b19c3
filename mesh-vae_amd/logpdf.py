"""`logpdf` -- stand-alone versions of the three reference functions cheb_VAE uses
(reference logpdf.py: KLD :7-8, gaussian_nll :22-23, softclip :24-28).

Inside the model these are fused into the HIP loss kernel (mvh_vae_loss_fwd/_bwd in
libmeshvae_hip); the functions below only keep the module's public names importable for
callers of the reference API and are ordinary tensor expressions.
"""
import math

import torch
from torch.nn.functional import softplus

HALF_LOG_2PI = 0.5 * math.log(2.0 * math.pi)
C = -HALF_LOG_2PI


def KLD(mu, logvar):
    """KL(N(mu, e^logvar) || N(0, 1)) summed over the last axis."""
    inner = 1 + logvar - mu.pow(2) - logvar.exp()
    return inner.sum(dim=-1).mul(-0.5)


def gaussian_nll(mu, log_sigma, x):
    """Element-wise negative log-likelihood of x under N(mu, e^(2 log_sigma))."""
    standardized = (x - mu) / log_sigma.exp()
    return 0.5 * torch.pow(standardized, 2) + log_sigma + HALF_LOG_2PI


def softclip(tensor, min):
    """Soft lower clip: min + softplus(tensor - min)."""
    return softplus(tensor - min) + min
