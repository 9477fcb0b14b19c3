"""Device-side helpers of the contrastive-reconstruction step (reference crecon.py:53-61, 160-198).

The reference's crecon.py has these as script-level functions around `net.encoder / classifier /
z_mean / sample`; with mesh-vae_amd ahead of the reference on sys.path they already run on the HIP
kernels unchanged.  This module is the same pair for callers that drive the package directly, with the
two decodes (own label, opposite label) batched into one 2B-mesh decoder pass.
"""
import torch
import torch.nn.functional as F

from meshvae_hip import functional as F_hip


def classifier_(net, x):
    """argmax of the VAE's own classifier head (crecon.py:53-61)."""
    return torch.argmax(net.classifier(net.encoder(x)), dim=1)


def estimate_diff_device(net, x, y, dtype):
    """estimate_diff without the host read-back: -> (diff [B,N,6], #correct as a 0-d device tensor).
    Nothing here synchronises, so the call can sit inside a hipGraph capture (engine.ClassifierStep)."""
    with torch.no_grad():
        h = net.encoder(x)
        index_pred = torch.argmax(net.classifier(h), dim=1)
        correct = torch.sum(index_pred == y)
        sex_hot = F.one_hot(y if dtype == "train" else index_pred, num_classes=2)
        x_mean = F_hip.linear(torch.cat([sex_hot.to(h.dtype), h], -1), net.z_mean.weight, net.z_mean.bias)
        B = x_mean.shape[0]
        # own-label and opposite-label decodes share one decoder pass (the decoder is per-mesh)
        both = net.sample(torch.cat([sex_hot, 1 - sex_hot]), torch.cat([x_mean, x_mean]))
        recon, recon_oppo = both[:B], both[B:]
        return torch.cat((x - recon_oppo, x - recon), dim=-1), correct


def estimate_diff(net, x, y, dtype):
    """([x - recon_opposite, x - recon] on the channel axis, #correct) -- crecon.py:160-198.

    x [B, N, 3] (or one mesh [N, 3] with an int label), y [B] int64 labels.  "train" conditions on the
    true label, anything else on the predicted one.  As in the reference the VAE runs under no_grad in
    whatever mode (`net.training`) the caller left it.
    """
    if x.dim() == 2:
        x = x.reshape(1, -1, 3)
        y = torch.as_tensor(y, device=x.device).unsqueeze(0)
    diff, correct = estimate_diff_device(net, x, y, dtype)
    return diff, correct.item()
