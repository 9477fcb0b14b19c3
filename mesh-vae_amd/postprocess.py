"""Around the train/eval step: what the reference does on the HOST after every batch
(main.py:88-93, :139-145; inference.py:50-51) as one device pass (libmeshvae_hip: mvh_recon_postprocess).

    recon_mesh = out.cpu() * std + mean                       # de-normalise (norm.npz of data.py:166-173)
    recon_mesh = torch.bmm(recon_mesh * s.unsqueeze(1), R) + m  # undo the Procrustes alignment (data.py:144)
    diff = euclidean_distances(recon_mesh, gt_mesh)           # per-vertex error [B, N]

Tensors may live on the device the model runs on; nothing is copied to the host, so the caller can
accumulate `dist.mean()` on the device and read it once per epoch instead of once per batch.
MI355X only: there is no CPU fallback (the oracle's restatement is test infrastructure).
"""
import torch

from meshvae_hip import check, lib


def _prep(out, std, mean, R, m, s):
    if not out.is_cuda:
        raise RuntimeError("postprocess runs on MI355X only (there is no CPU fallback)")
    dev = out.device
    B, N, C = out.shape
    if C != 3:
        raise ValueError("expected vertex tensors of shape [B, N, 3]")
    f = lambda t: torch.as_tensor(t, dtype=torch.float32).to(dev).contiguous()  # noqa: E731
    std, mean, R, m, s = f(std), f(mean), f(R), f(m), f(s)
    if std.shape != (N, 3) or mean.shape != (N, 3):
        raise ValueError(f"std/mean must be [{N}, 3]")
    if R.shape != (B, 3, 3) or m.numel() != B * 3 or s.numel() != B:
        raise ValueError("R must be [B,3,3], m [B,1,3] or [B,3], s [B,1] or [B]")
    return out.contiguous().to(torch.float32), std, mean, R, m.reshape(B, 3), s.reshape(B)


def reconstruction_error(out, std, mean, R, m, s, gt_mesh):
    """-> (recon_mesh [B,N,3], dist [B,N]) of main.py:88-93 in one launch."""
    out, std, mean, R, m, s = _prep(out, std, mean, R, m, s)
    B, N, _ = out.shape
    gt = torch.as_tensor(gt_mesh, dtype=torch.float32).to(out.device).contiguous()
    if gt.shape != out.shape:
        raise ValueError("gt_mesh must have the shape of the reconstruction")
    mesh, dist = torch.empty_like(out), torch.empty(B, N, dtype=torch.float32, device=out.device)
    with torch.cuda.device(out.device):
        check(lib().mvh_recon_postprocess(torch.cuda.current_stream(out.device).cuda_stream, out.data_ptr(),
                                          std.data_ptr(), mean.data_ptr(), R.data_ptr(), m.data_ptr(), s.data_ptr(),
                                          gt.data_ptr(), mesh.data_ptr(), dist.data_ptr(), B, N))
    return mesh, dist


def reconstruct(out, std, mean, R, m, s):
    """-> recon_mesh [B,N,3]: `torch.bmm((out * std + mean) * s, R) + m` (main.py:88-90)."""
    out, std, mean, R, m, s = _prep(out, std, mean, R, m, s)
    B, N, _ = out.shape
    mesh = torch.empty_like(out)
    with torch.cuda.device(out.device):
        check(lib().mvh_recon_postprocess(torch.cuda.current_stream(out.device).cuda_stream, out.data_ptr(),
                                          std.data_ptr(), mean.data_ptr(), R.data_ptr(), m.data_ptr(), s.data_ptr(),
                                          None, mesh.data_ptr(), None, B, N))
    return mesh


def euclidean_distances(gt, pred):
    """Per-vertex distance [.., N] (reference inference.py:50-51 / crecon.py:61-62) for tensors on any device."""
    return (torch.as_tensor(gt) - torch.as_tensor(pred)).pow(2).sum(-1).sqrt()
