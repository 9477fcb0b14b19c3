"""Input side of the step: dataset alignment and per-batch normalisation on the device
(reference data.py:103-111,139-161 and utils.py:58-157).

The reference aligns every mesh to the template with a Procrustes fit in numpy when the dataset is
built, keeps the aligned meshes on the host, and per item computes `(mesh - mean) / std` in fp64 and a
float copy, which the loader then collates and copies to the device every step.  Here the whole
aligned dataset lives in HBM (1076 meshes x 4998 x 3 doubles = 129 MB of 288 GB):

  procrustes_batch   centroid / norm / 3x3 cross-covariance per mesh (mvh_procrustes_stats), the 3x3 SVD
                     of scipy's orthogonal_procrustes on the host (13 numbers per mesh), rotation + scale
                     + disparity on the device (mvh_procrustes_apply)
  procrustes         the reference's single-mesh signature over the same kernels
  DeviceDataset      aligned meshes, mean/std, labels and the per-mesh (R, s, m) of data.py:159-161 on
                     the device; `batch(idx)` gathers and normalises a batch in one launch
                     (mvh_gather_normalize, bit-identical to the reference's torch ops) -- no H2D per step

MI355X only: there is no CPU fallback (the oracle's restatement is test infrastructure).
"""
import os

import numpy as np
import torch
from scipy.linalg import svd as _svd

import mesh_operations
from meshvae_hip import check, lib


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def standardize(template):
    """Centred, unit-Frobenius-norm copy of the template (utils.py:120,136,141,147) -> numpy [N,3] double."""
    mtx1 = np.array(template, dtype=np.double, copy=True)
    if mtx1.ndim != 2:
        raise ValueError("Input matrices must be two-dimensional")
    if mtx1.size == 0:
        raise ValueError("Input matrices must be >0 rows and >0 cols")
    mtx1 -= np.mean(mtx1, 0)
    norm1 = np.linalg.norm(mtx1)
    if norm1 == 0:
        raise ValueError("Input matrices must contain >1 unique points")
    mtx1 /= norm1
    return mtx1


def procrustes_batch(template, pts, device="cuda:0"):
    """Align pts [B,N,3] to template [N,3] (utils.procrustes per mesh).  Returns a dict of fp64 tensors
    on `device`: mtx1 [N,3], aligned [B,N,3] (the reference's mtx2), disparity [B], R [B,3,3],
    s [B] (= norm2 / scale, the reference's res[1]) and m [B,3] (the centroid, res[2])."""
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("preprocess runs on MI355X only (there is no CPU fallback)")
    mtx1 = torch.from_numpy(standardize(template)).to(device)
    pts = torch.as_tensor(pts, dtype=torch.float64).to(device).contiguous()
    if pts.dim() != 3 or pts.shape[1:] != mtx1.shape:
        raise ValueError("Input matrices must be of same shape")
    B, N, _ = pts.shape
    stats = torch.empty(B, 13, dtype=torch.float64, device=device)
    L = lib()
    with torch.cuda.device(device):
        check(L.mvh_procrustes_stats(_stream(device), mtx1.data_ptr(), pts.data_ptr(), stats.data_ptr(), B, N))
        st = stats.cpu().numpy()
        if np.any(st[:, 3] == 0):
            raise ValueError("Input matrices must contain >1 unique points")
        R, scale = np.empty((B, 3, 3)), np.empty(B)
        for b in range(B):                              # scipy.linalg.orthogonal_procrustes (utils.py:151)
            u, w, vt = _svd(st[b, 4:].reshape(3, 3))
            R[b], scale[b] = u.dot(vt), w.sum()
        R_d, s_d = torch.from_numpy(R).to(device), torch.from_numpy(scale).to(device)
        aligned = torch.empty_like(pts)
        disparity = torch.empty(B, dtype=torch.float64, device=device)
        check(L.mvh_procrustes_apply(_stream(device), mtx1.data_ptr(), pts.data_ptr(), stats.data_ptr(), R_d.data_ptr(),
                                     s_d.data_ptr(), aligned.data_ptr(), disparity.data_ptr(), B, N))
    return {"mtx1": mtx1, "aligned": aligned, "disparity": disparity, "R": R_d,
            "s": torch.from_numpy(st[:, 3] / scale).to(device), "m": stats[:, :3].clone()}


def procrustes(data1, data2, device="cuda:0"):
    """The reference's signature (utils.py:58): -> (mtx1, mtx2, disparity, [R, norm2 / s, centroid2]) as numpy."""
    d2 = np.asarray(data2, dtype=np.double)
    if np.asarray(data1).ndim != 2 or d2.ndim != 2:
        raise ValueError("Input matrices must be two-dimensional")
    if np.asarray(data1).shape != d2.shape:
        raise ValueError("Input matrices must be of same shape")
    r = procrustes_batch(data1, d2[None], device)
    return (r["mtx1"].cpu().numpy(), r["aligned"][0].cpu().numpy(), float(r["disparity"][0]),
            [r["R"][0].cpu().numpy(), float(r["s"][0]), r["m"][0].cpu().numpy()])


def list_meshes(config, get_sex_from_file_name=True):
    """-> (dataset_index, labels): what the reference's listMeshes hands to MeshData (data.py:40-72).

    dataset_index = the `*.obj` entries of config['root_dir'] in sorted order, without those whose name is the
    first blank-separated token of a line of config['error_file'] (empty string = no reject list);
    labels[name] = 0 when the second `_`-separated field of the name is "f", else 1; -1 for every name when
    `get_sex_from_file_name` is false."""
    rejects = set()
    if config.get("error_file"):
        with open(config["error_file"]) as fh:
            rejects = {line.partition(" ")[0] for line in fh.read().split("\n")}
    found = sorted(e.name for e in os.scandir(config["root_dir"]) if e.name.endswith(".obj"))
    dataset_index = [n for n in found if n not in rejects]

    def label(name):
        if not get_sex_from_file_name:
            return -1
        return 0 if name.split("_")[1] == "f" else 1

    labels = {n: label(n) for n in dataset_index}
    print(f"{len(found)} OBJ files under {config['root_dir']}: {len(found) - len(dataset_index)} on the reject list, "
          f"{len(dataset_index)} kept")
    return dataset_index, labels


def save_obj(filename, vertices, faces):
    """`v x y z` / 1-based `f a b c` records (reference data.py:20-27)."""
    with open(filename, "w") as fp:
        for v in vertices:
            fp.write("v %f %f %f\n" % (v[0], v[1], v[2]))
        for f in np.asarray(faces) + 1:
            fp.write("f %d %d %d\n" % (f[0], f[1], f[2]))


class DeviceDataset:
    """MeshData (data.py:75-200) with everything resident on the device.

    meshes [M,N,3] raw vertex arrays, labels [M]; `norm` = (mean, std) of a training split (norm.npz of
    data.py:166-178) or None to compute them from this split (np.mean / np.std over meshes, axis 0).
    """

    def __init__(self, meshes, labels, template, norm=None, device="cuda:0"):
        device = torch.device(device)
        fit = procrustes_batch(template, meshes, device)
        self.device = device
        self.ori_data = fit["aligned"]                                         # data.py:145-146,184
        self.ori_mesh = torch.as_tensor(meshes, dtype=torch.float32).to(device)  # data.py:143
        self.R = fit["R"].float()                                              # data.py:159-161 (FloatTensor)
        self.s = fit["s"].float().reshape(-1, 1)
        self.m = fit["m"].float().reshape(-1, 1, 3)
        self.data_label = torch.as_tensor(labels).to(device)
        if norm is None:
            mean, std = self.ori_data.mean(0), self.ori_data.std(0, unbiased=False)
        else:
            mean, std = (torch.as_tensor(np.asarray(t), dtype=torch.float64).to(device) for t in norm)
        self.mean, self.std = mean.contiguous(), std.contiguous()

    @classmethod
    def from_directory(cls, dataset_index, config, labels, template, dtype="train", device="cuda:0"):
        """MeshData(dataset_index, config, label, template, dtype) (reference data.py:84-200): reads
        config['root_dir']/<name> for every name that exists, aligns, and hands the statistics over through
        config['checkpoint_dir']/norm.npz as the reference effectively does (data.py:166-178): its "already there"
        test looks for a file called `norm`, which np.savez never creates (it appends .npz), so EVERY 'train' split
        recomputes mean/std from its own meshes and overwrites norm.npz; every split then normalises with the
        file's current statistics (a later 'test' split sees the last 'train' split's, e.g. per fold of main.py's
        k-fold loop)."""
        names = [n for n in dataset_index if os.path.exists(os.path.join(config["root_dir"], n))]
        meshes = np.stack([mesh_operations.read_obj(os.path.join(config["root_dir"], n))[0] for n in names])
        ds = cls(meshes, [labels[n] for n in names], template, norm=None, device=device)
        ds.filename = [os.path.join(config["root_dir"], n) for n in names]
        norm_file = os.path.join(config["checkpoint_dir"], "norm.npz")
        if dtype == "train":
            os.makedirs(config["checkpoint_dir"], exist_ok=True)
            np.savez(norm_file, mean=ds.mean.cpu().numpy(), std=ds.std.cpu().numpy())
        stats = np.load(norm_file, allow_pickle=True)
        ds.mean = torch.as_tensor(stats["mean"], dtype=torch.float64).to(ds.device).contiguous()
        ds.std = torch.as_tensor(stats["std"], dtype=torch.float64).to(ds.device).contiguous()
        print(dtype, " dataset has been created, number of {} samples:".format(dtype), len(ds))
        return ds

    def __len__(self):
        return self.ori_data.shape[0]

    def batch(self, idx):
        """-> (x [B,N,3] float32, x_gt [B,N,3] float64, label [B], ori_mesh, R, m, s) for mesh indices idx:
        the tensors of the reference's 8-tuple (data.py:111) a step consumes (main.py:66-71,88-93)."""
        idx = torch.as_tensor(idx, dtype=torch.int64).to(self.device).contiguous()
        if idx.numel() and (int(idx.min()) < 0 or int(idx.max()) >= len(self)):
            raise IndexError("mesh index out of range")
        B, (M, N, _) = idx.numel(), self.ori_data.shape
        x32 = torch.empty(B, N, 3, dtype=torch.float32, device=self.device)
        x64 = torch.empty(B, N, 3, dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            check(lib().mvh_gather_normalize(_stream(self.device), self.ori_data.data_ptr(), M, idx.data_ptr(),
                                             self.mean.data_ptr(), self.std.data_ptr(), x32.data_ptr(), x64.data_ptr(),
                                             B, N * 3))
        return x32, x64, self.data_label[idx], self.ori_mesh[idx], self.R[idx], self.m[idx], self.s[idx]
