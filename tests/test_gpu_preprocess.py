"""GPU (-m gpu): input-side pre-processing -- device Procrustes alignment against the
vectors the reference's utils.procrustes produced, and the batch gather + normalise of MeshData.__getitem__
bit-exact against the torch CPU ops it replaces.

Tolerances: fp64 throughout; the device sums run in a different (fixed) order than numpy's, so alignment
agrees to ~1e-13 relative, asserted at 1e-11; the gather/normalise is integer-indexed IEEE sub/div: bit-exact.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


@pytest.mark.parametrize("tag", ["tiny", "5k"])
def test_procrustes_batch_matches_reference(tag):
    from preprocess import procrustes, procrustes_batch
    npz = load_golden("procrustes.npz")
    dev = _dev()
    r = procrustes_batch(npz[f"{tag}/template"], npz[f"{tag}/pts"], dev)
    np.testing.assert_allclose(r["mtx1"].cpu().numpy(), npz[f"{tag}/mtx1"][0], rtol=0, atol=1e-15)
    np.testing.assert_allclose(r["aligned"].cpu().numpy(), npz[f"{tag}/mtx2"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(r["R"].cpu().numpy(), npz[f"{tag}/R"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(r["s"].cpu().numpy(), npz[f"{tag}/s"], rtol=1e-11)
    np.testing.assert_allclose(r["m"].cpu().numpy(), npz[f"{tag}/m"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(r["disparity"].cpu().numpy(), npz[f"{tag}/disparity"], rtol=1e-9)
    # the reference's single-mesh signature
    mtx1, mtx2, disparity, (R, s, m) = procrustes(npz[f"{tag}/template"], npz[f"{tag}/pts"][1])
    np.testing.assert_allclose(mtx2, npz[f"{tag}/mtx2"][1], rtol=0, atol=1e-12)
    assert abs(disparity - npz[f"{tag}/disparity"][1]) < 1e-9 * npz[f"{tag}/disparity"][1]
    # bitwise reproducible (fixed-order sums)
    r2 = procrustes_batch(npz[f"{tag}/template"], npz[f"{tag}/pts"], dev)
    assert torch.equal(r["aligned"], r2["aligned"]) and torch.equal(r["disparity"], r2["disparity"])


def test_procrustes_properties_and_errors():
    """Size-independent properties at B = 64 x 4998: aligning an exact similarity copy gives disparity ~ 0
    and recovers the transform; the inverse map of main.py:88-90 (aligned * s @ R + m) returns the input."""
    from preprocess import procrustes_batch
    dev = _dev()
    g = np.random.default_rng(9)
    template = g.standard_normal((4998, 3))
    pts = []
    for b in range(64):
        q, _ = np.linalg.qr(g.standard_normal((3, 3)))
        pts.append(template @ q * g.uniform(0.1, 100.0) + g.standard_normal(3) * 50.0)
    pts = np.stack(pts)
    r = procrustes_batch(template, pts, dev)
    assert float(r["disparity"].max()) < 1e-20
    back = torch.bmm(r["aligned"] * r["s"].reshape(-1, 1, 1), r["R"]) + r["m"].reshape(-1, 1, 3)
    np.testing.assert_allclose(back.cpu().numpy(), pts, rtol=0, atol=1e-10)
    eye = torch.bmm(r["R"], r["R"].transpose(1, 2)).cpu().numpy()
    np.testing.assert_allclose(eye, np.broadcast_to(np.eye(3), eye.shape), atol=1e-13)
    with pytest.raises(ValueError, match="same shape"):
        procrustes_batch(template, pts[:, :-1], dev)
    with pytest.raises(ValueError, match="unique points"):
        procrustes_batch(template, np.ones((2, 4998, 3)), dev)
    assert procrustes_batch(template, pts[:0], dev)["aligned"].shape == (0, 4998, 3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        procrustes_batch(template, pts[:1], "cpu")


def test_device_dataset_batches_bit_exact():
    """DeviceDataset.batch vs the oracle's restatement of data.py:103-111 on the same aligned meshes."""
    from oracle import cheb_oracle as O
    from preprocess import DeviceDataset
    npz = load_golden("procrustes.npz")
    dev = _dev()
    g = np.random.default_rng(4)
    template = npz["tiny/template"]
    meshes = np.stack([template @ np.linalg.qr(g.standard_normal((3, 3)))[0] * g.uniform(1, 5)
                       + g.standard_normal(template.shape) * 0.02 for _ in range(37)])
    labels = np.arange(37) % 2
    ds = DeviceDataset(meshes, labels, template, device=dev)
    assert len(ds) == 37 and ds.R.dtype == torch.float32 and ds.s.shape == (37, 1) and ds.m.shape == (37, 1, 3)
    aligned = ds.ori_data.cpu().numpy()
    np.testing.assert_allclose(ds.mean.cpu().numpy(), np.mean(aligned, axis=0), rtol=0, atol=1e-15)   # data.py:169-170
    np.testing.assert_allclose(ds.std.cpu().numpy(), np.std(aligned, axis=0), rtol=1e-12)
    for idx in ([5, 0, 36, 5, 12], list(range(37)), [3]):
        x, x_gt, y, ori, R, m, s = ds.batch(idx)
        want32, want64 = O.normalize_items(aligned[idx], ds.mean.cpu().numpy(), ds.std.cpu().numpy())
        assert x.dtype == torch.float32 and x_gt.dtype == torch.float64
        assert torch.equal(x_gt.cpu(), want64) and torch.equal(x.cpu(), want32)
        assert torch.equal(y.cpu(), torch.as_tensor(labels[idx])) and ori.shape == (len(idx), 162, 3)
        assert torch.equal(R, ds.R[idx]) and s.shape == (len(idx), 1) and m.shape == (len(idx), 1, 3)
    # a held-out split normalised with the training statistics (norm.npz, data.py:176-184)
    test = DeviceDataset(meshes[:5], labels[:5], template, norm=(ds.mean.cpu().numpy(), ds.std.cpu().numpy()), device=dev)
    assert torch.equal(test.batch([1])[1], ds.batch([1])[1])
    with pytest.raises(IndexError):
        ds.batch([37])
    assert ds.batch([])[0].shape == (0, 162, 3)


def test_every_train_split_rewrites_norm_npz(tmp_path, capsys):
    """data.py:166-173 looks for `norm` (np.savez writes norm.npz), so the reference recomputes and overwrites the
    statistics for EVERY 'train' split: two train splits in one checkpoint_dir must leave the second one's
    statistics in norm.npz, and a 'test' split built afterwards normalises with those."""
    from preprocess import DeviceDataset, list_meshes, save_obj
    from meshgen import torus_mesh
    dev = _dev()
    v, f = torus_mesh(8, 12)
    g = np.random.default_rng(3)
    root = tmp_path / "data"
    root.mkdir()
    for k in range(12):
        save_obj(str(root / f"{k:03d}_{'f' if k % 2 else 'm'}_0.obj"), v * (1.0 + 0.1 * k) + g.standard_normal(v.shape) * 0.02, f)
    cfg = {"root_dir": str(root), "error_file": "", "checkpoint_dir": str(tmp_path / "ckpt")}
    index, labels = list_meshes(cfg)
    a = DeviceDataset.from_directory(index[:6], cfg, labels, v, dtype="train", device=dev)
    first = np.load(tmp_path / "ckpt" / "norm.npz")["mean"].copy()
    b = DeviceDataset.from_directory(index[6:], cfg, labels, v, dtype="train", device=dev)
    second = np.load(tmp_path / "ckpt" / "norm.npz")["mean"]
    assert not np.array_equal(first, second)                                   # overwritten, not kept
    np.testing.assert_array_equal(second, b.ori_data.mean(0).cpu().numpy())    # ... with the second split's own
    np.testing.assert_array_equal(a.mean.cpu().numpy(), first)
    t = DeviceDataset.from_directory(index[:3], cfg, labels, v, dtype="test", device=dev)
    np.testing.assert_array_equal(t.mean.cpu().numpy(), second)
    capsys.readouterr()


def test_dataset_round_trip_through_postprocess():
    """pre- and post-processing are inverses: de-normalising and un-aligning the network *input* with
    postprocess.reconstruction_error (main.py:88-93) returns the original mesh (fp32 tolerance)."""
    from postprocess import reconstruction_error
    from preprocess import DeviceDataset
    npz = load_golden("procrustes.npz")
    dev = _dev()
    ds = DeviceDataset(npz["5k/pts"], [0, 1], npz["5k/template"], device=dev)
    x, x_gt, y, ori, R, m, s = ds.batch([1, 0])
    mesh, dist = reconstruction_error(x, ds.std.float(), ds.mean.float(), R, m, s, ori)
    scale = float(ori.abs().max())
    assert float(dist.max()) < 2e-5 * scale, (float(dist.max()), scale)


def test_end_to_end_example_on_a_fake_dataset(tmp_path, capsys, monkeypatch):
    """examples/train_fake_dataset.py (BASELINE configs[0]-shaped plumbing): OBJ files -> loader -> hierarchy from the
    template OBJ -> native train steps with the LR table -> on-device evaluation.  The loss must fall and stay finite."""
    import importlib.util
    import os
    import re
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("train_fake_dataset", os.path.join(root, "examples", "train_fake_dataset.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.setattr(sys, "argv", ["train_fake_dataset.py", "--meshes", "32", "--epochs", "2", "--batch", "8",
                                      "--workdir", str(tmp_path)])
    mod.main()
    out = capsys.readouterr().out
    losses = [float(v) for v in re.findall(r"train loss ([0-9.]+)", out)]
    assert len(losses) == 2 and all(np.isfinite(losses)) and losses[1] < losses[0]
    assert "test: 8 meshes" in out and os.path.exists(os.path.join(str(tmp_path), "ckpt", "norm.npz"))
    assert os.path.exists(os.path.join(str(tmp_path), "ckpt", "initial_weight.pt"))
