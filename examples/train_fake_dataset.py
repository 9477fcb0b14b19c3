#!/usr/bin/env python3
"""BASELINE configs[0]-shaped plumbing on the MI355X path: a synthetic fake dataset of OBJ files (template + noise,
named `<id>_f_<k>.obj` / `<id>_m_<k>.obj` so the label parsing of data.py:64-69 applies) goes through this package's
own loader (preprocess.list_meshes / DeviceDataset: Procrustes alignment and normalisation on the device), the model
factory (model.get_model builds the hierarchy from the template OBJ), the native train step with the reference's
optimizer and LR table (main.py:251,266-269), and the evaluation block of main.py:129-159 (de-normalise, inverse
Procrustes, per-vertex error) -- nothing touches the host between the loader and the reported numbers.

    python examples/train_fake_dataset.py [--meshes 64] [--epochs 3] [--batch 16] [--template some.obj]
"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mesh-vae_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--meshes", type=int, default=64)
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--template", default=None, help="template OBJ (default: this repo's procedural 4998-vertex torus)")
    ap.add_argument("--workdir", default=None)
    a = ap.parse_args()
    from meshvae_hip.engine import TrainStep
    from model import get_model
    from postprocess import reconstruction_error
    from preprocess import DeviceDataset, list_meshes, save_obj

    work = a.workdir or tempfile.mkdtemp(prefix="meshvae_fake_")
    if a.template:
        import mesh_operations
        verts, faces = mesh_operations.read_obj(a.template)
    else:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from meshgen import torus_mesh
        verts, faces = torus_mesh(51, 98)
    data_dir = os.path.join(work, "data")
    os.makedirs(data_dir, exist_ok=True)
    template_obj = os.path.join(work, "template.obj")
    save_obj(template_obj, verts, faces)
    g = np.random.default_rng(0)
    span = np.ptp(verts, axis=0).max()
    for k in range(a.meshes):                       # females slightly wider, males slightly taller: a learnable label
        sex = "f" if k % 2 == 0 else "m"
        scale = np.array([1.06, 1.0, 1.0]) if sex == "f" else np.array([1.0, 1.06, 1.0])
        q, _ = np.linalg.qr(g.standard_normal((3, 3)))
        pts = (verts * scale + g.standard_normal(verts.shape) * 0.004 * span) @ q * g.uniform(0.8, 1.2) + g.standard_normal(3)
        save_obj(os.path.join(data_dir, f"{k:04d}_{sex}_0.obj"), pts, faces)
    config = {"root_dir": data_dir, "error_file": "", "checkpoint_dir": os.path.join(work, "ckpt"), "template": template_obj,
              "type": "cheb_VAE", "model": "optimal_sigma_VAE", "n_layers": 4, "num_conv_filters": [16, 16, 16, 32, 32],
              "polygon_order": [6] * 5, "downsampling_factors": [4, 4, 4, 4], "num_classes": 2, "num_style": 16,
              "num_hidden": 512, "dropout": 0.2, "learning_rate": 1e-3, "weight_decay": 5e-4,
              "learning_rates_epochs": [1, 2], "learning_rates": [5e-4, 2.5e-4]}
    dev = torch.device("cuda:0")
    t0 = time.perf_counter()
    index, labels = list_meshes(config)
    n_train = int(0.75 * len(index)) // a.batch * a.batch
    template = verts
    train = DeviceDataset.from_directory(index[:n_train], config, labels, template, dtype="train", device=dev)
    test = DeviceDataset.from_directory(index[n_train:], config, labels, template, dtype="test", device=dev)
    print(f"loader: {time.perf_counter() - t0:.2f} s for {len(index)} OBJ files (read + align + normalise)")
    torch.manual_seed(666)
    t0 = time.perf_counter()
    net = get_model(config, dev)
    print(f"get_model (hierarchy from the template OBJ): {time.perf_counter() - t0:.2f} s")
    net.train()
    step = TrainStep(net, a.batch, lr=config["learning_rate"], weight_decay=config["weight_decay"], use_graph=False)
    std, mean = train.std.float(), train.mean.float()
    gen = torch.Generator().manual_seed(1)
    for epoch in range(1, a.epochs + 1):
        lr = step.set_epoch(config, epoch)
        perm = torch.randperm(len(train), generator=gen)
        tot, correct, err = 0.0, 0, 0.0
        for i in range(0, len(train), a.batch):
            x, x_gt, label, ori, R, m, s = train.batch(perm[i:i + a.batch])
            step.load(x, x_gt, torch.nn.functional.one_hot(label, 2))
            loss, corr, recon = step.step()
            _, dist = reconstruction_error(recon, std, mean, R, m, s, ori)       # main.py:88-93, on the device
            tot, correct, err = tot + float(loss), correct + int(corr), err + float(dist.mean())
        nb = len(train) // a.batch
        print(f"epoch {epoch} lr {lr:g}  train loss {tot / nb:.1f}  acc {correct / len(train):.2f}  mean vertex error {err / nb:.4f}")
    net.eval()
    with torch.no_grad():
        x, x_gt, label, ori, R, m, s = test.batch(list(range(len(test))))
        h = net.encoder(x)
        y_hat = net.classifier(h)
        pred = torch.argmax(y_hat, 1)
        mu = torch.nn.functional.linear(torch.cat([torch.nn.functional.one_hot(label, 2).float(), h], -1),
                                        net.z_mean.weight, net.z_mean.bias)
        recon = net.sample(torch.nn.functional.one_hot(label, 2), mu)
        _, dist = reconstruction_error(recon, std, mean, R, m, s, ori)
    print(f"test: {len(test)} meshes, classifier acc {float((pred == label).float().mean()):.2f}, "
          f"mean vertex error {float(dist.mean()):.4f} (mesh extent {span:.2f})")


if __name__ == "__main__":
    main()
