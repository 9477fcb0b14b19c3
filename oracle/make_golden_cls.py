#!/usr/bin/env python3
"""Golden vectors for the crecon classifier path, by RUNNING THE REFERENCE.

TEST INFRASTRUCTURE, build container only.  Imports the unmodified reference modules
``models/cheb_cls.py`` (cheb_GCN) and ``crecon.py`` (estimate_diff) from /root/reference with
``oracle.refshim`` standing in for the absent third-party packages -- for this path that includes
torch-geometric 2.0.4's ChebConv, restated there from its published algorithm -- and writes

  tests/golden/cls_tiny.npz   tiny hierarchy: seed-666 state_dict, logits, CE loss, every gradient
  tests/golden/cls_5k.npz     crecon.cfg's classifier on the 5k template (B = 4), same contents,
                              plus crecon.estimate_diff of the seed-666 cheb_VAE in "train" and
                              "test" mode on the model_5k.npz input

    python oracle/make_golden_cls.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import make_golden as mg  # noqa: E402  (installs refshim, puts the reference on sys.path)

from models.cheb_cls import cheb_GCN  # noqa: E402  (reference)
from models.cheb_VAE import cheb_VAE  # noqa: E402  (reference)
import crecon  # noqa: E402  (reference)

OUT = mg.OUT


def classifier_vectors(topo, config, B, out):
    A, D, U, nn_ = mg.sparse_lists(topo)
    torch.manual_seed(666)
    cfg = dict(config, num_conv_filters=list(config["num_conv_filters"]))   # cheb_cls.py:60-61 mutates the list
    net = cheb_GCN(6, cfg, D, U, A, nn_)
    sd = net.state_dict()
    out["sd_keys"] = np.asarray(list(sd.keys()))
    for k, v in sd.items():
        out[f"sd/{k}"] = v.numpy().copy()
    x = torch.randn(B, nn_[0], 6, generator=torch.Generator().manual_seed(7))
    label = torch.arange(B) % 2
    out["x"], out["label"] = x.numpy(), label.numpy()
    net.train()
    logits = net(x)
    loss = torch.nn.CrossEntropyLoss()(logits, label)                         # crecon.py:83,262
    loss.backward()
    out["logits"], out["loss"] = logits.detach().numpy(), loss.detach().numpy()
    names = []
    for k, p in net.named_parameters():
        names.append(k)
        out[f"grad/{k}"] = p.grad.numpy().copy()
    out["grad_names"] = np.asarray(names)


def diff_vectors(topo, config, B, out):
    A, D, U, nn_ = mg.sparse_lists(topo)
    torch.manual_seed(666)
    dvae = cheb_VAE(3, dict(config, num_conv_filters=list(config["num_conv_filters"])), D, U, A, nn_,
                    model="optimal_sigma_VAE")
    dvae.eval()           # crecon.py leaves the VAE in train mode (dropout on); the fixture needs determinism
    crecon.device = torch.device("cpu")
    x = torch.randn(B, nn_[0], 3, generator=torch.Generator().manual_seed(0))  # == model_5k.npz "x"
    label = torch.tensor([1, 1, 0, 0][:B])
    for mode in ("train", "test"):
        diff, correct = crecon.estimate_diff(dvae, x, label, mode)
        out[f"diff/{mode}"], out[f"diff/{mode}_correct"] = diff.numpy(), np.int64(correct)
    out["diff/x"], out["diff/label"] = x.numpy(), label.numpy()


def main():
    torch.set_num_threads(8)
    tiny = dict(np.load(os.path.join(OUT, "topology_tiny.npz")))
    t = {}
    classifier_vectors(tiny, {"n_layers": 2, "num_conv_filters": [8, 16, 16], "polygon_order": [6, 6, 6],
                              "num_classes": 2}, 4, t)
    np.savez_compressed(os.path.join(OUT, "cls_tiny.npz"), **t)
    topo5k = dict(np.load(os.path.join(OUT, "topology_5k.npz")))
    cfg = {"n_layers": 4, "num_conv_filters": [16, 16, 16, 32, 32], "polygon_order": [6, 6, 6, 6, 6],
           "num_classes": 2, "num_style": 16, "num_hidden": 512, "dropout": 0.2}
    c = {}
    classifier_vectors(topo5k, cfg, 4, c)
    diff_vectors(topo5k, cfg, 4, c)
    print("5k logits", c["logits"], "loss", c["loss"], "correct", c["diff/train_correct"], c["diff/test_correct"])
    np.savez_compressed(os.path.join(OUT, "cls_5k.npz"), **c)
    for fn in ("cls_tiny.npz", "cls_5k.npz"):
        print(fn, os.path.getsize(os.path.join(OUT, fn)) // 1024, "KiB")


if __name__ == "__main__":
    main()
