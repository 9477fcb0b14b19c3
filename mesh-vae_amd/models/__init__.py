"""Drop-in for the reference's `models` package: cheb_VAE (models/cheb_VAE.py, the hot path) and the
crecon classifier cheb_GCN (models/cheb_cls.py), both on the HIP kernels."""
