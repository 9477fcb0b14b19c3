"""Drop-in for the reference's `models` package (cheb_VAE only; cheb_cls is out of scope)."""
