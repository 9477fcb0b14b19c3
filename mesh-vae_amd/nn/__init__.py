"""Drop-in for the reference's `nn` package (nn/conv.py, nn/pool.py) on MI355X."""
