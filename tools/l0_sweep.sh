#!/bin/bash
# base (K=1) vs per-order cost of the level-0 16->16 kernels, isolated
for k in 1 2 6; do
  python tools/microbench_conv.py --level 0 --cin 16 --cout 16 --k $k --iters 50
  python tools/microbench_conv.py --level 0 --cin 16 --cout 16 --k $k --iters 50 --bwd
  python tools/microbench_conv.py --level 0 --cin 16 --cout 16 --k $k --iters 50 --dwonly
done
for lvl in 1 2 3; do
  python tools/microbench_conv.py --level $lvl --cin 16 --cout 16 --k 6 --iters 50
  python tools/microbench_conv.py --level $lvl --cin 16 --cout 16 --k 1 --iters 50
  python tools/microbench_conv.py --level $lvl --cin 16 --cout 16 --k 6 --iters 50 --dwonly
  python tools/microbench_conv.py --level $lvl --cin 16 --cout 16 --k 1 --iters 50 --dwonly
done
