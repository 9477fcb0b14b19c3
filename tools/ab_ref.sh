#!/bin/bash
# usage: tools/ab_ref.sh [rounds] -- the reference-API loop (bench.py --only-reference-loop) with and without the asynchronous launcher
R="${1:-2}"
run() { env "$@" python bench.py --only-reference-loop --steps 300 --warmup 30 --prewarm-steps 300 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', round(d['ms_per_step']*1e3,1), 'us/step', round(d['value']), 'meshes/s async', d['async_launcher'])"; }
for i in $(seq $R); do run MESHVAE_ASYNC=0; run MESHVAE_ASYNC=1; done
