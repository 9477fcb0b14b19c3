#!/bin/bash
# usage: tools/ab3.sh rounds "<debug A>" "<debug B>" ... -- alternating bench runs of any number of MESHVAE_DEBUG settings
R="$1"; shift
run() { MESHVAE_DEBUG="$1" timeout -k 10 300 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-kernel-roofline --no-variants | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$1]', round(d['value']), 'meshes/s', round(d['ms_per_step']*1e3,1), 'us')"; }
for i in $(seq $R); do for cfg in "$@"; do run "$cfg" || exit 1; done; done
