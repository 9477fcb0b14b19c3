#!/bin/bash
# usage: tools/ab_lib.sh <libA.so> <libB.so> [rounds] [extra bench args] -- alternating bench runs of two BUILDS of the library
A="$1"; B="$2"; R="${3:-2}"; shift 3 2>/dev/null
run() { MESHVAE_LIB="$PWD/mesh-vae_amd/meshvae_hip/$1" python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-kernel-roofline --no-variants "${@:3}" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2 [$1]', round(d['value']), 'meshes/s', round(d['ms_per_step']*1e3,1), 'us')"; }
for i in $(seq $R); do run "$A" A "$@"; run "$B" B "$@"; done
