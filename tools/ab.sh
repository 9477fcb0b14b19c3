#!/bin/bash
# usage: tools/ab.sh "<env for A>" "<env for B>" [rounds]  -- prints meshes/s of alternating bench runs
A="$1"; B="$2"; R="${3:-2}"
run() { env $1 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-kernel-roofline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', round(d['value']), 'meshes/s', round(d['ms_per_step']*1e3,1), 'us')"; }
for i in $(seq $R); do run "$A" A; run "$B" B; done
