#!/bin/bash
# hipGraph replay of the train step under the runtime's graph knobs (one bench run each; eager first and last for scale)
#   tools/graph_env_sweep.sh > gpurun_out/<tag>/graph_env_sweep.txt
run() { env "$@" python bench.py --steps 200 --warmup 30 --prewarm-steps 200 --no-cpu-baseline --no-kernel-roofline --no-variants $EXTRA 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', '$EXTRA', round(d['ms_per_step']*1e3,1), 'us')"; }
EXTRA="" run A=0
EXTRA="--graph" run A=0
EXTRA="--graph" run DEBUG_HIP_FORCE_GRAPH_QUEUES=1
EXTRA="--graph" run DEBUG_HIP_FORCE_GRAPH_QUEUES=2
EXTRA="--graph" run DEBUG_HIP_FORCE_GRAPH_QUEUES=4
EXTRA="--graph" run DEBUG_HIP_FORCE_GRAPH_QUEUES=8
EXTRA="--graph" run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
EXTRA="--graph" run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
EXTRA="--graph" run DEBUG_HIP_GRAPH_BATCH_SIZE=1
EXTRA="--graph" run DEBUG_HIP_GRAPH_BATCH_SIZE=16
EXTRA="--graph" run DEBUG_HIP_GRAPH_BATCH_SIZE=256
EXTRA="" run A=0
