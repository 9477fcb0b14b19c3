run() { MESHVAE_DEBUG="$1" python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-kernel-roofline --no-variants 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$1]', round(d['ms_per_step']*1e3,1), 'us')"; }
for s in "" "skip_conv_dw=1" "skip_conv_dw=2" "skip_conv_dw=3" "skip_conv_dw=4" "no_side=1" ""; do run "$s"; done
