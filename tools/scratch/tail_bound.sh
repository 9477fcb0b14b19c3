run() { env $2 MESHVAE_DEBUG="$1" python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-kernel-roofline --no-variants 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$1 $2]', round(d['ms_per_step']*1e3,1), 'us')"; }
for i in 1 2; do run "" "A=1"; run "skip_pack=1" "A=1"; run "" "MESHVAE_SKIP_ADAM=1"; run "skip_pack=1" "MESHVAE_SKIP_ADAM=1"; done
