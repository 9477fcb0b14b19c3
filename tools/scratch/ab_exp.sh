# A/B of the regular build against the EXP build (libmeshvae_hip_exp.so): isolated patch kernels + the fp32 step, alternating
EXPLIB=$GRAFT_REPO_ROOT/mesh-vae_amd/meshvae_hip/libmeshvae_hip_exp.so
for i in 1 2 3; do
  for lib in "" "$EXPLIB"; do
    tag=$([ -z "$lib" ] && echo new || echo exp)
    f=$(MESHVAE_LIB=$lib timeout -k 10 120 python tools/microbench_conv.py --iters 50 2>&1 | tail -1 | sed 's/.*fwd: //')
    b=$(MESHVAE_LIB=$lib timeout -k 10 120 python tools/microbench_conv.py --iters 50 --bwd 2>&1 | tail -1 | sed 's/.*bwd: //')
    s=$(MESHVAE_LIB=$lib timeout -k 10 200 python bench.py --steps 300 --warmup 20 --prewarm-steps 100 --no-cpu-baseline --no-kernel-roofline --no-variants 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1))")
    echo "[$tag] fwd $f | bwd $b | step $s us"
  done
done
