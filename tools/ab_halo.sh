# A/B on one box: the product library against a variant built without the on-chip tile insides (bash tools/ab_bf16.sh nohalo "-DSR_FZ_HALO=0"):
# the reference recipe's step (8-row fused instance forward, 64-tile transposed blocks), the C3 step, bf16 inference.
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
  for lib in "" $R/image_restoration_amd/lib/libsr_hip_nohalo.so; do
    export SR_HIP_LIB_PATH=$lib
    [ -z "$lib" ] && unset SR_HIP_LIB_PATH
    tag=${lib:+nohalo}; tag=${tag:-product}
    python3 $R/bench.py --mode train --lq 32 --batch 32 --steps 20 --warmup 3 --dtype bf16 --disc-dtype bf16 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('$tag recipe bf16:', d['ms_per_step'], 'ms')"
    python3 $R/bench.py --dtype bf16 --no-cpu-baseline --no-secondary --steps 20 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag inference bf16:', d['value'], 'img/s')"
  done
done
for lib in "" $R/image_restoration_amd/lib/libsr_hip_nohalo.so; do
  export SR_HIP_LIB_PATH=$lib
  [ -z "$lib" ] && unset SR_HIP_LIB_PATH
  tag=${lib:+nohalo}; tag=${tag:-product}
  python3 $R/bench.py --mode train --dtype bf16 --disc unet --lq 128 --batch 32 --steps 3 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('$tag c3:', d['ms_per_step'], 'ms')"
done
