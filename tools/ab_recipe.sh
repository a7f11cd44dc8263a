# A/B of the recipe step (32x32 LR patches, batch 32, VGG discriminator) on one box: backward weight gradients on the caller's stream
# (--overlap 0) against the side lane (automatic).  usage (GPU box): bash tools/ab_recipe.sh
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "fp32:--dtype fp32" "bf16:--dtype bf16 --disc-dtype bf16"; do
  k=${v%%:*}; fl=${v#*:}
  for ov in 0 -1; do
    python3 $R/bench.py --mode train --lq 32 --batch 32 --steps 20 --warmup 3 $fl --overlap $ov 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('$k overlap $ov:', d['ms_per_step'], 'ms', d['value'], 'img/s')"
  done
done
done
python3 $R/bench.py --mode train --dtype bf16 --disc unet --lq 128 --batch 32 --steps 3 --warmup 1 --overlap 1 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('c3 overlap 1:', d['ms_per_step'])"
python3 $R/bench.py --mode train --dtype bf16 --disc unet --lq 128 --batch 32 --steps 3 --warmup 1 --overlap 0 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('c3 overlap 0:', d['ms_per_step'])"
