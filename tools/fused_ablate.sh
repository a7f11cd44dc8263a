# Development aid: variant libraries of the fused dense-block kernel with parts switched off (SR_FZ_ABL bits, fused_block.inc) —
# timing only, the results of an ablated kernel are wrong.  usage: bash tools/fused_ablate.sh "0 1 2 4 8 16 31"; then on the GPU box
# python tools/fused_ablate.py
set -e
cd "$(dirname "$0")/.."
for v in ${1:-0 1 2 4 8 16 31}; do
  ( bash tools/ab_bf16.sh abl$v "-DSR_FZ_ABL=$v" && echo "built abl$v" ) &
  while [ $(jobs -r | wc -l) -ge 4 ]; do sleep 1; done
done
wait
