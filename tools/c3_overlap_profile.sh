# A/B of the C3 step with G's weight gradients (a) before the critic phase on the caller's stream, (b) on the second lane with the
# join deferred to G's optimiser step, (c) the same, the dense blocks' weight gradients gated behind the data-gradient chain:
# wall time of each, and the per-queue timeline of the last step out of a rocprofv3 kernel trace.  usage: bash tools/c3_overlap_profile.sh r04
set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-r04}
O=$R/gpurun_out/prof_${TAG}_overlap
mkdir -p $O
S=$R/tools/rocpd_summary.py
db() { find $1 -name "*_results.db" | head -1; }
C3="python3 $R/bench.py --mode train --dtype bf16 --disc unet --disc-dtype bf16 --lq 128 --batch 32 --steps 4 --warmup 2"
OUT=$R/gpurun_out/${TAG}_c3_overlap.txt
: > $OUT
for cfg in "0 1" "1 1" "1 2"; do
  set -- $cfg
  export SR_BENCH_OVERLAP_G=$1 SR_DEFER_MODE=$2
  echo "== overlap_g_wgrad=$1 defer_mode=$2 (untraced run, then traced)" >> $OUT
  $C3 2> $O/run_$1_$2.err | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('ms_per_step', j['ms_per_step'], 'l_g_pix', j['losses']['l_g_pix'], 'l_d_real', j['losses']['l_d_real'])" >> $OUT
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace -d $O/kt_$1_$2 -- $C3 > $O/kt_$1_$2.log 2>&1)
  python3 $S timeline $(db $O/kt_$1_$2) $O/tl_$1_$2.txt >> $OUT
done
cat $OUT
find $R/gpurun_out -name "*_results.db" -size +8M -delete
