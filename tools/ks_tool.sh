# rocprofv3 kernel statistics of one python tool.  usage (on the GPU box): bash tools/ks_tool.sh <tag> tools/<script>.py [args]
set -e
R=$GRAFT_REPO_ROOT
TAG=$1
shift
O=$R/gpurun_out/ks_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O -- python3 $R/"$@" > $O/run.log 2>&1
python3 $R/tools/rocpd_summary.py stats $(find $O -name "*_results.db" | head -1) $O/stats.csv > $O/stats.txt
head -${KS_LINES:-14} $O/stats.txt
find $O -name "*.db" -size +20M -delete
