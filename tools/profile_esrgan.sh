# rocprofv3 kernel statistics of whole training steps (bench.py --mode train); args: extra bench flags
set -e
R=$GRAFT_REPO_ROOT
TAG=${TAG:-esrgan}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r01b_$TAG -- python3 $R/bench.py --mode train "$@" > $R/gpurun_out/r01b_$TAG.log 2>&1
tail -1 $R/gpurun_out/r01b_$TAG.log | cut -c1-300
find $R/gpurun_out -name "*kernel_trace.csv" -delete
