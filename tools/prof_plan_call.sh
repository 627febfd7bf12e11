# on the GPU box: a plan call end to end (wall clock, three calls and one call) and its kernel timeline.  usage: bash tools/prof_plan_call.sh <tag> [workload]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-x}
W=${2:-headline}
mkdir -p $R/gpurun_out/r05
FCPP_ONE_CALL=0 python3 $R/tools/trace_create.py 300 $W > $R/gpurun_out/r05/trace_${TAG}_$W.log 2>&1
python3 $R/tools/trace_create.py 300 $W >> $R/gpurun_out/r05/trace_${TAG}_$W.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r05/pc_${TAG}_$W -o t --output-format csv -- python3 $R/tools/trace_create.py 60 $W > $R/gpurun_out/r05/pc_${TAG}_$W.log 2>&1
F=$(find $R/gpurun_out/r05/pc_${TAG}_$W -name '*kernel_trace.csv' | head -1)
python3 $R/tools/call_timeline.py $F > $R/gpurun_out/r05/timeline_${TAG}_$W.txt 2>&1
grep median $R/gpurun_out/r05/trace_${TAG}_$W.log
cat $R/gpurun_out/r05/timeline_${TAG}_$W.txt
find $R/gpurun_out/r05/pc_${TAG}_$W -name '*.csv' ! -name '*kernel_stats.csv' -delete
