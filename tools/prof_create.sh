cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in base new; do
  if [ $v = base ]; then export FCPP_LIBRARY=$R/build/libfcpp_base.so; else unset FCPP_LIBRARY; fi
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pc_$v -o t --output-format csv -- python3 $R/tools/trace_create.py 100 > $R/gpurun_out/pc_$v.log 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob('$R/gpurun_out/pc_$v/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    print('$v', r['Name'][:60], r['Calls'], r['AverageNs'])
PY
  tail -2 $R/gpurun_out/pc_$v.log
done
