# Runs ON THE GPU BOX: instruction counters of the setup kernels of the headline plan call (tools/trace_create.py, 20 calls).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES -d $R/gpurun_out/pmc_create -o p --output-format csv -- python3 $R/tools/trace_create.py 20 > $R/gpurun_out/pmc_create.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob('$R/gpurun_out/pmc_create/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in acc.items():
    if 'fcpp' not in k: continue
    w = sum(c['SQ_WAVES']) / len(c['SQ_WAVES'])
    print(k[:60], 'waves %d' % w, ' '.join('%s/w %.0f' % (n[3:], sum(v) / len(v) / w) for n, v in sorted(c.items()) if n != 'SQ_WAVES'))
PY
