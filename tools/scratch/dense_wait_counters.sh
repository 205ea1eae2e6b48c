cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in ${VARIANTS:-p0 nostore p4}; do
  for CNT in "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "WRITE_SIZE"; do
    rm -rf gpurun_out/wc_$v; LTRACE_LIB=$PWD/ab_builds/lib$v.so rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d gpurun_out/wc_$v -- python3 tools/dense_bench.py 4194304 224 0.9 1 > gpurun_out/wc_$v.json 2>/dev/null
    python3 - gpurun_out/wc_$v $v <<'PY'
import csv,glob,sys,os
from collections import defaultdict
a=defaultdict(float);c=defaultdict(int)
for f in glob.glob(os.path.join(sys.argv[1],'**','*counter_collection.csv'),recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_dense_tracks' in r['Kernel_Name']: a[r['Counter_Name']]+=float(r['Counter_Value']); c[r['Counter_Name']]+=1
print(sys.argv[2], {k: round(v/c[k]/1e6,1) for k,v in sorted(a.items())})
PY
  done
  LTRACE_LIB=$PWD/ab_builds/lib$v.so python3 tools/dense_bench.py 4194304 224 0.9 1 2>/dev/null | cut -c95-150
done
