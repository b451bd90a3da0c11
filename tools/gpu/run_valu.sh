# instruction counts of the step and k-step kernels (rocprofv3 PMC pass, counters only)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmc_valu -- python3 tools/kernel_sweep.py --geom 1x128 --fp16 --fused --rounds 1 --launches 64 > $O/pmc_valu.log 2>&1
echo rc=$?
python3 - <<'PY'
import csv,glob,os,collections,re
f=max(glob.glob('gpurun_out/pmc_valu/**/*_counter_collection.csv',recursive=True),key=os.path.getmtime)
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k=re.sub(r'\(anonymous namespace\)::','',r['Kernel_Name']).split('(')[0].replace('void ','')
    if 'fpv_' in k: agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,c in agg.items():
    w=max(c['SQ_WAVES']) if c.get('SQ_WAVES') else 0
    v=max(c['SQ_INSTS_VALU']) if c.get('SQ_INSTS_VALU') else 0
    s=max(c['SQ_INSTS_SALU']) if c.get('SQ_INSTS_SALU') else 0
    print(f"{k[:70]:70s} waves {w:9.0f} valu/wave {v/max(w,1):9.1f} salu/wave {s/max(w,1):8.1f}  (max over dispatches)")
PY
