# where does the k-step kernel's time go: VALU pipeline busy cycles per instruction, calibrated on the issue probe
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
rm -rf $O/pmc6_*
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU2 SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_LEVEL_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc6_probe_$i -- tools/exp/issue/issue_probe > $O/pmc6_probe_$i.log 2>&1; echo "probe $i rc=$?"
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc6_fused_$i -- python3 tools/kernel_sweep.py --fused --rounds 1 --launches 64 > $O/pmc6_fused_$i.log 2>&1; echo "fused $i rc=$?"
done
python3 - <<'PY'
import csv,glob,os,collections,re
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc6_*/**/*_counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        k=re.sub(r'\(anonymous namespace\)::','',r['Kernel_Name']).split('(')[0].replace('void ','')
        if k.startswith('probe') or 'rollout_kernel<false, false, false' in k or k.startswith('fpv_drone_step_kernel<false, false, false, false'):
            agg[k][r['Counter_Name']].append((float(r['Counter_Value']), int(r['End_Timestamp'])-int(r['Start_Timestamp']), int(r['Grid_Size']) if 'Grid_Size' in r else 0))
names=["SQ_INSTS_VALU","SQ_ACTIVE_INST_VALU","SQ_ACTIVE_INST_VALU2","SQ_VALU_MFMA_BUSY_CYCLES","SQ_THREAD_CYCLES_VALU","SQ_BUSY_CYCLES","SQ_WAVE_CYCLES","SQ_WAIT_INST_ANY","SQ_WAIT_ANY","SQ_ACTIVE_INST_ANY","SQ_INST_CYCLES_SALU","SQ_IFETCH","SQ_IFETCH_LEVEL","SQ_LEVEL_WAVES","GRBM_GUI_ACTIVE","SQ_WAVES"]
print("kernel | dispatch(us, largest) | per-VALU-instruction: " + " ".join(n.replace("SQ_","") for n in names[1:]))
for k,c in sorted(agg.items()):
    # the largest dispatch of the kernel (probe: the 8-waves-per-SIMD launch with 2000 iterations)
    def big(name):
        v=c.get(name)
        if not v: return None
        return max(v, key=lambda t: t[1])
    iv=big("SQ_INSTS_VALU")
    if not iv: continue
    row=[f"{k[:58]:58s}", f"{iv[1]/1e3:9.1f}"]
    for n in names[1:]:
        b=big(n)
        row.append("     -" if not b else f"{b[0]/iv[0]:7.3f}")
    print(" ".join(row))
PY
