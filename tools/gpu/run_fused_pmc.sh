# where the k-step kernel's time goes: VALU-busy / wait / wave cycles (rocprofv3 PMC passes, counters only)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
rm -rf $O/pmc_fused_*
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM" "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32" "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64 SQ_IFETCH" "SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc_fused_$tag -- python3 tools/kernel_sweep.py --geom 1x128 --fused --rounds 1 --launches 64 > $O/pmc_fused_$tag.log 2>&1
  echo "$tag rc=$?"
done
python3 - <<'PY'
import csv,glob,os,collections,re
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc_fused_*/**/*_counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        k=re.sub(r'\(anonymous namespace\)::','',r['Kernel_Name']).split('(')[0].replace('void ','')
        if ('rollout_kernel<128, false, false, false, false' in k) or ('fpv_drone_step_kernel<128, 1, false, false, false, false, false' in k):
            agg[k][r['Counter_Name']].append((float(r['Counter_Value']), int(r['End_Timestamp'])-int(r['Start_Timestamp'])))
for k,c in agg.items():
    print(k)
    for name,v in sorted(c.items()):
        v.sort(); val,ns=v[len(v)//2]
        print(f"   {name:28s} median {val:16.0f}   (dispatch {ns/1e3:8.1f} us, {len(v)} dispatches)")
PY
