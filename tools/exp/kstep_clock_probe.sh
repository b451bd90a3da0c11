# does the k-step kernel run at a lower clock when it streams its actions from HBM?  SQ_BUSY_CYCLES and GRBM_GUI_ACTIVE per
# dispatch over the dispatch duration, streamed vs held actions (tools/exp/kstep_held_vs_stream.py launches both)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
rm -rf $O/pmc_clock
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/pmc_clock -- python3 tools/exp/kstep_held_vs_stream.py > $O/pmc_clock.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = max(glob.glob('gpurun_out/pmc_clock/**/*_counter_collection.csv', recursive=True))
rows = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    if 'rollout_kernel' in r['Kernel_Name']:
        rows[int(r['Dispatch_Id'])][r['Counter_Name']] = float(r['Counter_Value'])
        rows[int(r['Dispatch_Id'])]['ns'] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
ids = sorted(rows)
# the script alternates: 120 streamed launches, 120 held, 120 streamed, 120 held
for name, sl in (('streamed', ids[0:120]), ('held', ids[120:240]), ('streamed', ids[240:360]), ('held', ids[360:480])):
    if not sl: continue
    n = len(sl)
    ns = sum(rows[i]['ns'] for i in sl) / n
    busy = sum(rows[i]['SQ_BUSY_CYCLES'] for i in sl) / n
    gui = sum(rows[i]['GRBM_GUI_ACTIVE'] for i in sl) / n
    wave = sum(rows[i]['SQ_WAVE_CYCLES'] for i in sl) / n
    wait = sum(rows[i]['SQ_WAIT_ANY'] for i in sl) / n
    print(f"{name:9s}: {ns / 1e3:7.1f} us per launch   SQ_BUSY_CYCLES / ns = {busy / ns:6.2f}   GRBM_GUI_ACTIVE / ns = {gui / ns:6.2f}   SQ_WAIT_ANY / SQ_WAVE_CYCLES = {wait / wave:5.3f}")
PY
