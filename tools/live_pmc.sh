# lane utilisation of the live shader: SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU) per dispatch of tools/live_probe.py
set -e
export TMPDIR=/tmp
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d /tmp/livepmc -- python3 $R/tools/live_probe.py > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
f=glob.glob('/tmp/livepmc/**/*_counter_collection.csv', recursive=True)[0]
per=collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    if 'k_shade' in r['Kernel_Name']:
        per[(r['Dispatch_Id'], r['Kernel_Name'].split('(')[0])][r['Counter_Name']]=float(r['Counter_Value'])
seen=set()
for (d,k),c in sorted(per.items(), key=lambda x:int(x[0][0])):
    key=(k, round(c.get('SQ_INSTS_VALU',0)/1e6))
    if key in seen: continue
    seen.add(key)
    util=c['SQ_THREAD_CYCLES_VALU']/(64*c['SQ_ACTIVE_INST_VALU']) if c.get('SQ_ACTIVE_INST_VALU') else 0
    print(f"dispatch {d} {k}: VALU instr {c.get('SQ_INSTS_VALU',0)/1e6:.1f}M, per wave {c.get('SQ_INSTS_VALU',0)/max(c.get('SQ_WAVES',1),1):.0f}, lane utilisation {util:.3f}")
PY
