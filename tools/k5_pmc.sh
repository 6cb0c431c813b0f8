set -e
export TMPDIR=/tmp PROBE_WORKERS=1
R=$PWD
cd /tmp
P="python3 $R/tools/shade_probe.py 7680 4320"
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d /tmp/k5a -- $P > $R/gpurun_out/k5pmc_a.log 2>&1
echo A
rocprofv3 --kernel-trace --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d /tmp/k5b -- $P > $R/gpurun_out/k5pmc_b.log 2>&1
echo B
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/k5c -- $P > $R/gpurun_out/k5pmc_c.log 2>&1
echo C
rocprofv3 --kernel-trace --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_BUFFER_TOTAL_CYCLES_sum --output-format csv -d /tmp/k5d -- $P > $R/gpurun_out/k5pmc_d.log 2>&1
echo D
python3 - <<'PY'
import csv, glob, collections
for d in 'abcd':
    f=glob.glob(f'/tmp/k5{d}/**/*_counter_collection.csv', recursive=True)
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if 'shade' in r['Kernel_Name']:
            agg[(r['Kernel_Name'].split('(')[0], r['Grid_Size'], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k,v in sorted(agg.items()):
        print(k, len(v), max(v))
PY
