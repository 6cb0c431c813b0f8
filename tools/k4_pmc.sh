# SQ counters of the K4b region kernel on one level shape (default: C4 mip 1 = n_src 128, out 2048, roughness 0.03)
#   bash tools/k4_pmc.sh [n_src out rough [tag]]    -> gpurun_out/k4pmc_<tag>.txt
set -e
export TMPDIR=/tmp
R=$PWD
NS=${1:-128}; OUT=${2:-2048}; RO=${3:-0.03}; TAG=${4:-mip1}
cd /tmp
rm -rf /tmp/k4a /tmp/k4b
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/k4a -- python3 $R/tools/mc_probe.py $NS $OUT $RO > $R/gpurun_out/k4pmc_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/k4b -- python3 $R/tools/mc_probe.py $NS $OUT $RO > $R/gpurun_out/k4pmc_b.log 2>&1
python3 - <<PY > $R/gpurun_out/k4pmc_$TAG.txt
import csv, glob, collections
agg=collections.defaultdict(dict)
for d in 'ab':
    f=glob.glob(f'/tmp/k4{d}/**/*_counter_collection.csv', recursive=True)
    per=collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f[0])):
        if 'k_mc_region' in r['Kernel_Name']:
            per[r['Dispatch_Id']][r['Counter_Name']] += float(r['Counter_Value'])
    # last dispatch of the probe
    last=per[sorted(per, key=int)[-1]]
    agg.update(last)
w=agg.get('SQ_WAVES',1)
for k in sorted(agg): print(f"{k:28s} {agg[k]:18.0f}  per wave {agg[k]/w:12.1f}")
PY
cat $R/gpurun_out/k4pmc_$TAG.txt
