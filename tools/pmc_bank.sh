#!/bin/bash
# PMC passes over the filter-bank kernel alone (tools/prof_kernels.py bank): usage bash tools/pmc_bank.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_bank
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"; do
  N=$(echo $C | tr " " "_" | cut -c1-20)
  timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $O/$N -- python3 $R/tools/prof_kernels.py bank > $O/$N.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob('$O/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'bank256' in r['Kernel_Name'] or 'conv_kernel' in r['Kernel_Name']:
            agg[(r['Kernel_Name'][:40], r['Counter_Name'])].append(float(r['Counter_Value']))
for k in sorted(agg): print(k, sum(agg[k]) / len(agg[k]))
PY
