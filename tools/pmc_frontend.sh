#!/bin/bash
# PMC passes over the front-end kernels (bench.py --workload frontend): instruction counts, issue / wait split, LDS, HBM bytes.
# usage (GPU box, repo root): bash tools/pmc_frontend.sh OUTDIR
O=${1:-gpurun_out/pmc_fe}; R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/$O
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
         "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $R/$O/p$i -- python3 $R/bench.py --workload frontend --steps 10 --warmup 2 --no-cpu-baseline > $R/$O/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$R/$O/p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'].split('(')[0][-40:]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items():
    if 'fe' not in k: continue
    print(k)
    for c, x in sorted(v.items()):
        print('   %-26s %14.1f  (n=%d)' % (c, sum(x) / len(x), len(x)))
PY
