#!/bin/bash
# Collects the round's evidence on the GPU box into gpurun_out/$RND (default r03; copied to profiles/$RND afterwards).
# usage: bash tools/collect_profiles.sh     (from the repo root, on the MI355X box)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
RND=${RND:-r03}
O=$R/gpurun_out/$RND
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke exit $?" >> $O/smoke.log
timeout -k 10 400 python bench.py > $O/bench_full.log 2>&1
timeout -k 10 200 python bench.py --workload frontend --steps 50 --warmup 5 > $O/bench_frontend.log 2>&1
timeout -k 10 300 python bench.py --workload train --steps 10 --warmup 3 > $O/bench_train.log 2>&1
timeout -k 10 200 python bench.py --workload vocoder --steps 5 --warmup 1 > $O/bench_vocoder.log 2>&1
for f in bench_full bench_frontend bench_train bench_vocoder; do tail -1 $O/$f.log > $O/$f.json; done
cd /tmp && export TMPDIR=/tmp
for W in full frontend train vocoder; do
  EXTRA="--steps 5 --warmup 1 --no-cpu-baseline --no-f32 --no-side"; [ $W = train ] && EXTRA="--steps 2 --warmup 1 --no-cpu-baseline"; [ $W = vocoder ] && EXTRA="--steps 2 --warmup 1 --no-cpu-baseline"
  [ $W = frontend ] && EXTRA="--steps 50 --warmup 5 --no-cpu-baseline"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$W -- python3 $R/bench.py --workload $W $EXTRA > $O/trace_$W.log 2>&1
  cp $O/trace_$W/*/*kernel_stats.csv $O/${W}_kernel_stats.csv 2>/dev/null
done
python3 $R/tools/train_timeline.py $O/trace_train/*/*kernel_trace.csv > $O/train_timeline.log 2>&1
# the roofline kernel alone, >= 50 launches (bench.py's rocprof average mixes the step-1 and step-2 filter banks under one name)
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_bank -- python3 $R/tools/prof_kernels.py bank > $O/trace_bank.log 2>&1
cp $O/trace_bank/*/*kernel_stats.csv $O/bank_step2_kernel_stats.csv 2>/dev/null
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr " " "_" | cut -c1-20)
  timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $O/pmc_$N -- python3 $R/tools/prof_kernels.py all > $O/pmc_$N.log 2>&1
done
python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
names = {'bank256_kernel': 'bank256_kernel_bf16_step2', 'conv256_kernel': 'conv256_kernel_bf16', 'gru_mfma_kernelILi256': 'gru_mfma_256',
         'gru_mfma_kernel<256>': 'gru_mfma_256', 'fe400_kernel<true>': 'fe400_stats_pass', 'fe400_kernel<false>': 'fe400_feature_pass', 'fe400_fused_kernel': 'fe400_one_launch',
         'gl_iter400_kernel<false>': 'gl_iter400_kernel', 'cbhg_small_kernel': 'cbhg_small_kernel',
         'gemm16_kernel': 'gemm16_bank_step2_train'}
for f in glob.glob('$O/pmc_*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        for k, v in names.items():
            if k in r['Kernel_Name']:
                agg[v][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for k, v in agg.items():
    d = {c: sum(x) / len(x) for c, x in v.items()}
    if 'FETCH_SIZE' in d and 'WRITE_SIZE' in d:
        # MI355X_MICROARCH.md: FETCH_SIZE/WRITE_SIZE are KiB; FETCH_SIZE reads 1/2 of a wide coalesced stream on gfx950
        d['traffic_bytes_per_launch'] = int((2 * d['FETCH_SIZE'] + d['WRITE_SIZE']) * 1024)
    out[k] = d
json.dump(out, open('$O/pmc_summary.json', 'w'), indent=1, sort_keys=True)
print(json.dumps({k: v.get('traffic_bytes_per_launch') for k, v in out.items()}))
PY
cd $R
timeout -k 10 200 python tools/fe_batch_sweep.py > $O/frontend_batch_sweep.log 2>&1
# training convolutions on split-float16 operands: per-launch A/B against the f32-MFMA kernels, K-split forms, host enqueue time
timeout -k 10 300 python tools/ab_gemm16.py > $O/ab_gemm16.log 2>&1
timeout -k 10 300 python tools/ab_gemm16_split.py > $O/ab_gemm16_split.log 2>&1
timeout -k 10 200 python tools/train_cpu_profile.py > $O/train_cpu_profile.log 2>&1
rm -rf $O/trace_* $O/pmc_*/                  # keep the summaries (csv / json / logs), not the raw traces
tail -2 $O/smoke.log; for f in bench_full bench_frontend bench_train bench_vocoder; do cut -c1-400 $O/$f.json; done
