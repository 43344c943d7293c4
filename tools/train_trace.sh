#!/bin/bash
# Kernel trace of two decoder training steps: every launch of the last step in order with its duration.
# usage (GPU box, repo root): bash tools/train_trace.sh OUTDIR
O=$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/tt -- python3 $GRAFT_REPO_ROOT/bench.py --workload train --steps 2 --warmup 1 --no-cpu-baseline > $O/tt.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - $O <<'PY'
import csv, glob, sys
O = sys.argv[1]
rows = list(csv.DictReader(open(glob.glob(O + '/tt/*/*kernel_trace.csv')[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
a, b = adam[-2], adam[-1]
with open(O + '/train_step_launches.txt', 'w') as f:
    t0 = int(rows[a]['End_Timestamp'])
    for r in rows[a + 1:b + 1]:
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        f.write('%9.1f us at %9.1f  %s\n' % (d, (int(r['Start_Timestamp']) - t0) / 1e3, r['Kernel_Name'][:100]))
    f.write('step wall %.1f us, %d launches\n' % ((int(rows[b]['End_Timestamp']) - t0) / 1e3, b - a))
PY
rm -rf $O/tt
