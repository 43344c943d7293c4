#!/bin/bash
# per-setting kernel time of cbhg_small_kernel: tools/cbhg_front_dbg.sh "2:0 2:1 4:0 ..."  (MI:DBG)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for s in $1; do
  export CFD_MI=${s%%:*} CFD_DBG=${s##*:} VC_LIB_PATH=$GRAFT_REPO_ROOT/build/libvc_hip_ablate.so   # tools/build_ablate.sh
  rm -rf gpurun_out/cfd
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cfd -- python3 tools/time_cbhg_front_dbg.py > gpurun_out/cfd.log 2>&1 || exit 1
  python3 - "$s" <<PY
import csv, glob, sys
f = glob.glob("gpurun_out/cfd/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "cbhg_small" in r["Name"]: print("MI:DBG", sys.argv[1], "calls", r["Calls"], "avg us %.1f" % (float(r["AverageNs"]) / 1e3))
PY
done
