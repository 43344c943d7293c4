#!/bin/bash
# Builds build/libvc_hip_ablate.so: the library with the timing-only ablation paths compiled in (-DVC_ABLATE).
# The shipped speech-cloner_amd/libvc_hip.so never contains them.  Use: VC_LIB_PATH=build/libvc_hip_ablate.so python tools/time_bank_dbg.py 1
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/build/ablate
cd $R/speech-cloner_amd/csrc
for f in *.hip; do
  /opt/rocm/bin/hipcc -O3 -fno-slp-vectorize -std=c++17 -fPIC --offload-arch=gfx950 -DVC_ABLATE -I$R/include -I. -c $f -o $R/build/ablate/${f%.hip}.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $R/build/ablate/*.o -o $R/build/libvc_hip_ablate.so
echo built $R/build/libvc_hip_ablate.so
