"""Measurement hook driver: python tools/time_bank_dbg.py <dbg> [only]  (options ablate_bank256 / ablate_bank256_only of the -DVC_ABLATE build)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'speech-cloner_amd')):
    sys.path.insert(0, p)
import torch, modules, bench
import _vc
assert _vc.lib().vc_ablate_build(), 'needs the -DVC_ABLATE library: VC_LIB_PATH=build/libvc_hip_ablate.so (tools/build_ablate.sh)'
_vc.set_option('ablate_bank256', int(sys.argv[1]))
if len(sys.argv) > 2:
    _vc.set_option('ablate_bank256_only', int(sys.argv[2]))
st = modules.VariableStore('bfloat16')
W, T, Cin = 64, 400, 256
with modules.variable_store(st), modules.variable_scope('d'):
    pre = (torch.randn(W, T, Cin, device='cuda') * 0.5).to(st.dtype)
    ms = bench.time_events(lambda: modules.conv1d_banks(pre, K=32, is_training=False), 20)
print('dbg=%s only=%s  %.4f ms (event timing, host-bound below ~0.08 ms)' % (sys.argv[1], sys.argv[2:] or None, ms))
