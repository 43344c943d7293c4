"""Interleaved A/B (one process, one box) of bank256_kernel variants that exist only in the -DVC_ABLATE build (same results):
bit 16 = static priority for waves 4-7 instead of per-cluster s_setprio flips, bit 32 = coefficient load behind the first
tiles' requests.  VC_LIB_PATH=build/libvc_hip_ablate.so python tools/ab_bank_variants.py [rounds]"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'speech-cloner_amd')):
    sys.path.insert(0, p)
import torch, modules, bench, _vc
assert _vc.lib().vc_ablate_build()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
st = modules.VariableStore('bfloat16')
W, T = 64, 400
x = (torch.randn(W, T, 256, device='cuda') * 0.5).to(st.dtype)
def f():
    with modules.variable_store(st), modules.variable_scope('b'):
        return modules.conv1d_banks(x, K=32, is_training=False, pool_output='auto')
f()
for v in list(st.vars.values()):
    if v.dim() == 3:
        v.copy_(torch.randn(v.shape, device='cuda') * 0.05)
st.invalidate()
ref = f()[0].clone()
fl = 2.0 * 256 * 128 * 528 * W * T
res = {m: [] for m in (0, 16, 32, 48)}
for r in range(rounds):
    for m in res:
        _vc.set_option('ablate_bank256', m)
        assert torch.equal(f()[0], ref)
        res[m].append(bench.time_events(f, 20))
base = statistics.median(res[0])
for m, v in res.items():
    print('variant %2d: median %.4f ms (min %.4f) %.0f TF  %+.2f %%' % (m, statistics.median(v), min(v), fl / statistics.median(v) / 1e9,
                                                                       100 * (base / statistics.median(v) - 1)))
