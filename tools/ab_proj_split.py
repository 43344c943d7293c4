"""Interleaved A/B (one process, one box) of decoder stage 2's first k = 3 projection (25,600 x 12,288 -> 256, bf16) on the
bank tiles: K split over two workgroups per row tile (default) against one workgroup per row tile (proj256_split = 0)
and against conv256_kernel (proj256 = 0); events over 50 launches, ABAB rounds.  Then the same switch inside the
pipelined step (bench._Pipeline, 10 streams).   python tools/ab_proj_split.py [rounds]"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'speech-cloner_amd')):
    sys.path.insert(0, p)
import torch, modules, bench, _vc
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
W, T = 64, 400
st = modules.VariableStore('bfloat16')
x = (torch.randn(W, T, 4096, device='cuda').abs() * 0.3).to(st.dtype)
def f():
    with modules.variable_store(st), modules.variable_scope('p'):
        return modules.conv1d(x, filters=256, size=3, scope='c', bn_scope='c', activation_fn='relu', pool_input=0)
f()
fl = 2.0 * 3 * 4096 * 256 * W * T
variants = {'split2': {}, 'unsplit': {'proj256_split': 0}, 'conv256': {'proj256': 0}}
res = {k: [] for k in variants}
for r in range(rounds):
    for name, opts in variants.items():
        with _vc.options(**opts):
            res[name].append(bench.time_events(f, 50))
for name, v in res.items():
    m = statistics.median(v)
    print('alone  %-8s median %.4f ms (min %.4f)  %.0f TFLOP/s' % (name, m, min(v), fl / m / 1e9))
# inside the pipelined step
class A: pass
wav = bench.synth_audio(32, 64000, seed=0).cuda()
enc, dec = bench.load_models('bfloat16', 0)
pipe = bench._Pipeline(wav, dec, 64, 10)
pipe.setup()
res = {k: [] for k in ('split2', 'unsplit')}
for r in range(rounds):
    for name in res:
        with _vc.options(**variants[name]):
            dt = pipe.timed(40, 6, 1)
        res[name].append(dt / 40 * 1e3)
for name, v in res.items():
    print('pipelined step %-8s median %.4f ms (min %.4f)' % (name, statistics.median(v), min(v)))
