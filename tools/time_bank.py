import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'speech-cloner_amd')):
    sys.path.insert(0, p)
import torch, modules, bench
st = modules.VariableStore('bfloat16')
W, T = 64, 400
with modules.variable_store(st), modules.variable_scope('d'):
    pre = torch.randn(W, T, 256, device='cuda').to(st.dtype)
    ms = bench.time_events(lambda: modules.conv1d_banks(pre, K=32, is_training=False), 20)
print(os.environ.get('VC_LIB_PATH', 'default'), 'bank ms %.4f  TF %.1f' % (ms, 2.0 * 256 * 128 * 528 * W * T / ms / 1e9))
