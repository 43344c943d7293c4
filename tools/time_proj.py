"""Times the k = 3 projection (decoder step-2 shape: 4096 -> 256) on conv256 vs conv_kernel, with and
without the fused max-pool of the operand."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'speech-cloner_amd')):
    sys.path.insert(0, p)
import torch, modules, bench
import _vc
st = modules.VariableStore('bfloat16')
W, T = 64, 400
for (cin, f) in ((4096, 256), (4096, 128), (512, 256)):
    with modules.variable_store(st), modules.variable_scope('p%d_%d' % (cin, f)):
        x = torch.rand(W, T, cin, device='cuda').to(st.dtype)
        for pool in (2, 0):
            for env in ('0', '1'):
                _vc.set_option('conv256', int(env) if env == '0' else -1)
                ms = bench.time_events(lambda: modules.conv1d(x, filters=f, size=3, scope='c', bn_scope='c', activation_fn='relu', pool_input=pool), 10)
                print('cin %4d f %3d pool %d conv256=%s: %.4f ms  %.0f TFLOP/s' % (cin, f, pool, env, ms, 2.0 * 3 * cin * f * W * T / ms / 1e9))
