"""Per-pair timing of the paired filter-bank kernel (option ablate_bank256_only, -DVC_ABLATE build): separates the per-tile
rate from the per-block fixed cost.  100 blocks on 256 CUs: every block has a CU to itself."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'speech-cloner_amd')):
    sys.path.insert(0, p)
import torch, modules, bench
import _vc
st = modules.VariableStore('bfloat16')
W, T, Cin = 64, 400, 256
with modules.variable_store(st), modules.variable_scope('d'):
    pre = (torch.randn(W, T, Cin, device='cuda') * 0.5).to(st.dtype)
    res = []
    for p in (0, 1, 3, 7, 11, 15):
        _vc.set_option('ablate_bank256_only', p)
        ms = bench.time_events(lambda: modules.conv1d_banks(pre, K=32, is_training=False), 20)
        tiles = 4 * (2 * p + 2)
        res.append((p, tiles, ms))
        print('pair %2d  tiles %3d  %.4f ms  -> %.3f us/tile incl. fixed' % (p, tiles, ms, ms * 1e3 / tiles))
    (p0, t0, m0), (p1, t1, m1) = res[0], res[-1]
    per_tile = (m1 - m0) / (t1 - t0) * 1e3
    print('per tile %.3f us (ideal 2048 cycles = %.3f us at 2.1 GHz); fixed per block %.2f us' % (per_tile, 2048 / 2100.0, m0 * 1e3 - per_tile * t0))
    _vc.set_option('ablate_bank256_only', -1)
