"""bank256_kernel, decoder step-2 filter bank at 64 windows: the three block -> XCD maps (option bank256_xcd: 0 plain 2-D
grid, 1 whole filter-width pairs per XCD -- the tallest row-tile columns, a pair's weight stream in ONE L2 --, 2 pairs
split over two XCDs: the default), HIP events over 50 launches each, interleaved three times.
python tools/ab_bank_xcd.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'speech-cloner_amd'), os.path.join(ROOT, 'tools')):
    sys.path.insert(0, p)
import torch, modules, _vc
from ab_gemm16 import timed

W, T = 64, 400
st = modules.VariableStore('bfloat16')
torch.manual_seed(0)
pre = torch.randn(W, T, 256, device='cuda').to(st.dtype)
flop = 2.0 * 256 * 128 * 528 * W * T
with modules.variable_store(st), modules.variable_scope('decoder'), modules.variable_scope('step2'), modules.variable_scope('CBHG'):
    run = lambda: modules.conv1d_banks(pre, K=32, is_training=False, pool_output='auto')
    run()
    for rep in range(3):
        line = []
        for mode in (0, 1, 2):
            with _vc.options(bank256_xcd=mode):
                ms = timed(run, 50)
            line.append('map %d: %.4f ms = %6.1f TFLOP/s (%.1f %%)' % (mode, ms, flop / ms / 1e9, flop / ms / 1e9 / 25.0))
        print('   '.join(line))
