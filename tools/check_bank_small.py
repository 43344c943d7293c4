"""bank256 vs conv_kernel on small / ragged row counts (bit-exact expected); mismatch anatomy."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'speech-cloner_amd')):
    sys.path.insert(0, p)
import torch, modules
import _vc
st = modules.VariableStore('bfloat16')
torch.manual_seed(0)
for (W, T, Cin) in ((9, 400, 256), (3, 100, 128), (5, 333, 64), (16, 400, 256)):
    with modules.variable_store(st), modules.variable_scope('d%d_%d' % (T, Cin)):
        pre = (torch.randn(W, T, Cin, device='cuda') * 0.5).to(st.dtype)
        _vc.set_option('bank256', 0)
        ref = modules.conv1d_banks(pre, K=32, is_training=False).float().view(W * T, -1)
        _vc.set_option('bank256', -1)
        outs = [modules.conv1d_banks(pre, K=32, is_training=False).float().view(W * T, -1) for _ in range(6)]
    torch.cuda.synchronize()
    for k, out in enumerate(outs):
        bad = torch.nonzero((out - ref).abs() > 0).cpu().numpy()
        if len(bad) == 0:
            print('W=%d T=%d Cin=%d run %d: exact' % (W, T, Cin, k)); continue
        tiles = collections.Counter((int(r) // 256, int(c) // 256) for r, c in bad)
        print('W=%d T=%d Cin=%d run %d: %d mismatches in (row tile, pair): %s' % (W, T, Cin, k, len(bad), dict(tiles)))
        for (rt, pr) in list(tiles)[:2]:
            sel = bad[(bad[:, 0] // 256 == rt) & (bad[:, 1] // 256 == pr)]
            rows = sorted(set(int(r) % 256 for r in sel[:, 0])); cols = sorted(set(int(c) % 256 for c in sel[:, 1]))
            print('   tile (%d,%d): rows %s  cols %s' % (rt, pr, rows[:40], cols[:40]))
