"""A/B of the filter-bank launch: paired 256x256 kernel (vc_bank256.hip) vs conv_kernel (option bank256 = 0).
Prints the largest difference between the two outputs and both timings."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'speech-cloner_amd')):
    sys.path.insert(0, p)
import torch, modules, bench
import _vc
st = modules.VariableStore('bfloat16')
W, T = 64, 400
Cin = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.manual_seed(0)
with modules.variable_store(st), modules.variable_scope('d'):
    pre = (torch.randn(W, T, Cin, device='cuda') * 0.5).to(st.dtype)
    # non-trivial weights / BN so that a wrong tap or column shows
    modules.conv1d_banks(pre, K=32, is_training=False)
    g = torch.Generator(device='cuda').manual_seed(1)
    for n, v in st.vars.items():
        if n.endswith('kernel'):
            v.copy_(torch.randn(v.shape, device='cuda', generator=g) * 0.05)
        elif n.endswith('gamma') or n.endswith('moving_variance'):
            v.copy_(torch.rand(v.shape, device='cuda', generator=g) + 0.5)
        else:
            v.copy_(torch.randn(v.shape, device='cuda', generator=g) * 0.1)
    st.invalidate()
    _vc.set_option('bank256', 0)
    ref = modules.conv1d_banks(pre, K=32, is_training=False).float()
    ms0 = bench.time_events(lambda: modules.conv1d_banks(pre, K=32, is_training=False), 20)
    _vc.set_option('bank256', -1)
    out = modules.conv1d_banks(pre, K=32, is_training=False).float()
    ms1 = bench.time_events(lambda: modules.conv1d_banks(pre, K=32, is_training=False), 20)
torch.cuda.synchronize()
d = (out - ref).abs()
print('max |new - old| = %.4g (ref max %.3g); mismatching elements %d of %d' % (d.max().item(), ref.abs().max().item(), int((d > 0).sum()), d.numel()))
if d.max().item() > 0:
    idx = torch.nonzero(d > 1e-2)
    print('first large mismatches (window, frame, channel):', idx[:8].tolist())
fl = 2.0 * Cin * 128 * 528 * W * T
print('conv_kernel  %.4f ms  %.1f TFLOP/s' % (ms0, fl / ms0 / 1e9))
print('bank256      %.4f ms  %.1f TFLOP/s' % (ms1, fl / ms1 / 1e9))
for mode, what in (('1', 'whole pairs per XCD'), ('0', 'plain block order'), ('', 'pairs split over two XCDs (default)')):
    _vc.set_option('bank256_xcd', int(mode) if mode else -1)
    with modules.variable_store(st), modules.variable_scope('d'):
        out2 = modules.conv1d_banks(pre, K=32, is_training=False).float()
        ms2 = bench.time_events(lambda: modules.conv1d_banks(pre, K=32, is_training=False), 20)
    print('bank256 (%s) %.4f ms  %.1f TFLOP/s; equal to default: %s' % (what, ms2, fl / ms2 / 1e9, bool(torch.equal(out, out2))))
