"""vc_gemm16 single-pair launches at 32 x 400 frames: ways the K walk is split over workgroups and where the splits run
(option gemm16_split = ways + 16 * map).  python tools/ab_gemm16_split.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'speech-cloner_amd')):
    sys.path.insert(0, p)
import torch, gemm16, _vc
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from ab_gemm16 import timed

dev = torch.device('cuda')
N, T, H, K = 32, 400, 256, 32
M, CB = N * T, 128 * K
g = torch.Generator().manual_seed(1)
Ws = [(torch.randn(k, H, 128, generator=g) * 0.05).to(dev) for k in range(1, K + 1)]
W1 = (torch.randn(3, CB, H, generator=g) * 0.02).to(dev)
zb = torch.randn(M, CB, generator=g).to(dev)
w16 = gemm16.Weights16(dev)
pp, pcs = gemm16.conv_forward_operands(w16, W1)
bp, bcs = gemm16.bank_dgrad_operands(w16, Ws, H)
w16.refresh()
z16, zrs = gemm16.split16(zb, M, CB, CB, T)
ws = torch.empty(_vc.lib().vc_gemm16_workspace_bytes(M, CB, 1), dtype=torch.uint8, device=dev)
q1 = torch.empty((M, H), device=dev)
dd = torch.zeros((M, H), device=dev)
ref = {}
for ways, mp in ((1, 0), (2, 0), (4, 0), (5, 0), (8, 0), (2, 1), (4, 1), (8, 1), (-1, 0)):
    _vc.set_option('gemm16_split', -1 if ways < 0 else ways + 16 * mp)
    tp = timed(lambda: gemm16.gemm16(z16, zrs, M, T, CB, pp, q1, H, col_scale=pcs, workspace=ws))
    a = q1.clone()
    tb = timed(lambda: gemm16.gemm16(z16, zrs, M, T, CB, bp, dd, H, col_scale=bcs, ragged=True, workspace=ws))
    b = dd.clone()
    if not ref:
        ref = {'p': a, 'b': b}
    ep = float((a - ref['p']).abs().max() / ref['p'].abs().max())
    eb = float((b - ref['b']).abs().max() / ref['b'].abs().max())
    print('ways %2d map %d: projection forward %.3f ms   bank data gradient %.3f ms   (vs unsplit: %.1e / %.1e)' % (ways, mp, tp, tb, ep, eb))
_vc.set_option('gemm16_split', -1)
