"""Localise a gemm16 mismatch: error per bank and per frame position (debug aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'speech-cloner_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch, numpy as np
import gemm16, test_gemm16_gpu as tg
H, K, N, T = 128, 8, 3, 400
M = N * T
x = tg._rand_acts(N, T, H, 10 + K)
g = torch.Generator().manual_seed(77)
Ws = [(torch.randn(k, H, 128, generator=g) * (0.3 / (k * H) ** 0.5) * (1 + k % 5)).float() for k in range(1, K + 1)]
ref = torch.cat([tg._conv_same64(x.double(), W.double()) for W in Ws], dim=2).view(M, 128 * K)
dev = torch.device('cuda')
Wd = [W.to(dev).contiguous() for W in Ws]
w16 = gemm16.Weights16(dev)
pairs, cs = gemm16.bank_forward_operands(w16, Wd, H)
w16.refresh()
xd = x.to(dev).view(M, H)
x16, rs = gemm16.split16(xd, M, H, H, T)
out = torch.full((M, 128 * K), float('nan'), device=dev)
gemm16.gemm16(x16, rs, M, T, H, pairs, out, 128 * K, col_scale=cs)
e = (out.double().cpu() - ref).abs() / ref.abs().max()
print('nan count', int(torch.isnan(out).sum()))
for k in range(1, K + 1):
    ek = e[:, 128 * (k - 1):128 * k]
    rows = torch.nonzero(ek.max(dim=1).values > 1e-5).flatten().tolist()
    print('bank %d: max err %.2e, bad rows %d: %s' % (k, float(ek.max()), len(rows), rows[:24]))
# weights check: reconstruct bank 8's operand
k = 8
bt = pairs[3][1].double().cpu()          # [128][k*2H]
sc = cs[128 * (k - 1)].item()
rec = (bt.view(128, k, 2, H)[:, :, 0] + bt.view(128, k, 2, H)[:, :, 1]) * sc     # [o][j][c]
wref = Ws[k - 1].double().permute(2, 0, 1)
print('weight operand bank 8: max rel err %.2e' % float((rec - wref).abs().max() / wref.abs().max()))
