"""Determinism / agreement of the MFMA GRU (option gru_mfma = 1) vs the register-resident VALU kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'speech-cloner_amd')):
    sys.path.insert(0, p)
import torch, modules
import _vc
st = modules.VariableStore('bfloat16')
torch.manual_seed(0)
streams = [torch.cuda.Stream() for _ in range(3)]
for (N, T, H) in ((2, 400, 256), (1, 400, 256), (9, 400, 128), (33, 100, 256)):
    with modules.variable_store(st), modules.variable_scope('g%d' % H):
        x = (torch.randn(N, T, H, device='cuda') * 0.5).to(st.dtype)
        _vc.set_option('gru_mfma', 0)
        ref = modules.gru(x, num_units=H, bidirection=True).float()
        _vc.set_option('gru_mfma', 1)
        outs = [modules.gru(x, num_units=H, bidirection=True).float() for _ in range(3)]
        torch.cuda.synchronize()
        par = []
        for s in streams:
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                par.append(modules.gru(x, num_units=H, bidirection=True).float())
        torch.cuda.synchronize()
    print('N=%d T=%d H=%d: |mfma - valu| max %.4g; repeat equal %s; concurrent equal %s' % (
        N, T, H, (outs[0] - ref).abs().max().item(), all(torch.equal(o, outs[0]) for o in outs),
        [torch.equal(o, outs[0]) for o in par]))
