"""float32 training recurrences, forward and backward: weights partly resident in registers (option gru_train_resident = 1) against
the form that streams them from L2 every step: largest differences of outputs / saved gates, event timings."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'speech-cloner_amd')):
    sys.path.insert(0, p)
import torch, _vc, bench
lib = _vc.lib()
P = lambda t: C.c_void_p(t.data_ptr())
torch.manual_seed(0)
if '--floor' in sys.argv:          # -DVC_ABLATE build: the streaming kernels without their weight stream (timing only)
    assert lib.vc_ablate_build()
    fn = lib.vc_ablate_set_gru_train; fn.restype = C.c_int; fn.argtypes = [C.c_int32]
    assert fn(1) == 0
for H in (256, 128):
    N, T = 32, 400
    xproj = torch.randn(N * T, 6 * H, device='cuda') * 0.5
    wh = [torch.randn(H, 3 * H, device='cuda') / H ** 0.5 for _ in range(2)]
    res = {}
    for mode in (0, 1):
        _vc.set_option('gru_train_resident', mode)
        out = torch.empty(N * T, 2 * H, device='cuda'); gates = torch.empty(2, N * T, 3 * H, device='cuda'); rh = torch.empty(2, N * T, H, device='cuda')
        f = lambda: _vc.check(lib.vc_gru_train_forward(P(xproj), P(wh[0]), P(wh[1]), N, T, H, P(out), P(gates), P(rh), _vc.current_stream()))
        f()
        ms = min(bench.time_events(f, 3) for _ in range(2))
        res[mode] = (out.clone(), gates.clone(), rh.clone(), ms)
    d = [float((a - b).abs().max()) for a, b in zip(res[0][:3], res[1][:3])]
    print('H %d forward : streaming %.3f ms, resident %.3f ms; max |diff| out %.2e gates %.2e r*h %.2e' % (H, res[0][3], res[1][3], *d))
    out, gates = res[0][0], res[0][1]
    dout = torch.randn(N * T, 2 * H, device='cuda') * 0.1
    whT = [w.t().contiguous() for w in wh]
    rb = {}
    for mode in (0, 1):
        _vc.set_option('gru_train_resident', mode)
        dpre = torch.empty(N * T, 6 * H, device='cuda')
        f = lambda: _vc.check(lib.vc_gru_backward(P(dout), P(out), P(gates), P(wh[0]), P(wh[1]), P(whT[0]), P(whT[1]), N, T, H, P(dpre), _vc.current_stream()))
        f()
        ms = min(bench.time_events(f, 3) for _ in range(2))
        rb[mode] = (dpre.clone(), ms)
    print('H %d backward: streaming %.3f ms, resident %.3f ms; max |diff| dpre %.2e (max |dpre| %.2e)' % (
        H, rb[0][1], rb[1][1], float((rb[0][0] - rb[1][0]).abs().max()), float(rb[0][0].abs().max())))
