"""Interleaved A/B (one process, one box) of the bf16 MFMA recurrence: four waves with all weights in registers (gru_mfma4 = 1)
against the eight-wave form (default), H = 256 and 128, 64 windows x 400 steps; events over 10 launches, ABAB
rounds; then bench.py's single-stream figure both ways.   python tools/ab_gru_mfma4.py [rounds]"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'speech-cloner_amd')):
    sys.path.insert(0, p)
import torch, modules, bench, _vc
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
W, T = 64, 400
for H in (256, 128):
    st = modules.VariableStore('bfloat16')
    x = (torch.randn(W, T, H, device='cuda') * 0.5).to(st.dtype)
    def f():
        with modules.variable_store(st), modules.variable_scope('g'):
            return modules.gru(x, num_units=H, bidirection=True)
    f()
    res = {'four': [], 'eight': []}
    for r in range(rounds):
        for name, v in (('four', 1), ('eight', -1)):
            with _vc.options(gru_mfma4=v):
                res[name].append(bench.time_events(f, 10))
    for name, v in res.items():
        m = statistics.median(v)
        print('H=%d %-5s median %.4f ms (min %.4f) = %.3f us per step' % (H, name, m, min(v), m * 1e3 / T))
wav = bench.synth_audio(32, 64000, seed=0).cuda()
enc, dec = bench.load_models('bfloat16', 0)
import audio_lib
fe = audio_lib.calc_MFCC_input_batch(wav, None, out_frames=800, **bench.FE_KW)
xw = fe[0].view(64, 400, 80)
res = {'four': [], 'eight': []}
for r in range(rounds):
    for name, v in (('four', 1), ('eight', -1)):
        with _vc.options(gru_mfma4=v):
            res[name].append(bench.time_events(lambda: dec.forward(xw), 5))
for name, v in res.items():
    m = statistics.median(v)
    print('encode+decode of 64 windows, one stream, %-5s: median %.4f ms (min %.4f)' % (name, m, min(v)))
