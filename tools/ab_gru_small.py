"""Interleaved A/B (one process, one box) of the encoder's H = 40 bf16 recurrence: 16 sequences per wave on MFMA (gru_small_mfma = 1)
against one wave per sequence (default): the recurrence alone (64 windows x 400 steps), the
encoder forward, and the pipelined step (bench._Pipeline, 10 streams).   python tools/ab_gru_small.py [rounds]"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'speech-cloner_amd')):
    sys.path.insert(0, p)
import torch, modules, bench, _vc
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
W, T, H = 64, 400, 40
st = modules.VariableStore('bfloat16')
x = (torch.randn(W, T, H, device='cuda') * 0.5).to(st.dtype)
def f():
    with modules.variable_store(st), modules.variable_scope('g'):
        return modules.gru(x, num_units=H, bidirection=True)
f()
var = (('mfma', 1), ('wave', -1))
res = {k: [] for k, _ in var}
for r in range(rounds):
    for name, v in var:
        with _vc.options(gru_small_mfma=v):
            res[name].append(bench.time_events(f, 10))
for name, v in res.items():
    m = statistics.median(v)
    print('recurrence alone (incl. input projection) %-5s median %.4f ms (min %.4f) = %.3f us per step' % (name, m, min(v), m * 1e3 / T))
wav = bench.synth_audio(32, 64000, seed=0).cuda()
enc, dec = bench.load_models('bfloat16', 0)
import audio_lib
fe = audio_lib.calc_MFCC_input_batch(wav, None, out_frames=800, **bench.FE_KW)
xw = fe[0].view(64, 400, 80)
res = {k: [] for k, _ in var}
res2 = {k: [] for k, _ in var}
for r in range(rounds):
    for name, v in var:
        with _vc.options(gru_small_mfma=v):
            res[name].append(bench.time_events(lambda: enc.forward(xw), 10))
            res2[name].append(bench.time_events(lambda: dec.forward(xw), 5))
for name in res:
    print('%-5s encoder forward of 64 windows: median %.4f ms; encode + decode, one stream: median %.4f ms' % (
        name, statistics.median(res[name]), statistics.median(res2[name])))
pipe = bench._Pipeline(wav, dec, 64, 10)
pipe.setup()
res = {k: [] for k, _ in var}
for r in range(rounds):
    for name, v in var:
        with _vc.options(gru_small_mfma=v):
            res[name].append(pipe.timed(40, 6, 1) / 40 * 1e3)
for name, v in res.items():
    print('pipelined step %-5s median %.4f ms (min %.4f)' % (name, statistics.median(v), min(v)))
