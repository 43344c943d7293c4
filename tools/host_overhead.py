"""How long does the host need to ENQUEUE one full step (no synchronisation) vs. the GPU to run it?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'speech-cloner_amd')):
    sys.path.insert(0, p)
import torch, bench, audio_lib
enc, dec = bench.load_models('bfloat16', 0)
wav = bench.synth_audio(32, 64000, 0).cuda()
fe = audio_lib.calc_MFCC_input_batch(wav, None, **bench.FE_KW)
x = fe[0][:, :800, :].reshape(64, 400, 80).contiguous()
for _ in range(3):
    dec.forward(x)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    audio_lib.calc_MFCC_input_batch(wav, None, out=fe, **bench.FE_KW)
    dec.forward(x)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('host enqueue %.3f ms/step, total %.3f ms/step' % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    dec.forward(x)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
