"""Timing probe of the one-launch front-end (events over 50 launches): batch sizes, stored rows, against the two-pass form.
python tools/fe_fused_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'speech-cloner_amd')):
    sys.path.insert(0, p)
import torch, bench, _vc, audio_lib
for B in (4, 8, 16, 17, 24, 32, 64):
    wav = bench.synth_audio(B, 64000, seed=0).cuda()
    for R in (801, 14):
        out = audio_lib.calc_MFCC_input_batch(wav, None, out_frames=R, **bench.FE_KW)
        f = lambda: audio_lib.calc_MFCC_input_batch(wav, None, out=out, out_frames=R, **bench.FE_KW)
        with _vc.options(fe_fused=1):
            t1 = bench.time_events(f, 50)
        with _vc.options(fe_fused=0):
            t2 = bench.time_events(f, 50)
        print('B=%2d rows=%3d  one launch %.4f ms   two launches %.4f ms' % (B, R, t1, t2), flush=True)
