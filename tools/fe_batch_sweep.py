"""Front-end launch times against the batch size: do the two passes scale with the work, or step with the number of
rounds of resident blocks (256 CUs x 4 blocks)?  python tools/fe_batch_sweep.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'speech-cloner_amd')):
    sys.path.insert(0, p)
import torch, bench, audio_lib
for B in [int(x) for x in sys.argv[1:]] or (8, 16, 20, 24, 32, 40, 48, 64, 96, 128):
    wav = bench.synth_audio(B, 64000, 0).cuda()
    out = audio_lib.calc_MFCC_input_batch(wav, None, **bench.FE_KW)
    for _ in range(3):
        audio_lib.calc_MFCC_input_batch(wav, None, out=out, **bench.FE_KW)
    t = {}
    for name, mask in (('stats', 2), ('feature', 4)):
        f = lambda: audio_lib.calc_MFCC_input_batch(wav, None, out=out, stage_mask=mask, **bench.FE_KW)
        t[name] = bench.time_events(f, 50) * 1e3
    nb1, nb2 = 51 * B, 58 * B
    print('B %3d  stats %6.1f us (%5d blocks = %.2f rounds)  feature %6.1f us (%5d blocks = %.2f rounds)   us per utterance %.2f' % (
        B, t['stats'], nb1, nb1 / 1024, t['feature'], nb2, nb2 / 1024, (t['stats'] + t['feature']) / B))
