"""Front-end launch times against the batch size: do the two passes scale with the work, or step with the number of
rounds of resident blocks (256 CUs x 4 blocks)?  python tools/fe_batch_sweep.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'speech-cloner_amd')):
    sys.path.insert(0, p)
import statistics
import torch, bench, audio_lib, _vc
_vc.set_option('fe_fused', 0)        # the two launches, each timed alone
# Five repeats of 20 launches each, median and max: a single mean over 50 launches (the round-2 form of this script)
# once read 230 us at 12 utterances and 170 us at 28 between smooth neighbours -- one stalled launch in fifty (the first
# launches after a 64 MB allocation + host upload) is enough for that; the medians below are what the kernels take.
for B in [int(x) for x in sys.argv[1:]] or (8, 12, 16, 20, 24, 28, 32, 40, 48, 64, 96, 128):
    wav = bench.synth_audio(B, 64000, 0).cuda()
    out = audio_lib.calc_MFCC_input_batch(wav, None, **bench.FE_KW)
    for _ in range(3):
        audio_lib.calc_MFCC_input_batch(wav, None, out=out, **bench.FE_KW)
    t = {}
    for name, mask in (('stats', 2), ('feature', 4)):
        f = lambda: audio_lib.calc_MFCC_input_batch(wav, None, out=out, stage_mask=mask, **bench.FE_KW)
        reps = [bench.time_events(f, 20) * 1e3 for _ in range(5)]
        t[name] = statistics.median(reps)
        t[name + '_max'] = max(reps)
    nb1, nb2 = 51 * B, 58 * B
    print('B %3d  stats %6.1f us (max of 5 repeats %6.1f; %5d blocks = %.2f rounds)  feature %6.1f us (max %6.1f; %5d blocks = %.2f rounds)   us per utterance %.2f' % (
        B, t['stats'], t['stats_max'], nb1, nb1 / 1024, t['feature'], t['feature_max'], nb2, nb2 / 1024, (t['stats'] + t['feature']) / B))
