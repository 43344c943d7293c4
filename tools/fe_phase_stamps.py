"""Where a front-end block spends its cycles: reads the s_memtime stamps a -DVC_ABLATE build of fe400_kernel writes at every
phase boundary (wave 0 of each block).  VC_LIB_PATH=build/libvc_hip_ablate.so python tools/fe_phase_stamps.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'speech-cloner_amd')):
    sys.path.insert(0, p)
import numpy as np, torch, bench, audio_lib, _vc
assert _vc.lib().vc_ablate_build(), 'needs the -DVC_ABLATE library (tools/build_ablate.sh, VC_LIB_PATH)'
B, L = 32, 64000
wav = bench.synth_audio(B, L, 0).cuda()
kw = dict(bench.FE_KW)
plan = audio_lib._get_plan(kw['sr'], kw['pre_emphasis'], kw['hop_length'], kw['win_length'], kw['n_mels'], kw['n_mfcc'], kw['n_fft'],
                           kw['window'], kw['mfcc_normaleze_first_mfcc'], kw['mfcc_norm_factor'], kw['calc_mfcc_derivate'],
                           kw['M_dB_norm_factor'], kw['P_dB_norm_factor'], kw['mean_abs_amp_norm'], kw['clip_output'])
ws = plan.workspace(B, L, wav.device)
mf = 1 + L // 80
nt1 = (mf + 15) // 16
a256 = lambda v: (v + 255) // 256 * 256
o_stats = a256(B * 16 * 4)
o_mel = a256(o_stats + B * nt1 * 8 * 4)
out = None
names1 = ['entry -> own samples in LDS (wave 0)', '25-point stage, twiddle, row stores', '16-point stage, |.|^2', 'power rows + weights',
          'barrier (power tile complete)', 'mel + reduce + record']
names2 = ['entry -> own samples in LDS (wave 0, incl. the utterance constants)', '25-point stage', '16-point stage', 'power rows',
          'barrier (power tile complete)', 'P_dB out', 'mel dB + barrier', 'M_dB out + sum/diff + barrier', 'DCT + barrier',
          'MFCC / delta out']
names3 = ['entry -> own samples in LDS', '25-point stage', '16-point stage, power rows', 'barrier (power tile complete)',
          'mel power + reductions + barrier', 'publish + WAIT for the utterance', 'gather + constants + barrier', 'P_dB out', 'mel dB + barrier',
          'M_dB out + sum/diff + barrier', 'DCT + barrier', 'MFCC / delta out']
passes = ((6, (mf + 13) // 14, names3),) if (len(sys.argv) > 1 and sys.argv[1] == 'fused') else ((2, nt1, names1), (4, (mf + 13) // 14, names2))
for mask, nblk, names in passes:
    for _ in range(3):
        if mask == 6:
            with _vc.options(fe_fused=1):
                out = audio_lib.calc_MFCC_input_batch(wav, None, out=out, **kw)
        else:
            with _vc.options(fe_fused=0):
                out = audio_lib.calc_MFCC_input_batch(wav, None, out=out, stage_mask=mask, **kw)
    torch.cuda.synchronize()
    raw = ws.view(torch.uint8)[o_mel + 65536 * 4: o_mel + 65536 * 4 + B * nblk * 16 * 8].view(torch.int64).cpu().numpy().reshape(B * nblk, 16)
    n = len(names) + 1
    t = raw[:, :n].astype(np.float64)
    d = np.diff(t, axis=1)
    print('pass mask %d: %d blocks, block lifetime median %.0f cycles (s_memtime ticks); launch span %.0f' % (
        mask, B * nblk, np.median(t[:, n - 1] - t[:, 0]), t[:, n - 1].max() - t[:, 0].min()))
    for i, nm in enumerate(names):
        print('   %-44s median %7.0f   p90 %7.0f' % (nm, np.median(d[:, i]), np.percentile(d[:, i], 90)))
    start = t[:, 0] - t[:, 0].min()
    print('   block start times: median %.0f, p90 %.0f, max %.0f' % (np.median(start), np.percentile(start, 90), start.max()))
