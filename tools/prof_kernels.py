#!/usr/bin/env python
"""Runs the dominant kernels of the hot path a few times each, stand-alone, so rocprofv3 (kernel
trace or --pmc passes) sees clean dispatches:  python tools/prof_kernels.py [bank|proj1|gru|encfront|frontend|vocoder|train16|all]

Shapes = the bench's full workload (64 windows x 400 frames, decoder step 2, bf16; front-end on
32 x 4 s).  Prints the algorithmic bytes / FLOPs per launch used by bench.py's roofline."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'speech-cloner_amd')):
    sys.path.insert(0, p)
import torch          # noqa: E402
import modules        # noqa: E402
import audio_lib      # noqa: E402
import bench          # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else 'all'
W, T, reps = 64, 400, (60 if what == 'bank' else 3)      # >= 50 launches of the roofline kernel for its rocprof average
st = modules.VariableStore('bfloat16')
torch.manual_seed(0)
with modules.variable_store(st), modules.variable_scope('decoder'), modules.variable_scope('step2'), modules.variable_scope('CBHG'):
    pre = torch.randn(W, T, 256, device='cuda').to(st.dtype)
    # (the two launches exactly as modules.CBHG issues them: the bank stores its max-pooled result, the projection reads a
    # plain operand and -- same kernel name, bank256_kernel -- runs only when asked for by name, so that the 'all' passes
    # keep the filter bank's counters clean)
    if what in ('bank', 'all', 'proj1'):
        for _ in range(reps):
            bank, pooled = modules.conv1d_banks(pre, K=32, is_training=False, pool_output='auto')
    if what == 'proj1':
        for _ in range(20):
            modules.conv1d(bank, filters=256, size=3, scope="conv1d_1", bn_scope="conv1d_1", activation_fn='relu',
                           pool_input=0 if pooled else 2)
    if what in ('gru', 'all'):
        for _ in range(reps):
            modules.gru(pre, num_units=256, bidirection=True)
if what in ('encfront', 'all'):
    with modules.variable_store(st), modules.variable_scope('encoder'):
        xe = torch.rand(W, T, 80, device='cuda') * 0.4 - 0.2
        for _ in range(reps):
            modules._cbhg_front(xe, 80, 6, 1, 'prenet', 'CBHG')
if what in ('frontend', 'all'):
    wav = bench.synth_audio(32, 64000, 0).cuda()
    out = None
    import _vc
    for _ in range(reps):                       # the form the library picks at this batch (two launches)
        out = audio_lib.calc_MFCC_input_batch(wav, None, out=out, **bench.FE_KW)
    with _vc.options(fe_fused=1):               # and the one-launch form, for its counters
        for _ in range(reps):
            out = audio_lib.calc_MFCC_input_batch(wav, None, out=out, **bench.FE_KW)
if what in ('vocoder', 'all'):
    amp = torch.rand(16, 1000, 201, device='cuda') * 0.1
    ph = torch.rand(16, 1000, 201, device='cuda') * 3.14159
    audio_lib.griffin_lim_batch(amp, None, 400, 80, num_iters=4, phase0=ph)
if what in ('train16', 'all'):
    # the training step's dominant launch: the step-2 filter bank forward at 32 x 400 frames on gemm16_kernel
    import gemm16
    Mt, Ht, Kt = 32 * T, 256, 32
    gg = torch.Generator().manual_seed(3)
    kern = [(torch.randn(k, Ht, 128, generator=gg) * 0.05).cuda() for k in range(1, Kt + 1)]
    w16 = gemm16.Weights16(torch.device('cuda'))
    pairs16, cs16 = gemm16.bank_forward_operands(w16, kern, Ht)
    w16.refresh()
    xt = torch.randn(Mt, Ht, device='cuda')
    x16, rs16 = gemm16.split16(xt, Mt, Ht, Ht, T)
    zt = torch.empty((Mt, 128 * Kt), device='cuda')
    for _ in range(reps if what == 'all' else 60):
        gemm16.gemm16(x16, rs16, Mt, T, Ht, pairs16, zt, 128 * Kt, col_scale=cs16)
    print('train16: %.4g algorithmic FLOP/launch (x3 executed); operands: X16 %d B + W16 %d B, out %d B' % (
        2.0 * 256 * 128 * 528 * Mt, Mt * 512 * 2, 256 * 128 * 528 * 2 * 2, Mt * 4096 * 4))
torch.cuda.synchronize()
print('bank  : %.4g FLOP/launch ; operands: X %d B + W %d B, out %d B' % (
    2.0 * 256 * 128 * 528 * W * T, W * T * 256 * 2, 256 * 128 * 528 * 2, W * T * 4096 * 2))
print('proj1 : %.4g FLOP/launch ; in %d B (x3 taps x2 pool via L2), W %d B, out %d B' % (
    2.0 * 3 * 4096 * 256 * W * T, W * T * 4096 * 2, 3 * 4096 * 256 * 2, W * T * 256 * 2))
print('encfront: %.4g FLOP/launch ; in %d B, out %d B' % (2.0 * 226880 * W * T, W * T * 80 * 4, W * T * 240 * 4))
print('front : %d B/launch algorithmic (1,764 B/frame x 25,632 frames)' % (1764 * 25632))
