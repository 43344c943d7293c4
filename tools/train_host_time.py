"""Decoder training step at the bench's shape: how long the host takes to enqueue a step (until it blocks on the loss
read-back) against the step's wall time."""
import os, sys, time, contextlib, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'speech-cloner_amd')):
    sys.path.insert(0, p)
import torch
with contextlib.redirect_stdout(io.StringIO()):
    from aux_func import load_cfg_d
    from encoder import encoder_spec_phn
    from decoder import decoder_specs
    hp = os.path.join(ROOT, 'speech-cloner_amd', 'hp')
    ec = load_cfg_d(os.path.join(hp, 'encoder_cfg_d.json')); dc = load_cfg_d(os.path.join(hp, 'decoder_cfg_d.json'))
    ec.update(is_training=False, model_path=os.path.join(ROOT, 'tests', 'golden', 'enc_14_ckpt')); dc.update(is_training=True)
    enc = encoder_spec_phn(ec, None); dec = decoder_specs(dc, None, enc)
g = torch.Generator().manual_seed(100)
mfcc = (torch.rand(32, 400, 80, generator=g) * 0.4 - 0.2).cuda()
mel = (torch.rand(32, 400, 80, generator=g) * 0.8).cuda()
stft = (torch.rand(32, 400, 201, generator=g) * 0.8).cuda()
for _ in range(2):
    dec.exec_train_step(mfcc, mel, stft)
tr = dec._get_trainer()
x = dec._to_device(mfcc, dec._input_width(), 'x'); tm = dec._to_device(mel, 80, 'm'); ts = dec._to_device(stft, 201, 's')
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    losses = tr.forward_backward(x, tm, ts)
    tr.apply_gradients(1)
    t1 = time.perf_counter()
    losses.cpu()
    t2 = time.perf_counter()
    print('host enqueue %.1f ms, step wall %.1f ms' % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
