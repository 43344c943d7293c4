"""Upper bound of what the per-step weight re-layouts (torch glue behind store.invalidate()) cost the training step: the same
step with the kernel-layout caches kept (stale weights in the copies: WRONG numerics, timing only).
python tools/train_glue_bound.py"""
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
def run(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        dec.exec_train_step(mfcc, mel, stft)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for _ in range(3):
    dec.exec_train_step(mfcc, mel, stft)
a = [run(5) for _ in range(3)]
inv = dec.store.invalidate
dec.store.invalidate = lambda: None
b = [run(5) for _ in range(3)]
dec.store.invalidate = inv
c = [run(5) for _ in range(3)]
print('shipped           ms/step', ['%.2f' % v for v in a])
print('caches kept stale ms/step', ['%.2f' % v for v in b], '(wrong numerics: upper bound of the gain)')
print('shipped again     ms/step', ['%.2f' % v for v in c])
