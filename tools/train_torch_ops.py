"""Which lines of the training step still launch torch kernels: counts of copy_/clone/contiguous/zeros/cat/flip/t calls
per source line over one decoder training step (after two warm-up steps)."""
import os, sys, contextlib, io, collections, traceback
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
counts = collections.Counter()
def where():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if 'speech-cloner_amd' in fr.filename:
            return '%s:%d' % (os.path.basename(fr.filename), fr.lineno)
    return '?'
def wrap(obj, name):
    orig = getattr(obj, name)
    def f(*a, **k):
        counts[(name, where())] += 1
        return orig(*a, **k)
    setattr(obj, name, f)
for n in ('copy_', 'clone', 'contiguous', 'flip', 'fill_', 'zero_', '__setitem__', 'float', 'to'):
    wrap(torch.Tensor, n)
for n in ('zeros', 'cat', 'zeros_like', 'flip', 'stack'):
    wrap(torch, n)
dec.exec_train_step(mfcc, mel, stft)
for (n, w), c in sorted(counts.items(), key=lambda kv: -kv[1])[:45]:
    print('%5d  %-12s %s' % (c, n, w))
