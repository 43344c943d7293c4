"""A/B of the encoder's fused pre-recurrence launch (vc_cbhg_front) against the per-layer launches
(modules.OPTIONS["cbhg_front"] = False): largest differences of the GRU output / logits / posteriors, and both timings."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'speech-cloner_amd')):
    sys.path.insert(0, p)
import numpy as np, torch, bench, modules
from encoder import encoder_spec_phn
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'encoder_fwd.npz'))
cfg = json.load(open(os.path.join(ROOT, 'speech-cloner_amd', 'hp', 'encoder_cfg_d.json')))
cfg.update(is_training=False, model_path=os.path.join(ROOT, 'tests', 'golden', 'enc_14_ckpt'), compute_dtype='bfloat16')
enc = encoder_spec_phn(cfg, None); enc.restore()
x3 = torch.from_numpy(g['x']).cuda()
x = torch.cat([x3] * 22, 0)[:64].contiguous()
outs = {}
for mode in ('0', '1'):
    modules.OPTIONS['cbhg_front'] = mode == '1'
    o = enc.forward(x)
    torch.cuda.synchronize()
    outs[mode] = {k: v.float().cpu() for k, v in o.items() if k in ('CBHG_out', 'y_logits', 'y_pred')}
    ms = bench.time_events(lambda: enc.forward(x), 20)
    print('cbhg_front=%s  encoder forward %.3f ms' % (mode, ms))
for k in outs['0']:
    d = (outs['0'][k] - outs['1'][k]).abs()
    print('%-9s max |fused - layers| = %.4g  mean %.3g  (ref max %.3g)  nan: %s' % (k, d.max(), d.mean(), outs['0'][k].abs().max(), bool(torch.isnan(outs['1'][k]).any())))
ref = torch.from_numpy(g['y_pred'])
for mode in ('0', '1'):
    d = (outs[mode]['y_pred'][:3] - ref).abs()
    print('mode %s vs f64 oracle posteriors: max %.4g mean %.3g; argmax agreement %.4f' % (
        mode, d.max(), d.mean(), (outs[mode]['y_pred'][:3].argmax(-1) == ref.argmax(-1)).float().mean()))
