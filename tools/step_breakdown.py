"""One encode+decode pass (64 windows, bf16, single stream) bracketed by marker launches, for
rocprofv3 --kernel-trace:  per-kernel time inside one step.
  rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/step_breakdown.py
  python3 tools/step_breakdown.py --parse OUT/*/*kernel_trace.csv"""
import os, sys, csv, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'speech-cloner_amd')):
    sys.path.insert(0, p)
if len(sys.argv) > 2 and sys.argv[1] == '--parse':
    rows = list(csv.DictReader(open(sys.argv[2])))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    marks = [i for i, r in enumerate(rows) if 'fill_kernel' in r['Kernel_Name']]
    a, b = marks[-2], marks[-1]
    if '--list' in sys.argv:
        for r in rows[a + 1:b]:
            d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
            print('%8.1f us  grid %-9s wg %-5s lds %-7s %s' % (d, r.get('Grid_Size', '?'), r.get('Workgroup_Size', '?'),
                                                          r.get('LDS_Block_Size', '?'), r['Kernel_Name'][:90]))
    agg = collections.OrderedDict()
    for r in rows[a + 1:b]:
        n = r['Kernel_Name'][:70]
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        c = agg.setdefault(n, [0, 0.0]); c[0] += 1; c[1] += d
    tot = sum(v[1] for v in agg.values())
    span = (int(rows[b]['Start_Timestamp']) - int(rows[a]['End_Timestamp'])) / 1e3
    for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('%8.1f us %5.1f%%  x%-3d %s' % (d, 100 * d / tot, c, n))
    print('kernel sum %.1f us, wall span %.1f us' % (tot, span))
    sys.exit(0)
import contextlib, io, json
import numpy as np, torch
import bench, _vc, ctypes as C
from encoder import encoder_spec_phn
from decoder import decoder_specs
from aux_func import load_cfg_d
hp = os.path.join(ROOT, 'speech-cloner_amd', 'hp')
with contextlib.redirect_stdout(io.StringIO()):
    ec = load_cfg_d(os.path.join(hp, 'encoder_cfg_d.json')); dc = load_cfg_d(os.path.join(hp, 'decoder_cfg_d.json'))
    ec.update(is_training=False, compute_dtype='bfloat16', model_path=os.path.join(ROOT, 'tests', 'golden', 'enc_14_ckpt'))
    dc.update(is_training=False, compute_dtype='bfloat16')
    enc = encoder_spec_phn(ec, None); enc.restore()
    dec = decoder_specs(dc, None, enc)
x = (torch.rand(64, 400, 80, device='cuda') * 0.4 - 0.2)
mark = torch.zeros(1024, device='cuda')
def marker():
    _vc.check(_vc.lib().vc_fill(_vc.ptr(mark), 0.0, 1024, _vc.current_stream()))
for _ in range(3):
    dec.forward(x)
torch.cuda.synchronize()
marker(); dec.forward(x); marker()
torch.cuda.synchronize()
