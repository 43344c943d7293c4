"""Ablation timings of cbhg_small_kernel (option ablate_cbhg_front bits, -DVC_ABLATE build; results are wrong by design): 20 launches of
the fused front for rocprofv3 --kernel-trace --stats (host overhead per call exceeds the kernel: no event timing)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'speech-cloner_amd')):
    sys.path.insert(0, p)
import numpy as np, torch, bench, modules, _vc
assert _vc.lib().vc_ablate_build(), 'needs the -DVC_ABLATE library (tools/build_ablate.sh, VC_LIB_PATH)'
_vc.set_option('cbhg_front_mi', int(os.environ.get('CFD_MI', '2')))
_vc.set_option('ablate_cbhg_front', int(os.environ.get('CFD_DBG', '0')))
from encoder import encoder_spec_phn
cfg = json.load(open(os.path.join(ROOT, 'speech-cloner_amd', 'hp', 'encoder_cfg_d.json')))
cfg.update(is_training=False, model_path=os.path.join(ROOT, 'tests', 'golden', 'enc_14_ckpt'), compute_dtype='bfloat16')
enc = encoder_spec_phn(cfg, None); enc.restore()
x = (torch.rand(64, 400, 80, device='cuda') * 0.4 - 0.2)
def front():
    with modules.variable_store(enc.store), modules.variable_scope(enc._scope):
        return modules._cbhg_front(x, 80, 6, 1, 'prenet', 'CBHG')
for _ in range(20):
    front()
torch.cuda.synchronize()
