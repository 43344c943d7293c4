"""Per-tensor gradient error of one decoder training step at the shipped sizes vs autograd on the float64 oracle
(the body of tests/test_training_gpu.py::test_hp_size_train_step_matches_autograd, printing every tensor).
  python tools/train_parity_report.py [N_windows] [dropout]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'speech-cloner_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np, torch
import test_training_gpu as tt
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cfg = tt._hp_cfg()
if len(sys.argv) > 2:
    cfg['dropout_rate'] = float(sys.argv[2])
dec, w, ppg, t_mel, t_stft = tt._setup(cfg, N=N)
tr = dec._get_trainer()
losses = tr.forward_backward(*(torch.from_numpy(a).cuda() for a in (ppg, t_mel, t_stft)))
ml, sl, grads, stats, ym, ys = tt._oracle_step(cfg, w, ppg, t_mel, t_stft, tr.seed)
print('losses', losses.cpu().numpy(), ml, sl)
print('y_mel err %.3e  y_stft err %.3e' % (np.abs(tr.y_mel.cpu().numpy().reshape(ym.shape) - ym).max(),
                                             np.abs(tr.y_stft.cpu().numpy().reshape(ys.shape) - ys).max()))
rows = []
for n in tr.names:
    g = tr.g(n).cpu().numpy().astype(np.float64)
    ref = grads[n]
    rows.append((np.abs(g - ref).max() / max(np.abs(ref).max(), 1e-12), np.abs(ref).max(), n))
for e, m, n in rows:
    print('%.3e  max|g| %.3e  %s' % (e, m, n))
