"""Where a step of the MFMA recurrence spends its cycles (-DVC_ABLATE build: s_memtime sums per phase, wave 0 of one
workgroup).  VC_LIB_PATH=build/libvc_hip_ablate.so python tools/gru_phase_stamps.py [H] [windows]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'speech-cloner_amd')):
    sys.path.insert(0, p)
import torch, modules, _vc
assert _vc.lib().vc_ablate_build()
H = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
T = 400
st = modules.VariableStore('bfloat16')
x = (torch.randn(N, T, H, device='cuda') * 0.5).to(st.dtype)
_vc.set_option('gru_mfma', 1)
if len(sys.argv) > 3:
    _vc.set_option('gru_mfma4', int(sys.argv[3]))
with modules.variable_store(st), modules.variable_scope('g'):
    for _ in range(3):
        modules.gru(x, num_units=H, bidirection=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        modules.gru(x, num_units=H, bidirection=True)
    e1.record()
torch.cuda.synchronize()
buf = (C.c_ulonglong * 16)()
fn = _vc.lib().vc_ablate_read_gru_stamps
fn.restype = C.c_int
assert fn(buf) == 0
four = _vc.get_option('gru_mfma4') == 1
if four:
    names = ["h fragments read, next step's projections requested", "r and u products issued, r's sigmoids in u's gaps", 'r*h stored', 'barrier A',
             "candidate products, u's sigmoids, tanh, update, h stored", 'output stores issued', 'barrier B']
else:
    names = ['h fragments read + gate MFMAs issued', 'gate results, sigmoids, r*h stored', 'barrier A', 'r*h fragments read + candidate MFMAs issued',
             'candidate results, tanh, update, stores', 'barrier B']
tot = sum(buf[i] for i in range(len(names)))
print('H = %d, %d windows: %.3f ms per bidirectional GRU incl. projection (events); stamped workgroup: %d cycles per step' % (H, N, e0.elapsed_time(e1) / 5, tot / T))
for i, nm in enumerate(names):
    print('   %-50s %7.0f cycles per step  (%4.1f %%)' % (nm, buf[i] / T, 100.0 * buf[i] / tot))
