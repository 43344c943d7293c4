"""ctypes binding of libvc_hip.so (C ABI declared in include/vc_hip.h).

This is the only place Python touches the native library.  There is NO CPU fallback: if the
library is missing or an entry point fails, an exception is raised (``VCError``).
"""
import contextlib
import ctypes as C
import os
import threading

# Caller requirement (documented in INTEGRATION.md, not enforced here -- this module never writes os.environ):
# independent work (window chunks in decoder.predict(n_streams=...), consecutive batches) is pipelined over HIP streams,
# and the HIP runtime folds streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); two streams sharing a queue
# serialise.  A caller that keeps more than ~3 batches in flight exports GPU_MAX_HW_QUEUES=16 BEFORE the process makes
# its first GPU call (bench.py does); results do not depend on it.

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('VC_LIB_PATH', os.path.join(_HERE, 'libvc_hip.so'))   # override: kernel A/B experiments

VC_OK = 0
VC_ABI_VERSION = 4      # include/vc_hip.h: VC_ABI_VERSION -- lib() refuses a library that reports another one


class VCError(RuntimeError):
    pass


class FrontendCfg(C.Structure):
    """struct vc_frontend_cfg (include/vc_hip.h)."""
    _fields_ = [('sample_rate', C.c_int32), ('hop_length', C.c_int32), ('win_length', C.c_int32),
                ('n_fft', C.c_int32), ('n_mels', C.c_int32), ('n_mfcc', C.c_int32),
                ('pre_emphasis', C.c_float), ('mean_abs_amp_norm', C.c_float),
                ('mfcc_norm_factor', C.c_float), ('M_dB_norm_factor', C.c_float),
                ('P_dB_norm_factor', C.c_float),
                ('mfcc_normaleze_first_mfcc', C.c_int32), ('calc_mfcc_derivate', C.c_int32),
                ('clip_output', C.c_int32)]


VC_F32, VC_BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH = 0, 1, 2, 3
GEMM_PLAIN, GEMM_HIGHWAY = 0, 1
GEMM_MAX_GROUPS = 32


class GemmGroup(C.Structure):
    """struct vc_gemm_group (include/vc_hip.h)."""
    _fields_ = [('d_Bt', C.c_void_p), ('K', C.c_int32), ('taps', C.c_int32), ('pad_l', C.c_int32),
                ('c_off', C.c_int32)]


class GemmDesc(C.Structure):
    """struct vc_gemm_desc (include/vc_hip.h)."""
    _fields_ = [('dtype', C.c_int32), ('mode', C.c_int32), ('d_X', C.c_void_p),
                ('M', C.c_int32), ('T', C.c_int32), ('Cin', C.c_int32), ('ldx', C.c_int32),
                ('N', C.c_int32), ('n_groups', C.c_int32),
                ('groups', GemmGroup * GEMM_MAX_GROUPS),
                ('d_pro_scale', C.c_void_p), ('d_pro_shift', C.c_void_p),
                ('pro_relu', C.c_int32), ('pro_pool', C.c_int32),
                ('d_epi_scale', C.c_void_p), ('d_epi_shift', C.c_void_p), ('act', C.c_int32),
                ('d_R', C.c_void_p), ('ldr', C.c_int32), ('d_C', C.c_void_p), ('ldc', C.c_int32),
                ('out_f32', C.c_int32), ('drop_keep', C.c_float), ('drop_seed', C.c_ulonglong),
                ('sum_groups', C.c_int32), ('epi_pool', C.c_int32),
                ('d_workspace', C.c_void_p), ('workspace_bytes', C.c_size_t)]


CBHG_FRONT_MAX_HIGHWAY = 4


class CbhgFrontDesc(C.Structure):
    """struct vc_cbhg_front_desc (include/vc_hip.h)."""
    _fields_ = [('d_x', C.c_void_p), ('x_f32', C.c_int32), ('ldx', C.c_int32), ('n_windows', C.c_int32), ('T', C.c_int32),
                ('n_features', C.c_int32), ('prenet_units', C.c_int32), ('width', C.c_int32), ('n_banks', C.c_int32),
                ('bank_filters', C.c_int32), ('n_highway', C.c_int32), ('gru_units', C.c_int32),
                ('d_pk_dense1', C.c_void_p), ('d_pk_dense2', C.c_void_p), ('d_pk_bank', C.c_void_p),
                ('d_pk_proj1', C.c_void_p), ('d_pk_proj2', C.c_void_p), ('d_pk_gru', C.c_void_p),
                ('d_pk_highway', C.c_void_p * CBHG_FRONT_MAX_HIGHWAY),
                ('d_coef', C.c_void_p),
                ('d_xproj', C.c_void_p), ('ldp', C.c_int32)]


class LayoutItem(C.Structure):
    """struct vc_layout_item (include/vc_hip.h)."""
    _fields_ = [('src', C.c_void_p), ('dst', C.c_void_p), ('k', C.c_int32), ('cin', C.c_int32), ('cout', C.c_int32),
                ('mode', C.c_int32)]


class W16Item(C.Structure):
    """struct vc_w16_item (include/vc_hip.h)."""
    _fields_ = [('src', C.c_void_p), ('dst', C.c_void_p), ('scale_dst', C.c_void_p), ('k', C.c_int32), ('cin', C.c_int32),
                ('cout', C.c_int32), ('mode', C.c_int32), ('row_len', C.c_int32), ('tap_stride', C.c_int32),
                ('plane_stride', C.c_int32), ('base', C.c_int32), ('group', C.c_int32), ('scale_n', C.c_int32)]


class Gemm16Pair(C.Structure):
    """struct vc_gemm16_pair (include/vc_hip.h)."""
    _fields_ = [('d_Bt0', C.c_void_p), ('d_Bt1', C.c_void_p), ('taps0', C.c_int32), ('extra', C.c_int32),
                ('pad_l', C.c_int32), ('c_off0', C.c_int32), ('c_off1', C.c_int32), ('row0', C.c_int32),
                ('nrows0', C.c_int32), ('nrows1', C.c_int32), ('s_off0', C.c_int32), ('s_off1', C.c_int32)]


class Gemm16Desc(C.Structure):
    """struct vc_gemm16_desc (include/vc_hip.h)."""
    _fields_ = [('d_X16', C.c_void_p), ('d_row_scale', C.c_void_p), ('M', C.c_int32), ('T', C.c_int32), ('C', C.c_int32),
                ('ldx', C.c_int32), ('n_pairs', C.c_int32), ('ragged', C.c_int32), ('pairs', Gemm16Pair * 16),
                ('d_col_scale', C.c_void_p), ('d_col_shift', C.c_void_p), ('d_C', C.c_void_p), ('ldc', C.c_int32),
                ('accumulate', C.c_int32), ('act', C.c_int32), ('atomic_splits', C.c_int32), ('d_workspace', C.c_void_p),
                ('workspace_bytes', C.c_size_t)]


class WgradGroup(C.Structure):
    """struct vc_wgrad_group (include/vc_hip.h)."""
    _fields_ = [('d_dYT', C.c_void_p), ('d_dW', C.c_void_p), ('N', C.c_int32), ('taps', C.c_int32),
                ('shift0', C.c_int32), ('ldw', C.c_int32)]


class WgradDesc(C.Structure):
    """struct vc_wgrad_desc (include/vc_hip.h)."""
    _fields_ = [('d_XT', C.c_void_p), ('ldxt', C.c_int32), ('ldyt', C.c_int32), ('Cin', C.c_int32),
                ('M', C.c_int32), ('T', C.c_int32), ('margin', C.c_int32), ('n_groups', C.c_int32),
                ('splits_allowed', C.c_int32), ('groups', WgradGroup * GEMM_MAX_GROUPS)]


_lib = None
_lock = threading.Lock()

_P = C.c_void_p
_SIGS = {
    'vc_version': (C.c_int, []),
    'vc_last_error': (C.c_char_p, []),
    'vc_target_arch': (C.c_char_p, []),
    'vc_set_option': (C.c_int, [C.c_char_p, C.c_int]),
    'vc_get_option': (C.c_int, [C.c_char_p, C.POINTER(C.c_int)]),
    'vc_ablate_build': (C.c_int, []),
    'vc_frontend_host_tables': (C.c_int, [C.POINTER(FrontendCfg), _P, _P]),
    'vc_frontend_plan_create': (C.c_int, [C.POINTER(FrontendCfg), _P, C.POINTER(_P)]),
    'vc_frontend_plan_destroy': (None, [_P]),
    'vc_frontend_num_frames': (C.c_int32, [_P, C.c_int32]),
    'vc_frontend_mfcc_width': (C.c_int32, [_P]),
    'vc_frontend_power_width': (C.c_int32, [_P]),
    'vc_frontend_get_mel': (C.c_int, [_P, _P]),
    'vc_frontend_get_dct': (C.c_int, [_P, _P]),
    'vc_frontend_workspace_bytes': (C.c_size_t, [_P, C.c_int32, C.c_int32]),
    'vc_frontend_f32': (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P,
                                  C.c_size_t, _P]),
    'vc_frontend_stages_f32': (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P,
                                         C.c_size_t, _P, C.c_int32]),
    'vc_conv_gemm_workspace_bytes': (C.c_size_t, [C.POINTER(GemmDesc)]),
    'vc_conv_gemm': (C.c_int, [C.POINTER(GemmDesc), _P]),
    'vc_softmax_argmax': (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32, C.c_int32, _P, _P]),
    'vc_softmax_argmax_dual': (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32, _P, C.c_int32, _P, _P]),
    'vc_gru_workspace_bytes': (C.c_size_t, [C.c_int32, C.c_int32]),
    'vc_gru_bidir': (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32, _P,
                               C.c_size_t, _P]),
    'vc_lstm_bidir': (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32, _P]),
    'vc_convert': (C.c_int, [_P, C.c_int32, _P, C.c_int32, C.c_size_t, _P]),
    'vc_conv_wgrad': (C.c_int, [C.POINTER(WgradDesc), _P]),
    'vc_transpose_pad': (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, C.c_int32, C.c_int32,
                                   C.c_int32, _P, C.c_int32, C.c_int32, C.c_int32, _P]),
    'vc_stats_workspace_floats': (C.c_size_t, [C.c_int32, C.c_int32]),
    'vc_bn_train_stats': (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, C.c_float, C.c_float,
                                    _P, _P, _P, _P, _P, _P]),
    'vc_affine_act': (C.c_int, [_P, _P, _P, C.c_int32, _P, _P, C.c_size_t, C.c_int32, _P]),
    'vc_bn_backward': (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P,
                                 C.c_int32, _P, _P, _P, _P, _P]),
    'vc_lstm_train_forward': (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P]),
    'vc_lstm_backward': (C.c_int, [_P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    'vc_bn_post_routing': (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P]),
    'vc_relu_dropout_backward': (C.c_int, [_P, _P, C.c_float, _P, C.c_size_t, _P]),
    'vc_highway_backward': (C.c_int, [_P, C.c_int32, _P, _P, C.c_int32, C.c_int32, _P, _P, _P]),
    'vc_col_sum': (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32, _P, _P]),
    'vc_fill': (C.c_int, [_P, C.c_float, C.c_size_t, _P]),
    'vc_weight_layouts': (C.c_int, [_P, C.c_int32, _P]),
    'vc_split16': (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, C.c_int32, C.c_int32, _P, _P, _P]),
    'vc_weights16': (C.c_int, [_P, C.c_int32, _P, C.c_int32, _P]),
    'vc_transpose_split16': (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, C.c_int32, C.c_int32, C.c_int32,
                                       C.c_int32, _P, _P, _P]),
    'vc_gemm16_workspace_bytes': (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    'vc_gemm16': (C.c_int, [C.POINTER(Gemm16Desc), _P]),
    'vc_axpby': (C.c_int, [_P, C.c_int32, C.c_float, _P, C.c_int32, C.c_float, _P, C.c_int32, C.c_int32, C.c_int32, _P]),
    'vc_mse_loss': (C.c_int, [_P, _P, C.c_size_t, C.c_float, _P, C.c_int32, C.c_int32, _P, _P, _P]),
    'vc_softmax_ce': (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32, _P, _P, _P]),
    'vc_adam_step': (C.c_int, [_P, _P, _P, _P, C.c_size_t, C.c_float, C.c_float, C.c_float, C.c_float,
                               C.c_float, _P]),
    'vc_gru_train_forward': (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P]),
    'vc_gru_backward': (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    'vc_conv_gemm_epi_pool_supported': (C.c_int, [C.POINTER(GemmDesc)]),
    'vc_highway_pack': (C.c_int, [_P, C.c_int32, C.c_int32, _P, _P]),
    'vc_highway_chain': (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(_P), C.POINTER(_P), _P, C.c_int32,
                                   _P, _P, C.c_int32, _P, C.c_int32, _P]),
    'vc_mfma_pack': (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    'vc_prenet_chain_supported': (C.c_int, [C.c_int32] * 3),
    'vc_prenet_chain': (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, C.c_int32, _P]),
    'vc_cbhg_front_coef_floats': (C.c_int32, []),
    'vc_cbhg_front_supported': (C.c_int, [C.c_int32] * 8),
    'vc_cbhg_front': (C.c_int, [C.POINTER(CbhgFrontDesc), _P]),
    'vc_gather_rows': (C.c_int, [_P, _P, _P, C.c_int64, C.c_int32, _P, _P]),
    'vc_vocoder_plan_create': (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _P, C.POINTER(_P)]),
    'vc_vocoder_plan_destroy': (None, [_P]),
    'vc_vocoder_num_samples': (C.c_int32, [_P, C.c_int32]),
    'vc_vocoder_workspace_bytes': (C.c_size_t, [_P, C.c_int32, C.c_int32, C.c_int32]),
    'vc_power_to_amp': (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, _P, _P]),
    'vc_griffin_lim_f32': (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32, _P, _P,
                                     C.c_size_t, _P]),
    'vc_inv_preemphasis_normalize': (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, _P]),
}


def lib():
    """Load (once) and return the ctypes handle of libvc_hip.so; raises VCError if absent."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise VCError('native library %s not found -- run `python -c "import __graft_entry__ as g; '
                          'g.build()"` (or make -C speech-cloner_amd/csrc); there is no CPU fallback'
                          % LIB_PATH)
        # torch ships its own libamdhip64; it must be the HIP runtime of the process.  Loading this
        # library first would pull in the system copy and leave two runtimes ("no ROCm-capable
        # device" from the second one), so torch is imported before the dlopen.
        import torch  # noqa: F401
        h = C.CDLL(LIB_PATH)
        h.vc_version.restype, h.vc_version.argtypes = C.c_int, []
        got = h.vc_version()
        if got != VC_ABI_VERSION:
            # a stale build (or a VC_LIB_PATH override built from older sources) would take shifted pointer / size
            # arguments: refuse it here instead of faulting on the device
            raise VCError('native library %s reports ABI version %d, this binding is written for %d -- rebuild it '
                          '(make -C speech-cloner_amd/csrc)' % (LIB_PATH, got, VC_ABI_VERSION))
        for name, (res, args) in _SIGS.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


def set_option(name, value):
    """vc_set_option (include/vc_hip.h): pick between equivalent HIP kernels; -1 restores the default."""
    check(lib().vc_set_option(name.encode(), int(value)))


def get_option(name):
    v = C.c_int(0)
    check(lib().vc_get_option(name.encode(), C.byref(v)))
    return v.value


@contextlib.contextmanager
def options(**kw):
    """with _vc.options(gru_mfma=1): ...  -- sets, then restores, library options."""
    old = {k: get_option(k) for k in kw}
    try:
        for k, v in kw.items():
            set_option(k, v)
        yield
    finally:
        for k, v in old.items():
            set_option(k, v)


# Two kernels fill an otherwise idle chip at the price of extra workgroup time: the k = 3 projection's K split over two
# workgroups per row tile (-43 % for that launch alone, +0.8 % per step when ten batches are in flight) and the
# front-end's one-launch form, whose blocks WAIT for their utterance (-14 % alone, +5.5 % per pipelined step: a waiting
# block keeps its CU from the register-filling MFMA launches of the other streams).  The library's defaults serve a
# caller with one batch at a time (the reference's test.py); a caller that keeps several batches in flight on several
# streams runs its loop under this context (bench.py's throughput loop does).
THROUGHPUT_OPTIONS = {'proj256_split': 0, 'fe_fused': 0}


def throughput_mode():
    """with _vc.throughput_mode(): ...  -- the option set for callers that keep several batches in flight."""
    return options(**THROUGHPUT_OPTIONS)


def check(rc):
    if rc != VC_OK:
        msg = lib().vc_last_error()
        raise VCError('libvc_hip error %d: %s' % (rc, msg.decode('utf-8', 'replace') if msg else '?'))


def ptr(t):
    """Device/host pointer of a torch tensor or numpy array (None -> NULL)."""
    if t is None:
        return None
    if hasattr(t, 'data_ptr'):
        return C.c_void_p(t.data_ptr())
    return C.c_void_p(t.ctypes.data)


def current_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
