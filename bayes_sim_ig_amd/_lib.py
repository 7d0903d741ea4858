"""ctypes binding of libbsig_hip.so (include/bsig.h).

The product path has NO fallback: if the library is missing or there is no
GPU, every compute entry point raises.  torch is used only for device memory,
streams and torch.distributed.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# (BSIG_LIB_PATH: another build of the same library, for A/B measurements of kernel variants)
LIB_PATH = os.environ.get('BSIG_LIB_PATH') or os.path.join(_HERE, 'lib', 'libbsig_hip.so')

BSIG_OK, BSIG_EINVAL, BSIG_ELAUNCH, BSIG_EUNSUPPORTED, BSIG_ENONFINITE = 0, -1, -2, -3, -4
EPI_NONE, EPI_BIAS, EPI_BIAS_ACT, EPI_COS_SIN, EPI_COS_OFF, EPI_MUL_DACT = range(6)
ACT_TANH, ACT_RELU, ACT_LEAKY_RELU, ACT_SIGMOID, ACT_IDENTITY = range(5)
MAX_HIDDEN = 8
FIT_GRAPH, FIT_SPLIT_ADAM = 1, 2
PLAN_NO_PERSISTENT = 1
X_ROWS, X_CROSSCORR_FACTORS = 0, 1

i64, i32, u64, f32, vp, sz = (C.c_int64, C.c_int32, C.c_uint64, C.c_float,
                              C.c_void_p, C.c_size_t)


class HeadDims(C.Structure):
    _fields_ = [('out_dim', i32), ('n_comp', i32), ('full_cov', i32),
                ('eps_noise', f32), ('min_weight', f32), ('ll_limit', f32)]


class MdnCfg(C.Structure):
    _fields_ = [('input_dim', i32), ('n_hidden', i32),
                ('hidden', i32 * MAX_HIDDEN), ('activation', i32),
                ('rff_feats', i32), ('rff_cos_only', i32),
                ('rff_scale', f32), ('head', HeadDims),
                ('lr', f32), ('beta1', f32), ('beta2', f32), ('adam_eps', f32)]


class FitBuffers(C.Structure):
    _fields_ = [('params', vp), ('grads', vp), ('exp_avg', vp),
                ('exp_avg_sq', vp),
                ('rff_coeff', vp), ('ld_coeff', i64), ('rff_offset', vp),
                ('x_train', vp), ('ldx_train', i64), ('n_train', i64),
                ('y_train', vp), ('ldy_train', i64),
                ('x_test', vp), ('ldx_test', i64), ('n_test', i64),
                ('y_test', vp), ('ldy_test', i64),
                ('ids_table', vp), ('train_loss', vp), ('test_loss', vp),
                ('state', vp), ('workspace', vp), ('workspace_bytes', sz),
                ('x_kind', i32), ('x_s', i32), ('x_a', i32),
                ('x_test_factors', vp), ('ldx_test_factors', i64)]


_PROTOS = {
    'bsig_last_error': (C.c_char_p, []),
    'bsig_version': (C.c_int, []),
    'bsig_abi_info': (u64, [C.c_int]),
    'bsig_device_count': (C.c_int, []),
    'bsig_summary_dim': (i64, [C.c_int] * 5),
    'bsig_summary_start': (C.c_int, [vp, vp, vp, i64] + [C.c_int] * 5 + [i64, vp]),
    'bsig_crosscorr': (C.c_int, [vp, vp, vp, i64] + [C.c_int] * 5 + [i64, vp, vp]),
    'bsig_crosscorr_factor_dims': (C.c_int, [C.c_int] * 3 + [C.POINTER(i32)] * 2),
    'bsig_crosscorr_factors': (C.c_int, [vp, vp, vp, i64] + [C.c_int] * 5 + [i64, vp, vp]),
    'bsig_crosscorr_expand': (C.c_int, [vp, i64, i64, C.c_int, C.c_int, vp, i64, vp]),
    'bsig_signature': (C.c_int, [vp, vp, vp, i64] + [C.c_int] * 4 + [i64, vp]),
    'bsig_gemm_workspace_bytes': (sz, [i64, i64, i64]),
    'bsig_gemm_f32': (C.c_int, [vp, i64, C.c_int, vp, vp, i64, C.c_int, vp, vp,
                                i64, i64, i64, i64, C.c_int, C.c_int, vp, vp,
                                i64, f32, vp, sz, vp]),
    'bsig_rff_coeff': (C.c_int, [vp, vp, vp, i64, i64, i64, vp]),
    'bsig_rff_project': (C.c_int, [vp, i64, vp, vp, i64, vp, vp, i64, i64, i64,
                                   i64, f32, C.c_int, vp, sz, vp]),
    'bsig_head_width': (i64, [C.POINTER(HeadDims)]),
    'bsig_head_workspace_bytes': (sz, [C.POINTER(HeadDims), i64]),
    'bsig_mdn_head_outputs': (C.c_int, [C.POINTER(HeadDims), vp, i64, i64, vp,
                                        u64, u64, vp, vp, vp, vp, vp, vp, sz, vp]),
    'bsig_mdn_nll_from_tuple': (C.c_int, [C.POINTER(HeadDims), vp, vp, vp, vp,
                                          vp, i64, i64, vp, vp, vp, sz, vp]),
    'bsig_mdn_head_nll': (C.c_int, [C.POINTER(HeadDims), vp, i64, vp, i64, vp,
                                    i64, i64, vp, u64, u64, vp, vp, vp, vp, sz,
                                    vp]),
    'bsig_adam_flat': (C.c_int, [vp, vp, vp, vp, i64, f32, f32, f32, f32, i64, vp]),
    'bsig_colsum': (C.c_int, [vp, i64, i64, i64, vp, vp, sz, vp]),
    'bsig_normalize_rows': (C.c_int, [vp, i64, vp, vp, vp, i64, i64, i64, vp]),
    'bsig_copy_rows': (C.c_int, [vp, i64, vp, vp, i64, i64, i64, vp]),
    'bsig_mdn_param_count': (i64, [C.POINTER(MdnCfg)]),
    'bsig_mdn_param_offsets': (C.c_int, [C.POINTER(MdnCfg), C.POINTER(i64), C.c_int]),
    'bsig_mdn_workspace_bytes': (sz, [C.POINTER(MdnCfg), i64]),
    'bsig_mdn_head_forward': (C.c_int, [C.POINTER(MdnCfg), vp, vp, i64, vp, vp,
                                        i64, vp, i64, vp, i64, vp, sz, vp]),
    'bsig_mdn_loss_grad': (C.c_int, [C.POINTER(MdnCfg), vp, vp, i64, vp, vp, i64,
                                     vp, i64, vp, i64, i64, vp, u64, u64, vp, vp,
                                     vp, vp, sz, vp]),
    'bsig_fit_create': (C.c_int, [C.POINTER(MdnCfg), i64, i64, i64, C.POINTER(vp)]),
    'bsig_fit_create_sized': (C.c_int, [C.POINTER(MdnCfg), i64, i64, i64, i64, C.POINTER(vp)]),
    'bsig_fit_create_ex': (C.c_int, [C.POINTER(MdnCfg), i64, i64, i64, i64, C.c_int, C.POINTER(vp)]),
    'bsig_fit_destroy': (None, [vp]),
    'bsig_fit_workspace_bytes': (sz, [vp]),
    'bsig_fit_bind': (C.c_int, [vp, C.POINTER(FitBuffers), C.c_int]),
    'bsig_fit_begin': (C.c_int, [vp, u64, i64, vp]),
    'bsig_fit_set_features': (C.c_int, [vp, vp, i64, i64, vp]),
    'bsig_fit_run': (C.c_int, [vp, i64, vp]),
    'bsig_fit_grad': (C.c_int, [vp, vp]),
    'bsig_fit_apply': (C.c_int, [vp, vp]),
    'bsig_fit_flush': (C.c_int, [vp, vp]),
    'bsig_fit_is_persistent': (C.c_int, [vp]),
    'bsig_fit_accepts_factors': (C.c_int, [vp]),
    'bsig_fit_accepts_factor_rows': (C.c_int, [vp, C.c_int, C.c_int]),
    'bsig_fit_evaluates_from_factors': (C.c_int, [vp, C.c_int, C.c_int, C.c_int]),
    'bsig_fit_takes_features': (C.c_int, [vp, i64]),
    'bsig_fit_eval': (C.c_int, [vp, vp]),
    'bsig_fit_updates': (C.c_int, [vp, i64, vp]),
    'bsig_debug_persist_profile': (None, [vp]),
    'bsig_debug_spin': (C.c_int, [C.c_int, sz, C.c_int, vp]),
    'bsig_debug_persist_geometry': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32)]),
    'bsig_debug_mfma_vs_fma': (C.c_int, [vp, vp, C.c_int, vp, vp]),
    'bsig_debug_persist_mdnn_geometry': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32)]),
    'bsig_comm_unique_id': (C.c_int, [vp]),
    'bsig_comm_init': (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]),
    'bsig_comm_init_external': (C.c_int, [C.c_int, C.c_int, vp, vp, C.POINTER(vp)]),
    'bsig_comm_transport': (C.c_int, [vp]),
    'bsig_comm_world': (C.c_int, [vp]),
    'bsig_fit_dp_graph_status': (C.c_int, [vp, C.c_char_p, sz]),
    'bsig_comm_rank': (C.c_int, [vp]),
    'bsig_comm_resident_calls': (i64, [vp]),
    'bsig_comm_set_resident': (None, [vp, C.c_int]),
    'bsig_comm_resident_mode': (C.c_int, [vp]),
    'bsig_comm_allreduce': (C.c_int, [vp, vp, i64, vp]),
    'bsig_comm_broadcast': (C.c_int, [vp, vp, i64, C.c_int, vp]),
    'bsig_comm_destroy': (None, [vp]),
    'bsig_fit_pack_logs': (C.c_int, [vp, i64, vp, vp]),
    'bsig_fit_run_dp': (C.c_int, [vp, vp, i64, vp, vp]),
}
COMM_ID_BYTES = 128
EXCHANGE_SUM, EXCHANGE_BROADCAST = 0, 1
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, vp, C.c_int, vp, i64, C.c_int, vp)

_lib = None


def exported_symbols():
    """Names every declaration of include/bsig.h must resolve to."""
    return sorted(_PROTOS)


def load():
    """Load the shared library (no GPU needed for loading)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                'libbsig_hip.so not built (%s): run ./build.sh or '
                '__graft_entry__.build(); there is no CPU fallback' % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _check_abi(lib)
        _lib = lib
    return _lib


HEADER_PATH = os.path.join(os.path.dirname(_HERE), 'include', 'bsig.h')


def header_hash(path=HEADER_PATH):
    """64-bit FNV-1a of include/bsig.h (what build.sh hands the compiler as BSIG_HEADER_HASH)."""
    h = 0xcbf29ce484222325
    with open(path, 'rb') as f:
        for b in f.read():
            h = ((h ^ b) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return h


def _check_abi(lib):
    """The library's view of include/bsig.h against this binding's: struct sizes always, the
    header text when the header is next to the package (a library or an object file built against
    another revision of the header must not be driven through these ctypes mirrors)."""
    want = {1: C.sizeof(HeadDims), 2: C.sizeof(MdnCfg), 3: C.sizeof(FitBuffers),
            4: FitBuffers.x_kind.offset}
    for which, size in want.items():
        got = int(lib.bsig_abi_info(which))
        if got != size:
            raise RuntimeError('libbsig_hip.so was built against another include/bsig.h (layout %d: '
                               'library %d, binding %d): rebuild with ./build.sh' % (which, got, size))
    built = int(lib.bsig_abi_info(0))
    if built and os.path.exists(HEADER_PATH) and built != header_hash():
        raise RuntimeError('libbsig_hip.so was built from another revision of include/bsig.h '
                           '(hash %016x, header %016x): rebuild with ./build.sh'
                           % (built, header_hash()))


def require_gpu():
    """The product path runs on MI355X only — fail loudly otherwise."""
    lib = load()
    if not torch.cuda.is_available():
        raise RuntimeError('bayes_sim_ig_amd needs a ROCm GPU (MI355X): '
                           'torch.cuda.is_available() is False and there is '
                           'no CPU fallback')
    return lib


def check(rc):
    if rc == BSIG_OK:
        return
    msg = load().bsig_last_error().decode('utf-8', 'replace')
    if rc == BSIG_EINVAL:
        raise AssertionError(msg)        # the reference asserts on bad shapes
    if rc == BSIG_EUNSUPPORTED:
        raise NotImplementedError(msg)
    raise RuntimeError('libbsig_hip: %s (code %d)' % (msg, rc))


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream(device=None):
    """torch's current stream of ``device`` (default: the current device) as a hipStream_t."""
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def on_device(device):
    """Context: make ``device`` the current HIP device (kernels launch on the current
    device; a model built on cuda:1 must not launch on cuda:0's stream)."""
    return torch.cuda.device(device)


def as_f32_rows(t, device=None):
    """fp32, last dim contiguous, on the GPU.  Returns (tensor, ld)."""
    if hasattr(t, 'materialize'):      # a lazy summary handle (summarizers.CrossCorrFactors)
        t = t.materialize()
    if t.dtype != torch.float32:
        t = t.float()
    if device is not None and t.device != torch.device(device):
        t = t.to(device)
    if not t.is_cuda:
        raise RuntimeError('expected a GPU tensor (no CPU fallback)')
    if t.dim() != 2:
        raise AssertionError('expected a 2-D tensor')
    if t.stride(1) != 1 or (t.shape[0] > 1 and t.stride(0) < t.shape[1]):
        t = t.contiguous()
    ld = t.stride(0) if t.shape[0] > 1 else max(t.shape[1], 1)
    return t, ld


def round_up(x, m):
    return (x + m - 1) // m * m
