"""ctypes binding of libadenet_hip.so (the C ABI declared in include/adenet.h).

The library is the ONLY compute path: if it is missing or cannot be loaded this module raises --
there is deliberately no CPU fallback (the CPU restatement under oracle/ is test infrastructure
and is never imported from here).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libadenet_hip.so")

ADN_MAX_STREAMS = 8
ADN_MAX_ENC_LAYERS = 8
ADN_MAX_CLASSES = 64

ADN_OK, ADN_ERR_INVALID, ADN_ERR_HIP, ADN_ERR_NO_DEVICE, ADN_ERR_STATE = 0, 1, 2, 3, 4     # enum adn_status
ACT = {"linear": 0, "identity": 0, "rectify": 1, "sigmoid": 2, "tanh": 3, "leaky_rectify": 4,
       "very_leaky_rectify": 5, "scaled_tanh": 6, "scaled_tanh_lecun": 7}
FUSION = {"none": 0, "sum": 1, "adasum": 2, "concat": 3}
PRECISION = {"f32": 0, "fp32": 0, "float32": 0, "bf16": 1, "bfloat16": 1, "bf16x3": 2, "mixed": 3}
FLAG_DEVICE_INPUTS = 1
FLAG_STOCHASTIC = 4
FLAG_DETERMINISTIC = 8
FLAG_BF16_INPUTS = 16
FLAG_PLANE_INPUTS = 32
HEAD = {"frames": 0, "last": 1}
FLAG_DEVICE_OUTPUTS = 2
BUF_PARAM, BUF_GRAD, BUF_ADAM_M, BUF_ADAM_V = 0, 1, 2, 3


class StreamConfig(C.Structure):
    _fields_ = [("input_dim", C.c_int32), ("n_enc", C.c_int32),
                ("enc_units", C.c_int32 * ADN_MAX_ENC_LAYERS), ("enc_act", C.c_int32 * ADN_MAX_ENC_LAYERS),
                ("use_delta", C.c_int32), ("bidirectional", C.c_int32), ("peepholes", C.c_int32),
                ("dropout_p", C.c_float), ("batchnorm", C.c_int32), ("aux_dim", C.c_int32)]


class Config(C.Structure):
    _fields_ = [("n_streams", C.c_int32), ("streams", StreamConfig * ADN_MAX_STREAMS),
                ("fusion", C.c_int32), ("agg", C.c_int32), ("agg_peepholes", C.c_int32),
                ("lstm_size", C.c_int32), ("classes", C.c_int32), ("precision", C.c_int32),
                ("head", C.c_int32), ("agg_dropout_p", C.c_float), ("stream_lstm_units", C.c_int32),
                ("reserved", C.c_int32 * 5)]


class CaeConfig(C.Structure):
    _fields_ = [("image_h", C.c_int32), ("image_w", C.c_int32), ("dense", C.c_int32), ("bottleneck", C.c_int32),
                ("precision", C.c_int32), ("variant", C.c_int32), ("reserved", C.c_int32 * 2)]


class RbmConfig(C.Structure):
    _fields_ = [("num_vis", C.c_int32), ("num_hid", C.c_int32), ("vis_type", C.c_int32), ("hid_type", C.c_int32),
                ("cd_type", C.c_int32), ("batchsize", C.c_int32), ("lr_w", C.c_float), ("lr_vb", C.c_float), ("lr_hb", C.c_float),
                ("weight_penalty", C.c_float), ("reserved", C.c_int32 * 2)]


class ParamInfo(C.Structure):
    _fields_ = [("name", C.c_char * 96), ("ndim", C.c_int32), ("dims", C.c_int64 * 2), ("numel", C.c_int64)]


class BatchStream(C.Structure):
    _fields_ = [("frames", C.c_void_p), ("width", C.c_int32), ("elem_bytes", C.c_int32), ("out", C.c_void_p)]


class ProfileEntry(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("launches", C.c_int64), ("ms", C.c_double), ("flops", C.c_double),
                ("bytes", C.c_double)]


class AdenetError(RuntimeError):
    pass


# every symbol include/adenet.h declares: (restype, argtypes)
_P = C.c_void_p
_SIGNATURES = {
    "adn_version": (C.c_char_p, []),
    "adn_abi_sizes": (None, [C.POINTER(C.c_int32)]),
    "adn_last_error": (C.c_char_p, []),
    "adn_device_count": (C.c_int, []),
    "adn_create": (C.c_int, [C.POINTER(Config), C.POINTER(_P)]),
    "adn_destroy": (None, [_P]),
    "adn_set_stream": (C.c_int, [_P, _P]),
    "adn_set_precision": (C.c_int, [_P, C.c_int]),
    "adn_num_params": (C.c_int, [_P]),
    "adn_param_info": (C.c_int, [_P, C.c_int, C.POINTER(ParamInfo)]),
    "adn_read_tensor": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "adn_write_tensor": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "adn_total_param_count": (C.c_int64, [_P]),
    "adn_flat_buffer": (C.c_int, [_P, C.c_int, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "adn_flat_buffer_const": (C.c_int, [_P, C.c_int, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "adn_grad_buckets": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    "adn_set_bucket_events": (C.c_int, [_P, C.POINTER(_P), C.c_int]),
    "adn_forward": (C.c_int, [_P, C.POINTER(_P), _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "adn_loss": (C.c_int, [_P, C.POINTER(_P), _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "adn_read_probs": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P]),
    "adn_compute_grads": (C.c_int, [_P, C.POINTER(_P), _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, _P]),
    "adn_zero_grads": (C.c_int, [_P]),
    "adn_apply_adam": (C.c_int, [_P, C.c_float]),
    "adn_grad_bucket_groups": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "adn_adam_begin": (C.c_int, [_P, C.c_float]),
    "adn_adam_range": (C.c_int, [_P, C.c_int64, C.c_int64]),
    "adn_adam_ranges": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int]),
    "adn_adam_end": (C.c_int, [_P]),
    "adn_apply_adam_vlr": (C.c_int, [_P, C.POINTER(C.c_float), C.c_int]),
    "adn_adam_step_count": (C.c_int, [_P]),
    "adn_set_adam_step_count": (C.c_int, [_P, C.c_int]),
    "adn_train_step": (C.c_int, [_P, C.POINTER(_P), _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _P]),
    "adn_read_encoder_activation": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "adn_synchronize": (C.c_int, [_P]),
    "adn_set_deterministic": (C.c_int, [C.c_int]),
    "adn_set_batch_lengths": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_int]),
    "adn_get_compact_rows": (C.c_int, [C.c_void_p]),
    "adn_set_auto_compaction": (C.c_int, [C.c_void_p, C.c_int]),
    "adn_set_length_buckets": (C.c_int, [C.c_void_p, C.c_int]),
    "adn_get_bucket_rows": (C.c_int, [C.c_void_p]),
    "adn_set_relu_grad_at_zero": (C.c_int, [C.c_void_p, C.c_float]),
    "adn_get_deterministic": (C.c_int, []),
    "adn_debug_raise_exchange_error": (C.c_int, [C.c_int]),
    "adn_debug_occupy_cus": (C.c_int, [C.c_int, C.c_int, C.c_double, _P]),
    "adn_debug_lstm_family_counts": (C.c_int, [C.POINTER(C.c_int64)]),
    "adn_debug_lstm_backward_family_counts": (C.c_int, [C.POINTER(C.c_int64)]),
    "adn_debug_plan_lstm_launches": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int]),
    "adn_profile_enable": (C.c_int, [_P, C.c_int]),
    "adn_profile_read": (C.c_int, [_P, C.POINTER(ProfileEntry), C.c_int, C.POINTER(C.c_int)]),
    "adn_op_gemm": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P, C.c_int, _P, C.c_int, _P,
                              C.c_int, C.c_int, _P]),
    "adn_op_gemm_ex": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P, C.c_int, _P, C.c_int, _P,
                                 C.c_int, C.c_int, C.c_int, _P]),
    "adn_op_gemm_shadow": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P, C.c_int, _P, C.c_int, _P, _P, _P,
                                     C.c_int, _P]),
    "adn_op_to_bf16": (C.c_int, [_P, _P, C.c_int64, _P]),
    "adn_op_delta_forward": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "adn_op_delta_backward": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "adn_op_adam": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_float, _P]),
    "adn_op_copy_bench": (C.c_int, [_P, _P, C.c_int64, C.c_int, _P, C.POINTER(C.c_float)]),
    "adn_set_dropout_state": (C.c_int, [_P, C.c_uint32, C.c_uint32]),
    "adn_apply_sgd": (C.c_int, [_P, C.c_float, C.c_float, C.c_int]),
    "adn_apply_adadelta": (C.c_int, [_P, C.c_float, C.c_float, C.c_float]),
    "adn_cae_create": (C.c_int, [_P, _P]),
    "adn_cae_destroy": (None, [_P]),
    "adn_cae_set_stream": (C.c_int, [_P, _P]),
    "adn_cae_num_params": (C.c_int, [_P]),
    "adn_cae_param_info": (C.c_int, [_P, C.c_int, _P]),
    "adn_cae_read_tensor": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "adn_cae_write_tensor": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "adn_cae_flat_buffer": (C.c_int, [_P, C.c_int, _P, _P]),
    "adn_cae_forward": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P]),
    "adn_cae_loss": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P]),
    "adn_cae_compute_grads": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P]),
    "adn_cae_apply_adadelta": (C.c_int, [_P, C.c_float, C.c_float, C.c_float]),
    "adn_cae_apply_adam": (C.c_int, [_P, C.c_float]),
    "adn_cae_synchronize": (C.c_int, [_P]),
    "adn_cae_set_dropout_state": (C.c_int, [_P, C.c_uint32, C.c_uint32]),
    "adn_rbm_create": (C.c_int, [_P, _P]),
    "adn_rbm_destroy": (None, [_P]),
    "adn_rbm_set_stream": (C.c_int, [_P, _P]),
    "adn_rbm_read": (C.c_int, [_P, C.c_int, _P]),
    "adn_rbm_write": (C.c_int, [_P, C.c_int, _P]),
    "adn_rbm_up": (C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    "adn_rbm_train_batch": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_float, C.c_uint32, C.c_uint32, _P]),
    "adn_prep_seq_deltas": (C.c_int, [_P, C.c_int, _P, C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "adn_prep_diff_images": (C.c_int, [_P, _P, C.c_int, _P, _P, C.c_int, C.c_int, _P]),
    "adn_prep_mean_image_subtraction": (C.c_int, [_P, _P, C.c_int, _P, _P, C.c_int, C.c_int, _P]),
    "adn_prep_normalize_rows": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P]),
    "adn_prep_column_stats": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P]),
    "adn_prep_apply_column_norm": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "adn_prep_gather_columns": (C.c_int, [_P, C.c_int, _P, C.c_int, _P, C.c_int, C.c_int, _P]),
    "adn_prep_lcn": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_float, _P]),
    "adn_batch_gather": (C.c_int, [_P, C.c_int, _P, _P, _P, C.c_int, _P, C.c_int, C.c_int, _P, _P, _P, _P]),
}
EXPORTED_SYMBOLS = tuple(sorted(_SIGNATURES))

_lib = None


def _preload_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (same soname as /opt/rocm's).  Two HIP
    runtimes in one process cannot both own the GPU, and whichever is loaded second sees no device.
    Loading torch's copy first makes the dynamic loader bind libadenet_hip.so to it (soname match), so
    this library, torch tensors and torch.distributed (RCCL) all share ONE runtime no matter which of
    them is imported first.  Without torch installed the system runtime is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """Load (once) and return the ctypes handle; raises AdenetError when the HIP library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AdenetError(
            "HIP extension not built: %s is missing. Build it with `python -m ip_avsr_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the compute path." % LIB_PATH)
    _preload_torch_hip_runtime()
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:
        raise AdenetError("cannot load %s: %s" % (LIB_PATH, e))
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise AdenetError("%s does not export %s (stale build?)" % (LIB_PATH, name))
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status):
    if status != ADN_OK:
        msg = load().adn_last_error()
        raise AdenetError("libadenet_hip error %d: %s" % (status, msg.decode() if msg else "?"))
