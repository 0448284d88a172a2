"""ctypes binding of libyolo_fastest_hip.so (C ABI: include/yolo_fastest_hip.h).

The product path has NO fallback: if the HIP library is missing or a call fails, this raises.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libyolo_fastest_hip.so")
ABI_VERSION = 1

YF_OK, YF_E_INVALID, YF_E_BLOB, YF_E_HIP, YF_E_WORKSPACE, YF_E_NOPROBE = 0, -1, -2, -3, -4, -5

_c = ctypes
_SIGS = {
    "yf_abi_version": (_c.c_int, []),
    "yf_last_error_string": (_c.c_char_p, []),
    "yf_create": (_c.c_int, [_c.c_void_p, _c.c_size_t, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.POINTER(_c.c_void_p)]),
    "yf_create_ex": (_c.c_int, [_c.c_void_p, _c.c_size_t, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.POINTER(_c.c_void_p)]),
    "yf_f32_to_f16_bits": (_c.c_uint16, [_c.c_float]),
    "yf_destroy": (_c.c_int, [_c.c_void_p]),
    "yf_io_params": (_c.c_int, [_c.c_void_p] + [_c.POINTER(_c.c_int)] * 4),
    "yf_workspace_bytes": (_c.c_int, [_c.c_void_p, _c.c_int, _c.POINTER(_c.c_size_t)]),
    "yf_forward": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_size_t,
                              _c.c_void_p]),
    "yf_forward_probe": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_char_p, _c.c_void_p, _c.c_size_t,
                                    _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "yf_decode_nms": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_double, _c.c_double,
                                 _c.POINTER(_c.c_double), _c.c_int, _c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p,
                                 _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    "yf_decode_nms_packed": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_double, _c.c_double, _c.POINTER(_c.c_double), _c.c_int,
                                        _c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p]),
    "yf_detect_packed": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_double, _c.c_double, _c.POINTER(_c.c_double), _c.c_int, _c.c_int,
                                    _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "yf_detect": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_double, _c.c_double, _c.POINTER(_c.c_double),
                             _c.c_int, _c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                             _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "yf_preprocess_u8": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p]),
    "yf_nms_sorted": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_double, _c.c_void_p, _c.c_void_p]),
    "yf_forward_u8": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                                 _c.c_size_t, _c.c_void_p]),
    "yf_val_decode_head": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.POINTER(_c.c_double), _c.c_int, _c.c_int,
                                      _c.c_void_p, _c.c_void_p]),
    "yf_val_nms": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_double, _c.c_double, _c.c_int, _c.c_void_p, _c.c_void_p,
                              _c.c_void_p]),
    "yf_val_nms_ex": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_double, _c.c_double, _c.c_int, _c.c_void_p,
                                 _c.c_void_p, _c.c_void_p]),
    "yf_train_loss_workspace_bytes": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.POINTER(_c.c_size_t)]),
    "yf_train_loss": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.POINTER(_c.c_double), _c.c_void_p, _c.c_int,
                                 _c.c_double, _c.c_void_p, _c.c_size_t, _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    "yf_train_head_loss_workspace_bytes": (_c.c_int, [_c.c_int, _c.c_int, _c.c_int, _c.POINTER(_c.c_size_t)]),
    "yf_train_head_loss": (_c.c_int, [_c.c_int, _c.c_int, _c.c_int, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.POINTER(_c.c_double),
                                      _c.c_void_p, _c.c_int, _c.c_double, _c.c_void_p, _c.c_size_t, _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    "yf_train_head_loss_workspace_bytes_ex": (_c.c_int, [_c.c_int] * 5 + [_c.POINTER(_c.c_size_t)]),
    "yf_train_head_loss_ex": (_c.c_int, [_c.c_int, _c.c_int, _c.c_int, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.POINTER(_c.c_double),
                                         _c.c_int, _c.c_int, _c.c_void_p, _c.c_int, _c.c_double, _c.c_void_p, _c.c_size_t, _c.c_void_p,
                                         _c.c_void_p, _c.c_void_p]),
    "yf_train_conv_forward": (_c.c_int, [_c.c_int, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p] + [_c.c_int] * 8 + [_c.c_void_p]),
    "yf_train_conv_backward_data": (_c.c_int, [_c.c_int, _c.c_void_p, _c.c_void_p, _c.c_void_p] + [_c.c_int] * 8 + [_c.c_void_p]),
    "yf_train_conv_backward_weight": (_c.c_int, [_c.c_int, _c.c_void_p, _c.c_void_p, _c.c_void_p] + [_c.c_int] * 8 +
                                      [_c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "yf_train_deconv_forward": (_c.c_int, [_c.c_int, _c.c_void_p, _c.c_void_p, _c.c_void_p] + [_c.c_int] * 5 + [_c.c_void_p]),
    "yf_train_deconv_backward_data": (_c.c_int, [_c.c_int, _c.c_void_p, _c.c_void_p, _c.c_void_p] + [_c.c_int] * 5 + [_c.c_void_p]),
    "yf_train_deconv_backward_weight": (_c.c_int, [_c.c_int, _c.c_void_p, _c.c_void_p, _c.c_void_p] + [_c.c_int] * 5 +
                                        [_c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "yf_train_unit_forward": (_c.c_int, [_c.c_int, _c.c_int] + [_c.c_void_p] * 9 + [_c.c_int] * 9 + [_c.c_void_p, _c.c_void_p]),
    "yf_train_unit_backward": (_c.c_int, [_c.c_int, _c.c_int] + [_c.c_void_p] * 12 + [_c.c_int] * 9 + [_c.c_void_p, _c.c_size_t, _c.c_void_p]),  # x z gy stats w gamma beta dgamma dbeta gz dw dx
    "yf_trainer_create": (_c.c_int, [_c.c_int, _c.c_int, _c.c_int, _c.POINTER(_c.c_void_p)]),
    "yf_trainer_create_ex": (_c.c_int, [_c.c_int] * 5 + [_c.POINTER(_c.c_void_p)]),
    "yf_trainer_destroy": (None, [_c.c_void_p]),
    "yf_trainer_num_params": (_c.c_int, [_c.c_void_p, _c.POINTER(_c.c_int), _c.POINTER(_c.c_int)]),
    "yf_trainer_workspace_bytes": (_c.c_int, [_c.c_void_p, _c.c_int, _c.POINTER(_c.c_size_t)]),
    "yf_trainer_forward": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                                      _c.c_size_t, _c.c_void_p]),
    "yf_trainer_backward": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                                       _c.c_size_t, _c.c_void_p]),
    "yf_trainer_graph_replays": (_c.c_int, [_c.c_void_p, _c.POINTER(_c.c_long), _c.POINTER(_c.c_long)]),
    "yf_trainer_graph_stats": (_c.c_int, [_c.c_void_p, _c.POINTER(_c.c_long)]),
    "yf_trainer_set_graphs": (_c.c_int, [_c.c_void_p, _c.c_int]),
    "yf_train_scratch_bytes": (_c.c_int, [_c.POINTER(_c.c_size_t)]),
    "yf_train_bn_forward": (_c.c_int, [_c.c_int] + [_c.c_void_p] * 7 + [_c.c_int, _c.c_int, _c.c_long, _c.c_int, _c.c_void_p, _c.c_void_p]),
    "yf_train_bn_backward": (_c.c_int, [_c.c_int] + [_c.c_void_p] * 8 + [_c.c_int, _c.c_int, _c.c_long, _c.c_int, _c.c_void_p, _c.c_void_p]),
    "yf_train_adam_multi": (_c.c_int, [_c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_double, _c.c_double,
                                       _c.c_double, _c.c_double, _c.c_int, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "yf_train_adam_multi_pinned": (_c.c_int, [_c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_double,
                                              _c.c_double, _c.c_double, _c.c_double, _c.c_int, _c.c_void_p, _c.c_size_t, _c.c_void_p, _c.c_int,
                                              _c.c_void_p]),
    "yf_train_channel_sum": (_c.c_int, [_c.c_int, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_long, _c.c_void_p]),
    "yf_train_channel_sum_split": (_c.c_int, [_c.c_int, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_long, _c.c_void_p, _c.c_void_p]),
    "yf_train_add": (_c.c_int, [_c.c_int, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_long, _c.c_void_p]),
    "yf_train_channel_slice": (_c.c_int, [_c.c_int, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_long, _c.c_int, _c.c_int, _c.c_int, _c.c_int,
                                          _c.c_void_p]),
    "yf_train_adam_step": (_c.c_int, [_c.c_int, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_long, _c.c_double, _c.c_double, _c.c_double,
                                      _c.c_double, _c.c_int, _c.c_void_p]),
    "yf_num_launches": (_c.c_int, [_c.c_void_p, _c.POINTER(_c.c_int)]),
    "yf_op_info": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_char_p, _c.c_int, _c.POINTER(_c.c_double), _c.POINTER(_c.c_double)]),
    "yf_op_info_ex": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_char_p, _c.c_int, _c.POINTER(_c.c_double), _c.POINTER(_c.c_double),
                                 _c.POINTER(_c.c_double)]),
    "yf_op_dtype": (_c.c_int, [_c.c_void_p, _c.c_int, _c.POINTER(_c.c_int)]),
    "yf_profile_forward": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_size_t, _c.c_void_p,
                                      _c.POINTER(_c.c_float), _c.c_int]),
    "yf_profile_forward_u8": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p, _c.c_size_t, _c.c_void_p,
                                         _c.POINTER(_c.c_float), _c.c_int]),
    "yf_set_chunk": (_c.c_int, [_c.c_void_p, _c.c_int]),
    "yf_set_fusion": (_c.c_int, [_c.c_void_p, _c.c_int]),
    "yf_streams_overlap": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.POINTER(_c.c_int)]),
    "yf_set_branches": (_c.c_int, [_c.c_void_p, _c.c_int]),
    "yf_set_lanes": (_c.c_int, [_c.c_void_p, _c.c_int]),
    "yf_set_split_sums": (_c.c_int, [_c.c_void_p, _c.c_int]),
    "yf_set_post_split": (_c.c_int, [_c.c_void_p, _c.c_int]),
    "yf_set_profile_repeats": (_c.c_int, [_c.c_void_p, _c.c_int]),
    "yf_cv_preprocess_u8": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p]),
    "yf_forward_bgr_u8": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                                     _c.c_size_t, _c.c_void_p]),
    "yf_op_dispatches": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_int, _c.POINTER(_c.c_int)]),
    "yf_profile_head_offsets": (_c.c_int, [_c.c_void_p, _c.c_int, _c.POINTER(_c.c_size_t), _c.POINTER(_c.c_size_t)]),
}
EXPORTS = tuple(_SIGS)
_lib = None


def build(verbose=False):
    """Compile csrc/*.hip for gfx950 into libyolo_fastest_hip.so (hipcc cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "-j4"] + ([] if verbose else ["-s"]))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: the HIP extension is the only implementation of this package "
                "(no CPU fallback). Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C yolo-fastest-and-embedded-deployment_amd/csrc`.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
            fn.restype, fn.argtypes = res, args
        if L.yf_abi_version() != ABI_VERSION:
            raise ImportError("libyolo_fastest_hip.so ABI version mismatch")
        _lib = L
    return _lib


class YFError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        msg = lib().yf_last_error_string().decode(errors="replace")
        raise YFError(f"yolo_fastest_hip error {rc}: {msg}")
