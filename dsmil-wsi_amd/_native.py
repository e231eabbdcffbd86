"""ctypes binding of libdsmil_hip.so (include/dsmil_hip.h).  Thin: it passes ``data_ptr()``s,
sizes and the current HIP stream, and turns negative status codes into RuntimeError.

There is deliberately NO fallback: if the shared object is missing or lacks a symbol the import
of the HIP path fails loudly (``NativeLibraryError``)."""
import ctypes
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# DSMIL_NATIVE_LIB: file name of an alternative in-tree build next to this module (profiling /
# experiment builds made by build.py --variant, e.g. libdsmil_hip_expt.so); default = the product build
LIB_PATH = os.path.join(PKG_DIR, os.path.basename(os.environ.get("DSMIL_NATIVE_LIB", "libdsmil_hip.so")))

c_f32p = ctypes.c_void_p
c_i64p = ctypes.c_void_p


class NativeLibraryError(RuntimeError):
    pass


class AggParams(ctypes.Structure):
    """struct dsmil_agg_params (include/dsmil_hip.h)."""
    _fields_ = [("fc_w", ctypes.c_void_p), ("fc_b", ctypes.c_void_p),
                ("q0_w", ctypes.c_void_p), ("q0_b", ctypes.c_void_p),
                ("q2_w", ctypes.c_void_p), ("q2_b", ctypes.c_void_p),
                ("fcc_w", ctypes.c_void_p), ("fcc_b", ctypes.c_void_p),
                ("K", ctypes.c_int32), ("Kv", ctypes.c_int32),
                ("C", ctypes.c_int32), ("nonlinear", ctypes.c_int32)]


class AggOpts(ctypes.Structure):
    """struct dsmil_agg_opts (include/dsmil_hip.h)."""
    _fields_ = [("packed_split", ctypes.c_void_p), ("row_map", ctypes.c_void_p), ("packed_f2", ctypes.c_void_p)]


class AggGrads(ctypes.Structure):
    """struct dsmil_agg_grads (include/dsmil_hip.h)."""
    _fields_ = [(n, ctypes.c_void_p) for n in
                ("fc_w", "fc_b", "q0_w", "q0_b", "q2_w", "q2_b", "fcc_w", "fcc_b")]


class AdamState(ctypes.Structure):
    """struct dsmil_adam_state (include/dsmil_hip.h)."""
    _fields_ = [("exp_avg", ctypes.POINTER(ctypes.c_void_p)), ("exp_avg_sq", ctypes.POINTER(ctypes.c_void_p)),
                ("step", ctypes.c_int64), ("lr", ctypes.c_double), ("beta1", ctypes.c_double), ("beta2", ctypes.c_double),
                ("eps", ctypes.c_double), ("weight_decay", ctypes.c_double)]


# symbol -> (restype, argtypes); must list every function include/dsmil_hip.h declares
SIGNATURES = {
    "dsmil_abi_version": (ctypes.c_int, []),
    "dsmil_strerror": (ctypes.c_char_p, [ctypes.c_int]),
    "dsmil_agg_mlp_form": (ctypes.c_int, []),
    "dsmil_agg_inline_query": (ctypes.c_int, [ctypes.c_int]),
    "dsmil_agg_batch_form": (ctypes.c_int, [ctypes.c_int]),
    "dsmil_agg_persistent_grid": (ctypes.c_int, [ctypes.c_int]),
    "dsmil_agg_logits_form": (ctypes.c_int, [ctypes.c_int]),
    "dsmil_device_cus": (ctypes.c_int, []),
    "dsmil_agg_packed_f2_bytes": (ctypes.c_size_t, [ctypes.c_int32]),
    "dsmil_agg_pack_f2": (ctypes.c_int, [c_f32p, c_f32p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]),
    "dsmil_agg_packed_split_bytes": (ctypes.c_size_t, [ctypes.c_int32, ctypes.c_int32]),
    "dsmil_agg_pack_split": (ctypes.c_int, [c_f32p, c_f32p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]),
    "dsmil_agg_forward_ex": (ctypes.c_int, [c_f32p, c_f32p, c_i64p, ctypes.c_int32, ctypes.c_int64,
                                            ctypes.c_int64, ctypes.POINTER(AggParams), ctypes.POINTER(AggOpts), c_f32p,
                                            c_f32p, c_f32p, c_f32p, c_f32p, c_i64p, ctypes.c_void_p,
                                            ctypes.c_size_t, ctypes.c_void_p]),
    "dsmil_agg_loss_head": (ctypes.c_int, [c_f32p, c_f32p, c_i64p, c_f32p, ctypes.c_int32, c_f32p, c_f32p, c_f32p,
                                           c_f32p, ctypes.c_void_p]),
    "dsmil_agg_backward_ex": (ctypes.c_int, [c_f32p, c_f32p, ctypes.c_int64, ctypes.POINTER(AggParams), c_f32p,
                                             c_f32p, c_i64p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                             ctypes.POINTER(AggGrads), c_f32p, c_i64p, ctypes.c_void_p, ctypes.c_size_t,
                                             ctypes.c_void_p]),
    "dsmil_agg_shard_argmax": (ctypes.c_int, [c_f32p, ctypes.c_int64, ctypes.POINTER(AggParams), c_f32p, c_f32p,
                                              c_i64p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "dsmil_agg_shard_attend": (ctypes.c_int, [c_f32p, c_f32p, ctypes.c_int64, ctypes.POINTER(AggParams), c_f32p,
                                              c_f32p, c_f32p, c_f32p, ctypes.c_void_p, ctypes.c_size_t,
                                              ctypes.c_void_p]),
    "dsmil_agg_tile_rows": (ctypes.c_int, [ctypes.c_int32, ctypes.c_int64]),
    "dsmil_agg_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int32, ctypes.c_int64, ctypes.c_int32,
                                                    ctypes.c_int32, ctypes.c_int32]),
    "dsmil_agg_packed_bf16_bytes": (ctypes.c_size_t, [ctypes.c_int32]),
    "dsmil_agg_pack_bf16": (ctypes.c_int, [c_f32p, c_f32p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]),
    "dsmil_agg_forward_bf16": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, c_i64p, ctypes.c_int32,
                                              ctypes.c_int64, ctypes.c_int64, ctypes.POINTER(AggParams),
                                              ctypes.c_void_p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_i64p,
                                              ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "dsmil_profile_enable": (ctypes.c_int, [ctypes.c_int]),
    "dsmil_profile_collect": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64)]),
    "dsmil_resnet18_packed_bytes": (ctypes.c_size_t, []),
    "dsmil_resnet18_pack": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), c_f32p, ctypes.c_void_p]),
    "dsmil_resnet18_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
    "dsmil_resnet18in_forward": (ctypes.c_int, [c_f32p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                                c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_int32, c_f32p,
                                                c_f32p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "dsmil_resnet18in_forward_u8": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                                   c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_int32, c_f32p,
                                                   c_f32p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "dsmil_resnet18_norm_channels": (ctypes.c_int32, []),
    "dsmil_resnet18bn_forward": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                                ctypes.c_int32, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                                ctypes.c_int32, c_f32p, c_f32p, ctypes.c_void_p, ctypes.c_size_t,
                                                ctypes.c_void_p]),
    "dsmil_resnet_num_convs": (ctypes.c_int32, [ctypes.c_int32]),
    "dsmil_resnet_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
    "dsmil_resnet_feature_dim": (ctypes.c_int32, [ctypes.c_int32]),
    "dsmil_resnet_mfma_forms": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]),
    "dsmil_resnet_norm_channels": (ctypes.c_int32, [ctypes.c_int32]),
    "dsmil_resnet_packed_bytes": (ctypes.c_size_t, [ctypes.c_int32]),
    "dsmil_resnet_packed_bytes_ex": (ctypes.c_size_t, [ctypes.c_int32, ctypes.c_int32]),
    "dsmil_resnet_pack": (ctypes.c_int, [ctypes.c_int32, ctypes.POINTER(ctypes.c_void_p), c_f32p, ctypes.c_void_p]),
    "dsmil_resnet_forward": (ctypes.c_int, [ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                                            ctypes.c_int32, ctypes.c_int32, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                            c_f32p, ctypes.c_int32, c_f32p, c_f32p, ctypes.c_void_p,
                                            ctypes.c_size_t, ctypes.c_void_p]),
    "dsmil_resnet_pack_ex": (ctypes.c_int, [ctypes.c_int32, ctypes.POINTER(ctypes.c_void_p), c_f32p, ctypes.c_int32, ctypes.c_void_p]),
    "dsmil_resnet_forward_ex": (ctypes.c_int, [ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                                               ctypes.c_int32, ctypes.c_int32, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                               c_f32p, ctypes.c_int32, c_f32p, c_f32p, ctypes.c_void_p,
                                               ctypes.c_size_t, ctypes.c_int32, ctypes.c_void_p]),
    "dsmil_tile_stats": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p,
                                        ctypes.c_void_p]),
    "dsmil_jpeg_plan_bytes": (ctypes.c_size_t, [ctypes.c_int32]),
    "dsmil_jpeg_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int64]),
    "dsmil_jpeg_parse": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]),
    "dsmil_read_files": (ctypes.c_int64, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]),
    "dsmil_csv_parse_f32": (ctypes.c_int64, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64]),
    "dsmil_csv_format_f32": (ctypes.c_int64, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32,
                                              ctypes.c_void_p, ctypes.c_int64]),
    "dsmil_jpeg_decode": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "dsmil_fc_forward": (ctypes.c_int, [c_f32p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                        c_f32p, c_f32p, c_f32p, ctypes.c_void_p]),
    "dsmil_agg_backward_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                                             ctypes.c_int32]),
    "dsmil_agg_backward": (ctypes.c_int, [c_f32p, c_f32p, ctypes.c_int64, ctypes.POINTER(AggParams), c_f32p,
                                          c_f32p, c_i64p, c_f32p, c_f32p, c_f32p, c_f32p,
                                          ctypes.POINTER(AggGrads), c_f32p, ctypes.c_void_p, ctypes.c_size_t,
                                          ctypes.c_void_p]),
    "dsmil_agg_train_step_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
    "dsmil_agg_train_step": (ctypes.c_int, [c_f32p, ctypes.c_int64, c_i64p, c_f32p, ctypes.POINTER(AggParams),
                                            ctypes.POINTER(AdamState), c_f32p, ctypes.c_void_p, ctypes.c_size_t,
                                            ctypes.c_void_p]),
    "dsmil_adam_step": (ctypes.c_int, [ctypes.c_int32, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p),
                                       ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p),
                                       ctypes.POINTER(ctypes.c_int64), ctypes.c_int64, ctypes.c_double, ctypes.c_double,
                                       ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_void_p]),
    "dsmil_agg_forward": (ctypes.c_int, [c_f32p, c_f32p, c_i64p, ctypes.c_int32, ctypes.c_int64,
                                         ctypes.c_int64, ctypes.POINTER(AggParams), c_f32p, c_f32p,
                                         c_f32p, c_f32p, c_f32p, c_i64p, ctypes.c_void_p,
                                         ctypes.c_size_t, ctypes.c_void_p]),
}

DSMIL_E_INVALID = -1        # include/dsmil_hip.h
DSMIL_E_UNSUPPORTED = -2
DSMIL_E_WORKSPACE = -3

_lib = None


def lib():
    """The loaded shared library (loads on first use)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryError(
                f"{LIB_PATH} not found: build it with `python __graft_entry__.py` "
                f"(or dsmil-wsi_amd/build.py); the HIP path has no fallback")
        try:
            L = ctypes.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise NativeLibraryError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(L, name)
            except AttributeError as e:
                raise NativeLibraryError(f"{LIB_PATH} does not export {name}") from e
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(code, what):
    if code != 0:
        msg = lib().dsmil_strerror(code).decode()
        raise RuntimeError(f"{what} failed: {msg} (code {code})")
