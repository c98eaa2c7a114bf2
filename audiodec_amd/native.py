"""ctypes binding of libaudiodec_hip.so (C ABI: include/audiodec_hip.h).

There is no CPU fallback: if the shared library is missing or does not export the ABI this module
raises, and so does everything that computes.  The library is built in-tree by
``__graft_entry__.build()`` (hipcc --offload-arch=gfx950).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ADK_LIB_PATH") or os.path.join(_HERE, "libaudiodec_hip.so")   # override: tuning builds only
ABI_VERSION = 14

ADK_OK = 0
ACT_NONE, ACT_ELU, ACT_LEAKY, ACT_TANH = 0, 1, 2, 3
IMPL_AUTO, IMPL_DIRECT, IMPL_MFMA, IMPL_MFMA_ROWS, IMPL_SPLIT16, IMPL_SPLIT16_ROWS, IMPL_SPLIT16_SK, IMPL_SPLIT16_UP = 0, 1, 2, 3, 4, 5, 6, 7
OP_CONV, OP_RING_WRITE, OP_MEAN, OP_HIST_REPLICATE = 0, 1, 2, 3
STEP_REPLAY = 1
POST_SLOTS = 32


class RingView(C.Structure):
    _fields_ = [("base", C.c_void_p), ("rows", C.c_int32), ("channels", C.c_int32),
                ("cursor", C.c_int32), ("ch_off", C.c_int32)]


class ConvDesc(C.Structure):
    _fields_ = [("cin_g", C.c_int32), ("cout_g", C.c_int32), ("groups", C.c_int32),
                ("taps", C.c_int32), ("stride", C.c_int32), ("dilation", C.c_int32),
                ("hist", C.c_int32), ("up", C.c_int32), ("cout_real", C.c_int32),
                ("in_group_stride", C.c_int32), ("res_group_stride", C.c_int32),
                ("act_in", C.c_int32), ("act_in_slope", C.c_float), ("act_out", C.c_int32),
                ("w", C.c_void_p), ("w_frag", C.c_void_p), ("bias", C.c_void_p)]


class RingDesc(C.Structure):
    _fields_ = [("channels", C.c_int32), ("hist", C.c_int32), ("rate", C.c_int32),
                ("external", C.c_int32), ("arena_off", C.c_int64), ("extra_rows", C.c_int32), ("reserved_", C.c_int32)]


class OpDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("in_ring", C.c_int32), ("out_ring", C.c_int32), ("res_ring", C.c_int32),
                ("in_ch_off", C.c_int32), ("out_ch_off", C.c_int32), ("res_ch_off", C.c_int32),
                ("rate_out", C.c_int32), ("conv", ConvDesc), ("w_off", C.c_int64), ("wf_off", C.c_int64), ("b_off", C.c_int64),
                ("mean_off", C.c_int64), ("scale_off", C.c_int64), ("ext_src", C.c_int32),
                ("mean_rings", C.c_int32 * 4), ("n_mean", C.c_int32), ("impl", C.c_int32), ("fuse_next", C.c_int32), ("chain", C.c_int32),
                ("in_shadow", C.c_int32), ("out_shadow", C.c_int32), ("shadow_act", C.c_int32), ("shadow_slope", C.c_float)]


# every symbol include/audiodec_hip.h declares: (restype, argtypes)
_vp, _i32, _i64 = C.c_void_p, C.c_int32, C.c_int64
SYMBOLS = {
    "adk_last_error": (C.c_char_p, []),
    "adk_abi_version": (C.c_int, []),
    "adk_debug_flags": (C.c_int, [C.POINTER(_i32)]),
    "adk_set_conv_cfg": (C.c_int, [_i32]),
    "adk_set_option": (C.c_int, [C.c_char_p, _i32]),
    "adk_streamk_plan": (C.c_int, [C.c_int64, _i32, _i32, C.POINTER(C.c_int32)]),
    "adk_streamk_range_start": (C.c_int64, [C.c_int64, _i32, C.POINTER(C.c_int32), _i32]),
    "adk_causal_conv": (C.c_int, [C.POINTER(ConvDesc), RingView, RingView, RingView, _i32, _i32, _i32, _vp]),
    "adk_causal_conv_describe": (C.c_int, [C.POINTER(ConvDesc), RingView, RingView, RingView, _i32, _i32, _i32, C.c_char_p, _i32]),
    "adk_causal_conv_time": (C.c_int, [C.POINTER(ConvDesc), RingView, RingView, RingView, _i32, _i32, _i32, _i32, _vp, C.POINTER(C.c_float)]),
    "adk_packed_weight_floats": (C.c_int64, [_i32, _i32, _i32]),
    "adk_pack_weights_mfma": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp]),
    "adk_ring_write": (C.c_int, [_vp, RingView, _vp, _vp, _i32, _i32, _vp]),
    "adk_rvq_encode": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "adk_rvq_lookup": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "adk_packed_weight_floats_split16": (C.c_int64, [_i32, _i32, _i32]),
    "adk_pack_weights_split16": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp]),
    "adk_codes_frame_bytes": (C.c_int32, [_i32, _i32]),
    "adk_codes_pack": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "adk_codes_unpack": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "adk_codes_lookup": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "adk_program_create": (C.c_int, [C.POINTER(OpDesc), _i32, C.POINTER(RingDesc), _i32, _i32, _i32, _vp, _i64,
                                     _vp, _i64, C.POINTER(_vp)]),
    "adk_program_destroy": (None, [_vp]),
    "adk_program_step": (C.c_int, [_vp, _i32, C.POINTER(_vp), _i32, _vp]),
    "adk_program_step_ex": (C.c_int, [_vp, _i32, C.POINTER(_vp), _i32, _vp, _i32]),
    "adk_program_reset": (C.c_int, [_vp, _vp]),
    "adk_program_flags": (C.c_int, [_vp, _vp, C.POINTER(_i32)]),
    "adk_program_rewind": (C.c_int, [_vp, _i32]),
    "adk_program_flags_post": (C.c_int, [_vp, _vp, C.POINTER(_i64)]),
    "adk_program_flags_poll": (C.c_int, [_vp, _i64, _i32, C.POINTER(_i32), C.POINTER(_i32)]),
    "adk_program_get_fresh": (C.c_int, [_vp]),
    "adk_program_set_fresh": (C.c_int, [_vp, _i32]),
    "adk_program_get_cursors": (C.c_int, [_vp, C.POINTER(_i32), _i32]),
    "adk_program_set_cursors": (C.c_int, [_vp, C.POINTER(_i32), _i32]),
    "adk_program_describe_op": (C.c_int, [_vp, _i32, _i32, C.c_char_p, _i32]),
    "adk_program_set_profiling": (C.c_int, [_vp, _i32]),
    "adk_program_set_workgroups": (C.c_int, [_vp, _i32]),
    "adk_program_last_op_ms": (C.c_int, [_vp, C.POINTER(C.c_float), _i32]),
    "adk_program_set_graph": (C.c_int, [_vp, _i32]),
    "adk_program_graph_stats": (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i32)]),
}

_lib = None


class NativeError(RuntimeError):
    pass


def lib():
    """Load (once) and return the shared library; raises NativeError if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError(
            f"{LIB_PATH} is missing: the AudioDec HIP kernels are not built. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback.")
    import torch  # noqa: F401  -- loads the HIP runtime (libamdhip64.so.7) this library binds to
    l = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(l, name)
        except AttributeError as e:
            raise NativeError(f"{LIB_PATH} does not export {name}; rebuild it") from e
        fn.restype = res
        fn.argtypes = args
    if l.adk_abi_version() != ABI_VERSION:
        raise NativeError(f"{LIB_PATH}: ABI version {l.adk_abi_version()} != {ABI_VERSION}; rebuild it")
    _lib = l
    return l


def check(rc, what=""):
    if rc != ADK_OK:
        msg = lib().adk_last_error().decode(errors="replace")
        if rc in (-1, -2):
            raise ValueError(f"{what}: {msg} (adk error {rc})")
        raise NativeError(f"{what}: {msg} (adk error {rc})")


FLAG_BAD_INDEX, FLAG_STREAMK_TIMEOUT, FLAG_BAD_CODE, FLAG_F16_OVERFLOW = 1, 2, 4, 8


def set_option(name, value):
    """Process-wide tuning / test option of the library (adk_set_option in the header)."""
    check(lib().adk_set_option(name.encode(), int(value)), "adk_set_option")


def device_flags():
    """Read and clear the sticky device-side error flags (adk_debug_flags); synchronises the device(s)."""
    v = C.c_int32(0)
    check(lib().adk_debug_flags(C.byref(v)), "adk_debug_flags")
    return int(v.value)


def raise_on_device_flags(where=""):
    """Turn a device-side failure into the exception the reference's PyTorch path would have raised (or the closest
    one).  Called where the facade synchronises anyway (payload / waveform leaves the device, end of an utterance,
    streamer tick); kernels never stop a launch sequence, they record the failure in a sticky word."""
    raise_for_flags(device_flags(), where)


def raise_for_flags(v, where=""):
    """The exception for a flag word (adk_debug_flags / adk_program_flags bits); nothing for 0."""
    if not v:
        return
    pre = f"{where}: " if where else ""
    if v & FLAG_BAD_INDEX:
        raise IndexError(pre + "index out of range in self (a code index outside the codebook reached lookup; "
                               "F.embedding raises the same in the reference, layers/vq_module.py:160)")
    if v & FLAG_BAD_CODE:
        raise ValueError(pre + "an index that is not a code of its stage was packed (wire format)")
    if v & FLAG_F16_OVERFLOW:
        raise NativeError(pre + "a split-f16 conv produced non-finite values: an operand beyond the f16 range "
                                "(|v| > 65504) or non-finite input; results since the last check are invalid.  A generator with its "
                                "guard on (the default of AudioDec, streaming and offline programs alike) repeats such a step on the "
                                "exact-f32 kernels by itself and does not get here; this is the report for unguarded steps (guard=False / "
                                "ADK_GUARD=0: asynchronous pipelines) -- run those on the exact-f32 kernels: ADK_SPLIT16=0 / set_split16(False)")
    if v & FLAG_STREAMK_TIMEOUT:
        raise NativeError(pre + "a stream-K conv workgroup timed out waiting for a partial tile; results since the "
                                "last check are invalid")
    raise NativeError(pre + f"device error flags {v}")


def current_stream(device):
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_gpu(device):
    """The product path runs on a HIP device only."""
    import torch
    dev = torch.device(device)
    if dev.type != "cuda":
        raise NativeError(f"device {device!r}: the AudioDec HIP path needs a 'cuda:N' (HIP) device; "
                          "there is no CPU implementation in this package")
    if not torch.cuda.is_available():
        raise NativeError("no HIP device visible (torch.cuda.is_available() is False)")
    lib()
    return dev


_warned_cpu = False


def resolve_device(device):
    """Device string a facade object should live on.  The reference's constructors default to 'cpu' and its demos pass
    'cpu' unless --cuda is given (utils/audiodec.py:20-30, demoFile.py:32-37); this package computes on HIP devices only.
    So that such callers port unchanged, 'cpu' is mapped to the first HIP device with a one-time warning when one is
    visible; without a HIP device it is an error (there is no CPU fallback)."""
    import warnings
    import torch
    global _warned_cpu
    dev = torch.device(device)
    if dev.type != "cpu":
        return str(device)
    if not torch.cuda.is_available():
        raise NativeError(f"device {device!r}: the AudioDec HIP path needs a HIP device ('cuda:N') and none is visible; "
                          "there is no CPU implementation in this package")
    if not _warned_cpu:
        warnings.warn("audiodec_amd has no CPU compute path: device 'cpu' is mapped to 'cuda:0' (the first HIP device)", UserWarning, stacklevel=3)
        _warned_cpu = True
    return "cuda:0"
