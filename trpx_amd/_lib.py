"""ctypes binding of libtrpx_hip.so (the C ABI declared in include/trpx_hip.h).

There is NO CPU fallback: if the HIP library is missing or cannot be loaded, importing any
compute entry point raises.  Build it with ``python -c 'import __graft_entry__ as g; g.build()'``
or ``make -C trpx_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TRPX_LIB") or os.path.join(_HERE, "libtrpx_hip.so")   # TRPX_LIB: kernel experiments only

OK, ERR_INVALID_ARG, ERR_UNSUPPORTED, ERR_CAPACITY, ERR_HIP, ERR_CORRUPT, ERR_NO_DEVICE, ERR_TIMEOUT = range(8)
U8, I8, U16, I16, U32, I32, F32, F64, U64, I64 = range(10)
STATUS_WORDS = 8
ABI_VERSION = 3          # TRPX_ABI_VERSION of include/trpx_hip.h (tests/test_abi.py compares them)


class TrpxError(RuntimeError):
    def __init__(self, code: int, text: str):
        super().__init__(f"trpx error {code}: {text}")
        self.code = code


class trpx_header(C.Structure):
    _fields_ = [("prolix_bits", C.c_uint), ("is_signed", C.c_int), ("block", C.c_uint),
                ("memory_size", C.c_uint64), ("number_of_values", C.c_uint64),
                ("number_of_frames", C.c_uint64), ("n_dims", C.c_uint), ("dims", C.c_uint64 * 8)]


# every symbol include/trpx_hip.h declares: (restype, argtypes)
_P, _SZ, _U, _I, _U64 = C.c_void_p, C.c_size_t, C.c_uint, C.c_int, C.c_uint64
SYMBOLS = {
    "trpx_abi_version": (_I, []),
    "trpx_last_error_string": (C.c_char_p, []),
    "trpx_dtype_size": (_SZ, [_I]),
    "trpx_dtype_is_signed": (_I, [_I]),
    "trpx_device_count": (_I, []),
    "trpx_worst_case_bytes": (_SZ, [_I, _SZ, _U]),
    "trpx_encode_workspace_bytes": (_SZ, [_I, _SZ, _SZ, _U]),
    "trpx_decode_workspace_bytes": (_SZ, [_I, _SZ, _SZ, _U]),
    "trpx_decode_parts_per_frame": (_U, [_I, _SZ, _SZ, _U]),
    "trpx_encode": (_I, [_I, _P, _SZ, _SZ, _U, _P, _SZ, _P, _P, _P, _SZ, _P]),
    "trpx_decode": (_I, [_I, _I, _P, _SZ, _P, _SZ, _SZ, _U, _P, _P, _P, _SZ, _P]),
    "trpx_index_bytes": (_SZ, [_I, _SZ, _SZ, _U]),
    "trpx_encode_indexed": (_I, [_I, _P, _SZ, _SZ, _U, _P, _SZ, _P, _P, _P, _P, _SZ, _P]),
    "trpx_encode_checked": (_I, [_I, _P, _SZ, _SZ, _U, _P, _SZ, _P, _P, _P, _P, _SZ, _P, _P]),
    "trpx_build_index": (_I, [_I, _P, _SZ, _P, _SZ, _SZ, _U, _P, _P, _P]),
    "trpx_decode_indexed": (_I, [_I, _I, _P, _SZ, _P, _P, _SZ, _SZ, _U, _P, _P, _P]),
    "trpx_decode_convert": (_I, [_I, _I, _P, _SZ, _P, _SZ, _SZ, _U, _P, _P, _P, _SZ, _P]),
    "trpx_encode_host": (_I, [_I, _P, _SZ, _SZ, _U, _P, _SZ, C.POINTER(_SZ), _P, C.POINTER(_U), _I]),
    "trpx_decode_host": (_I, [_I, _I, _P, _SZ, _P, _SZ, _SZ, _U, _P, _I]),
    "trpx_frame_offsets_host": (_I, [_P, _SZ, _SZ, _SZ, _U, _U, _P, _I]),
    "trpx_gather_workspace_bytes": (_SZ, [_SZ, _I]),
    "trpx_gather_frame_offsets": (_I, [_P, _P, _SZ, _SZ, _P, _P, _P, _P, _P, _SZ, _P]),
    "trpx_encode_sharded_workspace_bytes": (_SZ, [_I, _SZ, _SZ, _SZ, _U, _I]),
    "trpx_encode_sharded": (_I, [_P, _I, _P, _SZ, _SZ, _SZ, _U, _P, _SZ, _P, _P, _P, _P, _P, _P, _SZ, _P, _P]),
    "trpx_decode_sharded_workspace_bytes": (_SZ, [_I, _SZ, _SZ, _U]),
    "trpx_decode_sharded": (_I, [_I, _I, _P, _SZ, _P, _SZ, _SZ, _SZ, _U, _P, _P, _P, _SZ, _P]),
    "trpx_gather_pack": (_I, [_P, _SZ, _SZ, _P, _P, _P]),
    "trpx_gather_scan": (_I, [_P, _I, _SZ, _P, _P, _P, _P]),
    "trpx_comm_unique_id": (_I, [_P]),
    "trpx_comm_init": (_I, [C.POINTER(_P), _I, _I, _P]),
    "trpx_comm_destroy": (_I, [_P]),
    "trpx_comm_info": (_I, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "trpx_shard_last_error": (C.c_char_p, []),
    "trpx_stack_open": (_I, [C.POINTER(_P), _I, _P, _SZ, _P, _P, _SZ, _SZ, _U, _U, _I]),
    "trpx_group_count": (_SZ, [_SZ, _U]),
    "trpx_index_group_states": (_I, [_P, _SZ, _SZ, _U, _P, _P]),
    "trpx_index_from_group_states": (_I, [_I, _P, _SZ, _P, _P, _SZ, _SZ, _U, _P, _P, _P]),
    "trpx_group_states_host": (_I, [_P, _SZ, _P, _SZ, _SZ, _U, _U, _P, _I]),
    "trpx_decode_host_grouped": (_I, [_I, _I, _P, _SZ, _P, _P, _SZ, _SZ, _U, _P, _I]),
    "trpx_header_format_grouped": (_SZ, [C.POINTER(trpx_header), C.c_void_p, _SZ, C.c_void_p, _SZ, C.c_char_p, _SZ]),
    "trpx_header_group_states": (_SZ, [C.c_char_p, _SZ, C.c_void_p, _SZ]),
    "trpx_stack_read": (_I, [_P, _SZ, _I, _P]),
    "trpx_stack_close": (None, [_P]),
    "trpx_host_release": (None, []),
    "trpx_set_encode_path": (_I, [_I]),
    "trpx_set_decode_path": (_I, [_I]),
    "trpx_profile_enable": (_I, [_I]),
    "trpx_profile_read": (_I, [C.POINTER(C.c_float), _I]),
    "trpx_bench_stream": (_I, [_I, _P, _P, _SZ, _P]),
    "trpx_workspace_invalidate": (_I, [_P, _SZ]),
    "trpx_synth_fill": (_I, [_I, _U64, _U64, _SZ, _SZ, _P, _P]),
    "trpx_header_format": (_SZ, [C.POINTER(trpx_header), C.c_char_p, _SZ]),
    "trpx_header_parse": (_I, [C.c_char_p, _SZ, C.POINTER(trpx_header), C.POINTER(_SZ)]),
    "trpx_header_format_indexed": (_SZ, [C.POINTER(trpx_header), C.c_void_p, _SZ, C.c_char_p, _SZ]),
    "trpx_header_frame_sizes": (_SZ, [C.c_char_p, _SZ, C.c_void_p, _SZ]),
}

_lib = None


def lib() -> C.CDLL:
    """Load libtrpx_hip.so; raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: the HIP extension has not been built "
                              "(run __graft_entry__.build()); trpx_amd has no CPU fallback")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)      # AttributeError if the ABI is incomplete
            fn.restype = res
            fn.argtypes = args
        if L.trpx_abi_version() != ABI_VERSION:
            raise ImportError(f"libtrpx_hip.so ABI version {L.trpx_abi_version()} != {ABI_VERSION} (include/trpx_hip.h: TRPX_ABI_VERSION)")
        _lib = L
    return _lib


def check(rc: int) -> None:
    if rc != OK:
        raise TrpxError(rc, lib().trpx_last_error_string().decode(errors="replace"))
