"""ctypes binding of librtldavis_hip.so (include/rtldavis_hip.h).

There is no CPU fallback: if the shared library is missing this module raises at import
of the symbol table, and every compute call raises ``HipError`` when no MI355X is usable.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# RTLDAVIS_HIP_LIB overrides the path (A/B builds of the kernels); the default is the in-tree build
LIB_PATH = os.environ.get("RTLDAVIS_HIP_LIB") or os.path.join(HERE, "librtldavis_hip.so")

RD_MAX_PREAMBLE = 64
RD_MAX_PKT_BYTES = 32
RD_OK, RD_ERR_ARG, RD_ERR_DEVICE, RD_ERR_CAPACITY, RD_ERR_STATE = 0, -1, -2, -3, -4


class HipError(RuntimeError):
    """A HIP/device failure reported by librtldavis_hip (never swallowed, never retried on CPU)."""


class RdConfig(C.Structure):
    _fields_ = [("bit_rate", C.c_int32), ("symbol_length", C.c_int32), ("preamble_symbols", C.c_int32),
                ("packet_symbols", C.c_int32), ("block_size", C.c_int32), ("preamble", C.c_uint8 * RD_MAX_PREAMBLE)]


class RdPacket(C.Structure):
    _fields_ = [("stream", C.c_int32), ("call", C.c_int32), ("index", C.c_int32), ("nbytes", C.c_int32),
                ("data", C.c_uint8 * RD_MAX_PKT_BYTES), ("rssi", C.c_double), ("snr", C.c_double)]


class RdParsed(C.Structure):
    _fields_ = [("stream", C.c_int32), ("call", C.c_int32), ("index", C.c_int32), ("freq_err", C.c_int32),
                ("id", C.c_int32), ("nbytes", C.c_int32), ("data", C.c_uint8 * RD_MAX_PKT_BYTES),
                ("rssi", C.c_double), ("snr", C.c_double)]


class RdTiming(C.Structure):
    _fields_ = [("demod_ms", C.c_float), ("fixup_ms", C.c_float), ("search_ms", C.c_float),
                ("slice_ms", C.c_float), ("total_ms", C.c_float), ("runs", C.c_int32)]


class RdChanConfig(C.Structure):
    _fields_ = [("out_rate", C.c_int32), ("decim", C.c_int32), ("n_taps", C.c_int32), ("n_channels", C.c_int32),
                ("gain", C.c_double)]


# name -> (restype, argtypes); exactly the functions include/rtldavis_hip.h declares
_P = C.c_void_p
SIGNATURES = {
    "rd_last_error": (C.c_char_p, []),
    "rd_device_count": (C.c_int, []),
    "rd_set_device": (C.c_int, [C.c_int]),
    "rd_set_wait_timeout_ms": (C.c_int, [C.c_int]),
    "rd_set_input_push": (C.c_int, [C.c_int]),
    "rd_demod_input_mode": (C.c_int, [C.c_void_p]),
    "rd_create": (C.c_int, [C.POINTER(RdConfig), C.POINTER(_P)]),
    "rd_create_multi": (C.c_int, [C.POINTER(RdConfig), C.c_int, C.POINTER(_P)]),
    "rd_demod_blocks": (C.c_int, [_P, _P, C.c_size_t, C.POINTER(RdPacket), C.c_int, C.POINTER(C.c_int)]),
    "rd_copy_discriminated_stream": (C.c_int, [_P, C.c_int, _P, C.c_size_t]),
    "rd_demod_submit": (C.c_int, [_P, _P, C.c_size_t, C.c_int]),
    "rd_demod_register_input": (C.c_int, [_P, _P, C.c_size_t]),
    "rd_demod_submit_from": (C.c_int, [_P, C.c_size_t, C.c_size_t, C.c_int]),
    "rd_demod_fetch": (C.c_int, [_P, C.POINTER(RdPacket), C.c_int, C.POINTER(C.c_int)]),
    "rd_demod_refetch": (C.c_int, [_P, C.POINTER(RdPacket), C.c_int, C.POINTER(C.c_int)]),
    "rd_demod_inflight": (C.c_int, [_P]),
    "rd_destroy": (None, [_P]),
    "rd_reset": (C.c_int, [_P]),
    "rd_demod_block": (C.c_int, [_P, _P, C.c_size_t, C.c_int, C.POINTER(RdPacket), C.c_int, C.POINTER(C.c_int)]),
    "rd_copy_discriminated": (C.c_int, [_P, _P, C.c_size_t]),
    "rd_copy_filtered": (C.c_int, [_P, _P, C.c_size_t]),
    "rd_copy_quantized": (C.c_int, [_P, _P, C.c_size_t]),
    "rd_batch_create": (C.c_int, [C.POINTER(RdConfig), C.c_int, C.c_int, C.POINTER(_P)]),
    "rd_batch_destroy": (None, [_P]),
    "rd_batch_input_ptr": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "rd_batch_upload": (C.c_int, [_P, _P, C.c_size_t]),
    "rd_batch_upload_async": (C.c_int, [_P, _P, C.c_size_t, _P]),
    "rd_batch_run": (C.c_int, [_P, _P]),
    "rd_batch_results": (C.c_int, [_P, C.POINTER(RdPacket), C.c_int, C.POINTER(C.c_int)]),
    "rd_batch_copy_bits": (C.c_int, [_P, C.c_int, _P, C.c_size_t]),
    "rd_batch_copy_discriminated": (C.c_int, [_P, C.c_int, C.c_size_t, _P, C.c_size_t]),
    "rd_batch_set_parse": (C.c_int, [_P, C.c_int]),
    "rd_batch_parsed": (C.c_int, [_P, C.POINTER(RdParsed), C.c_int, C.POINTER(C.c_int)]),
    "rd_batch_set_timing": (C.c_int, [_P, C.c_int]),
    "rd_batch_set_pipelined": (C.c_int, [_P, C.c_int]),
    "rd_batch_last_run_forms": (C.c_int, [_P, C.POINTER(C.c_uint32)]),
    "rd_batch_get_timing": (C.c_int, [_P, C.POINTER(RdTiming)]),
    "rd_batch_get_counters": (C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "rd_lut_execute": (C.c_int, [_P, C.c_size_t, _P, C.c_size_t]),
    "rd_rotate_fs4": (C.c_int, [_P, _P, C.c_size_t]),
    "rd_fir9": (C.c_int, [_P, C.c_size_t, _P, C.c_size_t]),
    "rd_discriminate": (C.c_int, [_P, C.c_size_t, _P, C.c_size_t]),
    "rd_quantize": (C.c_int, [_P, _P, C.c_size_t]),
    "rd_search": (C.c_int, [C.POINTER(RdConfig), _P, C.c_size_t, _P, C.c_int, C.POINTER(C.c_int)]),
    "rd_chan_create": (C.c_int, [C.POINTER(RdChanConfig), _P, _P, C.POINTER(_P)]),
    "rd_chan_destroy": (None, [_P]),
    "rd_chan_upload": (C.c_int, [_P, _P, C.c_size_t]),
    "rd_chan_input_ptr": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    "rd_chan_run": (C.c_int, [_P, C.c_size_t, _P, C.c_size_t, _P]),
    "rd_chan_run_host": (C.c_int, [_P, C.c_size_t, _P, C.c_size_t]),
    "rd_debug_mfma_taps": (None, [_P]),
    "rd_debug_mfma_taps8": (None, [_P]),
    "rd_debug_mfma_taps8s": (None, [_P, _P]),
    "rd_debug_demod_mfma": (C.c_int, [_P, C.c_int, C.c_uint32, C.c_int, C.c_uint32, _P, _P, _P, C.c_uint32,
                                      C.POINTER(C.c_uint32)]),
}

_lib = None


def lib():
    """The loaded library.  Raises ImportError (loudly) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C rtldavis_amd/csrc`.  rtldavis_amd has no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def last_error() -> str:
    return (lib().rd_last_error() or b"").decode("utf-8", "replace")


def check(rc: int) -> None:
    """Map rd_status to the exceptions the reference raises (dsp.py:32-36,145-149 -> ValueError)."""
    if rc == RD_OK:
        return
    msg = last_error()
    if rc == RD_ERR_ARG:
        raise ValueError(msg)
    if rc == RD_ERR_CAPACITY:
        raise BufferError(msg)
    if rc == RD_ERR_STATE:
        raise RuntimeError(msg)
    raise HipError(msg)


def make_config(bit_rate, symbol_length, preamble_symbols, packet_symbols, preamble, block_size) -> RdConfig:
    if len(preamble) > RD_MAX_PREAMBLE:
        raise ValueError("preamble longer than 64 symbols is not supported")
    c = RdConfig(int(bit_rate), int(symbol_length), int(preamble_symbols), int(packet_symbols), int(block_size))
    for i, ch in enumerate(preamble):
        c.preamble[i] = int(ch)
    return c
