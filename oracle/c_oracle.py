"""ctypes binding of oracle/dsp_oracle.c.  TEST INFRASTRUCTURE ONLY (see dsp_oracle.c)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "liboracle.so")


class OracleCfg(C.Structure):
    _fields_ = [("bit_rate", C.c_int32), ("symbol_length", C.c_int32), ("preamble_symbols", C.c_int32),
                ("packet_symbols", C.c_int32), ("block_size", C.c_int32), ("preamble", C.c_uint8 * 64)]


class OraclePkt(C.Structure):
    _fields_ = [("stream", C.c_int32), ("call", C.c_int32), ("index", C.c_int32), ("nbytes", C.c_int32),
                ("data", C.c_uint8 * 32), ("rssi", C.c_double), ("snr", C.c_double)]


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "dsp_oracle.c")
    newest = os.path.join(HERE, "_build", "liboracle_v4.so")
    if force or not os.path.exists(LIB_PATH) or not os.path.exists(newest) or \
            os.path.getmtime(newest) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", HERE])
    return LIB_PATH


def best_isa_level() -> int:
    """Highest x86-64 micro-architecture level (2, 3 or 4) this host's /proc/cpuinfo flags cover."""
    try:
        with open("/proc/cpuinfo") as fh:
            flags = set()
            for line in fh:
                if line.startswith("flags"):
                    flags = set(line.split(":", 1)[1].split())
                    break
    except OSError:
        return 2
    v3 = {"avx", "avx2", "bmi1", "bmi2", "f16c", "fma", "movbe", "xsave", "abm"}  # abm = lzcnt
    v4 = {"avx512f", "avx512bw", "avx512cd", "avx512dq", "avx512vl"}
    if v3 <= flags:
        return 4 if v4 <= flags else 3
    return 2


def use_isa_level(level: int) -> int:
    """Switch the loaded library to the build for `level` (2..4, clamped to what the host runs);
    returns the level in use.  The tests stay on the default v2 build."""
    global _lib, LIB_PATH
    level = max(2, min(int(level), best_isa_level()))
    LIB_PATH = os.path.join(HERE, "_build", "liboracle.so" if level == 2 else f"liboracle_v{level}.so")
    _lib = None
    return level


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.oracle_demod_stream.restype = C.c_long
        L.oracle_demod_stream.argtypes = [C.c_void_p, C.c_long, C.POINTER(OracleCfg), C.c_int32, C.c_void_p,
                                          C.c_void_p, C.POINTER(OraclePkt), C.c_long]
        L.oracle_demod_batch.restype = C.c_long
        L.oracle_demod_batch.argtypes = [C.c_void_p, C.c_long, C.c_long, C.POINTER(OracleCfg), C.c_int, C.c_void_p,
                                         C.POINTER(OraclePkt), C.c_long, C.c_void_p]
        L.oracle_stages.restype = None
        L.oracle_stages.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_crc16_ccitt.restype = C.c_uint16
        L.oracle_crc16_ccitt.argtypes = [C.c_char_p, C.c_long]
        L.oracle_swap_bit_order.restype = C.c_uint8
        L.oracle_swap_bit_order.argtypes = [C.c_uint8]
        _lib = L
    return _lib


def make_cfg(bit_rate=19200, symbol_length=14, preamble_symbols=16, packet_symbols=80,
             preamble="1100101110001001", block_size=8192) -> OracleCfg:
    c = OracleCfg(bit_rate, symbol_length, preamble_symbols, packet_symbols, block_size)
    for i, ch in enumerate(preamble):
        c.preamble[i] = int(ch)
    return c


class Pkt:
    __slots__ = ("stream", "call", "index", "data", "rssi", "snr")

    def __init__(self, p: OraclePkt):
        self.stream, self.call, self.index = p.stream, p.call, p.index
        self.data = np.frombuffer(bytes(p.data[: p.nbytes]), dtype=np.uint8)
        self.rssi, self.snr = p.rssi, p.snr


def demod_stream(raw: np.ndarray, cfg: OracleCfg, want_disc=False, cap=4096):
    """Returns (calls: list per call of Pkt, bits_le: uint8[(n+7)//8], disc or None)."""
    raw = np.ascontiguousarray(raw, dtype=np.uint8)
    n = raw.size // 2
    bits = np.zeros((n + 7) // 8, dtype=np.uint8)
    disc = np.zeros(n, dtype=np.float64) if want_disc else None
    out = (OraclePkt * cap)()
    r = lib().oracle_demod_stream(raw.ctypes.data, n, C.byref(cfg), 0, bits.ctypes.data,
                                  disc.ctypes.data if want_disc else None, out, cap)
    if r < 0:
        raise ValueError("oracle_demod_stream: bad arguments")
    calls = [[] for _ in range(n // cfg.block_size)]
    for i in range(r):
        calls[out[i].call].append(Pkt(out[i]))
    return calls, bits, disc


def demod_batch(raw: np.ndarray, cfg: OracleCfg, threads: int, want_bits=False, cap_per_stream=64):
    """raw: [S, 2n] uint8.  Returns (list per stream of list of Pkt, bits or None)."""
    raw = np.ascontiguousarray(raw, dtype=np.uint8)
    S, n = raw.shape[0], raw.shape[1] // 2
    bits = np.zeros((S, (n + 7) // 8), dtype=np.uint8) if want_bits else None
    out = (OraclePkt * (cap_per_stream * S))()
    counts = np.zeros(S, dtype=np.int64)
    r = lib().oracle_demod_batch(raw.ctypes.data, S, n, C.byref(cfg), threads,
                                 bits.ctypes.data if want_bits else None, out, cap_per_stream, counts.ctypes.data)
    if r < 0:
        raise ValueError("oracle_demod_batch failed")
    res = [[Pkt(out[s * cap_per_stream + i]) for i in range(int(counts[s]))] for s in range(S)]
    return res, bits


def stages(raw: np.ndarray):
    """Returns (filtered complex128[n+1] starting at f[-1], disc f64[n], bits u8[n])."""
    raw = np.ascontiguousarray(raw, dtype=np.uint8)
    n = raw.size // 2
    filt = np.zeros(2 * (n + 1), dtype=np.float64)
    disc = np.zeros(n, dtype=np.float64)
    bits = np.zeros(n, dtype=np.uint8)
    lib().oracle_stages(raw.ctypes.data, n, filt.ctypes.data, disc.ctypes.data, bits.ctypes.data)
    return filt.view(np.complex128), disc, bits
