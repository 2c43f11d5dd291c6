"""Wideband front end: one uint8 IQ capture -> one 268.8 kSPS uint8 IQ stream
per hop channel, channelized on the GPU straight into a BatchDemodulator's input buffer.

rtldavis itself has no channelizer: it retunes one narrow-band dongle per hop
(/root/reference/src/rtldavis/runners/rtlsdr.py:51,72), so parity is unpinned.  The arithmetic is
defined in csrc/rd_channelizer.hip; oracle/channelizer_oracle.py restates it in float64 for the tests.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _lib

# protocol.py:119-171 (US band): 51 hop channels, 501.75 kHz apart
US_CHANNELS_HZ = (
    902419338, 902921088, 903422839, 903924589, 904426340, 904928090, 905429841, 905931591, 906433342,
    906935092, 907436843, 907938593, 908440344, 908942094, 909443845, 909945595, 910447346, 910949096,
    911450847, 911952597, 912454348, 912956099, 913457849, 913959599, 914461350, 914963100, 915464850,
    915966601, 916468351, 916970102, 917471852, 917973603, 918475353, 918977104, 919478854, 919980605,
    920482355, 920984106, 921485856, 921987607, 922489357, 922991108, 923492858, 923994609, 924496359,
    924998110, 925499860, 926001611, 926503361, 927005112, 927506862,
)
OUT_RATE = 268800          # 19200 bit/s x 14 samples per symbol (protocol.py:68-76, :309)
DEFAULT_DECIM = 100        # 26.88 MS/s covers the 25.1 MHz the 51 channels span
DEFAULT_CENTRE_HZ = 914963100


def design_taps(n_taps: int = 512, cutoff_hz: float = 110e3, wide_rate: float = OUT_RATE * DEFAULT_DECIM,
                beta: float = 7.0) -> np.ndarray:
    """Kaiser-windowed sinc low-pass, unit DC gain (float64).  The defaults pass the channel
    (carrier at -67.2 kHz +- deviation and data) and stop the neighbours 501.75 kHz away (~-70 dB),
    whose aliases would otherwise land beside it after the decimation."""
    n = np.arange(n_taps) - (n_taps - 1) / 2.0
    h = np.sinc(2.0 * cutoff_hz / wide_rate * n) * np.kaiser(n_taps, beta)
    return h / h.sum()


class Channelizer:
    """``Channelizer(channels_hz, centre_hz)`` moves each channel's centre to -out_rate/4, where the
    demodulator's Fs/4 rotation (dsp.py:42-49) expects the carrier, low-passes, decimates by
    ``decim`` and re-quantises to uint8 with ``gain``."""

    def __init__(self, channels_hz: Sequence[int] = US_CHANNELS_HZ, centre_hz: int = DEFAULT_CENTRE_HZ,
                 decim: int = DEFAULT_DECIM, taps: Optional[np.ndarray] = None, gain: float = 3.0,
                 out_rate: int = OUT_RATE, if_hz: Optional[int] = None) -> None:
        self.out_rate = int(out_rate)
        self.decim = int(decim)
        if self.decim < 1 or self.out_rate < 1:
            raise ValueError("decim and out_rate must be positive")
        self.wide_rate = self.out_rate * self.decim
        self.if_hz = -self.out_rate // 4 if if_hz is None else int(if_hz)
        self.taps = np.ascontiguousarray(design_taps(wide_rate=self.wide_rate) if taps is None else taps, np.float64)
        self.gain = float(gain)
        # the wideband frequency that lands on 0 Hz of the output: channel offset minus the IF
        self.shift_hz = np.ascontiguousarray([int(f) - int(centre_hz) - self.if_hz for f in channels_hz], np.int64)
        self.n_channels = self.shift_hz.size
        if self.n_channels and np.abs(self.shift_hz).max() > self.wide_rate // 2:
            raise ValueError("a channel lies outside the captured band")
        self._h = C.c_void_p()
        cfg = _lib.RdChanConfig(self.out_rate, self.decim, int(self.taps.size), int(self.n_channels), self.gain)
        _lib.check(_lib.lib().rd_chan_create(C.byref(cfg), self.taps.ctypes.data, self.shift_hz.ctypes.data,
                                             C.byref(self._h)))
        self.n_wide = 0

    def __del__(self):
        try:
            if self._h:
                _lib.lib().rd_chan_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def upload(self, wide_iq: np.ndarray) -> None:
        """Copy a capture (uint8, I,Q interleaved) to the device."""
        a = np.ascontiguousarray(wide_iq, dtype=np.uint8).reshape(-1)
        _lib.check(_lib.lib().rd_chan_upload(self._h, a.ctypes.data, a.size))
        self.n_wide = a.size // 2

    def run_host(self, n_out: Optional[int] = None) -> np.ndarray:
        """Channelized streams as a host array uint8 [n_channels, 2*n_out]."""
        n_out = self.n_wide // self.decim if n_out is None else int(n_out)
        out = np.empty((self.n_channels, 2 * n_out), np.uint8)
        _lib.check(_lib.lib().rd_chan_run_host(self._h, n_out, out.ctypes.data, out.size))
        return out

    def run_into(self, bd, hip_stream: int = 0) -> None:
        """Channelize straight into a BatchDemodulator's resident input (n_streams == n_channels)."""
        if bd.n_streams != self.n_channels:
            raise ValueError("Incompatible array sizes")
        ptr, nbytes = bd.input_ptr()
        _lib.check(_lib.lib().rd_chan_run(self._h, bd.n_samples, ptr, 2 * bd.n_samples, hip_stream))
