"""TEST INFRASTRUCTURE - float64 restatement of the wideband channelizer's definition
(rtldavis_amd/csrc/rd_channelizer.hip).  PARITY UNPINNED: rtldavis has no channelizer (it retunes
one dongle per hop, /root/reference/src/rtldavis/runners/rtlsdr.py:51,72), so there is no reference
output to pin this against; the tests tie it to the reference through the packets the (pinned)
demodulator recovers from its output.  Only tests/ may import this module."""
import numpy as np


def channelize(raw, shift_hz, taps, decim, out_rate, gain, n_out=None):
    """raw: uint8 I,Q interleaved capture at decim*out_rate; returns uint8 [n_channels, 2*n_out].
    z_c[t] = sum_k h[k] x[D t - k] exp(-2j pi shift_c (D t - k) / Fw), x = LUT of dsp.py:20-39,
    out = clip(rint(gain z 127.6 + 127.4), 0, 255)."""
    raw = np.asarray(raw, np.uint8).reshape(-1)
    x = (raw[0::2].astype(np.float64) - 127.4) / 127.6 + 1j * ((raw[1::2].astype(np.float64) - 127.4) / 127.6)
    fw = int(out_rate) * int(decim)
    n = x.size
    n_out = n // decim if n_out is None else int(n_out)
    taps = np.asarray(taps, np.float64)
    T = taps.size
    nn = np.arange(n, dtype=np.int64)
    outs = np.empty((len(shift_hz), 2 * n_out), np.uint8)
    for c, sh in enumerate(shift_hz):
        ph = ((int(sh) * nn) % fw).astype(np.float64) / fw          # exact integer remainder
        y = x * np.exp(-2j * np.pi * ph)
        ypad = np.concatenate([np.zeros(T - 1, np.complex128), y])
        z = np.empty(n_out, np.complex128)
        # z[t] = sum_k h[k] y[D t - k]: windows of ypad ending at D t
        idx = decim * np.arange(n_out)
        step = 4096
        hr = taps[::-1]
        for a in range(0, n_out, step):
            b = min(n_out, a + step)
            win = np.lib.stride_tricks.sliding_window_view(ypad, T)[idx[a:b]]
            z[a:b] = win @ hr
        z *= gain
        outs[c, 0::2] = np.clip(np.rint(z.real * 127.6 + 127.4), 0, 255)
        outs[c, 1::2] = np.clip(np.rint(z.imag * 127.6 + 127.4), 0, 255)
    return outs
