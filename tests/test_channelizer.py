"""Wideband front end.  PARITY UNPINNED: rtldavis has no channelizer, so the
oracle here (oracle/channelizer_oracle.py) is this repo's own float64 restatement; what ties it to
the reference is that the reference-pinned demodulator recovers the injected packets from its
output.  CPU tests cover the definition and the tap design, GPU tests the HIP kernel against it."""
import os

import numpy as np
import pytest

from rtldavis_amd import synth


def _cz():
    from rtldavis_amd import channelizer
    return channelizer


def test_tap_design_passes_the_channel_and_stops_the_neighbours():
    CZ = _cz()
    h = CZ.design_taps()
    assert h.size == 512 and abs(h.sum() - 1.0) < 1e-12 and np.allclose(h, h[::-1])
    fw = CZ.OUT_RATE * CZ.DEFAULT_DECIM
    f = np.array([0.0, 67.2e3 + 15e3, 501750.0 - 67.2e3 - 20e3, 501750.0 + 67.2e3 + 20e3])
    H = np.abs(np.exp(-2j * np.pi * np.outer(f / fw, np.arange(h.size))) @ h)
    assert H[0] > 0.999 and H[1] > 0.7          # carrier at |IF| + deviation + data still passed
    assert 20 * np.log10(H[2:].max()) < -60.0    # both edges of the neighbouring channel


def test_oracle_channelizer_feeds_the_pinned_demodulator():
    """Three bursts at the band edges and next to the capture's centre, one wideband capture ->
    float64 channelizer -> C oracle demodulator: every injected packet comes back."""
    from oracle import c_oracle as CO
    from oracle import channelizer_oracle as CHO
    CZ = _cz()
    chans = [0, 25, 50]
    off = [CZ.US_CHANNELS_HZ[c] - CZ.DEFAULT_CENTRE_HZ for c in chans]
    raw, info = synth.synth_wideband([1, 2, 3], off, 3 * 8192)
    shifts = [f + CZ.OUT_RATE // 4 for f in off]
    nb = CHO.channelize(raw, shifts, CZ.design_taps(), CZ.DEFAULT_DECIM, CZ.OUT_RATE, 3.0)
    assert nb.shape == (3, 2 * 3 * 8192)
    res, _ = CO.demod_batch(nb, CO.make_cfg(), threads=3)
    for (payload, start), pk in zip(info, res):
        hits = [(p.call, p.index) for p in pk if bytes(p.data).hex() == payload]
        assert hits, payload
        # the preamble sits 32 symbols after the burst start; the filters and the oversampled match add ~20 samples
        call, idx = hits[0]
        pos = (call - 1) * 8192 + idx
        assert 0 <= pos - (start + 32 * 14) <= 30


@pytest.mark.gpu
def test_channelizer_matches_float64_model():
    from oracle import channelizer_oracle as CHO
    CZ = _cz()
    chans = [0, 7, 24, 25, 26, 50]
    off = [CZ.US_CHANNELS_HZ[c] - CZ.DEFAULT_CENTRE_HZ for c in chans]
    raw, _ = synth.synth_wideband([11, 12, 13, 14, 15, 16], off, 3 * 8192)
    cz = CZ.Channelizer([CZ.US_CHANNELS_HZ[c] for c in chans])
    cz.upload(raw)
    got = cz.run_host()
    want = CHO.channelize(raw, cz.shift_hz, cz.taps, cz.decim, cz.out_rate, cz.gain)
    d = got.astype(np.int32) - want.astype(np.int32)
    # fp32 sums of 512 products against float64: never more than the last bit, and that rarely
    assert np.abs(d).max() <= 1
    assert (d != 0).mean() < 1e-3
    # a shorter output and a ragged capture length
    cz.upload(raw[: 2 * (8192 * 100 + 37)])
    got2 = cz.run_host(8000)
    assert np.array_equal(got2, got[:, : 2 * 8000])
    with pytest.raises(ValueError):
        cz.run_host(8193)


@pytest.mark.gpu
def test_51_hop_channels_from_one_capture():
    """BASELINE configs[2]: 51 channels, one capture, one channelizer launch straight into the
    batch demodulator's input; every channel's packet is recovered where it was injected."""
    from rtldavis_amd import batch, dsp
    CZ = _cz()
    nb = 3
    off = [f - CZ.DEFAULT_CENTRE_HZ for f in CZ.US_CHANNELS_HZ]
    raw, info = synth.synth_wideband(range(100, 151), off, nb * 8192, amplitude=0.05)
    cz = CZ.Channelizer()
    cz.upload(raw)
    cfg = dsp.PacketConfig(19200, 14, 16, 80, "1100101110001001", 8192)
    bd = batch.BatchDemodulator(cfg, 51, nb)
    cz.run_into(bd)
    bd.run()
    recs = bd.results()
    for c, (payload, start) in enumerate(info):
        hits = [(int(r["call"]), int(r["index"])) for r in recs
                if int(r["stream"]) == c and r["data"][: int(r["nbytes"])].tobytes().hex() == payload]
        assert hits, (c, payload)
        pos = (hits[0][0] - 1) * 8192 + hits[0][1]
        assert 0 <= pos - (start + 32 * 14) <= 30, (c, pos, start)
    # the demodulator saw exactly the channelizer's bytes
    host = cz.run_host(nb * 8192)
    bd2 = batch.BatchDemodulator(cfg, 51, nb)
    got = bd2.demodulate(host)
    flat = sorted((s, c, p.index, bytes(p.data)) for s in range(51) for c, ps in enumerate(got[s]) for p in ps)
    assert flat == sorted((int(r["stream"]), int(r["call"]), int(r["index"]), r["data"][: int(r["nbytes"])].tobytes())
                          for r in recs)
