"""BASELINE configs[2] made literal: 51 US hop channels demodulated concurrently from ONE synthetic
wideband capture (26.88 MS/s uint8 IQ) on one MI355X: channelizer -> batch demodulator, everything
resident on the device.  Diagnostic tool, not the bench.
usage: wideband_demo.py [n_blocks=8]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rtldavis_amd import batch, channelizer as CZ, dsp, synth


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    n_out = nb * 8192
    cz = CZ.Channelizer()
    t0 = time.perf_counter()
    raw, info = synth.synth_wideband(range(100, 151), [f - CZ.DEFAULT_CENTRE_HZ for f in CZ.US_CHANNELS_HZ], n_out,
                                     amplitude=0.05)
    print(f"synthetic capture: {raw.size // 2 / 1e6:.1f} M samples ({n_out / CZ.OUT_RATE:.2f} s of air), 51 bursts, "
          f"{time.perf_counter() - t0:.1f} s to generate")
    cz.upload(raw)
    cfg = dsp.PacketConfig(19200, 14, 16, 80, "1100101110001001", 8192)
    bd = batch.BatchDemodulator(cfg, cz.n_channels, nb)
    cz.run_into(bd); cz.run_host(1)  # warm-up; the synchronous copy drains the default stream
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        cz.run_into(bd)
    cz.run_host(1)
    dt_c = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        cz.run_into(bd)
        bd.run()
        recs = bd.results()
    dt_all = (time.perf_counter() - t0) / reps
    ok = 0
    for c, (payload, _start) in enumerate(info):
        ok += payload in [r["data"][: int(r["nbytes"])].tobytes().hex() for r in recs if int(r["stream"]) == c]
    flops = 8.0 * cz.n_channels * n_out * cz.taps.size
    print(f"channelizer: {1e3 * dt_c:.3f} ms per capture = {n_out * cz.decim / dt_c / 1e6:.0f} wideband MS/s "
          f"({flops / dt_c / 1e12:.1f} TFLOP/s fp32), {n_out * cz.decim / dt_c / (CZ.OUT_RATE * cz.decim):.0f}x real time")
    print(f"channelizer + demod + results: {1e3 * dt_all:.3f} ms per capture; packets recovered: {ok} of {len(info)}")


main()
