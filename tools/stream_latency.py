"""Latency and throughput of the streaming handle (one 8192-sample block = 30.5 ms of air time):
the synchronous call (rd_demod_block), the same call split in two with two blocks in flight
(rd_demod_submit / rd_demod_fetch: block i+1's host-to-device copy beside block i's kernels), and
sixteen receivers in lock step through one handle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rtldavis_amd import dsp, synth
cfg = dsp.PacketConfig(19200, 14, 16, 80, "1100101110001001", 8192)
B = cfg.block_size
raw = synth.synth_stream(0)
blocks = [raw[2 * B * b: 2 * B * (b + 1)] for b in range(33)]
dem = dsp.Demodulator(cfg)
ts = []
npk = 0
for rep in range(6):
    dem.reset()
    for blk in blocks:
        t0 = time.perf_counter()
        pk = dem.demodulate(blk)
        ts.append(time.perf_counter() - t0)
        npk += len(pk)
ts = np.array(ts[5:]) * 1e3
print(f"demodulate(): median {np.median(ts):.3f} ms, p99 {np.percentile(ts, 99):.3f} ms, max {ts.max():.3f} ms over {ts.size} calls; "
      f"{1e3 / np.mean(ts):.0f} blocks/s = {1e3 / np.mean(ts) * B / 1e6:.1f} MS/s; {npk} packets")
# the first state-mirror read after the first blocks (protocol.py:307-309 reads .discriminated for the first CRC-valid packet)
t0 = time.perf_counter(); d = dem.discriminated; t1 = time.perf_counter()
print(f"first .discriminated materialisation {1e3*(t1-t0):.3f} ms")
# complex input (py:144-150): what pyrtlsdr's sdr.stream() yields (runners/rtlsdr.py:100-103); (k - 127.5) / 127.5 stands in
# for pyrtlsdr's own scaling
cblocks = [((b[0::2].astype(np.float64) - 127.5) / 127.5 + 1j * (b[1::2].astype(np.float64) - 127.5) / 127.5) for b in blocks]
demc = dsp.Demodulator(cfg)
tc, npc = [], 0
for rep in range(6):
    demc.reset()
    for blk in cblocks:
        t0 = time.perf_counter()
        pk = demc.demodulate(blk)
        tc.append(time.perf_counter() - t0)
        npc += len(pk)
tc = np.array(tc[5:]) * 1e3
print(f"complex input, demodulate(): median {np.median(tc):.3f} ms, p99 {np.percentile(tc, 99):.3f} ms, max {tc.max():.3f} ms over {tc.size} calls; "
      f"{1e3 / np.mean(tc):.0f} blocks/s; {npc} packets (form: {os.environ.get('RD_STREAM_IMPL', 'one launch per block')})")
t0 = time.perf_counter()
nq = 0
for rep in range(10):
    demc.reset()
    demc.submit(cblocks[0])
    for blk in cblocks[1:]:
        demc.submit(blk)
        nq += len(demc.fetch())
    nq += len(demc.fetch())
dt = time.perf_counter() - t0
print(f"complex input, submit()/fetch(), two blocks in flight: {10 * 33 / dt:.0f} blocks/s ({1e3 * dt / 330:.3f} ms per block); {nq} packets")
t0 = time.perf_counter(); d = demc.discriminated; t1 = time.perf_counter()
print(f"complex input, first .discriminated materialisation {1e3*(t1-t0):.3f} ms")
# pipelined: two blocks in flight
npk2 = 0
t0 = time.perf_counter()
reps = 20
for rep in range(reps):
    dem.reset()
    dem.submit(blocks[0])
    for blk in blocks[1:]:
        dem.submit(blk)
        npk2 += len(dem.fetch())
    npk2 += len(dem.fetch())
dt = time.perf_counter() - t0
print(f"submit()/fetch(), two blocks in flight: {reps * 33 / dt:.0f} blocks/s = {reps * 33 * B / dt / 1e6:.1f} MS/s "
      f"({1e3 * dt / (reps * 33):.3f} ms per block); {npk2} packets ({npk2 // reps} per stream)")
# sixteen receivers in lock step
NS = 16
raws = synth.synth_streams(range(NS))
md = dsp.MultiDemodulator(cfg, NS)
md.demodulate(raws[:, : 2 * B])  # device state is allocated by the first call: keep that out of the timing
for mode in ("demodulate", "submit/fetch"):
    md.reset()
    t0 = time.perf_counter()
    n = 0
    if mode == "demodulate":
        for b in range(33):
            n += sum(len(x) for x in md.demodulate(raws[:, 2 * B * b: 2 * B * (b + 1)]))
    else:
        md.submit(raws[:, : 2 * B])
        for b in range(1, 33):
            md.submit(raws[:, 2 * B * b: 2 * B * (b + 1)])
            n += sum(len(x) for x in md.fetch())
        n += sum(len(x) for x in md.fetch())
    dt = time.perf_counter() - t0
    print(f"{NS} receivers, {mode}: {33 / dt:.0f} rounds/s = {33 * NS * B / dt / 1e6:.1f} MS/s ({1e3 * dt / 33:.3f} ms per round), {n} packets")
# per-call latency of the 16-receiver round
md.reset()
tr = []
for rep in range(4):
    md.reset()
    for b in range(33):
        t0 = time.perf_counter(); md.demodulate(raws[:, 2 * B * b: 2 * B * (b + 1)]); tr.append(time.perf_counter() - t0)
tr = np.array(tr[5:]) * 1e3
print(f"{NS} receivers, demodulate(): median {np.median(tr):.3f} ms, p99 {np.percentile(tr, 99):.3f} ms per round "
      f"(form: {os.environ.get('RD_STREAM_IMPL', 'one launch per block')})")
t0 = time.perf_counter(); d = dem.discriminated; t1 = time.perf_counter()
print(f".discriminated materialisation {1e3*(t1-t0):.3f} ms")
for _ in range(2):
    t0 = time.perf_counter(); d = dem.discriminated; t1 = time.perf_counter(); f = dem.filtered; t2 = time.perf_counter(); q = dem.quantized; t3 = time.perf_counter()
    print(f"again: discriminated {1e3*(t1-t0):.3f} ms, filtered {1e3*(t2-t1):.3f} ms, quantized {1e3*(t3-t2):.3f} ms")
