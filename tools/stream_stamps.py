"""Phase stamps of the streaming handle's one-launch blocks (diagnostic library, s_memrealtime inside the kernels): where the
kernel time of one demodulate() call goes, for a uint8 block (k_stream_block) and a complex128 block (k_stream_block_cplx).
  RTLDAVIS_HIP_LIB=$PWD/rtldavis_amd/librtldavis_hip_diag.so RD_SB_STAMPS=1 python3 tools/stream_stamps.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rtldavis_amd import dsp, synth, _lib
L = _lib.lib()
L.rd_diag_read_sb_stamps.argtypes = [C.c_void_p]
cfg = dsp.PacketConfig(19200, 14, 16, 80, "1100101110001001", 8192)
B = 8192
raw = synth.synth_stream(0)
blocks = [raw[2 * B * b: 2 * B * (b + 1)] for b in range(33)]
cblocks = [((b[0::2].astype(np.float64) - 127.5) / 127.5 + 1j * (b[1::2].astype(np.float64) - 127.5) / 127.5) for b in blocks]
PHASES = {
    "uint8": ["loads issued (barrier)", "block in (device memory, or the link), ring + LDS stores", "exact sign bits", "window out + search", "slice, RSSI, flag"],
    "complex128": ["workgroup 0: piece in (device memory, or the link), ring + LDS", "workgroup 0: signs (float64, from LDS)",
                   "workgroup 0: words out, stores acknowledged, counted in", "last workgroup: (arrival of the others,) window in",
                   "last workgroup: search", "last workgroup: slice, RSSI, flag"],
}
for name, blks in (("uint8", blocks), ("complex128", cblocks)):
    ph = PHASES[name]
    n = len(ph)
    dem = dsp.Demodulator(cfg)
    rows, withpk = [], []
    for rep in range(4):
        dem.reset()
        for blk in blks:
            pk = dem.demodulate(blk)
            st = np.zeros(8, np.uint64)
            assert L.rd_diag_read_sb_stamps(st.ctypes.data) == 0
            d = st.astype(np.int64)
            rows.append([(d[i + 1] - d[i]) / 100.0 for i in range(n)] + [(d[n] - d[0]) / 100.0])
            withpk.append(len(pk) > 0)
    rows = np.array(rows[3:]); withpk = np.array(withpk[3:])
    print(f"{name} block of {B} samples: {len(rows)} calls, {int(withpk.sum())} with packets; us, first stamp to last")
    for i, t in enumerate(ph + ["whole kernel (first stamp to last)"]):
        print(f"  {t:62s} median {np.median(rows[:, i]):6.2f}   p90 {np.percentile(rows[:, i], 90):6.2f}   calls with packets {np.median(rows[withpk, i]):6.2f}")
