"""(uint8: 0 start, 1 loads, 2 stores, 3 signs, 4 search, 5 end.  complex128: 0 start, 1 loads + LDS, 2 signs, 3 counted in (workgroup 0);
4 window loaded, 5 search, 6 end (the last workgroup).)
Phase stamps of the streaming one-launch blocks (diagnostic library, RD_SB_STAMPS=1): where a block's kernel time goes.
RTLDAVIS_HIP_LIB=$PWD/rtldavis_amd/librtldavis_hip_diag.so RD_SB_STAMPS=1 python3 tools/experiments/r4_sb_stamps.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np
from rtldavis_amd import dsp, synth, _lib
L = _lib.lib()
L.rd_diag_read_sb_stamps.argtypes = [C.c_void_p]
cfg = dsp.PacketConfig(19200, 14, 16, 80, "1100101110001001", 8192)
B = 8192
raw = synth.synth_stream(0)
blocks = [raw[2 * B * b: 2 * B * (b + 1)] for b in range(33)]
cblocks = [((b[0::2].astype(np.float64) - 127.5) / 127.5 + 1j * (b[1::2].astype(np.float64) - 127.5) / 127.5) for b in blocks]
names = ["0>1", "1>2", "2>3", "3>4", "4>5", "5>6", "whole"]
for name, blks in (("uint8", blocks), ("complex128", cblocks)):
    dem = dsp.Demodulator(cfg)
    rows, withpk = [], []
    for rep in range(4):
        dem.reset()
        for blk in blks:
            pk = dem.demodulate(blk)
            st = np.zeros(8, np.uint64)
            assert L.rd_diag_read_sb_stamps(st.ctypes.data) == 0
            d = st.astype(np.int64)
            last = 6 if d[6] > d[5] else 5
            rows.append([(d[i + 1] - d[i]) / 100.0 for i in range(6)] + [(d[last] - d[0]) / 100.0])
            withpk.append(len(pk) > 0)
    rows = np.array(rows[3:]); withpk = np.array(withpk[3:])
    print(f"{name}: {len(rows)} blocks, {withpk.sum()} with packets")
    for i, n in enumerate(names):
        print(f"  {n:28s} median {np.median(rows[:, i]):6.2f} us   p90 {np.percentile(rows[:, i], 90):6.2f}   with packets {np.median(rows[withpk, i]) if withpk.any() else 0:6.2f}")
