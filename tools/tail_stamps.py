"""Phase stamps of the one-launch tail (k_tail, diagnostic library, RD_FT_STAMPS=1): where a workgroup's time goes.
Stamps are s_memrealtime (100 MHz): 0 start, 1 fix-up done (barrier), 6 wave 0's end of the search, 2 search done (barrier),
3 slice done + totals published, 4 the prefix of the groups in front known (wave 3), 5 wave 0's last record stored.
usage: RTLDAVIS_HIP_LIB=rtldavis_amd/librtldavis_hip_diag.so RD_FT_STAMPS=1 python tools/tail_stamps.py [streams] [blocks]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtldavis_amd import _lib, batch, dsp, synth  # noqa: E402

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 33
cfg = dsp.PacketConfig(19200, 14, 16, 80, "1100101110001001", 8192)
uniq = synth.synth_streams(range(64))[:, : 2 * 8192 * nb]
host = np.tile(uniq, ((ns + 63) // 64, 1))[:ns]
bds = [batch.BatchDemodulator(cfg, ns, nb) for _ in range(2)]
for bd in bds:
    bd.upload(host)
    bd.set_timing(1)
for _ in range(40):
    for bd in bds:
        bd.run()
    for bd in bds:
        bd.results()
tm = bds[0].timing()
L = _lib.lib()
groups = (ns + 3) // 4
buf = (C.c_uint64 * (groups * 8))()
n = C.c_uint32(0)
L.rd_diag_read_tail_stamps.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
rc = L.rd_diag_read_tail_stamps(buf, groups, C.byref(n))
a = np.frombuffer(buf, dtype=np.uint64).reshape(groups, 8)[: n.value].astype(np.float64) / 100.0   # us
t0 = a[:, 0].min()
a = a - t0
print(f"rc {rc} groups {n.value}; demod {tm['demod_ms']:.4f} ms, whole run {tm['total_ms']:.4f} ms -> tail {1e3 * (tm['total_ms'] - tm['demod_ms']):.1f} us")
def col(name, v):
    print(f"  {name:34s} mean {v.mean():7.2f}  p10 {np.percentile(v, 10):7.2f}  median {np.median(v):7.2f}  p90 {np.percentile(v, 90):7.2f}  max {v.max():7.2f} us")
col("start after the first group's", a[:, 0])
col("0 fix-up (to barrier)", a[:, 1] - a[:, 0])
col("1 search, wave 0 alone", a[:, 6] - a[:, 1])
col("1 search (to barrier)", a[:, 2] - a[:, 1])
col("2 slice: words loaded, entries made", a[:, 7] - a[:, 2])
col("2 slice + publish (to barrier)", a[:, 3] - a[:, 2])
col("3 prefix known (wave 3) after 3", a[:, 4] - a[:, 3])
col("3 rssi + records (wave 0) after 3", np.maximum(a[:, 5] - a[:, 3], 0))
col("end (wave 0) after the first start", np.maximum(a[:, 5], a[:, 4]))
print(f"  kernel span by the stamps: {np.maximum(a[:, 5], a[:, 4]).max():.2f} us")
sr = a[:, 6] - a[:, 1]
print("  search (wave 0) by group % 8:", " ".join(f"{sr[k::8].mean():.1f}" for k in range(8)))
print("  search (wave 0) by eighth of the grid:", " ".join(f"{c.mean():.1f}" for c in np.array_split(sr, 8)))
print("  search (wave 0) by group % 32 // 8 (SE?):", " ".join(f"{sr[(np.arange(sr.size) % 32) // 8 == k].mean():.1f}" for k in range(4)))
end = np.maximum(a[:, 5], a[:, 4])
print("  end by eighth of the grid:", " ".join(f"{c.mean():.1f}" for c in np.array_split(end, 8)), "| max per eighth:", " ".join(f"{c.max():.1f}" for c in np.array_split(end, 8)))
late = np.argsort(-end)[:10]
for g in late:
    print(f"  late group {g}: start {a[g,0]:.1f} fix {a[g,1]-a[g,0]:.1f} search {a[g,2]-a[g,1]:.1f} slice {a[g,3]-a[g,2]:.1f} prefix +{a[g,4]-a[g,3]:.1f} records +{a[g,5]-a[g,3]:.1f} end {end[g]:.1f}")
