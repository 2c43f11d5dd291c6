"""In-kernel stamps of the demod kernel (diagnostic library, RD_K1_STAMPS=1): where a wave's time goes.

Per wave the kernel sums, in shader cycles (s_memtime), the loop-top wait for the tile (vmcnt), the stretch from
there to the last load of the next tile issued (window read, word store, address arithmetic, LDS-DMA issue) and the
rest (the arithmetic); it also stamps s_memrealtime (100 MHz) at both ends, so that the clock the chip really held is
d(memtime) / d(memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS give-back item 6).

usage: k1_stamps.py NAME=ENV1=V1,ENV2=V2 ...   one subprocess per variant (the library reads its switches once)
RD_K1_STAMPS=1 must be part of the variant.  A stamped build is ~10 % slower than the real kernel:
read the SHARES and the clock, not the length.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIAG = os.path.join(ROOT, "rtldavis_amd", "librtldavis_hip_diag.so")
CHILD = r'''
import sys, os, json, ctypes as C
sys.path.insert(0, os.environ["RD_REPO_ROOT"])
import numpy as np
from rtldavis_amd import _lib, batch, dsp, synth
cfg = dsp.PacketConfig(19200, 14, 16, 80, "1100101110001001", 8192)
uniq = synth.synth_streams(range(64))
host = np.tile(uniq, (64, 1))
bds = [batch.BatchDemodulator(cfg, 4096, 33) for _ in range(2)]
for bd in bds:
    bd.upload(host)
    bd.set_timing(1)
import time
t0 = time.time()
i = 0
while time.time() - t0 < 2.5:          # >= 2 s of back-to-back launches before the stamps are read
    for bd in bds:
        for _ in range(8): bd.run()
    for bd in bds: bd.results()
    i += 1
for bd in bds: bd.timing()
for _ in range(20):
    for bd in bds: bd.run()
for bd in bds: bd.results()
tm = bds[0].timing()
L = _lib.lib()
W = 12
cap = 8192
buf = (C.c_uint64 * (cap * W))()
n = C.c_uint32(0)
L.rd_diag_read_stamps.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
rc = L.rd_diag_read_stamps(buf, cap, C.byref(n))
a = np.frombuffer(buf, dtype=np.uint64).reshape(cap, W)[: n.value].astype(np.float64)
a = a[a[:, 4] > 0]
out = {"demod_ms": tm["demod_ms"], "rc": rc, "waves": int(a.shape[0])}
if a.shape[0]:
    it = a[:, 4]
    tot = a[:, 0] + a[:, 1] + a[:, 2]
    out.update({
        "tiles_per_wave_med": float(np.median(it)), "tiles_per_wave_min": float(it.min()), "tiles_per_wave_max": float(it.max()),
        "cycles_per_tile": float((tot / it).mean()),
        "wait_per_tile": float((a[:, 0] / it).mean()), "gap_per_tile": float((a[:, 1] / it).mean()),
        "comp_per_tile": float((a[:, 2] / it).mean()),
        "share_wait": float(a[:, 0].sum() / tot.sum()), "share_gap": float(a[:, 1].sum() / tot.sum()),
        "share_comp": float(a[:, 2].sum() / tot.sum()),
        "wait_after_store_per_iter": float(a[:, 3].sum() / max(1.0, a[:, 5].sum())),
        "wait_other_per_iter": float((a[:, 0].sum() - a[:, 3].sum()) / max(1.0, (a[:, 4] - a[:, 5]).sum())),
        "wait_max_med": float(np.median(a[:, 8])), "wait_max_max": float(a[:, 8].max()),
        "clock_GHz_med": float(np.median(a[:, 6] / a[:, 7]) * 0.1), "clock_GHz_min": float((a[:, 6] / a[:, 7]).min() * 0.1),
        "clock_GHz_max": float((a[:, 6] / a[:, 7]).max() * 0.1),
        "wave_lifetime_us_med": float(np.median(a[:, 7]) / 100.0), "wave_lifetime_us_min": float(a[:, 7].min() / 100.0),
        "wave_lifetime_us_max": float(a[:, 7].max() / 100.0),
        "kernel_span_us": float((a[:, 10].max() - a[:, 9].min()) / 100.0),
        "first_wave_start_spread_us": float((a[:, 9].max() - a[:, 9].min()) / 100.0),
        "last_wave_end_spread_us": float((a[:, 10].max() - a[:, 10].min()) / 100.0),
    })
    # per XCD: tiles done and mean cycles per tile
    xs = {}
    for x in range(8):
        m = a[:, 11] == x
        if m.any():
            xs[str(x)] = {"waves": int(m.sum()), "tiles": float(it[m].sum()), "cycles_per_tile": float((tot[m] / it[m]).mean()),
                          "wait_share": float(a[m, 0].sum() / tot[m].sum())}
    out["per_xcd"] = xs
print(json.dumps(out))
'''


def main():
    variants = []
    for a in sys.argv[1:]:
        name, _, envs = a.partition("=")
        env = dict(kv.split("=", 1) for kv in envs.split(",") if kv) if envs else {}
        variants.append((name, env))
    for name, env in variants:
        e = dict(os.environ)
        e.update(env)
        e["RD_REPO_ROOT"] = ROOT
        e.setdefault("RTLDAVIS_HIP_LIB", DIAG)
        out = subprocess.run([sys.executable, "-c", CHILD], env=e, capture_output=True, text=True, cwd=ROOT)
        try:
            t = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception as ex:  # noqa: BLE001
            print(name, "FAILED", repr(ex), out.stderr[-600:])
            continue
        print(f"== {name}  ({' '.join(f'{k}={v}' for k, v in env.items())})")
        per = t.pop("per_xcd", None)
        for k, v in t.items():
            print(f"   {k:28s} {v:.4f}" if isinstance(v, float) else f"   {k:28s} {v}")
        if per:
            print("   per XCD: " + "  ".join(f"[{x}] {d['waves']}w {d['tiles']:.0f}t {d['cycles_per_tile']:.0f}c/t wait {d['wait_share']:.2f}"
                                             for x, d in sorted(per.items())))
        sys.stdout.flush()


main()
