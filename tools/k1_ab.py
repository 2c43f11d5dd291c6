"""A/B timing of kernel variants under continuous load (diagnostic).
Each variant is selected through environment variables read at library load, so every variant
runs in its own subprocess; all share one GPU box and are interleaved round-robin.
usage: k1_ab.py [--key demod_ms] [--rounds N] 'NAME=ENV1=V1,ENV2=V2' ...   (NAME alone = defaults)
The variants live in the DIAGNOSTIC library (rtldavis_amd/librtldavis_hip_diag.so, `make -C rtldavis_amd/csrc diag`):
the product library has none of the wrong-result switches.  RTLDAVIS_HIP_LIB in a variant's environment overrides."""
import json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIAG = os.path.join(ROOT, "rtldavis_amd", "librtldavis_hip_diag.so")
CHILD = r'''
import sys, os, json
sys.path.insert(0, os.environ["RD_REPO_ROOT"])
import numpy as np
from rtldavis_amd import batch, dsp, synth
cfg = dsp.PacketConfig(19200, 14, 16, 80, "1100101110001001", 8192)
uniq = synth.synth_streams(range(64))
host = np.tile(uniq, (64, 1))
bd = batch.BatchDemodulator(cfg, 4096, 33)
bd.upload(host)
bd.set_timing(int(os.environ.get("RD_AB_TIMING", "2")))
for _ in range(10): bd.run()
bd.results(); bd.timing()
for _ in range(40): bd.run()
bd.results()
print(json.dumps(bd.timing()))
'''

def main():
    args = sys.argv[1:]
    key = "demod_ms"
    rounds = 3
    while args and args[0] in ("--key", "--rounds"):
        if args[0] == "--key":
            key = args[1]
        else:
            rounds = int(args[1])
        args = args[2:]
    variants = []
    for a in args:
        name, _, envs = a.partition("=")
        env = dict(kv.split("=", 1) for kv in envs.split(",") if kv) if envs else {}
        variants.append((name, env))
    res = {n: [] for n, _ in variants}
    for rnd in range(rounds):
        for name, env in variants:
            e = dict(os.environ); e["RTLDAVIS_HIP_LIB"] = DIAG; e.update(env); e["RD_REPO_ROOT"] = ROOT
            out = subprocess.run([sys.executable, "-c", CHILD], env=e, capture_output=True, text=True, cwd=ROOT)
            try:
                t = json.loads(out.stdout.strip().splitlines()[-1])
                res[name].append(t if key == "all" else t[key])
            except Exception as ex:
                print(name, "FAILED", repr(ex), out.stderr[-400:])
    if key == "all":
        for name, v in res.items():
            if not v:
                continue
            print(f"{name:24s} " + "  ".join(f"{k[:-3]} {min(x[k] for x in v):.4f}" for k in v[0] if k != "runs"))
        return
    for name, v in res.items():
        print(f"{name:24s} {key} min {min(v):.4f}  med {sorted(v)[len(v)//2]:.4f}  all {['%.4f' % x for x in v]}")

main()
