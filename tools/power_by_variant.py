"""Package power and shader clock the chip reports (rocm-smi) while a variant of the demod kernel runs back to back
(diagnostic library; wrong results for the ablated ones): what the 1400 W cap is spent on.
usage: power_by_variant.py NAME=ENV1=V1,... ...   (NAME alone = the product kernel)"""
import json, os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIAG = os.path.join(ROOT, "rtldavis_amd", "librtldavis_hip_diag.so")
CHILD = r'''
import sys, os, json, time
sys.path.insert(0, os.environ["RD_REPO_ROOT"])
import numpy as np
from rtldavis_amd import batch, dsp, synth
cfg = dsp.PacketConfig(19200, 14, 16, 80, "1100101110001001", 8192)
host = np.tile(synth.synth_streams(range(64)), (64, 1))
bds = [batch.BatchDemodulator(cfg, 4096, 33) for _ in range(2)]
for bd in bds:
    bd.upload(host); bd.set_timing(1)
print("READY", flush=True)
t0 = time.time(); n = 0
while time.time() - t0 < float(os.environ.get("RD_PW_SECONDS", "7")):
    for bd in bds:
        for _ in range(8): bd.run()
    for bd in bds: bd.results()
    n += 16
tm = bds[0].timing()
print("DONE " + json.dumps({"demod_ms": tm["demod_ms"], "total_ms": tm["total_ms"], "runs": n}), flush=True)
'''
def smi():
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "-d", "0"], capture_output=True, text=True).stdout
    p = re.search(r"Power \(W\): ([0-9.]+)", out); s = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
    return (float(p.group(1)) if p else float("nan"), int(s.group(1)) if s else -1)
for a in sys.argv[1:] or ["product"]:
    name, _, envs = a.partition("=")
    env = dict(os.environ); env["RTLDAVIS_HIP_LIB"] = DIAG; env["RD_REPO_ROOT"] = ROOT
    env.update(dict(kv.split("=", 1) for kv in envs.split(",") if kv) if envs else {})
    pr = subprocess.Popen([sys.executable, "-c", CHILD], env=env, stdout=subprocess.PIPE, text=True, cwd=ROOT)
    assert pr.stdout.readline().startswith("READY")
    time.sleep(2.5)
    samples = []
    for _ in range(6):
        samples.append(smi()); time.sleep(0.5)
    line = pr.stdout.readline(); pr.wait()
    t = json.loads(line[5:]) if line.startswith("DONE") else {}
    pw = [s[0] for s in samples]; ck = [s[1] for s in samples]
    print(f"{name:20s} demod {t.get('demod_ms', float('nan')):.4f} ms  whole run {t.get('total_ms', float('nan')):.4f} ms   package power {min(pw):.0f}-{max(pw):.0f} W   sclk {min(ck)}-{max(ck)} MHz", flush=True)
