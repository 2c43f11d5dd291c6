"""Summaries for tools/profile_round.sh (rocprofv3 csv output -> the small files kept under profiles/)."""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


DEMOD_KERNEL_SYMBOL = "_Z12k_demod_mfmaILi0ELb0EE"   # k_demod_mfma<0, false>: the product kernel
STAMP_FILE = os.path.join(ROOT, "rtldavis_amd", "librtldavis_hip.stamp")


def isa_sha256(asm_text: str, symbol_prefix: str = DEMOD_KERNEL_SYMBOL) -> str:
    """sha256 over the INSTRUCTIONS of the dominant kernel, from the device assembly hipcc writes for
    rd_demod_mfma.hip (-S --cuda-device-only, the product's flags): the kernel's body from its label to s_endpgm,
    comments and local labels dropped, basic-block label numbers normalised (they count the functions in front of the
    kernel).  Any change to the kernel's code changes it; other kernels, host code, diagnostic variants and comments
    do not.  (The bytes inside the .so are no use for this: their pc-relative literals move with every function added
    to the file.)  '' when the kernel is not in the text."""
    import re
    body, inside = [], False
    for line in asm_text.splitlines():
        if not inside:
            if line.startswith(symbol_prefix) and line.rstrip().split(";")[0].rstrip().endswith(":"):
                inside = True
            continue
        t = line.split(";")[0].strip()
        if not t or t.startswith(".L") or t.startswith("."):
            continue
        body.append(re.sub(r"LBB\d+_", "LBB_", t))
        if t == "s_endpgm":
            return hashlib.sha256("\n".join(body).encode()).hexdigest()
    return ""


def kernel_isa_stamp() -> str:
    """The stamp the build left next to the product library (rtldavis_amd/csrc/Makefile: librtldavis_hip.stamp =
    isa_sha256 of the demod kernel as compiled for that library); '' when there is none."""
    try:
        with open(STAMP_FILE) as fh:
            return fh.read().strip()
    except OSError:
        return ""


def short(name: str) -> str:
    return name.split("(")[0].replace("void ", "")


def stats(trace_dir: str, out_dir: str, prefix: str = "") -> None:
    """kernel_stats.csv as rocprofv3 wrote it + per-dispatch durations of the demod kernel in launch order."""
    st = glob.glob(trace_dir + "/**/*kernel_stats.csv", recursive=True)
    if st:
        with open(st[0]) as fh, open(os.path.join(out_dir, prefix + "kernel_stats.csv"), "w") as out:
            out.write(fh.read())
        with open(st[0]) as fh:
            for i, line in enumerate(fh):
                if i < 8:
                    print(line.rstrip()[:200])
    tr = glob.glob(trace_dir + "/**/*kernel_trace.csv", recursive=True)
    if not tr:
        return
    rows = list(csv.DictReader(open(tr[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    per = collections.defaultdict(list)
    for r in rows:
        per[short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    with open(os.path.join(out_dir, prefix + "kernel_durations.txt"), "w") as out:
        out.write("per-dispatch durations in launch order (us), from rocprofv3 --kernel-trace\n")
        for name, v in per.items():
            d = [(e - s) / 1e3 for s, e in v]
            if len(d) < 2 or "rocclr" in name:
                continue
            mean = sum(d) / len(d)
            sd = (sum((x - mean) ** 2 for x in d) / len(d)) ** 0.5
            out.write(f"{name}: n={len(d)} mean {mean:.1f} min {min(d):.1f} max {max(d):.1f} std {sd:.1f}\n")
            if "demod" in name or "channelize" in name:
                out.write("  first 12: " + " ".join(f"{x:.0f}" for x in d[:12]) + "\n")
                out.write("  last 12:  " + " ".join(f"{x:.0f}" for x in d[-12:]) + "\n")
                # does the other batch's readback (copyBuffer) overlap the slow ones?
                copies = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows
                          if "copyBuffer" in r["Kernel_Name"]]
                ov = [sum(1 for cs, ce in copies if cs < e and ce > s) for s, e in v]
                slow = [x for x, o in zip(d, ov) if o]
                fast = [x for x, o in zip(d, ov) if not o]
                if slow and fast:
                    out.write(f"  with a copyBuffer dispatch overlapping: n={len(slow)} mean {sum(slow)/len(slow):.1f}; "
                              f"without: n={len(fast)} mean {sum(fast)/len(fast):.1f}\n")


def pmc(pmc_dir: str) -> None:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(pmc_dir + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("kernel,counter,mean_value,n")
    for k, d in agg.items():
        if "rocclr" in k:
            continue
        for c, v in sorted(d.items()):
            print(f"{k},{c},{sum(v)/len(v):.1f},{len(v)}")


def traffic(out_dir: str, commit: str) -> None:
    """FETCH_SIZE / WRITE_SIZE (KiB per dispatch) of the demod kernel -> bytes, with the gfx950 correction
    of MI355X_MICROARCH.md (FETCH_SIZE reports half the bytes of a wide coalesced streaming read)."""
    vals = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        acc = []
        for f in glob.glob(f"{out_dir}/pmc_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "k_demod" in r["Kernel_Name"] and r["Counter_Name"] == c:
                    acc.append(float(r["Counter_Value"]))
        vals[c] = sum(acc) / len(acc) if acc else None
    lib = os.path.join(ROOT, "rtldavis_amd", "librtldavis_hip.so")
    with open(lib, "rb") as fh:
        lib_sha = hashlib.sha256(fh.read()).hexdigest()
    read_b = vals["FETCH_SIZE"] * 1024 * 2 if vals["FETCH_SIZE"] is not None else None
    write_b = vals["WRITE_SIZE"] * 1024 if vals["WRITE_SIZE"] is not None else None
    alg = 4096 * 33 * 8192 * 2
    print(json.dumps({
        "kernel": "k_demod_mfma", "commit": commit, "kernel_isa_sha256": kernel_isa_stamp(), "library_sha256": lib_sha,
        "workload": {"streams": 4096, "blocks": 33, "block_size": 8192},
        "FETCH_SIZE_KiB_per_dispatch": vals["FETCH_SIZE"], "WRITE_SIZE_KiB_per_dispatch": vals["WRITE_SIZE"],
        "read_bytes": read_b, "write_bytes": write_b,
        "traffic_bytes": int(read_b + write_b) if read_b is not None and write_b is not None else None,
        "algorithmic_bytes": alg,
        "traffic_over_algorithmic": round((read_b + write_b) / alg, 4) if read_b is not None and write_b is not None else None,
        "method": "rocprofv3 --pmc, one counter per run; FETCH_SIZE x2 (gfx950: 128-byte requests tallied at 64 B), "
                  "WRITE_SIZE as reported; mean over the dispatches of bench.py --steps 6",
    }, indent=1))


if __name__ == "__main__":
    cmd = sys.argv[1]
    if cmd == "stats":
        stats(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else "")
    elif cmd == "pmc":
        pmc(sys.argv[2])
    elif cmd == "traffic":
        traffic(sys.argv[2], sys.argv[3])
    elif cmd == "isa-stamp":   # isa-stamp <device assembly of rd_demod_mfma.hip>
        with open(sys.argv[2]) as fh:
            st = isa_sha256(fh.read())
        if not st:
            sys.exit("k_demod_mfma not found in " + sys.argv[2])
        print(st)
    elif cmd == "stamp":
        print(kernel_isa_stamp())
