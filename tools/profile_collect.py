"""Summaries for tools/profile_round.sh (rocprofv3 csv output -> the small files kept under profiles/)."""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


DEMOD_KERNEL_SYMBOL = b"_Z12k_demod_mfmaILi0ELi1ELi10EE"   # k_demod_mfma<0, 1, RD_MF_PRODUCT_OPT>: the product's variant


def kernel_code_sha256(lib_path: str = None, symbol_prefix: bytes = DEMOD_KERNEL_SYMBOL) -> str:
    """sha256 over the MACHINE CODE of the dominant kernel inside the built library - what bench.py checks a traffic
    file against.  The library's clang offload bundles are walked, the gfx950 code object that defines the kernel is
    parsed as ELF64 and the bytes of the function symbol are hashed: any change to the kernel changes the stamp,
    changes elsewhere (tail kernels, host code, diagnostic variants, comments) do not.  '' when the library or the
    symbol is missing."""
    import struct
    lib_path = lib_path or os.path.join(ROOT, "rtldavis_amd", "librtldavis_hip.so")
    try:
        with open(lib_path, "rb") as fh:
            data = fh.read()
    except OSError:
        return ""
    magic, pos = b"__CLANG_OFFLOAD_BUNDLE__", 0
    while True:
        i = data.find(magic, pos)
        if i < 0:
            return ""
        pos = i + len(magic)
        n = struct.unpack_from("<Q", data, i + 24)[0]
        q = i + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, q)
            q += 24
            triple = data[q:q + tl]
            q += tl
            if b"gfx950" not in triple or size == 0:
                continue
            elf = data[i + off:i + off + size]
            if elf[:4] != b"\x7fELF":
                continue
            shoff, = struct.unpack_from("<Q", elf, 0x28)
            shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
            secs = [struct.unpack_from("<IIQQQQIIQQ", elf, shoff + k * shentsize) for k in range(shnum)]
            for sec in secs:
                if sec[1] != 2:  # SHT_SYMTAB
                    continue
                strtab = secs[sec[6]]
                for k in range(sec[5] // 24):
                    st_name, st_info, _o, st_shndx, st_value, st_size = struct.unpack_from("<IBBHQQ", elf, sec[4] + 24 * k)
                    if (st_info & 0xF) != 2 or st_shndx == 0 or st_shndx >= shnum:  # STT_FUNC, defined
                        continue
                    name_at = strtab[4] + st_name
                    if elf[name_at:name_at + len(symbol_prefix)] != symbol_prefix:
                        continue
                    text = secs[st_shndx]
                    start = text[4] + (st_value - text[3])
                    return hashlib.sha256(elf[start:start + st_size]).hexdigest()


def sources_sha256() -> str:
    """The stamp bench.py checks a traffic file against (the name is historic: since round 3 it is the hash of the
    demod kernel's machine code in the built product library, see kernel_code_sha256)."""
    return kernel_code_sha256()


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
        "kernel": "k_demod_mfma", "commit": commit, "kernel_code_sha256": kernel_code_sha256(), "library_sha256": lib_sha,
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
    elif cmd == "stamp":
        print(sources_sha256())
