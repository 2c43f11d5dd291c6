#!/usr/bin/env python3
"""Generate tests/golden/ by running the REAL reference (build container only).

Imports /root/reference/src/rtldavis (never copied into this repo, never sent to
the GPU box), runs it on the canonical synthetic inputs of rtldavis_amd.synth and
writes small data fixtures: inputs, expected per-call packets, packed bits,
per-stage float arrays and protocol.Parser.parse() results.

    PYTHONDONTWRITEBYTECODE=1 python3 tools/gen_golden.py

Everything written is data (inputs / expected outputs); no reference source.
"""
from __future__ import annotations

import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("RTLDAVIS_REFERENCE", "/root/reference/src")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

from rtldavis import dsp as ref_dsp  # noqa: E402  (the real reference)
from rtldavis import protocol as ref_protocol  # noqa: E402
from rtldavis_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
B = 8192


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def pack_le(bits: np.ndarray) -> np.ndarray:
    return np.packbits(bits.astype(np.uint8), bitorder="little")


def fnum(x: float):
    """JSON-safe float (NaN/inf as strings)."""
    x = float(x)
    if x != x:
        return "nan"
    if x in (float("inf"), float("-inf")):
        return "inf" if x > 0 else "-inf"
    return x


def run_reference(raw, cfg_kwargs, complex_input=None, with_parse=False):
    """Run dsp.Demodulator call by call.  Returns dict(calls, bits, disc_all, parse)."""
    if with_parse:
        parser = ref_protocol.Parser(symbol_length=cfg_kwargs["symbol_length"])
        assert parser.cfg.block_size == cfg_kwargs["block_size"]
        dem = parser.demodulator
        cfg = parser.cfg
    else:
        cfg = ref_dsp.PacketConfig(**cfg_kwargs)
        dem = ref_dsp.Demodulator(cfg)
        parser = None
    bs = cfg.block_size
    n = (raw.size // 2) if complex_input is None else complex_input.size
    assert n % bs == 0
    calls, bits, disc, parsed = [], [], [], []
    for b in range(n // bs):
        if complex_input is None:
            blk = raw[2 * bs * b: 2 * bs * (b + 1)]
        else:
            blk = complex_input[bs * b: bs * (b + 1)]
        pk = dem.demodulate(blk)
        calls.append([
            {"index": int(p.index), "data": bytes(p.data).hex(), "rssi": fnum(p.rssi), "snr": fnum(p.snr)}
            for p in pk
        ])
        bits.append(dem.quantized[cfg.buffer_length - bs:].copy())
        disc.append(dem.discriminated[bs:].copy())
        if parser is not None:
            # freq_err is not part of Message; recompute exactly as protocol.py:304-311
            msgs = parser.parse(pk)
            rec = []
            for m in msgs:
                idx = m.packet.index
                mean = np.mean(dem.discriminated[idx: idx + cfg.preamble_length])
                fe = -int((mean * float(cfg.sample_rate)) / (2 * np.pi))
                rec.append({"index": int(idx), "id": int(m.id), "sensor_type": m.sensor_type.name,
                            "freq_err": int(fe), "data": bytes(m.packet.data).hex()})
            parsed.append(rec)
    return {"calls": calls, "bits": np.concatenate(bits), "disc": np.concatenate(disc),
            "parse": parsed, "dem": dem, "cfg": cfg}


PROD = dict(bit_rate=19200, symbol_length=14, preamble_symbols=16, packet_symbols=80,
            preamble="1100101110001001", block_size=8192)


def main() -> None:
    os.makedirs(OUT, exist_ok=True)
    manifest = {"numpy": np.__version__, "generator": "tools/gen_golden.py",
                "reference": "2bitoperations/rtldavis src/rtldavis/dsp.py + protocol.py",
                "bit_packing": "LSB-first (np.packbits bitorder=little): sample t -> byte t//8 bit t%8"}

    # ---- 1. canonical streams, seeds 0..63 (bench set); full bits for seeds 0..3
    streams = {}
    bits_full = {}
    for seed in range(64):
        raw = synth.synth_stream(seed)
        r = run_reference(raw, PROD, with_parse=True)
        packed = pack_le(r["bits"])
        streams[str(seed)] = {
            "raw_sha256": sha(raw), "bits_sha256": sha(packed), "payload": synth.payload_of(seed),
            "calls": {str(i): c for i, c in enumerate(r["calls"]) if c},
            "parse": {str(i): c for i, c in enumerate(r["parse"]) if c},
        }
        if seed < 4:
            bits_full[f"seed{seed}"] = packed
        print(f"seed {seed}: {sum(len(c) for c in r['calls'])} packets, "
              f"{sum(len(c) for c in r['parse'])} parsed", flush=True)
    with open(os.path.join(OUT, "streams.json"), "w") as fh:
        json.dump(streams, fh, indent=0, sort_keys=True)
    np.savez(os.path.join(OUT, "streams_bits.npz"), **bits_full)

    # ---- 2. config 1: the seed-0 burst cut (blocks 20..22) as raw bytes
    raw0 = synth.synth_stream(0)
    cut = raw0[2 * B * 20: 2 * B * 23].copy()
    cut.tofile(os.path.join(OUT, "burst_seed0_b20_22.u8"))
    r = run_reference(cut, PROD, with_parse=True)
    dem = r["dem"]
    burst = {"raw_sha256": sha(cut), "calls": r["calls"], "parse": r["parse"],
             "bits_sha256": sha(pack_le(r["bits"]))}
    np.savez(os.path.join(OUT, "burst_seed0_b20_22_state.npz"),
             bits=pack_le(r["bits"]),
             # state after the last call (call 2): newest-block filtered, two-block discriminated
             filtered=dem.filtered.copy(), discriminated=dem.discriminated.copy(),
             quantized=pack_le(dem.quantized), disc_all=r["disc"])
    with open(os.path.join(OUT, "burst_seed0_b20_22.json"), "w") as fh:
        json.dump(burst, fh, indent=1, sort_keys=True)

    # ---- 3. default block size 512 (buffer_length = 4 blocks) on part of the cut
    small = dict(PROD, block_size=512)
    # in the cut the sync word starts at sample 5502; keep 12288 samples from 4096 on
    lo = 4096
    part = cut[2 * lo: 2 * (lo + 12288)].copy()
    r = run_reference(part, small)
    dem = r["dem"]
    np.savez(os.path.join(OUT, "b512_stages.npz"),
             raw=part, bits=pack_le(r["bits"]),
             last_iq=dem.iq.copy(), last_filtered=dem.filtered.copy(),
             last_discriminated=dem.discriminated.copy(), last_quantized=dem.quantized.copy(),
             last_raw_samples=dem.raw_samples[dem.cfg.buffer_length - 512:].copy())
    with open(os.path.join(OUT, "b512_calls.json"), "w") as fh:
        json.dump({"config": small, "calls": r["calls"]}, fh, indent=0)

    # ---- 4. quirk: q_idx == B is emitted twice (call b as B, call b+1 as 0)
    edge = None
    for seed in range(100, 140):
        raw = synth.synth_stream(seed, n_samples=6 * B)
        r = run_reference(raw, PROD)
        true_hex = synth.payload_of(seed)
        hits = [(b, p["index"]) for b, c in enumerate(r["calls"]) for p in c if p["data"] == true_hex]
        if not hits:
            continue
        b, q = hits[0]
        p_abs = (b - 1) * B + q
        rng = np.random.default_rng(seed)
        rng.integers(0, 5)
        drawn = int(rng.integers(B, 6 * B - 120 * 14 - B))
        new_start = drawn + (2 * B - p_abs)
        raw2 = synth.synth_stream(seed, n_samples=6 * B, start=new_start)
        r2 = run_reference(raw2, PROD)
        hits2 = [(bb, p["index"]) for bb, c in enumerate(r2["calls"]) for p in c if p["data"] == true_hex]
        if (2, B) in hits2 and (3, 0) in hits2:
            edge = {"seed": seed, "start": new_start, "n_samples": 6 * B, "raw_sha256": sha(raw2),
                    "calls": r2["calls"], "bits_sha256": sha(pack_le(r2["bits"]))}
            print("edge q==B fixture: seed", seed, "start", new_start, hits2)
            break
    assert edge is not None, "could not craft q_idx == B case"
    with open(os.path.join(OUT, "edge_q_eq_B.json"), "w") as fh:
        json.dump(edge, fh, indent=0)

    # ---- 5. another symbol length / block size (generic PacketConfig)
    alt = dict(PROD, symbol_length=8, block_size=1024)
    raw = synth.synth_stream(7, n_samples=16 * 1024, symbol_length=8, margin=1024)
    r = run_reference(raw, alt)
    with open(os.path.join(OUT, "alt_s8_b1024.json"), "w") as fh:
        json.dump({"config": alt, "seed": 7, "n_samples": 16 * 1024, "margin": 1024, "raw_sha256": sha(raw),
                   "calls": r["calls"], "bits_sha256": sha(pack_le(r["bits"]))}, fh, indent=0)
    np.savez(os.path.join(OUT, "alt_s8_b1024_bits.npz"), bits=pack_le(r["bits"]))

    # ---- 6. complex-input branch (dsp.py:144-150): samples normalised pyrtlsdr-style
    cplx = (cut[0::2].astype(np.float64) - 127.5) / 127.5 + 1j * (cut[1::2].astype(np.float64) - 127.5) / 127.5
    r = run_reference(None, PROD, complex_input=cplx)
    with open(os.path.join(OUT, "complex_input.json"), "w") as fh:
        json.dump({"source": "burst_seed0_b20_22.u8 as (k-127.5)/127.5 complex128",
                   "calls": r["calls"], "bits_sha256": sha(pack_le(r["bits"]))}, fh, indent=0)
    np.savez(os.path.join(OUT, "complex_input_state.npz"), bits=pack_le(r["bits"]),
             discriminated=r["dem"].discriminated.copy(), filtered=r["dem"].filtered.copy())

    # ---- 7. zero-history start-up: first byte pair in each quadrant (signed-zero behaviour)
    quad = {}
    rng = np.random.default_rng(2024)
    for name, (i0, q0) in {"pp": (200, 200), "np": (50, 200), "nn": (50, 50), "pn": (200, 50)}.items():
        raw = rng.integers(0, 256, size=2 * 512, dtype=np.uint8)
        raw[0], raw[1] = i0, q0
        r = run_reference(raw, small)
        quad[name] = {"raw": raw.tobytes().hex(), "bits": pack_le(r["bits"]).tobytes().hex(),
                      "first_disc": [fnum(v) for v in r["disc"][:4]],
                      "first_disc_signbit": [int(np.signbit(v)) for v in r["disc"][:4]]}
    with open(os.path.join(OUT, "startup_quadrants.json"), "w") as fh:
        json.dump(quad, fh, indent=0)

    # ---- 8. the reference's own quantize tests (tests/test_dsp.py:4-33) as data
    q_in = np.array([-5.0, 5.0, -0.1, 0.1, 0.0, -0.0])
    q_out = np.zeros(q_in.size, dtype=np.uint8)
    ref_dsp.quantize(q_in, q_out)
    rq = np.random.default_rng(42).uniform(-10, 10, 1000)
    rq_out = np.zeros(1000, dtype=np.uint8)
    ref_dsp.quantize(rq, rq_out)
    with open(os.path.join(OUT, "quantize.json"), "w") as fh:
        json.dump({"in": [repr(float(v)) for v in q_in], "out": q_out.tolist(),
                   "rng42_out_packed": pack_le(rq_out).tobytes().hex()}, fh)

    # ---- 9. two identical bursts inside one search window: per-call dedupe keeps the first
    #         occurrence in phase-major order (dsp.py:175-186,203-205)
    two = {}
    for seed, gap in ((300, 2100), (301, 1700), (302, 2507)):
        raw = synth.synth_two_bursts(seed, gap)
        r = run_reference(raw, PROD)
        two[str(seed)] = {"gap": gap, "raw_sha256": sha(raw), "calls": r["calls"],
                          "bits_sha256": sha(pack_le(r["bits"]))}
        print("two bursts seed", seed, [[(p["index"], p["data"]) for p in c] for c in r["calls"]])
    with open(os.path.join(OUT, "two_bursts.json"), "w") as fh:
        json.dump(two, fh, indent=0)

    with open(os.path.join(OUT, "manifest.json"), "w") as fh:
        json.dump(manifest, fh, indent=1, sort_keys=True)
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
