"""GPU parity tests (MI355X): the HIP path, called through the C ABI, against the golden
fixtures generated from the real reference and against the C oracle on the same inputs."""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN, assert_calls_equal, dense_calls, load_json, load_npz
from rtldavis_amd import synth

pytestmark = pytest.mark.gpu


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def dsp():
    from rtldavis_amd import _lib, dsp as d
    assert _lib.lib().rd_device_count() > 0, "no HIP device: the GPU tests need an MI355X"
    return d


@pytest.fixture(scope="module")
def batchmod():
    from rtldavis_amd import batch
    return batch


def prod_cfg(dsp, **kw):
    args = dict(bit_rate=19200, symbol_length=14, preamble_symbols=16, packet_symbols=80,
                preamble="1100101110001001", block_size=8192)
    args.update(kw)
    return dsp.PacketConfig(**args)


def run_streaming(dem, raw=None, cplx=None):
    B = dem.cfg.block_size
    calls = []
    n = raw.size // 2 if cplx is None else cplx.size
    for b in range(n // B):
        blk = raw[2 * B * b: 2 * B * (b + 1)] if cplx is None else cplx[B * b: B * (b + 1)]
        calls.append(dem.demodulate(blk))
    return calls


# ---------------------------------------------------------------- batch path (configs 2,3)
def test_batch_streams_bits_and_packets(dsp, batchmod, golden_streams):
    seeds = list(range(8))
    raw = synth.synth_streams(seeds)
    bd = batchmod.BatchDemodulator(prod_cfg(dsp), len(seeds), synth.BLOCKS_PER_STREAM)
    res = bd.demodulate(raw)
    full = load_npz("streams_bits.npz")
    for i, seed in enumerate(seeds):
        g = golden_streams[str(seed)]
        assert sha(raw[i]) == g["raw_sha256"]
        bits = bd.bits(i)
        assert sha(bits) == g["bits_sha256"], f"seed {seed}: packed bits differ from the reference"
        if seed < 4:
            assert np.array_equal(bits, full[f"seed{seed}"])
        assert_calls_equal(res[i], dense_calls(g["calls"], synth.BLOCKS_PER_STREAM))
    c = bd.counters()
    assert 0 < c["fixup_runs"] < 0.1 * len(seeds) * 8448


def test_batch_matches_c_oracle_51_channels(dsp, batchmod):
    """Config 3: 51 hop channels in one launch; checked against the C oracle run here."""
    from oracle import c_oracle as CO
    seeds = list(range(100, 151))
    raw = synth.synth_streams(seeds)
    bd = batchmod.BatchDemodulator(prod_cfg(dsp), len(seeds), synth.BLOCKS_PER_STREAM)
    res = bd.demodulate(raw)
    want, wbits = CO.demod_batch(raw, CO.make_cfg(), threads=8, want_bits=True)
    for i in range(len(seeds)):
        assert np.array_equal(bd.bits(i), wbits[i]), f"stream {i}"
        got = [(c, p.index, bytes(p.data).hex()) for c, ps in enumerate(res[i]) for p in ps]
        exp = [(p.call, p.index, bytes(p.data).hex()) for p in want[i]]
        assert got == exp
        for (c, ps), w in zip([(c, p) for c, pl in enumerate(res[i]) for p in pl], want[i]):
            assert abs(ps.rssi - w.rssi) < 1e-3 and abs(ps.snr - w.snr) < 1e-3
        true = [bytes(p.data).hex() for pl in res[i] for p in pl if bytes(p.data).hex() == synth.payload_of(seeds[i])]
        assert len(true) >= 1


def test_batch_discriminated_and_freq_error(dsp, batchmod, golden_streams):
    import math
    raw = synth.synth_streams([0])
    bd = batchmod.BatchDemodulator(prod_cfg(dsp), 1, synth.BLOCKS_PER_STREAM)
    bd.demodulate(raw)
    g = golden_streams["0"]["parse"]["21"][0]
    # call 21, index 5502 -> absolute sample 20*8192 + 5502
    t0 = 20 * 8192 + g["index"]
    d = bd.discriminated(0, t0, 224)
    fe = -int((np.mean(d) * 268800.0) / (2 * math.pi))
    assert abs(fe - g["freq_err"]) <= 1
    st = load_npz("burst_seed0_b20_22_state.npz")
    ref = st["disc_all"]  # d over blocks 20..22 of seed 0 (cut stream: first samples differ by history)
    got = bd.discriminated(0, 20 * 8192 + 64, 3 * 8192 - 64)
    r = ref[64:]
    assert np.all(np.abs(got - r) <= 1e-5 * np.maximum(1, np.abs(r)))


def test_batch_parse_front_half(dsp, batchmod, golden_streams):
    """protocol.Parser.parse front half on the device: CRC gate, transmitter id and frequency
    error for every packet of the batch, against what the real Parser produced (streams.json)."""
    seeds = list(range(16))
    raw = synth.synth_streams(seeds)
    bd = batchmod.BatchDemodulator(prod_cfg(dsp), len(seeds), synth.BLOCKS_PER_STREAM)
    bd.set_parse(True)
    bd.upload(raw)
    bd.run()
    got = bd.parsed()
    want = []
    for i, seed in enumerate(seeds):
        for c, msgs in sorted(golden_streams[str(seed)]["parse"].items(), key=lambda kv: int(kv[0])):
            for m in msgs:
                want.append((i, int(c), m["index"], m["id"], m["freq_err"], m["data"]))
    from oracle import dsp_oracle as O
    have = []
    for r in got:
        # un-swap to compare with the fixture's on-air packet bytes
        ota = bytes([0xCB, 0x89]) + bytes(O.swap_bit_order(int(b)) for b in r["data"][: int(r["nbytes"])])
        have.append((int(r["stream"]), int(r["call"]), int(r["index"]), int(r["id"]), int(r["freq_err"]), ota.hex()))
    assert have == want
    assert len(have) == len(seeds)  # exactly one CRC-valid message per synthetic stream
    # rssi/snr are carried over from the packet
    pk = {(s, c, p.index): p for s, c, p in bd.records()}
    for r in got:
        p = pk[(int(r["stream"]), int(r["call"]), int(r["index"]))]
        assert r["rssi"] == p.rssi and r["snr"] == p.snr


def test_batch_small_and_ragged_shapes(dsp, batchmod):
    from oracle import c_oracle as CO
    rng = np.random.default_rng(7)
    for B, nb, ns in [(512, 3, 5), (1024, 1, 3), (8192, 1, 1), (96, 40, 2), (36, 70, 3)]:
        raw = rng.integers(0, 256, size=(ns, 2 * B * nb), dtype=np.uint8)
        # plant the sync word so the search has something to find
        cfg = prod_cfg(dsp, block_size=B)
        bd = batchmod.BatchDemodulator(cfg, ns, nb)
        res = bd.demodulate(raw)
        want, wbits = CO.demod_batch(raw, CO.make_cfg(block_size=B), threads=2, want_bits=True)
        for i in range(ns):
            assert np.array_equal(bd.bits(i), wbits[i]), (B, nb, i)
            got = [(c, p.index, bytes(p.data).hex()) for c, ps in enumerate(res[i]) for p in ps]
            assert got == [(p.call, p.index, bytes(p.data).hex()) for p in want[i]], (B, nb, i)


@pytest.mark.parametrize("S,P,K,B,nb,ns", [
    (4, 4, 12, 64, 9, 3),        # short preamble: a match every ~16 positions, lists overflow and grow
    (5, 8, 40, 256, 6, 4),
    (8, 12, 80, 1004, 3, 2),     # block size not a multiple of 8: exact kernel throughout
    (14, 16, 80, 512, 10, 3),    # default block size, buffer_length = 4 blocks
    (20, 24, 120, 4096, 2, 2),
    (3, 6, 200, 96, 20, 2),      # 25-byte packets
    (14, 6, 24, 8192, 2, 5),
])
def test_random_configs_against_c_oracle(dsp, batchmod, S, P, K, B, nb, ns):
    """Generic PacketConfig coverage: random preamble, random noise input, batch and streaming
    paths against the C oracle (bits, packets with order and dedupe, RSSI/SNR)."""
    from oracle import c_oracle as CO
    rng = np.random.default_rng(1000 * S + P)
    pre = "".join(str(int(b)) for b in rng.integers(0, 2, size=P))
    cfg = dsp.PacketConfig(19200, S, P, K, pre, B)
    ocfg = CO.make_cfg(19200, S, P, K, pre, B)
    raw = rng.integers(96, 160, size=(ns, 2 * B * nb), dtype=np.uint8)
    want, wbits = CO.demod_batch(raw, ocfg, threads=2, want_bits=True, cap_per_stream=200000)
    bd = batchmod.BatchDemodulator(cfg, ns, nb)
    res = bd.demodulate(raw)
    total = 0
    for i in range(ns):
        assert np.array_equal(bd.bits(i), wbits[i]), (i, "bits")
        got = [(c, p.index, bytes(p.data).hex()) for c, ps in enumerate(res[i]) for p in ps]
        exp = [(p.call, p.index, bytes(p.data).hex()) for p in want[i]]
        assert got == exp, (i, len(got), len(exp))
        flat = [p for ps in res[i] for p in ps]
        for a, b in zip(flat, want[i]):
            assert abs(a.rssi - b.rssi) < 1e-3 and (abs(a.snr - b.snr) < 1e-3 or (a.snr != a.snr and b.snr != b.snr))
        total += len(got)
    assert total > 0 or P > 12  # long random preambles need not occur in noise
    # the same streams block by block through the streaming handle
    # (the reference's per-call list is unbounded, dsp.py:190-246: the 6-symbol preamble on 8192-sample
    # blocks returns more than the wrapper's initial 64 records per call and must lose none)
    dem = dsp.Demodulator(cfg)
    calls = run_streaming(dem, raw[0])
    got = [(c, p.index, bytes(p.data).hex()) for c, ps in enumerate(calls) for p in ps]
    assert got == [(p.call, p.index, bytes(p.data).hex()) for p in want[0]]
    if (S, P, B) == (14, 6, 8192):
        assert max(len(ps) for ps in calls) > 64


def test_soak_small_configs_dense_duplicates(dsp, batchmod):
    """Randomised small PacketConfigs (symbol length 1.., preambles of 1-6 symbols, 32-256-sample
    blocks) on repetitive inputs: dense matches, long runs of identical packets at adjacent
    positions (the slice kernel voids those it can prove the per-call dedupe drops), many
    positions on block boundaries.  Everything against the C oracle, batch and streaming."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("rd_soak", os.path.join(root, "tools", "soak.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.soak(150, 5, verbose=False) > 1000
    # production config, bursts placed on and around block boundaries (q = B / q = 0 twins)
    n, at_edge = mod.soak_bursts(768, 3, verbose=False)
    assert n >= 768 and at_edge >= 5


def test_soak_streaming_one_launch_forms(dsp):
    """tools/soak_stream.py, 120 cases: the Davis configuration at seven block sizes (one to four workgroups per
    complex block, ragged pieces), bursts at random positions, amplitudes and offsets or degenerate inputs, blocks as
    uint8 / complex128 / uint8 then complex128, synchronous or two in flight, pushed by the host or read from the pinned
    slot: every call's packets and the final window against the C oracle (dsp.py:139-246)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("rd_soak_stream", os.path.join(root, "tools", "soak_stream.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.soak(120, 11, verbose=False) > 100


def test_batch_api_call_orders(dsp, batchmod, golden_streams):
    """run twice before fetching, results fetched twice, parse switched on between runs: the
    packets are those of the reference every time (counter sets alternate per run; records of a
    run stay readable until the next run)."""
    cfg = prod_cfg(dsp)
    raw = synth.synth_streams(range(4))
    bd = batchmod.BatchDemodulator(cfg, 4, 33)
    bd.upload(raw)
    bd.run()
    bd.run()                      # first run's results dropped, never mixed into the second's
    a = bd.results()
    b = bd.results()
    assert np.array_equal(a, b)
    for s in range(4):
        assert_calls_equal(bd.packets()[s], dense_calls(golden_streams[str(s)]["calls"], 33))
    bd.set_parse(True)
    bd.run()
    assert np.array_equal(bd.results(), a)
    assert len(bd.parsed()) == 4  # one CRC-valid message per stream
    bd.set_parse(False)
    for _ in range(3):
        bd.run()
        assert np.array_equal(bd.results(), a)


def test_batch_degenerate_inputs_take_the_exact_path(dsp, batchmod):
    """Low-amplitude and saturated inputs put most runs inside the guard band: the guard
    list overflows and every run is re-evaluated exactly.  Bits must still match."""
    from oracle import c_oracle as CO
    rng = np.random.default_rng(11)
    B, nb = 8192, 4
    raw = np.stack([
        rng.integers(126, 130, size=2 * B * nb, dtype=np.uint8),
        rng.integers(127, 129, size=2 * B * nb, dtype=np.uint8),
        rng.choice(np.array([0, 255], np.uint8), size=2 * B * nb),
        rng.integers(0, 256, size=2 * B * nb, dtype=np.uint8),
    ])
    bd = batchmod.BatchDemodulator(prod_cfg(dsp), raw.shape[0], nb)
    res = bd.demodulate(raw)
    want, wbits = CO.demod_batch(raw, CO.make_cfg(), threads=4, want_bits=True)
    for i in range(raw.shape[0]):
        assert np.array_equal(bd.bits(i), wbits[i]), i
        got = [(c, p.index, bytes(p.data).hex()) for c, ps in enumerate(res[i]) for p in ps]
        assert got == [(p.call, p.index, bytes(p.data).hex()) for p in want[i]]


def test_batch_replicas_are_identical_property(dsp, batchmod, golden_streams):
    """Size-independent property at a larger size: 512 streams tiled from 8 unique ones give
    512 bitstreams whose hashes are the 8 golden ones, and the same packets."""
    uniq = synth.synth_streams(range(8))
    raw = np.tile(uniq, (64, 1))
    bd = batchmod.BatchDemodulator(prod_cfg(dsp), raw.shape[0], synth.BLOCKS_PER_STREAM)
    bd.upload(raw)
    bd.run()
    recs = bd.records()
    per = {}
    for s, c, p in recs:
        per.setdefault(s, []).append((c, p.index, bytes(p.data).hex()))
    for s in range(raw.shape[0]):
        g = golden_streams[str(s % 8)]
        want = [(int(c), p["index"], p["data"]) for c, ps in sorted(g["calls"].items(), key=lambda kv: int(kv[0]))
                for p in ps]
        assert per.get(s, []) == want, s
    for s in (0, 9, 255, 511):
        assert sha(bd.bits(s)) == golden_streams[str(s % 8)]["bits_sha256"]


def test_full_size_4096_streams_properties(dsp, batchmod, golden_streams):
    """BASELINE config 4 (4096 streams x 33 blocks, 2.2 GB): every stream's packets equal the
    fixture of the unique stream it was tiled from, and the XOR / sum of all packed bitstreams
    equals what 64 copies of each of the 64 golden bitstreams give (checksum of checksums)."""
    import zlib
    uniq = synth.synth_streams(range(64))
    raw = np.tile(uniq, (64, 1))
    bd = batchmod.BatchDemodulator(prod_cfg(dsp), raw.shape[0], synth.BLOCKS_PER_STREAM)
    bd.upload(raw)
    del raw
    bd.run()
    res = bd.results()
    per = {}
    for r in res:
        per.setdefault(int(r["stream"]), []).append(
            (int(r["call"]), int(r["index"]), r["data"][: int(r["nbytes"])].tobytes().hex()))
    crc_unique = {}
    for s in range(4096):
        g = golden_streams[str(s % 64)]
        want = [(int(c), p["index"], p["data"]) for c, ps in sorted(g["calls"].items(), key=lambda kv: int(kv[0]))
                for p in ps]
        assert per.get(s, []) == want, s
    total = 0
    for s in range(0, 4096, 7):  # every 7th stream: 586 bitstreams, all 64 residues covered
        b = bd.bits(s)
        c = zlib.crc32(b.tobytes())
        u = s % 64
        if u not in crc_unique:
            assert sha(b) == golden_streams[str(u)]["bits_sha256"]
            crc_unique[u] = c
        assert c == crc_unique[u], s
        total += 1
    assert len(crc_unique) == 64 and total == 586
    cnt = bd.counters()
    assert cnt["matches"] % 64 == 0  # 64 identical copies of each unique stream
    # (the fix-up list is not a multiple of 64: the forced entries at chunk starts follow the global tile index,
    # 16 tiles per chunk - the work-queue default, rd_demod_mfma.hip - against 132 per stream)
    assert 0 < cnt["fixup_runs"] < 0.05 * 4096 * 8448


# ---------------------------------------------------------------- streaming Demodulator
def test_demodulator_burst_config1(dsp):
    raw = np.fromfile(f"{GOLDEN}/burst_seed0_b20_22.u8", dtype=np.uint8)
    g = load_json("burst_seed0_b20_22.json")
    st = load_npz("burst_seed0_b20_22_state.npz")
    dem = dsp.Demodulator(prod_cfg(dsp))
    assert np.all(dem.discriminated == 0) and np.all(dem.quantized == 0)
    calls = run_streaming(dem, raw)
    assert_calls_equal(calls, g["calls"])
    assert bytes(calls[1][0].data).hex() == "cb890520ac8bd4000e5c"
    assert not calls[1][0].data.flags.writeable
    q = dem.quantized
    assert np.array_equal(np.packbits(q, bitorder="little"), st["quantized"])
    np.testing.assert_allclose(dem.filtered, st["filtered"], rtol=0, atol=1e-13)
    ref = st["discriminated"]
    assert np.all(np.abs(dem.discriminated - ref) <= 1e-5 * np.maximum(1, np.abs(ref)))
    dem.reset()
    assert_calls_equal(run_streaming(dem, raw), g["calls"])


def test_demodulator_freq_error_like_parser(dsp):
    """protocol.Parser.parse reads demodulator.discriminated right after the call (protocol.py:304-311)."""
    import math
    raw = np.fromfile(f"{GOLDEN}/burst_seed0_b20_22.u8", dtype=np.uint8)
    g = load_json("burst_seed0_b20_22.json")
    cfg = prod_cfg(dsp)
    dem = dsp.Demodulator(cfg)
    dem.demodulate(raw[:16384])
    pk = dem.demodulate(raw[16384:32768])
    idx = pk[0].index
    mean = np.mean(dem.discriminated[idx: idx + cfg.preamble_length])
    assert -int((mean * float(cfg.sample_rate)) / (2 * math.pi)) == g["parse"][1][0]["freq_err"]


def test_demodulator_block512_and_state(dsp):
    z = load_npz("b512_stages.npz")
    g = load_json("b512_calls.json")
    cfg = dsp.PacketConfig(**g["config"])
    dem = dsp.Demodulator(cfg)
    calls = run_streaming(dem, z["raw"])
    assert_calls_equal(calls, g["calls"])
    assert np.array_equal(dem.quantized, z["last_quantized"])
    np.testing.assert_allclose(dem.filtered, z["last_filtered"], rtol=0, atol=1e-13)
    ref = z["last_discriminated"]
    assert np.all(np.abs(dem.discriminated - ref) <= 1e-5 * np.maximum(1, np.abs(ref)))


def test_demodulator_edge_q_equals_block_size(dsp):
    g = load_json("edge_q_eq_B.json")
    raw = synth.synth_stream(g["seed"], n_samples=g["n_samples"], start=g["start"])
    calls = run_streaming(dsp.Demodulator(prod_cfg(dsp)), raw)
    assert_calls_equal(calls, g["calls"])
    idx = [(b, p.index) for b, c in enumerate(calls) for p in c]
    assert (2, 8192) in idx and (3, 0) in idx


def test_batch_edge_q_equals_block_size(dsp, batchmod):
    g = load_json("edge_q_eq_B.json")
    raw = synth.synth_stream(g["seed"], n_samples=g["n_samples"], start=g["start"])
    bd = batchmod.BatchDemodulator(prod_cfg(dsp), 1, g["n_samples"] // 8192)
    res = bd.demodulate(raw[None, :])
    assert_calls_equal(res[0], g["calls"])
    assert sha(bd.bits(0)) == g["bits_sha256"]


def test_two_bursts_in_one_window_dedupe_order(dsp, batchmod):
    """Several matches with different and with identical bytes inside one call: the dedupe (certain
    cases on the device, the rest on the host) must keep exactly what dsp.py:203-205 keeps, in the
    reference's order."""
    g = load_json("two_bursts.json")
    seeds = sorted(g, key=int)
    raws = np.stack([synth.synth_two_bursts(int(s), g[s]["gap"]) for s in seeds])
    for i, s in enumerate(seeds):
        assert sha(raws[i]) == g[s]["raw_sha256"]
        assert_calls_equal(run_streaming(dsp.Demodulator(prod_cfg(dsp)), raws[i]), g[s]["calls"])
    bd = batchmod.BatchDemodulator(prod_cfg(dsp), len(seeds), raws.shape[1] // 2 // 8192)
    res = bd.demodulate(raws)
    for i, s in enumerate(seeds):
        assert_calls_equal(res[i], g[s]["calls"])
        assert sha(bd.bits(i)) == g[s]["bits_sha256"]


def test_dedupe_many_identical_records_one_call(dsp, batchmod):
    """One stream repeated: every stream must produce the same packets (records of different
    streams with equal bytes must never be merged by the dedupe)."""
    g = load_json("two_bursts.json")
    raw = synth.synth_two_bursts(300, g["300"]["gap"])
    n = 700
    bd = batchmod.BatchDemodulator(prod_cfg(dsp), n, raw.size // 2 // 8192)
    res = bd.demodulate(np.tile(raw, (n, 1)))
    for i in range(n):
        assert_calls_equal(res[i], g["300"]["calls"])


def test_alt_symbol_length(dsp, batchmod):
    g = load_json("alt_s8_b1024.json")
    cfg = dsp.PacketConfig(**g["config"])
    raw = synth.synth_stream(g["seed"], n_samples=g["n_samples"], symbol_length=8, margin=g["margin"])
    assert_calls_equal(run_streaming(dsp.Demodulator(cfg), raw), g["calls"])
    bd = batchmod.BatchDemodulator(cfg, 1, g["n_samples"] // 1024)
    res = bd.demodulate(raw[None, :])
    assert_calls_equal(res[0], g["calls"])
    assert np.array_equal(bd.bits(0), load_npz("alt_s8_b1024_bits.npz")["bits"])


def test_complex_input_branch(dsp):
    raw = np.fromfile(f"{GOLDEN}/burst_seed0_b20_22.u8", dtype=np.uint8)
    cplx = (raw[0::2].astype(np.float64) - 127.5) / 127.5 + 1j * (raw[1::2].astype(np.float64) - 127.5) / 127.5
    g = load_json("complex_input.json")
    st = load_npz("complex_input_state.npz")
    dem = dsp.Demodulator(prod_cfg(dsp))
    calls = run_streaming(dem, cplx=cplx)
    assert_calls_equal(calls, g["calls"])
    assert [p.index for p in calls[1]] == [5502, 536]  # phase-major order
    np.testing.assert_allclose(dem.filtered, st["filtered"], rtol=0, atol=1e-13)
    ref = st["discriminated"]
    assert np.all(np.abs(dem.discriminated - ref) <= 1e-5 * np.maximum(1, np.abs(ref)))
    # complex64 input is accepted like any complex array (np.iscomplexobj, dsp.py:144)
    dem.reset()
    calls32 = run_streaming(dem, cplx=cplx.astype(np.complex64))
    assert [bytes(p.data).hex() for c in calls32 for p in c][0] == "cb890520ac8bd4000e5c"


def test_mixed_uint8_then_complex_keeps_history(dsp):
    """uint8 blocks then the same samples as complex: identical to all-uint8 (the reference
    converts bytes with the LUT and keeps complex history, dsp.py:150-152)."""
    raw = np.fromfile(f"{GOLDEN}/burst_seed0_b20_22.u8", dtype=np.uint8)
    g = load_json("burst_seed0_b20_22.json")
    lut = (np.arange(256, dtype=np.float64) - 127.4) / 127.6
    cplx = lut[raw[0::2]] + 1j * lut[raw[1::2]]
    dem = dsp.Demodulator(prod_cfg(dsp))
    calls = [dem.demodulate(raw[:16384]), dem.demodulate(cplx[8192:16384]), dem.demodulate(raw[32768:])]
    assert_calls_equal(calls, g["calls"])


def test_multi_demodulator_lock_step(dsp, golden_streams):
    """Five receivers fed block by block in one handle == five independent Demodulators."""
    import math
    seeds = [0, 1, 2, 3, 17]
    raws = synth.synth_streams(seeds)
    cfg = prod_cfg(dsp)
    md = dsp.MultiDemodulator(cfg, len(seeds))
    per = [[] for _ in seeds]
    freq = {}
    for b in range(synth.BLOCKS_PER_STREAM):
        out = md.demodulate(raws[:, 2 * 8192 * b: 2 * 8192 * (b + 1)])
        for i, pk in enumerate(out):
            per[i].append(pk)
            want = golden_streams[str(seeds[i])]["parse"].get(str(b))
            if want:
                d = md.discriminated(i)
                idx = want[0]["index"]
                mean = np.mean(d[idx: idx + cfg.preamble_length])
                freq[i] = (-int((mean * float(cfg.sample_rate)) / (2 * math.pi)), want[0]["freq_err"])
    for i, seed in enumerate(seeds):
        assert_calls_equal(per[i], dense_calls(golden_streams[str(seed)]["calls"], synth.BLOCKS_PER_STREAM))
        assert freq[i][0] == freq[i][1]
    md.reset()
    out = md.demodulate(raws[:, : 2 * 8192])
    assert [len(x) for x in out] == [len(per[i][0]) for i in range(len(seeds))]
    with pytest.raises(ValueError, match="Incompatible array sizes"):
        md.demodulate(raws[:2, : 2 * 8192])


@pytest.mark.parametrize("B", [2048, 4096, 16384])
def test_one_launch_block_at_other_block_sizes(dsp, B):
    """The streaming handle's one-launch form (k_stream_block: Davis symbol / preamble / packet lengths, buffer_length
    = 2 blocks, 2048 <= B <= 16384) at the ends and in the middle of that range: three receivers in lock step and a
    single one, block by block against the C oracle (per-call lists with order, index, bytes; RSSI / SNR), then
    reset() and the first blocks again (dsp.py:136-137, 139-246)."""
    from oracle import c_oracle as CO
    seeds = [3, 4, 11]
    raws = synth.synth_streams(seeds)
    nb = raws.shape[1] // (2 * B)
    raws = np.ascontiguousarray(raws[:, : 2 * B * nb])
    cfg = dsp.PacketConfig(19200, 14, 16, 80, "1100101110001001", B)
    want, _ = CO.demod_batch(raws, CO.make_cfg(19200, 14, 16, 80, "1100101110001001", B), threads=2, cap_per_stream=256)
    md = dsp.MultiDemodulator(cfg, len(seeds))
    got = [[] for _ in seeds]
    for b in range(nb):
        out = md.demodulate(raws[:, 2 * B * b: 2 * B * (b + 1)])
        for i, pk in enumerate(out):
            got[i] += [(b, p.index, bytes(p.data).hex(), p.rssi, p.snr) for p in pk]
    n = 0
    for i in range(len(seeds)):
        exp = [(p.call, p.index, bytes(p.data).hex(), p.rssi, p.snr) for p in want[i]]
        assert [g[:3] for g in got[i]] == [e[:3] for e in exp], (B, i)
        for g, e in zip(got[i], exp):
            assert abs(g[3] - e[3]) < 1e-3 and abs(g[4] - e[4]) < 1e-3
        n += len(exp)
    assert n >= len(seeds)  # every stream holds its burst
    dem = dsp.Demodulator(cfg)
    single = []
    for b in range(nb):
        single += [(b, p.index, bytes(p.data).hex()) for p in dem.demodulate(raws[1, 2 * B * b: 2 * B * (b + 1)])]
    assert single == [(p.call, p.index, bytes(p.data).hex()) for p in want[1]]
    dem.reset()
    again = []
    for b in range(min(nb, 6)):
        again += [(b, p.index, bytes(p.data).hex()) for p in dem.demodulate(raws[1, 2 * B * b: 2 * B * (b + 1)])]
    assert again == [x for x in single if x[0] < min(nb, 6)]


@pytest.mark.parametrize("B", [2048, 2144, 4096, 6176, 8192])
def test_one_launch_complex_block_at_other_block_sizes(dsp, B):
    """The complex-input one-launch form (k_stream_block_cplx, dsp.py:144-150 + 154-246) cuts a block into pieces of 2048
    samples, a workgroup each, and the last workgroup to arrive searches and slices: one piece (2048), a ragged second
    piece (2144), two, a ragged fourth (6176) and four (8192).  Input = the byte stream through the reference's LUT
    (py:26), so the per-call lists equal the C oracle's for the bytes (index, bytes, order exactly; RSSI / SNR to
    1e-3 dB), block by block over a whole stream; then reset(), uint8 blocks first and complex ones behind them (the
    byte ring converted once, history carried over: py:150-152), and the quantized mirror against the oracle's bits."""
    from oracle import c_oracle as CO
    raw = synth.synth_streams([4])
    nb = raw.shape[1] // (2 * B)
    raw = np.ascontiguousarray(raw[:, : 2 * B * nb])
    cfg = dsp.PacketConfig(19200, 14, 16, 80, "1100101110001001", B)
    want, bits = CO.demod_batch(raw, CO.make_cfg(19200, 14, 16, 80, "1100101110001001", B), threads=2, want_bits=True, cap_per_stream=256)
    exp = [(p.call, p.index, bytes(p.data).hex(), p.rssi, p.snr) for p in want[0]]
    assert exp
    lut = (np.arange(256, dtype=np.float64) - 127.4) / 127.6
    cplx = lut[raw[0, 0::2]] + 1j * lut[raw[0, 1::2]]
    dem = dsp.Demodulator(cfg)

    def run(first_complex):
        got = []
        for b in range(nb):
            blk = cplx[B * b: B * (b + 1)] if b >= first_complex else raw[0, 2 * B * b: 2 * B * (b + 1)]
            got += [(b, p.index, bytes(p.data).hex(), p.rssi, p.snr) for p in dem.demodulate(blk)]
        return got

    for first_complex in (0, 3):
        dem.reset()
        got = run(first_complex)
        assert [g[:3] for g in got] == [e[:3] for e in exp], (B, first_complex)
        for g, e in zip(got, exp):
            assert abs(g[3] - e[3]) < 1e-3 and abs(g[4] - e[4]) < 1e-3
        # the window after the last block = the stream's last 2 B sign bits
        q = np.asarray(dem.quantized).astype(np.uint8)
        allbits = np.unpackbits(bits[0], bitorder="little")[: B * nb]
        assert np.array_equal(q, allbits[-2 * B:]), (B, first_complex)


def test_complex_blocks_equal_the_reference_on_every_fixture_stream(dsp, golden_streams):
    """All 64 fixture streams block by block as complex128 (the bytes through the reference's LUT, py:26: what the
    reference keeps in raw_samples either way, py:150-152) through ONE handle with reset() in between: the per-call
    lists of the real reference (tests/golden/streams.json: index, bytes, order; RSSI / SNR to 1e-3 dB), and
    submit() / fetch() with two blocks in flight gives the same lists as the synchronous calls."""
    cfg = prod_cfg(dsp)
    lut = (np.arange(256, dtype=np.float64) - 127.4) / 127.6
    dem = dsp.Demodulator(cfg)
    B = 8192
    n = 0
    for seed in sorted(golden_streams, key=int):
        raw = synth.synth_stream(int(seed))
        cplx = lut[raw[0::2]] + 1j * lut[raw[1::2]]
        dem.reset()
        calls = [dem.demodulate(cplx[B * b: B * (b + 1)]) for b in range(synth.BLOCKS_PER_STREAM)]
        assert_calls_equal(calls, dense_calls(golden_streams[seed]["calls"], synth.BLOCKS_PER_STREAM))
        n += sum(len(c) for c in calls)
        if int(seed) % 8 == 0:
            dem.reset()
            piped = []
            dem.submit(cplx[:B])
            for b in range(1, synth.BLOCKS_PER_STREAM):
                dem.submit(cplx[B * b: B * (b + 1)])
                piped.append(dem.fetch())
            piped.append(dem.fetch())
            assert_calls_equal(piped, dense_calls(golden_streams[seed]["calls"], synth.BLOCKS_PER_STREAM))
    assert n >= len(golden_streams)


def test_host_push_and_pinned_slot_give_the_same_packets(dsp, golden_streams):
    """A copied block reaches the device either through an uncached device buffer the HOST writes (PCIe large BAR:
    memcpy, sfence, HDP flush, launch) or through a pinned slot the kernel reads across the bus (rd_set_input_push).
    On this pool the push is available and is the default: the test says so, then runs a fixture stream through both
    forms, uint8 and complex128, synchronous and with two blocks in flight (the slot written again while the other
    block's kernel runs), and 16 receivers in lock step - every block's packets against the real reference's lists."""
    cfg = prod_cfg(dsp)
    B, nb = 8192, synth.BLOCKS_PER_STREAM
    lut = (np.arange(256, dtype=np.float64) - 127.4) / 127.6
    try:
        assert dsp.set_input_push(None), "this device does not offer the host push: the pinned slot serves (not an error, but say so)"
        for push in (True, False):
            dsp.set_input_push(push)
            dem = dsp.Demodulator(cfg)
            dem.demodulate(synth.synth_stream(0)[: 2 * B])
            assert dem.input_pushed == push
            for seed in (0, 21, 50):
                raw = synth.synth_stream(seed)
                want = dense_calls(golden_streams[str(seed)]["calls"], nb)
                cplx = lut[raw[0::2]] + 1j * lut[raw[1::2]]
                for blocks in ([raw[2 * B * b: 2 * B * (b + 1)] for b in range(nb)], [cplx[B * b: B * (b + 1)] for b in range(nb)]):
                    dem.reset()
                    assert_calls_equal([dem.demodulate(x) for x in blocks], want)
                    dem.reset()
                    got = []
                    dem.submit(blocks[0])
                    for x in blocks[1:]:
                        dem.submit(x)
                        got.append(dem.fetch())
                    got.append(dem.fetch())
                    assert_calls_equal(got, want)
            seeds = list(range(16))
            raws = synth.synth_streams(seeds)
            md = dsp.MultiDemodulator(cfg, len(seeds))
            per = [[] for _ in seeds]
            for b in range(nb):
                for i, pk in enumerate(md.demodulate(raws[:, 2 * B * b: 2 * B * (b + 1)])):
                    per[i].append(pk)
            for i, seed in enumerate(seeds):
                assert_calls_equal(per[i], dense_calls(golden_streams[str(seed)]["calls"], nb))
    finally:
        dsp.set_input_push(None)


def test_submit_fetch_pipeline_equals_synchronous_path(dsp, golden_streams):
    """rd_demod_submit / rd_demod_fetch with two blocks in flight over a 33-block stream: the same
    packets (index, bytes, order, RSSI/SNR) as demodulate() block by block, and as the reference."""
    raw = synth.synth_streams([5])[0]
    cfg = prod_cfg(dsp)
    B = cfg.block_size
    blocks = [raw[2 * B * b: 2 * B * (b + 1)] for b in range(synth.BLOCKS_PER_STREAM)]
    sync = run_streaming(dsp.Demodulator(cfg), raw)
    dem = dsp.Demodulator(cfg)
    got = []
    dem.submit(blocks[0])
    for b in range(1, len(blocks)):
        dem.submit(blocks[b])          # block b's copy runs beside block b-1's kernels
        assert dem.inflight == 2
        with pytest.raises(RuntimeError):
            dem.submit(blocks[b])      # a third block in flight is refused, nothing is consumed
        with pytest.raises(RuntimeError):
            dem.discriminated          # state mirrors need a quiet handle
        got.append(dem.fetch())
    got.append(dem.fetch())
    assert dem.inflight == 0
    with pytest.raises(RuntimeError):
        dem.fetch()
    assert_calls_equal(got, dense_calls(golden_streams["5"]["calls"], synth.BLOCKS_PER_STREAM))
    for a, b in zip(got, sync):
        assert [(p.index, bytes(p.data), p.rssi, p.snr) for p in a] == [(p.index, bytes(p.data), p.rssi, p.snr) for p in b]
    # the mirrors refer to the last fetched block once the handle is quiet
    ref = dsp.Demodulator(cfg)
    for blk in blocks[:3]:
        ref.demodulate(blk)
    dem.reset()
    for blk in blocks[:3]:
        dem.submit(blk)
        dem.fetch()
    assert np.array_equal(dem.discriminated, ref.discriminated)
    # several receivers through the same pipeline
    raws = synth.synth_streams([0, 1, 2])
    md, md2 = dsp.MultiDemodulator(cfg, 3), dsp.MultiDemodulator(cfg, 3)
    outs = []
    md.submit(raws[:, : 2 * B])
    for b in range(1, 6):
        md.submit(raws[:, 2 * B * b: 2 * B * (b + 1)])
        outs.append(md.fetch())
    outs.append(md.fetch())
    for b in range(6):
        want = md2.demodulate(raws[:, 2 * B * b: 2 * B * (b + 1)])
        assert [[(p.index, bytes(p.data)) for p in s_] for s_ in outs[b]] == [[(p.index, bytes(p.data)) for p in s_] for s_ in want]


def test_worker_loops_on_the_gpu(dsp, golden_streams):
    """rtldavis_amd.worker: the reference's worker loop (worker.py:34-58) on the pipelined
    demodulator, and several data queues drained into one MultiDemodulator."""
    import queue
    import threading
    from rtldavis_amd import worker
    cfg = prod_cfg(dsp)
    B = cfg.block_size

    class Parser:  # what worker_main needs of protocol.Parser: cfg, demodulator, parse()
        def __init__(self):
            self.cfg = cfg
            self.demodulator = dsp.Demodulator(cfg)
            self.seen = 0

        def parse(self, packets):
            self.seen += 1
            d = self.demodulator.discriminated  # protocol.py:307-309 reads it for every packet
            assert d.shape == (2 * B,)
            return [(self.seen - 1, p.index, bytes(p.data).hex()) for p in packets]

    raw = synth.synth_streams([3])[0]
    dq, rq = queue.Queue(), queue.Queue()
    for b in range(synth.BLOCKS_PER_STREAM):
        dq.put(raw[2 * B * b: 2 * B * (b + 1)])
    dq.put(np.zeros(7, np.uint8))  # a bad block is logged and dropped (worker.py:56-58), the loop goes on
    dq.put(None)
    t = threading.Thread(target=worker.worker_loop, args=(dq, rq, Parser), kwargs=dict(poll_s=0.05))
    t.start(); t.join(60)
    assert not t.is_alive()
    got = []
    while not rq.empty():
        got.append(rq.get())
    want = [(int(c), p["index"], p["data"]) for c, ps in golden_streams["3"]["calls"].items() for p in ps]
    assert sorted(got) == sorted(want) and len(got) > 0
    # three receivers in lock step
    seeds = [0, 1, 2]
    raws = synth.synth_streams(seeds)
    dqs = [queue.Queue() for _ in seeds]
    for b in range(8, 24):
        for k in range(3):
            dqs[k].put(raws[k, 2 * B * b: 2 * B * (b + 1)])
    for q in dqs:
        q.put(None)

    class P2(Parser):
        def __init__(self, k):
            self.cfg, self.demodulator, self.seen = cfg, None, 0

    rq = queue.Queue()
    t = threading.Thread(target=worker.multi_worker_main, args=(dqs, rq, P2), kwargs=dict(poll_s=0.05))
    t.start(); t.join(60)
    assert not t.is_alive()
    per = {k: [] for k in range(3)}
    while not rq.empty():
        k, m = rq.get()
        per[k].append(m[2])
    ref = dsp.MultiDemodulator(cfg, 3)
    exp = {k: [] for k in range(3)}
    for b in range(8, 24):
        for k, pk in enumerate(ref.demodulate(raws[:, 2 * B * b: 2 * B * (b + 1)])):
            exp[k] += [bytes(p.data).hex() for p in pk]
    assert per == exp and sum(len(v) for v in per.values()) > 0


def test_startup_signed_zero_quadrants(dsp):
    g = load_json("startup_quadrants.json")
    cfg = prod_cfg(dsp, block_size=512)
    for name, rec in g.items():
        raw = np.frombuffer(bytes.fromhex(rec["raw"]), dtype=np.uint8)
        dem = dsp.Demodulator(cfg)
        dem.demodulate(raw)
        bits = np.packbits(dem.quantized[-512:], bitorder="little")
        assert bits.tobytes().hex() == rec["bits"], name


def test_errors_like_reference(dsp):
    dem = dsp.Demodulator(prod_cfg(dsp))
    with pytest.raises(ValueError, match="Incompatible array sizes"):
        dem.demodulate(np.zeros(100, dtype=np.uint8))
    with pytest.raises(ValueError, match="Incompatible array sizes"):
        dem.demodulate(np.zeros(100, dtype=np.complex128))
    with pytest.raises(ValueError, match="Incompatible array sizes"):
        dsp.ByteToCmplxLUT().execute(np.zeros(10, np.uint8), np.zeros(4, np.complex128))
    # a non-contiguous slice is accepted
    raw = np.fromfile(f"{GOLDEN}/burst_seed0_b20_22.u8", dtype=np.uint8)
    wide = np.zeros(2 * 16384, dtype=np.uint8)
    wide[::2] = raw[:16384]
    assert dem.demodulate(wide[::2]) == []


# ---------------------------------------------------------------- stage functions
def test_stage_functions_against_fixture(dsp):
    z = load_npz("b512_stages.npz")
    raw = z["raw"]
    x = np.zeros(512, dtype=np.complex128)
    dsp.ByteToCmplxLUT().execute(raw[-1024:], x)
    np.testing.assert_array_equal(x, z["last_raw_samples"])
    y = np.empty_like(x)
    dsp.rotate_fs4(x, y)
    np.testing.assert_array_equal(y, z["last_iq"][9:])
    dsp.rotate_fs4(x, x)  # in place, like dsp.py:160
    np.testing.assert_array_equal(x, z["last_iq"][9:])
    f = np.zeros(513, dtype=np.complex128)
    dsp.fir9(z["last_iq"], f[1:])
    np.testing.assert_allclose(f[1:], z["last_filtered"][1:], rtol=0, atol=1e-14)
    d = np.zeros(512)
    dsp.discriminate(z["last_filtered"], d)
    ref = z["last_discriminated"][512:]
    assert np.all(np.abs(d - ref) <= 1e-9 * np.maximum(1, np.abs(ref)))
    q = np.zeros(512, dtype=np.uint8)
    dsp.quantize(z["last_discriminated"][512:], q)
    np.testing.assert_array_equal(q, z["last_quantized"][-512:])
    assert dsp.search(z["last_quantized"], dsp.PacketConfig(19200, 14, 16, 80, "1100101110001001", 512)) == \
        [i for i in _ref_search(z["last_quantized"], 14, "1100101110001001")]


def _ref_search(q, S, pre):
    from oracle import dsp_oracle as O
    return O.search(q, O.OracleConfig(symbol_length=S, preamble=pre, preamble_symbols=len(pre)))


def test_quantize_reference_tests(dsp):
    """The reference's tests/test_dsp.py:4-33, verbatim in behaviour."""
    in_float = np.array([-5.0, 5.0, -0.1, 0.1, 0.0])
    out_byte = np.zeros(len(in_float), dtype=np.uint8)
    dsp.quantize(in_float, out_byte)
    assert out_byte.tolist() == [1, 0, 1, 0, 0]
    rng = np.random.default_rng(42)
    in_float = rng.uniform(-10, 10, 1000)
    out_byte = np.zeros(len(in_float), dtype=np.uint8)
    dsp.quantize(in_float, out_byte)
    assert np.array_equal(out_byte, (in_float < 0).astype(np.uint8))
    z = np.zeros(1, np.uint8)
    dsp.quantize(np.array([-0.0]), z)
    assert z[0] == 1


def test_search_many_matches_order(dsp):
    """A buffer full of preambles: order must be phase-major (dsp.py:175-186)."""
    pre = "1100101110001001"
    cfg = dsp.PacketConfig(19200, 14, 16, 80, pre, 512)
    q = np.zeros(4096, dtype=np.uint8)
    for start in (5, 19, 300, 301, 1000, 2000 + 13, 3500):
        q[start: start + 16 * 14: 14] = [int(c) for c in pre]
    assert dsp.search(q, cfg) == _ref_search(q, 14, pre)


def test_sharded_demodulation_with_the_real_batch_demodulator(dsp, batchmod, golden_streams):
    """shard.demodulate_sharded (the N > 1 entry point, tests/test_sharding_gloo.py covers its
    partition/gather logic with two CPU ranks) driving the real BatchDemodulator: world size 1 here,
    so the whole range lands on this GPU; records come back in global stream order and equal the
    reference fixtures."""
    from rtldavis_amd.shard import demodulate_sharded, shard_range
    seeds = list(range(6))
    cfg = prod_cfg(dsp)

    def load(lo, hi):
        return synth.synth_streams(seeds[lo:hi])

    def demod(raw):
        bd = batchmod.BatchDemodulator(cfg, raw.shape[0], synth.BLOCKS_PER_STREAM)
        res = bd.demodulate(raw)
        return [(s, c, (p.index, bytes(p.data).hex())) for s, calls in enumerate(res) for c, ps in enumerate(calls)
                for p in ps]

    got = demodulate_sharded(len(seeds), load, demod)
    want = [(s, int(c), (p["index"], p["data"])) for s in range(len(seeds))
            for c, ps in sorted(golden_streams[str(seeds[s])]["calls"].items(), key=lambda kv: int(kv[0])) for p in ps]
    assert got == want
    # the shards a 4-rank job would take, one after the other on this GPU, concatenate to the same list
    parts = []
    for r in range(4):
        lo, hi = shard_range(len(seeds), 4, r)
        parts += [(lo + s, c, p) for (s, c, p) in (demod(load(lo, hi)) if hi > lo else [])]
    assert parts == want


def test_degenerate_patterns_match_the_reference(dsp, batchmod):
    """Constant, period-2 and period-4 byte patterns against the REAL reference's output
    (tests/golden/degenerate_patterns.json, tools/gen_golden_degenerate.py): the inputs where the discriminator is
    closest to a tie (|d| down to 1.5e-5 on a -Fs/4 tone), streaming handle and batch path."""
    g = load_json("degenerate_patterns.json")
    B = g["block_size"]
    names = sorted(g["patterns"])
    blocks = []
    for name in names:
        fx = g["patterns"][name]
        flat = np.array([v for pair in fx["period"] for v in pair], dtype=np.uint8)
        blk = np.tile(flat, 2 * B // flat.size)
        blocks.append(blk)
        dem = dsp.Demodulator(prod_cfg(dsp))
        for c, want in enumerate(fx["calls"]):
            pk = dem.demodulate(blk)
            q = np.asarray(dem.quantized).astype(np.uint8)
            assert q.size == want["n"]
            assert int(q.sum()) == want["ones"], f"{name} call {c}"
            assert sha(np.packbits(q, bitorder="little")) == want["quantized_packed_sha256"], f"{name} call {c}"
            assert [[int(p.index), bytes(p.data).hex()] for p in pk] == want["packets"], f"{name} call {c}"
    # batch path: every pattern as one stream of g["blocks"] identical blocks; the last call's window is the
    # stream's last 2B bits
    nb = g["blocks"]
    raw = np.stack([np.tile(b, nb) for b in blocks])
    bd = batchmod.BatchDemodulator(prod_cfg(dsp), len(names), nb)
    res = bd.demodulate(raw)
    for i, name in enumerate(names):
        want = g["patterns"][name]["calls"]
        bits = np.unpackbits(bd.bits(i).view(np.uint8), bitorder="little")[: nb * B]
        assert sha(np.packbits(bits[(nb - 2) * B:], bitorder="little")) == want[-1]["quantized_packed_sha256"], name
        assert [[[int(p.index), bytes(p.data).hex()] for p in call] for call in res[i]] == [w["packets"] for w in want], name


def test_zero_signal_power_raises_like_reference(dsp):
    """dsp.py:231-236: signal_power == 0 over a non-zero noise estimate ends in math.log10(0), a ValueError out of
    demodulate().  Reachable with complex zeros and an all-zero preamble (checked against the real reference:
    'math domain error' on the first call)."""
    cfg = dsp.PacketConfig(19200, 4, 4, 8, "0000", 512)
    dem = dsp.Demodulator(cfg)
    with pytest.raises(ValueError, match="math domain error"):
        dem.demodulate(np.zeros(512, dtype=np.complex128))


def test_batch_discriminated_whole_streams_against_c_oracle(dsp, batchmod):
    """`discriminated` over WHOLE streams (33 blocks each, four streams incl. a quiet and a loud one), every
    sample, against the C oracle's float64 values: the 1e-5 bound of the north_star on more than the three blocks
    the fixture holds."""
    from oracle import c_oracle as CO
    seeds = [3, 17, 42, 63]
    raw = synth.synth_streams(seeds)
    bd = batchmod.BatchDemodulator(prod_cfg(dsp), len(seeds), synth.BLOCKS_PER_STREAM)
    bd.demodulate(raw)
    n = raw.shape[1] // 2
    worst = 0.0
    for i in range(len(seeds)):
        _, _, ref = CO.demod_stream(raw[i], CO.make_cfg(), want_disc=True)
        got = bd.discriminated(i, 0, n)
        err = np.abs(got - ref) / np.maximum(1.0, np.abs(ref))
        worst = max(worst, float(err.max()))
        assert np.all(err <= 1e-5), f"stream {i}: {err.max()} at {int(err.argmax())}"
    assert worst < 1e-5


def test_slice_forms_agree(dsp, batchmod, tmp_path):
    """The two-kernel slice of the batch path (k_classify + k_rssi_u8: dense records, RSSI windows on the matrix pipe)
    and the one-kernel form (RD_SLICE_IMPL=wave, read once per process: hence the child) return the same packets;
    RSSI and SNR agree to 1e-3 dB (the fixtures' tolerance: fp32 window sums against the exact integer filter)."""
    import subprocess
    import sys
    seeds = list(range(24))
    raw = synth.synth_streams(seeds)
    bd = batchmod.BatchDemodulator(prod_cfg(dsp), len(seeds), synth.BLOCKS_PER_STREAM)
    bd.upload(raw)
    bd.run()
    mine = bd.results().copy()
    out = tmp_path / "wave.npy"
    child = (
        "import sys, numpy as np\n"
        "from rtldavis_amd import batch, dsp, synth\n"
        "cfg = dsp.PacketConfig(19200, 14, 16, 80, '1100101110001001', 8192)\n"
        f"raw = synth.synth_streams(list(range({len(seeds)})))\n"
        f"bd = batch.BatchDemodulator(cfg, {len(seeds)}, synth.BLOCKS_PER_STREAM)\n"
        "bd.upload(raw); bd.run()\n"
        f"np.save(r'{out}', bd.results())\n")
    env = dict(os.environ, RD_SLICE_IMPL="wave")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run([sys.executable, "-c", child], check=True, env=env, cwd=root, timeout=300)
    other = np.load(out)
    assert len(mine) == len(other) > 0
    for f in ("stream", "call", "index", "nbytes"):
        assert np.array_equal(mine[f], other[f]), f
    assert np.array_equal(mine["data"], other["data"])
    assert np.all(np.abs(mine["rssi"] - other["rssi"]) < 1e-3)
    assert np.all(np.abs(mine["snr"] - other["snr"]) < 1e-3)


def test_match_list_overflow_is_transparent(dsp, batchmod, golden_streams, monkeypatch):
    """A first match list that is far too small (RD_TEST_MATCH_CAP, read when the handle allocates): the run's
    counters say so, rd_batch_results grows the lists, searches and slices again, and the packets are the
    fixtures' - for the dense (two-kernel) record layout of the production shape and twice in a row."""
    monkeypatch.setenv("RD_TEST_MATCH_CAP", "7")
    seeds = list(range(12))
    raw = synth.synth_streams(seeds)
    bd = batchmod.BatchDemodulator(prod_cfg(dsp), len(seeds), synth.BLOCKS_PER_STREAM)
    for _ in range(2):
        res = bd.demodulate(raw)
        for i, seed in enumerate(seeds):
            assert_calls_equal(res[i], dense_calls(golden_streams[str(seed)]["calls"], synth.BLOCKS_PER_STREAM))
    assert bd.counters()["matches"] > 7


def _records_of_child(tmp_path, n_seeds, env_extra, name):
    """The batch path's records from a fresh process (the library reads its switches once per process)."""
    import subprocess
    import sys
    out = tmp_path / f"{name}.npy"
    child = (
        "import sys, numpy as np\n"
        "from rtldavis_amd import batch, dsp, synth\n"
        "cfg = dsp.PacketConfig(19200, 14, 16, 80, '1100101110001001', 8192)\n"
        f"raw = synth.synth_streams(list(range({n_seeds})))\n"
        f"bd = batch.BatchDemodulator(cfg, {n_seeds}, synth.BLOCKS_PER_STREAM)\n"
        "bd.upload(raw); bd.run()\n"
        f"np.save(r'{out}', bd.results())\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run([sys.executable, "-c", child], check=True, env=dict(os.environ, **env_extra), cwd=root, timeout=300)
    return np.load(out)


@pytest.mark.gpu
def test_ordered_tail_equals_the_unordered_kernels_plus_host_ordering(dsp, batchmod, tmp_path):
    """The batch path's default tail orders and dedupes on the device (k_tail: dsp.py:171-188 order, dsp.py:203-205
    first occurrence wins); RD_TAIL_IMPL=legacy is the form every other shape takes - separate kernels, radix sort and
    dedupe on the host.  Same records, field for field (the child process keeps the two forms' environments apart)."""
    seeds = list(range(40))
    raw = synth.synth_streams(seeds)
    bd = batchmod.BatchDemodulator(prod_cfg(dsp), len(seeds), synth.BLOCKS_PER_STREAM)
    bd.upload(raw)
    bd.run()
    mine = bd.results().copy()
    other = _records_of_child(tmp_path, len(seeds), {"RD_TAIL_IMPL": "legacy"}, "legacy")
    assert len(mine) == len(other) > 0
    for f in ("stream", "call", "index", "nbytes"):
        assert np.array_equal(mine[f], other[f]), f
    assert np.array_equal(mine["data"], other["data"])
    assert np.all(np.abs(mine["rssi"] - other["rssi"]) < 1e-3)
    assert np.all(np.abs(mine["snr"] - other["snr"]) < 1e-3)
    assert bd.counters()["matches"] > 0


@pytest.mark.gpu
def test_pipelined_completion_returns_the_same_packets(dsp, batchmod, golden_streams):
    """rd_batch_set_pipelined: a run ends without an event of its own and is adopted by the next demod kernel launched
    on the stream (another handle's or its own).  Three handles with different inputs round-robin on one stream, runs
    queued ahead as bench.py does; then the paths nobody adopts: the last run of the loop, a single run, two runs in a
    row without fetching, timing switched on, a handle destroyed with its run still waiting.  Every result = the
    fixtures' packets (dsp.py:128-253)."""
    groups = [list(range(0, 6)), list(range(6, 12)), list(range(12, 18))]
    bds = []
    for g in groups:
        bd = batchmod.BatchDemodulator(prod_cfg(dsp), len(g), synth.BLOCKS_PER_STREAM)
        bd.upload(synth.synth_streams(g))
        bd.set_pipelined(True)
        bds.append(bd)

    def check(bd, g):
        res = bd.packets()
        for i, seed in enumerate(g):
            assert_calls_equal(res[i], dense_calls(golden_streams[str(seed)]["calls"], synth.BLOCKS_PER_STREAM))

    R, k = len(bds), 9
    for i in range(R - 1):
        bds[i].run()
    for i in range(k):
        if i + R - 1 < k:
            bds[(i + R - 1) % R].run()
        check(bds[i % R], groups[i % R])      # adopted by the run launched behind it; the last one by nobody
    bds[0].run()
    check(bds[0], groups[0])                  # a single run: flushed by results()
    bds[1].run()
    bds[1].run()                              # run again without fetching: the first one is flushed by the second
    check(bds[1], groups[1])
    for bd in bds:
        bd.set_timing(1)
    bds[0].run()
    bds[1].run()
    bds[2].run()
    for bd, g in zip(bds, groups):
        check(bd, g)
    tm = [bd.timing() for bd in bds]
    assert all(t["runs"] == 1 and t["demod_ms"] > 0 for t in tm)
    assert tm[0]["total_ms"] == 0 and tm[1]["total_ms"] == 0   # adopted: no end-of-run event
    bds[2].run()
    bds[2].close()                            # destroyed while waiting for an adopter
    bds[0].run()
    check(bds[0], groups[0])


@pytest.mark.gpu
def test_ordered_tail_falls_back_when_a_stream_overflows_its_bucket(dsp, batchmod, golden_streams, monkeypatch):
    """RD_TEST_BUCKET_CAP=2: every stream has more than two matches, the one-launch tail raises its overflow flag, and
    rd_batch_results re-runs the unordered kernels on the same bits - the fixtures' packets, twice in a row (the
    second run goes straight to the unordered path: no new upload in between), and again after a new upload."""
    monkeypatch.setenv("RD_TEST_BUCKET_CAP", "2")
    seeds = list(range(10))
    raw = synth.synth_streams(seeds)
    bd = batchmod.BatchDemodulator(prod_cfg(dsp), len(seeds), synth.BLOCKS_PER_STREAM)
    for _ in range(2):
        res = bd.demodulate(raw)
        for i, seed in enumerate(seeds):
            assert_calls_equal(res[i], dense_calls(golden_streams[str(seed)]["calls"], synth.BLOCKS_PER_STREAM))
    bd.run()
    res = bd.packets()
    for i, seed in enumerate(seeds):
        assert_calls_equal(res[i], dense_calls(golden_streams[str(seed)]["calls"], synth.BLOCKS_PER_STREAM))


# ---------------------------------------------------------------- round 4: host waits, stream hand-over
def _hip_stream():
    """A second hipStream_t (integer handle) from the HIP runtime the library already loaded."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    st = C.c_void_p()
    assert hip.hipStreamCreate(C.byref(st)) == 0
    return st.value


@pytest.mark.gpu
def test_host_wait_deadline_drops_the_block_and_goes_on(dsp, batchmod, golden_streams):
    """Every host wait has a deadline (rd_set_wait_timeout_ms / RD_WAIT_TIMEOUT_MS): a wait that passes it raises
    HipError instead of spinning for ever, so that the worker's "log, drop the block, continue"
    (/root/reference/src/rtldavis/worker.py:56-58) fires.  Nothing is made to hang: with the deadline at 0 an ordinary
    block times out.  The dropped block's packets are lost, the stream's state is not: every later call equals the
    fixture; so does a whole stream after reset(); a batch run whose wait timed out is fetched by the next call."""
    from rtldavis_amd import _lib
    L = _lib.lib()
    B = 8192
    raw = synth.synth_stream(0)
    want = dense_calls(golden_streams["0"]["calls"], synth.BLOCKS_PER_STREAM)
    for make, feed in ((lambda: dsp.Demodulator(prod_cfg(dsp)), lambda d, blk: d.demodulate(blk)),):
        dem = make()
        calls = []
        for b in range(synth.BLOCKS_PER_STREAM):
            blk = raw[2 * B * b: 2 * B * (b + 1)]
            if b == 7:  # (a call without packets in the fixture)
                assert want[b] == []
                prev = L.rd_set_wait_timeout_ms(0)
                try:
                    with pytest.raises(_lib.HipError, match="timed out"):
                        feed(dem, blk)
                    with pytest.raises(RuntimeError):   # its packets are gone: nothing to fetch
                        dem.fetch()
                finally:
                    assert L.rd_set_wait_timeout_ms(-1) == 0
                assert prev > 0
                calls.append([])
                continue
            calls.append(feed(dem, blk))
        assert_calls_equal(calls, want)
        # a timed-out fetch of a submitted block, then reset(): waits for the block that was given up on, clears
        dem.submit(raw[: 2 * B])
        L.rd_set_wait_timeout_ms(0)
        try:
            with pytest.raises(_lib.HipError, match="timed out"):
                dem.fetch()
        finally:
            L.rd_set_wait_timeout_ms(-1)
        dem.reset()
        assert_calls_equal(run_streaming(dem, raw), want)
    # batch handle: the wait for a run's results
    bd = batchmod.BatchDemodulator(prod_cfg(dsp), 2, synth.BLOCKS_PER_STREAM)
    bd.upload(synth.synth_streams([0, 1]))
    bd.run()
    bd.packets()          # (first run: buffers and kernels warm)
    L.rd_set_wait_timeout_ms(0)
    try:
        bd.run()
        with pytest.raises(_lib.HipError, match="timed out"):
            bd.results()
    finally:
        L.rd_set_wait_timeout_ms(-1)
    res = bd.packets()    # the run is still there to be fetched
    for i, seed in enumerate((0, 1)):
        assert_calls_equal(res[i], dense_calls(golden_streams[str(seed)]["calls"], synth.BLOCKS_PER_STREAM))
    bd.upload(synth.synth_streams([2, 3]))   # upload -> run -> results goes on as before
    res = bd.demodulate(synth.synth_streams([2, 3]))
    for i, seed in enumerate((2, 3)):
        assert_calls_equal(res[i], dense_calls(golden_streams[str(seed)]["calls"], synth.BLOCKS_PER_STREAM))


@pytest.mark.gpu
def test_pipelined_run_adopted_by_an_untimed_handle_without_the_fast_kernel(dsp, batchmod, golden_streams):
    """ADVICE r3 (rd_batch_run, third branch): a pipelined run waiting on a stream is adopted by the next run launched
    there; when that run belongs to an untimed handle whose streams are not 16-byte multiples (no fused demod kernel,
    no dispatch event) the carrier must be an event of THAT run.  A's packets = the fixtures, B's = the C oracle."""
    from oracle import c_oracle as CO
    seeds = [0, 1, 2]
    a = batchmod.BatchDemodulator(prod_cfg(dsp), len(seeds), synth.BLOCKS_PER_STREAM)
    a.upload(synth.synth_streams(seeds))
    a.set_pipelined(True)
    rng = np.random.default_rng(11)
    Bb, nb, ns = 36, 70, 3   # 2 * 36 * 70 bytes per stream = 5040: not a multiple of 16
    raw_b = rng.integers(0, 256, size=(ns, 2 * Bb * nb), dtype=np.uint8)
    b = batchmod.BatchDemodulator(prod_cfg(dsp, block_size=Bb), ns, nb)
    b.upload(raw_b)
    want_b, _ = CO.demod_batch(raw_b, CO.make_cfg(block_size=Bb), threads=2, want_bits=True)
    for _ in range(3):
        a.run()
        b.run()          # adopts a's run
        res_a = a.packets()
        for i, seed in enumerate(seeds):
            assert_calls_equal(res_a[i], dense_calls(golden_streams[str(seed)]["calls"], synth.BLOCKS_PER_STREAM))
        res_b = b.packets()
        for i in range(ns):
            got = [(c, p.index, bytes(p.data).hex()) for c, ps in enumerate(res_b[i]) for p in ps]
            assert got == [(p.call, p.index, bytes(p.data).hex()) for p in want_b[i]]


@pytest.mark.gpu
def test_pipelined_handle_rerun_on_another_stream(dsp, batchmod, golden_streams):
    """ADVICE r3 (rd_batch_run): a pipelined run nobody fetched, then the same handle run again on ANOTHER stream: the
    first run is flushed on the stream it was launched on (its completion event there, its g_tail entry removed), the
    second waits for it; then a third handle launched on the first stream must not adopt anything stale, and destroying
    the handles leaves nothing behind.  Results = the fixtures every time."""
    seeds = [4, 5, 6, 7]
    st2 = _hip_stream()
    bd = batchmod.BatchDemodulator(prod_cfg(dsp), len(seeds), synth.BLOCKS_PER_STREAM)
    bd.upload(synth.synth_streams(seeds))
    bd.set_pipelined(True)
    other = batchmod.BatchDemodulator(prod_cfg(dsp), 2, synth.BLOCKS_PER_STREAM)
    other.upload(synth.synth_streams([0, 1]))

    def check(h, sd):
        res = h.packets()
        for i, seed in enumerate(sd):
            assert_calls_equal(res[i], dense_calls(golden_streams[str(seed)]["calls"], synth.BLOCKS_PER_STREAM))

    for _ in range(3):
        bd.run(0)        # deferred on the null stream
        bd.run(st2)      # not fetched: flushed where it was launched, then this run on st2
        other.run(0)     # the null stream holds no waiting run any more
        check(other, [0, 1])
        check(bd, seeds)
        bd.run(st2)
        bd.run(0)
        check(bd, seeds)
    bd.run(st2)
    bd.close()           # destroyed with a run waiting on st2
    other.run(st2)
    check(other, [0, 1])


# ---------------------------------------------------------------- round 4: the one-launch tail (k_tail)
@pytest.mark.gpu
def test_one_launch_tail_is_the_default_and_equals_the_other_forms(dsp, batchmod, golden_streams, monkeypatch):
    """Everything behind the demod kernel - exact bits for the listed groups, Demodulator._search and ._slice
    (dsp.py:171-246: order, dedupe, RSSI / SNR) - runs as ONE launch by default; RD_TAIL_IMPL=legacy selects the
    separate kernels + host ordering.  Both: the fixtures' packets, the same records field for field, the same bits.
    13 streams: three whole groups of four and a partial one."""
    seeds = list(range(13))
    raw = synth.synth_streams(seeds)
    out = {}
    for impl in (None, "legacy"):
        if impl is None:
            monkeypatch.delenv("RD_TAIL_IMPL", raising=False)
        else:
            monkeypatch.setenv("RD_TAIL_IMPL", impl)
        bd = batchmod.BatchDemodulator(prod_cfg(dsp), len(seeds), synth.BLOCKS_PER_STREAM)
        bd.upload(raw)
        for _ in range(3):      # both counter sets, the sequence numbers of the groups' totals
            bd.run()
            rec = bd.results().copy()
        forms = bd.last_run_forms()
        assert forms["one_launch_tail"] == (impl is None) and forms["ordered_tail"] == (impl is None) and not forms["second_pass"]
        res = bd.packets()
        for i, seed in enumerate(seeds):
            assert_calls_equal(res[i], dense_calls(golden_streams[str(seed)]["calls"], synth.BLOCKS_PER_STREAM))
            assert sha(bd.bits(i)) == golden_streams[str(seed)]["bits_sha256"]
        out[impl] = (rec, bd.counters())
    for impl in ("legacy",):
        a, b = out[None][0], out[impl][0]
        assert len(a) == len(b)
        for f in ("stream", "call", "index", "nbytes", "data"):
            assert np.array_equal(a[f], b[f]), (impl, f)
        assert np.allclose(a["rssi"], b["rssi"], atol=1e-4, rtol=0) and np.allclose(a["snr"], b["snr"], atol=1e-4, rtol=0)
        assert out[None][1] == out[impl][1], "fix-up and match counts differ between the forms"


@pytest.mark.gpu
@pytest.mark.parametrize("ns,parts", [(6, 10), (3, 24)])
def test_long_streams_size_their_match_lists(dsp, batchmod, ns, parts):
    """the per-stream match lists are sized from the stream's length (rd_host.h:
    rd_ord_bucket_cap): 330 blocks of noise with ten bursts (each stream = ten fixture-style streams back to back) hold
    ~45 raw preamble matches per stream - more than the 32 of round 3's buckets - and still take the one-launch
    tail in its first pass; 792 blocks: lists of 416 matches, 98 KiB of dynamic LDS per workgroup (above the 64 KiB a
    kernel gets without asking).  Packets and bits against the C oracle (dsp.py:171-246)."""
    from oracle import c_oracle as CO
    raw = np.stack([np.concatenate([synth.synth_stream(100 + 17 * s + k) for k in range(parts)]) for s in range(ns)])
    nb = parts * synth.BLOCKS_PER_STREAM
    bd = batchmod.BatchDemodulator(prod_cfg(dsp), ns, nb)
    res = bd.demodulate(raw)
    forms = bd.last_run_forms()
    assert forms["one_launch_tail"] and forms["ordered_tail"] and not forms["second_pass"]
    want, wbits = CO.demod_batch(raw, CO.make_cfg(), threads=4, want_bits=True, cap_per_stream=4096)
    assert max(len(w) for w in want) > 32
    for i in range(ns):
        assert np.array_equal(bd.bits(i), wbits[i]), i
        got = [(c, p.index, bytes(p.data).hex()) for c, ps in enumerate(res[i]) for p in ps]
        assert got == [(p.call, p.index, bytes(p.data).hex()) for p in want[i]], i
        flat = [p for ps in res[i] for p in ps]
        for p, q in zip(flat, want[i]):
            assert abs(p.rssi - q.rssi) < 1e-3 and abs(p.snr - q.snr) < 1e-3


@pytest.mark.gpu
def test_one_launch_tail_overflows_fall_back(dsp, batchmod, golden_streams, monkeypatch):
    """The one-launch tail's two overflows on ordinary inputs.  A stream with more matches than its list
    (RD_TEST_BUCKET_CAP=2): the separate kernels finish the run on the bits k_tail has already made exact.  A group's
    fix-up bucket (RD_TEST_FIX_BCAP=8): every run is re-evaluated exactly, then the separate kernels.
    Both: second_pass reported, results = fixtures / C oracle."""
    from oracle import c_oracle as CO
    monkeypatch.setenv("RD_TEST_BUCKET_CAP", "2")
    seeds = list(range(6))
    bd = batchmod.BatchDemodulator(prod_cfg(dsp), len(seeds), synth.BLOCKS_PER_STREAM)
    res = bd.demodulate(synth.synth_streams(seeds))
    f = bd.last_run_forms()
    assert f["second_pass"] and not f["ordered_tail"]
    for i, seed in enumerate(seeds):
        assert_calls_equal(res[i], dense_calls(golden_streams[str(seed)]["calls"], synth.BLOCKS_PER_STREAM))
    monkeypatch.delenv("RD_TEST_BUCKET_CAP")
    monkeypatch.setenv("RD_TEST_FIX_BCAP", "8")   # a group lists ~40 words (chunk starts, first runs): every bucket overflows
    rng = np.random.default_rng(3)
    nb = 8
    raw = np.stack([rng.integers(126, 130, size=2 * 8192 * nb, dtype=np.uint8) for _ in range(5)] +
                   [synth.synth_stream(9)[: 2 * 8192 * nb]])
    bd = batchmod.BatchDemodulator(prod_cfg(dsp), raw.shape[0], nb)
    want, wbits = CO.demod_batch(raw, CO.make_cfg(), threads=4, want_bits=True)
    for _ in range(2):   # (the second run of the same input goes straight to the separate kernels)
        bd.upload(raw)
        bd.run()
        res = bd.packets()
        f = bd.last_run_forms()
        assert f["second_pass"] and not f["one_launch_tail"]
        for i in range(raw.shape[0]):
            assert np.array_equal(bd.bits(i), wbits[i]), i
            got = [(c, p.index, bytes(p.data).hex()) for c, ps in enumerate(res[i]) for p in ps]
            assert got == [(p.call, p.index, bytes(p.data).hex()) for p in want[i]]


@pytest.mark.gpu
def test_zero_copy_ring_input_equals_the_fixtures(dsp, golden_streams):
    """SURVEY section 8f-4 literally: blocks that lie in a multiprocessing.shared_memory ring (rtldavis_amd.ring) are
    demodulated where the producer wrote them - Demodulator.register_input pins and maps the segment, submit_from
    launches on a slot - uint8 and complex128 blocks, two in flight, several times round a 4-slot ring: the fixtures'
    packets (dsp.py:139-246), and the reference's size error for a wrong count."""
    from rtldavis_amd.ring import BlockRing, KIND_C128
    B = 8192
    raw = synth.synth_stream(0)
    want = dense_calls(golden_streams["0"]["calls"], synth.BLOCKS_PER_STREAM)
    ring = BlockRing.create(n_slots=4, block_size=B)
    try:
        dem = dsp.Demodulator(prod_cfg(dsp))
        dem.register_input(ring.data)
        with pytest.raises(ValueError):
            dem.submit_from(0, 2 * B - 2)
        with pytest.raises(ValueError):
            dem.submit_from(8, 2 * B)              # not 16-byte aligned
        calls, inflight = [], 0
        for b in range(synth.BLOCKS_PER_STREAM):
            assert ring.put(raw[2 * B * b: 2 * B * (b + 1)], timeout=5.0)
            slot, off, kind, count = ring.get(taken=inflight, timeout=5.0)
            dem.submit_from(off, count, kind == KIND_C128)
            inflight += 1
            if inflight == 2:
                calls.append(dem.fetch())
                ring.release()
                inflight -= 1
        calls.append(dem.fetch())
        ring.release()
        assert_calls_equal(calls, want)
        assert ring.backlog == (synth.BLOCKS_PER_STREAM, synth.BLOCKS_PER_STREAM)
        # complex blocks through the same ring: the complex-input fixture (dsp.py:144-150)
        g = load_json("complex_input.json")
        rawc = np.fromfile(f"{GOLDEN}/burst_seed0_b20_22.u8", dtype=np.uint8)
        x = (rawc[0::2].astype(np.float64) - 127.5) / 127.5 + 1j * (rawc[1::2].astype(np.float64) - 127.5) / 127.5
        dem.reset()
        ccalls = []
        for b in range(x.size // B):
            assert ring.put(x[B * b: B * (b + 1)], timeout=5.0)
            slot, off, kind, count = ring.get(timeout=5.0)
            assert kind == KIND_C128 and count == B
            dem.submit_from(off, count, True)
            ccalls.append(dem.fetch())
            ring.release()
        assert_calls_equal(ccalls, g["calls"])
        dem.register_input(None)
        with pytest.raises(RuntimeError):
            dem.submit_from(0, 2 * B)              # nothing registered any more
    finally:
        ring.close()


@pytest.mark.gpu
def test_upload_async_on_a_copy_stream(dsp, batchmod, golden_streams):
    """rd_batch_upload_async: the copy goes to a copy stream and the handle's next run waits for it on the device; the
    next upload of a handle queues behind its previous run.  Two handles alternate inputs - upload(i + 1) issued before
    run(i)'s results are fetched - and every result equals the fixtures of the input that run was given."""
    cs = _hip_stream()
    sets = [list(range(0, 5)), list(range(5, 10)), list(range(10, 15))]
    inputs = [np.ascontiguousarray(synth.synth_streams(s)) for s in sets]
    bds = [batchmod.BatchDemodulator(prod_cfg(dsp), 5, synth.BLOCKS_PER_STREAM) for _ in range(2)]

    def check(bd, seeds):
        res = bd.packets()
        for i, seed in enumerate(seeds):
            assert_calls_equal(res[i], dense_calls(golden_streams[str(seed)]["calls"], synth.BLOCKS_PER_STREAM))

    k = 7
    bds[0].upload_async(inputs[0], cs)
    for i in range(k):
        if i + 1 < k:
            bds[(i + 1) % 2].upload_async(inputs[(i + 1) % 3], cs)
        bds[i % 2].run()
        check(bds[i % 2], sets[i % 3])
    # a second upload of the same handle without a fetch in between: it queues behind the run that still reads the input
    bds[0].upload_async(inputs[1], cs)
    bds[0].run()
    bds[0].upload_async(inputs[2], cs)
    bds[0].run()
    check(bds[0], sets[2])
    with pytest.raises(ValueError):
        bds[0].upload_async(inputs[0][:, :-2].copy(), cs)
