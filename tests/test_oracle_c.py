"""Pin oracle/dsp_oracle.c (the CPU baseline that travels to the GPU box) against the
golden fixtures from the real reference and against the NumPy oracle.  CPU only."""
import hashlib

import numpy as np
import pytest

from conftest import GOLDEN, assert_calls_equal, dense_calls, load_json, load_npz
from oracle import c_oracle as CO
from oracle import dsp_oracle as O
from rtldavis_amd import synth


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module", autouse=True)
def _build():
    CO.build()


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 31, 63])
def test_streams(seed, golden_streams):
    g = golden_streams[str(seed)]
    calls, bits, _ = CO.demod_stream(synth.synth_stream(seed), CO.make_cfg())
    assert sha(bits) == g["bits_sha256"]
    assert_calls_equal(calls, dense_calls(g["calls"], synth.BLOCKS_PER_STREAM))


def test_burst_state_and_stages():
    raw = np.fromfile(f"{GOLDEN}/burst_seed0_b20_22.u8", dtype=np.uint8)
    g = load_json("burst_seed0_b20_22.json")
    st = load_npz("burst_seed0_b20_22_state.npz")
    calls, bits, disc = CO.demod_stream(raw, CO.make_cfg(), want_disc=True)
    assert_calls_equal(calls, g["calls"])
    assert np.array_equal(bits, st["bits"])
    ref = st["disc_all"]
    assert np.all(np.abs(disc - ref) <= 1e-9 * np.maximum(1, np.abs(ref)))
    filt, disc2, b = CO.stages(raw)
    np.testing.assert_allclose(filt[2 * 8192:], st["filtered"], rtol=0, atol=1e-14)
    f, d, bits_np = O.demod_stream_oneshot(raw)
    assert np.array_equal(b, bits_np)
    np.testing.assert_allclose(filt[1:], f, rtol=0, atol=1e-14)


def test_block_512_edge_and_alt():
    z = load_npz("b512_stages.npz")
    g = load_json("b512_calls.json")
    calls, bits, _ = CO.demod_stream(z["raw"], CO.make_cfg(**g["config"]))
    assert_calls_equal(calls, g["calls"])
    assert np.array_equal(bits, z["bits"])
    e = load_json("edge_q_eq_B.json")
    raw = synth.synth_stream(e["seed"], n_samples=e["n_samples"], start=e["start"])
    calls, bits, _ = CO.demod_stream(raw, CO.make_cfg())
    assert_calls_equal(calls, e["calls"])
    a = load_json("alt_s8_b1024.json")
    raw = synth.synth_stream(a["seed"], n_samples=a["n_samples"], symbol_length=8, margin=a["margin"])
    calls, bits, _ = CO.demod_stream(raw, CO.make_cfg(**a["config"]))
    assert_calls_equal(calls, a["calls"])
    assert sha(bits) == a["bits_sha256"]


def test_two_bursts_in_one_window():
    for seed, rec in load_json("two_bursts.json").items():
        raw = synth.synth_two_bursts(int(seed), rec["gap"])
        calls, bits, _ = CO.demod_stream(raw, CO.make_cfg())
        assert_calls_equal(calls, rec["calls"])
        assert sha(bits) == rec["bits_sha256"]


def test_startup_quadrants():
    for name, rec in load_json("startup_quadrants.json").items():
        raw = np.frombuffer(bytes.fromhex(rec["raw"]), dtype=np.uint8)
        calls, bits, _ = CO.demod_stream(raw, CO.make_cfg(block_size=512))
        assert bits.tobytes().hex() == rec["bits"], name


def test_batch_threads_match_single(golden_streams):
    raw = synth.synth_streams(range(4))
    res, bits = CO.demod_batch(raw, CO.make_cfg(), threads=4, want_bits=True)
    for s in range(4):
        g = golden_streams[str(s)]
        assert sha(bits[s]) == g["bits_sha256"]
        want = [(int(c), p["index"], p["data"]) for c, ps in sorted(g["calls"].items(), key=lambda kv: int(kv[0]))
                for p in ps]
        assert [(p.call, p.index, bytes(p.data).hex()) for p in res[s]] == want


def test_crc_kats():
    L = CO.lib()
    for hx in synth.OTA_PACKETS:
        d = bytes(L.oracle_swap_bit_order(b) for b in bytes.fromhex(hx))
        assert L.oracle_crc16_ccitt(d[2:], 8) == 0
        assert L.oracle_crc16_ccitt(d[2:9] + bytes([d[9] ^ 1]), 8) != 0
