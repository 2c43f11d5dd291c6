"""CPU-side checks: the C-ABI library loads and exports every symbol the header declares,
argument errors surface as the reference's exceptions, and no compute happens without a GPU."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "rtldavis_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rd_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from rtldavis_amd import _lib
    L = _lib.lib()
    names = header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/rtldavis_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes table and header disagree"


def test_struct_layouts_match_header():
    import ctypes as C
    from rtldavis_amd import _lib
    assert C.sizeof(_lib.RdConfig) == 5 * 4 + 64
    assert C.sizeof(_lib.RdPacket) == 4 * 4 + 32 + 2 * 8
    assert _lib.RdPacket.data.offset == 16 and _lib.RdPacket.rssi.offset == 48
    assert C.sizeof(_lib.RdTiming) == 24
    assert C.sizeof(_lib.RdParsed) == 6 * 4 + 32 + 2 * 8
    assert C.sizeof(_lib.RdChanConfig) == 4 * 4 + 8 and _lib.RdChanConfig.gain.offset == 16


def test_packet_config_mirrors_reference():
    from rtldavis_amd import dsp
    cfg = dsp.PacketConfig(19200, 14, 16, 80, "1100101110001001", 8192)
    assert (cfg.sample_rate, cfg.block_size2, cfg.preamble_length, cfg.packet_length, cfg.buffer_length) == \
        (268800, 16384, 224, 1120, 16384)
    assert cfg.preamble_str == bytes([1, 1, 0, 0, 1, 0, 1, 1, 1, 0, 0, 0, 1, 0, 0, 1])
    assert dsp.PacketConfig(19200, 14, 16, 80, "1100101110001001").buffer_length == 2048  # default block 512


def test_argument_errors_without_gpu():
    """Size checks come before any device work (dsp.py:32-36,145-149)."""
    from rtldavis_amd import dsp
    dem = dsp.Demodulator(dsp.PacketConfig(19200, 14, 16, 80, "1100101110001001", 8192))  # no HIP init here
    with pytest.raises(ValueError, match="Incompatible array sizes"):
        dem.demodulate(np.zeros(7, dtype=np.uint8))
    with pytest.raises(ValueError, match="Incompatible array sizes"):
        dem.demodulate(np.zeros(7, dtype=np.complex64))
    with pytest.raises(ValueError, match="Incompatible array sizes"):
        dsp.ByteToCmplxLUT().execute(np.zeros(10, np.uint8), np.zeros(4, np.complex128))
    with pytest.raises(ValueError):
        dsp.Demodulator(dsp.PacketConfig(19200, 14, 16, 80, "1100101110001001", 30))  # not a multiple of 4


def test_create_argument_checks_without_gpu():
    """Bad shapes/configs are rejected by the host part of the C ABI before any device work."""
    import ctypes as C
    from rtldavis_amd import _lib, batch, dsp
    good = dsp.PacketConfig(19200, 14, 16, 80, "1100101110001001", 8192)
    for ns, nb in [(0, 1), (1, 0), (-3, 2)]:
        with pytest.raises(ValueError):
            batch.BatchDemodulator(good, ns, nb)
    with pytest.raises(ValueError):  # packet shorter than the preamble
        batch.BatchDemodulator(dsp.PacketConfig(19200, 14, 16, 8, "1100101110001001", 8192), 1, 1)
    with pytest.raises(ValueError):  # preamble symbols must be 0/1
        dsp.Demodulator(dsp.PacketConfig(19200, 14, 4, 80, "1120", 512))
    with pytest.raises(ValueError):
        dsp.MultiDemodulator(good, 0)
    h = C.c_void_p()
    assert _lib.lib().rd_create(None, C.byref(h)) == _lib.RD_ERR_ARG
    assert b"null config" in _lib.lib().rd_last_error()
    # results before run is a state error, not a crash
    b = batch.BatchDemodulator(good, 1, 1)
    n = C.c_int()
    assert _lib.lib().rd_batch_results(b._b, None, 0, C.byref(n)) in (_lib.RD_ERR_STATE, _lib.RD_ERR_DEVICE)


def test_no_silent_cpu_fallback():
    """Without a GPU every compute entry point must raise, never return numbers."""
    from rtldavis_amd import _lib, batch, dsp
    if _lib.lib().rd_device_count() > 0:
        pytest.skip("a GPU is present")
    cfg = dsp.PacketConfig(19200, 14, 16, 80, "1100101110001001", 8192)
    with pytest.raises(_lib.HipError):
        dsp.Demodulator(cfg).demodulate(np.zeros(16384, dtype=np.uint8))
    with pytest.raises(_lib.HipError):
        batch.BatchDemodulator(cfg, 2, 2).demodulate(np.zeros((2, 32768), dtype=np.uint8))
    with pytest.raises(_lib.HipError):
        dsp.quantize(np.zeros(4), np.zeros(4, np.uint8))


def test_channelizer_argument_checks_and_no_fallback():
    """rd_chan_create validates without touching the device; without a GPU the compute calls raise."""
    from rtldavis_amd import _lib, channelizer
    with pytest.raises(ValueError):
        channelizer.Channelizer([914963100], centre_hz=914963100, decim=0)          # no decimation factor
    with pytest.raises(ValueError):
        channelizer.Channelizer([902419338], centre_hz=990000000)                   # outside the captured band
    with pytest.raises(ValueError):
        channelizer.Channelizer([914963100], gain=0.0)
    with pytest.raises(ValueError):
        channelizer.Channelizer([914963100], taps=np.ones(9000))                    # more taps than the kernel stages
    with pytest.raises(ValueError):
        channelizer.Channelizer([914963100], decim=98)                              # decimation not a multiple of 4
    cz = channelizer.Channelizer()                                                  # host state only
    assert cz.n_channels == 51 and cz.shift_hz[25] == 67200 and cz.taps.size == 512
    if _lib.lib().rd_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.HipError):
        cz.upload(np.zeros(2 * 100 * 64, np.uint8))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "rtldavis_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "dsp_oracle" not in text, f


def test_shard_ranges_cover_and_balance():
    from rtldavis_amd.shard import shard_range
    for n, w in [(32768, 8), (4096, 3), (5, 8), (51, 4), (1, 1)]:
        parts = [shard_range(n, w, r) for r in range(w)]
        assert parts[0][0] == 0 and parts[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
        sizes = [hi - lo for lo, hi in parts]
        assert max(sizes) - min(sizes) <= 1


def test_traffic_stamp_of_the_built_demod_kernel():
    """bench.py accepts a profiles/rNN_traffic.json only when its stamp equals the one the build left next to the
    library: a sha256 over the instructions of k_demod_mfma (tools/profile_collect.py::isa_sha256 - the kernel's body
    in hipcc's device assembly, comments and label numbers dropped)."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("profile_collect", os.path.join(ROOT, "tools", "profile_collect.py"))
    pc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pc)
    stamp = pc.kernel_isa_stamp()
    assert len(stamp) == 64 and int(stamp, 16) >= 0, "make -C rtldavis_amd/csrc all writes librtldavis_hip.stamp"
    asm = "\n".join(["_Z5otherv:", "\ts_endpgm", pc.DEMOD_KERNEL_SYMBOL + "v9rd_layout: ; @k", "; %bb.0:", "\ts_load_dword s0, s[4:5], 0x0 ; c",
                     ".LBB7_2:", "\ts_cbranch_scc1 .LBB7_2", "\ts_endpgm", "\t.section x"])
    again = asm.replace("LBB7_", "LBB3_").replace("; c", "; another comment")
    assert pc.isa_sha256(asm) == pc.isa_sha256(again) != ""
    assert pc.isa_sha256(asm.replace("0x0", "0x4")) != pc.isa_sha256(asm)
    assert pc.isa_sha256("_Z5otherv:\n\ts_endpgm\n") == ""
    with open(os.path.join(ROOT, "profiles", "r03_traffic.json")) as fh:
        tj = json.load(fh)
    assert len(tj["kernel_isa_sha256"]) == 64 and tj["traffic_bytes"] > tj["algorithmic_bytes"]
