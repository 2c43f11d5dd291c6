"""INTEGRATION.md option A: the reference's own protocol.Parser running on rtldavis_amd.dsp through a
module swap.  CPU only - constructing the Parser / Demodulator must not touch the GPU (the parent
process builds one before it forks the worker, runners/rtlsdr.py:30, __main__.py:277).  Needs the
reference checkout, which exists in the build container only: skipped elsewhere."""
import importlib
import os
import sys

import numpy as np
import pytest

REF_SRC = "/root/reference/src"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF_SRC, "rtldavis")),
                                reason="reference checkout not present (GPU box)")


@pytest.fixture()
def swapped():
    saved = {k: v for k, v in sys.modules.items() if k == "rtldavis" or k.startswith("rtldavis.")}
    for k in saved:
        del sys.modules[k]
    sys.path.insert(0, REF_SRC)
    sys.dont_write_bytecode, old_flag = True, sys.dont_write_bytecode  # the reference tree is read-only
    try:
        import rtldavis  # noqa: F401  (the package itself: no dsp import yet)
        import rtldavis_amd.dsp as hip_dsp
        sys.modules["rtldavis.dsp"] = hip_dsp  # what INTEGRATION.md section 2 adds to rtldavis/__init__.py
        protocol = importlib.import_module("rtldavis.protocol")
        yield protocol, hip_dsp
    finally:
        sys.dont_write_bytecode = old_flag
        sys.path.remove(REF_SRC)
        for k in [k for k in sys.modules if k == "rtldavis" or k.startswith("rtldavis.")]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_reference_parser_builds_on_the_hip_dsp_without_a_gpu(swapped):
    protocol, hip_dsp = swapped
    p = protocol.Parser(symbol_length=14)  # protocol.py:115-116 -> dsp.Demodulator(cfg)
    assert type(p.demodulator) is hip_dsp.Demodulator
    assert type(p.cfg) is hip_dsp.PacketConfig
    c = p.cfg
    assert (c.bit_rate, c.symbol_length, c.block_size, c.preamble_length, c.buffer_length) == (19200, 14, 8192, 224, 16384)
    from rtldavis_amd import _lib
    assert _lib.lib().rd_demod_inflight(p.demodulator._h) == 0
    # size errors are the reference's (dsp.py:145-149) and come before any device work
    with pytest.raises(ValueError, match="Incompatible array sizes"):
        p.demodulator.demodulate(np.zeros(10, np.uint8))
    # parse() of an empty list touches nothing on the demodulator (protocol.py:282-337)
    assert p.parse([]) == []


def test_demodulator_accepts_the_references_packet_config(swapped):
    """A PacketConfig built by the reference's own (unswapped) class is accepted as is."""
    _protocol, hip_dsp = swapped
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_dsp_plain", os.path.join(REF_SRC, "rtldavis", "dsp.py"))
    ref_dsp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_dsp)
    ref_cfg = ref_dsp.PacketConfig(19200, 14, 16, 80, "1100101110001001", 8192)
    dem = hip_dsp.Demodulator(ref_cfg)  # host state only
    assert dem.cfg is ref_cfg
    assert dem.quantized.shape == (ref_cfg.buffer_length,) and not dem.quantized.any()  # zeros before the first call
