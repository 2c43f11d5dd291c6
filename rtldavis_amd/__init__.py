"""rtldavis_amd - MI355X (gfx950) implementation of rtldavis's IQ -> bits -> packets path.

``rtldavis_amd.dsp`` mirrors ``rtldavis.dsp`` (drop-in for protocol.Parser / worker);
``rtldavis_amd.batch.BatchDemodulator`` demodulates many independent streams per launch.
"""
__all__ = ["dsp", "batch", "synth"]

import os as _os

# Device->host copies of a few MB (the packet records of a batch) go through the SDMA engines instead
# of a blit kernel: ROCclr's default threshold sends copies below 16 MiB to a compute kernel, which
# has to share the CUs with the persistent demod kernel of the next batch (measured on MI355X: the
# 4 MB readback took 420 us instead of 97 and slowed that kernel by 7 %; whole-step throughput +10 %
# with the SDMA path, profiles/r02_readback_sdma.txt).  Read by the HIP runtime when it initialises,
# so it has to be in the environment before the first HIP call of the process; an explicit setting wins.
if "GPU_FORCE_BLIT_COPY_SIZE" not in _os.environ:
    _os.environ["GPU_FORCE_BLIT_COPY_SIZE"] = "0"
    # Too late when something in this process has initialised HIP already (torch.cuda does on first use): the
    # runtime has read its flags by then and the setting silently does nothing.  Say so once.
    import sys as _sys
    _torch = _sys.modules.get("torch")
    try:
        _late = bool(_torch is not None and _torch.cuda.is_initialized())
    except Exception:  # noqa: BLE001
        _late = False
    if _late:
        import warnings as _warnings
        _warnings.warn("rtldavis_amd: HIP was initialised before this import, so GPU_FORCE_BLIT_COPY_SIZE=0 cannot take "
                       "effect: device-to-host copies of the packet records will run as blit kernels beside the demod "
                       "kernel (about 10 % less batch throughput).  Import rtldavis_amd first, or export "
                       "GPU_FORCE_BLIT_COPY_SIZE=0 in the environment.", RuntimeWarning, stacklevel=2)
