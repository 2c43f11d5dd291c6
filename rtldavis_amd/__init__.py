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
_os.environ.setdefault("GPU_FORCE_BLIT_COPY_SIZE", "0")
