"""rtldavis_amd - MI355X (gfx950) implementation of rtldavis's IQ -> bits -> packets path.

``rtldavis_amd.dsp`` mirrors ``rtldavis.dsp`` (drop-in for protocol.Parser / worker);
``rtldavis_amd.batch.BatchDemodulator`` demodulates many independent streams per launch.
"""
__all__ = ["dsp", "batch", "synth"]
