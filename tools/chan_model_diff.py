import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from rtldavis_amd import channelizer as CZ, synth
from oracle import channelizer_oracle as CHO
chans = [0, 7, 24, 25, 26, 50]
off = [CZ.US_CHANNELS_HZ[c] - CZ.DEFAULT_CENTRE_HZ for c in chans]
raw, _ = synth.synth_wideband([11, 12, 13, 14, 15, 16], off, 3 * 8192)
cz = CZ.Channelizer([CZ.US_CHANNELS_HZ[c] for c in chans])
cz.upload(raw)
got = cz.run_host()
want = CHO.channelize(raw, cz.shift_hz, cz.taps, cz.decim, cz.out_rate, cz.gain)
d = got.astype(np.int32) - want.astype(np.int32)
print("max |diff|", np.abs(d).max(), "fraction differing", (d != 0).mean(), "of", d.size)
