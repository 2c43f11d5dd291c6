import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the in-tree shared library is a build artefact (git-ignored): build it when it is missing
    # (hipcc cross-compiles for gfx950 without a GPU); tests never fall back to anything else
    so = os.path.join(ROOT, "rtldavis_amd", "librtldavis_hip.so")
    stamp = os.path.join(ROOT, "rtldavis_amd", "librtldavis_hip.stamp")   # (written by the same make target)
    if not os.path.exists(so) or not os.path.exists(stamp):
        import subprocess
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "rtldavis_amd", "csrc")])


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as fh:
        return json.load(fh)


def load_npz(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def gfloat(v):
    """Inverse of tools/gen_golden.py:fnum."""
    if isinstance(v, str):
        return float(v)
    return float(v)


def assert_calls_equal(got_calls, want_calls, db_tol=1e-3):
    """got_calls: list (per call) of packet-likes with .index/.data/.rssi/.snr;
    want_calls: list (per call) of golden dicts.  Order inside a call matters."""
    assert len(got_calls) == len(want_calls)
    for b, (g, w) in enumerate(zip(got_calls, want_calls)):
        assert [(int(p.index), bytes(p.data).hex()) for p in g] == \
               [(int(p["index"]), p["data"]) for p in w], f"call {b}"
        for p, q in zip(g, w):
            for key in ("rssi", "snr"):
                a, e = float(getattr(p, key)), gfloat(q[key])
                if e != e:
                    assert a != a, f"call {b} {key}"
                else:
                    assert abs(a - e) <= db_tol, f"call {b} {key}: {a} vs {e}"


def dense_calls(sparse, n_calls):
    """streams.json stores only non-empty calls, keyed by call number."""
    return [sparse.get(str(i), []) for i in range(n_calls)]


@pytest.fixture(scope="session")
def golden_streams():
    return load_json("streams.json")
