import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gold():
    vec = np.load(os.path.join(GOLD, "reference_vectors.npz"))
    meta = json.load(open(os.path.join(GOLD, "reference_meta.json")))
    return vec, meta


@pytest.fixture(scope="session")
def po():
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def native():
    """The product library through its C ABI (ctypes)."""
    import niqki_amd
    niqki_amd.lib()
    return niqki_amd


def synth_case(native, m, which="genomes"):
    """Regenerates the seeded genomes / queries of a golden case."""
    seed = m.get("seed", json.load(open(os.path.join(GOLD, "reference_meta.json")))["seed"])
    if which == "genomes":
        f, mm, r = m["fam"], m["mem"], m["rate"]
    else:
        f, mm, r = m["qfam"], m["qmem"], m["qrate"]
    return [native.synth_genome_host(seed, a, b, c, m["len"]) for a, b, c in zip(f, mm, r)]


def family_spec(n_fam, n_mem, lo=16, hi=820, fam0=0):
    fam, mem, rate = [], [], []
    for f in range(n_fam):
        for k in range(n_mem):
            fam.append(fam0 + f)
            mem.append(k)
            rate.append(0 if k == 0 else int(round(lo * (hi / lo) ** ((k - 1) / max(n_mem - 2, 1)))))
    return np.array(fam, np.uint32), np.array(mem, np.uint32), np.array(rate, np.uint32)
