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


def make_cli_workdir(td, native, meta):
    """The synthetic input files of the CLI goldens (oracle/make_goldens.py wrote the reference
    CLI's outputs for exactly these): 12 FASTA genomes of 40 kbp, their list, 30 reads."""
    fam, mem, rate = family_spec(4, 8)
    genomes = [native.synth_genome_host(meta["seed"], int(f), int(m), int(r), 40000)
               for f, m, r in zip(fam, mem, rate)]
    names = []
    for i, g in enumerate(genomes[:12]):
        fn = "syn%02d.fa" % i
        with open(td / fn, "wb") as f:
            f.write(b">syn%02d\n" % i)
            for a in range(0, len(g), 70):
                f.write(bytes(g[a:a + 70]) + b"\n")
        names.append(fn)
    (td / "fof.txt").write_text("\n".join(names) + "\n")
    with open(td / "reads.fa", "wb") as f:
        for i in range(30):
            f.write(b">read%d some text\n" % i + bytes(genomes[i % 12][200 * i:200 * i + 150]) + b"\n")
    return td
