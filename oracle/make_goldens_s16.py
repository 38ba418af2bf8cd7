#!/usr/bin/env python3
"""Golden case D5 from the REAL reference (oracle/_ref): S = 16, the reference's lF > 15 branch
with uint32 counters (src/niqki_index.cpp:668-682).  Kept apart from make_goldens.py so that the
round-1 fixtures stay byte-identical.  TEST INFRASTRUCTURE ONLY; run in the build container:
    make -C oracle && python oracle/make_goldens_s16.py
Writes tests/golden/reference_s16.npz (+ .json): sketches of 8 genomes of 300 kbp (4.6 k-mers per
slot: ~1 % of the cells are filled by densification) and of a 500-base record (densification does
nearly all the work), hit lists of self queries (count 2^16 = F, which no u16 counter holds),
mutants and an unrelated genome, the dump's md5."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import make_goldens as mg  # noqa: E402
from oracle import pyoracle as po  # noqa: E402


def main():
    po.build()
    assert po.have_ref(), "oracle/_ref missing: run make -C oracle (needs /root/reference)"
    out = {}
    K, S, W, H, J, L = 31, 16, 8, 3, 0.2, 300_000
    fam, mem, rate = mg.family_spec(2, 4, lo=30, hi=500)
    genomes = mg.synth(fam, mem, rate, L)
    qf = np.array(list(fam) + [0, 1, 7], np.uint32)
    qm = np.array(list(mem) + [60, 61, 0], np.uint32)
    qr = np.array(list(rate) + [80, 400, 0], np.uint32)
    queries = mg.synth(qf, qm, qr, L)
    meta = {"seed": mg.SEED, "D5": {"K": K, "S": S, "W": W, "H": H, "J": J, "len": L,
                                     "fam": fam.tolist(), "mem": mem.tolist(), "rate": rate.tolist(),
                                     "qfam": qf.tolist(), "qmem": qm.tolist(), "qrate": qr.tolist()}}
    meta["D5"].update(mg.ref_index_case("D5", K, S, W, H, J, genomes, queries, out))
    # a short record: 470 k-mers in 65 536 slots
    r = po.Ref(K=K, S=S, W=W, H=H, J=J)
    short = genomes[3][1000:1500].copy()
    out["D5_short_seq"] = short
    out["D5_short_sketch"] = r.compute_sketch(short)
    r.close()
    # the reference CLI at -S 16 on the CLI golden inputs (tests/conftest.py make_cli_workdir)
    import gzip
    import pathlib
    import subprocess
    import tempfile
    import niqki_amd
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import make_cli_workdir
    base_meta = json.load(open(os.path.join(mg.GOLD, "reference_meta.json")))
    env = dict(os.environ, OMP_NUM_THREADS="1")
    with tempfile.TemporaryDirectory() as td:
        td = make_cli_workdir(pathlib.Path(td), niqki_amd, base_meta)
        subprocess.check_call([po.REF_BIN_PATH, "-I", "fof.txt", "-Q", "fof.txt", "-S", "16", "-W", "8", "-J", "0.1", "-O", "h.gz"],
                              cwd=td, env=env, stdout=subprocess.DEVNULL)
        subprocess.check_call([po.REF_BIN_PATH, "-M", "fof.txt", "-S", "16", "-W", "8", "-J", "0.1", "-O", "m.gz"],
                              cwd=td, env=env, stdout=subprocess.DEVNULL)
        meta["cli_s16"] = {"hits": gzip.open(td / "h.gz", "rb").read().decode(), "matrix": gzip.open(td / "m.gz", "rb").read().decode()}
    out = {k: (v.astype(np.int16) if k.endswith("sketches") or k.endswith("_sketch") else v) for k, v in out.items()}   # W = 8: cells in -1..255
    np.savez_compressed(os.path.join(mg.GOLD, "reference_s16.npz"), **out)
    json.dump(meta, open(os.path.join(mg.GOLD, "reference_s16.json"), "w"), indent=1)
    print("hits per query:", np.diff(out["D5_hit_off"]).tolist(), "max count", int(out["D5_hit_counts"].max()))


if __name__ == "__main__":
    main()
