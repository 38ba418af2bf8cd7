"""GPU: nq_group.hip's RCCL branch with MORE THAN ONE RANK on a one-GPU box.

The product loads librccl by name (dlopen("librccl.so.1")); here a stand-in of that name
(tests/fake_rccl/fake_rccl.hip: the nccl* entry points the product uses, for ranks that all live in one
process, made of hipMemcpyAsync + a summing kernel) is put first on the loader's path of a fresh process, so
that every count, offset, datatype, communicator and stream of the RCCL call sites -- which the real library
only ever saw at world 1 on this pool -- runs at world 2, 3 and 8: sparse, dense and overflowing exchange,
S = 16 (u32 sums), and an injected ncclSend failure (the group call must be closed on the way out).  Hit lists =
whole-range handle = oracle (tests/fake_rccl_worker.py).  Sum being sharded: src/niqki_index.cpp:652-661."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE_DIR = os.path.join(ROOT, "tests", "fake_rccl")


@pytest.fixture(scope="module")
def fake_lib():
    so = os.path.join(FAKE_DIR, "librccl.so.1")
    src = os.path.join(FAKE_DIR, "fake_rccl.hip")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", FAKE_DIR])
    return so


@pytest.mark.parametrize("world,exchange,S", [(2, "sparse", 9), (2, "dense", 9), (3, "sparse", 9), (3, "overflow", 9),
                                              (8, "sparse", 9), (8, "dense", 9), (8, "overflow", 9), (2, "sparse", 16),
                                              (2, "dense", 16), (3, "fail", 9)])
def test_rccl_branch_with_several_ranks(fake_lib, world, exchange, S):
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = FAKE_DIR + ":" + env.get("LD_LIBRARY_PATH", "")
    env["NIQKI_GROUP_TRANSPORT"] = "rccl"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fake_rccl_worker.py"), str(world), exchange, str(S)],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "fake-rccl ok" in r.stdout
