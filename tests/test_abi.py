"""CPU: the C-ABI library loads and exports every symbol include/niqki_hip.h
declares; without a GPU the product refuses to run (no CPU fallback)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


HEADERS = ("niqki_hip.h", "niqki_hip_bench.h")     # the drop-in boundary; measurement / diagnosis / test support


def declared_symbols(header=None):
    out = set()
    for h in ((header,) if header else HEADERS):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        out |= set(re.findall(r"\b(niqki_[A-Za-z0-9_]+)\s*\(", text))
    return sorted(out)


def test_header_symbols_all_exported(native):
    import ctypes
    L = ctypes.CDLL(native.lib_path())
    syms = declared_symbols()
    assert len(syms) >= 30
    # the two headers do not overlap, and what a host program needs -- the reference's operators, files, dump / load,
    # groups -- is all in the first one: the bench header holds no create / sketch / insert / query / stage / dump call
    main, extra = set(declared_symbols("niqki_hip.h")), set(declared_symbols("niqki_hip_bench.h"))
    assert not (main & extra) and len(extra) <= 20
    for need in ("niqki_create", "niqki_sketch", "niqki_insert", "niqki_query", "niqki_query_sequences", "niqki_matrix_range",
                 "niqki_stage_raw", "niqki_staged_insert", "niqki_staged_query", "niqki_export_dump", "niqki_import_dump",
                 "niqki_group_create", "niqki_group_query", "niqki_sketch_shared", "niqki_pack_fasta", "niqki_select_best_H"):
        assert need in main, need
    for tool in ("niqki_synth_genomes", "niqki_measure_alu", "niqki_profile_read", "niqki_query_survivors"):
        assert tool in extra, tool
    for s in syms:
        assert hasattr(L, s), "libniqki_hip.so lacks %s" % s
    # the ctypes table covers the whole header too
    from niqki_amd import capi
    assert sorted(n for n, _, _ in capi.ABI) == syms


def test_header_cites_the_reference_interfaces():
    text = open(os.path.join(ROOT, "include", "niqki_hip.h")).read()
    for cite in ("src/niqki_index.cpp:335-358", "src/niqki_index.cpp:362-370",
                 "src/niqki_index.cpp:633-687", "src/niqki_index.cpp:42-55"):
        assert cite in text


def test_abi_version_and_min_score(native):
    L = native.lib()
    assert L.niqki_abi_version() == 2
    assert native.min_score(0.9, 10) == 921
    assert native.min_score(0.1, 15) == 3276
    assert L.niqki_status_string(6) == b"no gfx950 device"


def test_public_struct_layouts_match_the_ctypes_table(native, tmp_path):
    """sizeof / offsetof of the header's public structs as a C compiler sees them == the ctypes mirrors in capi.py
    (a struct that grows must bump NIQKI_ABI_VERSION and its mirror together)."""
    import ctypes
    import subprocess
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "niqki_hip.h"\n#include "niqki_hip_bench.h"\n'
                   'int main(void) { printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(niqki_params), sizeof(niqki_raw_batch), '
                   'offsetof(niqki_raw_batch, file_status), sizeof(niqki_stage_info), sizeof(niqki_group_plan), '
                   '(size_t)NIQKI_ABI_VERSION); return 0; }\n')
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    from niqki_amd import capi
    assert got == [ctypes.sizeof(capi.Params), ctypes.sizeof(capi.RawBatch), capi.RawBatch.file_status.offset,
                   ctypes.sizeof(capi.StageInfo), ctypes.sizeof(capi.GroupPlan), native.lib().niqki_abi_version()]


def test_no_gpu_means_no_engine(native):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(native.NiqkiError) as ei:
        native.Engine()
    assert ei.value.code == 6  # NIQKI_E_NODEVICE: there is no CPU path to fall back to


def test_invalid_parameters_rejected_before_touching_a_device(native):
    for kw in (dict(K=32), dict(S=17), dict(W=16), dict(S=16, W=15), dict(H=13, W=12), dict(K=0)):
        with pytest.raises(native.NiqkiError) as ei:
            native.Engine(**kw)
        assert ei.value.code == 1, kw


def test_host_synth_is_deterministic_acgt(native):
    import numpy as np
    a = native.synth_genome_host(5, 3, 0, 0, 1000)
    b = native.synth_genome_host(5, 3, 0, 0, 1000)
    c = native.synth_genome_host(5, 3, 7, 400, 1000)
    assert np.array_equal(a, b) and set(a.tolist()) <= set(b"ACGT")
    d = int((a != c).sum())
    assert 5 <= d <= 60  # ~2.4 % substitutions
    assert np.array_equal(native.synth_genome_host(5, 3, 0, 0, 333), a[:333])


def test_product_never_touches_the_oracle():
    """No file of the product (package, host program, headers) may reference oracle/."""
    bad = []
    for base in ("niqki_amd", "include"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            for fn in fns:
                if fn.endswith((".py", ".h", ".hip", ".cpp", ".hpp", "Makefile")):
                    t = open(os.path.join(dp, fn), errors="ignore").read()
                    # ... nor the stand-in librccl of tests/fake_rccl (the product loads "librccl.so.1" by name only)
                    if re.search(r"oracle|liboracle|nqo_|fake_rccl|fake-rccl|host_san", t):
                        bad.append(os.path.join(dp, fn))
    assert not bad, bad
