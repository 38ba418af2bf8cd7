"""niqki_amd -- MI355X-native NIQKI sketch/query engine.

The product is libniqki_hip.so (hand-written HIP kernels for gfx950 behind the C
ABI of include/niqki_hip.h) and the C++17 host program `niqki` built on it
(niqki_amd/host).  This Python package is only the thin ctypes view of that C
ABI used by the tests and by bench.py; it holds no compute and no fallback.
"""
from .capi import (Engine, NiqkiError, Params, lib, lib_path, build_native,  # noqa: F401
                   MEM_HOST, MEM_DEVICE, SEQ_PAD, KC_SKETCH, KC_DENSIFY,
                   KC_GATHER, KC_HITS, KC_BUILD, KC_INGEST, KC_EXCHANGE, min_score, synth_genome_host, pack_fasta, unpack_fasta,
                   Group, group_slot_range, group_new_id, group_plan, row_stride)
