"""CPU: the oracle (oracle/niqki_oracle.c) against the golden vectors the REAL
reference produced (oracle/make_goldens.py -> tests/golden/).  This is what pins
the oracle."""
import hashlib
import os

import numpy as np
import pytest

from conftest import family_spec, synth_case


def test_hash_and_fingerprint_kats(po, gold):
    vec, _ = gold
    # SURVEY.md 8a rows a5/a6 known answers
    assert po.rev64(1) == 0x4179B061E0C0E0D0
    assert po.unrev64(1) == 0xB471E5C8635F305A
    assert po.rev64(0) == 0 and po.unrev64(0) == 0
    assert po.fingerprint(1) == 1
    assert po.fingerprint(1 << 63) == 3840
    assert po.fingerprint((1 << 63) - 1) == 3839
    assert po.fingerprint(0x0001000000000ABC) == 188
    assert po.fingerprint(0) == 0
    L = po.lib()
    for i, x in enumerate(vec["kat_x"].tolist()):
        assert po.rev64(x) == int(vec["kat_rev"][i])
        assert po.unrev64(x) == int(vec["kat_unrev"][i])
        assert po.fingerprint(x, 12, 4) == int(vec["kat_fp_w12h4"][i])
        assert po.fingerprint(x, 8, 3) == int(vec["kat_fp_w8h3"][i])
        assert L.nqo_hash_family(x & 0xFFF, 5) == int(vec["kat_hashfam_step5"][i])
        # the pair is mutually inverse
        assert po.unrev64(po.rev64(x)) == x


def test_select_best_H_and_stale_fingerprints(po, gold):
    """-G: select_best_H / score_H (src/niqki_index.cpp:126-164) and get_fingerprint with the
    constructor's stale mask_M / maximal_remainder, against the reference's own answers."""
    vec, meta = gold
    for S, W, H, G, chosen in meta["select_best_H"]:
        assert po.select_best_H(G, S, W, H) == chosen, (S, W, H, G)
    xs = vec["kat_x"].tolist()
    for tag, (W, H0) in {"kat_fp_stale_w12_h4_g150": (12, 4), "kat_fp_stale_w12_h2_g5e6": (12, 2),
                         "kat_fp_stale_w8_h5_g1e4": (8, 5), "kat_fp_stale_w10_h0_g1": (10, 0)}.items():
        v = vec[tag]
        Hn = int(v[0])
        assert Hn != H0
        for i, x in enumerate(xs):
            assert po.fingerprint_stale(x, W, Hn, H0) == int(v[1 + i]), (tag, hex(x))


def test_min_score_truncation(po):
    assert po.lib().nqo_min_score(0.9, 10) == 921
    assert po.lib().nqo_min_score(0.1, 15) == 3276
    assert po.lib().nqo_min_score(0.0, 15) == 0


@pytest.mark.parametrize("case", ["A", "D1", "D2", "D3", "D4", "G1", "G2", "G3"])
def test_sketch_index_query_dump_vs_reference(po, native, gold, case):
    vec, meta = gold
    m = meta[case]
    p = po.make_params(m["K"], m["S"], m["W"], m["H"], m["J"], genome_size=m.get("G", 0.0))
    if "G" in m:
        assert p.H == m["H_final"] != m["H"]
    assert p.min_score == int(vec[case + "_min_score"][0])
    genomes = synth_case(native, m)
    sk = np.stack([po.compute_sketch(p, g) for g in genomes])
    assert np.array_equal(sk, vec[case + "_sketches"])
    ix = po.Index(p, sk)
    qsk = vec[case + "_qsketches"]
    off = vec[case + "_hit_off"]
    for q in range(qsk.shape[0]):
        hc, hg = ix.query(qsk[q])
        lo, hi = int(off[q]), int(off[q + 1])
        assert np.array_equal(hc, vec[case + "_hit_counts"][lo:hi])
        assert np.array_equal(hg, vec[case + "_hit_gids"][lo:hi])
    # dump bytes: payload + the names the harness used ("g0\n"...)
    raw = ix.dump_bytes() + "".join("g%d\n" % i for i in range(len(genomes))).encode()
    assert len(raw) == m["dump_len"]
    assert hashlib.md5(raw).hexdigest() == m["dump_md5"]
    # load(dump) round trip
    ix2 = po.Index.load_bytes(raw)
    assert ix2.n == len(genomes) and np.array_equal(ix2.gids(), ix.gids())
    if case == "A":
        qs = synth_case(native, m, "queries")
        assert np.array_equal(np.stack([po.compute_sketch(p, q) for q in qs]), qsk)


def test_north_star_parameters_vs_reference(po, native, gold):
    vec, meta = gold
    m = meta["B"]
    p = po.make_params(31, 15, 12, 4, 0.0)
    genomes = synth_case(native, m)
    sk = np.stack([po.compute_sketch(p, g) for g in genomes])
    assert ["%016x" % po.fnv1a64(s) for s in sk] == m["sketch_fnv"]
    assert sk[:, :8].tolist() == m["sketch_head"]
    qs = synth_case(native, m, "queries")
    qsk = np.stack([po.compute_sketch(p, q) for q in qs])
    assert ["%016x" % po.fnv1a64(s) for s in qsk] == m["qsketch_fnv"]
    ix = po.Index(p, sk)
    off = vec["B_hit_off"]
    for q in range(len(qs)):
        hc, hg = ix.query(qsk[q])
        lo, hi = int(off[q]), int(off[q + 1])
        assert np.array_equal(hc, vec["B_hit_counts"][lo:hi])
        assert np.array_equal(hg, vec["B_hit_gids"][lo:hi])


def test_short_and_edge_reads_vs_reference(po, gold):
    vec, meta = gold
    m = meta["C"]
    p = po.make_params(m["K"], m["S"], m["W"], m["H"], 0.0)
    off = vec["C_read_off"]
    for i in range(m["n"]):
        rd = vec["C_reads"][int(off[i]):int(off[i + 1])]
        assert np.array_equal(po.compute_sketch(p, rd), vec["C_sketches"][i]), i


def test_densify_guard_on_reference_hangs(po):
    # poly-A: every canonical k-mer is 0, rev(0) = 0 is even: the reference spins
    # forever; the oracle reports it instead.
    p = po.make_params(31, 12, 10, 4, 0.0)
    sk = po.sketch_accumulate(p, b"A" * 100)
    assert (sk != -1).sum() == 1
    out, rc = po.densify(p, sk)
    assert rc == -1 and np.array_equal(out, sk)
    # an all-empty sketch likewise
    out, rc = po.densify(p, np.full(4096, -1, np.int32))
    assert rc == -1


def test_records_not_longer_than_k_contribute_nothing(po):
    p = po.make_params(31, 10, 12, 4, 0.0)
    for L in (0, 1, 30, 31):
        sk = po.sketch_accumulate(p, b"ACGT" * 8)[:0]  # noqa: F841
        s = po.sketch_accumulate(p, (b"ACGT" * 8)[:L])
        assert (s == -1).all()
    s = po.sketch_accumulate(p, (b"ACGTTGCA" * 5)[:32])
    assert (s != -1).sum() == 1  # exactly one k-mer: the last one is skipped


def test_matrix_equals_pairwise_queries(po, native, gold):
    """query_range counts (bucket co-occurrence) equal the query counts of the
    stored sketches -- the identity the GPU matrix path relies on."""
    vec, meta = gold
    m = meta["A"]
    p = po.make_params(m["K"], m["S"], m["W"], m["H"], m["J"])
    sk = vec["A_sketches"]
    ix = po.Index(p, sk)
    mat = ix.matrix_range(3, 11)
    for t in range(3, 11):
        assert np.array_equal(mat[:, t - 3].astype(np.uint32), ix.counts(sk[t]))


def test_cli_golden_first_line_shape(gold):
    _, meta = gold
    hits = meta["cli"]["hits"].splitlines()
    assert len(hits) == 12 and hits[0].startswith("syn00.fa syn00.fa:1 ")
    assert meta["cli"]["matrix"].startswith("##Names\tsyn00.fa\t")


def test_record_framing_restatement_vs_reference_cli(po, native, gold):
    """Pins oracle/pyoracle.py frame_records (Index::Biogetline + its callers' loops,
    src/niqki_index.cpp:383-430, :890-941) on what the reference CLI printed for the
    hand-made FASTA / FASTQ files with every framing oddity (lines mode, J=0: every
    record lists all 12 genomes with its exact counts)."""
    _, meta = gold
    cli = meta["cli"]
    fam, mem, rate = family_spec(4, 8)
    genomes = [native.synth_genome_host(meta["seed"], int(f), int(m), int(r), 40000)
               for f, m, r in zip(fam[:12], mem[:12], rate[:12])]
    p = po.make_params(31, 10, 12, 4, 0.0)
    ix = po.Index(p, np.stack([po.compute_sketch(p, g) for g in genomes]))
    names = ["syn%02d.fa" % i for i in range(12)]
    for key, ty in (("nasty_fa", "A"), ("nasty_fq", "Q")):
        data = cli[key + "_input"].encode("latin1")
        lines = []
        for _, header, seq in po.frame_records(data, ty, 31):
            hc, hg = ix.query(po.compute_sketch(p, seq))
            lines.append(header.decode("latin1") + " " + "".join("%s:%g " % (names[g], c / 1024) for c, g in zip(hc, hg)))
        assert "\n".join(lines) + "\n" == cli[key], key
    # -i: the headers become the genome names
    data = cli["nasty_fa_input"].encode("latin1")
    recs = po.frame_records(data, "A", 31)
    ix2 = po.Index(po.make_params(31, 10, 12, 4, 0.02), np.stack([po.compute_sketch(p, s) for _, _, s in recs]))
    lines = []
    for i, g in enumerate(genomes):
        hc, hg = ix2.query(po.compute_sketch(p, g), min_score=20)
        lines.append(names[i] + " " + "".join("%s:%g " % (recs[g_][1].decode("latin1"), c / 1024) for c, g_ in zip(hc, hg)))
    assert "\n".join(lines) + "\n" == cli["nasty_idx"]


def test_ecoli_pins(po, gold):
    """the 9 E. coli genomes the reference ships (tests/golden/ecoli)."""
    import gzip
    import os
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ecoli")   # data fixtures
    _, meta = gold
    m = meta["ecoli"]
    p = po.make_params(31, 15, 12, 4, 0.0)
    sks = []
    for fn in m["files"]:
        lines = gzip.open(os.path.join(d, fn), "rb").read().split(b"\n")
        seq = np.frombuffer(b"".join(l for l in lines if not l.startswith(b">")), np.uint8)
        sks.append(po.compute_sketch(p, seq))
    assert ["%016x" % po.fnv1a64(s) for s in sks] == m["sketch_fnv"]
    # values SURVEY.md 8c recorded from its own probe of the reference (its checksum
    # variant differs from nqo_fnv1a64, so only the raw values are compared)
    assert sks[0][:8].tolist() == [1895, 2012, 2088, 1521, 2142, 1625, 2631, 2590] == m["g1_head"]
    ix = po.Index(p, np.stack(sks))
    hc, hg = ix.query(sks[0])
    assert hc.tolist() == m["q1_counts"] == [32768, 31712, 30737, 29845, 28993, 28220, 27415, 26677, 25930]
    assert hg.tolist() == m["q1_gids"]
    # README matrix value (README.md:118-128): ecoli01p vs 02p = 31712/32768
    assert "%g" % (31712 / 32768) == "0.967773"
    # the whole matrix the reference CLI printed for these files (query_matrix / query_range,
    # src/niqki_index.cpp:570-628: min_score 0.1*F, rows in list order, trailing tabs)
    mat = ix.matrix_range(0, 9)
    text = "##Names\t" + "".join(f + "\t" for f in m["files"]) + "\n"
    for a in range(9):
        text += m["files"][a] + "\t" + "".join("%g\t" % ((mat[a, t] / 32768) if mat[a, t] >= 3276 else 0.0) for t in range(9)) + "\n"
    assert text == meta["ecoli_cli"]["matrix"]


def test_reference_binary_loads_our_dump_fixture(tmp_path, po, native, gold):
    """Drop-in check of the dump format in the other direction, in the build container: the REAL
    reference binary (oracle/_ref/niqki_ref, compiled from /root/reference by oracle/Makefile;
    it never travels to the GPU box) loads tests/golden/ours_cli_dump.gz -- the multi-member
    gzip dump OUR host program wrote on a GPU box (tools/make_dump_fixture.py; the GPU suite
    checks that the program still writes it) -- and answers like it does from its own dump."""
    import gzip
    import shutil
    import subprocess
    from conftest import GOLD, make_cli_workdir
    fixture = os.path.join(GOLD, "ours_cli_dump.gz")
    if not os.path.exists(po.REF_BIN_PATH):
        pytest.skip("oracle/_ref/niqki_ref not built (no /root/reference here)")
    assert os.path.exists(fixture)
    _, meta = gold
    td = make_cli_workdir(tmp_path, native, meta)
    shutil.copy(fixture, td / "ours.dump")
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run([po.REF_BIN_PATH, "-L", "ours.dump", "-Q", "fof.txt", "-O", "ref_on_ours.gz"], cwd=td,
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert gzip.open(td / "ref_on_ours.gz", "rb").read().decode() == meta["cli"]["hits_loaded"]


def test_s16_case_vs_reference(po, native):
    """Golden D5 (oracle/make_goldens_s16.py): S = 16, the reference's uint32-counter branch
    (src/niqki_index.cpp:668-682) -- self hits count 2^16."""
    import json
    from conftest import GOLD
    vec = np.load(os.path.join(GOLD, "reference_s16.npz"))
    m = json.load(open(os.path.join(GOLD, "reference_s16.json")))["D5"]
    p = po.make_params(m["K"], m["S"], m["W"], m["H"], m["J"])
    assert p.min_score == int(vec["D5_min_score"][0])
    seed = json.load(open(os.path.join(GOLD, "reference_s16.json")))["seed"]
    genomes = [native.synth_genome_host(seed, a, b, c, m["len"]) for a, b, c in zip(m["fam"], m["mem"], m["rate"])]
    sk = np.stack([po.compute_sketch(p, g) for g in genomes])
    assert np.array_equal(sk, vec["D5_sketches"].astype(np.int32))
    assert np.array_equal(po.compute_sketch(p, vec["D5_short_seq"]), vec["D5_short_sketch"].astype(np.int32))
    ix = po.Index(p, sk)
    qsk = vec["D5_qsketches"].astype(np.int32)
    off = vec["D5_hit_off"]
    for q in range(qsk.shape[0]):
        hc, hg = ix.query(qsk[q])
        assert np.array_equal(hc, vec["D5_hit_counts"][int(off[q]):int(off[q + 1])])
        assert np.array_equal(hg, vec["D5_hit_gids"][int(off[q]):int(off[q + 1])])
    assert int(vec["D5_hit_counts"].max()) == 1 << 16
    raw = ix.dump_bytes() + "".join("g%d\n" % i for i in range(len(genomes))).encode()
    assert len(raw) == m["dump_len"] and hashlib.md5(raw).hexdigest() == m["dump_md5"]


def test_index_build_over_slot_ranges_gives_the_single_thread_arrays(po):
    """nqo_index_build_mt (what the bench and the full-size tests build their large oracle indexes with): the same
    offsets and ids as the plain build, whatever the thread count."""
    import numpy as np
    p = po.make_params(31, 9, 8, 4, 0.1)
    rng = np.random.default_rng(17)
    sk = rng.integers(-1, 1 << 8, (300, 1 << 9)).astype(np.int32)
    sk[7] = -1                                   # an empty sketch
    sk[:, 100] = 5                               # one bucket with every genome
    one = po.Index(p, sk, threads=1)
    for th in (2, 3, 8, 600):
        many = po.Index(p, sk, threads=th)
        assert np.array_equal(one.offsets(), many.offsets()) and np.array_equal(one.gids(), many.gids()), th
        assert np.array_equal(one.counts(sk[3]), many.counts(sk[3]))
