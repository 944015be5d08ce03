"""The CPU restatement (oracle/) against the committed golden vectors that were captured from
the reference FastK build (tests/golden/make_golden.py).  CPU only."""
import hashlib
import os

import numpy as np
import pytest

from oracle import orc
from tests import util


@pytest.mark.parametrize("name", util.golden_names())
def test_oracle_matches_reference_golden(name, tmp_path):
    case, bases, boff = util.load_case(name)
    res = orc.fastk(case["k"], bases, boff, cutoff=case["cutoff"], nthreads=case["T"])
    util.check_against_golden(case, res.hist, res.max_inst, res.table)
    # stricter: the files the oracle writes are byte-identical to the reference's, part by part
    orc.write_outputs(res, case["cutoff"], case["T"], str(tmp_path), "x")
    for fname, digest in case["expected"]["file_sha256"].items():
        got = hashlib.sha256(open(os.path.join(str(tmp_path), fname), "rb").read()).hexdigest()
        assert got == digest, fname
    t = orc.read_ktab(os.path.join(str(tmp_path), "x"))
    assert t["part_sizes"] == case["expected"]["ktab"]["part_sizes"]


def test_oracle_matches_reference_at_configs0(tmp_path):
    """BASELINE.json configs[0] at its stated size (1 M x 150 bp reads of a 10 Mbp genome, k=40 -t1 -T4):
    the restatement reproduces every file digest of the reference run (make_golden.py --large)."""
    case, bases, boff = util.load_case("configs0_k40_t1_T4")
    assert len(boff) - 1 == 1000000
    res = orc.fastk(case["k"], bases, boff, cutoff=case["cutoff"], nthreads=case["T"])
    util.check_against_golden(case, res.hist, res.max_inst, res.table)
    orc.write_outputs(res, case["cutoff"], case["T"], str(tmp_path), "x")
    for fname, digest in case["expected"]["file_sha256"].items():
        assert util.sha_file(os.path.join(str(tmp_path), fname)) == digest, fname


def test_oracle_follows_the_reference_scheme_at_configs0_with_small_sort_memory(tmp_path):
    """SURVEY 8(a) D3 / D4.  BASELINE configs[0] with -M1: the reference cuts the input into two buckets (Determine_Scheme,
    assign_pieces with its unseeded drand48) and the hidden part files by bucket 0's weighted k-mers.  The restated
    scheme + bucket loop reproduce every file digest of that run (make_golden.py --only=configs0_k40_t1_T4_M1)."""
    case, bases, boff = util.load_case("configs0_k40_t1_T4_M1")
    res = orc.fastk_parts(case["k"], bases, boff, int(case["ref_extra"][0][2:]) * 1000000000, cutoff=case["cutoff"],
                          nthreads=case["T"])
    assert res.nparts == 2
    util.check_against_golden(case, res.hist, res.max_inst, res.table)
    orc.write_outputs(res, case["cutoff"], case["T"], str(tmp_path), "x")
    for fname, digest in case["expected"]["file_sha256"].items():
        assert util.sha_file(os.path.join(str(tmp_path), fname)) == digest, fname


@pytest.mark.parametrize("name", ["edge_k40_t1_T4", "edge_k21_t2_T3", "synth_tiny_k40_t1_T2"])
def test_brute_force_definition_matches_golden(name):
    case, bases, boff = util.load_case(name)
    res = orc.brute(case["k"], bases, boff, cutoff=case["cutoff"])
    util.check_against_golden(case, res.hist, res.max_inst, res.table)


def test_oracle_stage_pipeline_consistency():
    """distribute -> msd sort -> kmer_list -> msd sort -> count, stage by stage, equals orc_fastk
    and the LSD engine orders records exactly like the MSD engine on full keys."""
    case, bases, boff = util.load_case("synth_tiny_k40_t1_T2")
    k = case["k"]
    P = orc.params(k)
    smers, inst = orc.distribute(P, bases, boff)
    assert inst == sum(max(0, int(boff[i + 1] - boff[i]) - 1 - k + 1) for i in range(len(boff) - 1))
    ss = orc.msd_sort(smers, P.smer_word)
    ls = orc.lsd_sort(smers, list(range(P.smer_word - 1, -1, -1)))
    assert np.array_equal(ss, ls)
    kl, ovf, nd = orc.kmer_list(P, ss)
    ks = orc.msd_sort(kl, P.kmer_bytes)
    res = orc.count_sorted(P, ks, case["cutoff"])
    util.check_against_golden(case, res.hist, res.max_inst + ovf, res.table)


def test_empty_and_short_inputs():
    for reads in ([], ["acgt"], ["a" * 39], ["n" * 100]):
        bases, boff = orc.block_from_reads(reads)
        res = orc.fastk(40, bases, boff, cutoff=1)
        assert res.ninst == 0 and res.ntable == 0 and res.hist.sum() == 0
    bases, boff = orc.block_from_reads(["acgtacgtacgtacgtacgtacgtacgtacgtacgtacgt"])
    res = orc.fastk(40, bases, boff, cutoff=1)
    assert res.ninst == 1 and res.ntable == 1 and res.hist[1] == 1


@pytest.mark.parametrize("name", util.golden_names())
def test_oracle_profiles_match_golden_digest(name):
    """Decoded -p profiles captured from the reference (make_golden.py) against the oracle's per-read
    counts, and the codec: the reference's own bytes of the first reads decode to them."""
    case, bases, boff = util.load_case(name)
    k = case["k"]
    g = case["expected"]["prof"]
    table = orc.fastk(k, bases, boff, cutoff=1).table
    counts = orc.profile_counts(k, bases, boff, table)
    assert len(counts) == g["nreads"]
    assert orc.profiles_digest(counts) == g["decoded_sha256"]
    for hx, c in zip(g["first_ref_hex"], counts):
        assert orc.profile_decode(bytes.fromhex(hx)) == c.tolist()
    assert sum(len(orc.profile_encode(c)) for c in counts) <= g["ref_bytes"]
