"""CPU-only checks of the C-ABI: the library loads, exports every symbol include/fastk_amd.h
declares, the host-only helpers agree with the oracle's widths, and the product refuses to run
without a GPU (no CPU fallback)."""
import os
import re
import subprocess

import numpy as np
import pytest

import fastk_amd
from oracle import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _build_lib():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "fastk_amd", "csrc")])


def test_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "fastk_amd.h")).read()
    declared = set(re.findall(r"\b(fk_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations found"
    L = fastk_amd.load_library()
    for name in sorted(declared):
        assert hasattr(L, name), name
    assert declared == set(fastk_amd.EXPORTS)


@pytest.mark.parametrize("k", [8, 21, 31, 40, 51, 64, 100, 128])
def test_widths_match_reference_formulas(k):
    w = fastk_amd.widths(k)
    P = orc.params(k)
    for f in ("kmer", "max_super", "smer_bytes", "slen_bytes", "smer_word", "kmer_bytes",
              "kmer_word"):
        assert getattr(w, f) == getattr(P, f), f
    assert w.smer_stride % 4 == 0 and w.smer_stride >= w.smer_word
    assert w.kmer_stride % 4 == 0 and w.kmer_stride >= w.kmer_word


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(fastk_amd.FastKError):
        fastk_amd.Context(kmer=40)


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "fastk_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".c", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.replace("no CPU", ""), os.path.join(dirpath, f)


def test_bench_roofline_record_of_the_reference_stage():
    """bench.py's roofline object when no pass runs over the weighted k-mers (fk_result.nrefs > 0, round 6): the graded
    kernel is the table sort's digit pass; figures checked on a synthetic result (no GPU involved)."""
    import types
    import bench
    loc = types.SimpleNamespace(nrefs=2_786_531_096, ntable=3_000_010_434, passes_final=4, ms_scatter_final=70.7544,
                                ms_pass_final=75.4956, launches_super=96, passes_super=2, nsuper=8_633_378_085,
                                ms_scatter_super=191.07, ms_pass_super=202.68, launches_kmer=144, passes_kmer=3,
                                ms_scatter_kmer=33.51, ms_pass_kmer=44.32, nweighted=22_204_340_673)
    w = types.SimpleNamespace(kmer_word=12, smer_word=20)
    r = bench.roofline_record(loc, w, 2, 5415.9, 5617.2)
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["traffic"] is None
    assert r["launches_per_step"] == 4 and r["records_per_launch"] == loc.ntable
    assert abs(r["achieved"] - 2.0 * loc.ntable * 12 * 4 / 70.7544e-3 / 1e9) < 1.0
    assert abs(r["frac"] - r["achieved"] / 8000.0) < 1e-3 and 0.5 < r["frac"] < 0.52
    assert r["kmer_grouping"]["launches_over_W"] == 0 and r["kmer_grouping"]["references"] == loc.nrefs
    assert r["reference_sort"]["launches"] == 144 and r["supermer_pass"]["launches"] == 96
    loc.nrefs = 0                                   # the hashed grouping: round 5's record
    loc.ncollapsed = loc.ntable
    r5 = bench.roofline_record(loc, w, 2, 5415.9, 5617.2)
    assert "weighted k-mer records" in r5["kernel"] and r5["launches_per_step"] == 144
