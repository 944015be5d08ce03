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
