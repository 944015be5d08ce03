#!/usr/bin/env python3
"""The DEFAULT splitter through the C driver on reads full of ties (homopolymers, tandem repeats, N runs; the generator of
test_exact_splitter_on_low_complexity_reads), with -p, -bc<n>, -c, k = 9 ... 64: FastK_amd on one GPU, -G2, -G4 and the
reference run live agree on .hist, the .ktab stream and the decoded profiles (GPU box).
    python tools/cli_low_complexity_probe.py"""
import os, sys, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
def load(n):
    spec = importlib.util.spec_from_file_location(n, os.path.join(ROOT, "tools", n + ".py")); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m); return m
deg = load("cli_degenerate_probe"); xp = load("exact_prof_low_complexity_probe")
import numpy as np
bad = 0
for k, seed, shape, flags, fq in ((40, 51, (260, (40, 60, 150, 400, 1500, 6000)), ("-t1",), False),
                              (21, 52, (260, (40, 60, 150, 400, 1500, 6000)), ("-t1", "-p"), True),
                              (40, 53, (60, (39, 5000, 30000, 120000)), ("-t2", "-p"), False),
                              (33, 54, (3000, (100, 150, 151, 250)), ("-t1", "-bc8", "-p"), True),
                              (40, 55, (3000, (100, 150, 151, 250)), ("-t1", "-c"), False),
                              (64, 56, (260, (40, 64, 65, 150, 400, 1500)), ("-t1", "-p"), False),
                              (9, 57, (260, (8, 9, 10, 40, 150)), ("-t1", "-p"), False)):
    bases, boff = xp.reads_of(20260000 + seed, *shape)
    reads = [bytes(bases[boff[i]:boff[i + 1] - 1]) for i in range(len(boff) - 1)]
    ok, res = deg.case("low complexity k=%d %s" % (k, " ".join(flags)), reads, k=k, flags=flags, fastq=fq)
    bad += 0 if ok else 1
print("differences:", bad)
sys.exit(1 if bad else 0)
