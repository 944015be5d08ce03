#!/usr/bin/env python3
"""VERDICT r5 item 1(a), the gate: if the weighted k-mers were binned by the VALUE of a longer minimizer (every copy of a
k-mer then shares its bin without any pass over the W records), how even are 48 x 65,536 such bins at BASELINE
configs[2] (3 Gbp genome, k = 40)?  A scale model: a random genome of 30 Mbp, windows of w = k - m + 1 m-mer starts,
bins of 960 genomic k-mer positions on average (= 7,080 weighted k-mers at configs[2]: W / positions = 7.4), and m chosen
so that lambda = genome positions per canonical m-mer value matches: configs[2] with m = 14 has lambda = 22 (between
the rows m = 10 and 11 here), with m = 16 lambda = 1.4 (row m = 13 here: 0.9).  Printed per m: the coefficient of
variation of the bin loads, the share of bins above 8192 / 7080 = 1.157 x the mean (more than one fill of the
aggregation's LDS table) and above 1.5 x, the largest bin, and the mean length of a minimizer's domain in k-mers
(pieces per super-mer = 1 + (n - 1) / that).  Output: profiles/r06_minbin_census.txt."""
import numpy as np, sys
from scipy.ndimage import minimum_filter1d
rng = np.random.default_rng(7)
G = 30_000_000
K = 40
def run(m, G, nbins_per_960=960.0, seed=7):
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, G, dtype=np.uint8)
    # forward / reverse-complement m-mer codes
    n = G - m + 1
    f = np.zeros(n, dtype=np.uint64); r = np.zeros(n, dtype=np.uint64)
    for i in range(m):
        f = (f << np.uint64(2)) | g[i:i+n].astype(np.uint64)
        r = r | ((np.uint64(3) - g[i:i+n].astype(np.uint64)) << np.uint64(2*i))
    c = np.minimum(f, r)
    h = (c * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(32)
    h ^= h >> np.uint64(15); h = (h * np.uint64(0x2C1B3C6D)) & np.uint64(0xffffffff); h ^= h >> np.uint64(13)
    w = K - m + 1
    # window min over [j, j+w-1]
    M = minimum_filter1d(h, size=w, mode='nearest', origin=-(w//2))[:n-w+1] if False else None
    # explicit: min over next w values
    hh = h.astype(np.uint64)
    M = hh[:n-w+1].copy()
    for d in range(1, w):
        np.minimum(M, hh[d:d+n-w+1], out=M)
    nk = len(M)
    key = (M * np.uint64(0x9E3779B1)) & np.uint64(0xffffffff)
    key ^= key >> np.uint64(16); key = (key * np.uint64(0x85EBCA6B)) & np.uint64(0xffffffff)
    nbins = int(nk / nbins_per_960)
    b = (key * np.uint64(nbins)) >> np.uint64(32)
    cnt = np.bincount(b.astype(np.int64), minlength=nbins)
    changes = np.count_nonzero(M[1:] != M[:-1])
    lam = G / (4**m/2)
    return dict(m=m, w=w, lam=round(lam,2), nbins=nbins, mean=cnt.mean(), cv=cnt.std()/cnt.mean(),
                p_over_1p157=float((cnt > 1.157*cnt.mean()).mean()), p_over_1p5=float((cnt>1.5*cnt.mean()).mean()),
                max_over_mean=cnt.max()/cnt.mean(), domain_len=nk/(changes+1))
for m in (10, 11, 12, 13, 14, 16):
    print(run(m, G))
