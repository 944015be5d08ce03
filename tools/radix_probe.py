#!/usr/bin/env python3
"""(needs a library built with `make -C fastk_amd/csrc ABLATION=1`: the look-back engine and its variants are not in the shipped build)
Measure the radix digit pass in isolation on random records (run on the GPU box).
Prints achieved algorithmic GB/s (2*n*R / avg pass ms) for the real kernel and for the ablated
variants, plus a hipMemcpy device-to-device copy ceiling of the same byte count."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import fastk_amd

def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200_000_000
    rsize = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    nkeys = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    dev = torch.device("cuda", 0)
    ctx = fastk_amd.Context(kmer=40)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    a = torch.randint(0, 256, (n * rsize,), dtype=torch.uint8, device=dev)
    b = torch.empty_like(a)
    out = {}
    # copy ceiling
    torch.cuda.synchronize()
    for _ in range(2): b.copy_(a)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): b.copy_(a)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    out["copy_GBs"] = round(2 * n * rsize / dt / 1e9, 1)
    items = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    ctx._ck(ctx.L.fk_debug_set(ctx.h, b"radix_items", items))
    out["items"] = items
    best = None
    for rep in range(3):
        ctx.lsd_sort(a.data_ptr(), b.data_ptr(), n, rsize, list(range(nkeys)))
        st = ctx.sort_stats()
        ms = st["pass_ms_total"] / max(st["passes"], 1)
        best = ms if best is None else min(best, ms)
    out["stream"] = dict(avg_pass_ms=round(best, 4), passes=st["passes"],
                         GBs=round(2 * n * rsize / (best * 1e-3) / 1e9, 1), hist_ms=round(st["hist_ms"], 3))
    ctx._ck(ctx.L.fk_debug_set(ctx.h, b"radix_engine", 1))
    for variant in (0, 1, 3):
        ctx._ck(ctx.L.fk_debug_set(ctx.h, b"radix_variant", variant))
        best = None
        for rep in range(3):
            ctx.lsd_sort(a.data_ptr(), b.data_ptr(), n, rsize, list(range(nkeys)))
            st = ctx.sort_stats()
            ms = st["pass_ms_total"] / max(st["passes"], 1)
            best = ms if best is None else min(best, ms)
        out["variant%d" % variant] = dict(avg_pass_ms=round(best, 4), passes=st["passes"],
                                          GBs=round(2 * n * rsize / (best * 1e-3) / 1e9, 1),
                                          hist_ms=round(st["hist_ms"], 3))
    ctx._ck(ctx.L.fk_debug_set(ctx.h, b"radix_variant", 0))
    ctx._ck(ctx.L.fk_debug_set(ctx.h, b"radix_engine", 0))
    print(json.dumps(dict(n=n, rsize=rsize, **out)))

if __name__ == "__main__":
    main()
