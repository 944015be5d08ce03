#!/usr/bin/env python3
"""What a digit pass of the stream engine would cost without one of its parts (run on the GPU box; builds the library
with ABLATION=1 first: `make -C fastk_amd/csrc ABLATION=1 -B`, and rebuild without it afterwards).

A hashed two-pass grouping (fk_group_records: the pipeline's sort of the weighted k-mers, R = 12, and of the super-mers,
R = 20) of n random records, scatter kernels timed by the library's own event pairs (fk_get_sort_stats), with the
ablation bits of fk_radix.hip set one at a time -- the OUTPUT IS WRONG, the times bound what removing that part can buy:
  nohash   the next pass's digit is a byte of the record (another byte every pass) instead of a hash of it: what carrying
           a second digit stream from the producer could save at most
  linear   the records leave in tile order: whole lines, no scatter (the bound of any write combining through LDS)
  noperm   no LDS gather by the permutation
  norank   no ballots / LDS atomics
  python tools/scatter_ablation.py [n_r12=462e6] [n_r20=180e6] > profiles/r05_scatter_ablation.json
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("FASTK_AMD_TEST_KNOBS", "1")
import torch            # noqa: E402
import fastk_amd        # noqa: E402

BITS = dict(real=0, nohash=0x100, linear=0x200, noperm=0x400, norank=0x800, nohash_linear=0x300, all=0xf00)


def main():
    sizes = {12: int(float(sys.argv[1])) if len(sys.argv) > 1 else 462_000_000,
             20: int(float(sys.argv[2])) if len(sys.argv) > 2 else 180_000_000}
    dev = torch.device("cuda", 0)
    ctx = fastk_amd.Context(kmer=40)
    out = dict(note="two hashed digit passes (fk_group_records) over random records; ms = scatter kernels only, per pass; "
                    "GBs = 2 n R / ms; every row but `real` computes WRONG output")
    for rsize, n in sizes.items():
        a = torch.randint(0, 2 ** 31 - 1, (n * rsize // 4,), dtype=torch.int32, device=dev)
        b = torch.empty_like(a)
        rows = {}
        for name, bits in BITS.items():
            ctx.debug_set("scatter_abl", bits)
            best = None
            for _ in range(3):
                ctx.group(a.data_ptr(), b.data_ptr(), n, rsize)
                st = ctx.sort_stats()
                ms = st["scatter_ms_total"] / max(st["passes"], 1)
                tot = st["pass_ms_total"] / max(st["passes"], 1)
                if best is None or ms < best[0]:
                    best = (ms, tot)
            rows[name] = dict(scatter_ms=round(best[0], 4), pass_ms_with_helpers=round(best[1], 4),
                              GBs=round(2 * n * rsize / (best[0] * 1e-3) / 1e9, 1))
        ctx.debug_set("scatter_abl", 0)
        for name in rows:
            rows[name]["vs_real"] = round(rows[name]["scatter_ms"] / rows["real"]["scatter_ms"], 3)
        out["R%d" % rsize] = dict(n=n, **rows)
        del a, b
        torch.cuda.empty_cache()
    ctx.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
