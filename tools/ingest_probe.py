#!/usr/bin/env python3
"""Where the wall time of FastK_amd's ingest goes: the configs[1] FASTA file (or a FASTQ of short reads) on the RAM
disk, then FastK_amd -v under several settings of its reader threads (FASTK_AMD_READERS, FASTK_AMD_PIECE,
FASTK_AMD_MMAP, FASTK_AMD_DEVICE_TEXT), each run twice; prints the -v lines of the faster run.

  python tools/ingest_probe.py [--scale 1.0] [--fastq] --set READERS=32 --set "MMAP=1 PIECE=268435456"
"""
import argparse
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--fastq", action="store_true", help="150 bp reads in a FASTQ file instead of 15 kbp reads in FASTA")
    ap.add_argument("--set", action="append", default=[], help="space-separated NAME=VALUE (FASTK_AMD_ is prefixed)")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--sleep", type=float, default=0., help="seconds to wait before every run (the driver wipes the device "
                    "memory the previous process released in the background; see tools/probe/malloc_probe.cpp)")
    args = ap.parse_args()
    import bench
    import fastk_amd
    glen = int(3000e6 * args.scale)
    L = 150 if args.fastq else 15000
    nreads = int(50 * glen / L)
    ctx = fastk_amd.Context(kmer=40)
    d = tempfile.mkdtemp(prefix="fkprobe", dir="/dev/shm")
    try:
        path = os.path.join(d, "reads.fastq" if args.fastq else "reads.fasta")
        bench.write_synth_file(ctx, path, args.fastq, 20251001, glen, L, 1000, nreads)
        ctx.close()
        exe = os.path.join(ROOT, "fastk_amd", "bin", "FastK_amd")
        for setting in [""] + args.set:
            env = dict(os.environ)
            for kv in setting.split():
                k, v = kv.split("=")
                env["FASTK_AMD_" + k] = v
            best = None
            for _ in range(args.reps):
                time.sleep(args.sleep)
                t0 = time.perf_counter()
                p = subprocess.run([exe, "-v", "-k40", "-t4", "-T32", "-M256", "-N" + os.path.join(d, "out"), path],
                                   env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
                dt = time.perf_counter() - t0
                if p.returncode != 0:
                    print("[%s] FAILED: %s" % (setting, p.stderr[-400:]))
                    break
                print("   run: %.3f s" % dt)
                if best is None or dt < best[0]:
                    best = (dt, p.stderr)
            if best:
                print("==== [%s] %.3f s" % (setting or "default", best[0]))
                lines = best[1].splitlines()
                lines = [x for x in lines if "so far" not in x] + [x for x in lines if "so far" in x][-1:]
                for line in lines:
                    if any(w in line for w in ("reader threads", "Wall s", "so far", "Device ms", "  chunk ", ".ktab part", "fk_write_ktab_device", "release")):
                        print("   " + line.strip())
            sys.stdout.flush()
    finally:
        subprocess.run(["rm", "-rf", d])


if __name__ == "__main__":
    main()
