#!/usr/bin/env python3
"""exact_parts on LONG reads with a sort memory so small that the reference cuts the input into several buckets: the
segments of the exact splitter and the trie walk of the scheme together (GPU box).  FastK_amd -x -M1 against the
reference's -M1, every output file.     python tools/exact_long_reads_buckets_probe.py [mbases=200] [read_len=15000]"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402
from tests import util  # noqa: E402


def run(mb=200.0, L=15000, extra=()):
    """extra: further options for both programs (("-p",): profiles too -- the .prof / .pidx parts are compared as well)"""
    nreads = int(mb * 1e6 / L)
    bases, boff = orc.synth_block(20250905, int(mb * 1e6 / 50), L, 2000, 0, nreads)
    d = tempfile.mkdtemp(prefix="fkxl", dir="/dev/shm")
    bad = 0
    try:
        for sub in ("ref", "ours"):
            os.mkdir(os.path.join(d, sub))
        path = os.path.join(d, "ref", "x.fasta")
        util.write_fastx(path, bases, boff, False)
        os.link(path, os.path.join(d, "ours", "x.fasta"))
        for T, mem in ((4, 1), (3, 2)):
            for sub in ("ref", "ours"):
                for f in os.listdir(os.path.join(d, sub)):
                    if f != "x.fasta":
                        os.remove(os.path.join(d, sub, f))
            p = subprocess.run([os.path.join(orc.REF_DIR, "FastK"), "-k40", "-t1", "-T%d" % T, "-M%d" % mem, "-v", "-P" + os.path.join(d, "ref")] + list(extra) + [
                                os.path.join(d, "ref", "x.fasta")], cwd=os.path.join(d, "ref"), capture_output=True, text=True)
            parts = [l for l in (p.stdout + p.stderr).split("\n") if "part" in l.lower() or "bucket" in l.lower()]
            q = subprocess.run([os.path.join(ROOT, "fastk_amd", "bin", "FastK_amd"), "-k40", "-t1", "-T%d" % T, "-M%d" % mem, "-x", "-v"] + list(extra) + [
                                os.path.join(d, "ours", "x.fasta")], cwd=os.path.join(d, "ours"), capture_output=True, text=True)
            if p.returncode != 0 or q.returncode != 0:
                print("-T%d -M%d: reference rc %d, ours rc %d: %s" % (T, mem, p.returncode, q.returncode, (q.stdout + q.stderr)[-400:]))
                bad += 1
                continue
            names = sorted(f for f in os.listdir(os.path.join(d, "ref")) if f != "x.fasta")
            diff = [f for f in names if not os.path.exists(os.path.join(d, "ours", f))
                    or util.sha_file(os.path.join(d, "ref", f)) != util.sha_file(os.path.join(d, "ours", f))]
            ours_b = [l for l in (q.stdout + q.stderr).split("\n") if "bucket" in l.lower() or "segments" in l.lower()]
            print("-T%d -M%d: %d files, different: %s" % (T, mem, len(names), diff or "none"))
            if os.environ.get("FK_PROBE_VERBOSE"):
                print("  reference:", (p.stdout + p.stderr)[:1500])
                print("  ours:", (q.stdout + q.stderr)[:1500])
            bad += 1 if diff else 0
    finally:
        subprocess.run(["rm", "-rf", d])
    return bad


def main():
    bad = run(float(sys.argv[1]) if len(sys.argv) > 1 else 200.0, int(sys.argv[2]) if len(sys.argv) > 2 else 15000,
              tuple(sys.argv[3:]))
    print("differences:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
