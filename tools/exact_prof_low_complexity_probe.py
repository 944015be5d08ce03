#!/usr/bin/env python3
"""`FastK_amd -x -p` against the reference's `-p` on reads full of ties (the generator of
test_exact_splitter_on_low_complexity_reads): every output file, the .prof / .pidx bytes included (GPU box).
    python tools/exact_prof_low_complexity_probe.py"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402
from tests import util  # noqa: E402


def reads_of(seed, nreads, lengths):
    src = open(os.path.join(ROOT, "tests", "test_gpu_parity.py")).read()
    i = src.index("def _low_complexity_reads")
    j = src.index("@pytest.mark.parametrize", i)
    ns = {"np": np, "orc": orc}
    exec(src[i:j], ns)
    return ns["_low_complexity_reads"](seed, nreads, lengths)


def run():
    bad = 0
    for case in ((40, 4, 31, (260, (40, 60, 150, 400, 1500, 6000)), False),
                                     (21, 2, 32, (260, (40, 60, 150, 400, 1500, 6000)), True),
                                     (51, 3, 33, (60, (39, 5000, 30000, 120000)), False),
                                     (12, 1, 34, (200, (5, 12, 13, 100, 900)), False),
                                     (40, 4, 35, (9000, (100, 150, 151, 250)), False),      # (several input threads)
                                     (31, 6, 36, (9000, (100, 150, 151, 250)), True),
                                     (40, 8, 37, (700, (150, 2000, 9000)), False),
                                     # several input files: plain (one stream of bytes cut at record starts), compressed (whole files)
                                     (40, 4, 38, (9000, (100, 150, 151, 250)), False, 3, False),
                                     (40, 3, 39, (9000, (100, 150, 151, 250)), True, 2, False),
                                     (33, 4, 40, (3000, (100, 150, 151, 250)), True, 3, True),
                                     (40, 2, 41, (3000, (100, 150, 151, 250)), False, 5, True)):
        k, T, seed, shape, fastq = case[:5]
        nfiles, gz = (case[5], case[6]) if len(case) > 5 else (1, False)
        bases, boff = reads_of(20260000 + seed, *shape)
        d = tempfile.mkdtemp(prefix="fkxp")
        try:
            out = {}
            for sub, cmd in (("ref", [os.path.join(orc.REF_DIR, "FastK"), "-k%d" % k, "-t1", "-T%d" % T, "-p"]),
                             ("ours", [os.path.join(ROOT, "fastk_amd", "bin", "FastK_amd"), "-k%d" % k, "-t1", "-T%d" % T, "-p", "-x"]),
                             ("shim", [os.path.join(orc.REF_DIR, "FastK_gpu"), "-k%d" % k, "-t1", "-T%d" % T, "-p"])):
                os.mkdir(os.path.join(d, sub))
                paths = []
                nr = len(boff) - 1
                for fi in range(nfiles):
                    lo, hi = fi * nr // nfiles, (fi + 1) * nr // nfiles
                    path = os.path.join(d, sub, "xyzuv"[fi] + (".fastq" if fastq else ".fasta"))
                    if fastq:
                        orc.write_fastq(path, bases[boff[lo]:boff[hi]], boff[lo:hi + 1] - boff[lo])
                    else:
                        orc.write_fasta(path, bases[boff[lo]:boff[hi]], boff[lo:hi + 1] - boff[lo], width=0)
                    if gz:
                        subprocess.run(["gzip", "-1", path], check=True)
                        path += ".gz"
                    paths.append(path)
                p = subprocess.run(cmd + (["-P" + os.path.join(d, sub)] if sub != "ours" else []) + paths, cwd=os.path.join(d, sub),
                                   capture_output=True, text=True, env=dict(os.environ, FASTK_AMD_EXACT="1"))
                out[sub] = (p.returncode, (p.stdout + p.stderr)[-300:])
            if any(v[0] != 0 for v in out.values()):
                print("k %d T %d: rc %s" % (k, T, out))
                bad += 1
                continue
            inputs = set(os.path.basename(q) for q in paths)
            names = sorted(f for f in os.listdir(os.path.join(d, "ref")) if f not in inputs)
            for sub in ("ours", "shim"):
                diff = [f for f in names if not os.path.exists(os.path.join(d, sub, f))
                        or util.sha_file(os.path.join(d, "ref", f)) != util.sha_file(os.path.join(d, sub, f))]
                diff += ["+" + f for f in sorted(os.listdir(os.path.join(d, sub))) if f not in names and f not in inputs]
                print("k %d T %d %d %s file(s) %s: %d files, different: %s" % (k, T, nfiles, "gz" if gz else "plain", sub, len(names), diff or "none"))
                bad += 1 if diff else 0
        finally:
            subprocess.run(["rm", "-rf", d])
    return bad


if __name__ == "__main__":
    b = run()
    print("differences:", b)
    sys.exit(1 if b else 0)
