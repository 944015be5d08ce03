#!/usr/bin/env python3
"""`FastK_amd -x` and the reference's main() over the shim (FASTK_AMD_EXACT=1) with the reference's other options --
-bc<n>, -c, -t<n>, with and without -p -- against the reference run live on reads full of ties: every output file
(GPU box).     python tools/exact_flags_probe.py"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402
from tests import util  # noqa: E402
import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location("xp", os.path.join(ROOT, "tools", "exact_prof_low_complexity_probe.py"))
_xp = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_xp)


def run(cases=None):
    bad = 0
    shapes = {"short": (260, (40, 60, 150, 400, 1500, 6000)), "many": (9000, (100, 150, 151, 250))}
    cases = cases or [(40, 4, 41, "short", False, ("-bc8",)), (40, 4, 42, "short", False, ("-c",)),
                      (40, 3, 43, "short", True, ("-bc5", "-p")), (40, 2, 44, "short", False, ("-c", "-p")),
                      (25, 4, 45, "many", False, ("-t3",)), (25, 4, 46, "many", True, ("-t2", "-p")),
                      (40, 4, 47, "many", False, ("-c", "-bc3", "-t2", "-p")), (51, 5, 48, "many", True, ("-bc10", "-t4"))]
    for k, T, seed, shape, fastq, flags in cases:
        bases, boff = _xp.reads_of(20260000 + seed, *shapes[shape])
        d = tempfile.mkdtemp(prefix="fkxf")
        try:
            out = {}
            fn = "x.fastq" if fastq else "x.fasta"
            for sub, cmd in (("ref", [os.path.join(orc.REF_DIR, "FastK")]), ("ours", [os.path.join(ROOT, "fastk_amd", "bin", "FastK_amd"), "-x"]),
                             ("shim", [os.path.join(orc.REF_DIR, "FastK_gpu")])):
                os.mkdir(os.path.join(d, sub))
                path = os.path.join(d, sub, fn)
                (orc.write_fastq(path, bases, boff) if fastq else orc.write_fasta(path, bases, boff, width=0))
                args = ["-k%d" % k, "-T%d" % T] + list(flags) + ([] if any(f.startswith("-t") for f in flags) else ["-t1"])
                p = subprocess.run(cmd + args + (["-P" + os.path.join(d, sub)] if sub != "ours" else []) + [path], cwd=os.path.join(d, sub),
                                   capture_output=True, text=True, env=dict(os.environ, FASTK_AMD_EXACT="1"))
                out[sub] = (p.returncode, (p.stdout + p.stderr)[-300:])
            if any(v[0] != 0 for v in out.values()):
                print("k %d T %d %s: rc %s" % (k, T, " ".join(flags), out))
                bad += 1
                continue
            names = sorted(f for f in os.listdir(os.path.join(d, "ref")) if f != fn)
            for sub in ("ours", "shim"):
                diff = [f for f in names if not os.path.exists(os.path.join(d, sub, f))
                        or util.sha_file(os.path.join(d, "ref", f)) != util.sha_file(os.path.join(d, sub, f))]
                diff += ["+" + f for f in sorted(os.listdir(os.path.join(d, sub))) if f not in names and f != fn]
                print("k %d T %d %-18s %s: %d files, different: %s" % (k, T, " ".join(flags), sub, len(names), diff or "none"))
                bad += 1 if diff else 0
        finally:
            subprocess.run(["rm", "-rf", d])
    return bad


if __name__ == "__main__":
    b = run()
    print("differences:", b)
    sys.exit(1 if b else 0)
