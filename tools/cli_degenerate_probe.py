#!/usr/bin/env python3
"""Degenerate input files through the C driver (GPU box): FastK_amd on one GPU, FastK_amd -G2 / -G4 (ranks sharing the GPU)
and the reference itself -- .hist bytes and the .ktab canonical stream must agree; with -p the decoded profiles.
    python tools/cli_degenerate_probe.py
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402
from tests import util  # noqa: E402

EXE = os.path.join(ROOT, "fastk_amd", "bin", "FastK_amd")
REF = os.path.join(orc.REF_DIR, "FastK")
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
bad = 0


def rnd(rng, n):
    return bytes(ACGT[rng.integers(0, 4, size=n)])


def outputs(d, root="x"):
    h = util.sha_file(os.path.join(d, root + ".hist")) if os.path.exists(os.path.join(d, root + ".hist")) else None
    t = None
    if os.path.exists(os.path.join(d, root + ".ktab")):
        kt = orc.read_ktab(os.path.join(d, root))
        t = (kt["stream_sha256"], kt["nels"])
    pf = None
    if os.path.exists(os.path.join(d, root + ".prof")):
        nr, nb, npos, sha = orc.profiles_digest_files(d, root)
        pf = (nr, npos, sha)
    return h, t, pf


def run(cmd, cwd, env=None):
    p = subprocess.run(cmd, cwd=cwd, env=env, capture_output=True, text=True, timeout=600)
    return p.returncode, (p.stdout + p.stderr)[-400:]


def case(name, reads, k=40, flags=("-t1",), fastq=False):
    global bad
    d = tempfile.mkdtemp(prefix="fkdeg")
    path = os.path.join(d, "x.fastq" if fastq else "x.fasta")
    with open(path, "wb") as f:
        for i, r in enumerate(reads):
            if fastq:
                f.write(b"@r%d\n" % i + r + b"\n+\n" + b"I" * len(r) + b"\n")
            else:
                f.write(b">r%d\n" % i + r + b"\n")
    res = {}
    for label, cmd, env in (("ref", [REF, "-k%d" % k, "-T4", "-P" + d] + list(flags), None),
                            ("one", [EXE, "-k%d" % k, "-T4"] + list(flags), None),
                            ("G2", [EXE, "-k%d" % k, "-T4", "-G2"] + list(flags), dict(os.environ, FK_RANKS_SHARE_GPU="1")),
                            # (-M16: the four ranks of the rig share one GPU's memory; a rank plans with all of it otherwise)
                            ("G4", [EXE, "-k%d" % k, "-T4", "-G4", "-M16"] + list(flags), dict(os.environ, FK_RANKS_SHARE_GPU="1"))):
        sub = os.path.join(d, label)
        os.mkdir(sub)
        p2 = os.path.join(sub, os.path.basename(path))
        os.link(path, p2)
        rc, tail = run(cmd + [p2], sub, env)
        res[label] = (rc,) + (outputs(sub) if rc == 0 else (tail,))
    if res["ref"][0] < 0:                     # (the reference itself dies on this input: ours must agree among themselves)
        print("    (reference FastK ended with signal %d)" % -res["ref"][0])
        res.pop("ref")
        res["ref"] = res["one"]
    ok = all(res[x][0] == res["ref"][0] for x in res) and (res["ref"][0] != 0 or all(res[x][1:] == res["ref"][1:] for x in res))
    print("%-42s %s" % (name, "ok" if ok else "DIFFERENT"))
    if not ok:
        bad += 1
        for x, v in res.items():
            print("    %-4s %s" % (x, str(v)[:300]))
    subprocess.run(["rm", "-rf", d])
    return ok, res


def main():
    rng = np.random.default_rng(11)
    case("one read", [rnd(rng, 500)])
    case("one read of exactly k", [rnd(rng, 40)])
    case("three reads", [rnd(rng, 100), rnd(rng, 41), rnd(rng, 3000)])
    case("only reads shorter than k", [rnd(rng, 39), rnd(rng, 5), rnd(rng, 1)])
    case("reads shorter than k among others", [rnd(rng, 20), rnd(rng, 400), rnd(rng, 39), rnd(rng, 400)])
    case("all N", [b"N" * 100, b"N" * 1000])
    case("fewer reads than ranks (2 reads)", [rnd(rng, 200), rnd(rng, 200)])
    case("one long read, 200 kbp", [rnd(rng, 200000)])
    case("same k-mer everywhere", [b"A" * 5000] * 7)
    case("one read, fastq, -t3", [rnd(rng, 500)] * 5, flags=("-t3",), fastq=True)
    case("three reads with -p", [rnd(rng, 100), rnd(rng, 30), rnd(rng, 3000)], flags=("-t1", "-p"))
    case("a short read first, -p", [rnd(rng, 30), rnd(rng, 20), rnd(rng, 500)], flags=("-t1", "-p"))
    case("short reads only, -p", [rnd(rng, 30), rnd(rng, 20)], flags=("-t1", "-p"))
    case("three reads with -p -M1", [rnd(rng, 100), rnd(rng, 30), rnd(rng, 3000)], flags=("-t1", "-p", "-M1"))
    case("five reads with -t2 -p", [rnd(rng, 100)] * 2 + [rnd(rng, 30), rnd(rng, 3000), rnd(rng, 41)], flags=("-t2", "-p"))
    case("k 21, 1000 short reads", [rnd(rng, int(n)) for n in rng.integers(1, 60, size=1000)], k=21)
    print("differences:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
