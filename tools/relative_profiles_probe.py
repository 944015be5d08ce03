#!/usr/bin/env python3
"""-p:<table> (profiles of one read set against the table of another) on reads full of ties: FastK_amd, FastK_amd -x and
the reference run live; the decoded profiles must agree (GPU box).     python tools/relative_profiles_probe.py"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402
import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location("xp", os.path.join(ROOT, "tools", "exact_prof_low_complexity_probe.py"))
_xp = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_xp)


def run():
    bad = 0
    for k, T, cut in ((40, 4, 1), (21, 2, 3), (51, 3, 2)):
        a = _xp.reads_of(20260061 + k, 600, (40, 60, 150, 400, 1500))
        bset = _xp.reads_of(20260062 + k, 300, (40, 60, 150, 400, 1500))
        # half of b's reads are a's, so that many k-mers are in the table
        d = tempfile.mkdtemp(prefix="fkrp")
        try:
            res = {}
            for sub, exe, extra in (("ref", os.path.join(orc.REF_DIR, "FastK"), []), ("ours", os.path.join(ROOT, "fastk_amd", "bin", "FastK_amd"), []),
                                    ("exact", os.path.join(ROOT, "fastk_amd", "bin", "FastK_amd"), ["-x"])):
                w = os.path.join(d, sub)
                os.mkdir(w)
                orc.write_fasta(os.path.join(w, "a.fasta"), a[0], a[1], width=0)
                with open(os.path.join(w, "b.fasta"), "wb") as f:
                    f.write(open(os.path.join(w, "a.fasta"), "rb").read()[:200000].rsplit(b">", 1)[0])
                    tmp = os.path.join(w, "t.fasta")
                    orc.write_fasta(tmp, bset[0], bset[1], width=0)
                    f.write(open(tmp, "rb").read())
                    os.remove(tmp)
                pdir = ["-P" + w] if sub == "ref" else []
                p1 = subprocess.run([exe, "-k%d" % k, "-t%d" % cut, "-T%d" % T] + extra + pdir + [os.path.join(w, "a.fasta")], cwd=w, capture_output=True, text=True)
                p2 = subprocess.run([exe, "-k%d" % k, "-T%d" % T, "-p:a"] + extra + pdir + [os.path.join(w, "b.fasta")], cwd=w, capture_output=True, text=True)
                if p1.returncode or p2.returncode:
                    res[sub] = ("rc", p1.returncode, p2.returncode, (p2.stdout + p2.stderr)[-300:])
                else:
                    nr, nb, npos, sha = orc.profiles_digest_files(w, "b")
                    res[sub] = (nr, npos, sha)        # (the encoded bytes differ by design: the reference's follow its work panels)
            ok = all(res[x] == res["ref"] for x in res) and res["ref"][0] != "rc"
            print("k %d -t%d: %s %s" % (k, cut, "ok" if ok else "DIFFERENT", "" if ok else res))
            bad += 0 if ok else 1
        finally:
            subprocess.run(["rm", "-rf", d])
    return bad


if __name__ == "__main__":
    b = run()
    print("differences:", b)
    sys.exit(1 if b else 0)
