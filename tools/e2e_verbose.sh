python - <<'PY'
import numpy as np, os, sys
sys.path.insert(0, os.getcwd())
from oracle import orc
L=150; glen=20_000_000; nreads=50*glen//L
bases, boff = orc.synth_block(20251001, glen, L, 1000, 0, nreads)
mat = np.empty((nreads, 3 + L + 3 + L + 1), dtype=np.uint8)
mat[:, 0:3] = np.frombuffer(b"@r\n", dtype=np.uint8)
mat[:, 3:3 + L] = bases.reshape(nreads, L + 1)[:, :L]
mat[:, 3 + L:6 + L] = np.frombuffer(b"\n+\n", dtype=np.uint8)
mat[:, 6 + L:6 + 2 * L] = ord("I")
mat[:, 6 + 2 * L] = ord("\n")
mat.tofile("/tmp/s.fastq")
PY
cd /tmp
for i in 1 2; do $GRAFT_REPO_ROOT/fastk_amd/bin/FastK_amd -k40 -t1 -T4 -v -Nout s.fastq 2>&1 | tail -4; done
