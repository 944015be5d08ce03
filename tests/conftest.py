import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The several-ranks-on-one-GPU tests fail (instead of skipping) when RCCL refuses the rig: a skipped exchange test
# is an untested exchange.  FK_REQUIRE_RANKS=0 in the environment gives the skip back.
os.environ.setdefault("FK_REQUIRE_RANKS", "1")
# fk_debug_set's alternative code paths (fall-backs the parity tests force) are only open to test processes
os.environ.setdefault("FASTK_AMD_TEST_KNOBS", "1")
# ... and so do the tests that check against the reference itself (oracle/_ref: FastK, Tabex, libfkref.so built from
# /root/reference by oracle/Makefile; the binaries travel to the GPU box with the snapshot): a clone without them
# must not pass by skipping.  FK_REQUIRE_REF=0 gives the skip back (a box that never had the reference sources).
os.environ.setdefault("FK_REQUIRE_REF", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # FASTK_AMD_EMU=1 (tests/test_emu_suite.py sets it for the child pytest it starts): the `gpu` tests run WITHOUT a GPU
    # against tests/csrc/libfastk_emu.so -- the library's own sources compiled for the CPU under tests/csrc/hip_emu.h
    # (work-items as fibers, a stand-in HIP runtime).  Test infrastructure like the oracle: nothing in fastk_amd/ knows
    # about it; the product's loader is pointed at the other file from here, for this process only.
    if os.environ.get("FASTK_AMD_EMU") == "1":
        sys.path.insert(0, os.path.join(ROOT, "tests", "csrc"))
        import build_emu_lib
        import fastk_amd.api as api
        api.LIB_PATH = build_emu_lib.build()


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    """The oracle is test infrastructure: make sure the restatement is compiled."""
    from oracle import orc
    orc.build(ref=False)
    yield
