#!/usr/bin/env python3
"""Test harness (not shipped): builds tests/csrc/libfastk_emu.so -- the WHOLE library (every fastk_amd/csrc/*.hip: kernels
and host halves, the same C-ABI) compiled for the CPU with g++: tests/csrc/hip_emu.h stands in for the kernel language
(work-items as fibers, a workgroup at a time) and, with -DFK_EMU_FULL, for the HIP runtime (device memory is host memory,
streams and events do nothing, a launch runs the kernel to its end).  It exists so that the parity tests that need only a
small input can run WITHOUT a GPU (tests/test_emu_suite.py); nothing in fastk_amd/ knows about it, and the product still
fails loudly without a device."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SRC = os.path.join(ROOT, "fastk_amd", "csrc")
OUT = os.path.join(HERE, "libfastk_emu.so")
OBJ = os.path.join(HERE, "_emu_obj")


def build(force=False):
    os.makedirs(OBJ, exist_ok=True)
    deps = [os.path.join(HERE, "hip_emu.h"), os.path.join(SRC, "fk_common.h"), os.path.join(ROOT, "include", "fastk_amd.h"),
            os.path.join(ROOT, "include", "fk_synth.h")]
    newest = max(os.path.getmtime(d) for d in deps)
    objs = []
    procs = []
    for hip in sorted(glob.glob(os.path.join(SRC, "*.hip"))):
        name = os.path.basename(hip)[:-4]
        o = os.path.join(OBJ, name + ".o")
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(newest, os.path.getmtime(hip)):
            cmd = ["g++", "-std=c++17", "-O1", "-g0", "-fPIC", "-DFK_HOST_EMU", "-DFK_EMU_FULL", "-w",
                   "-I", SRC, "-I", HERE, "-I", os.path.join(HERE, "emu_include"), "-x", "c++", "-c", hip, "-o", o]
            if name == "fk_api":
                cmd.insert(1, "-DFK_EMU_DEFINE")
            procs.append((name, subprocess.Popen(cmd, stderr=subprocess.PIPE, text=True)))
            if len(procs) >= 6:
                _wait(procs)
    _wait(procs)
    if force or not os.path.exists(OUT) or any(os.path.getmtime(o) > os.path.getmtime(OUT) for o in objs):
        subprocess.check_call(["g++", "-shared", "-o", OUT] + objs + ["-lpthread", "-ldl"])
    return OUT


def build_driver(name):
    """host/<name>.c linked against libfastk_emu.so: the C drivers of the CLI tests, on the CPU"""
    lib = build()
    bindir = os.path.join(HERE, "_emu_bin")
    os.makedirs(bindir, exist_ok=True)
    exe = os.path.join(bindir, name)
    srcs = [os.path.join(SRC, "host", name + ".c")]
    if name == "FastK_amd":
        srcs.append(os.path.join(SRC, "host", "input_formats.c"))
    if not os.path.exists(exe) or os.path.getmtime(exe) < max([os.path.getmtime(lib)] + [os.path.getmtime(x) for x in srcs]):
        subprocess.check_call(["gcc", "-O2", "-w", "-o", exe] + srcs + ["-L" + HERE, "-lfastk_emu", "-lz", "-lpthread", "-lstdc++",
                               "-Wl,-rpath," + HERE])
    return exe


def _wait(procs):
    while procs:
        name, p = procs.pop(0)
        err = p.communicate()[1]
        if p.returncode != 0:
            sys.stderr.write(err[-4000:])
            raise SystemExit("tests/csrc/build_emu_lib.py: %s.hip does not compile for the CPU" % name)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
