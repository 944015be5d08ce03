#!/usr/bin/env python3
"""Facts about the GPU box that size the configs[2] run: host RAM, disk, HBM free, allocation and
copy rates (hipMalloc, hipHostMalloc, H2D, D2H, D2D).  Prints one JSON object."""
import ctypes as C
import json
import os
import shutil
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401  (one HIP runtime for the process)

hip = C.CDLL("libamdhip64.so")
vp = C.c_void_p


def ck(rc, what):
    if rc != 0:
        raise SystemExit("%s failed: %d" % (what, rc))


def main():
    out = {}
    out["cpus"] = os.cpu_count()
    with open("/proc/meminfo") as f:
        mi = dict((l.split(":")[0], int(l.split()[1])) for l in f)
    out["host_mem_total_gb"] = round(mi["MemTotal"] / 1e6, 1)
    out["host_mem_avail_gb"] = round(mi["MemAvailable"] / 1e6, 1)
    for p in ("/tmp", "/dev/shm", os.getcwd()):
        try:
            du = shutil.disk_usage(p)
            out["disk_free_gb:" + p] = round(du.free / 1e9, 1)
        except OSError:
            pass
    ck(hip.hipSetDevice(0), "hipSetDevice")
    free, total = C.c_size_t(), C.c_size_t()
    ck(hip.hipMemGetInfo(C.byref(free), C.byref(total)), "hipMemGetInfo")
    out["hbm_free_gb"] = round(free.value / 1e9, 2)
    out["hbm_total_gb"] = round(total.value / 1e9, 2)

    def dmalloc(n):
        p = vp()
        t0 = time.perf_counter()
        ck(hip.hipMalloc(C.byref(p), C.c_size_t(n)), "hipMalloc %d" % n)
        return p, time.perf_counter() - t0

    for gb in (1, 8, 32):
        p, dt = dmalloc(gb << 30)
        t0 = time.perf_counter()
        ck(hip.hipMemset(p, 0, C.c_size_t(gb << 30)), "memset")
        ck(hip.hipDeviceSynchronize(), "sync")
        dt2 = time.perf_counter() - t0
        t0 = time.perf_counter()
        ck(hip.hipMemset(p, 1, C.c_size_t(gb << 30)), "memset")
        ck(hip.hipDeviceSynchronize(), "sync")
        dt3 = time.perf_counter() - t0
        t0 = time.perf_counter()
        ck(hip.hipFree(p), "free")
        out["hipMalloc_%dGiB_s" % gb] = round(dt, 4)
        out["first_touch_%dGiB_s" % gb] = round(dt2, 4)
        out["second_touch_%dGiB_s" % gb] = round(dt3, 4)
        out["hipFree_%dGiB_s" % gb] = round(time.perf_counter() - t0, 4)

    n = 8 << 30
    a, _ = dmalloc(n)
    b, _ = dmalloc(n)
    ck(hip.hipMemset(a, 1, C.c_size_t(n)), "memset")
    ck(hip.hipMemset(b, 2, C.c_size_t(n)), "memset")
    ck(hip.hipDeviceSynchronize(), "sync")
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        ck(hip.hipMemcpy(b, a, C.c_size_t(n), 3), "d2d")
        ck(hip.hipDeviceSynchronize(), "sync")
        best = min(best, time.perf_counter() - t0)
    out["d2d_copy_GBs_of_payload"] = round(n / best / 1e9, 1)
    out["d2d_copy_GBs_read_plus_write"] = round(2 * n / best / 1e9, 1)

    h = vp()
    t0 = time.perf_counter()
    ck(hip.hipHostMalloc(C.byref(h), C.c_size_t(n), 0), "hipHostMalloc")
    out["hipHostMalloc_8GiB_s"] = round(time.perf_counter() - t0, 3)
    t0 = time.perf_counter()
    C.memset(h, 1, n)
    out["host_memset_8GiB_s"] = round(time.perf_counter() - t0, 3)
    for name, kind, dst, src in (("h2d", 1, a, h), ("d2h", 2, h, a)):
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            ck(hip.hipMemcpy(dst, src, C.c_size_t(n), kind), name)
            ck(hip.hipDeviceSynchronize(), "sync")
            best = min(best, time.perf_counter() - t0)
        out[name + "_pinned_GBs"] = round(n / best / 1e9, 1)
    # pageable D2H for comparison (what fetch_result_table did in round 1)
    import numpy as np
    pg = np.empty(1 << 30, dtype=np.uint8)
    pg[:] = 0
    t0 = time.perf_counter()
    ck(hip.hipMemcpy(vp(pg.ctypes.data), a, C.c_size_t(1 << 30), 2), "d2h pageable")
    out["d2h_pageable_GBs"] = round((1 << 30) / (time.perf_counter() - t0) / 1e9, 1)
    t0 = time.perf_counter()
    ck(hip.hipHostFree(h), "hostfree")
    out["hipHostFree_8GiB_s"] = round(time.perf_counter() - t0, 3)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
