// How fast can a large host buffer be made ready for a device-to-host copy?  hipcc -O2 -o pin_probe pin_probe.cpp -lpthread
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <sys/mman.h>
#include <thread>
#include <vector>
static double wall() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static void touch(char *p, size_t n, int T)
{ std::vector<std::thread> th;
  for (int t = 0; t < T; t++)
    th.emplace_back([=]() { size_t lo = n / T * t, hi = (t == T - 1) ? n : n / T * (t + 1); for (size_t i = lo; i < hi; i += 4096) p[i] = 0; });
  for (auto &x : th) x.join();
}
int main(int argc, char **argv)
{ const size_t GB = (size_t) 1 << 30, n = (argc > 1 ? atoi(argv[1]) : 8) * GB;
  void *d; hipMalloc(&d, n); hipMemset(d, 1, n); hipDeviceSynchronize();
  double t0 = wall(); void *h1; hipHostMalloc(&h1, n, hipHostMallocDefault); double t1 = wall();
  hipMemcpy(h1, d, n, hipMemcpyDeviceToHost); double t2 = wall();
  printf("hipHostMalloc %.1f GB: %.3f s (%.1f GB/s); D2H %.3f s (%.1f GB/s)\n", n / 1e9, t1 - t0, n / 1e9 / (t1 - t0), t2 - t1, n / 1e9 / (t2 - t1));
  hipHostFree(h1);
  for (int T : {1, 8, 32})
    { t0 = wall(); char *p = (char *) aligned_alloc(1 << 21, n); madvise(p, n, MADV_HUGEPAGE); touch(p, n, T); t1 = wall();
      hipError_t e = hipHostRegister(p, n, hipHostRegisterDefault); t2 = wall();
      hipMemcpy(p, d, n, hipMemcpyDeviceToHost); double t3 = wall();
      printf("touch with %2d threads %.3f s, hipHostRegister %.3f s (%s), D2H %.3f s (%.1f GB/s)\n", T, t1 - t0, t2 - t1, hipGetErrorString(e), t3 - t2, n / 1e9 / (t3 - t2));
      hipHostUnregister(p); free(p);
    }
  { t0 = wall(); char *p = (char *) aligned_alloc(1 << 21, n); madvise(p, n, MADV_HUGEPAGE); t1 = wall();
    hipMemcpy(p, d, n, hipMemcpyDeviceToHost); double t2 = wall();
    printf("pageable, untouched: D2H %.3f s (%.1f GB/s)\n", t2 - t1, n / 1e9 / (t2 - t1));
    hipMemcpy(p, d, n, hipMemcpyDeviceToHost); double t3 = wall();
    printf("pageable, touched:   D2H %.3f s (%.1f GB/s)\n", t3 - t2, n / 1e9 / (t3 - t2));
    free(p);
  }
  // bounce: two pinned 256 MB buffers + T copy threads
  { const size_t C = (size_t) 256 << 20; void *b[2]; hipHostMalloc(&b[0], C); hipHostMalloc(&b[1], C);
    hipStream_t s; hipStreamCreate(&s); hipEvent_t ev[2]; hipEventCreate(&ev[0]); hipEventCreate(&ev[1]);
    for (int T : {4, 8, 16})
      { t0 = wall(); char *p = (char *) aligned_alloc(1 << 21, n); madvise(p, n, MADV_HUGEPAGE);
        size_t nc = (n + C - 1) / C;
        hipMemcpyAsync(b[0], d, C < n ? C : n, hipMemcpyDeviceToHost, s); hipEventRecord(ev[0], s);
        for (size_t i = 0; i < nc; i++)
          { if (i + 1 < nc) { size_t o = (i + 1) * C, l = (n - o < C) ? n - o : C; hipMemcpyAsync(b[(i + 1) & 1], (char *) d + o, l, hipMemcpyDeviceToHost, s); hipEventRecord(ev[(i + 1) & 1], s); }
            hipEventSynchronize(ev[i & 1]);
            size_t o = i * C, l = (n - o < C) ? n - o : C; char *src = (char *) b[i & 1];
            std::vector<std::thread> th;
            for (int t = 0; t < T; t++) th.emplace_back([=]() { size_t lo = l / T * t, hi = (t == T - 1) ? l : l / T * (t + 1); memcpy(p + o + lo, src + lo, hi - lo); });
            for (auto &x : th) x.join();
          }
        t1 = wall();
        printf("bounce 2 x 256 MB, %2d copy threads: %.3f s (%.1f GB/s) incl. allocation\n", T, t1 - t0, n / 1e9 / (t1 - t0));
        free(p);
      }
  }
  return 0;
}
