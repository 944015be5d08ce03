// What device memory costs to get.  On this driver a fresh hipMalloc is cleared before it is handed out, at ~33 GB/s
// (8 GB: 0.24 s), and memory that was freed is wiped before it can be handed out again -- the first hipMalloc after
// freeing 200 GB waits ~6 s.  This probe times 200 GB taken as 8 GB slabs by 1, 2, 4 and 8 threads at once, and as
// slabs of other sizes, to see whether the clearing runs in parallel.
//   hipcc -O2 -o malloc_probe malloc_probe.cpp -lpthread && ./malloc_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <atomic>
#include <mutex>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void pass(int nthreads, size_t slab, size_t total)
{ std::vector<void *> slabs;
  std::mutex m;
  std::atomic<long> left((long) (total / slab));
  std::vector<std::thread> th;
  const double t0 = now();
  double first = 0;
  for (int t = 0; t < nthreads; t++)
    th.emplace_back([&]()
      { hipSetDevice(0);
        while (left.fetch_sub(1) > 0)
          { void *p = NULL;
            double a = now();
            if (hipMalloc(&p, slab) != hipSuccess) { printf("  hipMalloc failed\n"); break; }
            a = now() - a;
            std::lock_guard<std::mutex> g(m);
            if (slabs.empty()) first = a;
            slabs.push_back(p);
          }
      });
  for (auto &t : th) t.join();
  const double dt = now() - t0;
  printf("%d thread(s), slabs of %.2f GB: %zu slabs in %.3f s = %.1f GB/s (first call %.3f s)\n", nthreads, slab / 1073741824.,
         slabs.size(), dt, slabs.size() * (double) slab / 1e9 / dt, first);
  double f0 = now();
  for (void *p : slabs) hipFree(p);
  // the wipe of what was just freed is paid by the next allocation: take it here so that every pass starts alike
  void *p = NULL; double a = now(); hipMalloc(&p, 64 << 20); a = now() - a; hipFree(p);
  printf("    freed in %.3f s; next small hipMalloc took %.3f s\n", a > 0 ? f0 = now() - f0 - a : 0, a);
}

int main(int argc, char **argv)
{ hipSetDevice(0);
  const size_t total = 200ull << 30;
  pass(1, 8ull << 30, total);
  pass(2, 8ull << 30, total);
  pass(4, 8ull << 30, total);
  pass(8, 8ull << 30, total);
  pass(1, 1ull << 30, total);
  pass(1, 40ull << 30, total);
  pass(8, 1ull << 30, total);
  return 0;
}
