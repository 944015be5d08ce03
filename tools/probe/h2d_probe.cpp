// Host-to-device rate by block size and by what the host did to the block before: hipcc -O2 -o h2d_probe h2d_probe.cpp -lpthread
// (why do the reader threads' 17 MB pushes cross PCIe at ~31 GB/s when blocks of 1 GB reach 57?)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <sys/mman.h>
#include <thread>
#include <vector>
static double wall() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
int main()
{ const size_t total = (size_t) 8 << 30;
  void *d; hipMalloc(&d, total);
  char *p = (char *) aligned_alloc(1 << 21, total); madvise(p, total, MADV_HUGEPAGE);
  { std::vector<std::thread> th; const int T = 8;
    for (int t = 0; t < T; t++) th.emplace_back([=]() { for (size_t i = total / T * t; i < total / T * (t + 1); i += 4096) p[i] = 1; });
    for (auto &x : th) x.join();
  }
  hipHostRegister(p, total, hipHostRegisterDefault);
  hipStream_t s[2]; hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking); hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking);
  hipMemcpyAsync(d, p, total, hipMemcpyHostToDevice, s[0]); hipStreamSynchronize(s[0]);
  for (size_t blk : { (size_t) 4 << 20, (size_t) 17 << 20, (size_t) 64 << 20, (size_t) 256 << 20, (size_t) 1 << 30 })
    { // (a) one copy at a time, waited for (what fk_push_packed does)
      double t0 = wall();
      for (size_t o = 0; o + blk <= total; o += blk) { hipMemcpyAsync((char *) d + o, p + o, blk, hipMemcpyHostToDevice, s[0]); hipStreamSynchronize(s[0]); }
      double t1 = wall();
      // (b) all queued on one stream, one wait
      for (size_t o = 0; o + blk <= total; o += blk) hipMemcpyAsync((char *) d + o, p + o, blk, hipMemcpyHostToDevice, s[0]);
      hipStreamSynchronize(s[0]);
      double t2 = wall();
      // (c) alternating over two streams, one wait each
      int k = 0;
      for (size_t o = 0; o + blk <= total; o += blk, k ^= 1) hipMemcpyAsync((char *) d + o, p + o, blk, hipMemcpyHostToDevice, s[k]);
      hipStreamSynchronize(s[0]); hipStreamSynchronize(s[1]);
      double t3 = wall();
      // (d) the block rewritten by a CPU thread right before its copy (dirty in that core's cache), waited for
      double tw = 0.;
      double t4 = wall();
      for (size_t o = 0; o + blk <= total; o += blk)
        { double a = wall(); memset(p + o, (int) (o >> 20) & 0x7f, blk); tw += wall() - a;
          hipMemcpyAsync((char *) d + o, p + o, blk, hipMemcpyHostToDevice, s[0]); hipStreamSynchronize(s[0]);
        }
      double t5 = wall();
      const double g = (double) (total / blk * blk) / 1e9;
      printf("block %5zu MB: waited-for %.1f GB/s, queued %.1f GB/s, two streams %.1f GB/s, rewritten first %.1f GB/s (copies alone)\n",
             blk >> 20, g / (t1 - t0), g / (t2 - t1), g / (t3 - t2), g / (t5 - t4 - tw));
    }
  // many threads writing their own slices while copies run (memory traffic beside the DMA)
  { const size_t blk = (size_t) 17 << 20; const int T = 48;
    volatile int stop = 0;
    char *q = (char *) aligned_alloc(1 << 21, (size_t) T << 26);
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++) th.emplace_back([=, &stop]() { char *b = q + ((size_t) t << 26); while (!stop) memset(b, t, (size_t) 1 << 26); });
    double t0 = wall();
    for (size_t o = 0; o + blk <= total; o += blk) { hipMemcpyAsync((char *) d + o, p + o, blk, hipMemcpyHostToDevice, s[0]); hipStreamSynchronize(s[0]); }
    double t1 = wall();
    stop = 1; for (auto &x : th) x.join();
    printf("block 17 MB waited-for while %d threads write memory: %.1f GB/s\n", T, (double) (total / blk * blk) / 1e9 / (t1 - t0));
  }
  return 0;
}
