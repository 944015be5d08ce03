// Does the HIP runtime write into host memory it has already given back?   hipcc -O2 -o runtime_churn_probe runtime_churn_probe.cpp
// Round 5: tests/fuzz_parity.py in 32 processes on one GPU found, about once per 2,000 iterations, an ALIGNED 32-BIT ZERO in
// freshly built Python data (the reads before they ever reached the library, the FASTQ text of an iteration, a numpy
// temporary) -- the signature round 4 had seen once ("four consecutive bases read 0", and a byte two lower than it was).
// Nothing of libfastk_amd.so is in this program: it only does what fk_create / a few stream operations / fk_destroy do to the
// runtime -- streams, events, device and pinned buffers made and destroyed over and over, small kernels and copies in
// between -- and keeps a ring of malloc'ed canary buffers (all 'a'), a few of them freed and allocated again every round so
// that memory the runtime has just freed comes back as a canary.
//   runtime_churn_probe <seconds> <id> <mode>     mode 0: objects made and destroyed every round
//                                                 mode 1: streams and events made once and kept (no destroy until exit)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <time.h>
#include <vector>
static double wall() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
__global__ void k_touch(unsigned *p, size_t n, unsigned v)
{ for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) p[i] = p[i] * 3u + v; }

struct Canary { unsigned char *p; size_t n; };
static long check(const Canary &c, int id, long round, long &shown)
{ long bad = 0;
  const uint64_t *q = (const uint64_t *) c.p;
  for (size_t i = 0; i < c.n / 8; i++)
    if (q[i] != 0x6161616161616161ull)
      { for (size_t j = 8 * i; j < 8 * i + 8; j++)
          if (c.p[j] != 0x61)
            { bad += 1;
              if (shown++ < 24)
                printf("id %d round %ld: canary of %zu bytes at %p: byte %zu (address mod 64 = %d) reads 0x%02x\n", id, round, c.n,
                       (void *) c.p, j, (int) (((uintptr_t) c.p + j) & 63), c.p[j]);
            }
      }
  return bad;
}

int main(int argc, char **argv)
{ const double secs = argc > 1 ? atof(argv[1]) : 30.;
  const int id = argc > 2 ? atoi(argv[2]) : 0, mode = argc > 3 ? atoi(argv[3]) : 0;
  unsigned rng = 99991u * (unsigned) (id + 1);
  auto rnd = [&]() { rng = rng * 1664525u + 1013904223u; return rng >> 8; };
  std::vector<Canary> ring;
  for (int i = 0; i < 96; i++)
    { Canary c; c.n = (1024 + rnd() % (i % 8 == 0 ? (4u << 20) : (256u << 10))) & ~7ull; c.p = (unsigned char *) malloc(c.n); memset(c.p, 0x61, c.n); ring.push_back(c); }
  hipStream_t ks[2] = { NULL, NULL }; hipEvent_t kev[8] = { NULL };
  long round = 0, bad = 0, shown = 0;
  const double t0 = wall();
  while (wall() - t0 < secs)
    { hipStream_t s[2]; hipEvent_t ev[8];
      if (mode == 0 || ks[0] == NULL)
        { for (int i = 0; i < 2; i++) hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
          for (int i = 0; i < 8; i++) { if (i < 2) hipEventCreateWithFlags(&ev[i], hipEventDisableTiming); else hipEventCreate(&ev[i]); }
          if (mode != 0) { memcpy(ks, s, sizeof(s)); memcpy(kev, ev, sizeof(ev)); }
        }
      else { memcpy(s, ks, sizeof(s)); memcpy(ev, kev, sizeof(ev)); }
      // what a small context does: a few device buffers, two pinned ones, copies through pinned staging, kernels, timing events
      const size_t n = 4096 + rnd() % (1u << 20);
      unsigned *d[4]; unsigned *h[2];
      for (int i = 0; i < 4; i++) hipMalloc((void **) &d[i], n * 4 + (i << 12));
      for (int i = 0; i < 2; i++) hipHostMalloc((void **) &h[i], n * 4, hipHostMallocDefault);
      memset(h[0], 1, n * 4);
      unsigned *pg = (unsigned *) malloc(n * 4);             // a pageable result buffer, like a histogram
      for (int rep = 0; rep < 3; rep++)
        { hipEventRecord(ev[2 + rep], s[0]);
          hipMemcpyAsync(d[0], h[0], n * 4, hipMemcpyHostToDevice, s[0]);
          hipEventRecord(ev[0], s[0]);
          hipStreamWaitEvent(s[1], ev[0], 0);
          hipLaunchKernelGGL(k_touch, dim3(64), dim3(256), 0, s[1], d[0], n, (unsigned) rep);
          hipMemcpyAsync(d[1 + rep], d[0], n * 4, hipMemcpyDeviceToDevice, s[1]);
          hipMemcpyAsync(h[1], d[1 + rep], n * 4, hipMemcpyDeviceToHost, s[1]);
          hipEventRecord(ev[5 + rep], s[1]);
          hipEventRecord(ev[1], s[1]);
          hipEventSynchronize(ev[1]);
          hipMemcpyAsync(pg, d[1 + rep], n * 4, hipMemcpyDeviceToHost, s[1]);
          hipStreamSynchronize(s[1]);
          float ms; hipEventElapsedTime(&ms, ev[2 + rep], ev[5 + rep]);
        }
      free(pg);
      hipDeviceSynchronize();
      for (int i = 0; i < 4; i++) hipFree(d[i]);
      for (int i = 0; i < 2; i++) hipHostFree(h[i]);
      if (mode == 0)
        { for (int i = 0; i < 8; i++) hipEventDestroy(ev[i]);
          for (int i = 0; i < 2; i++) hipStreamDestroy(s[i]);
        }
      // the heap turns over: a few canaries go and come back (what was the runtime's a moment ago may be theirs now)
      for (int k = 0; k < 12; k++)
        { Canary &c = ring[rnd() % ring.size()];
          bad += check(c, id, round, shown);
          free(c.p);
          c.n = (1024 + rnd() % (k == 0 ? (4u << 20) : (256u << 10))) & ~7ull;
          c.p = (unsigned char *) malloc(c.n); memset(c.p, 0x61, c.n);
        }
      if (round % 8 == 7)
        for (auto &c : ring) { const long b = check(c, id, round, shown); if (b) { bad += b; memset(c.p, 0x61, c.n); } }
      round += 1;
    }
  for (auto &c : ring) bad += check(c, id, round, shown);
  printf("id %d mode %d: %ld rounds, %ld canary bytes changed\n", id, mode, round, bad);
  return bad ? 1 : 0;
}
