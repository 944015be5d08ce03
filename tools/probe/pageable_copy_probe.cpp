// Are asynchronous copies between device memory and PAGEABLE host memory complete when hipStreamSynchronize returns --
// also while dozens of processes share the GPU?    hipcc -O2 -o pageable_copy_probe pageable_copy_probe.cpp
// (tests/fuzz_parity.py in 32 processes side by side: histograms read from a malloc'ed buffer after
//  hipMemcpyAsync(DeviceToHost) + hipStreamSynchronize were sometimes not the device's, and the harness's own download
//  of the read buffer came back with a tail of something else -- round 5.)
//   pageable_copy_probe <seconds> <id>      D2H into malloc'ed memory of changing size and alignment, with the pattern of
//                                           the iteration; H2D from malloc'ed memory that is overwritten right after the
//                                           call returns; unaligned device offsets; a kernel reads what H2D brought
// Prints one line: iterations, and how many D2H / H2D results were wrong (first few described).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <time.h>
static double wall() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

__global__ void k_fill(unsigned char *p, size_t n, unsigned seed)
{ for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x)
    p[i] = (unsigned char) ((i * 2654435761u + seed) >> 13) | 1u;              // never 0
}

__global__ void k_check(const unsigned char *p, size_t n, unsigned seed, unsigned long long *bad, unsigned long long *zeros)
{ unsigned long long b = 0, z = 0;
  for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x)
    { const unsigned char want = (unsigned char) ((i * 2654435761u + seed) >> 13) | 1u;
      if (p[i] != want) b += 1;
      if (p[i] == 0) z += 1;
    }
  if (b) atomicAdd(bad, b);
  if (z) atomicAdd(zeros, z);
}

int main(int argc, char **argv)
{ const double secs = argc > 1 ? atof(argv[1]) : 20.;
  const int id = argc > 2 ? atoi(argv[2]) : 0;
  const size_t cap = (size_t) 8 << 20;
  unsigned char *d; unsigned long long *d_cnt, *h_cnt;
  hipMalloc((void **) &d, cap + 4096); hipMalloc((void **) &d_cnt, 16); hipHostMalloc((void **) &h_cnt, 16, hipHostMallocDefault);
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  unsigned rng = 12345u + 977u * (unsigned) id;
  long it = 0, bad_d2h = 0, bad_h2d = 0, shown = 0;
  const double t0 = wall();
  while (wall() - t0 < secs)
    { rng = rng * 1664525u + 1013904223u; const size_t n = 1000 + (rng >> 8) % (cap - 1000);
      rng = rng * 1664525u + 1013904223u; const size_t doff = (rng >> 8) % 64;         // unaligned device offset
      rng = rng * 1664525u + 1013904223u; const size_t hoff = (rng >> 8) % 64;
      const unsigned seed = (unsigned) it * 7919u + (unsigned) id;
      // ---- D2H into pageable memory
      hipLaunchKernelGGL(k_fill, dim3(512), dim3(256), 0, s, d + doff, n, seed);
      unsigned char *h = (unsigned char *) malloc(n + 128);
      memset(h, 0, n + 128);
      hipMemcpyAsync(h + hoff, d + doff, n, hipMemcpyDeviceToHost, s);
      hipStreamSynchronize(s);
      size_t wrong = 0, first = 0, zeros = 0;
      for (size_t i = 0; i < n; i++)
        { const unsigned char want = (unsigned char) ((i * 2654435761u + seed) >> 13) | 1u;
          if (h[hoff + i] != want) { if (!wrong) first = i; wrong += 1; if (h[hoff + i] == 0) zeros += 1; }
        }
      if (wrong)
        { bad_d2h += 1;
          if (shown++ < 6) printf("id %d it %ld: D2H of %zu bytes: %zu wrong (%zu still 0), first at %zu\n", id, it, n, wrong, zeros, first);
        }
      // ---- H2D from pageable memory that is reused the moment the call returns
      for (size_t i = 0; i < n; i++) h[hoff + i] = (unsigned char) ((i * 2654435761u + (seed ^ 0x5555u)) >> 13) | 1u;
      hipMemsetAsync(d_cnt, 0, 16, s);
      hipMemcpyAsync(d + doff, h + hoff, n, hipMemcpyHostToDevice, s);
      memset(h, 0, n + 128);                                   // the caller's buffer is the caller's again
      free(h);
      hipLaunchKernelGGL(k_check, dim3(512), dim3(256), 0, s, d + doff, n, seed ^ 0x5555u, d_cnt, d_cnt + 1);
      hipMemcpyAsync(h_cnt, d_cnt, 16, hipMemcpyDeviceToHost, s);
      hipStreamSynchronize(s);
      if (h_cnt[0])
        { bad_h2d += 1;
          if (shown++ < 6) printf("id %d it %ld: H2D of %zu bytes: %llu wrong on the device (%llu are 0)\n", id, it, n, h_cnt[0], h_cnt[1]);
        }
      it += 1;
    }
  printf("id %d: %ld iterations, %ld D2H into pageable memory wrong after hipStreamSynchronize, %ld H2D from pageable memory wrong\n",
         id, it, bad_d2h, bad_h2d);
  return (bad_d2h + bad_h2d) ? 1 : 0;
}
