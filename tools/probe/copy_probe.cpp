// What a plain copy reaches on this box: hipMemcpy device-to-device against hand-written uint4 copy kernels
// (grid-stride, one-shot, non-temporal), 5.5 GB each way -- the ceiling the radix scatter is priced against.
//   hipcc -O2 --offload-arch=gfx950 -o copy_probe copy_probe.cpp && ./copy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(256) void k_copy_stride(const uint4 *__restrict__ a, uint4 *__restrict__ b, size_t n)
{ for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t) gridDim.x * 256)
    b[i] = a[i];
}

template <int U>
__global__ __launch_bounds__(256) void k_copy_tile(const uint4 *__restrict__ a, uint4 *__restrict__ b, size_t n)
{ const size_t base = (size_t) blockIdx.x * 256 * U + threadIdx.x;
  uint4 v[U];
#pragma unroll
  for (int u = 0; u < U; u++)
    if (base + (size_t) u * 256 < n) v[u] = a[base + (size_t) u * 256];
#pragma unroll
  for (int u = 0; u < U; u++)
    if (base + (size_t) u * 256 < n) b[base + (size_t) u * 256] = v[u];
}

template <int U>
__global__ __launch_bounds__(256) void k_copy_tile_nt(const uint4 *__restrict__ a, uint4 *__restrict__ b, size_t n)
{ const size_t base = (size_t) blockIdx.x * 256 * U + threadIdx.x;
  uint4 v[U];
#pragma unroll
  for (int u = 0; u < U; u++)
    if (base + (size_t) u * 256 < n)
      { const uint4 *p = a + base + (size_t) u * 256;
        v[u].x = __builtin_nontemporal_load(&p->x); v[u].y = __builtin_nontemporal_load(&p->y);
        v[u].z = __builtin_nontemporal_load(&p->z); v[u].w = __builtin_nontemporal_load(&p->w);
      }
#pragma unroll
  for (int u = 0; u < U; u++)
    if (base + (size_t) u * 256 < n)
      { uint4 *p = b + base + (size_t) u * 256;
        __builtin_nontemporal_store(v[u].x, &p->x); __builtin_nontemporal_store(v[u].y, &p->y);
        __builtin_nontemporal_store(v[u].z, &p->z); __builtin_nontemporal_store(v[u].w, &p->w);
      }
}

__global__ __launch_bounds__(256) void k_fill_tile(uint4 *__restrict__ b, size_t n, unsigned x)
{ const size_t base = (size_t) blockIdx.x * 1024 + threadIdx.x;
  const uint4 v = make_uint4(x, x + 1, x + 2, x + 3);
#pragma unroll
  for (int u = 0; u < 4; u++)
    if (base + (size_t) u * 256 < n) b[base + (size_t) u * 256] = v;
}

__global__ __launch_bounds__(256) void k_read_tile(const uint4 *__restrict__ a, size_t n, unsigned *sink)
{ const size_t base = (size_t) blockIdx.x * 1024 + threadIdx.x;
  unsigned acc = 0;
#pragma unroll
  for (int u = 0; u < 4; u++)
    if (base + (size_t) u * 256 < n) { const uint4 v = a[base + (size_t) u * 256]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) *sink = acc;
}

template <typename F> static void timeit(const char *what, size_t bytes, F f)
{ hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  f(); f();
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < 5; i++) f();
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s %8.3f ms  %7.1f GB/s (read + write)\n", what, ms / 5, 2.0 * bytes / (ms / 5 * 1e-3) / 1e9);
}

int main()
{ const size_t bytes = 5551085160ull / 16 * 16;      // one bucket's weighted k-mers at configs[2]
  const size_t n = bytes / 16;
  uint4 *a, *b;
  hipMalloc((void **) &a, bytes); hipMalloc((void **) &b, bytes);
  hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
  timeit("hipMemcpyAsync device to device", bytes, [&]() { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); });
  for (int g : { 1024, 2048, 4096, 16384 })
    { char w[64]; snprintf(w, sizeof w, "grid-stride uint4, %d workgroups", g);
      timeit(w, bytes, [&]() { hipLaunchKernelGGL(k_copy_stride, dim3(g), dim3(256), 0, 0, a, b, n); });
    }
  timeit("one tile per workgroup, 4 x uint4 per thread", bytes, [&]() { hipLaunchKernelGGL(k_copy_tile<4>, dim3((unsigned) ((n + 1023) / 1024)), dim3(256), 0, 0, a, b, n); });
  timeit("one tile per workgroup, 8 x uint4 per thread", bytes, [&]() { hipLaunchKernelGGL(k_copy_tile<8>, dim3((unsigned) ((n + 2047) / 2048)), dim3(256), 0, 0, a, b, n); });
  timeit("the same, non-temporal, 4 x", bytes, [&]() { hipLaunchKernelGGL(k_copy_tile_nt<4>, dim3((unsigned) ((n + 1023) / 1024)), dim3(256), 0, 0, a, b, n); });
  timeit("the same, non-temporal, 8 x", bytes, [&]() { hipLaunchKernelGGL(k_copy_tile_nt<8>, dim3((unsigned) ((n + 2047) / 2048)), dim3(256), 0, 0, a, b, n); });
  // one direction only (the rate printed counts 2 x bytes: halve it)
  timeit("write only (fill), x2 convention", bytes, [&]() { hipLaunchKernelGGL(k_fill_tile, dim3((unsigned) ((n + 1023) / 1024)), dim3(256), 0, 0, b, n, 7u); });
  unsigned *sink; hipMalloc((void **) &sink, 4);
  timeit("read only, x2 convention", bytes, [&]() { hipLaunchKernelGGL(k_read_tile, dim3((unsigned) ((n + 1023) / 1024)), dim3(256), 0, 0, a, n, sink); });
  return 0;
}
