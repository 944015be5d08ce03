/* Test harness (not shipped): just enough of the HIP kernel language to run a kernel's SOURCE on the CPU.
 *
 *   g++ -std=c++17 -O1 -pthread -DFK_HOST_EMU -shared -fPIC -I fastk_amd/csrc -I tests/csrc tests/csrc/recut_emu.cpp
 *
 * fastk_amd/csrc/fk_common.h includes this file instead of <hip/hip_runtime.h> when FK_HOST_EMU is defined; the .hip
 * files keep their host halves (the ones that talk to the HIP runtime) behind #ifndef FK_HOST_EMU.  A launch runs the
 * workgroups ONE AFTER THE OTHER, every work-item of a workgroup as a host thread: __syncthreads() is a barrier over
 * the workgroup's threads, the wave barrier and __shfl_up one over the 64 threads of a wave, `__shared__` variables are
 * function-local statics (one workgroup at a time, so one copy is the workgroup's), dynamic LDS is a buffer the launch
 * hands out, atomics are the compiler's.  What it can show: index arithmetic, record layouts, barriers in the right
 * places, loops that end -- on the inputs the CPU tests give it.  What it cannot: anything about speed, about memory
 * ordering between waves beyond barriers, or about the real compiler.  Round 6 (the GPU pool closed after two boxes were
 * lost to the test suite): the kernels of the k-mer stage by references run through this against the oracle
 * (tests/test_recut_emu.py); on the MI355X the same stage had been checked against the oracle before (profiles/r06_b_*). */
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#define __HIPCC__ 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static

struct dim3
{ unsigned x, y, z;
  dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
struct uint4 { unsigned x, y, z, w; };
struct uint2 { unsigned x, y; };
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { uint4 r = { a, b, c, d }; return r; }
static inline uint2 make_uint2(unsigned a, unsigned b) { uint2 r = { a, b }; return r; }

typedef void *hipStream_t;
typedef void *hipEvent_t;
typedef int   hipError_t;
#define hipSuccess 0
#define hipErrorOutOfMemory 2
static inline const char *hipGetErrorString(hipError_t) { return "emulated"; }

/* ---- the work-item's coordinates and its workgroup ---------------------------------------------------------------- */
struct emu_barrier
{ std::mutex m;
  std::condition_variable cv;
  unsigned n = 0, waiting = 0, phase = 0;
  void init(unsigned count) { n = count; waiting = 0; phase = 0; }
  void wait()
  { std::unique_lock<std::mutex> lk(m);
    const unsigned ph = phase;
    if (++waiting == n)
      { waiting = 0; phase += 1; cv.notify_all(); }
    else
      cv.wait(lk, [&] { return phase != ph; });
  }
};

struct emu_group
{ emu_barrier all;
  emu_barrier wave[16];
  unsigned long long xchg[16][64];     /* __shfl_up: a wave's exchange buffer */
  void *dyn_lds;
};

extern thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
extern thread_local emu_group *emu_g;
#ifdef FK_EMU_DEFINE
thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
thread_local emu_group *emu_g = nullptr;
#endif

static inline void __syncthreads() { emu_g->all.wait(); }
static inline void emu_wave_barrier() { emu_g->wave[threadIdx.x >> 6].wait(); }
#define __builtin_amdgcn_wave_barrier emu_wave_barrier
#define FK_DYN_LDS(type, name) type *name = (type *) emu_g->dyn_lds
#define FK_DYN_LDS_ALIGNED(type, name, al) type *name = (type *) emu_g->dyn_lds

template <typename T>
static inline T __shfl_up(T x, unsigned o, int width = 64)
{ static_assert(sizeof(T) <= 8, "shuffles of up to 64 bits");
  const unsigned w = threadIdx.x >> 6, l = threadIdx.x & 63u;
  unsigned long long v = 0;
  memcpy(&v, &x, sizeof(T));
  emu_g->xchg[w][l] = v;
  emu_g->wave[w].wait();
  T y = x;
  if ((l & (unsigned) (width - 1)) >= o)               /* (inside segments of `width` lanes) */
    { v = emu_g->xchg[w][l - o]; memcpy(&y, &v, sizeof(T)); }
  emu_g->wave[w].wait();
  return (y);
}

/* run `body` once per work-item: grid x block threads, one workgroup at a time */
static inline void emu_launch(unsigned grid, unsigned block, size_t dyn_lds_bytes, const std::function<void()> &body)
{ emu_group g;
  std::vector<unsigned char> lds(dyn_lds_bytes + 64);
  g.dyn_lds = lds.data();
  for (unsigned b = 0; b < grid; b++)
    { g.all.init(block);
      for (unsigned w = 0; w < (block + 63) / 64; w++)
        g.wave[w].init(std::min(64u, block - 64 * w));
      std::vector<std::thread> th;
      for (unsigned t = 0; t < block; t++)
        th.emplace_back([&, t, b]
          { threadIdx = dim3(t); blockIdx = dim3(b); blockDim = dim3(block); gridDim = dim3(grid);
            emu_g = &g;
            body();
          });
      for (auto &x : th) x.join();
    }
}

/* ---- atomics, intrinsics --------------------------------------------------------------------------------------------- */
static inline unsigned atomicAdd(unsigned *p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
static inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
static inline unsigned long long atomicMax(unsigned long long *p, unsigned long long v)
{ unsigned long long o = __atomic_load_n(p, __ATOMIC_SEQ_CST);
  while (o < v && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) { }
  return (o);
}
static inline unsigned __funnelshift_l(unsigned lo, unsigned hi, unsigned s)
{ s &= 31u; return (unsigned) (((((unsigned long long) hi) << 32) | lo) << s >> 32); }
static inline unsigned __funnelshift_r(unsigned lo, unsigned hi, unsigned s)
{ s &= 31u; return (unsigned) ((((((unsigned long long) hi) << 32) | lo) >> s) & 0xffffffffull); }
static inline unsigned __builtin_amdgcn_perm(unsigned s0, unsigned s1, unsigned sel)     /* v_perm_b32: bytes of {s0, s1} */
{ const unsigned long long src = (((unsigned long long) s0) << 32) | s1;
  unsigned r = 0;
  for (int i = 0; i < 4; i++)
    { const unsigned c = (sel >> (8 * i)) & 0xffu;
      const unsigned b = (c < 8) ? (unsigned) ((src >> (8 * c)) & 0xffu) : (c == 12 ? 0u : (c > 12 ? 0xffu : 0u));
      r |= b << (8 * i);
    }
  return (r);
}
static inline unsigned __umul24(unsigned a, unsigned b) { return ((a & 0xffffffu) * (b & 0xffffffu)); }
static inline unsigned emu_brev32(unsigned x)
{ x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
  x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
  x = ((x >> 4) & 0x0f0f0f0fu) | ((x & 0x0f0f0f0fu) << 4);
  return (__builtin_bswap32(x));
}
#define __builtin_bitreverse32 emu_brev32
static inline unsigned __brev(unsigned x) { return (emu_brev32(x)); }
static inline int __popc(unsigned x) { return (__builtin_popcount(x)); }
static inline int __popcll(unsigned long long x) { return (__builtin_popcountll(x)); }
static inline int __clz(int x) { return (x == 0 ? 32 : __builtin_clz((unsigned) x)); }
static inline int __ffs(int x) { return (__builtin_ffs(x)); }
static inline int __ffsll(long long x) { return (__builtin_ffsll(x)); }
static inline unsigned __umulhi(unsigned a, unsigned b) { return ((unsigned) (((unsigned long long) a * b) >> 32)); }
static inline void __threadfence_block() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline unsigned atomicOr(unsigned *p, unsigned v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
using std::min;
using std::max;
static inline unsigned min(unsigned a, int b) { return (a < (unsigned) b ? a : (unsigned) b); }
