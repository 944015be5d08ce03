/* Test harness (not shipped): the HIP kernel language -- and, with -DFK_EMU_FULL, the HIP runtime -- on the CPU.
 *
 * fastk_amd/csrc/fk_common.h includes this file instead of <hip/hip_runtime.h> when FK_HOST_EMU is defined.  Two uses:
 *   * a kernel's SOURCE run by a small driver (tests/csrc/recut_emu.cpp, xs_emu.cpp, split_emu.cpp: -DFK_HOST_EMU; the
 *     .hip files keep the host halves that talk to the runtime behind #if !defined(FK_HOST_EMU) || defined(FK_EMU_FULL));
 *   * the WHOLE library -- every .hip file, kernels and host code, the same C-ABI -- as tests/csrc/libfastk_emu.so
 *     (tests/csrc/build_emu_lib.py: -DFK_HOST_EMU -DFK_EMU_FULL), so that the parity tests that need only a small input can
 *     run without a GPU (FASTK_AMD_EMU=1, tests/conftest.py, tests/test_emu_suite.py).
 *
 * Kernel language.  A launch runs the workgroups ONE AFTER THE OTHER, every work-item of a workgroup as a fiber of the
 * calling thread (ucontext): __syncthreads() parks a work-item until all of the workgroup's that are still alive have
 * arrived, the wave barrier and the wave collectives (__shfl*, __ballot, readlane / readfirstlane, the DPP controls the
 * kernels use) until the 64 of its wave have; `__shared__` variables are function-local statics (one workgroup at a time,
 * so one copy is the workgroup's), dynamic LDS is a buffer the launch hands out (FK_DYN_LDS), atomics are plain
 * read-modify-writes.  One work-item at a time and in a fixed order: a run is deterministic, and code that counts on the
 * lanes of a wave moving in lockstep BETWEEN two collectives fails the same way every time -- the three places of the
 * library that do are marked in the sources with macros that cost the device nothing (fk_common.h): FK_EMU_WAVE_SYNC()
 * (k_ag_count2's wave-local elections: a wave's LDS stores are all done before its loads), FK_WAVE_UNIFORM(x) (k_dd_table's
 * overflow flag: one LDS read for the whole wave), FK_BALLOT_ACTIVE(p) (a ballot among the lanes the exec mask keeps in a
 * divergent loop); k_dd_table's four inline-asm LDS accesses have plain C++ twins under FK_HOST_EMU.  The device code is
 * unchanged by all of this to the last instruction (hipcc -S --cuda-device-only before / after).
 *
 * Runtime (-DFK_EMU_FULL).  Device memory is host memory, streams and events do nothing, a launch runs the kernel to its
 * end, one device "gfx950 (tests/csrc/hip_emu.h)" with 8 compute units.
 *
 * What it can show: index arithmetic, record layouts, barriers in the right places, loops that end, host orchestration --
 * on the inputs the tests give it.  What it cannot: anything about speed, about memory ordering on a real chip, about
 * the real compiler, the real runtime (streams, events, pinned memory) or RCCL. */
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <time.h>
#include <functional>
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

/* ---- the work-items of a workgroup: fibers of ONE host thread ------------------------------------------------------------
   (The first version ran every work-item as a host thread with condition-variable barriers: right, and 50 x slower -- a
   barrier over 256 threads is 256 trips through the kernel's scheduler.)  A work-item runs until it reaches a barrier or
   ends; __syncthreads / the wave barrier park it until every work-item of the workgroup / wave that is still alive has
   arrived.  One at a time and in a fixed order: a run is deterministic, atomics need no hardware, and code that silently
   relies on the lanes of a wave moving in lockstep between two barriers fails the same way every time. */
#include <ucontext.h>
#include <pthread.h>
#include <sys/mman.h>

struct emu_fiber
{ ucontext_t ctx;
  void      *stack;
  unsigned   tid;
  int        state;                    /* 0 runnable, 1 at the workgroup's barrier, 2 at its wave's barrier, 3 done */
};

struct emu_group
{ unsigned nthreads, alive, at_all;
  unsigned wave_alive[16], at_wave[16];
  unsigned long long xchg[16][64];     /* __shfl_up: a wave's exchange buffer */
  void *dyn_lds;
  emu_fiber *fib;
  ucontext_t sched;
  emu_fiber *cur;
  const std::function<void()> *body;
  unsigned block, grid, bid;
};

extern thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
extern thread_local emu_group *emu_g;
extern thread_local const char *emu_kernel;            /* the kernel a launch runs, for the messages */
#ifdef FK_EMU_DEFINE
thread_local const char *emu_kernel = "?";
thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
thread_local emu_group *emu_g = nullptr;
#endif

static inline void emu_yield()
{ emu_group *g = emu_g;
  emu_fiber *f = g->cur;
  swapcontext(&f->ctx, &g->sched);
  threadIdx = dim3(f->tid);            /* (back in this work-item) */
}

static inline void emu_release_all(emu_group *g)
{ for (unsigned t = 0; t < g->nthreads; t++)
    if (g->fib[t].state == 1) g->fib[t].state = 0;
  g->at_all = 0;
}

static inline void emu_release_wave(emu_group *g, unsigned w)
{ const unsigned hi = std::min(g->nthreads, 64 * (w + 1));
  for (unsigned t = 64 * w; t < hi; t++)
    if (g->fib[t].state == 2) g->fib[t].state = 0;
  g->at_wave[w] = 0;
}

static inline void __syncthreads()
{ emu_group *g = emu_g;
  g->cur->state = 1;
  if (++g->at_all == g->alive)
    emu_release_all(g);
  emu_yield();
}

static inline int __syncthreads_count(int pred)      /* the workgroup's barrier + how many work-items brought a true */
{ static unsigned acc[2];
  static unsigned gen;
  emu_group *g = emu_g;
  if (g->at_all == 0) acc[gen & 1] = 0;              /* the first to arrive */
  const unsigned my = gen & 1;
  acc[my] += pred ? 1u : 0u;
  if (g->at_all + 1 == g->alive) gen += 1;           /* the last one turns the page before anybody goes on */
  __syncthreads();
  return ((int) acc[my]);
}

static inline void emu_wave_barrier()
{ emu_group *g = emu_g;
  const unsigned w = g->cur->tid >> 6;
  g->cur->state = 2;
  if (++g->at_wave[w] == g->wave_alive[w])
    emu_release_wave(g, w);
  emu_yield();
}
#define __builtin_amdgcn_wave_barrier emu_wave_barrier
#define FK_DYN_LDS(type, name) type *name = (type *) emu_g->dyn_lds
#define FK_DYN_LDS_ALIGNED(type, name, al) type *name = (type *) emu_g->dyn_lds
#define FK_OPAQUE(x) ((void) (x))
#define FK_EMU_WAVE_SYNC() emu_wave_barrier()
#define FK_WAVE_UNIFORM(x) ((unsigned) __builtin_amdgcn_readfirstlane((int) (x)))
/* a ballot among the lanes the hardware's exec mask keeps in a divergent loop: here every lane sees itself alone (the one
   use sums the set bits into a workgroup counter through the lowest set lane: the same total) */
#define FK_BALLOT_ACTIVE(p) ((p) ? (1ull << (threadIdx.x & 63u)) : 0ull)
#define FK_KEEP(x)   ((void) (x))

template <typename T>
static inline T __shfl_up(T x, unsigned o, int width = 64)
{ static_assert(sizeof(T) <= 8, "shuffles of up to 64 bits");
  const unsigned w = threadIdx.x >> 6, l = threadIdx.x & 63u;
  unsigned long long v = 0;
  memcpy(&v, &x, sizeof(T));
  emu_g->xchg[w][l] = v;
  emu_wave_barrier();
  T y = x;
  if ((l & (unsigned) (width - 1)) >= o)               /* (inside segments of `width` lanes) */
    { v = emu_g->xchg[w][l - o]; memcpy(&y, &v, sizeof(T)); }
  emu_wave_barrier();
  return (y);
}

static void emu_fiber_main()
{ emu_group *g = emu_g;
  emu_fiber *f = g->cur;
  threadIdx = dim3(f->tid);
  (*g->body)();
  /* the work-item ends: it no longer counts at any barrier */
  g = emu_g;
  f = g->cur;
  f->state = 3;
  g->alive -= 1;
  const unsigned w = f->tid >> 6;
  g->wave_alive[w] -= 1;
  if (g->alive > 0 && g->at_all == g->alive) emu_release_all(g);
  if (g->wave_alive[w] > 0 && g->at_wave[w] == g->wave_alive[w]) emu_release_wave(g, w);
  swapcontext(&f->ctx, &g->sched);
}

#define EMU_STACK ((size_t) 256 << 10)

/* run `body` once per work-item: grid x block work-items, one workgroup at a time */
static inline void emu_launch(unsigned grid, unsigned block, size_t dyn_lds_bytes, const std::function<void()> &body)
{ emu_group g;
  std::vector<unsigned char> lds(dyn_lds_bytes + 64);
  std::vector<emu_fiber> fib(block);
  char *stacks = (char *) mmap(NULL, EMU_STACK * block, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
  if (stacks == (char *) MAP_FAILED) { fprintf(stderr, "hip_emu: no memory for %u stacks\n", block); abort(); }
  g.dyn_lds = lds.data();
  g.fib = fib.data();
  g.body = &body;
  g.block = block; g.grid = grid; g.nthreads = block;
  emu_group *outer = emu_g;
  emu_g = &g;
  blockDim = dim3(block); gridDim = dim3(grid);
  for (unsigned b = 0; b < grid; b++)
    { g.bid = b;
      blockIdx = dim3(b);
      g.alive = block; g.at_all = 0;
      for (unsigned w = 0; w < 16; w++)
        { g.wave_alive[w] = (64 * w < block) ? std::min(64u, block - 64 * w) : 0; g.at_wave[w] = 0; }
      for (unsigned t = 0; t < block; t++)
        { emu_fiber &f = fib[t];
          f.tid = t; f.state = 0; f.stack = stacks + EMU_STACK * t;
          getcontext(&f.ctx);
          f.ctx.uc_stack.ss_sp = f.stack; f.ctx.uc_stack.ss_size = EMU_STACK; f.ctx.uc_link = NULL;
          makecontext(&f.ctx, emu_fiber_main, 0);
        }
      unsigned done = 0;
      while (done < block)
        { bool ran = false;
          for (unsigned t = 0; t < block; t++)
            if (fib[t].state == 0)
              { g.cur = &fib[t];
                swapcontext(&g.sched, &fib[t].ctx);
                ran = true;
                if (fib[t].state == 3) done += 1;
              }
          if (!ran && done < block)
            { fprintf(stderr, "hip_emu: %s: workgroup %u is stuck: %u of %u work-items wait at a barrier that the others never reach "
                              "(%u at the workgroup's; at their wave's:", emu_kernel, b, block - done, block, g.at_all);
              for (unsigned w = 0; w < 16; w++) if (g.at_wave[w]) fprintf(stderr, " wave %u: %u of %u", w, g.at_wave[w], g.wave_alive[w]);
              fprintf(stderr, ")\n");
              abort();
            }
        }
    }
  emu_g = outer;
  munmap(stacks, EMU_STACK * block);
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

/* ---- wave collectives (every lane of the wave takes part: the kernels call them in wave-uniform control flow) ------------ */
template <typename T>
static inline void emu_wave_gather(T x, T (&all)[64])
{ static_assert(sizeof(T) <= 8, "up to 64 bits");
  const unsigned w = threadIdx.x >> 6, l = threadIdx.x & 63u;
  unsigned long long v = 0;
  memcpy(&v, &x, sizeof(T));
  emu_g->xchg[w][l] = v;
  emu_wave_barrier();
  const unsigned n = emu_g->nthreads - 64 * w < 64 ? emu_g->nthreads - 64 * w : 64;
  for (unsigned i = 0; i < 64; i++)
    { unsigned long long u = (i < n) ? emu_g->xchg[w][i] : 0ull; memcpy(&all[i], &u, sizeof(T)); }
  emu_wave_barrier();
}
static inline unsigned long long __ballot(int pred)
{ unsigned all[64];
  emu_wave_gather<unsigned>(pred ? 1u : 0u, all);
  unsigned long long m = 0;
  for (int i = 0; i < 64; i++) if (all[i]) m |= 1ull << i;
  return (m);
}
template <typename T> static inline T __shfl(T x, int src, int width = 64)
{ T all[64]; emu_wave_gather<T>(x, all);
  const unsigned l = threadIdx.x & 63u;
  return (all[(l & ~(unsigned) (width - 1)) | ((unsigned) src & (unsigned) (width - 1))]);
}
template <typename T> static inline T __shfl_down(T x, unsigned o, int width = 64)
{ T all[64]; emu_wave_gather<T>(x, all);
  const unsigned l = threadIdx.x & 63u;
  return (((l & (unsigned) (width - 1)) + o < (unsigned) width) ? all[l + o] : x);
}
template <typename T> static inline T __shfl_xor(T x, int m, int width = 64)
{ T all[64]; emu_wave_gather<T>(x, all);
  (void) width;
  return (all[(threadIdx.x & 63u) ^ (unsigned) m]);
}
static inline int __builtin_amdgcn_readlane(int x, int l) { int all[64]; emu_wave_gather<int>(x, all); return (all[l & 63]); }
static inline int __builtin_amdgcn_readfirstlane(int x) { int all[64]; emu_wave_gather<int>(x, all); return (all[0]); }
/* v_mov_dpp with the controls the kernels use: row_shr:1..15 (0x111..0x11f), row_bcast:15 (0x142), row_bcast:31 (0x143);
   a lane whose source lies outside its row (or whose row is masked out) keeps `old` */
static inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl)
{ int all[64]; emu_wave_gather<int>(src, all);
  (void) bank_mask; (void) bound_ctrl;
  const unsigned l = threadIdx.x & 63u, row = l >> 4;
  if (!((row_mask >> row) & 1)) return (old);
  if (ctrl >= 0x111 && ctrl <= 0x11f)
    { const unsigned s = (unsigned) ctrl - 0x110;
      return (((l & 15u) >= s) ? all[l - s] : old);
    }
  if (ctrl == 0x142) return ((row & 1u) ? all[(row << 4) - 1] : old);        /* lane 15 of the row in front -> rows 1, 3 */
  if (ctrl == 0x143) return ((row >= 2) ? all[31] : old);                     /* lane 31 -> rows 2, 3 */
  fprintf(stderr, "hip_emu: DPP control 0x%x is not emulated\n", ctrl);
  abort();
}
static inline int atomicMin(int *p, int v) { int o = *p; if (v < o) *p = v; return (o); }
static inline int atomicMax(int *p, int v) { int o = *p; if (v > o) *p = v; return (o); }
static inline int atomicAdd(int *p, int v) { int o = *p; *p = o + v; return (o); }
static inline unsigned atomicMin(unsigned *p, unsigned v) { unsigned o = *p; if (v < o) *p = v; return (o); }
static inline unsigned atomicMax(unsigned *p, unsigned v) { unsigned o = *p; if (v > o) *p = v; return (o); }
static inline unsigned atomicCAS(unsigned *p, unsigned cmp, unsigned v) { unsigned o = *p; if (o == cmp) *p = v; return (o); }
static inline unsigned long long atomicCAS(unsigned long long *p, unsigned long long cmp, unsigned long long v)
{ unsigned long long o = *p; if (o == cmp) *p = v; return (o); }

/* ---- the runtime, for the host halves of the .hip files (-DFK_EMU_FULL): device memory is host memory, streams and events
        do nothing, a launch runs the kernel to its end ------------------------------------------------------------------- */
#ifdef FK_EMU_FULL
#define hipErrorUnknown 999
#define hipErrorNotReady 600
enum { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyHostToHost = 0, hipMemcpyDefault = 4 };
#define hipStreamNonBlocking 1
#define hipEventDisableTiming 2
#define hipHostMallocDefault 0
#define hipHostRegisterDefault 0
#define hipFuncAttributeMaxDynamicSharedMemorySize 8
enum { hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2 };
struct hipPointerAttribute_t { int type; };
struct hipDeviceProp_t { char gcnArchName[64]; int multiProcessorCount; size_t totalGlobalMem; };
static inline hipError_t hipMalloc(void **p, size_t n) { *p = aligned_alloc(256, (n + 255 + 256) & ~(size_t) 255); return (*p ? 0 : 2); }
static inline hipError_t hipFree(void *p) { free(p); return (0); }
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { return (hipMalloc(p, n)); }
static inline hipError_t hipHostFree(void *p) { free(p); return (0); }
static inline hipError_t hipHostRegister(void *, size_t, unsigned) { return (0); }
static inline hipError_t hipHostUnregister(void *) { return (0); }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, int) { memmove(d, s, n); return (0); }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, int, hipStream_t) { memmove(d, s, n); return (0); }
static inline hipError_t hipMemcpy2DAsync(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h, int, hipStream_t)
{ for (size_t r = 0; r < h; r++) memmove((char *) d + r * dp, (const char *) s + r * sp, w); return (0); }
static inline hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return (0); }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return (0); }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = malloc(8); return (0); }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return (0); }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return (0); }
static inline hipError_t hipDeviceSynchronize() { return (0); }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = malloc(8); return (0); }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = malloc(8); return (0); }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return (0); }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return (0); }
static inline hipError_t hipEventQuery(hipEvent_t) { return (0); }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.001f; return (0); }
static inline hipError_t hipGetLastError() { return (0); }
static inline hipError_t hipSetDevice(int) { return (0); }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return (0); }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int)
{ memset(p, 0, sizeof(*p)); strcpy(p->gcnArchName, "gfx950 (tests/csrc/hip_emu.h)"); p->multiProcessorCount = 8; p->totalGlobalMem = (size_t) 8 << 30; return (0); }
static inline hipError_t hipFuncSetAttribute(const void *, int, int) { return (0); }
static inline hipError_t hipMemGetInfo(size_t *f, size_t *t) { *f = *t = (size_t) 8 << 30; return (0); }
static inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t *a, const void *) { a->type = hipMemoryTypeHost; return (0); }
#define hipLaunchKernelGGL(kern, grid, block, lds, stream, ...) \
  do { const dim3 g_ = (grid), b_ = (block); (void) (stream); emu_kernel = #kern; \
       emu_launch(g_.x, b_.x, (size_t) (lds), [=] { kern(__VA_ARGS__); }); } while (0)
#endif
