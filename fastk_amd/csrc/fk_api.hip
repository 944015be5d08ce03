// fk_api.hip -- C-ABI of libfastk_amd.so (see include/fastk_amd.h): context, streaming
// interface, whole-path driver, output encodings.
#include "fk_common.h"
#include "../../include/fk_synth.h"

#include <pthread.h>
#include <math.h>
#include <stdarg.h>
#include <fcntl.h>
#include <unistd.h>
#include <sys/mman.h>
#include <algorithm>
#include <vector>
#include <thread>
#include <chrono>

static char g_last_error[512] = "";

void fk_set_error(fk_ctx *ctx, const char *fmt, ...)
{ va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
  va_end(ap);
  if (ctx != NULL)
    memcpy(ctx->err, g_last_error, sizeof(ctx->err));
}

void *fk_slot(fk_ctx *ctx, int slot, int64_t nbytes)
{ if (nbytes < 16) nbytes = 16;
  if (ctx->slot_cap[slot] >= nbytes)
    return (ctx->slot_ptr[slot]);
  if (ctx->slot_ptr[slot] != NULL)
    { hipFree(ctx->slot_ptr[slot]);
      ctx->slot_ptr[slot] = NULL;
      ctx->slot_cap[slot] = 0;
    }
  // a little headroom, so that run-to-run jitter of data-dependent sizes (hash collisions in the
  // super-mer grouping change W by a few records) does not re-allocate multi-GB buffers
  nbytes += nbytes / 32 + (1 << 20);
  void *p = NULL;
  const double w0 = fk_wall();
  hipError_t me = hipMalloc(&p, (size_t) nbytes);
  if (nbytes > (1ll << 30) && getenv("FK_FINISH_TIMING") != NULL)
    fprintf(stderr, "  finish timing: slot %d: hipMalloc of %.1f GB took %.3f s\n", slot, (double) nbytes / 1e9, fk_wall() - w0);
  if (me != hipSuccess && slot != FK_SLOT_SM_DIG && ctx->slot_ptr[FK_SLOT_SM_DIG] != NULL)
    { // The splitter's digit stream (a byte per super-mer record: 3 GB and more) only saves the grouping sort a pass
      // over the records: when memory runs short it goes first.  ctx->dig_lost tells the pipeline that the streams of
      // the buckets still to come are gone (fk_radix.hip and count_bucket look at it before they use one).
      (void) hipGetLastError();
      hipFree(ctx->slot_ptr[FK_SLOT_SM_DIG]);
      ctx->slot_ptr[FK_SLOT_SM_DIG] = NULL;
      ctx->slot_cap[FK_SLOT_SM_DIG] = 0;
      ctx->dig_lost = true;
      me = hipMalloc(&p, (size_t) nbytes);
    }
  if (me != hipSuccess)
    { size_t fr = 0, tot = 0;
      int64_t held = 0;
      (void) hipGetLastError();                 // (the failure is reported here: it must not surface again at the next launch check)
      char    big[256] = "";
      hipMemGetInfo(&fr, &tot);
      for (int i = 0; i < FK_NSLOTS; i++)
        { held += ctx->slot_cap[i];
          if (ctx->slot_cap[i] >= (1ll << 30) && strlen(big) < 220)
            snprintf(big + strlen(big), sizeof(big) - strlen(big), " %d:%.1f", i, (double) ctx->slot_cap[i] / 1e9);
        }
      fk_set_error(ctx, "out of HBM: cannot allocate %lld bytes (arena slot %d); %.1f GB free of %.1f, arena holds "
                        "%.1f GB, chunks %.1f GB; slots over 1 GB [slot:GB]%s", (long long) nbytes, slot,
                   (double) fr / 1e9, (double) tot / 1e9, (double) held / 1e9, (double) ctx->chunk_hbm_bytes / 1e9, big);
      return (NULL);
    }
  ctx->slot_ptr[slot] = p;
  ctx->slot_cap[slot] = nbytes;
  return (p);
}

uint8_t *fkx_dig_slot(fk_ctx *ctx, int64_t cap)
{ ctx->dig2_off = 0;
  if (ctx->dig_lost)
    return (NULL);
  const int64_t one = cap + 64;
  uint8_t *p = NULL;
  // The second plane is OFF unless FASTK_AMD_TWO_DIGIT_PLANES=1 asks for it: measured at configs[2] it costs the packed
  // splitter 21 ms per step (a second scattered byte store per record) and buys the first grouping pass nothing -- with
  // the digit carried or hashed the pass takes the same 0.35 ms per 35 M records (FK_SORT_TIMING=1): the pass is bound
  // by its write pattern, not by the hash.  (The ablation that promised 28 % had made the records sorted: DESIGN 11.)
  if (!ctx->dig_one_plane && getenv("FASTK_AMD_TWO_DIGIT_PLANES") != NULL)
    p = (uint8_t *) fk_slot(ctx, FK_SLOT_SM_DIG, 2 * one);
  if (p != NULL)
    { ctx->dig2_off = one;
      return (p);
    }
  ctx->dig_one_plane = true;                       // (no second try: a slot that is freed and allocated again every pass
                                                   //  waits for the driver's wipe of what it gave back)
  ctx->dig2_off = 0;
  p = (uint8_t *) fk_slot(ctx, FK_SLOT_SM_DIG, one);
  if (p == NULL)
    ctx->err[0] = 0;                             // (optional: the grouping sort then makes the stream itself)
  return (p);
}

// The result table's host copy lives in pinned memory (the D2H copy of a 36 GB table runs at the
// PCIe rate only from there); grown with headroom, kept until fk_destroy.
// ---- large pinned host buffers --------------------------------------------------------------------
// hipHostMalloc allocates, zeroes and pins at ~8 GB/s (17 GB: 2.1 s; the 36 GB result table of configs[2]: 5-6 s,
// more than the whole device pipeline).  Ordinary huge-page memory touched by a few threads and then registered
// is ready in 0.13 s per 17 GB and copies at the same 57 GB/s (tools/probe/pin_probe.cpp).  Buffers below
// FK_PIN_FAST come from hipHostMalloc as before.
#define FK_PIN_FAST ((int64_t) 64 << 20)
static pthread_mutex_t         g_pin_lock = PTHREAD_MUTEX_INITIALIZER;
static std::vector<void *>     g_pin_reg;             // buffers that were registered, not hipHostMalloc'ed

int fkx_pinned_alloc(void **out, int64_t bytes)
{ *out = NULL;
  if (bytes < FK_PIN_FAST)
    return (hipHostMalloc(out, (size_t) (bytes > 0 ? bytes : 1), hipHostMallocDefault) == hipSuccess ? FK_OK : FK_ENOMEM);
  const size_t n = ((size_t) bytes + ((size_t) 2 << 20) - 1) & ~(((size_t) 2 << 20) - 1);
  char *p = (char *) aligned_alloc((size_t) 2 << 20, n);
  if (p == NULL)
    return (FK_ENOMEM);
  madvise(p, n, MADV_HUGEPAGE);
  { const int T = 8;
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
      th.emplace_back([=]()
        { const size_t lo = n / T * t, hi = (t == T - 1) ? n : n / T * (t + 1);
          for (size_t i = lo; i < hi; i += 4096)
            p[i] = 0;
        });
    for (auto &x : th)
      x.join();
  }
  if (hipHostRegister(p, n, hipHostRegisterDefault) != hipSuccess)
    { (void) hipGetLastError();
      free(p);
      return (hipHostMalloc(out, (size_t) bytes, hipHostMallocDefault) == hipSuccess ? FK_OK : FK_ENOMEM);
    }
  pthread_mutex_lock(&g_pin_lock);
  g_pin_reg.push_back(p);
  pthread_mutex_unlock(&g_pin_lock);
  *out = p;
  return (FK_OK);
}

int fkx_pinned_free(void *p)
{ if (p == NULL)
    return (FK_OK);
  bool mine = false;
  pthread_mutex_lock(&g_pin_lock);
  for (size_t i = 0; i < g_pin_reg.size(); i++)
    if (g_pin_reg[i] == p)
      { g_pin_reg[i] = g_pin_reg.back();
        g_pin_reg.pop_back();
        mine = true;
        break;
      }
  pthread_mutex_unlock(&g_pin_lock);
  if (!mine)
    return (hipHostFree(p) == hipSuccess ? FK_OK : FK_EHIP);
  if (hipHostUnregister(p) != hipSuccess)
    { // still mapped for the device: a copy engine may yet write here.  The pages are left alone (leaked) rather
      // than handed back to malloc, where the next owner would see the late write.
      (void) hipGetLastError();
      return (FK_EHIP);
    }
  free(p);
  return (FK_OK);
}

static pthread_mutex_t g_stream_lock = PTHREAD_MUTEX_INITIALIZER;
static std::vector<std::pair<int, hipStream_t> > g_stream_pool;      // idle streams, by device

int fkx_stream_get(int device, hipStream_t *s)
{ *s = NULL;
  pthread_mutex_lock(&g_stream_lock);
  for (size_t i = 0; i < g_stream_pool.size(); i++)
    if (g_stream_pool[i].first == device)
      { *s = g_stream_pool[i].second;
        g_stream_pool[i] = g_stream_pool.back();
        g_stream_pool.pop_back();
        break;
      }
  pthread_mutex_unlock(&g_stream_lock);
  if (*s != NULL)
    return (FK_OK);
  return (hipStreamCreateWithFlags(s, hipStreamNonBlocking) == hipSuccess ? FK_OK : FK_EHIP);
}

void fkx_stream_put(int device, hipStream_t s)
{ if (s == NULL)
    return;
  (void) hipStreamSynchronize(s);                    // (what goes into the pool is idle)
  pthread_mutex_lock(&g_stream_lock);
  g_stream_pool.push_back(std::make_pair(device, s));
  pthread_mutex_unlock(&g_stream_lock);
}

struct fk_pooled_event { int device; bool timing; hipEvent_t ev; };
static std::vector<fk_pooled_event> g_event_pool;          // idle events (under g_stream_lock)

int fkx_event_get(int device, bool timing, hipEvent_t *e)
{ *e = NULL;
  pthread_mutex_lock(&g_stream_lock);
  for (size_t i = 0; i < g_event_pool.size(); i++)
    if (g_event_pool[i].device == device && g_event_pool[i].timing == timing)
      { *e = g_event_pool[i].ev;
        g_event_pool[i] = g_event_pool.back();
        g_event_pool.pop_back();
        break;
      }
  pthread_mutex_unlock(&g_stream_lock);
  if (*e != NULL)
    return (FK_OK);
  const hipError_t rc = timing ? hipEventCreate(e) : hipEventCreateWithFlags(e, hipEventDisableTiming);
  if (rc != hipSuccess)
    *e = NULL;
  return (rc == hipSuccess ? FK_OK : FK_EHIP);
}

void fkx_event_put(int device, bool timing, hipEvent_t *e)
{ if (e == NULL || *e == NULL)
    return;
  // What goes into the pool is complete (never recorded counts as complete).  An event that is still pending -- only on
  // an error path: a stream that waits for a peer that has died -- is neither waited for (hipEventDestroy never blocked
  // either, and the caller is on its way out) nor destroyed: its handle is dropped, the object stays alive.
  if (hipEventQuery(*e) != hipSuccess)
    { (void) hipGetLastError();
      *e = NULL;
      return;
    }
  fk_pooled_event p; p.device = device; p.timing = timing; p.ev = *e;
  pthread_mutex_lock(&g_stream_lock);
  g_event_pool.push_back(p);
  pthread_mutex_unlock(&g_stream_lock);
  *e = NULL;
}

int fkx_d2h_pageable(fk_ctx *ctx, hipStream_t s, void *dst, const void *d_src, size_t nbytes)
{ FK_HIP(ctx, hipStreamSynchronize(s));
  if (nbytes > 0)
    FK_HIP(ctx, hipMemcpy(dst, d_src, nbytes, hipMemcpyDeviceToHost));
  return (FK_OK);
}

int fkx_h2d_pageable(fk_ctx *ctx, hipStream_t s, void *d_dst, const void *src, size_t nbytes)
{ FK_HIP(ctx, hipStreamSynchronize(s));            // (what the stream still does with the destination comes first)
  if (nbytes > 0)
    FK_HIP(ctx, hipMemcpy(d_dst, src, nbytes, hipMemcpyHostToDevice));
  return (FK_OK);
}

int fkx_reserve_host_table(fk_ctx *ctx, int64_t bytes)
{ if (ctx->h_table_cap >= bytes)
    return (FK_OK);
  if (ctx->h_table != NULL)
    fkx_pinned_free(ctx->h_table);
  ctx->h_table = NULL;
  ctx->h_table_cap = 0;
  const int64_t want = bytes + bytes / 32 + 4096;
  if (fkx_pinned_alloc((void **) &ctx->h_table, want) != FK_OK)
    { ctx->h_table = NULL;
      fk_set_error(ctx, "out of host memory: cannot pin %lld bytes for the result table", (long long) want);
      return (FK_ENOMEM);
    }
  ctx->h_table_cap = want;
  return (FK_OK);
}

extern "C" const char *fk_last_error(const fk_ctx *ctx)
{ return (ctx != NULL ? ctx->err : g_last_error); }

extern "C" const char *fk_version(void)
{ return ("fastk_amd 0.1 (gfx950)"); }

// ---- widths: FastK.c:417,446-468 with PAD_LEN = MIN_LEN = 5 (split.c:56) ------------------------
extern "C" int fk_get_widths(int kmer, fk_widths *w)
{ if (w == NULL || kmer < 5)
    return (FK_EINVAL);
  int v, bits = 0;
  w->kmer       = kmer;
  w->min_len    = 5;      // the widths are those of PAD_LEN = 5; the splitter's 7-mer minimizers only make
                          // super-mers of at most kmer - 6 k-mers, which these records hold
  w->max_super  = kmer - 4;
  for (v = w->max_super; v > 0; v >>= 1)
    bits += 1;
  w->slen_bytes = (bits + 7) >> 3;
  w->smer_bytes = (2 * (w->max_super + kmer - 1) + 7) >> 3;
  w->smer_word  = w->smer_bytes + w->slen_bytes;
  w->kmer_bytes = (2 * kmer + 7) >> 3;
  w->kmer_word  = w->kmer_bytes + 2;
  w->smer_stride = (w->smer_word + 3) & ~3;
  w->kmer_stride = (w->kmer_word + 3) & ~3;
  return (FK_OK);
}

extern "C" void fk_default_params(fk_params *p)
{ memset(p, 0, sizeof(*p));
  p->kmer = 40;            // FastK.c:232
  p->table_cutoff = 0;
  p->nthreads = 4;         // FastK.c:234
  p->bc_prefix = 0;
  p->device = 0;
  p->nbuckets = 1;
  p->hbm_budget = 0;
}

// ---- minimizer order: canonical 7-mers ranked by fk_mrank14 ----------------------------------------
static uint32_t revcomp7(uint32_t v)
{ uint32_t rc = 0;
  for (int j = 0; j < FK_MIN_LEN; j++)
    rc |= (3u - ((v >> (2 * j)) & 3u)) << (2 * (FK_MIN_LEN - 1 - j));
  return (rc);
}

static void build_minimizer_tables(uint8_t *mbucket, int nbuckets)
{ // Buckets must carry equal loads (one bucket = one GPU).  A smaller rank wins more windows, so the
  // load of a rank falls monotonically with its value: deal the ranks that occur (the images of the
  // canonical codes) out in serpentine order.
  static bool used[FK_NRANKS];
  for (int r = 0; r < FK_NRANKS; r++)
    used[r] = false;
  for (uint32_t v = 0; v < FK_NRANKS; v++)
    used[fk_mrank14(std::min(v, revcomp7(v)))] = true;
  int p = 0;
  for (int r = 0; r < FK_NRANKS; r++)
    { mbucket[r] = 0;
      if (!used[r])
        continue;
      const int q = p % (2 * nbuckets);
      mbucket[r] = (uint8_t) (q < nbuckets ? q : 2 * nbuckets - 1 - q);
      p += 1;
    }
}

// ---- bucket training (the role of Determine_Scheme's trie balancing, split.c:617-766) -----------
// Host-side, on a sample of reads: how much work each canonical minimizer rank attracts.  counts[r]
// += 4 * super-mer starts + k-mer instances whose minimizer has rank r  (r < FK_NRANKS = 16384).
extern "C" int fk_bucket_census(fk_ctx *ctx, const char *bases, int64_t nbytes, int64_t *counts)
{ if (ctx == NULL || bases == NULL || counts == NULL || nbytes < 0) return (FK_EINVAL);
  const int K = ctx->prm.kmer, W = K - (FK_MIN_LEN - 1);
  std::vector<uint64_t> key((size_t) nbytes + 8);      // (rank, position) of the 7-mer starting at i
  std::vector<int32_t>  bad((size_t) nbytes + 8);      // invalid bases in [0, i)
  uint32_t code = 0;
  int32_t  nbad = 0;
  for (int64_t i = 0; i < nbytes; i++)
    { bad[i] = nbad;
      const unsigned char c = (unsigned char) bases[i] & 0xDF;
      uint32_t x = 0;
      if (c == 'A') x = 0; else if (c == 'C') x = 1; else if (c == 'G') x = 2; else if (c == 'T') x = 3;
      else nbad += 1;
      code = ((code << 2) | x) & 0x3fffu;
      if (i >= FK_MIN_LEN - 1)
        key[i - (FK_MIN_LEN - 1)] = ((uint64_t) fk_mrank14(std::min(code, revcomp7(code))) << 40)
                                  | (uint64_t) (i - (FK_MIN_LEN - 1));
    }
  bad[nbytes] = nbad;
  uint64_t prev = ~0ull;
  bool     pv = false;
  for (int64_t i = 0; i + K <= nbytes; i++)
    { const bool v = (bad[i + K] == bad[i]);
      if (v)
        { uint64_t m = ~0ull;
          for (int j = 0; j < W; j++)
            m = std::min(m, key[i + j]);
          const int r = (int) (m >> 40);
          counts[r] += 1;
          if (!pv || m != prev)
            counts[r] += 4;
          prev = m;
        }
      pv = v;
    }
  return (FK_OK);
}

// Deal the minimizer ranks to the buckets by measured weight (longest processing time first) instead
// of the default serpentine deal.  Every process of a sharded run must pass the SAME counts
// (all-reduce them first).  Any assignment gives correct results; this one balances the buckets.
extern "C" int fk_set_bucket_weights(fk_ctx *ctx, const int64_t *counts)
{ if (ctx == NULL || counts == NULL) return (FK_EINVAL);
  const int nb = ctx->prm.nbuckets;
  std::vector<std::pair<int64_t, int> > order(FK_NRANKS);
  for (int r = 0; r < FK_NRANKS; r++)
    order[r] = std::make_pair(-counts[r], r);
  std::sort(order.begin(), order.end());
  int64_t load[256];
  for (int b = 0; b < nb; b++)
    load[b] = 0;
  for (int i = 0; i < FK_NRANKS; i++)
    { int best = 0;
      for (int b = 1; b < nb; b++)
        if (load[b] < load[best])
          best = b;
      ctx->h_mbucket[order[i].second] = (uint8_t) best;
      load[best] += -order[i].first + 1;
    }
  FK_HIP(ctx, hipSetDevice(ctx->device));
  FK_HIP(ctx, hipMemcpy(ctx->d_mbucket, ctx->h_mbucket, FK_NRANKS, hipMemcpyHostToDevice));
  return (FK_OK);
}

extern "C" int fk_create(const fk_params *p, fk_ctx **out)
{ if (p == NULL || out == NULL)
    return (FK_EINVAL);
  *out = NULL;
  fk_ctx *ctx = (fk_ctx *) calloc(1, sizeof(fk_ctx));
  if (ctx == NULL)
    return (FK_ENOMEM);
  ctx->prm = *p;
  if (ctx->prm.nbuckets < 1) ctx->prm.nbuckets = 1;
  if (ctx->prm.nthreads < 1) ctx->prm.nthreads = 1;
  if (ctx->prm.nbuckets > 256 || fk_get_widths(p->kmer, &ctx->wid) != FK_OK || p->kmer > 64
      || p->kmer < 8)
    { fk_set_error(NULL, "fk_create: unsupported parameters (k=%d nbuckets=%d): k from 8 to 64 (the expansion "
                         "and counting kernels are built for k-mers of up to four 32-bit words), at most 256 buckets",
                   p->kmer, p->nbuckets);
      free(ctx);
      return (FK_EINVAL);
    }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    { fk_set_error(NULL, "fk_create: no HIP device visible (this library has no CPU path)");
      free(ctx);
      return (FK_ENODEVICE);
    }
  ctx->device = p->device;
  if (hipSetDevice(ctx->device) != hipSuccess)
    { fk_set_error(NULL, "fk_create: cannot select device %d", ctx->device);
      free(ctx);
      return (FK_ENODEVICE);
    }
  { hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess)
      { if (strstr(prop.gcnArchName, "gfx950") == NULL)
          { fk_set_error(NULL, "fk_create: device %d is %s; this library is built for gfx950 only",
                         ctx->device, prop.gcnArchName);
            free(ctx);
            return (FK_ENODEVICE);
          }
        ctx->num_cus = prop.multiProcessorCount;
      }
  }
#define CK(call) do { if ((call) != hipSuccess) { fk_set_error(NULL, "fk_create: %s failed", #call); \
                                                    fk_destroy(ctx); return (FK_EHIP); } } while (0)
  if (fkx_stream_get(ctx->device, &ctx->stream) != FK_OK)
    { fk_set_error(NULL, "fk_create: hipStreamCreateWithFlags failed");
      fk_destroy(ctx);
      return (FK_EHIP);
    }
  ctx->own_stream = true;                            // (set before anything else can fail: fk_destroy returns the stream to the pool)
  if (fkx_stream_get(ctx->device, &ctx->copy_stream) != FK_OK)
    { fk_set_error(NULL, "fk_create: hipStreamCreateWithFlags failed");
      fk_destroy(ctx);
      return (FK_EHIP);
    }
#define CKE(call) do { if ((call) != FK_OK) { fk_set_error(NULL, "fk_create: %s failed", #call); \
                                               fk_destroy(ctx); return (FK_EHIP); } } while (0)
  CKE(fkx_event_get(ctx->device, false, &ctx->reads_ev));
  CKE(fkx_event_get(ctx->device, true, &ctx->ev0));
  CKE(fkx_event_get(ctx->device, true, &ctx->ev1));
  CKE(fkx_event_get(ctx->device, true, &ctx->stage_ev[0]));
  CKE(fkx_event_get(ctx->device, true, &ctx->stage_ev[1]));
#undef CKE
  CK(hipMalloc((void **) &ctx->d_mbucket, FK_NRANKS));
  CK(hipMalloc((void **) &ctx->d_mbucket_pass, FK_NRANKS));
  CK(hipHostMalloc((void **) &ctx->h_mbucket_pass, FK_NRANKS, hipHostMallocDefault));
  CK(hipMalloc((void **) &ctx->d_scratch, 65536));
  CK(hipHostMalloc((void **) &ctx->h_scratch, 65536 + 32 * 256 * 8, hipHostMallocDefault));
  CK(hipMalloc((void **) &ctx->d_digit_hist, 32 * 256 * sizeof(u64)));
  if (ctx->prm.hbm_budget > 0 && !ctx->prm.exact_parts)
    ctx->chunk_bytes = std::min<int64_t>(std::max<int64_t>(ctx->prm.hbm_budget / 32, 64ll << 20), 2ll << 30);
  if (ctx->prm.hbm_budget > 0)
    // a quarter of the budget is left for what is alive while a bucket is counted (two read
    // buffers, a chunk's split output, the bucket's records and weighted k-mers, the growing table)
    // (never below half the budget: a budget under 512 MB used to make this negative, which fkx_slab_alloc reads as
    // "no limit" -- and then the first slab alone was 8 GB)
    ctx->spill_limit = std::max<int64_t>(ctx->prm.hbm_budget - std::max<int64_t>(ctx->prm.hbm_budget / 4, 256ll << 20),
                                         std::max<int64_t>(ctx->prm.hbm_budget / 2, 1));
  build_minimizer_tables(ctx->h_mbucket, ctx->prm.nbuckets);
  CK(hipMemcpy(ctx->d_mbucket, ctx->h_mbucket, FK_NRANKS, hipMemcpyHostToDevice));
#undef CK
  pthread_mutex_t *m = (pthread_mutex_t *) malloc(sizeof(pthread_mutex_t));
  pthread_mutex_init(m, NULL);
  ctx->push_lock = m;
  *out = ctx;
  return (FK_OK);
}

extern "C" void fk_destroy(fk_ctx *ctx)
{ if (ctx == NULL)
    return;
  hipSetDevice(ctx->device);
  (void) fkx_flush_join(ctx);
  // Nothing this context queued may still be running when its host buffers go back to the allocator: the copy and
  // compute streams, and the part writers' streams, whose strip kernels store into the pinned staging freed below
  // (a writer that failed half-way leaves its next piece in flight).  The context's own streams one by one (a stream
  // the caller installed with fk_set_stream is the caller's to keep alive), then the device as a whole for the rest.
  if (ctx->copy_stream != NULL)
    hipStreamSynchronize(ctx->copy_stream);
  if (ctx->stream != NULL)
    hipStreamSynchronize(ctx->stream);
  for (int i = 0; i < 4; i++)
    if (ctx->wstream[i] != NULL)
      hipStreamSynchronize(ctx->wstream[i]);
  // ... then the device as a whole.  ADVICE r5 notes that this stalls a host with other contexts or work of its own on
  // the device; it stays all the same (round 6): the hipFree calls below synchronise the device anyway, and the first
  // GPU box this round's library ran on WITHOUT it was lost under 32 fuzz processes -- cause unknown, the conservative
  // order is the one that ran 60,000 iterations in round 5.  (The stream / event pools assume no hipDeviceReset.)
  (void) hipDeviceSynchronize();
  (void) hipGetLastError();
  hipFree(ctx->d_mbucket); hipFree(ctx->d_mbucket_pass); hipFree(ctx->d_scratch); hipFree(ctx->d_cursors); hipFree(ctx->d_plan);
  if (ctx->h_mbucket_pass) hipHostFree(ctx->h_mbucket_pass);
  if (ctx->h_scratch) hipHostFree(ctx->h_scratch);
  hipFree(ctx->d_digit_hist);
  hipFree(ctx->d_reads);
  hipFree(ctx->d_reads_alt);
  for (int i = 0; i < 2; i++)
    { hipFree(ctx->pk[i].roff); hipFree(ctx->pk[i].inv); }
  if (ctx->h_pk) hipHostFree(ctx->h_pk);
  hipFree(ctx->d_min_part);
  free(ctx->min_part);
  fkx_stream_put(ctx->device, ctx->copy_stream);      // (streams go back to the pool, never to hipStreamDestroy: fk_common.h)
  fkx_event_put(ctx->device, false, &ctx->reads_ev);   // (events too: fk_common.h)
  free(ctx->h_prof);
  free(ctx->h_prof_off);
  free(ctx->h_prof_split);
  free(ctx->blocks);
  for (int i = 0; i < FK_NSLOTS; i++)
    if (ctx->slot_ptr[i] != NULL)
      hipFree(ctx->slot_ptr[i]);
  for (int i = 0; i < 2; i++)
    { if (ctx->h_stage[i]) fkx_pinned_free(ctx->h_stage[i]);
      fkx_event_put(ctx->device, true, &ctx->stage_ev[i]);
    }
  fkx_event_put(ctx->device, true, &ctx->ev0);
  fkx_event_put(ctx->device, true, &ctx->ev1);
  for (int i = 0; i < 128; i++)
    fkx_event_put(ctx->device, true, &ctx->pass_ev[i]);
  if (ctx->own_stream && ctx->stream != NULL)
    fkx_stream_put(ctx->device, ctx->stream);
  for (int i = 0; i < ctx->nchunks; i++)
    fkx_free_chunk(ctx, &ctx->chunks[i]);
  free(ctx->chunks);
  for (int i = 0; i < ctx->nslabs; i++)
    hipFree(ctx->slabs[i].ptr);
  free(ctx->slabs);
  for (int i = 0; i < ctx->nspill; i++)
    if (ctx->spill_buf[i].ptr != NULL)
      fkx_pinned_free(ctx->spill_buf[i].ptr);
  free(ctx->spill_buf);
  if (ctx->h_table) fkx_pinned_free(ctx->h_table);
  free(ctx->acc_res);
  free(ctx->h_roff);
  free(ctx->ktab_ends);
  if (ctx->h_wstage) hipHostFree(ctx->h_wstage);
  for (int i = 0; i < 4; i++)
    fkx_stream_put(ctx->device, ctx->wstream[i]);
  if (ctx->push_lock)
    { pthread_mutex_destroy((pthread_mutex_t *) ctx->push_lock);
      free(ctx->push_lock);
    }
  free(ctx);
}

/* Returns the context's device buffers (arena, slab store, read buffers) and its pinned staging and spill buffers
   while the caller still works on the results of the last fk_finish, which stay valid (the table in host memory,
   histogram, profiles; with keep_table also the sorted table in HBM, for fk_write_ktab_device).  May run in a thread of its own beside the file writers: returning 280 GB of HBM takes the
   driver seconds, as long as writing a 36 GB table does.  The context can be used again afterwards (the buffers come
   back on demand), but the device copy of the last table is gone. */
extern "C" int fk_release_device(fk_ctx *ctx, int keep_table)
{ if (ctx == NULL) return (FK_EINVAL);
  FK_HIP(ctx, hipSetDevice(ctx->device));
  (void) fkx_flush_join(ctx);
  if (ctx->copy_stream != NULL)
    FK_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
  if (ctx->stream != NULL)
    FK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < ctx->nchunks; i++)
    fkx_free_chunk(ctx, &ctx->chunks[i]);
  ctx->nchunks = 0;
  for (int i = 0; i < ctx->nslabs; i++)
    hipFree(ctx->slabs[i].ptr);
  ctx->nslabs = 0;
  ctx->chunk_hbm_bytes = 0;
  for (int i = 0; i < FK_NSLOTS; i++)
    if (ctx->slot_ptr[i] != NULL)
      { if (keep_table && ctx->last_table != NULL && (char *) ctx->last_table >= (char *) ctx->slot_ptr[i]
            && (char *) ctx->last_table < (char *) ctx->slot_ptr[i] + ctx->slot_cap[i])
          continue;                              // the sorted table lives here
        hipFree(ctx->slot_ptr[i]);
        ctx->slot_ptr[i] = NULL;
        ctx->slot_cap[i] = 0;
      }
  hipFree(ctx->d_reads);     ctx->d_reads = NULL;     ctx->reads_cap = 0; ctx->reads_len = 0;
  hipFree(ctx->d_reads_alt); ctx->d_reads_alt = NULL; ctx->reads_cap_alt = 0;
  for (int i = 0; i < 2; i++)
    { hipFree(ctx->pk[i].roff); hipFree(ctx->pk[i].inv);
      memset(&ctx->pk[i], 0, sizeof(ctx->pk[i]));
    }
  if (ctx->h_pk) { hipHostFree(ctx->h_pk); ctx->h_pk = NULL; ctx->h_pk_cap = 0; }
  ctx->push_form = 0; ctx->pk_ascii_len = 0;
  for (int i = 0; i < ctx->nspill; i++)
    if (ctx->spill_buf[i].ptr != NULL && !ctx->spill_buf[i].in_use)
      { fkx_pinned_free(ctx->spill_buf[i].ptr);
        ctx->spill_buf[i].ptr = NULL;
        ctx->spill_buf[i].cap = 0;
      }
  for (int i = 0; i < 2; i++)
    if (ctx->h_stage[i])
      { fkx_pinned_free(ctx->h_stage[i]);
        ctx->h_stage[i] = NULL;
      }
  ctx->stage_cap = 0;
  ctx->pf_dict_table = NULL;
  if (!keep_table)
    { ctx->have_table = ctx->have_part_table = false;
      ctx->last_table = NULL;
      ctx->last_ntab = 0;
    }
  return (FK_OK);
}

/* exact_parts only: the reference's -M in bytes (FastK.c:235,291: 12e9 by default, <int> x 1e9) and, when the caller
   knows it, the ratio "whole input / first block" in file bytes (io.c:528,749) -- with both an exact_parts run cuts
   the input into the reference's NPARTS buckets under the reference's scheme (fk_scheme.hip), which is what decides
   where the hidden .ktab part files are cut.  sort_memory 0 (the default): one bucket. */
extern "C" int fk_set_sort_memory(fk_ctx *ctx, int64_t sort_memory, double input_ratio)
{ if (ctx == NULL || sort_memory < 0 || input_ratio < 0.) return (FK_EINVAL);
  ctx->sort_memory = sort_memory;
  ctx->input_ratio = input_ratio;
  return (FK_OK);
}

extern "C" int fk_set_stream(fk_ctx *ctx, void *hip_stream)
{ if (ctx == NULL) return (FK_EINVAL);
  if (ctx->own_stream && ctx->stream != NULL)
    fkx_stream_put(ctx->device, ctx->stream);
  ctx->stream = (hipStream_t) hip_stream;
  ctx->own_stream = false;
  return (FK_OK);
}

extern "C" int fk_synchronize(fk_ctx *ctx)
{ if (ctx == NULL) return (FK_EINVAL);
  FK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return (FK_OK);
}

// ---- utilities ----------------------------------------------------------------------------------
extern "C" int fk_device_alloc(fk_ctx *ctx, int64_t nbytes, void **d_ptr)
{ if (ctx == NULL || d_ptr == NULL || nbytes < 0) return (FK_EINVAL);
  *d_ptr = NULL;
  FK_HIP(ctx, hipSetDevice(ctx->device));
  FK_HIP(ctx, hipMalloc(d_ptr, (size_t) (nbytes > 0 ? nbytes : 16)));
  return (FK_OK);
}

extern "C" int fk_device_free(fk_ctx *ctx, void *d_ptr)
{ if (ctx == NULL) return (FK_EINVAL);
  FK_HIP(ctx, hipFree(d_ptr));
  return (FK_OK);
}

extern "C" int fk_copy_to_device(fk_ctx *ctx, void *d_dst, const void *src, int64_t nbytes)
{ if (ctx == NULL) return (FK_EINVAL);
  return fkx_h2d_pageable(ctx, ctx->stream, d_dst, src, (size_t) nbytes);
}

extern "C" int fk_copy_to_host(fk_ctx *ctx, void *dst, const void *d_src, int64_t nbytes)
{ if (ctx == NULL) return (FK_EINVAL);
  return fkx_d2h_pageable(ctx, ctx->stream, dst, d_src, (size_t) nbytes);
}

/* Test and measurement knobs (never used by the product path).  Every knob of a normal build keeps
   results valid (alternative code paths, smaller limits); the kernels whose output is wrong on purpose
   exist only in -DFK_ABLATION builds. */
extern "C" int fk_debug_set(fk_ctx *ctx, const char *key, int64_t value)
{ if (ctx == NULL || key == NULL) return (FK_EINVAL);
  // Only "verbose" is for users.  Everything else selects another (valid, slower or smaller) code path and exists for
  // the parity tests and measurement scripts: the library takes those knobs only from a process that says it is
  // one (FASTK_AMD_TEST_KNOBS=1: tests/conftest.py and bench.py --debug set it), so that no product caller can
  // switch the path it runs by accident.
  if (strcmp(key, "verbose") != 0)
    { const char *e = getenv("FASTK_AMD_TEST_KNOBS");
      if (e == NULL || strcmp(e, "1") != 0)
        { fk_set_error(ctx, "fk_debug_set(%s): test and measurement knobs need FASTK_AMD_TEST_KNOBS=1 in the environment", key);
          return (FK_EUNSUPPORTED);
        }
    }
  if (strcmp(key, "radix_variant") == 0 || strcmp(key, "radix_items") == 0 || (strcmp(key, "radix_engine") == 0 && value == 1))
    { fk_set_error(ctx, "fk_debug_set(%s): the look-back radix engine and its ablated variants were removed in round 6", key);
      return (FK_EUNSUPPORTED);
    }
#ifdef FK_ABLATION
  if (strcmp(key, "scatter_abl") == 0)     // ablated stream-engine scatters: WRONG output (tools/scatter_ablation.py)
    { ctx->dbg_scatter_abl = (int) value;
      return (FK_OK);
    }
#endif
  if (strcmp(key, "radix_engine") == 0)    // 2 / 3: narrow / wide stream tiles whatever the width, 4: stable first pass
    { ctx->dbg_radix_engine = (int) value;
      return (FK_OK);
    }
  if (strcmp(key, "exact_chain") == 0)      // 1..5: entries of the minimizer chain the exact splitter keeps in registers
    { ctx->dbg_exact_chain = (int) value;
      return (FK_OK);
    }
  if (strcmp(key, "exact_segments") == 0)   // 0: the exact splitter keeps one thread per read
    { ctx->dbg_exact_segments = (value == 0) ? -1 : 0;
      return (FK_OK);
    }
  if (strcmp(key, "slab_bytes") == 0)       // size of the chunk store's slabs (fk_ingest.hip)
    { if (value < 0) return (FK_EINVAL);
      ctx->dbg_slab_bytes = value;
      return (FK_OK);
    }
  if (strcmp(key, "smer_stage") == 0)       // 1: four grouping passes + run detection instead of LDS de-duplication
    { ctx->dbg_smer_stage = (int) value;
      return (FK_OK);
    }
  if (strcmp(key, "table_sort") == 0)       // 1: plain full-key table sort, >= 2: prefix bytes of the short one
    { ctx->dbg_table_sort = (int) value;
      return (FK_OK);
    }
  if (strcmp(key, "verbose") == 0)          // per-bucket sizes and times on stderr
    { ctx->dbg_verbose = (int) value;
      return (FK_OK);
    }
  if (strcmp(key, "spill_limit") == 0)      // chunk records kept in HBM before spilling to the host (tests)
    { ctx->spill_limit = value;
      return (FK_OK);
    }
  if (strcmp(key, "chunk_bytes") == 0)      // split the pushed reads every so many bytes (tests)
    { ctx->chunk_bytes = value;
      return (FK_OK);
    }
  if (strcmp(key, "aggr_variant") == 0)     // bit 0: no inserts, bit 1: no histogram, bit 2: no table (WRONG output)
    {
#ifdef FK_ABLATION
      ctx->dbg_aggr_variant = (int) value;
      return (FK_OK);
#else
      fk_set_error(ctx, "fk_debug_set(aggr_variant): ablations are only in -DFK_ABLATION builds");
      return (FK_EUNSUPPORTED);
#endif
    }
  if (strcmp(key, "aggr_engine") == 0)      // 1: the counting-sort aggregation of round 3; 0: k_ag_count2
    { ctx->dbg_aggr_engine = (int) value;
      return (FK_OK);
    }
  if (strcmp(key, "aggr_limit") == 0)
    { ctx->dbg_aggr_limit = (int) value;
      return (FK_OK);
    }
  if (strcmp(key, "table_prefix") == 0)
    { ctx->dbg_table_prefix = (int) value;
      return (FK_OK);
    }
  if (strcmp(key, "aggr_gshift") == 0)      // bins merged per table fill = 2^(value - 1); 0 = automatic
    { ctx->dbg_aggr_gshift = (int) value;
      return (FK_OK);
    }
  if (strcmp(key, "split_replay") == 0)     // 0: every pass of a multi-pass split recomputes the minimizers
    { ctx->dbg_no_replay = (value == 0) ? 1 : 0;
      return (FK_OK);
    }
  if (strcmp(key, "kmer_stage") == 0)       // 1: sort / collapse / sort instead of hash aggregation; 2: two hashed passes over
                                            //    the weighted k-mers instead of sorted references (fk_recut.hip)
    { ctx->dbg_kmer_stage = (int) value;
      return (FK_OK);
    }
  return (FK_EINVAL);
}

extern "C" int fk_debug_get(fk_ctx *ctx, const char *key, int64_t *value)
{ if (ctx == NULL || key == NULL || value == NULL) return (FK_EINVAL);
  if (strcmp(key, "aggr_extra_rounds") == 0)
    { *value = ctx->aggr_extra_rounds;
      return (FK_OK);
    }
  if (strcmp(key, "scheme_nparts") == 0)    // buckets of the last exact_parts run (1: the unpadded one-bucket case)
    { *value = ctx->scheme_nparts;
      return (FK_OK);
    }
  if (strcmp(key, "spilled_bytes") == 0)    // super-mer records the last chunked run moved to host memory
    { *value = ctx->spilled_bytes;
      return (FK_OK);
    }
  if (strcmp(key, "table_sort_ties") == 0)
    { *value = ctx->tsort_ties;
      return (FK_OK);
    }
  // where the pushed reads of a resident run lie in HBM and how many bytes they take: for harnesses that compare the
  // device copy with what they pushed (tests/fuzz_parity.py)
  if (strcmp(key, "event_pool") == 0)       // idle events in the process-wide pool (fkx_event_get / _put)
    { pthread_mutex_lock(&g_stream_lock);
      *value = (int64_t) g_event_pool.size();
      pthread_mutex_unlock(&g_stream_lock);
      return (FK_OK);
    }
  if (strcmp(key, "stream_pool") == 0)      // idle streams in the process-wide pool (fkx_stream_get / _put)
    { pthread_mutex_lock(&g_stream_lock);
      *value = (int64_t) g_stream_pool.size();
      pthread_mutex_unlock(&g_stream_lock);
      return (FK_OK);
    }
  if (strcmp(key, "reads_ptr") == 0 || strcmp(key, "reads_len") == 0)
    { if (getenv("FASTK_AMD_TEST_KNOBS") == NULL)
        return (FK_EINVAL);
      *value = (key[6] == 'p') ? (int64_t) (uintptr_t) ctx->d_reads : ctx->reads_len;
      return (FK_OK);
    }
  return (FK_EINVAL);
}

extern "C" int fk_get_sort_stats(fk_ctx *ctx, fk_sort_stats *st)
{ if (ctx == NULL || st == NULL) return (FK_EINVAL);
  *st = ctx->sort_stats;
  return (FK_OK);
}

// ---- synthetic reads (include/fk_synth.h), one thread per base ----------------------------------
__global__ __launch_bounds__(256) void k_synth(fk_synth_spec sp, uint64_t first_read, int64_t nreads,
                                               unsigned char *out)
{ const int64_t stride = (int64_t) sp.read_len + 1;
  const int64_t total  = nreads * stride;
  for (int64_t g = (int64_t) blockIdx.x * 256 + threadIdx.x; g < total;
       g += (int64_t) gridDim.x * 256)
    { const int64_t r = g / stride;
      const uint32_t j = (uint32_t) (g - r * stride);
      unsigned char c = 0;
      if (j < sp.read_len)
        { uint64_t start; uint32_t strand;
          fk_synth_place(&sp, first_read + (uint64_t) r, &start, &strand);
          const uint32_t b = fk_synth_base(&sp, first_read + (uint64_t) r, j, start, strand);
          c = (unsigned char) ("acgt"[b]);
        }
      out[g] = c;
    }
}

int fkx_synth(fk_ctx *ctx, uint64_t seed, uint64_t genome_len, uint32_t read_len,
              uint32_t err_ppm, uint64_t first_read, int64_t nreads, void *d_bases)
{ if (read_len == 0 || genome_len < read_len)
    { fk_set_error(ctx, "fk_synth_reads: genome shorter than a read");
      return (FK_EINVAL);
    }
  fk_synth_spec sp;
  sp.seed = seed; sp.genome_len = genome_len; sp.read_len = read_len; sp.err_ppm = err_ppm;
  const int64_t total = nreads * ((int64_t) read_len + 1);
  int64_t nb = (total + 255) / 256;
  if (nb > 65536) nb = 65536;
  if (nb > 0)
    { hipLaunchKernelGGL(k_synth, dim3((unsigned) nb), dim3(256), 0, ctx->stream, sp, first_read,
                         nreads, (unsigned char *) d_bases);
      FK_LAUNCH_CHECK(ctx);
    }
  return (FK_OK);
}

extern "C" int fk_synth_reads(fk_ctx *ctx, uint64_t seed, uint64_t genome_len, uint32_t read_len,
                              uint32_t err_ppm, uint64_t first_read, int64_t nreads, void *d_bases)
{ if (ctx == NULL || d_bases == NULL || nreads < 0) return (FK_EINVAL);
  return fkx_synth(ctx, seed, genome_len, read_len, err_ppm, first_read, nreads, d_bases);
}

extern "C" int fk_pack_fixed_reads(fk_ctx *ctx, const void *d_bases, int64_t nreads, uint32_t read_len, void *d_codes)
{ if (ctx == NULL || d_bases == NULL || d_codes == NULL || nreads < 0 || read_len == 0) return (FK_EINVAL);
  return fkx_pack_fixed(ctx, d_bases, nreads, read_len, d_codes);
}

// ---- stage interface ----------------------------------------------------------------------------
extern "C" int fk_split_supermers(fk_ctx *ctx, const void *d_bases, int64_t nbytes, void *d_out,
                                  int64_t cap, int64_t *nsuper, int64_t *ninst,
                                  int64_t *bucket_counts)
{ if (ctx == NULL || d_bases == NULL || nbytes < 0) return (FK_EINVAL);
  if (((uintptr_t) d_bases & 15) != 0)
    { fk_set_error(ctx, "fk_split_supermers: read buffer must be 16-byte aligned");
      return (FK_EINVAL);
    }
  return fkx_split(ctx, d_bases, nbytes, d_out, cap, nsuper, ninst, bucket_counts, false);
}

/* Emit only: bucket_counts[] holds the result of an earlier fk_split_supermers(cap = 0) call on the
   same input, so the counting kernel is not run again. */
extern "C" int fk_split_supermers_emit(fk_ctx *ctx, const void *d_bases, int64_t nbytes, void *d_out,
                                       int64_t cap, const int64_t *bucket_counts)
{ if (ctx == NULL || d_bases == NULL || d_out == NULL || bucket_counts == NULL || nbytes < 0)
    return (FK_EINVAL);
  if (((uintptr_t) d_bases & 15) != 0)
    { fk_set_error(ctx, "fk_split_supermers_emit: read buffer must be 16-byte aligned");
      return (FK_EINVAL);
    }
  int64_t bc[256], ns = 0, ni = 0;
  for (int b = 0; b < ctx->prm.nbuckets; b++)
    { bc[b] = bucket_counts[b];
      ns += bc[b];
    }
  if (cap < ns)
    { fk_set_error(ctx, "fk_split_supermers_emit: buffer holds %lld records, %lld needed",
                   (long long) cap, (long long) ns);
      return (FK_EINVAL);
    }
  if (ns == 0)
    return (FK_OK);
  return fkx_split(ctx, d_bases, nbytes, d_out, cap, &ns, &ni, bc, true);
}

/* One-pass bucketed split for the sharded path: fk_split_plan sizes padded per-bucket regions from
   a tile sample (offsets[nbuckets+1], *cap records in total); fk_split_planned emits into them and
   reports the real per-bucket counts.  FK_ESTATE = a region overflowed: use the exact
   fk_split_supermers(cap = 0) + fk_split_supermers_emit pair instead. */
extern "C" int fk_split_plan(fk_ctx *ctx, const void *d_bases, int64_t nbytes, int64_t *cap,
                             int64_t *offsets)
{ if (ctx == NULL || d_bases == NULL || cap == NULL || offsets == NULL || nbytes < 0) return (FK_EINVAL);
  return fkx_split_plan(ctx, d_bases, nbytes, cap, offsets);
}

extern "C" int fk_split_planned(fk_ctx *ctx, const void *d_bases, int64_t nbytes, void *d_out,
                                int64_t cap, const int64_t *offsets, int64_t *counts, int64_t *ninst)
{ if (ctx == NULL || d_bases == NULL || d_out == NULL || offsets == NULL || counts == NULL
      || ninst == NULL || nbytes < 0)
    return (FK_EINVAL);
  if (((uintptr_t) d_bases & 15) != 0)
    { fk_set_error(ctx, "fk_split_planned: read buffer must be 16-byte aligned");
      return (FK_EINVAL);
    }
  return fkx_split_planned(ctx, d_bases, nbytes, d_out, cap, offsets, counts, ninst);
}

extern "C" int fk_lsd_sort_records(fk_ctx *ctx, int64_t nelem, void *d_src, void *d_trg, int rsize,
                                   const int *bytes, void **result)
{ if (ctx == NULL || bytes == NULL || result == NULL || nelem < 0) return (FK_EINVAL);
  int nb = 0;
  while (bytes[nb] >= 0)
    nb += 1;
  return fkx_lsd_sort(ctx, nelem, d_src, d_trg, rsize, bytes, nb, result);
}

extern "C" int fk_msd_sort_records(fk_ctx *ctx, void *d_array, void *d_tmp, int64_t nelem, int rsize,
                                   int ksize, void **result)
{ if (ctx == NULL || result == NULL || nelem < 0 || ksize < 0 || ksize > rsize || ksize > 60)
    return (FK_EINVAL);
  // the MSD engine (fk_tsort.hip): ceil(log256 n) levels whatever ksize is, small parts finished in LDS
  return fkx_msd_sort(ctx, nelem, d_array, d_tmp, rsize, ksize, result);
}

extern "C" int fk_group_records(fk_ctx *ctx, int64_t nelem, void *d_src, void *d_trg, int rsize,
                                void **result)
{ if (ctx == NULL || result == NULL || nelem < 0) return (FK_EINVAL);
  return fkx_group(ctx, nelem, d_src, d_trg, rsize, rsize, 5, result);
}

extern "C" int fk_expand_kmers(fk_ctx *ctx, const void *d_smers, int64_t nsuper, void *d_out,
                               int64_t cap, int64_t *nweighted, int64_t *ndistinct,
                               int64_t *overflow)
{ int64_t a, b, c;
  if (ctx == NULL || nsuper < 0) return (FK_EINVAL);
  return fkx_expand(ctx, d_smers, nsuper, d_out, cap, nweighted ? nweighted : &a,
                    ndistinct ? ndistinct : &b, overflow ? overflow : &c);
}

extern "C" int fk_count_kmers(fk_ctx *ctx, const void *d_kmers, int64_t nweighted, int cutoff,
                              int64_t *hist, int64_t *max_inst, int64_t *ndistinct, void *d_table,
                              int64_t cap, int64_t *ntable)
{ if (ctx == NULL || hist == NULL || max_inst == NULL || nweighted < 0) return (FK_EINVAL);
  return fkx_count(ctx, d_kmers, nweighted, cutoff, ctx->wid.kmer_bytes, hist, max_inst, ndistinct,
                   d_table, cap, ntable);
}

extern "C" int fk_count_presorted_kmers(fk_ctx *ctx, const void *d_kmers, int64_t nweighted, int cutoff,
                                        int sorted_bytes, int64_t *hist, int64_t *max_inst,
                                        int64_t *ndistinct, void *d_table, int64_t cap, int64_t *ntable)
{ if (ctx == NULL || hist == NULL || max_inst == NULL || nweighted < 0 || sorted_bytes < 1)
    return (FK_EINVAL);
  if (sorted_bytes > ctx->wid.kmer_bytes)
    sorted_bytes = ctx->wid.kmer_bytes;
  return fkx_count(ctx, d_kmers, nweighted, cutoff, sorted_bytes, hist, max_inst, ndistinct, d_table,
                   cap, ntable);
}

extern "C" int fk_count_unsorted_kmers(fk_ctx *ctx, void *d_kmers, void *d_tmp, int64_t nweighted,
                                       int cutoff, int64_t *hist, int64_t *max_inst, int64_t *ndistinct,
                                       void **d_table, int64_t *ntable)
{ if (ctx == NULL || hist == NULL || max_inst == NULL || nweighted < 0) return (FK_EINVAL);
  if (nweighted > 0 && (d_kmers == NULL || d_tmp == NULL)) return (FK_EINVAL);
  if (cutoff > 0 && (d_table == NULL || ntable == NULL)) return (FK_EINVAL);
  const fk_widths &w = ctx->wid;
  void *grouped = d_kmers;
  int rc = fkx_group(ctx, nweighted, d_kmers, d_tmp, w.kmer_stride, w.kmer_bytes, 2, &grouped);
  if (rc != FK_OK) return (rc);
  void *tbuf = (grouped == d_kmers) ? d_tmp : d_kmers;
  int64_t nt = 0;
  rc = fkx_aggregate(ctx, grouped, nweighted, cutoff, hist, max_inst, ndistinct,
                     cutoff > 0 ? tbuf : NULL, nweighted, &nt);
  if (rc != FK_OK) return (rc);
  if (cutoff > 0)
    { void *sorted = tbuf;
      if (nt > 0 && (rc = fkx_sort_table(ctx, nt, tbuf, grouped, &sorted, NULL)) != FK_OK)
        return (rc);
      *d_table = sorted;
      *ntable = nt;
    }
  return (FK_OK);
}

/* Sum-merge of k-mer tables (Fastmerge.c:168-457): records = the entries of all input tables, any
   order, KMER_WORD bytes each (host).  Equal k-mers are summed, the count saturates at 0x7fff
   exactly as Fastmerge.c:313-329: hist[0x7fff] counts the saturated k-mers and max_inst receives, for a
   k-mer whose sum exceeds 0x7fff, the input counts below 0x7fff (the instances behind inputs that
   were saturated already come from the inputs' own histograms: pass their sum as max_inst_in).
   Runs on the aggregation + table-sort kernels of the counting path. */
extern "C" int fk_merge_tables(fk_ctx *ctx, const uint8_t *records, int64_t n, int64_t max_inst_in,
                               fk_result *res)
{ if (ctx == NULL || res == NULL || n < 0 || (records == NULL && n > 0)) return (FK_EINVAL);
  const fk_widths &w = ctx->wid;
  hipStream_t s = ctx->stream;
  memset(res, 0, sizeof(*res));
  res->max_inst = max_inst_in;
  if (n == 0)
    return (FK_OK);
  FK_HIP(ctx, hipSetDevice(ctx->device));
  void *d_a = fk_slot(ctx, FK_SLOT_KM_A, n * w.kmer_stride);
  void *d_b = fk_slot(ctx, FK_SLOT_KM_B, n * w.kmer_stride);
  if (d_a == NULL || d_b == NULL)
    return (FK_ENOMEM);
  // device layout (stride), and an input that is saturated already gets weight 0x8000: a sum reaches
  // 0x8000 exactly when Fastmerge calls the k-mer saturated (sum > 0x7fff, or a saturated input)
  std::vector<uint8_t> stage((size_t) n * w.kmer_stride, 0);
  int64_t nsat = 0;
  for (int64_t i = 0; i < n; i++)
    { const uint8_t *r = records + i * w.kmer_word;
      uint8_t *o = stage.data() + i * w.kmer_stride;
      memcpy(o, r, w.kmer_bytes);
      uint16_t c;
      memcpy(&c, r + w.kmer_bytes, 2);
      if (c >= 0x7fff) { c = 0x8000; nsat += 1; }
      memcpy(o + w.kmer_stride - 2, &c, 2);
    }
  { const int hrc = fkx_h2d_pageable(ctx, s, d_a, stage.data(), stage.size());
    if (hrc != FK_OK) return (hrc);
  }
  void *grouped = d_a;
  int rc = fkx_group(ctx, n, d_a, d_b, w.kmer_stride, w.kmer_bytes, 2, &grouped);
  if (rc != FK_OK) return (rc);
  void *tbuf = (grouped == d_a) ? d_b : d_a;
  int64_t nt = 0, nd = 0, mx = 0;
  ctx->aggr_sat = 0x8000;
  rc = fkx_aggregate(ctx, grouped, n, 1, res->hist, &mx, &nd, tbuf, n, &nt);
  ctx->aggr_sat = 0;
  if (rc != FK_OK) return (rc);
  res->max_inst += mx - 0x8000ll * nsat;
  res->ndistinct = nd;
  res->nweighted = n;
  res->ntable = nt;
  void *sorted = tbuf;
  int64_t census[256];
  if ((rc = fkx_sort_table(ctx, nt, tbuf, grouped, &sorted, census)) != FK_OK)
    return (rc);
  for (int x = 0; x < 256; x++)
    res->wfirst[x] = census[x];
  const int64_t bytes = nt * w.kmer_word;
  if (fkx_reserve_host_table(ctx, bytes) != FK_OK)
    return (FK_ENOMEM);
  std::vector<uint8_t> back((size_t) nt * w.kmer_stride);
  { const int hrc = fkx_d2h_pageable(ctx, s, back.data(), sorted, back.size());
    if (hrc != FK_OK) return (hrc);
  }
  for (int64_t i = 0; i < nt; i++)
    { memcpy(ctx->h_table + i * w.kmer_word, back.data() + i * w.kmer_stride, w.kmer_bytes);
      memcpy(ctx->h_table + i * w.kmer_word + w.kmer_bytes, back.data() + i * w.kmer_stride + w.kmer_stride - 2, 2);
    }
  res->table = ctx->h_table;
  return (FK_OK);
}

