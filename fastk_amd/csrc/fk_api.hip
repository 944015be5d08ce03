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
  if (hipMalloc(&p, (size_t) nbytes) != hipSuccess)
    { size_t fr = 0, tot = 0;
      int64_t held = 0;
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
  (void) hipHostUnregister(p);
  free(p);
  return (FK_OK);
}

static int reserve_host_table(fk_ctx *ctx, int64_t bytes)
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
  CK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  ctx->own_stream = true;
  CK(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
  CK(hipEventCreateWithFlags(&ctx->reads_ev, hipEventDisableTiming));
  CK(hipEventCreate(&ctx->ev0));
  CK(hipEventCreate(&ctx->ev1));
  CK(hipEventCreate(&ctx->stage_ev[0]));
  CK(hipEventCreate(&ctx->stage_ev[1]));
  CK(hipMalloc((void **) &ctx->d_mbucket, FK_NRANKS));
  CK(hipMalloc((void **) &ctx->d_mbucket_pass, FK_NRANKS));
  CK(hipHostMalloc((void **) &ctx->h_mbucket_pass, FK_NRANKS, hipHostMallocDefault));
  CK(hipMalloc((void **) &ctx->d_scratch, 65536));
  CK(hipHostMalloc((void **) &ctx->h_scratch, 65536 + 32 * 256 * 8, hipHostMallocDefault));
  CK(hipMalloc((void **) &ctx->d_digit_hist, 32 * 256 * sizeof(u64)));
  CK(hipMalloc((void **) &ctx->d_ticket, 64 * sizeof(u32)));
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
  if (ctx->copy_stream != NULL)
    hipStreamSynchronize(ctx->copy_stream);
  if (ctx->stream != NULL)
    hipStreamSynchronize(ctx->stream);
  hipFree(ctx->d_mbucket); hipFree(ctx->d_mbucket_pass); hipFree(ctx->d_scratch); hipFree(ctx->d_cursors); hipFree(ctx->d_plan);
  if (ctx->h_mbucket_pass) hipHostFree(ctx->h_mbucket_pass);
  if (ctx->h_scratch) hipHostFree(ctx->h_scratch);
  hipFree(ctx->d_digit_hist); hipFree(ctx->d_status); hipFree(ctx->d_ticket);
  hipFree(ctx->d_reads);
  hipFree(ctx->d_reads_alt);
  for (int i = 0; i < 2; i++)
    { hipFree(ctx->pk[i].roff); hipFree(ctx->pk[i].inv); }
  if (ctx->h_pk) hipHostFree(ctx->h_pk);
  hipFree(ctx->d_min_part);
  free(ctx->min_part);
  if (ctx->copy_stream) hipStreamDestroy(ctx->copy_stream);
  if (ctx->reads_ev) hipEventDestroy(ctx->reads_ev);
  free(ctx->h_prof);
  free(ctx->h_prof_off);
  free(ctx->h_prof_split);
  free(ctx->blocks);
  for (int i = 0; i < FK_NSLOTS; i++)
    if (ctx->slot_ptr[i] != NULL)
      hipFree(ctx->slot_ptr[i]);
  for (int i = 0; i < 2; i++)
    { if (ctx->h_stage[i]) fkx_pinned_free(ctx->h_stage[i]);
      if (ctx->stage_ev[i]) hipEventDestroy(ctx->stage_ev[i]);
    }
  if (ctx->ev0) hipEventDestroy(ctx->ev0);
  if (ctx->ev1) hipEventDestroy(ctx->ev1);
  for (int i = 0; i < 128; i++)
    if (ctx->pass_ev[i]) hipEventDestroy(ctx->pass_ev[i]);
  if (ctx->own_stream && ctx->stream != NULL)
    hipStreamDestroy(ctx->stream);
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
    if (ctx->wstream[i]) hipStreamDestroy(ctx->wstream[i]);
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
    { hipStreamSynchronize(ctx->stream);
      hipStreamDestroy(ctx->stream);
    }
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
  FK_HIP(ctx, hipMemcpyAsync(d_dst, src, (size_t) nbytes, hipMemcpyHostToDevice, ctx->stream));
  FK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return (FK_OK);
}

extern "C" int fk_copy_to_host(fk_ctx *ctx, void *dst, const void *d_src, int64_t nbytes)
{ if (ctx == NULL) return (FK_EINVAL);
  FK_HIP(ctx, hipMemcpyAsync(dst, d_src, (size_t) nbytes, hipMemcpyDeviceToHost, ctx->stream));
  FK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return (FK_OK);
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
#ifdef FK_ABLATION
  if (strcmp(key, "radix_variant") == 0)   // ablated look-back kernels: WRONG output, isolates one cost each
    { ctx->dbg_radix_variant = (int) value;
      return (FK_OK);
    }
  if (strcmp(key, "radix_items") == 0)
    { ctx->dbg_radix_items = (int) value;
      return (FK_OK);
    }
#else
  if (strcmp(key, "radix_variant") == 0 || strcmp(key, "radix_items") == 0 || (strcmp(key, "radix_engine") == 0 && value == 1))
    { fk_set_error(ctx, "fk_debug_set(%s): the look-back radix engine and its ablations are only in builds made with "
                        "-DFK_ABLATION (make -C fastk_amd/csrc ABLATION=1)", key);
      return (FK_EUNSUPPORTED);
    }
#endif
  if (strcmp(key, "radix_engine") == 0)    // 2 / 3: narrow / wide stream tiles whatever the width, 4: stable first pass
    { ctx->dbg_radix_engine = (int) value;
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
  if (strcmp(key, "kmer_stage") == 0)       // 1: sort / collapse / sort instead of hash aggregation
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
  FK_HIP(ctx, hipMemcpyAsync(d_a, stage.data(), stage.size(), hipMemcpyHostToDevice, s));
  FK_HIP(ctx, hipStreamSynchronize(s));
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
  if (reserve_host_table(ctx, bytes) != FK_OK)
    return (FK_ENOMEM);
  std::vector<uint8_t> back((size_t) nt * w.kmer_stride);
  FK_HIP(ctx, hipMemcpyAsync(back.data(), sorted, back.size(), hipMemcpyDeviceToHost, s));
  FK_HIP(ctx, hipStreamSynchronize(s));
  for (int64_t i = 0; i < nt; i++)
    { memcpy(ctx->h_table + i * w.kmer_word, back.data() + i * w.kmer_stride, w.kmer_bytes);
      memcpy(ctx->h_table + i * w.kmer_word + w.kmer_bytes, back.data() + i * w.kmer_stride + w.kmer_stride - 2, 2);
    }
  res->table = ctx->h_table;
  return (FK_OK);
}

// ---- whole path: split -> sort -> expand -> sort -> count ----------------------------------------
#define FK_GROUP_PASSES 4      // hashed digit passes that group super-mers
#define FK_LOW_BYTES    4      // key bytes sorted over all W records before equal neighbours are collapsed
#define FK_PREFIX_BYTES 64
// FK_PREFIX_BYTES 64 = disabled: sort every key byte.  A shorter prefix (fk_count_presorted_kmers) does not pay on
// read data: one-substitution error k-mers share long prefixes with their true k-mer, so about half
// of all prefix runs are heterogeneous and would need a local sort (measured).


static double ms_between(hipEvent_t a, hipEvent_t b)
{ float ms = 0.f;
  hipEventElapsedTime(&ms, a, b);
  return (double) ms;
}

// One bucket of super-mer records -> histogram / totals accumulated into res, table records.
//   final = true  (single bucket): the table is sorted here, *table_out points at it (device).
//   final = false (bucket streaming): the table records of this bucket, in no particular order, are
//                  appended to the slot FK_SLOT_TABLE at record *ntab (grown as needed).
// sm_in is clobbered (it is one half of the grouping's ping-pong pair).
struct fk_stage_ms { double group_s, expand, radix_k, aggr; };

// dig != NULL: the stream of hash digit 0 of the ns records, written by the splitter beside them.
static int count_bucket(fk_ctx *ctx, void *sm_in, int64_t ns, fk_result *res, bool final,
                        void **table_out, int64_t *ntab, const int64_t *exact_roff, fk_stage_ms *tm,
                        int64_t ns_max = 0, const uint8_t *dig = NULL)
{ const fk_widths &w = ctx->wid;
  hipStream_t s = ctx->stream;
  const int cutoff = ctx->prm.table_cutoff;
  hipEvent_t ev[4];
  int rc = FK_OK;

  if (ns <= 0)
    return (FK_OK);
  for (int i = 0; i < 4; i++)
    if (hipEventCreate(&ev[i]) != hipSuccess)
      { fk_set_error(ctx, "fk_finish: cannot create events");
        return (FK_EHIP);
      }
  do
    { // The buffers of the two stages are never live together: the second half of the super-mer
      // grouping's ping-pong pair lives in the k-mer slot A and the de-duplicated records in the k-mer
      // slot B (which is only sized for the weighted k-mers once the expansion has consumed them).
      const bool lds_dedup = (ctx->dbg_smer_stage != 1 && (w.smer_stride >> 2) >= 2 && (w.smer_stride >> 2) <= 7);
      void *sm_b = fk_slot(ctx, lds_dedup ? FK_SLOT_KM_A : FK_SLOT_SM_B, ns * w.smer_stride);
      void *km_a = NULL, *km_b = NULL;
      if (sm_b == NULL) { rc = FK_ENOMEM; break; }
      hipEventRecord(ev[0], s);

      // super-mer "sort": only has to bring identical records together (Supermer_Sort's output is
      // consumed by the run-length pass of count.c:421-426).  Two hashed digit passes put all copies
      // of a record into one of 65,536 bins and the aggregation kernel de-duplicates every bin in an
      // LDS hash table (distinct record + multiplicity); records too wide for that table, or
      // fk_debug_set("smer_stage",1), take four hashed passes and the expansion finds the runs itself
      void *sm_sorted = sm_in;
      int64_t nsx = ns;                          // records handed to the expansion
      bool    dd = false;
      if (lds_dedup)
        { void *grouped = sm_in;
          ctx->pre_dig = dig; ctx->pre_dig_n = ns;
          if ((rc = fkx_group(ctx, ns, sm_in, sm_b, w.smer_stride, w.smer_stride, 2, &grouped)) != FK_OK)
            break;
          res->passes_super      = ctx->sort_stats.passes;
          res->launches_super   += ctx->sort_stats.passes;
          res->ms_pass_super    += ctx->sort_stats.pass_ms_total;
          res->ms_scatter_super += ctx->sort_stats.scatter_ms_total;
          void *dd_out = fk_slot(ctx, FK_SLOT_KM_B, ns * (w.smer_stride + 4));
          if (dd_out == NULL) { rc = FK_ENOMEM; break; }
          rc = fkx_dedup_supermers(ctx, grouped, ns, dd_out, ns, &nsx);
          if (rc == FK_OK)
            { dd = true;
              sm_sorted = dd_out;
            }
          else if (rc == FK_ESTATE)
            { // a bin did not fit: group fully; the pair must not touch the k-mer slots any more
              void *other = fk_slot(ctx, FK_SLOT_SM_B, ns * w.smer_stride);
              if (other == NULL) { rc = FK_ENOMEM; break; }
              if (grouped != sm_in)
                { if (hipMemcpyAsync(sm_in, grouped, (size_t) (ns * w.smer_stride), hipMemcpyDeviceToDevice, s) != hipSuccess)
                    { rc = FK_EHIP; break; }
                  grouped = sm_in;
                }
              sm_sorted = grouped;
              nsx = ns;
              if ((rc = fkx_group(ctx, ns, grouped, other, w.smer_stride, w.smer_stride, FK_GROUP_PASSES, &sm_sorted)) != FK_OK)
                break;
            }
          else
            break;
        }
      else
        { if ((rc = fkx_group(ctx, ns, sm_in, sm_b, w.smer_stride, w.smer_stride, FK_GROUP_PASSES, &sm_sorted)) != FK_OK)
            break;
          res->passes_super      = ctx->sort_stats.passes;
          res->launches_super   += ctx->sort_stats.passes;
          res->ms_pass_super    += ctx->sort_stats.pass_ms_total;
          res->ms_scatter_super += ctx->sort_stats.scatter_ms_total;
        }
      hipEventRecord(ev[1], s);

      // weighted k-mer list
      int64_t nw = 0, nd = 0, ovf = 0;
      if ((rc = fkx_expand(ctx, sm_sorted, nsx, NULL, 0, &nw, &nd, &ovf, false, false, dd)) != FK_OK) break;
      res->nweighted += nw;
      res->ndistinct_super += nd;
      if (nw > 0)
        { // (slots grow on demand: hipMalloc / hipFree of tens of GB take well under a millisecond here,
          // and buckets balanced by work differ too much in density for a prediction to be worth it)
          const int64_t want = nw;
          if (ctx->dbg_verbose)
            fprintf(stderr, "  bucket sizing: %lld records (%lld after de-duplication), largest bucket ~%lld, "
                            "%lld weighted k-mers, buffers for %lld\n", (long long) ns, (long long) nsx,
                    (long long) ns_max, (long long) nw, (long long) want);
          if ((km_a = fk_slot(ctx, FK_SLOT_KM_A, want * w.kmer_stride)) == NULL)
            { rc = FK_ENOMEM; break; }
          if ((rc = fkx_expand(ctx, sm_sorted, nsx, km_a, nw, &nw, &nd, &ovf, true,
                               ctx->dbg_kmer_stage != 1 && exact_roff == NULL, dd)) != FK_OK) break;
          // (the de-duplicated super-mers in slot B are dead from here on)
          if ((km_b = fk_slot(ctx, FK_SLOT_KM_B, want * w.kmer_stride)) == NULL)
            { rc = FK_ENOMEM; break; }
        }
      int64_t exact_census[256];
      if (exact_roff != NULL)
        for (int x = 0; x < 256; x++)
          ctx->exact_wfirst[x] = 0;
      if (exact_roff != NULL && nw > 0
          && (rc = fkx_first_byte_census(ctx, km_a, nw, w.kmer_stride, exact_census)) != FK_OK)
        break;
      if (exact_roff != NULL && nw > 0)
        for (int x = 0; x < 256; x++)
          ctx->exact_wfirst[x] = exact_census[x];
      hipEventRecord(ev[2], s);

      // weighted k-mer stage.  The reference sorts the W weighted records on KMER_BYTES and scans the
      // result (MSDsort.c:536, 491-509).  Only the table has to come out in k-mer order, so here two
      // hashed digit passes bring all copies of a k-mer into one of 65,536 bins, one workgroup per bin
      // sums them in an LDS hash table (fk_aggr.hip: histogram, max_inst, table candidates), and only
      // the table records (count >= cutoff) are sorted on KMER_BYTES.  fk_debug_set("kmer_stage",1)
      // or a bin that does not fit selects the sort / collapse / sort path below instead.
      int64_t nt = 0, ndk = 0, ovf2 = 0;
      void   *tab = NULL;                       // device address of this bucket's table records
      bool    tab_sorted = false;
      bool    aggregated = false;
      bool    direct = false;                   // the table records are already in the union buffer
      float   ms_aggr = 0.f;
      if (nw > 0 && ctx->dbg_kmer_stage != 1)
        { void *grouped = km_a;
          if ((rc = fkx_group(ctx, nw, km_a, km_b, w.kmer_stride, w.kmer_bytes, 2, &grouped)) != FK_OK)
            break;
          res->passes_kmer      = ctx->sort_stats.passes;
          res->launches_kmer   += ctx->sort_stats.passes;
          res->ms_pass_kmer    += ctx->sort_stats.pass_ms_total;
          res->ms_scatter_kmer += ctx->sort_stats.scatter_ms_total;
          void *tbuf = (grouped == km_a) ? km_b : km_a;
          // Bucket streaming: the table candidates go straight behind those of the earlier buckets when the union
          // buffer has clearly enough room left (1.5 x what a bucket has brought so far); the kernel checks the
          // bound, and a bucket that does not fit after all is aggregated again into its own buffer.
          int64_t room = 0;
          if (!final && cutoff > 0 && ctx->slot_ptr[FK_SLOT_TABLE] != NULL && res->buckets_counted > 0 && ntab != NULL)
            { room = ctx->slot_cap[FK_SLOT_TABLE] / w.kmer_stride - *ntab;
              const int64_t expect = *ntab / res->buckets_counted;
              if (room < expect + expect / 2 + 4096)
                room = 0;
            }
          hipEventRecord(ctx->ev0, s);
          if (room > 0)
            { rc = fkx_aggregate(ctx, grouped, nw, cutoff, res->hist, &res->max_inst, &ndk,
                                 (char *) ctx->slot_ptr[FK_SLOT_TABLE] + *ntab * w.kmer_stride, room, &nt);
              if (rc == FKX_TABLE_FULL)
                room = 0;
              else if (rc == FK_OK)
                direct = true;
            }
          if (room == 0)
            rc = fkx_aggregate(ctx, grouped, nw, cutoff, res->hist, &res->max_inst, &ndk,
                               cutoff > 0 ? tbuf : NULL, nw, &nt);     // adds to hist only on success
          hipEventRecord(ctx->ev1, s);
          if (rc == FK_OK)
            { aggregated = true;
              hipEventSynchronize(ctx->ev1);
              hipEventElapsedTime(&ms_aggr, ctx->ev0, ctx->ev1);
              tab = tbuf;
              km_a = tbuf; km_b = grouped;       // km_a: table records, km_b: free
            }
          else if (rc == FK_ESTATE)
            { km_a = grouped; km_b = tbuf;       // all records are still there, in another order
              rc = FK_OK;
            }
          else
            break;
        }
      if (nw > 0 && !aggregated)
        { // The LSD sort over the W weighted records is interrupted after the FK_LOW_BYTES least
          // significant key bytes: equal k-mers that are adjacent by then are collapsed into one record
          // (weights summed, clipped like count.c:455-458) -- the collapse keeps the order, so the
          // remaining passes simply continue the same LSD sort on ~3x fewer records.  k-mers that were
          // not adjacent yet meet at the end and are summed by the count kernel.
          const int nlow = (w.kmer_bytes > FK_LOW_BYTES + 1) ? FK_LOW_BYTES : 0;
          int bytes[64];
          int64_t nc = nw;
          void *low = km_a;
          if (nlow > 0)
            { for (int i = 0; i < nlow; i++)
                bytes[i] = w.kmer_bytes - 1 - i;
              if ((rc = fkx_lsd_sort(ctx, nw, km_a, km_b, w.kmer_stride, bytes, nlow, &low)) != FK_OK)
                break;
              res->passes_kmer      = ctx->sort_stats.passes;
              res->launches_kmer   += ctx->sort_stats.passes;
              res->ms_pass_kmer    += ctx->sort_stats.pass_ms_total;
              res->ms_scatter_kmer += ctx->sort_stats.scatter_ms_total;
              void *cbuf = (low == km_a) ? km_b : km_a;
              if ((rc = fkx_collapse(ctx, low, nw, cbuf, nw, &nc, &ovf2)) != FK_OK)
                break;
              km_b = low; km_a = cbuf;                 // km_a holds the collapsed records
            }
          const int nhigh = w.kmer_bytes - nlow;
          for (int i = 0; i < nhigh; i++)
            bytes[i] = nhigh - 1 - i;
          void *km_sorted = km_a;
          if ((rc = fkx_lsd_sort(ctx, nc, km_a, km_b, w.kmer_stride, bytes, nhigh, &km_sorted)) != FK_OK)
            break;
          res->ms_pass_final += ctx->sort_stats.pass_ms_total;
          if (nlow == 0)
            { res->passes_kmer      = ctx->sort_stats.passes;
              res->ms_pass_kmer    += ctx->sort_stats.pass_ms_total;
              res->ms_scatter_kmer += ctx->sort_stats.scatter_ms_total;
            }
          for (int x = 0; x < 256; x++)          // first-byte census of the sorted records
            res->wfirst[x] = (exact_roff != NULL) ? exact_census[x] : (int64_t) ctx->h_scratch[x];
          void *other = (km_sorted == km_a) ? km_b : km_a;
          if ((rc = fkx_count(ctx, km_sorted, nc, cutoff, w.kmer_bytes, res->hist, &res->max_inst, &ndk,
                              cutoff > 0 ? other : NULL, nc, &nt)) != FK_OK)
            break;
          tab = other;
          tab_sorted = true;
          km_a = other; km_b = km_sorted;
        }
      res->max_inst  += ovf + ovf2;                           // count.c:1551
      res->ndistinct += ndk;
      if (cutoff > 0 && nt > 0)
        { if (final)
            { if (!tab_sorted)
                { void *sorted = tab;
                  int64_t census[256];
                  if ((rc = fkx_sort_table(ctx, nt, tab, km_b, &sorted, census)) != FK_OK)
                    break;
                  res->passes_final   = ctx->sort_stats.passes;
                  res->ms_pass_final += ctx->sort_stats.pass_ms_total;
                  tab = sorted;
                  for (int x = 0; x < 256; x++)
                    res->wfirst[x] = (exact_roff != NULL) ? exact_census[x] : census[x];
                }
              *table_out = tab;
            }
          else if (!direct)
            { // keep what earlier buckets appended while the slot grows
              const int64_t need = (*ntab + nt) * w.kmer_stride;
              if (ctx->slot_cap[FK_SLOT_TABLE] < need)
                { void *nbuf = NULL;
                  // the other buckets are about as rich as the ones seen so far: size for all of them
                  // at once (growing by copies costs more than the counting at tens of GB)
                  int64_t ncap = need + need / 2 + (1 << 20);
                  if (ns_max > 0 && ns > 0 && ctx->prm.nbuckets > 1)
                    { // buckets carry about equal work (fk_set_bucket_weights) or about equal records
                      // (default deal): the smaller of the two extrapolations (the slot grows again if
                      // that was too little)
                      const double by_sm = (double) (*ntab + nt) / (double) (ctx->acc_ns + ns) * (double) ctx->acc_ns_total;
                      const double by_bk = (double) (*ntab + nt) / (double) (res->buckets_counted + 1) * (double) ctx->prm.nbuckets;
                      const int64_t all = (int64_t) (std::min(by_sm, by_bk) * 1.10) * w.kmer_stride;
                      if (all > ncap) ncap = all + (1 << 20);
                    }
                  if (hipMalloc(&nbuf, (size_t) ncap) != hipSuccess)
                    { ncap = need + (1 << 20);                // no room for the extrapolation: what is needed now
                      if (hipMalloc(&nbuf, (size_t) ncap) != hipSuccess)
                        { fk_set_error(ctx, "out of HBM: cannot allocate %lld bytes for the table records", (long long) ncap);
                          rc = FK_ENOMEM;
                          break;
                        }
                    }
                  if (*ntab > 0
                      && hipMemcpyAsync(nbuf, ctx->slot_ptr[FK_SLOT_TABLE], (size_t) (*ntab * w.kmer_stride),
                                        hipMemcpyDeviceToDevice, s) != hipSuccess)
                    { hipFree(nbuf); rc = FK_EHIP; break; }
                  hipStreamSynchronize(s);
                  if (ctx->slot_ptr[FK_SLOT_TABLE] != NULL)
                    hipFree(ctx->slot_ptr[FK_SLOT_TABLE]);
                  ctx->slot_ptr[FK_SLOT_TABLE] = nbuf;
                  ctx->slot_cap[FK_SLOT_TABLE] = ncap;
                }
              if (hipMemcpyAsync((char *) ctx->slot_ptr[FK_SLOT_TABLE] + *ntab * w.kmer_stride, tab,
                                 (size_t) (nt * w.kmer_stride), hipMemcpyDeviceToDevice, s) != hipSuccess)
                { rc = FK_EHIP; break; }
            }
        }
      *ntab += (cutoff > 0) ? nt : 0;
      ctx->acc_ns += ns;
      res->buckets_counted += 1;
      hipEventRecord(ev[3], s);
      if (hipEventSynchronize(ev[3]) != hipSuccess) { rc = FK_EHIP; break; }
      if (ctx->dbg_verbose)
        fprintf(stderr, "  bucket: %lld super-mers, %lld weighted k-mers, %lld distinct, %.2f ms "
                        "(super-mers %.2f, expand %.2f, k-mers %.2f of which aggregation %.2f)\n",
                (long long) ns, (long long) nw, (long long) ndk, ms_between(ev[0], ev[3]),
                ms_between(ev[0], ev[1]), ms_between(ev[1], ev[2]), ms_between(ev[2], ev[3]), ms_aggr);
      tm->group_s += ms_between(ev[0], ev[1]);
      tm->expand  += ms_between(ev[1], ev[2]);
      tm->radix_k += ms_between(ev[2], ev[3]) - ms_aggr;
      tm->aggr    += ms_aggr;
    }
  while (0);
  for (int i = 0; i < 4; i++)
    hipEventDestroy(ev[i]);
  return (rc);
}

// The table candidates the buckets appended to FK_SLOT_TABLE -> one table in k-mer order.
static int sort_union_table(fk_ctx *ctx, int64_t ntab, fk_result *res, void **table, fk_stage_ms *tm)
{ const fk_widths &w = ctx->wid;
  hipStream_t s = ctx->stream;
  // second buffer of the sort: every per-bucket buffer is idle by now -- take one that is large
  // enough as it is (after a multi-pass split the super-mer slot is) before growing one
  void *tmp = NULL;
  { static const int idle[5] = { FK_SLOT_SM_A, FK_SLOT_KM_A, FK_SLOT_KM_B, FK_SLOT_SM_G, FK_SLOT_SM_D };
    for (int i = 0; i < 5 && tmp == NULL; i++)
      if (ctx->slot_cap[idle[i]] >= ntab * w.kmer_stride)
        tmp = ctx->slot_ptr[idle[i]];
    if (tmp == NULL)
      tmp = fk_slot(ctx, FK_SLOT_KM_A, ntab * w.kmer_stride);
  }
  int64_t census[256];
  if (tmp == NULL) return (FK_ENOMEM);
  hipEvent_t te[2];                              // (the sort itself records ctx->ev0 / ev1)
  if (hipEventCreate(&te[0]) != hipSuccess || hipEventCreate(&te[1]) != hipSuccess)
    return (FK_EHIP);
  hipEventRecord(te[0], s);
  *table = ctx->slot_ptr[FK_SLOT_TABLE];
  int rc = fkx_sort_table(ctx, ntab, ctx->slot_ptr[FK_SLOT_TABLE], tmp, table, census);
  if (rc != FK_OK)
    { hipEventDestroy(te[0]); hipEventDestroy(te[1]);
      return (rc);
    }
  res->passes_final   = ctx->sort_stats.passes;
  res->ms_pass_final += ctx->sort_stats.pass_ms_total;
  for (int x = 0; x < 256; x++)
    res->wfirst[x] = census[x];
  hipEventRecord(te[1], s);
  hipEventSynchronize(te[1]);
  tm->radix_k += ms_between(te[0], te[1]);
  res->ms_table_sort += ms_between(te[0], te[1]);
  hipEventDestroy(te[0]); hipEventDestroy(te[1]);
  return (FK_OK);
}

// device stride -> reference width (k-mer bytes + uint16 count), one thread per record
__global__ __launch_bounds__(256) void k_repack_table(const uint8_t *__restrict__ in, int64_t n, int stride, int kbytes,
                                                      uint8_t *__restrict__ out)
{ const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint8_t *r = in + i * stride;
  uint8_t *o = out + i * (kbytes + 2);
  for (int j = 0; j < kbytes; j++) o[j] = r[j];
  o[kbytes] = r[stride - 2];
  o[kbytes + 1] = r[stride - 1];
}

int fkx_repack_table(fk_ctx *ctx, const void *d_in, int64_t n, void *d_out)
{ if (n <= 0) return (FK_OK);
  hipLaunchKernelGGL(k_repack_table, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, ctx->stream,
                     (const uint8_t *) d_in, n, ctx->wid.kmer_stride, ctx->wid.kmer_bytes, (uint8_t *) d_out);
  FK_LAUNCH_CHECK(ctx);
  return (FK_OK);
}

// ntable / ncollapsed and, if asked, the table itself (host copy in reference layout) into res
static int fetch_result_table(fk_ctx *ctx, fk_result *res, void *table, int64_t nt, bool fetch_table)
{ const fk_widths &w = ctx->wid;
  hipStream_t s = ctx->stream;
  const int cutoff = ctx->prm.table_cutoff;
  res->ncollapsed = nt;
  res->ntable = (cutoff > 0) ? nt : 0;
  if (!(cutoff > 0 && nt > 0 && fetch_table))
    return (FK_OK);
  const int64_t bytes = nt * w.kmer_word;
  if (reserve_host_table(ctx, bytes) != FK_OK)
    return (FK_ENOMEM);
  if (w.kmer_word == w.kmer_stride)
    { if (hipMemcpyAsync(ctx->h_table, table, (size_t) bytes, hipMemcpyDeviceToHost, s) != hipSuccess
          || hipStreamSynchronize(s) != hipSuccess)
        return (FK_EHIP);
    }
  else
    { // records wider than KMER_WORD on the device (k = 41..48, 49..64 ...): packed by a kernel into an idle
      // buffer, then one copy -- not a per-record loop on the host
      void *pk = NULL;
      static const int idle[4] = { FK_SLOT_SM_A, FK_SLOT_KM_B, FK_SLOT_SM_G, FK_SLOT_SM_D };
      for (int i = 0; i < 4 && pk == NULL; i++)
        if (ctx->slot_cap[idle[i]] >= bytes && ctx->slot_ptr[idle[i]] != table
            && !((char *) table >= (char *) ctx->slot_ptr[idle[i]]
                 && (char *) table < (char *) ctx->slot_ptr[idle[i]] + ctx->slot_cap[idle[i]]))
          pk = ctx->slot_ptr[idle[i]];
      bool own = false;
      if (pk == NULL)
        { if (hipMalloc(&pk, (size_t) bytes) != hipSuccess) return (FK_ENOMEM);
          own = true;
        }
      int rc = fkx_repack_table(ctx, table, nt, pk);
      if (rc == FK_OK
          && (hipMemcpyAsync(ctx->h_table, pk, (size_t) bytes, hipMemcpyDeviceToHost, s) != hipSuccess
              || hipStreamSynchronize(s) != hipSuccess))
        rc = FK_EHIP;
      if (own) hipFree(pk);
      if (rc != FK_OK) return (rc);
    }
  res->table = ctx->h_table;
  return (FK_OK);
}

// d_smers_in != NULL: start from caller-owned super-mer records (sharded path, after the exchange);
// the caller's buffer is used as one half of the sort's ping-pong pair and is clobbered.
// With nbuckets > 1 and reads as input the buckets are processed one after the other ("bucket
// streaming": the k-mer buffers only ever hold one bucket's weighted k-mers); equal k-mers share a
// minimizer, hence a bucket, so histograms add up and the table is the sorted union.
// pk != NULL: d_reads holds the reads in two bits per base, nbytes = their positions (fk_pkview).
int fkx_pipeline(fk_ctx *ctx, const void *d_reads, int64_t nbytes, void *d_smers_in,
                 int64_t nsmers_in, fk_result *res, bool fetch_table, int64_t *h_roff = NULL,
                 int64_t nreads = 0, const fk_pkview *pk = NULL)
{ const fk_widths &w = ctx->wid;
  hipStream_t s = ctx->stream;
  hipEvent_t ev[3];
  int rc = FK_OK;

  memset(res, 0, sizeof(*res));
  for (int i = 0; i < 3; i++)
    if (hipEventCreate(&ev[i]) != hipSuccess)
      { fk_set_error(ctx, "fk_finish: cannot create events");
        return (FK_EHIP);
      }
  do
    { void *sm_a = NULL;
      uint8_t *sm_dig = NULL;            // first digit stream of the super-mer grouping, from a one-pass split
      int64_t ns = 0, ni = 0;
      int64_t bc[256] = { 0 }, bo[256] = { 0 };
      int     nbk = 1;
      int     ngroups = 1;                 // split passes over resident reads (fk_params.split_passes)
      int64_t gcap_all = 0, goffs[257];
      fk_stage_ms tm = { 0., 0., 0., 0. };

      hipEventRecord(ev[0], s);
      void *sm_in = d_smers_in;
      const bool chunked = (d_smers_in == NULL && d_reads == NULL);   // the chunks of fkx_flush_chunk
      if (chunked)
        { nbk = ctx->prm.nbuckets;
          for (int b = 0; b < nbk; b++)
            { bc[b] = 0;
              for (int c = 0; c < ctx->nchunks; c++)
                bc[b] += ctx->chunks[c].cnt[b];
              ns += bc[b];
            }
          res->nsuper = ns;
          res->ninst  = ctx->chunk_ninst;
        }
      else if (d_smers_in != NULL)
        { ns = nsmers_in;
          res->nsuper = ns;
        }
      else
        { if (h_roff != NULL)
            { // exact_parts: the reference's own rule, so that Table_Split falls where it does there
              h_roff[nreads] = nbytes;
              int64_t train = 0, olen = 0;                 // Get_First_Block(io, 1e9), io.c:2606-2630
              const int64_t maxrds = 1000000000ll / 150, omax = 1000000000ll + maxrds;
              while (train < nreads)
                { olen += h_roff[train + 1] - h_roff[train];
                  train += 1;
                  if (olen > omax - 100000 || train >= maxrds)
                    break;
                }
              int tran[4];
              if (ctx->have_tran)
                for (int x = 0; x < 4; x++)
                  tran[x] = ctx->tran[x];
              else if ((rc = fkx_train_tran(ctx, d_reads, h_roff, train, ctx->prm.nthreads, tran)) != FK_OK) break;
              for (int x = 0; x < 4; x++)
                ctx->exact_tran[x] = tran[x];
              ctx->exact_tran_set = true;
              int64_t *d_roff = (int64_t *) fk_slot(ctx, FK_SLOT_ROFF, (nreads + 1) * 8);
              if (d_roff == NULL) { rc = FK_ENOMEM; break; }
              if (hipMemcpyAsync(d_roff, h_roff, (size_t) (nreads + 1) * 8, hipMemcpyHostToDevice, s) != hipSuccess)
                { rc = FK_EHIP; break; }
              // how many buckets the reference would use (FastK.c:417-429: the k-mer records of the whole input,
              // extrapolated from the training block, over the sort memory) and, with more than one, its scheme
              ctx->scheme_nparts = 1;
              if (ctx->sort_memory > 0 && train > 0)
                { const int64_t totlen = (h_roff[train] - h_roff[0]) - train;          // bases of the block
                  const int64_t all = nbytes - nreads;
                  const double ratio = (ctx->input_ratio > 0.) ? ctx->input_ratio
                                     : (train >= nreads ? 1.0 : (double) (all + nreads) / (double) (totlen + train));
                  int64_t gsize = totlen - (int64_t) ctx->prm.kmer * train;
                  gsize = (int64_t) ((double) gsize * ratio * (double) w.kmer_word);
                  const int64_t np = (gsize - 1) / ctx->sort_memory + 1;
                  if (np > 1)
                    { if (np > FK_EXACT_MAXPARTS)
                        { fk_set_error(ctx, "exact_parts: the reference would cut this input into %lld buckets; this engine "
                                            "follows its scheme up to %d", (long long) np, FK_EXACT_MAXPARTS);
                          rc = FK_EUNSUPPORTED;
                          break;
                        }
                      if ((rc = fkx_train_scheme(ctx, d_reads, d_roff, train, tran, (int) np)) != FK_OK) break;
                    }
                }
              if ((rc = fkx_split_exact(ctx, d_reads, d_roff, nreads, tran, &sm_a, &ns, &ni, bc, bo)) != FK_OK) break;
              nbk = (ctx->scheme_nparts > 1) ? ctx->scheme_nparts : 1;
            }
          else
            { nbk = ctx->prm.nbuckets;
              // several split passes over resident reads, each emitting one group of buckets?
              if (nbk > 1 && nbk <= 255 && (ctx->prm.split_passes > 1 || (ctx->prm.split_passes == 0 && ctx->prm.hbm_budget > 0)))
                { if ((rc = fkx_split_plan(ctx, d_reads, nbytes, &gcap_all, goffs, pk)) != FK_OK) break;
                  ngroups = ctx->prm.split_passes;
                  if (ngroups <= 0)
                    { const int64_t half = std::max<int64_t>(ctx->prm.hbm_budget / 2, 1);
                      ngroups = (int) std::min<int64_t>((gcap_all * w.smer_stride + half - 1) / half, 255);
                    }
                  if (ngroups > nbk) ngroups = nbk;
                  if (gcap_all == 0) ngroups = 1;
                }
              if (ngroups <= 1)
                { // split (sampled capacity + one emit pass; exact count-then-emit with several buckets)
                  ngroups = 1;
                  if ((rc = fkx_split_fast(ctx, d_reads, nbytes, &sm_a, &ns, &ni, bc, bo, pk, &sm_dig)) != FK_OK) break;
                }
            }
          res->nsuper = ns;
          res->ninst = ni;
          sm_in = sm_a;
        }
      hipEventRecord(ev[1], s);
      double ms_split_groups = 0.;

      void   *table = NULL;
      int64_t ntab = 0;
      int64_t ns_max = 0;
      if (nbk == 1)
        bc[0] = ns;
      for (int b = 0; b < nbk; b++)
        ns_max = std::max(ns_max, bc[b]);
      ctx->acc_ns = 0;
      ctx->acc_ns_total = ns;
      // chunked ingest: a bucket's records are gathered from the chunks right before it is counted
      auto gather = [&](int b, void **ptr) -> int
        { char *g = (char *) fk_slot(ctx, FK_SLOT_SM_G, ns_max * w.smer_stride);
          if (g == NULL)
            return (FK_ENOMEM);
          int64_t run = 0;
          for (int c = 0; c < ctx->nchunks; c++)
            { const fk_chunk *ch = &ctx->chunks[c];
              if (ch->cnt[b] > 0
                  && hipMemcpyAsync(g + run * w.smer_stride, ch->run[b],
                                    (size_t) (ch->cnt[b] * w.smer_stride),
                                    ch->on_host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, s)
                     != hipSuccess)
                return (FK_EHIP);
              run += ch->cnt[b];
            }
          *ptr = g;
          return (FK_OK);
        };
      if (ngroups > 1)
        { // Multi-pass split: the buckets are dealt into ngroups runs of consecutive buckets of about
          // equal (estimated) size; every pass re-reads the reads and keeps one run's super-mers.
          int     gb[257];
          int64_t est[256], tot = 0, gmax = 0;
          for (int b = 0; b < nbk; b++)
            { est[b] = goffs[b + 1] - goffs[b];
              ns_max = std::max(ns_max, est[b]);
            }
          { // runs of consecutive buckets, at most ngroups of them, with the smallest possible largest
            // run: binary search on the run capacity, greedy fill
            int64_t lo_c = ns_max, hi_c = gcap_all;
            auto fill = [&](int64_t capv, int *bounds) -> int
              { int g = 0;
                int64_t acc = 0;
                bounds[0] = 0;
                for (int b = 0; b < nbk; b++)
                  { if (acc + est[b] > capv && acc > 0)
                      { bounds[++g] = b;
                        acc = 0;
                      }
                    acc += est[b];
                  }
                bounds[++g] = nbk;
                return (g);
              };
            while (lo_c < hi_c)
              { const int64_t mid = lo_c + (hi_c - lo_c) / 2;
                if (fill(mid, gb) <= ngroups) hi_c = mid; else lo_c = mid + 1;
              }
            ngroups = fill(lo_c, gb);
          }
          for (int g = 0; g < ngroups; g++)
            { int64_t sum = 0;
              for (int b = gb[g]; b < gb[g + 1]; b++)
                sum += est[b];
              gmax = std::max(gmax, sum);
            }
          ctx->acc_ns_total = gcap_all;
          // Entry replay: the first pass also records the 4-byte entries (start, flip, length, bucket) of the
          // super-mers it does not emit; the later passes rebuild their records from those and the reads and
          // skip the minimizer computation (72 % of a pass).  Falls back to full passes when the entries do
          // not fit or cannot be allocated.
          bool replay = (ctx->dbg_no_replay == 0);
          if (replay)
            { int64_t first = 0;
              for (int b = gb[0]; b < gb[1]; b++)
                first += est[b];
              const int64_t ntiles = (nbytes - ctx->prm.kmer + 1 + 4095) / 4096;      // (packed or not: nbytes = positions)
              ctx->ent_cap = ((gcap_all - first) + (gcap_all - first) / 16 + 64 * 8192) / 64 * 64;
              const int64_t nchunks = (ntiles + 15) / 16;
              if (fk_slot(ctx, FK_SLOT_ENT, ctx->ent_cap * 4) == NULL || fk_slot(ctx, FK_SLOT_TENT, ntiles * 8) == NULL
                  || fk_slot(ctx, FK_SLOT_TCNT, ntiles * nbk * 2) == NULL
                  || fk_slot(ctx, FK_SLOT_CBASE, nchunks * nbk * 12 + FK_CBASE_EXTRA) == NULL)
                { replay = false;
                  ctx->err[0] = 0;
                }
            }
          hipEvent_t gev[2];
          if (hipEventCreate(&gev[0]) != hipSuccess || hipEventCreate(&gev[1]) != hipSuccess)
            { rc = FK_EHIP; break; }
          for (int g = 0; g < ngroups && rc == FK_OK; g++)
            { int64_t lo[257], cnt[256], nig = 0;
              double  grow = 1.0;
              for (int tries = 0; ; tries++)
                { int64_t run = 0;
                  for (int b = 0; b < nbk; b++)
                    { lo[b] = run;
                      if (b >= gb[g] && b < gb[g + 1])
                        run += (replay && g > 0) ? ctx->ent_totals[b]            // exact: the recording pass counted them
                                                 : (int64_t) ((double) est[b] * grow);
                    }
                  lo[nbk] = run;
                  void *out = fk_slot(ctx, FK_SLOT_SM_A, std::max(run, gmax) * w.smer_stride);
                  if (out == NULL) { rc = FK_ENOMEM; break; }
                  sm_a = out;
                  // (the passes write the first digit stream of every bucket's grouping sort beside the records)
                  sm_dig = (w.smer_stride == 20) ? (uint8_t *) fk_slot(ctx, FK_SLOT_SM_DIG, ctx->slot_cap[FK_SLOT_SM_A] / w.smer_stride + 64)
                                                 : NULL;
                  hipEventRecord(gev[0], s);
                  rc = fkx_split_planned(ctx, d_reads, nbytes, out, ctx->slot_cap[FK_SLOT_SM_A] / w.smer_stride,
                                         lo, cnt, &nig, gb[g], gb[g + 1], replay ? (g == 0 ? 1 : 2) : 0, pk, sm_dig);
                  if (replay && g == 0 && rc == FK_OK && !ctx->ent_valid)
                    replay = false;                 // the entries did not fit: full passes for the other groups
                  hipEventRecord(gev[1], s);
                  hipEventSynchronize(gev[1]);
                  ms_split_groups += ms_between(gev[0], gev[1]);
                  if (rc != FK_ESTATE || tries >= 3)
                    break;
                  if (tries < 2)
                    grow *= 1.5;              // the sample under-estimated a bucket: wider regions
                  else
                    { // very uneven input: count the buckets exactly (one more pass over the reads) instead of failing
                      int64_t ns_x = 0, ni_x = 0, bcx[256];
                      if ((rc = fkx_split(ctx, d_reads, nbytes, NULL, 0, &ns_x, &ni_x, bcx, false, NULL, pk)) != FK_OK)
                        break;
                      for (int b = 0; b < nbk; b++)
                        est[b] = bcx[b] + FK_REGION_SLACK;
                      grow = 1.0;
                    }
                }
              if (rc != FK_OK)
                break;
              if (g == 0)
                res->ninst = nig;
              res->replay_passes += (replay && g > 0) ? 1 : 0;
              for (int b = gb[g]; b < gb[g + 1] && rc == FK_OK; b++)
                { tot += cnt[b];
                  rc = count_bucket(ctx, (char *) sm_a + lo[b] * w.smer_stride, cnt[b], res, false, NULL,
                                    &ntab, NULL, &tm, ns_max, sm_dig != NULL ? sm_dig + lo[b] : NULL);
                }
            }
          hipEventDestroy(gev[0]);
          hipEventDestroy(gev[1]);
          if (rc != FK_OK)
            break;
          res->nsuper = tot;
          if (ntab > 0 && (rc = sort_union_table(ctx, ntab, res, &table, &tm)) != FK_OK)
            break;
        }
      else if (nbk == 1)
        { if (chunked && ns > 0)
            rc = gather(0, &sm_in);
          if (rc == FK_OK)
            rc = count_bucket(ctx, sm_in, ns, res, true, &table, &ntab, h_roff, &tm, 0, chunked ? NULL : sm_dig);
        }
      else
        { const bool tim = (getenv("FK_FINISH_TIMING") != NULL);
          double t_g = 0., t_c = 0., t_s = 0.;
          for (int b = 0; b < nbk && rc == FK_OK; b++)
            { void *p = (char *) sm_in + bo[b] * w.smer_stride;
              const double w0 = tim ? fk_wall() : 0.;
              if (chunked && bc[b] > 0)
                rc = gather(b, &p);
              if (tim) { hipStreamSynchronize(s); t_g += fk_wall() - w0; }
              const double w1 = tim ? fk_wall() : 0.;
              if (rc == FK_OK)
                rc = count_bucket(ctx, p, bc[b], res, false, NULL, &ntab, (h_roff != NULL && b == 0) ? h_roff : NULL, &tm,
                                  ns_max, (chunked || sm_dig == NULL) ? NULL : sm_dig + bo[b]);
              if (tim) t_c += fk_wall() - w1;
            }
          const double w2 = tim ? fk_wall() : 0.;
          if (rc == FK_OK && ntab > 0 && (rc = sort_union_table(ctx, ntab, res, &table, &tm)) != FK_OK)
            break;
          if (h_roff != NULL)                  // exact_parts: Table_Split goes by BUCKET 0's weighted k-mers (count.c:1560-1565)
            for (int x = 0; x < 256; x++)
              res->wfirst[x] = ctx->exact_wfirst[x];
          if (tim)
            { t_s = fk_wall() - w2;
              fprintf(stderr, "  finish timing: gather %.3f s, count %.3f s, table sort %.3f s\n", t_g, t_c, t_s);
            }
        }
      if (rc != FK_OK)
        break;
      { const double w3 = fk_wall();
        if ((rc = fetch_result_table(ctx, res, table, ntab, fetch_table)) != FK_OK)
          break;
        if (getenv("FK_FINISH_TIMING") != NULL)
          fprintf(stderr, "  finish timing: table fetch %.3f s\n", fk_wall() - w3);
      }
      // fk_make_profiles looks k-mers up in this table: it has to hold every k-mer of resident reads
      ctx->have_table = (d_smers_in == NULL && ctx->prm.table_cutoff == 1);
      ctx->have_part_table = (ctx->prm.table_cutoff == 1);
      ctx->last_table = table;
      ctx->pf_dict_table = NULL;
      ctx->last_ntab  = ntab;
      hipEventRecord(ev[2], s);
      if (hipStreamSynchronize(s) != hipSuccess) { rc = FK_EHIP; break; }
      res->ms_split      = ms_between(ev[0], ev[1]) + ms_split_groups;
      res->split_passes  = ngroups;
      res->spilled_bytes = chunked ? ctx->spilled_bytes : 0;
      res->ms_sort_super = tm.group_s;
      res->ms_expand     = tm.expand;
      res->ms_sort_kmer  = tm.radix_k;                  // radix passes (grouping + table sort)
      res->ms_count      = tm.aggr;
      res->ms_total      = ms_between(ev[0], ev[2]);
    }
  while (0);
  for (int i = 0; i < 3; i++)
    hipEventDestroy(ev[i]);
  if (rc == FK_EHIP && ctx->err[0] == 0)
    fk_set_error(ctx, "fk_finish: HIP failure: %s", hipGetErrorString(hipGetLastError()));
  return (rc);
}

static int finish_impl(fk_ctx *ctx, fk_result *res, bool fetch)
{ if (ctx == NULL || res == NULL) return (FK_EINVAL);
  FK_HIP(ctx, hipSetDevice(ctx->device));
  { pthread_mutex_lock((pthread_mutex_t *) ctx->push_lock);
    int rc = fkx_flush_join(ctx);
    pthread_mutex_unlock((pthread_mutex_t *) ctx->push_lock);
    if (rc != FK_OK)
      return (rc);
  }
  FK_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
  FK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->nchunks > 0)
    { // chunked ingest: the rest of the reads becomes the last chunk, then the buckets are counted
      pthread_mutex_lock((pthread_mutex_t *) ctx->push_lock);
      const double w0 = fk_wall();
      int rc = fkx_flush_chunk(ctx);
      pthread_mutex_unlock((pthread_mutex_t *) ctx->push_lock);
      if (getenv("FK_FINISH_TIMING") != NULL)
        fprintf(stderr, "  finish timing: last flush %.3f s\n", fk_wall() - w0);
      if (rc == FK_OK)
        rc = fkx_pipeline(ctx, NULL, 0, NULL, 0, res, fetch);
      for (int i = 0; i < ctx->nchunks; i++)
        fkx_free_chunk(ctx, &ctx->chunks[i]);
      ctx->nchunks = 0;
      fkx_rewind_slabs(ctx);
      ctx->chunk_ninst = 0;
      return (rc);
    }
  if (ctx->prm.exact_parts)
    { if (ctx->h_roff == NULL && ctx->reads_len > 0)
        { fk_set_error(ctx, "exact_parts needs reads pushed with fk_push_block");
          return (FK_ESTATE);
        }
      if (ctx->h_roff == NULL)
        { ctx->h_roff = (int64_t *) malloc(sizeof(int64_t) * 4);
          ctx->roff_cap = 4;
        }
      const void *rd = ctx->d_reads;
      int64_t     len = ctx->reads_len;
      if (ctx->push_form == 2)                 // the reference's rule walks reads byte by byte: restore the ASCII
        { void *asc = NULL;
          const int rc = fkx_unpack_store(ctx, &asc, &len);
          if (rc != FK_OK)
            return (rc);
          rd = asc;
        }
      return fkx_pipeline(ctx, rd, len, NULL, 0, res, fetch, ctx->h_roff, ctx->nroff);
    }
  if (ctx->push_form == 2)
    { const fk_pkstore *st = &ctx->pk[ctx->pk_cur];
      fk_pkview pv;
      pv.roff = st->roff; pv.nreads = st->nreads; pv.inv = st->inv; pv.ninv = st->ninv;
      if (st->nreads == 0)
        return fkx_pipeline(ctx, ctx->d_reads, 0, NULL, 0, res, fetch);
      return fkx_pipeline(ctx, ctx->d_reads, st->npos, NULL, 0, res, fetch, NULL, 0, &pv);
    }
  return fkx_pipeline(ctx, ctx->d_reads, ctx->reads_len, NULL, 0, res, fetch);
}

extern "C" int fk_finish(fk_ctx *ctx, fk_result *res)
{ return finish_impl(ctx, res, true); }

/* fk_finish without the host copy of the table (res->table NULL, res->ntable set): the sorted table stays in HBM for
   fk_write_ktab_device / fk_make_profiles. */
extern "C" int fk_finish_device(fk_ctx *ctx, fk_result *res)
{ int rc = finish_impl(ctx, res, false);
  if (rc == FK_OK && ctx->prm.table_cutoff > 0 && ctx->last_table != NULL && ctx->last_ntab == res->ntable)
    rc = fkx_ktab_prepare(ctx, res->ntable);           // what the part writers need, made before any memory goes back
  return (rc);
}

/* Forget the reads pushed so far (and any chunks split from them); arenas, staging buffers and the
   bucket assignment stay, so the next data set starts without allocations. */
extern "C" int fk_reset(fk_ctx *ctx)
{ if (ctx == NULL) return (FK_EINVAL);
  FK_HIP(ctx, hipSetDevice(ctx->device));
  pthread_mutex_lock((pthread_mutex_t *) ctx->push_lock);
  (void) fkx_flush_join(ctx);
  hipStreamSynchronize(ctx->copy_stream);
  hipStreamSynchronize(ctx->stream);
  ctx->reads_len = 0;
  ctx->nblocks = 0;
  ctx->blocks_bad = false;
  ctx->nroff = 0;
  ctx->push_form = 0;
  ctx->pk_ascii_len = 0;
  for (int i = 0; i < 2; i++)
    ctx->pk[i].nreads = ctx->pk[i].ninv = ctx->pk[i].npos = 0;
  for (int i = 0; i < ctx->nchunks; i++)
    fkx_free_chunk(ctx, &ctx->chunks[i]);
  ctx->nchunks = 0;
  fkx_rewind_slabs(ctx);
  ctx->chunk_ninst = 0;
  pthread_mutex_unlock((pthread_mutex_t *) ctx->push_lock);
  return (FK_OK);
}

/* Same pipeline on a caller-owned device buffer, table left out unless asked (bench path). */
extern "C" int fk_count_device_reads(fk_ctx *ctx, const void *d_bases, int64_t nbytes, int fetch_table,
                                     fk_result *res)
{ if (ctx == NULL || res == NULL || d_bases == NULL) return (FK_EINVAL);
  if (((uintptr_t) d_bases & 15) != 0)
    { fk_set_error(ctx, "fk_count_device_reads: read buffer must be 16-byte aligned");
      return (FK_EINVAL);
    }
  FK_HIP(ctx, hipSetDevice(ctx->device));
  return fkx_pipeline(ctx, d_bases, nbytes, NULL, 0, res, fetch_table != 0);
}

/* fk_count_device_reads for reads that are resident in TWO BITS PER BASE and stay owned by the caller (see
   include/fastk_amd.h): nothing is unpacked, the splitter's tile loader reads the codes. */
extern "C" int fk_count_device_packed(fk_ctx *ctx, const void *d_codes, int64_t nbases, const int64_t *d_roff, int64_t nreads,
                                      const int64_t *d_inv, int64_t ninv, int fetch_table, fk_result *res)
{ if (ctx == NULL || res == NULL || d_codes == NULL || d_roff == NULL || nreads <= 0 || nbases < 0 || ninv < 0
      || (ninv > 0 && d_inv == NULL))
    return (FK_EINVAL);
  if (((uintptr_t) d_codes & 3) != 0)
    { fk_set_error(ctx, "fk_count_device_packed: the codes must be 4-byte aligned");
      return (FK_EINVAL);
    }
  if (ctx->prm.exact_parts)
    { fk_set_error(ctx, "fk_count_device_packed: exact_parts needs reads pushed with fk_push_block / fk_push_packed");
      return (FK_EUNSUPPORTED);
    }
  FK_HIP(ctx, hipSetDevice(ctx->device));
  fk_pkview pv;
  pv.roff = d_roff; pv.nreads = nreads; pv.inv = d_inv; pv.ninv = ninv;
  return fkx_pipeline(ctx, d_codes, nbases, NULL, 0, res, fetch_table != 0, NULL, 0, &pv);
}

/* Sort + expand + sort + count over super-mer records that are already in HBM (the records a
   rank owns after the bucket exchange).  d_smers is clobbered. */
extern "C" int fk_count_device_supermers(fk_ctx *ctx, void *d_smers, int64_t nsuper, int fetch_table,
                                         fk_result *res)
{ if (ctx == NULL || res == NULL || nsuper < 0 || (d_smers == NULL && nsuper > 0)) return (FK_EINVAL);
  static char dummy[16];
  FK_HIP(ctx, hipSetDevice(ctx->device));
  return fkx_pipeline(ctx, NULL, 0, nsuper > 0 ? d_smers : (void *) dummy, nsuper, res,
                      fetch_table != 0);
}

/* Rounds: the records a rank owns may arrive in several pieces (one per exchange round, so that the
   exchange of piece i+1 overlaps the counting of piece i).  Every piece must be closed under k-mer
   identity (whole minimizer buckets).  begin -> add (once per piece; d_smers is clobbered) -> finish:
   histogram, totals and the table over all pieces, exactly as if they had been counted together. */
extern "C" int fk_rounds_begin(fk_ctx *ctx)
{ if (ctx == NULL) return (FK_EINVAL);
  if (ctx->acc_res == NULL && (ctx->acc_res = (fk_result *) malloc(sizeof(fk_result))) == NULL)
    return (FK_ENOMEM);
  memset(ctx->acc_res, 0, sizeof(fk_result));
  ctx->acc_ntab = 0;
  ctx->acc_ns = 0;
  ctx->acc_ns_total = 0;
  ctx->acc_tm[0] = ctx->acc_tm[1] = ctx->acc_tm[2] = ctx->acc_tm[3] = 0.;
  return (FK_OK);
}

extern "C" int fk_rounds_add(fk_ctx *ctx, void *d_smers, int64_t nsuper)
{ if (ctx == NULL || ctx->acc_res == NULL || nsuper < 0 || (d_smers == NULL && nsuper > 0)) return (FK_EINVAL);
  FK_HIP(ctx, hipSetDevice(ctx->device));
  fk_stage_ms tm = { 0., 0., 0., 0. };
  ctx->acc_res->nsuper += nsuper;
  int rc = count_bucket(ctx, d_smers, nsuper, ctx->acc_res, false, NULL, &ctx->acc_ntab, NULL, &tm);
  ctx->acc_tm[0] += tm.group_s; ctx->acc_tm[1] += tm.expand; ctx->acc_tm[2] += tm.radix_k; ctx->acc_tm[3] += tm.aggr;
  return (rc);
}

extern "C" int fk_rounds_finish(fk_ctx *ctx, int fetch_table, fk_result *res)
{ if (ctx == NULL || ctx->acc_res == NULL || res == NULL) return (FK_EINVAL);
  FK_HIP(ctx, hipSetDevice(ctx->device));
  *res = *ctx->acc_res;
  fk_stage_ms tm = { ctx->acc_tm[0], ctx->acc_tm[1], ctx->acc_tm[2], ctx->acc_tm[3] };
  void *table = NULL;
  int rc = FK_OK;
  if (ctx->acc_ntab > 0 && (rc = sort_union_table(ctx, ctx->acc_ntab, res, &table, &tm)) != FK_OK)
    return (rc);
  if ((rc = fetch_result_table(ctx, res, table, ctx->acc_ntab, fetch_table != 0)) != FK_OK)
    return (rc);
  ctx->have_table = false;                       // a rank's pieces: not the whole data set
  ctx->have_part_table = (ctx->prm.table_cutoff == 1);
  ctx->last_table = table;
  ctx->pf_dict_table = NULL;
  ctx->last_ntab  = ctx->acc_ntab;
  res->ms_sort_super = tm.group_s;
  res->ms_expand     = tm.expand;
  res->ms_sort_kmer  = tm.radix_k;
  res->ms_count      = tm.aggr;
  res->ms_total      = tm.group_s + tm.expand + tm.radix_k + tm.aggr;
  return (FK_OK);
}
