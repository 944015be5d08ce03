// fk_shard.hip -- the sharded run from a C host: one process per GPU, RCCL called directly.
//
// What the reference does with files between its phases, a node's GPUs do over xGMI:
//   C1  every rank splits the reads it was handed into super-mers grouped by minimizer bucket; bucket
//       r*world + d goes to rank d in round r: grouped ncclSend/ncclRecv of the SMER_WORD records
//       replaces the ".T" file shuffle (split.c:1263 <-> count.c:1347); round r+1 travels on its own
//       stream while round r is counted;
//   C2  the 0x8000-bin histogram, the totals and the first-byte census are all-reduced
//       (count.c:1543-1553);
//   C3  a second exchange keyed by the first k-mer byte gives rank r the table entries of a contiguous
//       first-byte range -- the ranges Table_Split picks for nparts parts (count.c:1560-1565) -- and
//       every rank writes the hidden part files of its range itself (table.c:346-533: parts are ordered
//       ranges of the table, README.md:988); the per-prefix entry counts are reduced to rank 0, which
//       writes the stub and the .hist file.
// Everything that travels stays in HBM; only the sorted part payloads cross PCIe (pinned) to be written.
//
// RCCL is bound at run time (dlopen of librccl.so.1): the library has no link-time dependency on it, a
// process that never shards never loads it, and inside a PyTorch process the copy torch already
// loaded is the one that gets used.
#include "fk_common.h"

#include <dlfcn.h>
#include <rccl/rccl.h>
#include <vector>
#include <algorithm>

struct fk_rccl
{ ncclResult_t (*GetUniqueId)(ncclUniqueId *);
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
  ncclResult_t (*CommDestroy)(ncclComm_t);
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
  ncclResult_t (*GroupStart)();
  ncclResult_t (*GroupEnd)();
  ncclResult_t (*CommCount)(const ncclComm_t, int *);
  const char  *(*GetErrorString)(ncclResult_t);
};

static fk_rccl g_rccl;
static bool    g_rccl_ok = false;

static int load_rccl(fk_ctx *ctx)
{ if (g_rccl_ok)
    return (FK_OK);
  void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (h == NULL) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (h == NULL) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (h == NULL)
    { fk_set_error(ctx, "sharded run: cannot load librccl.so.1 (%s)", dlerror());
      return (FK_EUNSUPPORTED);
    }
#define SYM(field, name) \
  if ((*(void **) &g_rccl.field = dlsym(h, name)) == NULL)                                   \
    { fk_set_error(ctx, "sharded run: librccl has no %s", name); return (FK_EUNSUPPORTED); }
  SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommDestroy, "ncclCommDestroy")
  SYM(Send, "ncclSend") SYM(Recv, "ncclRecv") SYM(AllReduce, "ncclAllReduce") SYM(AllGather, "ncclAllGather")
  SYM(GroupStart, "ncclGroupStart") SYM(GroupEnd, "ncclGroupEnd") SYM(GetErrorString, "ncclGetErrorString")
  SYM(CommCount, "ncclCommCount")
#undef SYM
  g_rccl_ok = true;
  return (FK_OK);
}

#define FK_NCCL(ctx, call)                                                                      \
  do { ncclResult_t r_ = (call);                                                                 \
       if (r_ != ncclSuccess)                                                                    \
         { fk_set_error(ctx, "%s failed: %s (%s:%d)", #call, g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
           return (FK_EHIP);                                                                     \
         }                                                                                       \
     } while (0)

#define FK_PAIR_BYTES (1ll << 30)      // one ncclSend/ncclRecv moves at most this much

struct fk_shard
{ fk_ctx     *ctx;
  int         rank, world, rounds;
  ncclComm_t  comm;
  hipStream_t xs;                 // the exchange runs here, counting on ctx->stream
  hipEvent_t  xev[2];
  int64_t    *d_small, *h_small;  // counts, histogram and totals on their way through the collectives
  int64_t     small_cap;          // in int64
  void       *inbox[2];
  int64_t     inbox_cap[2];
  fk_result   local;              // this rank's own result (its table stays in HBM, in ctx)
  int64_t     lfirst[256];        // first-byte census of this rank's table
  // C3 (fk_shard_gather): this rank's range of the whole table, ordered, in pinned host memory; kept for the next run
  char       *g_dev[2];
  int64_t     g_dev_cap;          // bytes each
  uint8_t    *g_host;
  int64_t     g_host_cap;
  int64_t     g_n;                // entries gathered (-1: none)
  int         g_nparts, g_ib;
  int         g_split[257];
  int         write_cutoff;       // > table_cutoff of the context: entries below it do not leave this rank (fk_shard_set_write_cutoff)
  int64_t     g_total;            // entries of the whole table that C3 moved (all ranks; = res->ntable without a write cutoff)
  // fk_shard_profiles: records + positions of the piece, counts made for the others, counts that came back
  void       *pf_buf[4];
  int64_t     pf_cap[4];
  // what the last fk_shard_count / fk_shard_gather moved (fk_shard_get_stats)
  fk_shard_stats st;
  hipEvent_t  tev[2 * 256];       // begin / end of every exchange round on xs (timing events, made on first use)
};

// all ranks learn whether any of them failed: every rank passes its own rc and gets the first failure (its own,
// or FK_EHIP for a peer's) -- so that a rank that ran out of memory does not leave the others in a collective
static int allreduce_i64(fk_shard *sh, int64_t *vals, int n);
static void *pf_reserve(fk_shard *sh, int i, int64_t bytes);
static int agree(fk_shard *sh, int rc)
{ int64_t bad = (rc != FK_OK) ? 1 : 0;
  const int rc2 = allreduce_i64(sh, &bad, 1);
  if (rc != FK_OK) return (rc);
  if (rc2 != FK_OK) return (rc2);
  if (bad != 0)
    { fk_set_error(sh->ctx, "sharded run: another rank failed (rank %d stops with it)", sh->rank);
      return (FK_EHIP);
    }
  return (FK_OK);
}

extern "C" int fk_shard_unique_id(char *id128)
{ if (id128 == NULL) return (FK_EINVAL);
  int rc = load_rccl(NULL);
  if (rc != FK_OK) return (rc);
  ncclUniqueId id;
  if (g_rccl.GetUniqueId(&id) != ncclSuccess)
    { fk_set_error(NULL, "ncclGetUniqueId failed");
      return (FK_EHIP);
    }
  static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
  memcpy(id128, &id, 128);
  return (FK_OK);
}

extern "C" void fk_shard_destroy(fk_shard *sh)
{ if (sh == NULL) return;
  hipSetDevice(sh->ctx->device);
  if (sh->xs) hipStreamSynchronize(sh->xs);
  if (sh->ctx->stream) hipStreamSynchronize(sh->ctx->stream);      // (the gather's D2H into g_host runs on the context's stream)
  if (sh->comm) g_rccl.CommDestroy(sh->comm);
  for (int i = 0; i < 2; i++)
    { fkx_event_put(sh->ctx->device, false, &sh->xev[i]);
      if (sh->inbox[i]) hipFree(sh->inbox[i]);
    }
  fkx_stream_put(sh->ctx->device, sh->xs);
  if (sh->d_small) hipFree(sh->d_small);
  if (sh->h_small) hipHostFree(sh->h_small);
  for (int i = 0; i < 2; i++)
    if (sh->g_dev[i]) hipFree(sh->g_dev[i]);
  if (sh->g_host) fkx_pinned_free(sh->g_host);
  for (int i = 0; i < 2 * 256; i++)
    fkx_event_put(sh->ctx->device, true, &sh->tev[i]);
  for (int i = 0; i < 4; i++)
    if (sh->pf_buf[i]) hipFree(sh->pf_buf[i]);
  free(sh);
}

/* element-wise sum of n int64 over all ranks, in place (host memory): what a C host needs to agree on totals */
extern "C" int fk_shard_sum_i64(fk_shard *sh, int64_t *vals, int n)
{ if (sh == NULL || vals == NULL || n < 1) return (FK_EINVAL);
  FK_HIP(sh->ctx, hipSetDevice(sh->ctx->device));
  return allreduce_i64(sh, vals, n);
}

extern "C" int fk_shard_get_stats(fk_shard *sh, fk_shard_stats *st)
{ if (sh == NULL || st == NULL) return (FK_EINVAL);
  *st = sh->st;
  st->rounds = sh->rounds;
  int n = 0;
  if (sh->comm != NULL && g_rccl.CommCount(sh->comm, &n) == ncclSuccess)
    st->comm_ranks = n;
  return (FK_OK);
}

extern "C" int fk_shard_create(fk_ctx *ctx, int rank, int world, const char *id128, fk_shard **out)
{ if (ctx == NULL || out == NULL || id128 == NULL || world < 1 || rank < 0 || rank >= world) return (FK_EINVAL);
  *out = NULL;
  if (ctx->prm.nbuckets % world != 0 || ctx->prm.nbuckets / world < 1)
    { fk_set_error(ctx, "fk_shard_create: the context's %d buckets are not a multiple of the %d ranks (bucket r*world+d "
                        "goes to rank d in round r)", ctx->prm.nbuckets, world);
      return (FK_EINVAL);
    }
  int rc = load_rccl(ctx);
  if (rc != FK_OK) return (rc);
  fk_shard *sh = (fk_shard *) calloc(1, sizeof(fk_shard));
  if (sh == NULL) return (FK_ENOMEM);
  sh->ctx = ctx; sh->rank = rank; sh->world = world; sh->rounds = ctx->prm.nbuckets / world;
  FK_HIP(ctx, hipSetDevice(ctx->device));
  ncclUniqueId id;
  memcpy(&id, id128, 128);
  { ncclResult_t r = g_rccl.CommInitRank(&sh->comm, world, id, rank);
    if (r != ncclSuccess)
      { fk_set_error(ctx, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, g_rccl.GetErrorString(r));
        free(sh);
        return (FK_EHIP);
      }
  }
  sh->small_cap = (int64_t) world * ctx->prm.nbuckets + 2 * (FK_HIST_BINS + 1024);
  if (fkx_stream_get(ctx->device, &sh->xs) != FK_OK
      || fkx_event_get(ctx->device, false, &sh->xev[0]) != FK_OK
      || fkx_event_get(ctx->device, false, &sh->xev[1]) != FK_OK
      || hipMalloc((void **) &sh->d_small, (size_t) sh->small_cap * 8) != hipSuccess
      || hipHostMalloc((void **) &sh->h_small, (size_t) sh->small_cap * 8, hipHostMallocDefault) != hipSuccess)
    { fk_set_error(ctx, "fk_shard_create: cannot set up streams and buffers");
      fk_shard_destroy(sh);
      return (FK_EHIP);
    }
  *out = sh;
  return (FK_OK);
}

// every rank contributes n int64; all[r*n .. r*n+n) = rank r's.  Host in, host out.
static int allgather_i64(fk_shard *sh, const int64_t *mine, int n, int64_t *all)
{ fk_ctx *ctx = sh->ctx;
  int64_t *d_in = sh->d_small, *d_out = sh->d_small + n;
  if ((int64_t) n * (sh->world + 1) > sh->small_cap) return (FK_EINVAL);
  memcpy(sh->h_small, mine, (size_t) n * 8);
  FK_HIP(ctx, hipMemcpyAsync(d_in, sh->h_small, (size_t) n * 8, hipMemcpyHostToDevice, sh->xs));
  FK_NCCL(ctx, g_rccl.AllGather(d_in, d_out, (size_t) n, ncclInt64, sh->comm, sh->xs));
  FK_HIP(ctx, hipMemcpyAsync(sh->h_small, d_out, (size_t) n * sh->world * 8, hipMemcpyDeviceToHost, sh->xs));
  FK_HIP(ctx, hipStreamSynchronize(sh->xs));
  memcpy(all, sh->h_small, (size_t) n * sh->world * 8);
  return (FK_OK);
}

static int allreduce_i64(fk_shard *sh, int64_t *vals, int n)
{ fk_ctx *ctx = sh->ctx;
  if (n > sh->small_cap) return (FK_EINVAL);
  memcpy(sh->h_small, vals, (size_t) n * 8);
  FK_HIP(ctx, hipMemcpyAsync(sh->d_small, sh->h_small, (size_t) n * 8, hipMemcpyHostToDevice, sh->xs));
  FK_NCCL(ctx, g_rccl.AllReduce(sh->d_small, sh->d_small, (size_t) n, ncclInt64, ncclSum, sh->comm, sh->xs));
  FK_HIP(ctx, hipMemcpyAsync(sh->h_small, sh->d_small, (size_t) n * 8, hipMemcpyDeviceToHost, sh->xs));
  FK_HIP(ctx, hipStreamSynchronize(sh->xs));
  memcpy(vals, sh->h_small, (size_t) n * 8);
  return (FK_OK);
}

// Rank `me` sends send_bytes[d] from send_ptr[d] to every rank d and receives recv_bytes[s] at recv_ptr[s]
// from every rank s, in grouped calls of at most FK_PAIR_BYTES per pair (the local share is a device copy).
static int exchange(fk_shard *sh, char *const *send_ptr, const int64_t *send_bytes, char *const *recv_ptr,
                    const int64_t *recv_bytes)
{ fk_ctx *ctx = sh->ctx;
  const int W = sh->world, me = sh->rank;
  if (send_bytes[me] != recv_bytes[me])
    { fk_set_error(ctx, "exchange: rank %d keeps %lld bytes but expects %lld", me, (long long) send_bytes[me],
                   (long long) recv_bytes[me]);
      return (FK_EHIP);
    }
  if (send_bytes[me] > 0)
    FK_HIP(ctx, hipMemcpyAsync(recv_ptr[me], send_ptr[me], (size_t) send_bytes[me], hipMemcpyDeviceToDevice, sh->xs));
  int64_t most = 0;
  for (int p = 0; p < W; p++)
    if (p != me)
      most = std::max(most, std::max(send_bytes[p], recv_bytes[p]));
  for (int64_t o = 0; o < most; o += FK_PAIR_BYTES)
    { FK_NCCL(ctx, g_rccl.GroupStart());
      for (int p = 0; p < W; p++)
        { if (p == me) continue;
          if (o < send_bytes[p])
            FK_NCCL(ctx, g_rccl.Send(send_ptr[p] + o, (size_t) std::min<int64_t>(FK_PAIR_BYTES, send_bytes[p] - o),
                                     ncclUint8, p, sh->comm, sh->xs));
          if (o < recv_bytes[p])
            FK_NCCL(ctx, g_rccl.Recv(recv_ptr[p] + o, (size_t) std::min<int64_t>(FK_PAIR_BYTES, recv_bytes[p] - o),
                                     ncclUint8, p, sh->comm, sh->xs));
        }
      FK_NCCL(ctx, g_rccl.GroupEnd());
    }
  return (FK_OK);
}

static int reserve_inbox(fk_shard *sh, int i, int64_t bytes)
{ if (sh->inbox_cap[i] >= bytes)
    return (FK_OK);
  if (sh->inbox[i] != NULL)
    hipFree(sh->inbox[i]);
  sh->inbox[i] = NULL;
  sh->inbox_cap[i] = 0;
  const int64_t want = bytes + bytes / 16 + (1 << 20);
  if (hipMalloc(&sh->inbox[i], (size_t) want) != hipSuccess)
    { fk_set_error(sh->ctx, "out of HBM: cannot allocate %lld bytes for the records of an exchange round", (long long) want);
      return (FK_ENOMEM);
    }
  sh->inbox_cap[i] = want;
  return (FK_OK);
}

/* C1 + per-rank counting + C2 over the reads pushed into the context (they stay resident: hbm_budget 0).
   res: the GLOBAL histogram, max_inst and totals (identical on every rank), wfirst = first-byte census of
   the whole table, ntable = its entries; res->table is NULL -- every rank's share stays in HBM for
   fk_shard_write.  Returns FK_EHIP with a message if the exchange did not conserve records or k-mers. */
static int shard_count(fk_shard *sh, const void *d_reads, int64_t reads_len, fk_result *res, const fk_pkview *pk = NULL,
                       bool chunked = false);

extern "C" int fk_shard_count(fk_shard *sh, fk_result *res)
{ if (sh == NULL || res == NULL) return (FK_EINVAL);
  fk_ctx *ctx = sh->ctx;
  FK_HIP(ctx, hipSetDevice(ctx->device));
  if (ctx->prm.exact_parts)
    { fk_set_error(ctx, "fk_shard_count: exact_parts runs on one GPU");
      return (FK_EUNSUPPORTED);
    }
  if (ctx->chunk_bytes > 0)
    { // hbm_budget > 0 (FastK_amd -G<n> -M<GB>): the stripe was split chunk by chunk as it was pushed and its
      // records wait in the slab store / in pinned host memory; the rest of the reads becomes the last chunk
      pthread_mutex_lock((pthread_mutex_t *) ctx->push_lock);
      int rc = fkx_flush_join(ctx);
      if (rc == FK_OK)
        rc = fkx_flush_chunk(ctx);
      pthread_mutex_unlock((pthread_mutex_t *) ctx->push_lock);
      if (rc != FK_OK)
        return (rc);
      FK_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
      FK_HIP(ctx, hipStreamSynchronize(ctx->stream));
      rc = shard_count(sh, NULL, 0, res, NULL, true);
      for (int i = 0; i < ctx->nchunks; i++)
        fkx_free_chunk(ctx, &ctx->chunks[i]);
      ctx->nchunks = 0;
      fkx_rewind_slabs(ctx);
      ctx->chunk_ninst = 0;
      return (rc);
    }
  FK_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
  FK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->push_form == 2)                         // the stripe came through fk_push_packed: split as it is
    { const fk_pkstore *st = &ctx->pk[ctx->pk_cur];
      fk_pkview pv;
      pv.roff = st->roff; pv.nreads = st->nreads; pv.inv = st->inv; pv.ninv = st->ninv;
      if (st->nreads > 0)
        return shard_count(sh, ctx->d_reads, st->npos, res, &pv);
      return shard_count(sh, ctx->d_reads, 0, res);
    }
  return shard_count(sh, ctx->d_reads, ctx->reads_len, res);
}

/* This rank's own share of the last fk_shard_count: its record / k-mer counts and the per-kernel timings
   (ms_scatter_kmer, launches_kmer ...) that the roofline accounting needs; table is NULL. */
extern "C" int fk_shard_local_result(fk_shard *sh, fk_result *res)
{ if (sh == NULL || res == NULL) return (FK_EINVAL);
  *res = sh->local;
  res->table = NULL;
  return (FK_OK);
}

/* The same over reads that are already resident in HBM and stay owned by the caller (16-byte aligned; any
   byte that is not acgtACGT separates reads), like fk_count_device_reads. */
extern "C" int fk_shard_count_device(fk_shard *sh, const void *d_bases, int64_t nbytes, fk_result *res)
{ if (sh == NULL || res == NULL || nbytes < 0 || (d_bases == NULL && nbytes > 0)) return (FK_EINVAL);
  fk_ctx *ctx = sh->ctx;
  if (((uintptr_t) d_bases & 15) != 0)
    { fk_set_error(ctx, "fk_shard_count_device: read buffer must be 16-byte aligned");
      return (FK_EINVAL);
    }
  FK_HIP(ctx, hipSetDevice(ctx->device));
  FK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return shard_count(sh, d_bases, nbytes, res);
}

/* ... and over reads resident in two bits per base (the arguments of fk_count_device_packed): the rank's stripe is
   split as it is, a quarter of the bytes of the ASCII form beside the records of the exchange. */
extern "C" int fk_shard_count_device_packed(fk_shard *sh, const void *d_codes, int64_t nbases, const int64_t *d_roff,
                                            int64_t nreads, const int64_t *d_inv, int64_t ninv, fk_result *res)
{ if (sh == NULL || res == NULL || nbases < 0 || nreads < 0 || ninv < 0 || (nreads > 0 && (d_codes == NULL || d_roff == NULL))
      || (ninv > 0 && d_inv == NULL))
    return (FK_EINVAL);
  fk_ctx *ctx = sh->ctx;
  if (((uintptr_t) d_codes & 3) != 0)
    { fk_set_error(ctx, "fk_shard_count_device_packed: the codes must be 4-byte aligned");
      return (FK_EINVAL);
    }
  FK_HIP(ctx, hipSetDevice(ctx->device));
  FK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (nreads == 0)
    return shard_count(sh, d_codes, 0, res);
  fk_pkview pv;
  pv.roff = d_roff; pv.nreads = nreads; pv.inv = d_inv; pv.ninv = ninv;
  return shard_count(sh, d_codes, nbases, res, &pv);
}

// chunked: the records are the chunks of the budgeted ingest (fk_ingest.hip) -- per bucket one run per chunk, in the
// slab store or spilled to pinned host memory; a round's buckets are gathered into one of two round buffers right
// before they travel (the role of the gather in front of a bucket's count in the one-GPU budgeted run).
static int shard_count(fk_shard *sh, const void *d_reads, int64_t reads_len, fk_result *res, const fk_pkview *pk, bool chunked)
{ fk_ctx *ctx = sh->ctx;
  const int W = sh->world, R = sh->rounds, me = sh->rank, nb = ctx->prm.nbuckets;
  const int stride = ctx->wid.smer_stride;

  // ---- split this rank's reads into bucketed super-mers (planned regions, exact pair on overflow)
  int64_t cap = 0, offs[257], cnt[256], ninst = 0;
  void   *outbox = NULL;
  char   *rbox[2] = { NULL, NULL };              // chunked: the round buffers
  memset(cnt, 0, sizeof(cnt));
  memset(offs, 0, sizeof(offs));
  int rc = FK_OK;
  if (chunked)
    { int64_t rmax = 1;
      for (int b = 0; b < nb; b++)
        for (int c = 0; c < ctx->nchunks; c++)
          cnt[b] += ctx->chunks[c].cnt[b];
      ninst = ctx->chunk_ninst;
      for (int r = 0; r < R; r++)
        { int64_t sum = 0;
          for (int p = 0; p < W; p++) sum += cnt[r * W + p];
          rmax = std::max(rmax, sum);
        }
      if ((rbox[0] = (char *) fk_slot(ctx, FK_SLOT_SM_A, rmax * stride)) == NULL
          || (R > 1 && (rbox[1] = (char *) fk_slot(ctx, FK_SLOT_SM_G, rmax * stride)) == NULL))
        return (FK_ENOMEM);
      // (offs[] are relative to the round buffer of the bucket's round)
      for (int r = 0; r < R; r++)
        { int64_t run = 0;
          for (int p = 0; p < W; p++) { offs[r * W + p] = run; run += cnt[r * W + p]; }
        }
    }
  else
  {
  rc = fkx_split_plan(ctx, d_reads, reads_len, &cap, offs, pk);
  if (rc != FK_OK) return (rc);
  if (cap > 0)
    { if ((outbox = fk_slot(ctx, FK_SLOT_SM_A, cap * stride)) == NULL) return (FK_ENOMEM);
      rc = fkx_split_planned(ctx, d_reads, reads_len, outbox, cap, offs, cnt, &ninst, 0, -1, 0, pk);
      if (rc == FK_ESTATE)
        { int64_t ns = 0;
          if ((rc = fkx_split(ctx, d_reads, reads_len, NULL, 0, &ns, &ninst, cnt, false, NULL, pk)) != FK_OK) return (rc);
          if ((outbox = fk_slot(ctx, FK_SLOT_SM_A, std::max<int64_t>(ns, 1) * stride)) == NULL) return (FK_ENOMEM);
          if (ns > 0 && (rc = fkx_split(ctx, d_reads, reads_len, outbox, ns, &ns, &ninst, cnt, true, NULL, pk)) != FK_OK)
            return (rc);
          int64_t run = 0;
          for (int b = 0; b < nb; b++) { offs[b] = run; run += cnt[b]; }
        }
      else if (rc != FK_OK)
        return (rc);
    }
  }

  // ---- who sends how much to whom: all[s*nb + b] = records of bucket b at rank s
  std::vector<int64_t> all((size_t) W * nb);
  if ((rc = allgather_i64(sh, cnt, nb, all.data())) != FK_OK) return (rc);
  int64_t sent = 0, expect = 0, inmax = 0;
  for (int b = 0; b < nb; b++) sent += cnt[b];
  for (int r = 0; r < R; r++)
    { int64_t in = 0;
      for (int s = 0; s < W; s++) in += all[(size_t) s * nb + r * W + me];
      inmax = std::max(inmax, in);
      expect += in;
    }
  for (int i = 0; i < (R > 1 ? 2 : 1); i++)
    if ((rc = reserve_inbox(sh, i, std::max<int64_t>(inmax, 1) * stride)) != FK_OK) return (rc);

  std::vector<int64_t> nin(R);
  auto post = [&](int r) -> int
    { char   *sp[256], *rp[256];
      int64_t sb[256], rb[256], run = 0;
      if (chunked)
        { // the round's buckets out of the chunk store, on the exchange stream in front of the sends (round buffer
          // r & 1 was last read by the sends of round r - 2, which round r - 1's count waited for)
          outbox = rbox[r & 1];
          for (int p = 0; p < W; p++)
            { const int b = r * W + p;
              int64_t at = offs[b];
              for (int c = 0; c < ctx->nchunks; c++)
                { const fk_chunk *ch = &ctx->chunks[c];
                  if (ch->cnt[b] > 0)
                    FK_HIP(ctx, hipMemcpyAsync((char *) outbox + at * stride, ch->run[b], (size_t) (ch->cnt[b] * stride),
                                               ch->on_host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, sh->xs));
                  at += ch->cnt[b];
                }
            }
        }
      for (int p = 0; p < W; p++)
        { const int b = r * W + p;
          sp[p] = (char *) outbox + offs[b] * stride;
          sb[p] = cnt[b] * stride;
          rp[p] = (char *) sh->inbox[r & 1] + run * stride;
          rb[p] = all[(size_t) p * nb + r * W + me] * stride;
          run += all[(size_t) p * nb + r * W + me];
        }
      nin[r] = run;
      for (int p = 0; p < W; p++)
        { if (p == me) sh->st.kept_bytes += sb[p];
          else         { sh->st.sent_bytes += sb[p]; sh->st.recv_bytes += rb[p]; }
        }
      for (int i = 0; i < 2; i++)
        if (sh->tev[2 * r + i] == NULL && fkx_event_get(ctx->device, true, &sh->tev[2 * r + i]) != FK_OK)
          { fk_set_error(ctx, "fk_shard_count: cannot create events");
            return (FK_EHIP);
          }
      FK_HIP(ctx, hipEventRecord(sh->tev[2 * r], sh->xs));
      int e = exchange(sh, sp, sb, rp, rb);
      if (e != FK_OK) return (e);
      FK_HIP(ctx, hipEventRecord(sh->tev[2 * r + 1], sh->xs));
      FK_HIP(ctx, hipEventRecord(sh->xev[r & 1], sh->xs));
      return (FK_OK);
    };
  sh->st.sent_bytes = sh->st.recv_bytes = sh->st.kept_bytes = 0;
  sh->st.exchange_ms = 0.;

  // ---- rounds: piece r+1 travels while piece r is counted
  if ((rc = fk_rounds_begin(ctx)) != FK_OK) return (rc);
  if ((rc = post(0)) != FK_OK) return (rc);
  for (int r = 0; r < R; r++)
    { FK_HIP(ctx, hipEventSynchronize(sh->xev[r & 1]));
      if (r + 1 < R && (rc = post(r + 1)) != FK_OK) return (rc);      // inbox (r+1)&1 was consumed by round r-1
      if ((rc = fk_rounds_add(ctx, sh->inbox[r & 1], nin[r])) != FK_OK) return (rc);
    }
  if ((rc = fk_rounds_finish(ctx, 0, &sh->local)) != FK_OK) return (rc);
  FK_HIP(ctx, hipStreamSynchronize(sh->xs));
  for (int r = 0; r < R; r++)
    { float ms = 0.f;
      if (hipEventElapsedTime(&ms, sh->tev[2 * r], sh->tev[2 * r + 1]) == hipSuccess)
        sh->st.exchange_ms += ms;
    }
  for (int x = 0; x < 256; x++)
    sh->lfirst[x] = sh->local.wfirst[x];

  // ---- C2: histogram, totals, census
  std::vector<int64_t> v(FK_HIST_BINS + 256 + 16, 0);
  for (int i = 0; i < FK_HIST_BINS; i++) v[i] = sh->local.hist[i];
  for (int x = 0; x < 256; x++) v[FK_HIST_BINS + x] = sh->local.wfirst[x];
  int64_t *t = v.data() + FK_HIST_BINS + 256;
  t[0] = sh->local.max_inst; t[1] = ninst; t[2] = sent; t[3] = sh->local.ndistinct_super; t[4] = sh->local.nweighted;
  t[5] = sh->local.ndistinct; t[6] = sh->local.ntable; t[7] = expect; t[8] = sh->local.nsuper;
  if ((rc = allreduce_i64(sh, v.data(), (int) v.size())) != FK_OK) return (rc);
  memset(res, 0, sizeof(*res));
  for (int i = 0; i < FK_HIST_BINS; i++) res->hist[i] = v[i];
  for (int x = 0; x < 256; x++) res->wfirst[x] = v[FK_HIST_BINS + x];
  res->max_inst = t[0]; res->ninst = t[1]; res->nsuper = t[2]; res->ndistinct_super = t[3]; res->nweighted = t[4];
  res->ndistinct = t[5]; res->ntable = t[6];
  res->ms_split = sh->local.ms_split; res->ms_sort_super = sh->local.ms_sort_super; res->ms_expand = sh->local.ms_expand;
  res->ms_sort_kmer = sh->local.ms_sort_kmer; res->ms_count = sh->local.ms_count; res->ms_total = sh->local.ms_total;
  res->buckets_counted = sh->local.buckets_counted;
  // conservation: every record sent was received and counted, every k-mer instance is in the histogram
  if (t[2] != t[7] || t[2] != t[8])
    { fk_set_error(ctx, "sharded run: the exchange lost records (%lld sent, %lld expected, %lld counted)",
                   (long long) t[2], (long long) t[7], (long long) t[8]);
      return (FK_EHIP);
    }
  { int64_t tot = res->max_inst;
    for (int c = 1; c < 0x7fff; c++) tot += (int64_t) c * res->hist[c];
    if (tot != res->ninst)
      { fk_set_error(ctx, "sharded run: %lld k-mer instances were split but the histogram holds %lld",
                     (long long) res->ninst, (long long) tot);
        return (FK_EHIP);
      }
  }
  return (FK_OK);
}


extern "C" int fk_shard_set_write_cutoff(fk_shard *sh, int cutoff)
{ if (sh == NULL || cutoff < 0 || cutoff > 0x7fff) return (FK_EINVAL);
  sh->write_cutoff = cutoff;
  return (FK_OK);
}

// ---- the write cutoff: the entries of a sorted device table whose count reaches `cutoff`, in order --------------------
// (records of `stw` dwords, the count in the upper half of the last one).  Three small kernels: survivors per tile of
// 256, their exclusive scan (k_exscan_tiles), the compaction; then the first-byte bounds of what is left.
__global__ __launch_bounds__(256) void k_shf_count(const u32 *__restrict__ t, int64_t n, int stw, u32 cutoff, u32 *__restrict__ cnt)
{ const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
  const bool keep = (i < n) && ((t[i * stw + stw - 1] >> 16) >= cutoff);
  const u32 c = (u32) __syncthreads_count(keep ? 1 : 0);
  if (threadIdx.x == 0) cnt[blockIdx.x] = c;
}

__global__ __launch_bounds__(256) void k_shf_compact(const u32 *__restrict__ t, int64_t n, int stw, u32 cutoff,
                                                      const u64 *__restrict__ off, u32 *__restrict__ out)
{ __shared__ u32 tmp[8];
  const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
  const bool keep = (i < n) && ((t[i * stw + stw - 1] >> 16) >= cutoff);
  u32 tot;
  const u32 ex = fk_block_exscan_256<u32>(keep ? 1u : 0u, tmp, &tot);
  if (keep)
    { const u64 o = off[blockIdx.x] + ex;
      for (int w = 0; w < stw; w++)
        out[o * stw + w] = t[i * stw + w];
    }
}

__global__ __launch_bounds__(256) void k_shf_bounds(const unsigned char *__restrict__ t, int64_t n, int stride, int64_t *__restrict__ bounds)
{ const int b = threadIdx.x;                                  // bounds[b] = first record whose first key byte is >= b
  int64_t lo = 0, hi = n;
  while (lo < hi)
    { const int64_t mid = (lo + hi) >> 1;
      if (t[mid * stride] < b) lo = mid + 1; else hi = mid;
    }
  bounds[b] = lo;
  if (b == 0) bounds[256] = n;
}

// this rank's table without the entries below sh->write_cutoff: *tab (a buffer of the shard) and pre[0..256]
static int shard_filter_table(fk_shard *sh, const char **tab, int64_t *pre)
{ fk_ctx *ctx = sh->ctx;
  const fk_widths &w = ctx->wid;
  const int64_t n = ctx->last_ntab;
  const int stw = w.kmer_stride / 4;
  hipStream_t s = ctx->stream;
  const int64_t nt = (n + 255) / 256;
  u32 *cnt = (u32 *) fk_slot(ctx, FK_SLOT_PF_ZC, std::max<int64_t>(nt, 1) * 4 + 64);
  u64 *off = (u64 *) fk_slot(ctx, FK_SLOT_PF_ZO, std::max<int64_t>(nt, 1) * 8 + 64);
  if (cnt == NULL || off == NULL) return (FK_ENOMEM);
  for (int x = 0; x <= 256; x++) pre[x] = 0;
  if (n == 0)
    return (FK_OK);
  hipLaunchKernelGGL(k_shf_count, dim3((unsigned) nt), dim3(256), 0, s, (const u32 *) ctx->last_table, n, stw, (u32) sh->write_cutoff, cnt);
  hipLaunchKernelGGL(k_exscan_tiles, dim3(1), dim3(256), 0, s, (const u32 *) cnt, nt, off, ctx->d_scratch);
  FK_LAUNCH_CHECK(ctx);
  FK_HIP(ctx, hipMemcpyAsync(ctx->h_scratch, ctx->d_scratch, 8, hipMemcpyDeviceToHost, s));
  FK_HIP(ctx, hipStreamSynchronize(s));
  const int64_t keep = (int64_t) ctx->h_scratch[0];
  char *out = (char *) pf_reserve(sh, 0, std::max<int64_t>(keep, 1) * w.kmer_stride);
  if (out == NULL) return (FK_ENOMEM);
  hipLaunchKernelGGL(k_shf_compact, dim3((unsigned) nt), dim3(256), 0, s, (const u32 *) ctx->last_table, n, stw, (u32) sh->write_cutoff,
                     (const u64 *) off, (u32 *) out);
  int64_t *d_b = (int64_t *) (ctx->d_scratch + 8);
  hipLaunchKernelGGL(k_shf_bounds, dim3(1), dim3(256), 0, s, (const unsigned char *) out, keep, w.kmer_stride, d_b);
  FK_LAUNCH_CHECK(ctx);
  FK_HIP(ctx, hipMemcpyAsync(ctx->h_scratch, d_b, 257 * 8, hipMemcpyDeviceToHost, s));
  FK_HIP(ctx, hipStreamSynchronize(s));
  for (int x = 0; x <= 256; x++) pre[x] = (int64_t) ctx->h_scratch[x];
  *tab = out;
  return (FK_OK);
}

/* C3, the final gather after fk_shard_count: rank d receives the entries of the first-byte ranges of parts
   d*m .. d*m+m-1 (m = nparts / world; the boundaries are Table_Split's, count.c:1560-1565, over the all-reduced
   census) from every rank's sorted table -- a second exchange over RCCL --, orders its `world` runs of disjoint
   k-mers with the radix engine and brings the range to pinned host memory.  Replaces the heap merge of the bucket
   tables (table.c:346-533).  *table stays valid until the next gather or fk_shard_destroy; device and host buffers
   are kept for the next run. */
extern "C" int fk_shard_gather(fk_shard *sh, const fk_result *res, int nparts, const uint8_t **table, int64_t *nentries)
{ if (sh == NULL || res == NULL) return (FK_EINVAL);
  fk_ctx *ctx = sh->ctx;
  const int W = sh->world, me = sh->rank;
  const fk_widths &w = ctx->wid;
  if (table) *table = NULL;
  if (nentries) *nentries = 0;
  sh->g_n = -1;
  FK_HIP(ctx, hipSetDevice(ctx->device));
  if (ctx->prm.table_cutoff <= 0)
    { fk_set_error(ctx, "fk_shard_gather: no table was asked for (table_cutoff 0)");
      return (FK_EINVAL);
    }
  if (nparts < W || nparts % W != 0 || nparts > 256)
    { fk_set_error(ctx, "fk_shard_gather: %d parts cannot be dealt to %d ranks", nparts, W);
      return (FK_EINVAL);
    }
  const int m = nparts / W;
  int *split = sh->g_split;
  int rc = fk_ktab_split(res->wfirst, ctx->prm.kmer, nparts, split);
  if (rc != FK_OK) return (rc);

  // this rank's sorted table: the entries of first-byte range d go to rank d
  const char *tab = (const char *) ctx->last_table;
  int64_t pre[257];
  pre[0] = 0;
  for (int x = 0; x < 256; x++) pre[x + 1] = pre[x] + sh->lfirst[x];
  if (pre[256] != ctx->last_ntab)
    { fk_set_error(ctx, "fk_shard_gather: census (%lld) and table (%lld entries) disagree", (long long) pre[256],
                   (long long) ctx->last_ntab);
      return (FK_ESTATE);
    }
  sh->g_total = res->ntable;
  if (sh->write_cutoff > ctx->prm.table_cutoff)
    { // -t<n> beside -p: the table in HBM keeps every k-mer for the look-ups, the files take those that reach n
      int64_t tot = 0;
      int frc = shard_filter_table(sh, &tab, pre);
      if ((rc = agree(sh, frc)) != FK_OK) return (rc);
      tot = pre[256];
      if ((rc = allreduce_i64(sh, &tot, 1)) != FK_OK) return (rc);
      sh->g_total = tot;
    }
  sh->st.table_entries_written = sh->g_total;
  std::vector<int64_t> mine(W), all((size_t) W * W);
  for (int d = 0; d < W; d++)
    mine[d] = pre[split[(d + 1) * m]] - pre[split[d * m]];
  if ((rc = allgather_i64(sh, mine.data(), W, all.data())) != FK_OK) return (rc);
  int64_t nin = 0;
  for (int s = 0; s < W; s++) nin += all[(size_t) s * W + me];
  const int64_t bytes = std::max<int64_t>(nin, 1) * w.kmer_stride;
  int arc = FK_OK;
  if (bytes > sh->g_dev_cap)
    { for (int i = 0; i < 2; i++)
        { if (sh->g_dev[i]) hipFree(sh->g_dev[i]);
          sh->g_dev[i] = NULL;
        }
      sh->g_dev_cap = 0;
      const int64_t want = bytes + bytes / 16;
      if (hipMalloc((void **) &sh->g_dev[0], (size_t) want) != hipSuccess
          || hipMalloc((void **) &sh->g_dev[1], (size_t) want) != hipSuccess)
        { fk_set_error(ctx, "out of HBM: cannot allocate 2 x %lld bytes for this rank's range of the table", (long long) want);
          (void) hipGetLastError();
          arc = FK_ENOMEM;
        }
      else
        sh->g_dev_cap = want;
    }
  const int64_t hbytes = std::max<int64_t>(nin, 1) * w.kmer_word;
  if (arc == FK_OK && hbytes > sh->g_host_cap)
    { if (sh->g_host) fkx_pinned_free(sh->g_host);
      sh->g_host = NULL;
      sh->g_host_cap = 0;
      const int64_t want = hbytes + hbytes / 16;
      if (fkx_pinned_alloc((void **) &sh->g_host, want) != FK_OK)
        { fk_set_error(ctx, "out of host memory: cannot pin %lld bytes for this rank's table range", (long long) want);
          arc = FK_ENOMEM;
        }
      else
        sh->g_host_cap = want;
    }
  if ((rc = agree(sh, arc)) != FK_OK)              // (a rank without memory must not leave the others in the exchange)
    return (rc);
  char *a = sh->g_dev[0], *b = sh->g_dev[1];
  char   *sp[256], *rp[256];
  int64_t sb[256], rb[256], run = 0;
  for (int p = 0; p < W; p++)
    { sp[p] = (char *) tab + pre[split[p * m]] * w.kmer_stride;
      sb[p] = mine[p] * w.kmer_stride;
      rp[p] = a + run * w.kmer_stride;
      rb[p] = all[(size_t) p * W + me] * w.kmer_stride;
      run += all[(size_t) p * W + me];
    }
  FK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  sh->st.gather_sent_bytes = 0;
  for (int p = 0; p < W; p++)
    if (p != me) sh->st.gather_sent_bytes += sb[p];
  const double g0 = fk_wall();
  if ((rc = exchange(sh, sp, sb, rp, rb)) != FK_OK) return (rc);
  FK_HIP(ctx, hipStreamSynchronize(sh->xs));
  const double g1 = fk_wall();
  // W sorted runs of disjoint k-mer sets -> one sorted range: the MSD engine (ceil(log256 n) levels + the LDS finish;
  // ten LSD passes over KMER_BYTES until round 3)
  void *sorted = a;
  if (W > 1 && nin > 0
      && (rc = fkx_msd_sort(ctx, nin, a, b, w.kmer_stride, w.kmer_bytes, &sorted)) != FK_OK)
    return (rc);
  FK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const double g2 = fk_wall();
  const void *src = sorted;
  if (w.kmer_word != w.kmer_stride && nin > 0)
    { void *other = (sorted == (void *) a) ? (void *) b : (void *) a;
      if ((rc = fkx_repack_table(ctx, sorted, nin, other)) != FK_OK)
        return (rc);
      src = other;
    }
  if (nin > 0)
    { FK_HIP(ctx, hipMemcpyAsync(sh->g_host, src, (size_t) (nin * w.kmer_word), hipMemcpyDeviceToHost, ctx->stream));
      FK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
  sh->st.gather_exchange_ms = 1e3 * (g1 - g0);
  sh->st.gather_sort_ms     = 1e3 * (g2 - g1);
  sh->st.gather_d2h_ms      = 1e3 * (fk_wall() - g2);
  sh->g_n = nin;
  sh->g_nparts = nparts;
  sh->g_ib = fk_ktab_idx_bytes(ctx->prm.kmer, sh->g_total);
  if (table) *table = sh->g_host;
  if (nentries) *nentries = nin;
  return (FK_OK);
}

// a device buffer of the shard itself (not an arena slot of the context: after the count those may hold the table)
static void *pf_reserve(fk_shard *sh, int i, int64_t bytes)
{ if (sh->pf_cap[i] >= bytes && sh->pf_buf[i] != NULL)
    return (sh->pf_buf[i]);
  if (sh->pf_buf[i]) hipFree(sh->pf_buf[i]);
  sh->pf_buf[i] = NULL; sh->pf_cap[i] = 0;
  const int64_t want = bytes + bytes / 8 + 4096;
  if (hipMalloc(&sh->pf_buf[i], (size_t) want) != hipSuccess)
    { (void) hipGetLastError();
      fk_set_error(sh->ctx, "out of HBM: cannot allocate %lld bytes for the profile exchange", (long long) want);
      sh->pf_buf[i] = NULL;
      return (NULL);
    }
  sh->pf_cap[i] = want;
  return (sh->pf_buf[i]);
}

/* Profiles in the sharded run from C (the reference carries run ordinals through both of its sorts for this,
   count.c:639-1181, and stitches per-bucket fragments, merge.c:761-1006): after fk_shard_count with table_cutoff 1 every
   rank holds the counts of the k-mers of ITS minimizer buckets.  A piece of reads (d_bases: 0-terminated ASCII, 16-byte
   aligned, device memory; whole reads) is cut into super-mers that remember where they were cut from; the records
   travel to the ranks that own their buckets (the C1 exchange again, round by round), the owner looks their k-mers up in
   its table, the counts travel back in record order (2 bytes per k-mer instance), land at the positions, and the piece
   is encoded (README.md:1029-1069).  out: as fk_make_profiles (host memory of the context, valid until the next call).
   COLLECTIVE: all ranks call it the same number of times; a rank that has run out of reads passes nbytes = 0 until
   *active_ranks -- the ranks that passed reads in this call -- comes back 0 (then nothing was exchanged). */
extern "C" int fk_shard_profiles(fk_shard *sh, const void *d_bases, int64_t nbytes, fk_profiles *out, int *active_ranks)
{ if (sh == NULL || out == NULL || nbytes < 0 || (nbytes > 0 && d_bases == NULL)) return (FK_EINVAL);
  fk_ctx *ctx = sh->ctx;
  const int W = sh->world, R = sh->rounds, me = sh->rank, nb = ctx->prm.nbuckets;
  const int stride = ctx->wid.smer_stride;
  memset(out, 0, sizeof(*out));
  FK_HIP(ctx, hipSetDevice(ctx->device));
  int rc = FK_OK;
  if (!ctx->have_part_table)
    { fk_set_error(ctx, "fk_shard_profiles: needs fk_shard_count with table_cutoff 1 first");
      rc = FK_ESTATE;
    }
  { int64_t v[2] = { (nbytes > 0) ? 1 : 0, (rc != FK_OK) ? 1 : 0 };
    const int rc2 = allreduce_i64(sh, v, 2);
    if (rc != FK_OK) return (rc);
    if (rc2 != FK_OK) return (rc2);
    if (v[1] != 0)
      { fk_set_error(ctx, "fk_shard_profiles: another rank failed");
        return (FK_EHIP);
      }
    if (active_ranks) *active_ranks = (int) v[0];
    if (v[0] == 0)
      return (FK_OK);
  }

  // ---- this rank's piece: super-mers by bucket, with the positions they were cut from
  int64_t bc[256], off[257], ns = 0, ni = 0;
  memset(bc, 0, sizeof(bc));
  if (nbytes > 0)
    rc = fk_split_supermers(ctx, d_bases, nbytes, NULL, 0, &ns, &ni, bc);
  char *outbox = NULL;
  u64  *pos = NULL;
  if (rc == FK_OK && ns > 0)
    { outbox = (char *) pf_reserve(sh, 0, ns * stride);
      pos    = (u64 *) pf_reserve(sh, 1, ns * 8);
      if (outbox == NULL || pos == NULL) rc = FK_ENOMEM;
      else rc = fk_split_supermers_emit_pos(ctx, d_bases, nbytes, outbox, ns, bc, pos);
    }
  if ((rc = agree(sh, rc)) != FK_OK) return (rc);
  off[0] = 0;
  for (int b = 0; b < nb; b++) off[b + 1] = off[b] + bc[b];
  std::vector<int64_t> all((size_t) W * nb);
  if ((rc = allgather_i64(sh, bc, nb, all.data())) != FK_OK) return (rc);

  // A step that can fail on ONE rank (a stream synchronisation, a look-up, the scatter) never returns on its own: its
  // code is carried into the next agreement, so that every rank leaves the call together -- a rank that returned between
  // two collectives left its peers in the next one for good.
  bool first = true;
  int  carry = FK_OK;
  auto sync = [&](hipStream_t st)
    { if (hipStreamSynchronize(st) != hipSuccess && carry == FK_OK)
        { fk_set_error(ctx, "fk_shard_profiles: hipStreamSynchronize failed: %s", hipGetErrorString(hipGetLastError()));
          carry = FK_EHIP;
        }
    };
  for (int r = 0; r < R; r++)
    { // ---- records of the round's buckets to their owners
      char   *sp[256], *rp[256];
      int64_t sb[256], rb[256], nin = 0, seg[257];
      seg[0] = 0;
      for (int p = 0; p < W; p++)
        { seg[p + 1] = seg[p] + all[(size_t) p * nb + r * W + me];
          nin = seg[p + 1];
        }
      int lrc = (carry != FK_OK) ? carry : reserve_inbox(sh, 0, std::max<int64_t>(nin, 1) * stride);
      if (lrc == FK_OK)
        { sync(ctx->stream);
          lrc = carry;
        }
      if ((rc = agree(sh, lrc)) != FK_OK) return (rc);
      for (int p = 0; p < W; p++)
        { const int b = r * W + p;
          sp[p] = outbox + off[b] * stride;
          sb[p] = bc[b] * stride;
          rp[p] = (char *) sh->inbox[0] + seg[p] * stride;
          rb[p] = (seg[p + 1] - seg[p]) * stride;
        }
      if ((rc = exchange(sh, sp, sb, rp, rb)) != FK_OK) return (rc);       // (a collective: it fails on every rank or on none)
      sync(sh->xs);

      // ---- the owner looks them up, source by source (the counts of a source stay together)
      std::vector<int64_t> nis(W, 0), allni((size_t) W * W);
      lrc = carry;
      for (int p = 0; p < W && lrc == FK_OK; p++)
        lrc = fk_profile_lookup_supermers(ctx, (char *) sh->inbox[0] + seg[p] * stride, seg[p + 1] - seg[p], NULL, 0, &nis[p]);
      int64_t ctot = 0, coff[257];
      coff[0] = 0;
      for (int p = 0; p < W; p++) { coff[p + 1] = coff[p] + nis[p]; ctot = coff[p + 1]; }
      uint16_t *cout = NULL;
      if (lrc == FK_OK && (cout = (uint16_t *) pf_reserve(sh, 2, std::max<int64_t>(ctot, 1) * 2)) == NULL)
        lrc = FK_ENOMEM;
      for (int p = 0; p < W && lrc == FK_OK; p++)
        if (nis[p] > 0)
          lrc = fk_profile_lookup_supermers(ctx, (char *) sh->inbox[0] + seg[p] * stride, seg[p + 1] - seg[p], cout + coff[p],
                                            nis[p], &nis[p]);
      if ((rc = agree(sh, lrc)) != FK_OK) return (rc);
      if ((rc = allgather_i64(sh, nis.data(), W, allni.data())) != FK_OK) return (rc);

      // ---- the counts back to where the records came from
      int64_t back[257];
      back[0] = 0;
      for (int o = 0; o < W; o++) back[o + 1] = back[o] + allni[(size_t) o * W + me];
      uint16_t *cin = (uint16_t *) pf_reserve(sh, 3, std::max<int64_t>(back[W], 1) * 2);
      lrc = (cin == NULL) ? FK_ENOMEM : FK_OK;
      if (lrc == FK_OK)
        { sync(ctx->stream);
          lrc = carry;
        }
      if ((rc = agree(sh, lrc)) != FK_OK) return (rc);
      for (int p = 0; p < W; p++)
        { sp[p] = (char *) (cout + coff[p]);
          sb[p] = nis[p] * 2;
          rp[p] = (char *) (cin + back[p]);
          rb[p] = (back[p + 1] - back[p]) * 2;
        }
      if ((rc = exchange(sh, sp, sb, rp, rb)) != FK_OK) return (rc);
      sync(sh->xs);
      for (int o = 0; o < W && nbytes > 0 && carry == FK_OK; o++)
        { const int b = r * W + o;
          if (bc[b] == 0 && !first)
            continue;
          carry = fk_profile_scatter(ctx, outbox + off[b] * stride, pos + off[b], bc[b], cin + back[o], nbytes, first ? 1 : 0);
          first = false;                          // (a failure rides into the next round's agreement, or the one below)
        }
    }
  if ((rc = agree(sh, carry)) != FK_OK) return (rc);
  rc = (nbytes == 0) ? FK_OK : fk_profile_encode(ctx, d_bases, nbytes, out);
  return (agree(sh, rc));                         // (the encoder is the last thing that can fail on one rank alone)
}

/* The .prof files of a sharded run: every rank holds the profiles of a contiguous range of the data set's reads (in file
   order, ranks in order) and writes the hidden parts .<root>.pidx.N / .<root>.prof.N of its own range -- nparts / world
   each, its reads divided evenly over them (README.md:1010-1027: a part names the index of its first read, which is the
   number of reads of all ranks in front) -- and rank 0 the <root>.prof stub. */
extern "C" int fk_shard_write_prof(fk_shard *sh, const fk_profiles *p, int kmer, int nparts, const char *dir, const char *root)
{ if (sh == NULL || p == NULL || dir == NULL || root == NULL) return (FK_EINVAL);
  fk_ctx *ctx = sh->ctx;
  const int W = sh->world, me = sh->rank;
  if (nparts < W || nparts % W != 0)
    { fk_set_error(ctx, "fk_shard_write_prof: %d parts cannot be dealt to %d ranks", nparts, W);
      return (FK_EINVAL);
    }
  std::vector<int64_t> all(W);
  int64_t mine = p->nreads;
  int rc = allgather_i64(sh, &mine, 1, all.data());
  if (rc != FK_OK) return (rc);
  int64_t base = 0;
  for (int r = 0; r < me; r++) base += all[r];
  const int m = nparts / W;
  rc = fk_write_prof_range(p, kmer, nparts, me * m, m, base, me == 0, dir, root);
  return (agree(sh, rc));
}

/* Output files after fk_shard_count: <dir>/<root>.hist and the .ktab stub from rank 0, the hidden parts
   .<root>.ktab.<rank*m+1 .. rank*m+m> from every rank (m = nparts / world; nparts must be a multiple of the
   ranks), from the range fk_shard_gather brings in (called here).  With table_cutoff 0 only the histogram is
   written.  The files are byte for byte those fk_write_hist / fk_write_ktab write in a one-GPU run with -T nparts. */
extern "C" int fk_shard_write(fk_shard *sh, const fk_result *res, int nparts, const char *dir, const char *root)
{ if (sh == NULL || res == NULL || dir == NULL || root == NULL) return (FK_EINVAL);
  fk_ctx *ctx = sh->ctx;
  const int W = sh->world, me = sh->rank;
  const int cutoff = (sh->write_cutoff > ctx->prm.table_cutoff) ? sh->write_cutoff : ctx->prm.table_cutoff;
  FK_HIP(ctx, hipSetDevice(ctx->device));
  if (me == 0)
    { char path[4096];
      snprintf(path, sizeof(path), "%s/%s.hist", dir, root);
      int rc = fk_write_hist(res, ctx->prm.kmer, path);
      if (rc != FK_OK) return (rc);
    }
  if (cutoff <= 0)
    return (FK_OK);
  const uint8_t *host = NULL;
  int64_t nin = 0;
  int rc = fk_shard_gather(sh, res, nparts, &host, &nin);
  if (rc != FK_OK) return (rc);
  const int m = nparts / W, ib = sh->g_ib;
  int64_t npre = 1;
  for (int i = 0; i < ib; i++) npre *= 256;
  // every rank says whether its part files were written and its buffer for the reduce exists before any of them
  // enters the reduce: a rank whose disk is full must not leave the others in a collective
  std::vector<int64_t> pc((size_t) npre, 0);
  int wrc = fk_write_ktab_range(host, nin, ctx->prm.kmer, ib, sh->g_split, me * m, m, dir, root, pc.data());
  int64_t *d_pc = NULL;
  if (wrc == FK_OK && hipMalloc((void **) &d_pc, (size_t) npre * 8) != hipSuccess)
    { (void) hipGetLastError();
      d_pc = NULL;
      fk_set_error(ctx, "fk_shard_write: out of HBM for the prefix counts");
      wrc = FK_ENOMEM;
    }
  if ((rc = agree(sh, wrc)) != FK_OK)
    { if (d_pc) hipFree(d_pc);
      return (rc);
    }
  // per-prefix entry counts of all ranks -> rank 0 writes the stub
  if (fkx_h2d_pageable(ctx, sh->xs, d_pc, pc.data(), (size_t) npre * 8) != FK_OK
      || g_rccl.AllReduce(d_pc, d_pc, (size_t) npre, ncclInt64, ncclSum, sh->comm, sh->xs) != ncclSuccess
      || fkx_d2h_pageable(ctx, sh->xs, pc.data(), d_pc, (size_t) npre * 8) != FK_OK)
    { hipFree(d_pc);
      fk_set_error(ctx, "fk_shard_write: reducing the prefix counts failed");
      return (FK_EHIP);
    }
  hipFree(d_pc);
  int64_t tot = 0;
  for (int64_t i = 0; i < npre; i++) tot += pc[(size_t) i];
  if (tot != sh->g_total)
    { fk_set_error(ctx, "sharded run: the table exchange lost entries (%lld written, %lld counted)", (long long) tot,
                   (long long) sh->g_total);
      return (FK_EHIP);
    }
  if (me == 0)
    rc = fk_write_ktab_stub(ctx->prm.kmer, nparts, cutoff, ib, pc.data(), dir, root);
  return (rc);
}
