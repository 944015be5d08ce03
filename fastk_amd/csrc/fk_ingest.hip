// fk_ingest.hip -- the streaming interface: reads pushed block by block (ASCII, two bits per base, FASTQ / FASTA text),
// chunked ingest with a helper thread, the slab store of a chunk's super-mer records and the spill to host memory
// (DESIGN.md section 5b).  Split from fk_api.hip.
#include "fk_common.h"
#include <thread>
#include <algorithm>
#include <chrono>
#include <pthread.h>

// ---- streaming interface ------------------------------------------------------------------------
// stream that carries the copies into the read buffer: in chunked mode a stream of its own, so that
// they overlap the split of the previous chunk (which runs on ctx->stream)
static inline hipStream_t push_stream(fk_ctx *ctx)
{ return (ctx->chunk_bytes > 0 ? ctx->copy_stream : ctx->stream); }

// A run takes its reads in one form: 0-terminated ASCII / text (1) or two bits per base (2), until fk_reset.
static int push_form(fk_ctx *ctx, int form)
{ if (ctx->push_form != 0 && ctx->push_form != form)
    { fk_set_error(ctx, "the reads of a run come in one form: fk_push_packed and the ASCII / text pushes do not mix (fk_reset starts a new run)");
      return (FK_ESTATE);
    }
  ctx->push_form = form;
  return (FK_OK);
}

static int reserve_reads(fk_ctx *ctx, int64_t extra)
{ const int64_t need = ctx->reads_len + extra + 64;
  if (need <= ctx->reads_cap)
    return (FK_OK);
  hipStream_t ps = push_stream(ctx);
  int64_t ncap = std::max<int64_t>(need, ctx->reads_cap * 2);
  ncap = std::max<int64_t>(ncap, 64ll << 20);
  if (ctx->chunk_bytes > 0)                      // a chunk's worth at once: no re-allocation while it fills
    ncap = std::max<int64_t>(ncap, ctx->chunk_bytes + ctx->chunk_bytes / 8 + (64ll << 20));
  char *nbuf = NULL;
  FK_HIP(ctx, hipMalloc((void **) &nbuf, (size_t) ncap));
  if (ctx->d_reads != NULL)
    { FK_HIP(ctx, hipStreamSynchronize(ctx->stream));       // device-side pushes write through this one
      if (ctx->reads_len > 0)
        FK_HIP(ctx, hipMemcpyAsync(nbuf, ctx->d_reads, (size_t) ctx->reads_len, hipMemcpyDeviceToDevice, ps));
      FK_HIP(ctx, hipStreamSynchronize(ps));
    }
  if (ctx->d_reads != NULL)
    FK_HIP(ctx, hipFree(ctx->d_reads));
  ctx->d_reads = nbuf;
  ctx->reads_cap = ncap;
  return (FK_OK);
}

// Pinned host buffer for a spilled chunk: one per chunk, kept in the context and used again by the next
// run (hipHostMalloc moves ~15 GB/s, slower than the spill copy itself).
static int spill_acquire(fk_ctx *ctx, int64_t bytes, int *slot)
{ int best = -1;
  for (int i = 0; i < ctx->nspill; i++)
    if (!ctx->spill_buf[i].in_use && ctx->spill_buf[i].cap >= bytes
        && (best < 0 || ctx->spill_buf[i].cap < ctx->spill_buf[best].cap))
      best = i;
  if (best < 0)
    { for (int i = 0; i < ctx->nspill && best < 0; i++)      // an idle one that is too small: replace it
        if (!ctx->spill_buf[i].in_use)
          { fkx_pinned_free(ctx->spill_buf[i].ptr);
            ctx->spill_buf[i].ptr = NULL;
            ctx->spill_buf[i].cap = 0;
            best = i;
          }
      if (best < 0)
        { if (ctx->nspill == ctx->spill_cap)
            { ctx->spill_cap = ctx->spill_cap * 2 + 16;
              ctx->spill_buf = (fk_spill_buf *) realloc(ctx->spill_buf, sizeof(fk_spill_buf) * (size_t) ctx->spill_cap);
              if (ctx->spill_buf == NULL) { ctx->nspill = ctx->spill_cap = 0; return (FK_ENOMEM); }
            }
          best = ctx->nspill++;
          ctx->spill_buf[best].ptr = NULL;
          ctx->spill_buf[best].cap = 0;
        }
      const int64_t want = bytes + bytes / 16 + 4096;
      if (fkx_pinned_alloc(&ctx->spill_buf[best].ptr, want) != FK_OK)
        { ctx->spill_buf[best].ptr = NULL;
          return (FK_ENOMEM);
        }
      ctx->spill_buf[best].cap = want;
    }
  ctx->spill_buf[best].in_use = 1;
  *slot = best;
  return (FK_OK);
}

void fkx_free_chunk(fk_ctx *ctx, fk_chunk *c)
{ if (c->on_host && c->total > 0)
    ctx->spill_buf[c->spill_slot].in_use = 0;
  c->total = 0;                        // (HBM runs live in the slabs, which are rewound as a whole)
}

void fkx_rewind_slabs(fk_ctx *ctx)
{ for (int i = 0; i < ctx->nslabs; i++)
    ctx->slabs[i].used = 0;
  ctx->chunk_hbm_bytes = 0;
}

#define FK_SLAB_BYTES (8ll << 30)

// room for `bytes` of records in the HBM store; NULL when that would exceed spill_limit (or HBM)
void *fkx_slab_alloc(fk_ctx *ctx, int64_t bytes)
{ bytes = (bytes + 255) & ~255ll;
  for (int i = 0; i < ctx->nslabs; i++)
    if (ctx->slabs[i].cap - ctx->slabs[i].used >= bytes)
      { void *p = ctx->slabs[i].ptr + ctx->slabs[i].used;
        ctx->slabs[i].used += bytes;
        ctx->chunk_hbm_bytes += bytes;
        return (p);
      }
  int64_t held = 0;
  for (int i = 0; i < ctx->nslabs; i++)
    held += ctx->slabs[i].cap;
  int64_t cap = std::max<int64_t>(ctx->dbg_slab_bytes > 0 ? ctx->dbg_slab_bytes : FK_SLAB_BYTES, bytes);
  if (ctx->spill_limit > 0 && held + cap > ctx->spill_limit)
    cap = std::max<int64_t>(ctx->spill_limit - held, 0);        // the last slab may be smaller
  if (cap < bytes)
    return (NULL);
  if (ctx->nslabs == ctx->slabs_cap)
    { ctx->slabs_cap = ctx->slabs_cap * 2 + 16;
      ctx->slabs = (fk_slab *) realloc(ctx->slabs, sizeof(fk_slab) * (size_t) ctx->slabs_cap);
      if (ctx->slabs == NULL) { ctx->nslabs = ctx->slabs_cap = 0; return (NULL); }
    }
  char *p = NULL;
  if (hipMalloc((void **) &p, (size_t) cap) != hipSuccess)
    { (void) hipGetLastError();
      return (NULL);
    }
  fk_slab *sl = &ctx->slabs[ctx->nslabs++];
  sl->ptr = p; sl->cap = cap; sl->used = bytes;
  ctx->chunk_hbm_bytes += bytes;
  return (p);
}

// Split `len` bytes of reads at `buf` into super-mers grouped by bucket and keep those (compacted)
// as a chunk: with hbm_budget set, the ASCII reads never have to be resident as a whole.  Runs on
// ctx->stream; called by the flush helper thread or, with no helper running, by the pushing thread.
static int flush_buffer(fk_ctx *ctx, const char *buf, int64_t len, const fk_pkstore *st = NULL)
{ const int stride = ctx->wid.smer_stride;
  hipStream_t s = ctx->stream;
  if (len == 0)
    return (FK_OK);
  void   *out = NULL;
  int64_t ns = 0, ni = 0, bc[256], bo[256];
  const auto tc0 = std::chrono::steady_clock::now();
  fk_pkview pv;                                  // packed reads: len counts positions
  if (st != NULL)
    { pv.roff = st->roff; pv.nreads = st->nreads; pv.inv = st->inv; pv.ninv = st->ninv; }
  int rc = fkx_split_fast(ctx, buf, len, &out, &ns, &ni, bc, bo, (st != NULL) ? &pv : NULL);
  if (rc != FK_OK)
    return (rc);
  const auto tc1 = std::chrono::steady_clock::now();
  ctx->chunk_ninst += ni;
  if (ns == 0)
    return (FK_OK);
  if (ctx->nchunks == ctx->chunks_cap)
    { ctx->chunks_cap = ctx->chunks_cap * 2 + 16;
      ctx->chunks = (fk_chunk *) realloc(ctx->chunks, sizeof(fk_chunk) * (size_t) ctx->chunks_cap);
      if (ctx->chunks == NULL) { ctx->nchunks = ctx->chunks_cap = 0; return (FK_ENOMEM); }
    }
  fk_chunk *c = &ctx->chunks[ctx->nchunks];
  memset(c, 0, sizeof(*c));
  if (ctx->nchunks == 0)
    ctx->spilled_bytes = 0;
  // the records stay in HBM (slab store) up to spill_limit; beyond that a chunk goes to pinned host
  // memory as a whole and comes back bucket by bucket when its buckets are counted
  const int64_t bytes = ns * stride;
  for (int b = 0; b < ctx->prm.nbuckets; b++)
    { c->cnt[b] = bc[b];
      c->run[b] = NULL;
    }
  { int b = 0;
    for (; b < ctx->prm.nbuckets; b++)
      if (bc[b] > 0 && (c->run[b] = fkx_slab_alloc(ctx, bc[b] * stride)) == NULL)
        break;
    c->on_host = (b < ctx->prm.nbuckets);
  }
  if (c->on_host)
    { if (spill_acquire(ctx, bytes, &c->spill_slot) != FK_OK)
        { fk_set_error(ctx, "out of host memory: cannot spill %lld super-mer records of a chunk", (long long) ns);
          return (FK_ENOMEM);
        }
      // (slab room taken for the first buckets of this chunk before the store ran out stays unused)
      char *h = (char *) ctx->spill_buf[c->spill_slot].ptr;
      int64_t run = 0;
      for (int b = 0; b < ctx->prm.nbuckets; b++)
        { c->run[b] = h + run * stride;
          run += bc[b];
        }
      ctx->spilled_bytes += bytes;
    }
  int64_t run = 0;
  for (int b = 0; b < ctx->prm.nbuckets; b++)
    { if (bc[b] > 0)
        FK_HIP(ctx, hipMemcpyAsync(c->run[b], (char *) out + bo[b] * stride, (size_t) (bc[b] * stride),
                                   c->on_host ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, s));
      run += bc[b];
    }
  c->total = run;
  ctx->nchunks += 1;
  const auto tc2 = std::chrono::steady_clock::now();
  FK_HIP(ctx, hipStreamSynchronize(s));
  if (ctx->dbg_verbose)
    { const auto tc3 = std::chrono::steady_clock::now();
      fprintf(stderr, "  chunk %d: %lld bytes of reads -> %lld records%s; split %.1f ms, allocation + copies issued %.1f ms, "
                      "copies done %.1f ms\n", ctx->nchunks - 1, (long long) len, (long long) ns, c->on_host ? " (host)" : "",
              std::chrono::duration<double, std::milli>(tc1 - tc0).count(),
              std::chrono::duration<double, std::milli>(tc2 - tc1).count(),
              std::chrono::duration<double, std::milli>(tc3 - tc2).count());
    }
  return (FK_OK);
}

// Wait for the flush helper, if one is running; returns its result.
int fkx_flush_join(fk_ctx *ctx)
{ if (ctx->flush_thread == NULL)
    return (FK_OK);
  std::thread *t = (std::thread *) ctx->flush_thread;
  t->join();
  delete t;
  ctx->flush_thread = NULL;
  if (ctx->flush_rc != FK_OK)
    memcpy(ctx->err, ctx->flush_err, sizeof(ctx->err));
  return (ctx->flush_rc);
}

// The reads pushed so far become a chunk.  Called with the push lock held.
//   async: the copies into the buffer were issued on copy_stream (fk_push_block, chunked mode) --
//          the buffer goes to a helper thread that waits for them and splits it, and the caller goes
//          on copying into the other buffer;
//   else:  the split runs here and now.
//   carry: the text pushes (fk_push_fastq / _fasta) may stop in the middle of a read; the chunk is
//          split as it is and the last K-1 bases of the unfinished read open the next chunk, so the k-mers
//          across the cut are counted exactly once (what a block with rem > 0 does, io.c:557-570).
int fkx_flush_chunk(fk_ctx *ctx, bool async, bool carry)
{ int rc = fkx_flush_join(ctx);
  if (rc != FK_OK || ctx->reads_len == 0)
    return (rc);
  char   *buf = ctx->d_reads;
  int64_t len = ctx->reads_len;
  fk_pkstore *st = (ctx->push_form == 2) ? &ctx->pk[ctx->pk_cur] : NULL;
  if (st != NULL)
    { // packed reads: the chunk is the store of the current buffer; the other store takes the next chunk's reads
      len = st->npos;
      if (!async)
        { FK_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
          FK_HIP(ctx, hipStreamSynchronize(ctx->stream));
          ctx->reads_len = 0;
          rc = flush_buffer(ctx, buf, len, st);
          st->nreads = st->ninv = st->npos = 0;
          return (rc);
        }
      FK_HIP(ctx, hipEventRecord(ctx->reads_ev, ctx->copy_stream));
      std::swap(ctx->d_reads, ctx->d_reads_alt);
      std::swap(ctx->reads_cap, ctx->reads_cap_alt);
      ctx->reads_len = 0;
      ctx->pk_cur ^= 1;
      ctx->pk[ctx->pk_cur].nreads = ctx->pk[ctx->pk_cur].ninv = ctx->pk[ctx->pk_cur].npos = 0;
      ctx->flush_rc = FK_OK;
      ctx->flush_thread = new std::thread([ctx, buf, len, st]()
        { int r = FK_EHIP;
          if (hipSetDevice(ctx->device) == hipSuccess
              && hipStreamWaitEvent(ctx->stream, ctx->reads_ev, 0) == hipSuccess)
            r = flush_buffer(ctx, buf, len, st);
          if (r != FK_OK)
            memcpy(ctx->flush_err, ctx->err, sizeof(ctx->flush_err));
          ctx->flush_rc = r;
        });
      return (FK_OK);
    }
  if (!async)
    { FK_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
      FK_HIP(ctx, hipStreamSynchronize(ctx->stream));
      int64_t keep = 0;
      if (carry)
        { const int64_t look = std::min<int64_t>(len, ctx->prm.kmer - 1);
          char tail[256];
          FK_HIP(ctx, hipMemcpy(tail, buf + len - look, (size_t) look, hipMemcpyDeviceToHost));
          while (keep < look && tail[look - 1 - keep] != 0)      // bases after the last read terminator
            keep += 1;
        }
      ctx->reads_len = 0;
      rc = flush_buffer(ctx, buf, len);
      if (rc == FK_OK && keep > 0)
        { // (source and destination cannot overlap: a chunk is far longer than 2 (K-1) bytes -- but be safe)
          if (len >= 2 * keep)
            FK_HIP(ctx, hipMemcpy(buf, buf + len - keep, (size_t) keep, hipMemcpyDeviceToDevice));
          else
            { char tmp[256];
              FK_HIP(ctx, hipMemcpy(tmp, buf + len - keep, (size_t) keep, hipMemcpyDeviceToHost));
              FK_HIP(ctx, hipMemcpy(buf, tmp, (size_t) keep, hipMemcpyHostToDevice));
            }
          ctx->reads_len = keep;
        }
      return (rc);
    }
  FK_HIP(ctx, hipEventRecord(ctx->reads_ev, ctx->copy_stream));
  std::swap(ctx->d_reads, ctx->d_reads_alt);
  std::swap(ctx->reads_cap, ctx->reads_cap_alt);
  ctx->reads_len = 0;
  ctx->flush_rc = FK_OK;
  ctx->flush_thread = new std::thread([ctx, buf, len]()
    { int r = FK_EHIP;
      if (hipSetDevice(ctx->device) == hipSuccess
          && hipStreamWaitEvent(ctx->stream, ctx->reads_ev, 0) == hipSuccess)
        r = flush_buffer(ctx, buf, len);
      if (r != FK_OK)
        memcpy(ctx->flush_err, ctx->err, sizeof(ctx->flush_err));
      ctx->flush_rc = r;
    });
  return (FK_OK);
}

// fk_push_device / _fastq / _fasta write the read buffer through ctx->stream: no flush helper may be
// running on it, and earlier host blocks must have landed.  Called with the push lock held.
static int device_push_begin(fk_ctx *ctx)
{ int rc = fkx_flush_join(ctx);
  if (rc != FK_OK)
    return (rc);
  if (ctx->chunk_bytes > 0)
    FK_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
  return (FK_OK);
}

extern "C" int fk_push_block(fk_ctx *ctx, const char *bases, const int32_t *boff, int nreads,
                             int rem, int tid)
{ if (ctx == NULL || bases == NULL || boff == NULL || nreads < 0) return (FK_EINVAL);
  if (nreads == 0)
    return (FK_OK);
  const int64_t len = (int64_t) boff[nreads] - boff[0];
  int rc = FK_OK;
  pthread_mutex_lock((pthread_mutex_t *) ctx->push_lock);
  do
    { hipSetDevice(ctx->device);
      if ((rc = push_form(ctx, 1)) != FK_OK || (rc = reserve_reads(ctx, len)) != FK_OK)
        break;
      hipStream_t ps = push_stream(ctx);
      // blocks that lie in pinned host memory are copied from where they are
      bool direct = false;
      if (!ctx->prm.exact_parts && ctx->prm.bc_prefix == 0)
        { hipPointerAttribute_t at;
          if (hipPointerGetAttributes(&at, bases) == hipSuccess)
            direct = (at.type == hipMemoryTypeHost);
          else
            (void) hipGetLastError();
        }
      if (direct)
        { if (hipMemcpyAsync(ctx->d_reads + ctx->reads_len, bases + boff[0], (size_t) len, hipMemcpyHostToDevice,
                             ps) != hipSuccess)
            { fk_set_error(ctx, "fk_push_block: host to device copy failed");
              rc = FK_EHIP;
              break;
            }
          if (hipStreamSynchronize(ps) != hipSuccess)      // the caller may reuse the block once we return
            { rc = FK_EHIP; break; }                       // (the split of the previous chunk overlaps anyway)
        }
      const int si = ctx->stage_idx;
      if (!direct && ctx->stage_cap < len)
        { for (int i = 0; i < 2; i++)
            { if (ctx->h_stage[i])
                { hipEventSynchronize(ctx->stage_ev[i]);
                  fkx_pinned_free(ctx->h_stage[i]);
                  ctx->h_stage[i] = NULL;
                }
            }
          ctx->stage_cap = std::max<int64_t>(len, 4ll << 20);
          for (int i = 0; i < 2; i++)
            if (fkx_pinned_alloc((void **) &ctx->h_stage[i], ctx->stage_cap) != FK_OK)
              { fk_set_error(ctx, "fk_push_block: cannot allocate pinned staging");
                rc = FK_ENOMEM;
              }
          if (rc != FK_OK)
            break;
        }
      if (!direct && hipEventSynchronize(ctx->stage_ev[si]) != hipSuccess)
        { rc = FK_EHIP; break; }
      char *st = ctx->h_stage[si];
      if (!direct)
        memcpy(st, bases + boff[0], (size_t) len);
      if (ctx->prm.exact_parts)
        { if (ctx->nroff + nreads + 1 > ctx->roff_cap)
            { ctx->roff_cap = std::max<int64_t>(ctx->nroff + nreads + 1, ctx->roff_cap * 2 + 1024);
              ctx->h_roff = (int64_t *) realloc(ctx->h_roff, sizeof(int64_t) * (size_t) ctx->roff_cap);
              if (ctx->h_roff == NULL) { rc = FK_ENOMEM; break; }
            }
          for (int i = 0; i < nreads; i++)
            ctx->h_roff[ctx->nroff++] = ctx->reads_len + (boff[i] - boff[0]);
        }
      else if (ctx->prm.bc_prefix > 0)      // -bc: the skipped prefix can never be inside a k-mer
        for (int i = 0; i < nreads; i++)
          { const int64_t o = boff[i] - boff[0];
            const int64_t e = boff[i + 1] - boff[0] - 1;
            for (int64_t j = o; j < e && j < o + ctx->prm.bc_prefix; j++)
              st[j] = 'N';                 // not a base, and not a read terminator either (profiles)
          }
      if (!direct)
        { if (hipMemcpyAsync(ctx->d_reads + ctx->reads_len, st, (size_t) len, hipMemcpyHostToDevice, ps) != hipSuccess
              || hipEventRecord(ctx->stage_ev[si], ps) != hipSuccess)
            { fk_set_error(ctx, "fk_push_block: host to device copy failed");
              rc = FK_EHIP;
              break;
            }
          ctx->stage_idx ^= 1;
        }
      ctx->reads_len += len;
      if (ctx->nblocks == ctx->blocks_cap)
        { ctx->blocks_cap = ctx->blocks_cap * 2 + 256;
          ctx->blocks = (fk_block *) realloc(ctx->blocks, sizeof(fk_block) * (size_t) ctx->blocks_cap);
          if (ctx->blocks == NULL) { ctx->nblocks = ctx->blocks_cap = 0; rc = FK_ENOMEM; break; }
        }
      ctx->blocks[ctx->nblocks].tid = tid;
      ctx->blocks[ctx->nblocks].rem = rem;
      ctx->blocks[ctx->nblocks].nreads = nreads;
      ctx->nblocks += 1;
      if (ctx->chunk_bytes > 0 && ctx->reads_len >= ctx->chunk_bytes)
        rc = fkx_flush_chunk(ctx, true);
    }
  while (0);
  pthread_mutex_unlock((pthread_mutex_t *) ctx->push_lock);
  return (rc);
}

// room for `nreads` more reads and `ninv` more stretches in a packed store (contents kept)
static int pk_reserve(fk_ctx *ctx, fk_pkstore *st, int64_t nreads, int64_t ninv, hipStream_t ps)
{ if (st->nreads + nreads + 1 > st->roff_cap)
    { const int64_t ncap = std::max<int64_t>(st->nreads + nreads + 1, st->roff_cap * 2 + (1 << 16));
      int64_t *n = NULL;
      FK_HIP(ctx, hipStreamSynchronize(ps));
      FK_HIP(ctx, hipMalloc((void **) &n, (size_t) ncap * 8));
      if (st->roff != NULL)
        { if (st->nreads > 0)
            FK_HIP(ctx, hipMemcpy(n, st->roff, (size_t) (st->nreads + 1) * 8, hipMemcpyDeviceToDevice));
          FK_HIP(ctx, hipFree(st->roff));
        }
      st->roff = n; st->roff_cap = ncap;
    }
  if (st->ninv + ninv > st->inv_cap)
    { const int64_t ncap = std::max<int64_t>(st->ninv + ninv, st->inv_cap * 2 + (1 << 12));
      int64_t *n = NULL;
      FK_HIP(ctx, hipStreamSynchronize(ps));
      FK_HIP(ctx, hipMalloc((void **) &n, (size_t) ncap * 16));
      if (st->inv != NULL)
        { if (st->ninv > 0)
            FK_HIP(ctx, hipMemcpy(n, st->inv, (size_t) st->ninv * 16, hipMemcpyDeviceToDevice));
          FK_HIP(ctx, hipFree(st->inv));
        }
      st->inv = n; st->inv_cap = ncap;
    }
  return (FK_OK);
}

/* The reads of a DATA_BLOCK in two bits per base (see include/fastk_amd.h).  They STAY packed: the codes are copied
   behind those of the earlier blocks (every block starts on a dword: up to 15 positions of padding behind a block,
   listed as one more stretch without acgt), the read offsets and the stretches go to the store's two sorted lists,
   and the splitter's tile loader reads that form (fk_split.hip, PACKED kernels).  Nothing is unpacked on the counting
   path; exact_parts runs and fk_make_profiles restore the ASCII reads on demand (fkx_unpack_store). */
extern "C" int fk_push_packed(fk_ctx *ctx, const uint8_t *codes, int64_t nbases, const int32_t *rlen, int nreads,
                              const int64_t *inv, int ninv, int rem, int tid)
{ if (ctx == NULL || nreads < 0 || nbases < 0 || ninv < 0 || (nbases > 0 && codes == NULL) || (nreads > 0 && rlen == NULL)
      || (ninv > 0 && inv == NULL))
    return (FK_EINVAL);
  if (nreads == 0)
    return (FK_OK);
  if (ctx->prm.bc_prefix > 0)
    { fk_set_error(ctx, "fk_push_packed: -bc needs the ASCII form (fk_push_block)");
      return (FK_EUNSUPPORTED);
    }
  const int64_t npos   = (nbases + 15) & ~15ll;          // positions the block takes
  const int64_t cbytes = (nbases + 3) / 4;
  int rc = FK_OK;
  pthread_mutex_lock((pthread_mutex_t *) ctx->push_lock);
  do
    { hipSetDevice(ctx->device);
      if ((rc = push_form(ctx, 2)) != FK_OK || (rc = reserve_reads(ctx, npos / 4)) != FK_OK)
        break;
      hipStream_t ps = push_stream(ctx);
      fk_pkstore *st = &ctx->pk[ctx->pk_cur];
      const int pad = (npos > nbases) ? 1 : 0;
      if ((rc = pk_reserve(ctx, st, nreads, (int64_t) ninv + pad, ps)) != FK_OK)
        break;
      // pinned staging for the two lists, sized on its own (a later block may bring fewer bases and more reads)
      const int64_t hneed = ((int64_t) nreads + 1) * 8 + ((int64_t) ninv + 1) * 16;
      if (ctx->h_pk_cap < hneed)
        { if (hipStreamSynchronize(ps) != hipSuccess) { rc = FK_EHIP; break; }
          if (ctx->h_pk) hipHostFree(ctx->h_pk);
          ctx->h_pk = NULL; ctx->h_pk_cap = 0;
          const int64_t cap = hneed + hneed / 2 + 4096;
          if (hipHostMalloc((void **) &ctx->h_pk, (size_t) cap, hipHostMallocDefault) != hipSuccess)
            { fk_set_error(ctx, "fk_push_packed: out of memory for %lld bytes of pinned staging", (long long) cap);
              rc = FK_ENOMEM;
              break;
            }
          ctx->h_pk_cap = cap;
        }
      const int64_t base = st->npos;
      int64_t *hro = ctx->h_pk, *hinv = ctx->h_pk + nreads + 1;
      int64_t  run = 0;
      for (int i = 0; i < nreads; i++)
        { if (rlen[i] < 0) { rc = FK_EINVAL; break; }
          hro[i] = base + run;
          run += rlen[i];
        }
      hro[nreads] = base + npos;                         // the next block's first read (the padding lies in front of it)
      if (rc != FK_OK || run != nbases)
        { fk_set_error(ctx, "fk_push_packed: the read lengths add up to %lld, not to %lld bases", (long long) run, (long long) nbases);
          rc = FK_EINVAL;
          break;
        }
      { int64_t prev = 0;                                // sorted, disjoint, inside the block
        for (int j = 0; j < ninv && rc == FK_OK; j++)
          { const int64_t s0 = inv[2 * j], n0 = inv[2 * j + 1];
            if (s0 < prev || n0 <= 0 || s0 + n0 > nbases)
              { fk_set_error(ctx, "fk_push_packed: stretch %d (%lld, %lld) is out of order or outside the block's %lld bases",
                             j, (long long) s0, (long long) n0, (long long) nbases);
                rc = FK_EINVAL;
              }
            hinv[2 * j] = base + s0;
            hinv[2 * j + 1] = n0;
            prev = s0 + n0;
          }
        if (rc != FK_OK)
          break;
        if (pad)
          { hinv[2 * ninv] = base + nbases;
            hinv[2 * ninv + 1] = npos - nbases;
          }
      }
      // (a copy KERNEL reading the pinned block across PCIe instead of hipMemcpyAsync was tried in round 4: 1.40 s
      //  instead of 1.15 s inside the 2,236 pushes of configs[2] -- the ~32 GB/s of these 17 MB blocks is not the
      //  copy engine's set-up)
      // the caller's codes: asynchronously only from memory the runtime knows as pinned (FastK_amd's reader threads pack
      // into fk_host_alloc'ed buffers); from pageable memory with a blocking copy (fkx_h2d_pageable, fk_common.h)
      bool pinned_src = false;
      { hipPointerAttribute_t at;
        if (cbytes > 0 && hipPointerGetAttributes(&at, codes) == hipSuccess)
          pinned_src = (at.type == hipMemoryTypeHost);
        else
          (void) hipGetLastError();
      }
      if (cbytes > 0 && !pinned_src && (rc = fkx_h2d_pageable(ctx, ps, ctx->d_reads + ctx->reads_len, codes, (size_t) cbytes)) != FK_OK)
        break;
      if ((cbytes > 0 && pinned_src
           && hipMemcpyAsync(ctx->d_reads + ctx->reads_len, codes, (size_t) cbytes, hipMemcpyHostToDevice, ps) != hipSuccess)
          || hipMemcpyAsync(st->roff + st->nreads, hro, (size_t) (nreads + 1) * 8, hipMemcpyHostToDevice, ps) != hipSuccess
          || (ninv + pad > 0
              && hipMemcpyAsync(st->inv + 2 * st->ninv, hinv, (size_t) (ninv + pad) * 16, hipMemcpyHostToDevice, ps) != hipSuccess))
        { fk_set_error(ctx, "fk_push_packed: host to device copy failed");
          (void) hipStreamSynchronize(ps);                 // the copies queued before the failing one still read the caller's
          rc = FK_EHIP;                                    // codes and h_pk: neither may be reused before they are through
          break;
        }
      if (hipStreamSynchronize(ps) != hipSuccess)          // the caller may reuse its buffers once we return
        { rc = FK_EHIP; break; }
      if (ctx->prm.exact_parts)
        { if (ctx->nroff + nreads + 1 > ctx->roff_cap)
            { ctx->roff_cap = std::max<int64_t>(ctx->nroff + nreads + 1, ctx->roff_cap * 2 + 1024);
              ctx->h_roff = (int64_t *) realloc(ctx->h_roff, sizeof(int64_t) * (size_t) ctx->roff_cap);
              if (ctx->h_roff == NULL) { rc = FK_ENOMEM; break; }
            }
          for (int i = 0; i < nreads; i++)                 // where the read starts in the restored ASCII
            ctx->h_roff[ctx->nroff++] = ctx->pk_ascii_len + (hro[i] - base) + i;
        }
      if (ctx->nblocks == ctx->blocks_cap)
        { ctx->blocks_cap = ctx->blocks_cap * 2 + 256;
          ctx->blocks = (fk_block *) realloc(ctx->blocks, sizeof(fk_block) * (size_t) ctx->blocks_cap);
          if (ctx->blocks == NULL) { ctx->nblocks = ctx->blocks_cap = 0; rc = FK_ENOMEM; break; }
        }
      fk_block *bl = &ctx->blocks[ctx->nblocks];
      bl->tid = tid; bl->rem = rem; bl->nreads = nreads;
      bl->pk_pos = base; bl->pk_nbases = nbases; bl->pk_read0 = st->nreads; bl->pk_inv0 = st->ninv; bl->pk_ninv = ninv;
      ctx->nblocks += 1;
      ctx->reads_len += npos / 4;
      ctx->pk_ascii_len += nbases + nreads;
      st->nreads += nreads;
      st->ninv   += ninv + pad;
      st->npos   += npos;
      if (ctx->chunk_bytes > 0 && st->npos >= ctx->chunk_bytes)
        rc = fkx_flush_chunk(ctx, true);
    }
  while (0);
  pthread_mutex_unlock((pthread_mutex_t *) ctx->push_lock);
  return (rc);
}

/* The packed reads of a resident run (no chunk was flushed) as 0-terminated ASCII in an arena slot, block by block:
   for the consumers that walk reads byte by byte (exact_parts, fk_make_profiles). */
int fkx_unpack_store(fk_ctx *ctx, void **d_ascii, int64_t *nbytes)
{ *d_ascii = NULL; *nbytes = 0;
  if (ctx->push_form != 2)
    return (FK_ESTATE);
  const fk_pkstore *st = &ctx->pk[ctx->pk_cur];
  const int64_t len = ctx->pk_ascii_len;
  char *dst = (char *) fk_slot(ctx, FK_SLOT_PK_ASCII, len + 64);
  if (dst == NULL)
    return (FK_ENOMEM);
  hipStream_t s = ctx->stream;
  FK_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
  int64_t at = 0;
  for (int64_t b = 0; b < ctx->nblocks; b++)
    { const fk_block *bl = &ctx->blocks[b];
      const int rc = fkx_unpack_reads(ctx, s, ctx->d_reads, bl->pk_pos, bl->pk_nbases, st->roff + bl->pk_read0, bl->nreads,
                                      st->inv + 2 * bl->pk_inv0, bl->pk_ninv, dst + at);
      if (rc != FK_OK)
        return (rc);
      at += bl->pk_nbases + bl->nreads;
    }
  if (at != len)
    { fk_set_error(ctx, "internal: %lld bytes of reads restored, %lld expected", (long long) at, (long long) len);
      return (FK_EHIP);
    }
  *d_ascii = dst;
  *nbytes = len;
  return (FK_OK);
}

extern "C" int fk_train_block(fk_ctx *ctx, const char *bases, const int32_t *boff, int nreads)
{ if (ctx == NULL || bases == NULL || boff == NULL || nreads < 0) return (FK_EINVAL);
  // frequency_thread x NTHREADS summed into thread 0's vector from j = 0 (split.c:95-112,536-539):
  // every byte once, read stripe 0 twice
  int64_t freq[256];
  memset(freq, 0, sizeof(freq));
  const int T = ctx->prm.nthreads;
  const int64_t stripe0 = (T > 1) ? ((int64_t) nreads * 1) / T : nreads;
  for (int64_t i = boff[0]; i < boff[nreads]; i++)
    freq[(unsigned char) bases[i]] += 1;
  for (int64_t i = boff[0]; i < boff[stripe0]; i++)
    freq[(unsigned char) bases[i]] += 1;
  const int64_t f4[4] = { freq['a'] + freq['A'], freq['c'] + freq['C'], freq['g'] + freq['G'],
                          freq['t'] + freq['T'] };
  for (int a = 0; a < 4; a++)
    { int rank = 0;
      for (int b = 0; b < 4; b++)
        if (f4[b] < f4[a] || (f4[b] == f4[a] && b < a))
          rank += 1;
      ctx->tran[a] = rank;
    }
  ctx->have_tran = 1;
  return (FK_OK);
}

extern "C" int fk_push_device(fk_ctx *ctx, const void *d_bases, int64_t nbytes)
{ if (ctx == NULL || d_bases == NULL || nbytes < 0) return (FK_EINVAL);
  if (ctx->prm.bc_prefix > 0 || ctx->prm.exact_parts)
    { fk_set_error(ctx, "fk_push_device: -bc and exact_parts need read offsets; use fk_push_block");
      return (FK_EUNSUPPORTED);
    }
  ctx->blocks_bad = true;
  int rc;
  pthread_mutex_lock((pthread_mutex_t *) ctx->push_lock);
  if ((rc = push_form(ctx, 1)) == FK_OK && (rc = device_push_begin(ctx)) == FK_OK && (rc = reserve_reads(ctx, nbytes + 1)) == FK_OK)
    { if (hipMemcpyAsync(ctx->d_reads + ctx->reads_len, d_bases, (size_t) nbytes,
                         hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess
          || hipMemsetAsync(ctx->d_reads + ctx->reads_len + nbytes, 0, 1, ctx->stream) != hipSuccess)
        { fk_set_error(ctx, "fk_push_device: device copy failed");
          rc = FK_EHIP;
        }
      else
        { ctx->reads_len += nbytes + 1;
          if (ctx->chunk_bytes > 0 && ctx->reads_len >= ctx->chunk_bytes)
            rc = fkx_flush_chunk(ctx);
        }
    }
  pthread_mutex_unlock((pthread_mutex_t *) ctx->push_lock);
  return (rc);
}

/* FASTQ text (any piece of a file, cut anywhere) -> reads, parsed on the device (fk_parse.hip).
   *line_phase: 0 before the first byte of a file, carried from call to call. */
extern "C" int fk_push_fastq(fk_ctx *ctx, const char *raw, int64_t nbytes, int flags, int *line_phase,
                             int64_t *nreads, int64_t *nbases)
{ if (ctx == NULL || raw == NULL || nbytes < 0 || line_phase == NULL) return (FK_EINVAL);
  if (ctx->prm.bc_prefix > 0 || ctx->prm.exact_parts)
    { fk_set_error(ctx, "fk_push_fastq: -bc and exact_parts need read offsets; use fk_push_block");
      return (FK_EUNSUPPORTED);
    }
  ctx->blocks_bad = true;
  if (nbytes == 0)
    return (FK_OK);
  int rc;
  pthread_mutex_lock((pthread_mutex_t *) ctx->push_lock);
  do
    { hipSetDevice(ctx->device);
      if ((rc = push_form(ctx, 1)) != FK_OK || (rc = device_push_begin(ctx)) != FK_OK || (rc = reserve_reads(ctx, nbytes + 16)) != FK_OK)
        break;
      void *d_raw = fk_slot(ctx, FK_SLOT_RAW, nbytes + 64);
      if (d_raw == NULL) { rc = FK_ENOMEM; break; }
      // one byte in front of the text: the last byte of the previous piece (homopolymer compression)
      d_raw = (char *) d_raw + 16;
      const unsigned char lastb = (unsigned char) ((*line_phase >> 8) & 0xff);
      // (a stack variable and the caller's text, which may be pageable: blocking copies)
      if ((rc = fkx_h2d_pageable(ctx, ctx->stream, (char *) d_raw - 1, &lastb, 1)) != FK_OK
          || (rc = fkx_h2d_pageable(ctx, ctx->stream, d_raw, raw, (size_t) nbytes)) != FK_OK)
        break;
      int64_t kept = 0, nr = 0;
      if ((rc = fkx_parse_fastq(ctx, d_raw, nbytes, flags, line_phase, ctx->d_reads + ctx->reads_len, &kept, &nr)) != FK_OK)
        break;
      *line_phase = (*line_phase & 3) | ((int) (unsigned char) raw[nbytes - 1] << 8);
      ctx->reads_len += kept;
      if (nreads) *nreads += nr;
      if (nbases) *nbases += kept - nr;
      if (ctx->chunk_bytes > 0 && ctx->reads_len >= ctx->chunk_bytes)
        rc = fkx_flush_chunk(ctx, false, true);
    }
  while (0);
  pthread_mutex_unlock((pthread_mutex_t *) ctx->push_lock);
  return (rc);
}

/* FASTA text (any piece of a file, cut anywhere) -> reads, parsed on the device (fk_parse.hip).
   *state: 2 before the first byte of a file, carried from call to call; last != 0 with the final piece
   of a file (ends its last record). */
extern "C" int fk_push_fasta(fk_ctx *ctx, const char *raw, int64_t nbytes, int last, int *state,
                             int64_t *nreads, int64_t *nbases)
{ if (ctx == NULL || (raw == NULL && nbytes > 0) || nbytes < 0 || state == NULL) return (FK_EINVAL);
  if (ctx->prm.bc_prefix > 0 || ctx->prm.exact_parts)
    { fk_set_error(ctx, "fk_push_fasta: -bc and exact_parts need read offsets; use fk_push_block");
      return (FK_EUNSUPPORTED);
    }
  ctx->blocks_bad = true;
  int rc = FK_OK;
  pthread_mutex_lock((pthread_mutex_t *) ctx->push_lock);
  do
    { hipSetDevice(ctx->device);
      if ((rc = push_form(ctx, 1)) != FK_OK || (rc = device_push_begin(ctx)) != FK_OK || (rc = reserve_reads(ctx, nbytes + 16)) != FK_OK)
        break;
      int64_t kept = 0, nr = 0;
      if (nbytes > 0)
        { void *d_raw = fk_slot(ctx, FK_SLOT_RAW, nbytes + 64);
          if (d_raw == NULL) { rc = FK_ENOMEM; break; }
          if ((rc = fkx_h2d_pageable(ctx, ctx->stream, d_raw, raw, (size_t) nbytes)) != FK_OK)
            break;
          if ((rc = fkx_parse_fasta(ctx, d_raw, nbytes, *state, ctx->d_reads + ctx->reads_len, &kept, &nr)) != FK_OK)
            break;
          // the state after this piece, from the host copy of the text
          int64_t p = nbytes - 1;
          while (p >= 0 && raw[p] != '\n')
            p -= 1;
          if (p >= 0)
            *state = (p == nbytes - 1) ? 2 : (raw[p + 1] == '>' ? 1 : 0);
          else if (*state & 2)
            *state = (raw[0] == '>') ? 1 : 0;
          ctx->reads_len += kept;
        }
      if (last)
        { if (hipMemsetAsync(ctx->d_reads + ctx->reads_len, 0, 1, ctx->stream) != hipSuccess)
            { rc = FK_EHIP; break; }
          ctx->reads_len += 1;
        }
      if (nreads) *nreads += nr;
      if (nbases) *nbases += kept - nr;
      if (ctx->chunk_bytes > 0 && ctx->reads_len >= ctx->chunk_bytes && !last)
        rc = fkx_flush_chunk(ctx, false, true);
    }
  while (0);
  pthread_mutex_unlock((pthread_mutex_t *) ctx->push_lock);
  return (rc);
}

extern "C" int fk_host_alloc(int64_t nbytes, void **ptr)
{ if (ptr == NULL || nbytes <= 0) return (FK_EINVAL);
  if (fkx_pinned_alloc(ptr, nbytes) != FK_OK)
    { fk_set_error(NULL, "fk_host_alloc: cannot pin %lld bytes", (long long) nbytes);
      return (FK_ENOMEM);
    }
  return (FK_OK);
}

extern "C" int fk_host_free(void *ptr)
{ return (fkx_pinned_free(ptr)); }

