// fk_profile_api.hip -- C-ABI entry points of the profile stage (fk_profile.hip holds the kernels).
#include "fk_common.h"
#include "../../include/fk_synth.h"
#include <pthread.h>
#include <stdarg.h>
#include <fcntl.h>
#include <unistd.h>
#include <algorithm>
#include <vector>
#include <thread>

// ---- profiles (-p) ---------------------------------------------------------------------------

// compressed profiles + offsets from HBM into the context's host buffers, in the data set's read order
// the codec on the host (README.md:1029-1069; same streams as k_pf_encode), for reads that arrived in pieces
static void fk_profile_decode_append(const uint8_t *b, int64_t len, std::vector<uint16_t> &out)
{ if (len <= 0) return;
  int64_t i;
  uint32_t c;
  if (b[0] & 0x80) { c = ((uint32_t) (b[0] & 0x7f) << 8) | b[1]; i = 2; }
  else             { c = b[0]; i = 1; }
  out.push_back((uint16_t) c);
  while (i < len)
    { const uint32_t x = b[i];
      if (x & 0x80)
        { c = (c + ((((x & 0x7f) << 8) | b[i + 1]))) & 0x7fff;
          out.push_back((uint16_t) c);
          i += 2;
        }
      else if (x & 0x40)
        { int d = (int) (x & 0x3f);
          if (d & 0x20) d -= 0x40;
          c = (uint32_t) ((int) c + d) & 0xffff;
          out.push_back((uint16_t) c);
          i += 1;
        }
      else
        { out.insert(out.end(), (size_t) x, (uint16_t) c);
          i += 1;
        }
    }
}

static void fk_profile_encode_append(const std::vector<uint16_t> &c, std::vector<uint8_t> &o)
{ if (c.empty()) return;
  uint32_t prev = c[0], run = 0;
  if (prev < 128) o.push_back((uint8_t) prev);
  else { o.push_back((uint8_t) (0x80 | (prev >> 8))); o.push_back((uint8_t) prev); }
  for (size_t j = 1; j < c.size(); j++)
    { const uint32_t x = c[j];
      if (x == prev)
        { if (++run == 63) { o.push_back(63); run = 0; }
          continue;
        }
      if (run) { o.push_back((uint8_t) run); run = 0; }
      const int d = (int) x - (int) prev;
      if (d > -32 && d < 32)
        o.push_back((uint8_t) (0x40 | (d & 0x3f)));
      else
        { const uint32_t dd = (uint32_t) d & 0x7fff;
          o.push_back((uint8_t) (0x80 | (dd >> 8)));
          o.push_back((uint8_t) dd);
        }
      prev = x;
    }
  if (run) o.push_back((uint8_t) run);
}

static int profiles_to_host(fk_ctx *ctx, int64_t *nreads_io, int64_t *nprof_io, void *d_data, uint64_t *d_offs,
                            bool own_reads)
{ const int64_t nreads = *nreads_io, nprof = *nprof_io;
  if (ctx->h_prof_cap < nprof + 1)
    { free(ctx->h_prof);
      ctx->h_prof = (uint8_t *) malloc((size_t) nprof + 1);
      ctx->h_prof_cap = nprof + 1;
      if (ctx->h_prof == NULL) { ctx->h_prof_cap = 0; return (FK_ENOMEM); }
    }
  if (ctx->h_prof_off_cap < nreads + 1)
    { free(ctx->h_prof_off);
      ctx->h_prof_off = (int64_t *) malloc(sizeof(int64_t) * (size_t) (nreads + 1));
      ctx->h_prof_off_cap = nreads + 1;
      if (ctx->h_prof_off == NULL) { ctx->h_prof_off_cap = 0; return (FK_ENOMEM); }
    }
  ctx->h_prof_off[0] = 0;
  if (nreads > 0)
    { int hrc = fkx_d2h_pageable(ctx, ctx->stream, ctx->h_prof, d_data, (size_t) nprof);       // (both malloc'ed)
      if (hrc == FK_OK)
        hrc = fkx_d2h_pageable(ctx, ctx->stream, ctx->h_prof_off, d_offs, (size_t) (nreads + 1) * 8);
      if (hrc != FK_OK)
        return (hrc);
    }
  // Blocks pushed by several input threads interleave in HBM; the data set's read order is thread 0's
  // reads, then thread 1's, ... (io.c gives every thread a contiguous range of the input), and the
  // reference's part files are exactly those ranges.
  ctx->h_prof_nsplit = 0;
  if (own_reads && !ctx->blocks_bad && ctx->nblocks > 0)
    { int64_t tot = 0;
      int     maxtid = 0;
      bool    sorted = true, joins = false;
      for (int64_t b = 0; b < ctx->nblocks; b++)
        { tot += ctx->blocks[b].nreads;
          if (ctx->blocks[b].tid > maxtid) maxtid = ctx->blocks[b].tid;
          if (b > 0 && ctx->blocks[b].tid < ctx->blocks[b - 1].tid) sorted = false;
          if (ctx->blocks[b].tid < 0) tot = -1 - nreads;
          if (ctx->blocks[b].rem > 0) joins = true;
        }
      if (tot == nreads && maxtid < 4096)
        { const int nt = maxtid + 1;
          free(ctx->h_prof_split);
          ctx->h_prof_split = (int64_t *) calloc((size_t) nt + 1, sizeof(int64_t));
          if (ctx->h_prof_split == NULL) return (FK_ENOMEM);
          ctx->h_prof_nsplit = nt;
          if (sorted && !joins)
            { for (int64_t b = 0; b < ctx->nblocks; b++)
                ctx->h_prof_split[ctx->blocks[b].tid + 1] += ctx->blocks[b].nreads;
              for (int t = 0; t < nt; t++)
                ctx->h_prof_split[t + 1] += ctx->h_prof_split[t];
            }
          else
            { // blocks in data-set order: by thread, then in push order; a block pushed with rem > 0 ends
              // in a read that goes on as the first read of the thread's next block (which repeats its
              // last K-1 bases, io.c:557-570): the two profiles are one read's -- their counts are
              // concatenated and encoded again
              std::vector<int64_t> first((size_t) ctx->nblocks), order;
              { int64_t r = 0;
                for (int64_t b = 0; b < ctx->nblocks; b++)
                  { first[(size_t) b] = r;
                    r += ctx->blocks[b].nreads;
                  }
              }
              for (int t = 0; t < nt; t++)
                for (int64_t b = 0; b < ctx->nblocks; b++)
                  if (ctx->blocks[b].tid == t && ctx->blocks[b].nreads > 0)
                    order.push_back(b);
              std::vector<uint8_t> nd;
              std::vector<int64_t> no(1, 0);
              std::vector<uint16_t> chain;
              bool open_chain = false;
              nd.reserve((size_t) nprof + 16);
              no.reserve((size_t) nreads + 1);
              for (size_t oi = 0; oi < order.size(); oi++)
                { const int64_t b = order[oi];
                  const int     t = ctx->blocks[b].tid;
                  const bool goes_on = ctx->blocks[b].rem > 0 && oi + 1 < order.size()
                                       && ctx->blocks[order[oi + 1]].tid == t;
                  for (int64_t i = 0; i < ctx->blocks[b].nreads; i++)
                    { const int64_t r  = first[(size_t) b] + i;
                      const uint8_t *p = ctx->h_prof + ctx->h_prof_off[r];
                      const int64_t  l = ctx->h_prof_off[r + 1] - ctx->h_prof_off[r];
                      const bool last  = (i == ctx->blocks[b].nreads - 1);
                      if (open_chain || (last && goes_on))
                        { fk_profile_decode_append(p, l, chain);
                          open_chain = true;
                          if (!(last && goes_on))              // the read ends here
                            { fk_profile_encode_append(chain, nd);
                              no.push_back((int64_t) nd.size());
                              chain.clear();
                              open_chain = false;
                              ctx->h_prof_split[t + 1] += 1;
                            }
                        }
                      else
                        { nd.insert(nd.end(), p, p + l);
                          no.push_back((int64_t) nd.size());
                          ctx->h_prof_split[t + 1] += 1;
                        }
                    }
                  if (open_chain && !goes_on)                  // (a dangling piece: close it as it is)
                    { fk_profile_encode_append(chain, nd);
                      no.push_back((int64_t) nd.size());
                      chain.clear();
                      open_chain = false;
                      ctx->h_prof_split[t + 1] += 1;
                    }
                }
              for (int t = 0; t < nt; t++)
                ctx->h_prof_split[t + 1] += ctx->h_prof_split[t];
              *nreads_io = (int64_t) no.size() - 1;
              *nprof_io  = (int64_t) nd.size();
              uint8_t *bd = (uint8_t *) malloc(nd.size() + 1);
              int64_t *bo = (int64_t *) malloc(sizeof(int64_t) * no.size());
              if (bd == NULL || bo == NULL) { free(bd); free(bo); return (FK_ENOMEM); }
              memcpy(bd, nd.data(), nd.size());
              memcpy(bo, no.data(), sizeof(int64_t) * no.size());
              free(ctx->h_prof);     ctx->h_prof = bd;      ctx->h_prof_cap = (int64_t) nd.size() + 1;
              free(ctx->h_prof_off); ctx->h_prof_off = bo;  ctx->h_prof_off_cap = (int64_t) no.size();
            }
        }
    }
  return (FK_OK);
}

extern "C" int fk_set_table(fk_ctx *ctx, const uint8_t *records, int64_t n)
{ if (ctx == NULL || n < 0 || (records == NULL && n > 0)) return (FK_EINVAL);
  const fk_widths &w = ctx->wid;
  hipStream_t s = ctx->stream;
  FK_HIP(ctx, hipSetDevice(ctx->device));
  ctx->have_table = false;
  ctx->have_part_table = false;
  ctx->last_table = NULL;
  ctx->pf_dict_table = NULL;
  ctx->last_ntab  = 0;
  if (n > 0)
    { void *d_t = fk_slot(ctx, FK_SLOT_TABLE, n * w.kmer_stride);
      if (d_t == NULL)
        return (FK_ENOMEM);
      if (w.kmer_word == w.kmer_stride)
        { const int hrc = fkx_h2d_pageable(ctx, s, d_t, records, (size_t) n * w.kmer_stride);       // (the caller's memory)
          if (hrc != FK_OK) return (hrc);
        }
      else
        { std::vector<uint8_t> stage((size_t) n * w.kmer_stride, 0);
          for (int64_t i = 0; i < n; i++)
            { memcpy(stage.data() + i * w.kmer_stride, records + i * w.kmer_word, w.kmer_bytes);
              memcpy(stage.data() + i * w.kmer_stride + w.kmer_stride - 2, records + i * w.kmer_word + w.kmer_bytes, 2);
            }
          const int hrc = fkx_h2d_pageable(ctx, s, d_t, stage.data(), stage.size());
          if (hrc != FK_OK) return (hrc);
        }
      FK_HIP(ctx, hipStreamSynchronize(s));
      ctx->last_table = d_t;
      ctx->pf_dict_table = NULL;                   // the look-ups hash the records: no order needed
    }
  ctx->last_ntab  = n;
  ctx->have_table = true;
  return (FK_OK);
}

extern "C" int fk_make_profiles(fk_ctx *ctx, const void *d_bases, int64_t nbytes, fk_profiles *out)
{ if (ctx == NULL || out == NULL || nbytes < 0) return (FK_EINVAL);
  memset(out, 0, sizeof(*out));
  FK_HIP(ctx, hipSetDevice(ctx->device));
  if (!ctx->have_table)
    { fk_set_error(ctx, "fk_make_profiles: needs fk_set_table or the table of a finished resident run with table_cutoff 1");
      return (FK_ESTATE);
    }
  const bool own_reads = (d_bases == NULL);
  if (((uintptr_t) d_bases & 15) != 0)
    { fk_set_error(ctx, "fk_make_profiles: d_bases must be 16-byte aligned");
      return (FK_EINVAL);
    }
  if (d_bases == NULL)
    { if (ctx->chunk_bytes > 0)
        { fk_set_error(ctx, "fk_make_profiles: the reads of a chunked run are not kept -- pass them again piece by piece");
          return (FK_ESTATE);
        }
      d_bases = ctx->d_reads;
      nbytes  = ctx->reads_len;
      if (ctx->push_form == 2)                  // pushed in two bits per base: the look-up kernels walk ASCII reads
        { void *asc = NULL;
          const int rcu = fkx_unpack_store(ctx, &asc, &nbytes);
          if (rcu != FK_OK)
            return (rcu);
          d_bases = asc;
        }
    }
  int64_t nreads = 0, nprof = 0;
  void *d_data = NULL;
  uint64_t *d_offs = NULL;
  ctx->pf_own_reads = own_reads;
  int rc = fkx_profiles(ctx, d_bases, nbytes, ctx->last_table, ctx->last_ntab, &nreads, &nprof, &d_data, &d_offs);
  ctx->pf_own_reads = false;
  if (rc != FK_OK)
    return (rc);
  if ((rc = profiles_to_host(ctx, &nreads, &nprof, d_data, d_offs, own_reads)) != FK_OK)
    return (rc);
  out->nreads  = nreads;
  out->nbytes  = nprof;
  out->data    = ctx->h_prof;
  out->offsets = ctx->h_prof_off;
  out->nsplit  = ctx->h_prof_nsplit;
  out->split   = ctx->h_prof_nsplit > 0 ? ctx->h_prof_split : NULL;
  return (FK_OK);
}

// ---- profiles in the sharded run: look-ups on the owning rank, counts sent back ----------------------

extern "C" int fk_split_supermers_emit_pos(fk_ctx *ctx, const void *d_bases, int64_t nbytes, void *d_out,
                                           int64_t cap, const int64_t *bucket_counts, void *d_pos)
{ if (ctx == NULL || d_bases == NULL || d_out == NULL || bucket_counts == NULL || d_pos == NULL || nbytes < 0)
    return (FK_EINVAL);
  if (((uintptr_t) d_bases & 15) != 0)
    { fk_set_error(ctx, "fk_split_supermers_emit_pos: read buffer must be 16-byte aligned");
      return (FK_EINVAL);
    }
  int64_t bc[256], ns = 0, ni = 0;
  for (int b = 0; b < ctx->prm.nbuckets; b++)
    { bc[b] = bucket_counts[b];
      ns += bc[b];
    }
  if (cap < ns)
    { fk_set_error(ctx, "fk_split_supermers_emit_pos: buffer holds %lld records, %lld needed",
                   (long long) cap, (long long) ns);
      return (FK_EINVAL);
    }
  if (ns == 0)
    return (FK_OK);
  FK_HIP(ctx, hipSetDevice(ctx->device));
  return fkx_split(ctx, d_bases, nbytes, d_out, cap, &ns, &ni, bc, true, d_pos);
}

extern "C" int fk_profile_lookup_supermers(fk_ctx *ctx, const void *d_smers, int64_t nsuper, void *d_counts,
                                           int64_t cap, int64_t *ninst)
{ if (ctx == NULL || ninst == NULL || nsuper < 0 || (nsuper > 0 && d_smers == NULL)) return (FK_EINVAL);
  FK_HIP(ctx, hipSetDevice(ctx->device));
  if (!ctx->have_part_table && !ctx->have_table)
    { fk_set_error(ctx, "fk_profile_lookup_supermers: needs the table of a finished run with table_cutoff 1");
      return (FK_ESTATE);
    }
  return fkx_profile_lookup_supermers(ctx, d_smers, nsuper, ctx->last_table, ctx->last_ntab, d_counts, cap, ninst);
}

extern "C" int fk_profile_scatter(fk_ctx *ctx, const void *d_smers, const void *d_pos, int64_t nsuper,
                                  const void *d_counts, int64_t nbytes, int reset)
{ if (ctx == NULL || nsuper < 0 || nbytes < 0 || (nsuper > 0 && (d_smers == NULL || d_pos == NULL || d_counts == NULL)))
    return (FK_EINVAL);
  FK_HIP(ctx, hipSetDevice(ctx->device));
  return fkx_profile_scatter(ctx, d_smers, d_pos, nsuper, d_counts, nbytes, reset != 0);
}

extern "C" int fk_profile_encode(fk_ctx *ctx, const void *d_bases, int64_t nbytes, fk_profiles *out)
{ if (ctx == NULL || out == NULL || d_bases == NULL || nbytes < 0) return (FK_EINVAL);
  memset(out, 0, sizeof(*out));
  FK_HIP(ctx, hipSetDevice(ctx->device));
  if (((uintptr_t) d_bases & 15) != 0)
    { fk_set_error(ctx, "fk_profile_encode: d_bases must be 16-byte aligned");
      return (FK_EINVAL);
    }
  int64_t nreads = 0, nprof = 0;
  void *d_data = NULL;
  uint64_t *d_offs = NULL;
  int rc = fkx_profile_encode_counts(ctx, d_bases, nbytes, &nreads, &nprof, &d_data, &d_offs);
  if (rc != FK_OK)
    return (rc);
  ctx->h_prof_nsplit = 0;
  if ((rc = profiles_to_host(ctx, &nreads, &nprof, d_data, d_offs, false)) != FK_OK)
    return (rc);
  out->nreads  = nreads;
  out->nbytes  = nprof;
  out->data    = ctx->h_prof;
  out->offsets = ctx->h_prof_off;
  return (FK_OK);
}

