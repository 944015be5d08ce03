// fk_pipeline.hip -- the whole path on the device: split -> group / de-duplicate -> expand -> group -> aggregate -> table
// sort, bucket by bucket (fkx_pipeline, count_bucket), and the entry points that run it (fk_finish, fk_finish_device,
// fk_count_device_reads / _packed / _supermers, fk_rounds_*, fk_reset).  Replaces the Sorting + Merge_Tables drivers
// (count.c:1202, table.c:346).  Split from fk_api.hip in round 4, which keeps the context, the stage entry points and
// the utilities.
#include "fk_common.h"

#include <pthread.h>
#include <algorithm>
#include <vector>

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
// A bucket's records gathered from the chunks of a chunked ingest with ONE launch: piece c of the bucket (its records in
// chunk c's slab) is src[c], its place among the bucket's dwords begins at off[c] (off[npieces] = the total); a piece
// that is not in HBM (spilled: copied by hipMemcpyAsync) has src[c] = NULL.  One copy call per chunk and bucket was
// 8,298 calls of ~20 MB at configs[2]: 208 ms of copy kernels too short to reach the memory's rate, and as much again
// in the gaps between them.  Records are 4-byte aligned, not more: dword copies (the L1 rate is not what bounds this).
#define GA_THREADS 256
#define GA_BLOCK   (GA_THREADS * 64)               // dwords of the destination per workgroup

struct __attribute__((packed, aligned(4))) ga_quad { u32 w[4]; };    // 16 bytes at a 4-byte aligned address: one dwordx4

__global__ __launch_bounds__(GA_THREADS) void k_gather_pieces(const u32 *const *__restrict__ src,
                                                              const int64_t *__restrict__ off, int npieces,
                                                              u32 *__restrict__ dst)
{ const int64_t total = off[npieces];
  const int64_t lo = (int64_t) blockIdx.x * GA_BLOCK;
  if (lo >= total)
    return;
  const int64_t hi = (lo + GA_BLOCK < total) ? lo + GA_BLOCK : total;
  int p = 0;                                       // the piece that holds dword lo
  { int a = 0, b = npieces;
    while (a + 1 < b)
      { const int m = (a + b) >> 1;
        if (off[m] <= lo) a = m; else b = m;
      }
    p = a;
  }
  int64_t pend = off[p + 1];
  for (int64_t i = lo + 4 * (int64_t) threadIdx.x; i < hi; i += 4 * GA_THREADS)
    { while (i >= pend)
        pend = off[++p + 1];
      if (i + 4 <= pend && i + 4 <= hi)            // four dwords of one piece
        { const u32 *sp = src[p];
          if (sp != NULL)
            *(ga_quad *) (dst + i) = *(const ga_quad *) (sp + (i - off[p]));
        }
      else                                         // a piece (or the bucket) ends inside: dword by dword
        { int     q = p;
          int64_t qend = pend;
          for (int64_t j = i; j < i + 4 && j < hi; j++)
            { while (j >= qend)
                qend = off[++q + 1];
              const u32 *sp = src[q];
              if (sp != NULL)
                dst[j] = sp[j - off[q]];
            }
        }
    }
}

struct fk_stage_ms { double group_s, expand, radix_k, aggr; };

// dig != NULL: the stream of hash digit 0 of the ns records, written by the splitter beside them.
static int count_bucket(fk_ctx *ctx, void *sm_in, int64_t ns, fk_result *res, bool final,
                        void **table_out, int64_t *ntab, const int64_t *exact_roff, fk_stage_ms *tm,
                        int64_t ns_max = 0, const uint8_t *dig = NULL)
{ const fk_widths &w = ctx->wid;
  hipStream_t s = ctx->stream;
  const int cutoff = ctx->prm.table_cutoff;
  hipEvent_t ev[6] = { NULL, NULL, NULL, NULL, NULL, NULL };      // [4], [5]: around the cut into references and their sort
  int rc = FK_OK;

  if (ns <= 0)
    return (FK_OK);
  for (int i = 0; i < 6; i++)                  // (pooled: no event is ever destroyed, fk_common.h)
    if (fkx_event_get(ctx->device, true, &ev[i]) != FK_OK)
      { fk_set_error(ctx, "fk_finish: cannot create events");
        for (int j = 0; j < i; j++)
          fkx_event_put(ctx->device, true, &ev[j]);
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
          ctx->pre_dig = ctx->dig_lost ? NULL : dig; ctx->pre_dig_n = ns;
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
      // Round 6: no grouping passes over the W records.  The de-duplicated super-mers are cut into the domains of a second
      // minimizer (16 bases), 8-byte references to the pieces are sorted by a key of that minimizer, the expansion walks
      // the references -- all copies of a k-mer then lie inside one key group of the W records as they are written -- and
      // the aggregation's fills are packed from whole groups (fk_recut.hip).  k from 32 to 64 with LDS de-duplication;
      // anything else, fk_debug_set("kmer_stage", 2), or pieces that outrun their room: the hashed grouping as before.
      bool     recut = (dd && nw > 0 && fkx_recut_applies(ctx, nsx));
      u64     *refs = NULL;
      int64_t  nref = 0;
      float    ms_recut = 0.f;
      if (recut)
        { hipEventRecord(ev[4], s);
          rc = fkx_recut(ctx, sm_sorted, nsx, &refs, &nref);
          if (rc == FK_ESTATE)
            { recut = false;
              rc = FK_OK;
              ctx->err[0] = 0;
            }
          else if (rc != FK_OK)
            break;
          else
            { res->passes_kmer      = ctx->sort_stats.passes;
              res->launches_kmer   += ctx->sort_stats.passes;
              res->ms_pass_kmer    += ctx->sort_stats.pass_ms_total;
              res->ms_scatter_kmer += ctx->sort_stats.scatter_ms_total;
              res->nrefs           += nref;
            }
          hipEventRecord(ev[5], s);
          hipEventSynchronize(ev[5]);
          hipEventElapsedTime(&ms_recut, ev[4], ev[5]);
        }
      const u64 *ref_koff = NULL;
      if (nw > 0)
        { // The k-mer slots are sized for the LARGEST bucket at once (this one's k-mers per super-mer applied to the largest
          // bucket's records): hipMalloc and hipFree take well under a millisecond each, but memory a process has freed is
          // wiped by the driver in the background (~34 GB/s) and an allocation that needs it waits -- slots that grew five
          // times over the 57 buckets of configs[2] left 38 GB of such memory behind, and the table sort's 37 GB buffer
          // then took 0.43 s to get (FK_FINISH_TIMING; every run of FastK_amd is a first run).
          int64_t want = nw;
          if (ns_max > ns && ns > 0 && ctx->slot_cap[FK_SLOT_KM_A] < nw * w.kmer_stride)
            { const int64_t big = (int64_t) ((double) nw * ((double) ns_max / (double) ns) * 1.05);
              size_t fr = 0, tot = 0;                      // (only with room to spare: a run that is short of memory grows on demand)
              if (hipMemGetInfo(&fr, &tot) == hipSuccess
                  && (int64_t) fr + ctx->slot_cap[FK_SLOT_KM_A] + ctx->slot_cap[FK_SLOT_KM_B]
                     > 2 * big * w.kmer_stride + (int64_t) tot / 8)
                want = big;
            }
          if (ctx->dbg_verbose)
            fprintf(stderr, "  bucket sizing: %lld records (%lld after de-duplication), largest bucket ~%lld, "
                            "%lld weighted k-mers, buffers for %lld\n", (long long) ns, (long long) nsx,
                    (long long) ns_max, (long long) nw, (long long) want);
          if ((km_a = fk_slot(ctx, FK_SLOT_KM_A, want * w.kmer_stride)) == NULL)
            { rc = FK_ENOMEM; break; }
          if (recut)
            { int64_t nw2 = 0;
              if ((rc = fkx_expand_refs(ctx, sm_sorted, nsx, refs, nref, km_a, want, &nw2, &ovf, &ref_koff)) != FK_OK) break;
              if (nw2 != nw)
                { fk_set_error(ctx, "the pieces hold %lld k-mers, their super-mers %lld", (long long) nw2, (long long) nw);
                  rc = FK_ESTATE;
                  break;
                }
            }
          else
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
          u64     *fills = NULL;
          int64_t  nfills = 0;
          if (recut)
            { int cap_fill = ((w.kmer_stride >> 2) <= 3) ? 8192 : 4096;      // AgCfg<KW>::CAP
              if (ctx->dbg_aggr_limit > 0 && ctx->dbg_aggr_limit < cap_fill)
                cap_fill = ctx->dbg_aggr_limit < 4 ? 4 : ctx->dbg_aggr_limit;
              const int target = std::max(1, cap_fill * 15 / 16);
              if ((rc = fkx_ref_bounds(ctx, refs, nref, ref_koff, nw, target, &fills, &nfills)) != FK_OK)
                break;
            }
          else
            { if ((rc = fkx_group(ctx, nw, km_a, km_b, w.kmer_stride, w.kmer_bytes, 2, &grouped)) != FK_OK)
                break;
              res->passes_kmer      = ctx->sort_stats.passes;
              res->launches_kmer   += ctx->sort_stats.passes;
              res->ms_pass_kmer    += ctx->sort_stats.pass_ms_total;
              res->ms_scatter_kmer += ctx->sort_stats.scatter_ms_total;
            }
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
            { rc = recut ? fkx_aggregate_fills(ctx, grouped, nw, cutoff, res->hist, &res->max_inst, &ndk,
                                               (char *) ctx->slot_ptr[FK_SLOT_TABLE] + *ntab * w.kmer_stride, room, &nt,
                                               fills, nfills)
                         : fkx_aggregate(ctx, grouped, nw, cutoff, res->hist, &res->max_inst, &ndk,
                                 (char *) ctx->slot_ptr[FK_SLOT_TABLE] + *ntab * w.kmer_stride, room, &nt);
              if (rc == FKX_TABLE_FULL)
                room = 0;
              else if (rc == FK_OK)
                direct = true;
            }
          if (room == 0)
            rc = recut ? fkx_aggregate_fills(ctx, grouped, nw, cutoff, res->hist, &res->max_inst, &ndk,
                                             cutoff > 0 ? tbuf : NULL, nw, &nt, fills, nfills)
                       : fkx_aggregate(ctx, grouped, nw, cutoff, res->hist, &res->max_inst, &ndk,
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
                  res->ms_scatter_final += ctx->sort_stats.scatter_ms_total;
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
                  const double wm = fk_wall();
                  const hipError_t me = hipMalloc(&nbuf, (size_t) ncap);
                  if (getenv("FK_FINISH_TIMING") != NULL)
                    fprintf(stderr, "  finish timing: table slot: hipMalloc of %.1f GB took %.3f s\n", (double) ncap / 1e9, fk_wall() - wm);
                  if (me != hipSuccess)
                    { ncap = need + (1 << 20);                // no room for the extrapolation: what is needed now
                      if (hipMalloc(&nbuf, (size_t) ncap) != hipSuccess && ctx->slot_ptr[FK_SLOT_SM_DIG] != NULL)
                        { (void) hipGetLastError();           // ... and the splitter's digit streams go first (fk_slot)
                          hipFree(ctx->slot_ptr[FK_SLOT_SM_DIG]);
                          ctx->slot_ptr[FK_SLOT_SM_DIG] = NULL;
                          ctx->slot_cap[FK_SLOT_SM_DIG] = 0;
                          ctx->dig_lost = true;
                        }
                      if (nbuf == NULL && hipMalloc(&nbuf, (size_t) ncap) != hipSuccess)
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
      tm->expand  += ms_between(ev[1], ev[2]) - ms_recut;
      tm->radix_k += ms_between(ev[2], ev[3]) - ms_aggr + ms_recut;      // (the cut into references and their sort: the k-mer "sort")
      tm->aggr    += ms_aggr;
    }
  while (0);
  for (int i = 0; i < 6; i++)
    fkx_event_put(ctx->device, true, &ev[i]);
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
  hipEvent_t te[2] = { NULL, NULL };             // (the sort itself records ctx->ev0 / ev1)
  if (fkx_event_get(ctx->device, true, &te[0]) != FK_OK || fkx_event_get(ctx->device, true, &te[1]) != FK_OK)
    { fkx_event_put(ctx->device, true, &te[0]);
      return (FK_EHIP);
    }
  hipEventRecord(te[0], s);
  *table = ctx->slot_ptr[FK_SLOT_TABLE];
  int rc = fkx_sort_table(ctx, ntab, ctx->slot_ptr[FK_SLOT_TABLE], tmp, table, census);
  if (rc != FK_OK)
    { fkx_event_put(ctx->device, true, &te[0]); fkx_event_put(ctx->device, true, &te[1]);
      return (rc);
    }
  res->passes_final   = ctx->sort_stats.passes;
  res->ms_pass_final += ctx->sort_stats.pass_ms_total;
  res->ms_scatter_final += ctx->sort_stats.scatter_ms_total;
  for (int x = 0; x < 256; x++)
    res->wfirst[x] = census[x];
  hipEventRecord(te[1], s);
  hipEventSynchronize(te[1]);
  tm->radix_k += ms_between(te[0], te[1]);
  res->ms_table_sort += ms_between(te[0], te[1]);
  fkx_event_put(ctx->device, true, &te[0]); fkx_event_put(ctx->device, true, &te[1]);
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
  if (fkx_reserve_host_table(ctx, bytes) != FK_OK)
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
  hipEvent_t ev[3] = { NULL, NULL, NULL };
  int rc = FK_OK;

  memset(res, 0, sizeof(*res));
  for (int i = 0; i < 3; i++)
    if (fkx_event_get(ctx->device, true, &ev[i]) != FK_OK)
      { fk_set_error(ctx, "fk_finish: cannot create events");
        for (int j = 0; j < i; j++)
          fkx_event_put(ctx->device, true, &ev[j]);
        return (FK_EHIP);
      }
  do
    { void *sm_a = NULL;
      uint8_t *sm_dig = NULL;            // first digit stream of the super-mer grouping, from a one-pass split
      ctx->dig_lost = false;
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
              if ((rc = fkx_h2d_pageable(ctx, s, d_roff, h_roff, (size_t) (nreads + 1) * 8)) != FK_OK)
                break;
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
      bool gather_tab_sent = false;
      auto gather = [&](int b, void **ptr) -> int
        { char *g = (char *) fk_slot(ctx, FK_SLOT_SM_G, ns_max * w.smer_stride);
          if (g == NULL)
            return (FK_ENOMEM);
          // the pieces in HBM with one kernel (tables of all buckets: made and sent once, on the first call)
          const int     nc = ctx->nchunks;
          const size_t  tb = (size_t) nc * 8 + (size_t) (nc + 1) * 8;           // per bucket: nc sources, nc + 1 offsets
          char *d_tab = (char *) fk_slot(ctx, FK_SLOT_GATHER, (int64_t) tb * nbk);
          if (d_tab == NULL)
            return (FK_ENOMEM);
          if (!gather_tab_sent)
            { char *h = (char *) malloc(tb * nbk);
              if (h == NULL)
                return (FK_ENOMEM);
              for (int x = 0; x < nbk; x++)
                { const void **hs = (const void **) (h + tb * x);
                  int64_t     *ho = (int64_t *) (h + tb * x + (size_t) nc * 8);
                  int64_t r = 0;
                  for (int c = 0; c < nc; c++)
                    { const fk_chunk *ch = &ctx->chunks[c];
                      hs[c] = (ch->on_host || ch->cnt[x] == 0) ? NULL : ch->run[x];
                      ho[c] = r;
                      r += ch->cnt[x] * (w.smer_stride / 4);
                    }
                  ho[nc] = r;
                }
              const hipError_t e = hipMemcpyAsync(d_tab, h, tb * nbk, hipMemcpyHostToDevice, s);
              const hipError_t e2 = hipStreamSynchronize(s);      // (pageable source: gone when this returns)
              free(h);
              if (e != hipSuccess || e2 != hipSuccess)
                return (FK_EHIP);
              gather_tab_sent = true;
            }
          int64_t run = 0;
          for (int c = 0; c < nc; c++)
            { const fk_chunk *ch = &ctx->chunks[c];
              if (ch->cnt[b] > 0 && ch->on_host
                  && hipMemcpyAsync(g + run * w.smer_stride, ch->run[b], (size_t) (ch->cnt[b] * w.smer_stride),
                                    hipMemcpyHostToDevice, s) != hipSuccess)
                return (FK_EHIP);
              run += ch->cnt[b];
            }
          if (run > 0)
            { const int64_t ndw = run * (w.smer_stride / 4);
              hipLaunchKernelGGL(k_gather_pieces, dim3((unsigned) ((ndw + GA_BLOCK - 1) / GA_BLOCK)), dim3(GA_THREADS), 0, s,
                                 (const u32 *const *) (d_tab + tb * b), (const int64_t *) (d_tab + tb * b + (size_t) nc * 8),
                                 nc, (u32 *) g);
              if (hipGetLastError() != hipSuccess)
                return (FK_EHIP);
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
          hipEvent_t gev[2] = { NULL, NULL };
          if (fkx_event_get(ctx->device, true, &gev[0]) != FK_OK || fkx_event_get(ctx->device, true, &gev[1]) != FK_OK)
            { fkx_event_put(ctx->device, true, &gev[0]);
              rc = FK_EHIP; break;
            }
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
                  sm_dig = (w.smer_stride == 20) ? fkx_dig_slot(ctx, ctx->slot_cap[FK_SLOT_SM_A] / w.smer_stride) : NULL;
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
          fkx_event_put(ctx->device, true, &gev[0]);
          fkx_event_put(ctx->device, true, &gev[1]);
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
    fkx_event_put(ctx->device, true, &ev[i]);
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
