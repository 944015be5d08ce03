// fk_ktab_device.hip -- <root>.ktab and its hidden parts written straight from the sorted table in HBM
// (fk_finish_device + fk_write_ktab_device; table.c:162-342,485-498 are the reference's writers), and the copy-rate
// measurement helper.  Split from fk_api.hip.
#include "fk_common.h"
#include <thread>
#include <vector>
#include <chrono>
#include <fcntl.h>
#include <unistd.h>

// lower bounds of the first key byte in a sorted device table: bounds[b] = first record whose byte 0 is >= b
__global__ __launch_bounds__(256) void k_first_byte_bounds(const unsigned char *__restrict__ t, int64_t n, int stride,
                                                           int64_t *__restrict__ bounds)
{ const int b = threadIdx.x;
  int64_t lo = 0, hi = n;
  while (lo < hi)
    { const int64_t mid = (lo + hi) >> 1;
      if (t[mid * stride] < b) lo = mid + 1; else hi = mid;
    }
  bounds[b] = lo;
  if (b == 0) bounds[256] = n;
}


// ends[p] = 1 + the index of the last record whose first ib key bytes spell p (ends zeroed before: 0 = no such record)
__global__ __launch_bounds__(256) void k_prefix_ends(const unsigned char *__restrict__ t, int64_t n, int stride, int ib,
                                                     int64_t *__restrict__ ends)
{ const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (i >= n)
    return;
  u32 a = 0, b = 0;
  for (int j = 0; j < ib; j++)
    { a = (a << 8) | t[i * stride + j];
      if (i + 1 < n) b = (b << 8) | t[(i + 1) * stride + j];
    }
  if (i + 1 == n || a != b)
    ends[a] = i + 1;
}

// cnt table records -> the bytes they take in a .ktab part file (table.c:162-342): the k-mer without its first ib bytes,
// then the count; pw = kb - ib + 2 bytes each.  `out` is pinned HOST memory: what the kernel stores crosses PCIe, so it
// stores nothing narrower than 16 bytes (a wave = 1 KB of consecutive addresses).  A workgroup takes KS_E entries:
// their records come into LDS with 16-byte loads from the aligned address below the first one, every thread then
// gathers the 16 bytes of an output quad from LDS -- the entry of its first byte by one multiply (magic = ceil(2^32 /
// pw); exact for offsets below 2^15 * 16) -- and stores it.  (Round 4's kernel stored a dword per thread after an
// integer division: 1.02 s of kernel time for the 27 GB of configs[2], twice what the link needs.)  The last quad of a
// launch may run up to 15 bytes past cnt * pw: the staging pieces are rounded up to 256 bytes.
#define KS_E 2048
__global__ __launch_bounds__(256) void k_ktab_strip(const unsigned char *__restrict__ t, int64_t cnt, int stride, int ib, int kb,
                                                    u32 magic, unsigned char *__restrict__ out)
{ FK_DYN_LDS_ALIGNED(unsigned char, ks_lds, 16);                             // KS_E * stride + 64 bytes
  const int     pw  = kb - ib + 2, tid = threadIdx.x;
  const int64_t e0  = (int64_t) blockIdx.x * KS_E;
  const int     ne  = (cnt - e0 < KS_E) ? (int) (cnt - e0) : KS_E;
  const unsigned char *src = t + e0 * stride;
  const int     mis = (int) ((uintptr_t) src & 15);
  const uint4  *g   = (const uint4 *) (src - mis);
  const int     nby = ne * stride + mis, n16 = nby >> 4;      // (the loads begin up to 15 bytes below the first record --
  for (int i = tid; i < n16; i += 256)                         //  never below the table: its buffer is 16-byte aligned --
    ((uint4 *) ks_lds)[i] = g[i];                              //  and end with the last record: the tail comes byte by byte)
  if (tid < (nby & 15))
    ks_lds[(n16 << 4) + tid] = ((const unsigned char *) g)[(n16 << 4) + tid];
  __syncthreads();
  const unsigned char *l = ks_lds + mis;
  const int nout = ne * pw, kx = kb - ib;
  uint4 *o4 = (uint4 *) (out + e0 * pw);                    // KS_E * pw is a multiple of 16
  for (int q = tid; q * 16 < nout; q += 256)
    { const u32 o = (u32) q * 16u;
      u32 i = __umulhi(o, magic);
      int j = (int) (o - i * (u32) pw);
      u32 w[4] = { 0u, 0u, 0u, 0u };
#pragma unroll
      for (int b = 0; b < 16; b++)
        { const int at = (int) i * stride + ((j < kx) ? ib + j : stride - 2 + (j - kx));
          w[b >> 2] |= (u32) l[at] << (8 * (b & 3));
          if (++j == pw) { j = 0; i += 1; }
        }
      o4[q] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

#define KTAB_PIECE_BYTES (16ll << 20)       // of table records per piece of a part writer

static int64_t ktab_piece_bytes(const fk_ctx *ctx, int ib)      // bytes of a stripped piece, rounded
{ const int64_t piece = std::max<int64_t>(KTAB_PIECE_BYTES / ctx->wid.kmer_stride, 1);
  return ((piece * (ctx->wid.kmer_word - ib) + 255) & ~255ll);
}

static int ktab_staging(fk_ctx *ctx, int nthreads, int ib)
{ const int64_t need = ktab_piece_bytes(ctx, ib) * 2 * nthreads;
  for (int i = 0; i < 4; i++)
    if (ctx->wstream[i] == NULL)
      { if (fkx_stream_get(ctx->device, &ctx->wstream[i]) != FK_OK)
          { fk_set_error(ctx, "fk_write_ktab_device: no stream for the part writers");
            return (FK_EHIP);
          }
      }
  if (ctx->wstage_cap >= need)
    return (FK_OK);
  if (ctx->h_wstage) hipHostFree(ctx->h_wstage);
  ctx->h_wstage = NULL;
  ctx->wstage_cap = 0;
  if (hipHostMalloc((void **) &ctx->h_wstage, (size_t) need, hipHostMallocDefault) != hipSuccess)
    { (void) hipGetLastError();
      fk_set_error(ctx, "fk_write_ktab_device: out of host memory for the staging of %d writers", nthreads);
      return (FK_ENOMEM);
    }
  ctx->wstage_cap = need;
  return (FK_OK);
}

// The prefix index of the table in HBM (n entries): ctx->ktab_ends[p] = entries whose first ib key bytes spell p.
// The pass needs npre * 8 bytes of device scratch: the read buffer when it is large enough (its reads are counted),
// else a buffer of its own.
static int ktab_prefix_index(fk_ctx *ctx, int64_t n, int ib)
{ int64_t npre = 1;
  for (int i = 0; i < ib; i++) npre *= 256;
  free(ctx->ktab_ends);
  ctx->ktab_ends = (int64_t *) calloc((size_t) npre, 8);
  ctx->ktab_ends_ntab = -1;
  if (ctx->ktab_ends == NULL)
    return (FK_ENOMEM);
  for (int b = 0; b <= 256; b++)
    ctx->ktab_first[b] = 0;
  if (n > 0)
    { int64_t *d_ends = NULL;
      bool own = false;
      if (ctx->d_reads != NULL && ctx->reads_cap >= npre * 8 && ctx->reads_len == 0 && ctx->flush_thread == NULL)
        d_ends = (int64_t *) ctx->d_reads;
      else
        { FK_HIP(ctx, hipMalloc((void **) &d_ends, (size_t) npre * 8));
          own = true;
        }
      hipError_t e = hipMemsetAsync(d_ends, 0, (size_t) npre * 8, ctx->stream);
      if (e == hipSuccess)
        { int64_t *d_b = (int64_t *) ctx->d_scratch;
          hipLaunchKernelGGL(k_first_byte_bounds, dim3(1), dim3(256), 0, ctx->stream, (const unsigned char *) ctx->last_table, n,
                             ctx->wid.kmer_stride, d_b);
          if (hipGetLastError() != hipSuccess || fkx_d2h_pageable(ctx, ctx->stream, ctx->ktab_first, d_b, 257 * 8) != FK_OK)
            e = hipErrorUnknown;                           // (ktab_first lies in the context: pageable)
        }
      if (e == hipSuccess)
        { hipLaunchKernelGGL(k_prefix_ends, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, ctx->stream,
                             (const unsigned char *) ctx->last_table, n, ctx->wid.kmer_stride, ib, d_ends);
          e = hipGetLastError();
        }
      if (e == hipSuccess && fkx_d2h_pageable(ctx, ctx->stream, ctx->ktab_ends, d_ends, (size_t) npre * 8) != FK_OK)
        e = hipErrorUnknown;                               // (calloc'ed)
      if (own)
        hipFree(d_ends);
      if (e != hipSuccess)
        { fk_set_error(ctx, "the prefix pass over the table failed: %s", hipGetErrorString(e));
          return (FK_EHIP);
        }
      int64_t last = 0;                                                   // ends -> entries per prefix
      for (int64_t p = 0; p < npre; p++)
        if (ctx->ktab_ends[p] > 0)
          { const int64_t end = ctx->ktab_ends[p];
            ctx->ktab_ends[p] = end - last;
            last = end;
          }
    }
  ctx->ktab_ends_ntab = n;
  ctx->ktab_ends_ib = ib;
  return (FK_OK);
}


// fk_finish_device: the prefix index, the first-byte bounds and the pinned staging of the part writers, made while
// nothing is being released yet (section 5b of DESIGN.md)
int fkx_ktab_prepare(fk_ctx *ctx, int64_t ntable)
{ const int ib = fk_ktab_idx_bytes(ctx->prm.kmer, ntable);
  int rc = ktab_prefix_index(ctx, ntable, ib);                         // what the .ktab stub holds
  if (rc == FK_OK && ntable > 0)
    rc = ktab_staging(ctx, ctx->prm.nthreads, ib);
  return (rc);
}

/* <root>.ktab + hidden parts straight from the sorted table fk_finish_device left in HBM.  The device makes the file
   bytes: one pass finds where every ib-byte prefix ends (the index of the stub file), and every part has a writer
   thread with a stream of its own that strips its range piece by piece (k_ktab_strip, which stores into one of two
   pinned host buffers) and hands the piece to write() as it is -- the next piece is made and crosses PCIe meanwhile.  The 36 GB
   table of a human-size run never exists in host memory and no host core touches an entry.
   The files are those of fk_write_ktab (table.c:162-342,485-498). */
extern "C" int fk_write_ktab_device(fk_ctx *ctx, const fk_result *res, int nthreads, const char *dir, const char *root)
{ if (ctx == NULL || res == NULL || dir == NULL || root == NULL || nthreads < 1 || nthreads > 256) return (FK_EINVAL);
  const fk_widths &w = ctx->wid;
  const int cutoff = ctx->prm.table_cutoff, kmer = ctx->prm.kmer;
  if (cutoff < 1)
    { fk_set_error(ctx, "fk_write_ktab_device: no table was asked for (table_cutoff 0)");
      return (FK_EINVAL);
    }
  const int64_t n = res->ntable;
  if (n > 0 && (ctx->last_table == NULL || ctx->last_ntab != n))
    { fk_set_error(ctx, "fk_write_ktab_device: the table of this result is not in HBM any more");
      return (FK_ESTATE);
    }
  FK_HIP(ctx, hipSetDevice(ctx->device));
  const double tw0 = fk_wall();
  std::vector<int> split((size_t) nthreads + 1);
  int rc = fk_ktab_split(res->wfirst, kmer, nthreads, split.data());
  if (rc != FK_OK) return (rc);
  const int ib = fk_ktab_idx_bytes(kmer, n);
  const int KW = w.kmer_word, ST = w.kmer_stride, pw = KW - ib;
  int64_t npre = 1;
  for (int i = 0; i < ib; i++) npre *= 256;
  std::vector<int64_t> pc((size_t) npre, 0);
  int64_t hb[257];
  const unsigned char *table = (const unsigned char *) ctx->last_table;
  const int64_t piece = std::max<int64_t>(KTAB_PIECE_BYTES / ST, 1);      // records per piece
  const int64_t pbytes = ktab_piece_bytes(ctx, ib);                       // bytes of a stripped piece (a multiple of 4)
  // Nothing is allocated or asked of the device here when fk_finish_device ran before (prefix index, first-byte
  // bounds, pinned staging): fk_release_device may be returning the rest of the context's HBM in another thread, every
  // hipFree of which holds up the other HIP calls of the process, and a hipMalloc that follows a large hipFree waits
  // until the driver has wiped what was freed (seconds; tools/probe/malloc_probe.cpp).  The strip kernels store into
  // pinned host memory.
  if (ctx->ktab_ends == NULL || ctx->ktab_ends_ntab != n || ctx->ktab_ends_ib != ib)
    { if ((rc = ktab_prefix_index(ctx, n, ib)) != FK_OK)
        return (rc);
    }
  memcpy(pc.data(), ctx->ktab_ends, (size_t) npre * 8);
  memcpy(hb, ctx->ktab_first, sizeof(hb));
  // The pinned staging fk_finish_device made serves ctx->prm.nthreads writers.  More parts than that do not get more
  // staging here -- hipHostFree / hipHostMalloc beside a running fk_release_device is the stall the prepare step
  // exists to avoid -- they take turns at what exists: `lanes` writers run at a time.
  if (n > 0 && ctx->wstage_cap < 2 * pbytes && (rc = ktab_staging(ctx, nthreads, ib)) != FK_OK)
    return (rc);                                       // (no fk_finish_device before: nothing is being released either)
  if (n > 0)
    for (int i = 0; i < 4; i++)
      if (ctx->wstream[i] == NULL)
        { if (fkx_stream_get(ctx->device, &ctx->wstream[i]) != FK_OK)
          { fk_set_error(ctx, "fk_write_ktab_device: no stream for the part writers");
            return (FK_EHIP);
          }
      }
  const int lanes = (n > 0) ? (int) std::min<int64_t>(nthreads, std::max<int64_t>(ctx->wstage_cap / (2 * pbytes), 1)) : nthreads;
  unsigned char *h_stage = ctx->h_wstage;
  const double tw1 = fk_wall();
  std::vector<int> prc((size_t) nthreads, FK_OK);
  auto write_part = [&](int t)
    { const int64_t lo = hb[split[t]], hi = hb[split[t + 1]], cnt = hi - lo;
      char pname[4096];
      snprintf(pname, sizeof(pname), "%s/.%s.ktab.%d", dir, root, t + 1);
      int fd = open(pname, O_WRONLY | O_CREAT | O_TRUNC, 0644);
      if (fd < 0) { prc[t] = FK_EINVAL; return; }
      hipStream_t st = ctx->wstream[t % 4];
      hipEvent_t  ev[2] = { NULL, NULL };
      bool ok = (write(fd, &kmer, 4) == 4 && write(fd, &cnt, 8) == 8);
      if (ok && cnt > 0)
        ok = (hipSetDevice(ctx->device) == hipSuccess
              && fkx_event_get(ctx->device, false, &ev[0]) == FK_OK
              && fkx_event_get(ctx->device, false, &ev[1]) == FK_OK);
      const int ln = t % lanes;                           // (parts t, t + lanes, ... are written one after the other)
      unsigned char *pin[2] = { h_stage + pbytes * (2 * ln), h_stage + pbytes * (2 * ln + 1) };
      auto fetch = [&](int64_t x, int which) -> bool
        { const int64_t m = std::min(hi, x + piece) - x;
          hipLaunchKernelGGL(k_ktab_strip, dim3((unsigned) ((m + KS_E - 1) / KS_E)), dim3(256), (size_t) KS_E * ST + 64, st,
                             table + x * ST, m, ST, ib, (int) w.kmer_bytes, (u32) ((0x100000000ull + (u64) pw - 1) / (u64) pw),
                             pin[which]);                                    // stored across PCIe as it is made
          return (hipGetLastError() == hipSuccess && hipEventRecord(ev[which], st) == hipSuccess);
        };
      int which = 0;
      double t_wait = 0., t_write = 0.;
      if (ok && lo < hi)
        ok = fetch(lo, 0);
      for (int64_t x = lo; ok && x < hi; x += piece, which ^= 1)
        { const int64_t e = std::min(hi, x + piece);
          if (e < hi)
            ok = fetch(e, which ^ 1);                       // the next piece is made and travels while this one is written
          const auto w0 = std::chrono::steady_clock::now();
          if (!ok || hipEventSynchronize(ev[which]) != hipSuccess) { ok = false; break; }
          const auto w1 = std::chrono::steady_clock::now();
          t_wait += std::chrono::duration<double>(w1 - w0).count();
          const unsigned char *q = pin[which];
          size_t left = (size_t) (e - x) * pw;
          while (left > 0)
            { const ssize_t wr = write(fd, q, left);
              if (wr <= 0) { ok = false; break; }
              q += wr; left -= (size_t) wr;
            }
          t_write += std::chrono::duration<double>(std::chrono::steady_clock::now() - w1).count();
        }
      if (ctx->dbg_verbose)
        fprintf(stderr, "  .ktab part %d: %lld entries; waited %.3f s for the device, %.3f s in write()\n", t + 1, (long long) cnt,
                t_wait, t_write);
      if (!ok && cnt > 0)                                 // a piece may still be on its way into the staging: nobody may
        (void) hipStreamSynchronize(st);                  // reuse or free that memory before it has landed
      for (int i = 0; i < 2; i++)
        fkx_event_put(ctx->device, false, &ev[i]);
      if (close(fd) != 0) ok = false;
      if (!ok) prc[t] = FK_EINVAL;
    };
  { auto lane = [&](int l) { for (int t = l; t < nthreads; t += lanes) write_part(t); };
    std::vector<std::thread> th;
    for (int l = 1; l < lanes; l++)
      th.emplace_back(lane, l);
    lane(0);
    for (auto &x : th)
      x.join();
  }
  const double tw2 = fk_wall();
  for (int t = 0; t < nthreads; t++)
    if (prc[t] != FK_OK)
      { fk_set_error(ctx, "Cannot write to %s/.%s.ktab.%d.  Enough disk space?", dir, root, t + 1);
        return (prc[t]);
      }
  rc = fk_write_ktab_stub(kmer, nthreads, cutoff, ib, pc.data(), dir, root);
  if (ctx->dbg_verbose)
    fprintf(stderr, "  fk_write_ktab_device: set-up %.3f s, part writers %.3f s, stub %.3f s\n", tw1 - tw0, tw2 - tw1,
            fk_wall() - tw2);
  return (rc);
}


__global__ __launch_bounds__(256) void k_copy_tile(const uint4 *__restrict__ a, uint4 *__restrict__ b, int64_t n)
{ const int64_t base = (int64_t) blockIdx.x * 1024 + threadIdx.x;
  uint4 v[4];
#pragma unroll
  for (int u = 0; u < 4; u++)
    if (base + u * 256 < n) v[u] = a[base + u * 256];
#pragma unroll
  for (int u = 0; u < 4; u++)
    if (base + u * 256 < n) b[base + u * 256] = v[u];
}

extern "C" int fk_copy_rate(fk_ctx *ctx, void *d_dst, const void *d_src, int64_t nbytes, int reps, double *gbps)
{ if (ctx == NULL || d_dst == NULL || d_src == NULL || nbytes < 16 || reps < 1 || gbps == NULL
      || (((uintptr_t) d_dst | (uintptr_t) d_src) & 15) != 0)
    return (FK_EINVAL);
  FK_HIP(ctx, hipSetDevice(ctx->device));
  const int64_t n = nbytes / 16;
  if ((n + 1023) / 1024 > 0x7fffffffll) return (FK_EINVAL);
  hipEvent_t e0 = NULL, e1 = NULL;
  if (fkx_event_get(ctx->device, true, &e0) != FK_OK || fkx_event_get(ctx->device, true, &e1) != FK_OK)
    { fkx_event_put(ctx->device, true, &e0);
      fk_set_error(ctx, "fk_copy_rate: cannot create events");
      return (FK_EHIP);
    }
  float best = 0.f;
  for (int r = 0; r <= reps; r++)                         // the first run is not timed
    { hipEventRecord(e0, ctx->stream);
      hipLaunchKernelGGL(k_copy_tile, dim3((unsigned) ((n + 1023) / 1024)), dim3(256), 0, ctx->stream, (const uint4 *) d_src,
                         (uint4 *) d_dst, n);
      const hipError_t le = hipGetLastError();
      hipEventRecord(e1, ctx->stream);
      if (le != hipSuccess || hipEventSynchronize(e1) != hipSuccess)
        { fkx_event_put(ctx->device, true, &e0); fkx_event_put(ctx->device, true, &e1);
          fk_set_error(ctx, "fk_copy_rate: the copy kernel failed: %s", hipGetErrorString(le));
          return (FK_EHIP);
        }
      float ms = 0.f;
      hipEventElapsedTime(&ms, e0, e1);
      if (r > 0 && (best == 0.f || ms < best)) best = ms;
    }
  float keep = best;
  fkx_event_put(ctx->device, true, &e0); fkx_event_put(ctx->device, true, &e1);
  best = keep;
  *gbps = 2.0 * (double) (n * 16) / ((double) best * 1e-3) / 1e9;
  return (FK_OK);
}

