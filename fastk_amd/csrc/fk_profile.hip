// fk_profile.hip -- read profiles (FastK -p): for every read the count of each of its k-mers, in read
// order, compressed with the reference's profile codec (README "K-mer Profile Files"; the reference
// builds them in count.c:868-947 from its sorted super-mer lists and stitches them in merge.c).
//
// Here the profiles are produced from the two things the counting path leaves in HBM: the reads and
// the sorted k-mer table with cutoff 1.
//   k_pf_index   24-bit prefix index over the sorted table
//   k_pf_counts  one lookup per base position: canonical k-mer from LDS-packed 2-bit codes, binary
//                search inside the prefix bucket -> u16 count (0 where the window is not all acgt)
//   k_pf_zeros   positions of the read terminators (0 bytes) -> read boundaries
//   k_pf_encode  one thread per read: length pass, then emit pass of the codec
// The codec output is the canonical one-byte-wherever-possible stream; it decodes to the same counts
// as the reference's files, whose run splits follow the reference's internal work panels (merge.c:65,711).
#include "fk_common.h"

#define PF_TILE  4096
#define PF_HALO  128          // >= kmer - 1, multiple of 16
#define PF_NG    ((PF_TILE + PF_HALO) / 16)
#define PF_ZCH   16384        // bytes per workgroup of k_pf_zeros

// ---------------------------------------------------------------------------------------
// prefix index: idx[p] = first table record whose leading key bits (>> pshift of the first
// big-endian key word) are >= p, idx[NP] = nt

static __global__ __launch_bounds__(256) void k_pf_index(const u32 *__restrict__ table, int64_t nt, int sdw,
                                                          int pshift, int64_t NP, u64 *__restrict__ idx)
{ const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (i > nt)
    return;
  const int64_t pprev = (i == 0) ? -1 : (int64_t) (__builtin_bswap32(table[(i - 1) * sdw]) >> pshift);
  const int64_t pcur  = (i == nt) ? NP : (int64_t) (__builtin_bswap32(table[i * sdw]) >> pshift);
  for (int64_t p = pprev + 1; p <= pcur; p++)
    idx[p] = (u64) i;
}

// ---------------------------------------------------------------------------------------
// per-position counts

__device__ __forceinline__ u32 pf_rev2(u32 x)      // reverse the order of the 16 2-bit groups
{ u32 y = __brev(x);
  return (((y & 0x55555555u) << 1) | ((y >> 1) & 0x55555555u));
}

template <int KW>
static __global__ __launch_bounds__(256) void k_pf_counts(const uint8_t *__restrict__ bases, int64_t n, int K,
                                                           const u32 *__restrict__ table, int sdw, int kbytes,
                                                           const u64 *__restrict__ idx, int pshift,
                                                           uint16_t *__restrict__ out)
{ __shared__ u32      fw[PF_NG + KW + 2];       // 2-bit codes, 16 per word, first base in the top bits
  __shared__ __attribute__((aligned(16))) uint16_t bad[PF_NG + 16];          // bit j of bad[g]: base 16 g + j is not acgt (or past n)
  const int     tid = threadIdx.x;
  const int64_t t0  = (int64_t) blockIdx.x * PF_TILE;

  for (int g = tid; g < PF_NG + KW + 2; g += 256)
    { const int64_t p = t0 + (int64_t) g * 16;
      u32 word = 0, bm = 0;
      if (g < PF_NG && p < n)
        { uint8_t c[16];
          if (p + 16 <= n)
            { const uint4 v = *(const uint4 *) (bases + p);
              const u32 d[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
              for (int j = 0; j < 16; j++)
                c[j] = (uint8_t) (d[j >> 2] >> (8 * (j & 3)));
            }
          else
            {
#pragma unroll
              for (int j = 0; j < 16; j++)
                c[j] = (p + j < n) ? bases[p + j] : (uint8_t) 0;
            }
#pragma unroll
          for (int j = 0; j < 16; j++)
            { const u32 l = c[j] | 0x20u;
              const bool ok = (l == 'a' || l == 'c' || l == 'g' || l == 't');
              const u32 x = (c[j] >> 1) & 3u;
              word |= (x ^ (x >> 1)) << (30 - 2 * j);
              bm   |= ok ? 0u : (1u << j);
            }
        }
      else
        bm = 0xffffu;
      fw[g] = word;
      if (g < PF_NG + 16)
        bad[g] = (uint16_t) bm;
    }
  __syncthreads();

  const int tb = K - 16 * (KW - 1);             // bases in the last key word
  const u32 tmask = (tb == 16) ? 0xffffffffu : ~(0xffffffffu >> (2 * tb));
  const int ps = 2 * (16 - tb);
  const int kb_last = kbytes - 4 * (KW - 1);    // key bytes in the last record word
  const u32 rmask = (kb_last == 4) ? 0xffffffffu : ~(0xffffffffu >> (8 * kb_last));
  const u32 *bad32 = (const u32 *) bad;

#pragma unroll 1
  for (int j = 0; j < PF_TILE / 256; j++)
    { const int     i = tid + j * 256;
      const int64_t p = t0 + i;
      if (p >= n)
        break;
      // any bad base in [i, i+K) ?
      bool ok = (p + K <= n);
      { const int w0 = i >> 5, w1 = (i + K - 1) >> 5;
        u32 acc = 0;
        for (int w = w0; w <= w1; w++)
          { u32 m = bad32[w];
            if (w == w0) m &= 0xffffffffu << (i & 31);
            if (w == w1) m &= 0xffffffffu >> (31 - ((i + K - 1) & 31));
            acc |= m;
          }
        ok = ok && (acc == 0);
      }
      u32 cnt = 0;
      if (ok)
        { const int q = i >> 4, sh = (i & 15) * 2;
          u32 a[KW], b[KW + 1];
#pragma unroll
          for (int w = 0; w < KW; w++)
            a[w] = (u32) (((((u64) fw[q + w]) << 32) | fw[q + w + 1]) << sh >> 32);
          a[KW - 1] &= tmask;
#pragma unroll
          for (int w = 0; w < KW; w++)
            b[w] = pf_rev2(~a[KW - 1 - w]);
          b[KW] = 0;
#pragma unroll
          for (int w = 0; w < KW; w++)
            b[w] = (u32) (((((u64) b[w]) << 32) | b[w + 1]) << ps >> 32);
          b[KW - 1] &= tmask;
          bool rc_less = false, decided = false;
#pragma unroll
          for (int w = 0; w < KW; w++)
            if (!decided && a[w] != b[w])
              { rc_less = b[w] < a[w];
                decided = true;
              }
          if (rc_less)
            {
#pragma unroll
              for (int w = 0; w < KW; w++)
                a[w] = b[w];
            }
          const u32 pre = a[0] >> pshift;
          int64_t lo = (int64_t) idx[pre], hi = (int64_t) idx[pre + 1];
          // -1 / 0 / +1: record m is below / equal to / above the key
          auto probe = [&](int64_t m) -> int
            { const u32 *r = table + m * sdw;
              int cmp = 0;
#pragma unroll
              for (int w = 0; w < KW; w++)
                { u32 x = __builtin_bswap32(r[w]);
                  if (w == KW - 1) x &= rmask;
                  if (cmp == 0 && x != a[w])
                    cmp = (x < a[w]) ? -1 : 1;
                }
              if (cmp == 0)
                cnt = r[sdw - 1] >> 16;
              return (cmp);
            };
          bool found = false;
          if (hi - lo > 8)
            { // the bits after the prefix are close to uniform inside a bucket: start at the interpolated
              // rank and gallop, so that the probes stay within a cache line or two of the answer
              const u32 nxt = (a[0] << (32 - pshift)) | ((KW > 1) ? (a[KW > 1 ? 1 : 0] >> pshift) : 0u);
              int64_t e = lo + (int64_t) (((u64) (hi - lo) * nxt) >> 32);
              int c = probe(e);
              if (c == 0)
                found = true;
              else if (c < 0)
                { lo = e + 1;
                  for (int64_t step = 2; ; step <<= 1)
                    { const int64_t m = lo + step - 1;
                      if (m >= hi) break;
                      c = probe(m);
                      if (c == 0) { found = true; break; }
                      if (c > 0) { hi = m; break; }
                      lo = m + 1;
                    }
                }
              else
                { hi = e;
                  for (int64_t step = 2; ; step <<= 1)
                    { const int64_t m = hi - step;
                      if (m < lo) break;
                      c = probe(m);
                      if (c == 0) { found = true; break; }
                      if (c < 0) { lo = m + 1; break; }
                      hi = m;
                    }
                }
            }
          while (!found && lo < hi)
            { const int64_t mid = (lo + hi) >> 1;
              const int c = probe(mid);
              if (c == 0) break;
              if (c < 0) lo = mid + 1;
              else       hi = mid;
            }
        }
      out[p] = (uint16_t) cnt;
    }
}

// ---------------------------------------------------------------------------------------
// read boundaries: positions of the 0 bytes

template <bool EMIT>
static __global__ __launch_bounds__(256) void k_pf_zeros(const uint8_t *__restrict__ bases, int64_t n,
                                                          u32 *__restrict__ cnt, const u64 *__restrict__ off,
                                                          int64_t *__restrict__ ends)
{ __shared__ u32 tmp[8];
  const int64_t p0 = (int64_t) blockIdx.x * PF_ZCH + (int64_t) threadIdx.x * (PF_ZCH / 256);
  u64 zm = 0;                                  // PF_ZCH / 256 = 64 bytes per thread
#pragma unroll
  for (int k = 0; k < 4; k++)
    { const int64_t p = p0 + 16 * k;
      if (p + 16 <= n)
        { const uint4 v = *(const uint4 *) (bases + p);
          const u32 d[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
          for (int j = 0; j < 16; j++)
            if (((d[j >> 2] >> (8 * (j & 3))) & 0xffu) == 0)
              zm |= 1ull << (16 * k + j);
        }
      else
        for (int j = 0; j < 16; j++)
          if (p + j < n && bases[p + j] == 0)
            zm |= 1ull << (16 * k + j);
    }
  u32 mine = (u32) __popcll(zm), tot;
  const u32 ex = fk_block_exscan_256<u32>(mine, tmp, &tot);
  if (!EMIT)
    { if (threadIdx.x == 0)
        cnt[blockIdx.x] = tot;
      return;
    }
  int64_t o = (int64_t) off[blockIdx.x] + ex;
  while (zm != 0)
    { const int j = __ffsll((unsigned long long) zm) - 1;
      ends[o++] = p0 + j;
      zm &= zm - 1;
    }
}

// ---------------------------------------------------------------------------------------
// the codec (README: first count in 1-2 bytes, then forward differences: 00x = run of x equal
// counts (x <= 63), 01x = 6-bit two's complement difference, 1x.y = 15-bit difference mod 2^15)

template <bool EMIT>
static __global__ __launch_bounds__(64) void k_pf_encode(const uint16_t *__restrict__ cnts,
                                                         const int64_t *__restrict__ ends, int64_t nreads,
                                                         int64_t nbytes, int K, u32 *__restrict__ lens,
                                                         const u64 *__restrict__ offs, uint8_t *__restrict__ out)
{ const int64_t r = (int64_t) blockIdx.x * 64 + threadIdx.x;
  if (r >= nreads)
    return;
  const int64_t s = (r == 0) ? 0 : ends[r - 1] + 1;
  const int64_t e = ends[r] < nbytes ? ends[r] : nbytes;
  const int64_t np = e - s - K + 1;
  uint8_t *o = EMIT ? out + offs[r] : NULL;
  u32 len = 0;
  if (np > 0)
    { const uint16_t *c = cnts + s;
      u32 prev = c[0];
      if (prev < 128)
        { if (EMIT) o[len] = (uint8_t) prev;
          len += 1;
        }
      else
        { if (EMIT) { o[len] = (uint8_t) (0x80u | (prev >> 8)); o[len + 1] = (uint8_t) prev; }
          len += 2;
        }
      u32 run = 0;
      for (int64_t j = 1; j < np; j++)
        { const u32 x = c[j];
          if (x == prev)
            { if (++run == 63)
                { if (EMIT) o[len] = 63;
                  len += 1;
                  run = 0;
                }
              continue;
            }
          if (run != 0)
            { if (EMIT) o[len] = (uint8_t) run;
              len += 1;
              run = 0;
            }
          const int d = (int) x - (int) prev;
          if (d > -32 && d < 32)
            { if (EMIT) o[len] = (uint8_t) (0x40u | ((u32) d & 0x3fu));
              len += 1;
            }
          else
            { const u32 dd = (u32) d & 0x7fffu;
              if (EMIT) { o[len] = (uint8_t) (0x80u | (dd >> 8)); o[len + 1] = (uint8_t) dd; }
              len += 2;
            }
          prev = x;
        }
      if (run != 0)
        { if (EMIT) o[len] = (uint8_t) run;
          len += 1;
        }
    }
  if (!EMIT)
    lens[r] = len;
}

// ---------------------------------------------------------------------------------------

template <int KW>
static void pf_launch_counts(hipStream_t s, int64_t ntiles, const uint8_t *bases, int64_t n, int K,
                             const u32 *table, int sdw, int kbytes, const u64 *idx, int pshift, uint16_t *out)
{ hipLaunchKernelGGL(k_pf_counts<KW>, dim3((unsigned) ntiles), dim3(256), 0, s, bases, n, K, table, sdw, kbytes,
                     idx, pshift, out);
}

// Profiles of the reads in d_bases[0..nbytes) (reads end at 0 bytes; a last read without terminator
// ends at nbytes) against the sorted table d_table (nt records of kmer_stride bytes, cutoff 1).
// Results stay in HBM: *d_data (nprof bytes) and *d_offs (nreads + 1 offsets).
int fkx_profiles(fk_ctx *ctx, const void *d_bases, int64_t nbytes, const void *d_table, int64_t nt,
                 int64_t *nreads_out, int64_t *nprof_out, void **d_data, uint64_t **d_offs)
{ const fk_widths &w = ctx->wid;
  hipStream_t s = ctx->stream;
  const int K = w.kmer;
  const int KW = (w.kmer_bytes + 3) / 4;
  const int sdw = w.kmer_stride / 4;
  const int PB = w.kmer_bytes < 3 ? w.kmer_bytes : 3;
  const int pshift = 32 - 8 * PB;
  const int64_t NP = 1ll << (8 * PB);
  const uint8_t *bases = (const uint8_t *) d_bases;

  *nreads_out = 0;
  *nprof_out = 0;
  *d_data = NULL;
  *d_offs = NULL;
  if (nbytes <= 0)
    return (FK_OK);
  if (K - 1 > PF_HALO)
    { fk_set_error(ctx, "profiles: k = %d not supported", K);
      return (FK_EUNSUPPORTED);
    }

  // 1. prefix index
  u64 *idx = (u64 *) fk_slot(ctx, FK_SLOT_PF_IDX, (NP + 2) * 8);
  if (idx == NULL) return (FK_ENOMEM);
  if (nt == 0)
    FK_HIP(ctx, hipMemsetAsync(idx, 0, (size_t) (NP + 2) * 8, s));
  else
    { hipLaunchKernelGGL(k_pf_index, dim3((unsigned) ((nt + 1 + 255) / 256)), dim3(256), 0, s,
                         (const u32 *) d_table, nt, sdw, pshift, NP, idx);
      FK_LAUNCH_CHECK(ctx);
    }

  // 2. counts per position
  uint16_t *cnts = (uint16_t *) fk_slot(ctx, FK_SLOT_PF_CNT, nbytes * 2 + 64);
  if (cnts == NULL) return (FK_ENOMEM);
  const int64_t ntiles = (nbytes + PF_TILE - 1) / PF_TILE;
  switch (KW)
    { case 1: pf_launch_counts<1>(s, ntiles, bases, nbytes, K, (const u32 *) d_table, sdw, w.kmer_bytes, idx, pshift, cnts); break;
      case 2: pf_launch_counts<2>(s, ntiles, bases, nbytes, K, (const u32 *) d_table, sdw, w.kmer_bytes, idx, pshift, cnts); break;
      case 3: pf_launch_counts<3>(s, ntiles, bases, nbytes, K, (const u32 *) d_table, sdw, w.kmer_bytes, idx, pshift, cnts); break;
      case 4: pf_launch_counts<4>(s, ntiles, bases, nbytes, K, (const u32 *) d_table, sdw, w.kmer_bytes, idx, pshift, cnts); break;
      case 5: pf_launch_counts<5>(s, ntiles, bases, nbytes, K, (const u32 *) d_table, sdw, w.kmer_bytes, idx, pshift, cnts); break;
      case 6: pf_launch_counts<6>(s, ntiles, bases, nbytes, K, (const u32 *) d_table, sdw, w.kmer_bytes, idx, pshift, cnts); break;
      case 7: pf_launch_counts<7>(s, ntiles, bases, nbytes, K, (const u32 *) d_table, sdw, w.kmer_bytes, idx, pshift, cnts); break;
      case 8: pf_launch_counts<8>(s, ntiles, bases, nbytes, K, (const u32 *) d_table, sdw, w.kmer_bytes, idx, pshift, cnts); break;
      default:
        fk_set_error(ctx, "profiles: k = %d not supported", K);
        return (FK_EUNSUPPORTED);
    }
  FK_LAUNCH_CHECK(ctx);

  // 3. read boundaries
  const int64_t nz = (nbytes + PF_ZCH - 1) / PF_ZCH;
  u32 *zc = (u32 *) fk_slot(ctx, FK_SLOT_PF_ZC, nz * 4 + 64);
  u64 *zo = (u64 *) fk_slot(ctx, FK_SLOT_PF_ZO, nz * 8 + 64);
  if (zc == NULL || zo == NULL) return (FK_ENOMEM);
  hipLaunchKernelGGL(k_pf_zeros<false>, dim3((unsigned) nz), dim3(256), 0, s, bases, nbytes, zc,
                     (const u64 *) NULL, (int64_t *) NULL);
  hipLaunchKernelGGL(k_exscan_tiles, dim3(1), dim3(256), 0, s, (const u32 *) zc, nz, zo, ctx->d_scratch);
  FK_LAUNCH_CHECK(ctx);
  uint8_t lastb = 0;
  FK_HIP(ctx, hipMemcpyAsync(ctx->h_scratch, ctx->d_scratch, 8, hipMemcpyDeviceToHost, s));
  FK_HIP(ctx, hipMemcpyAsync((char *) ctx->h_scratch + 8, bases + nbytes - 1, 1, hipMemcpyDeviceToHost, s));
  FK_HIP(ctx, hipStreamSynchronize(s));
  const int64_t nzero = (int64_t) ctx->h_scratch[0];
  lastb = *((uint8_t *) ctx->h_scratch + 8);
  const int64_t nreads = nzero + (lastb != 0 ? 1 : 0);
  *nreads_out = nreads;
  if (nreads == 0)
    return (FK_OK);
  int64_t *ends = (int64_t *) fk_slot(ctx, FK_SLOT_PF_ENDS, (nreads + 1) * 8);
  if (ends == NULL) return (FK_ENOMEM);
  hipLaunchKernelGGL(k_pf_zeros<true>, dim3((unsigned) nz), dim3(256), 0, s, bases, nbytes, (u32 *) NULL,
                     (const u64 *) zo, ends);
  FK_LAUNCH_CHECK(ctx);
  if (lastb != 0)
    FK_HIP(ctx, hipMemcpyAsync(ends + nzero, &nbytes, 8, hipMemcpyHostToDevice, s));

  // 4. codec: lengths, offsets, bytes
  u32 *lens = (u32 *) fk_slot(ctx, FK_SLOT_PF_LEN, nreads * 4 + 64);
  u64 *offs = (u64 *) fk_slot(ctx, FK_SLOT_PF_OFF, (nreads + 1) * 8 + 64);
  if (lens == NULL || offs == NULL) return (FK_ENOMEM);
  const unsigned nb = (unsigned) ((nreads + 63) / 64);
  hipLaunchKernelGGL(k_pf_encode<false>, dim3(nb), dim3(64), 0, s, (const uint16_t *) cnts, (const int64_t *) ends,
                     nreads, nbytes, K, lens, (const u64 *) NULL, (uint8_t *) NULL);
  hipLaunchKernelGGL(k_exscan_tiles, dim3(1), dim3(256), 0, s, (const u32 *) lens, nreads, offs, offs + nreads);
  FK_LAUNCH_CHECK(ctx);
  FK_HIP(ctx, hipMemcpyAsync(ctx->h_scratch, offs + nreads, 8, hipMemcpyDeviceToHost, s));
  FK_HIP(ctx, hipStreamSynchronize(s));       // also keeps the stack variable nbytes alive for the copy above
  const int64_t nprof = (int64_t) ctx->h_scratch[0];
  uint8_t *data = (uint8_t *) fk_slot(ctx, FK_SLOT_PF_OUT, nprof + 64);
  if (data == NULL) return (FK_ENOMEM);
  hipLaunchKernelGGL(k_pf_encode<true>, dim3(nb), dim3(64), 0, s, (const uint16_t *) cnts, (const int64_t *) ends,
                     nreads, nbytes, K, (u32 *) NULL, (const u64 *) offs, data);
  FK_LAUNCH_CHECK(ctx);
  *nprof_out = nprof;
  *d_data = data;
  *d_offs = (uint64_t *) offs;
  return (FK_OK);
}
