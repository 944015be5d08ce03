// fk_aggr.hip -- count weighted k-mers that two hashed digit passes have grouped by 16 hash bits.
//
// Replaces, with the same results, the reference's sort-then-scan of the weighted k-mer list
// (Weighted_Kmer_Sort MSDsort.c:536-544 -> hist_kmers MSDsort.c:491-509 -> table_write_thread
// count.c:564-616).  Only the table has to be in k-mer order, and only k-mers with count >= the
// cutoff are in the table, so the W weighted records are not sorted at all: fkx_group brings all
// copies of a k-mer into one of 65,536 hash bins (two 8-bit digit passes), and here one workgroup
// per bin sums the weights of equal k-mers in a hash table that lives in LDS, updates the 0x8000-bin
// histogram (saturation and max_inst exactly as MSDsort.c:498-506) and appends the qualifying
// (k-mer, count) records to the table buffer, which the caller then sorts on KMER_BYTES.
//
// A bin whose distinct k-mers do not fit the LDS table is processed in 2, 4, ... rounds, each
// round taking the records whose next hash bits select it (the bin is re-read, mostly from L2).
#include "fk_common.h"

#define AG_THREADS 1024
#define AG_WAVES   (AG_THREADS / 64)
#define AG_HB      4096                 // LDS-private histogram bins
#define AG_HIGH    (1u << 29)           // a carried count that reaches this is cut back by AG_CUT, the
#define AG_CUT     (1u << 28)           //   removed instances go straight to max_inst
#define AG_MAXR    64
#define AG_BINS    65536
#define AG_NSCAL   16                   // scalars behind the histogram (8 results + 8 phase timers of ablation builds)
#ifdef FK_ABLATION
#define AG_T(k) do { if (tid == 0) { const u64 now_ = __builtin_readcyclecounter(); ph[k] += now_ - tlast; tlast = now_; } } while (0)
#else
#define AG_T(k) do { } while (0)
#endif

// a record as one memory operation: three dword loads / stores per lane at a 12-byte stride are three requests per
// record where one (dwordx3) does (a quarter of the replay passes' time was this)
template <int N> struct __attribute__((packed, aligned(4))) ag_rec { u32 w[N]; };

template <int KW> struct AgCfg
{ static constexpr int CAP = (KW <= 3) ? 8192 : 4096;   // records one fill takes = cells of the counting sort
  static constexpr int NS  = CAP / AG_THREADS;          // records a thread holds in registers during a fill
  static constexpr size_t LDS = (size_t) CAP * 4 * (KW + 1) + AG_HB * 4 + 256;
};

// position of the first record of every bin: bounds[b] = lower bound of (hash16 >= b), bounds[65536] = n
template <int KW>
__global__ __launch_bounds__(256) void k_ag_bounds(const u32 *__restrict__ recs, int64_t n, int kbytes,
                                                   u64 *__restrict__ bounds)
{ const u32 b = blockIdx.x * 256 + threadIdx.x;
  if (b > AG_BINS) return;
  if (b == AG_BINS) { bounds[b] = (u64) n; return; }
  int64_t lo = 0, hi = n;
  while (lo < hi)
    { const int64_t mid = (lo + hi) >> 1;
      u32 r[KW];
#pragma unroll
      for (int w = 0; w < KW; w++)
        r[w] = recs[mid * KW + w];
      u32 ha, hb;
      fk_rec_hash<KW>(r, kbytes, ha, hb);
      if ((hb & 0xffffu) < b) lo = mid + 1;
      else hi = mid;
    }
  bounds[b] = (u64) lo;
}

// exclusive scan over the 1024 threads of the block.  tmp: AG_WAVES u32 of LDS.
__device__ __forceinline__ u32 ag_block_exscan(u32 v, u32 *tmp, u32 *total)
{ const u32 lane = fk_lane();
  const u32 wave = threadIdx.x >> 6;
  u32 x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1)
    { u32 y = __shfl_up(x, o, 64);
      if ((int) lane >= o) x += y;
    }
  if (lane == 63) tmp[wave] = x;
  __syncthreads();
  u32 base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < AG_WAVES; w++)
    { const u32 t = tmp[w];
      if ((u32) w < wave) base += t;
      tot += t;
    }
  __syncthreads();
  *total = tot;
  return (base + x - v);
}

// In-place exclusive scan of cell[0 .. NS * 1024): thread t owns cells NS*t .. NS*t + NS - 1.  Returns the total.
// The caller puts a barrier behind it before anybody reads a cell.  tmp: AG_WAVES u32 of LDS.
template <int NS>
__device__ __forceinline__ u32 ag_scan_cells(u32 *cell, u32 *tmp)
{ const u32 lane = fk_lane();
  const u32 wave = threadIdx.x >> 6;
  uint4 *c4 = (uint4 *) cell + threadIdx.x * (NS / 4);
  u32 c[NS];
#pragma unroll
  for (int i = 0; i < NS / 4; i++)
    { const uint4 q = c4[i];
      c[4 * i] = q.x; c[4 * i + 1] = q.y; c[4 * i + 2] = q.z; c[4 * i + 3] = q.w;
    }
  u32 s = 0;
#pragma unroll
  for (int i = 0; i < NS; i++)
    { const u32 t = c[i];
      c[i] = s;
      s += t;
    }
  u32 x = s;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1)
    { u32 y = __shfl_up(x, o, 64);
      if ((int) lane >= o) x += y;
    }
  if (lane == 63) tmp[wave] = x;
  __syncthreads();
  // the 16 wave totals scanned by every group of 16 lanes
  u32 y = tmp[lane & (AG_WAVES - 1)];
#pragma unroll
  for (int o = 1; o < AG_WAVES; o <<= 1)
    { u32 z = __shfl_up(y, o, AG_WAVES);
      if ((int) (lane & (AG_WAVES - 1)) >= o) y += z;
    }
  const u32 total = __shfl(y, AG_WAVES - 1, 64);
  const u32 prev  = __shfl(y, (int) ((wave + AG_WAVES - 1) & (AG_WAVES - 1)), 64);
  const u32 base  = ((wave == 0) ? 0u : prev) + x - s;
#pragma unroll
  for (int i = 0; i < NS / 4; i++)
    c4[i] = make_uint4(c[4 * i] + base, c[4 * i + 1] + base, c[4 * i + 2] + base, c[4 * i + 3] + base);
  return (total);
}

// scal: [0] max_inst  [1] distinct k-mers  [2] table entries  [3] failure flag  [4] bin ticket
//       [5] extra rounds taken  [6] table buffer too small (tcap records)
//
// One persistent workgroup per CU takes hash bins in turn.  A bin (<= CAP records, all copies of a k-mer among
// them) is summed by a COUNTING SORT IN LDS on 13 further hash bits followed by a leader search -- no hash-table
// protocol (compare-and-swap claims, locked slots, retries of lanes that lost a race):
//   A   every thread holds NS records in registers; cell = hash & (CAP - 1); rank = atomic counter of the cell
//   S   exclusive scan of the CAP cell counters in place
//   B   position p = scanned counter + rank; key -> K[.][p], weight -> cell[p] (the counters are dead by then)
//   C   a record with rank > 0 compares itself with the records in front of it in its cell (1.3 on average): the
//       first equal one is the k-mer's LEADER and takes the weight (one LDS add), the record's own count becomes 0
//   H   position p with a count != 0 is a distinct k-mer: histogram, max_inst, table candidate
// A bin of more than CAP records is taken in chunks: the leaders found so far stay in the fill as records that
// carry their count (heavy k-mers of any multiplicity cost LDS space once); if the distinct k-mers alone pass
// `limit` the bin is taken in 2, 4, ... selections by further hash bits, each re-reading the bin.
// DEDUP: the records are super-mers (whole record = key, weight 1); every distinct record comes out
// once, followed by a dword with its multiplicity (records of KW + 1 dwords), nothing else is computed.
template <int KW, bool DEDUP>
__global__ __launch_bounds__(AG_THREADS) void k_ag_count(const u32 *__restrict__ recs,
                                                         const u64 *__restrict__ bounds, int kbytes,
                                                         int cutoff, u64 *__restrict__ hist_g,
                                                         u64 *__restrict__ scal, u32 *__restrict__ table,
                                                         int cap_eff, int limit, int variant, int gshift, u32 sat, u64 tcap)
{ constexpr int CAP = AgCfg<KW>::CAP;
  constexpr int NS  = AgCfg<KW>::NS;
  extern __shared__ uint4 ag_lds[];
  u32 *cell   = (u32 *) ag_lds;                            // [CAP] counters -> offsets -> counts by position
  u32 *K      = cell + CAP;                                // [KW][CAP] keys by position
  u32 *lhist  = K + KW * CAP;                              // [AG_HB]
  u32 *sh_tmp = lhist + AG_HB;                             // [AG_WAVES]
  u64 *sh_base = (u64 *) (sh_tmp + AG_WAVES);
  const int tid = threadIdx.x;

  for (int i = tid; i < CAP; i += AG_THREADS)
    cell[i] = 0;
  for (int i = tid; i < AG_HB; i += AG_THREADS)
    lhist[i] = 0;
  u64 my_max = 0;
  u32 my_distinct = 0, my_rounds = 0;
  u32 R0 = 1;
#ifdef FK_ABLATION
  u64 ph[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, tlast = __builtin_readcyclecounter();
#endif
  u32 kmask[KW];                                // key bytes of each record dword (pad and weight off)
#pragma unroll
  for (int w = 0; w < KW; w++)
    kmask[w] = (4 * w + 4 <= kbytes) ? 0xffffffffu : (4 * w < kbytes) ? ((1u << (8 * (kbytes - 4 * w))) - 1u) : 0u;
  __syncthreads();

  // records fetched ahead for the first chunk of the next bin (or selection) while this one is swept
  u32     key[NS][KW], wgt[NS];          // a new record sits in key[] as loaded until step A takes its weight off
  int64_t raw_beg = -1;

  // groups of 2^gshift neighbouring bins (one bin when the input is large) are dealt round-robin:
  // hashing makes them equally heavy
  const u32 nbins = (u32) (AG_BINS >> gshift);
  for (u32 bin = blockIdx.x; bin < nbins; bin += gridDim.x)
    { const int64_t beg = (int64_t) bounds[bin << gshift], end = (int64_t) bounds[(bin + 1) << gshift];
      if (beg >= end)
        continue;
      int64_t nx_beg = -1, nx_end = -1;         // the bin this workgroup takes next
      if (bin + gridDim.x < nbins)
        { nx_beg = (int64_t) bounds[(bin + gridDim.x) << gshift];
          nx_end = (int64_t) bounds[(bin + gridDim.x + 1) << gshift];
        }

      // A bin is taken in R0 selections (records whose next hash bits equal r0); R0 is what the
      // previous bin of this workgroup needed (all bins are alike), so that a fill that is too
      // small for whole bins is not found out again bin after bin.  A selection that still does
      // not fit is halved on the spot (depth-first), exactly once.
      bool bin_ovf = false, failed = false;
      u32  bin_fill = 0;
      for (u32 r0 = 0; r0 < R0 && !failed; r0++)
      { u32 R = R0, r = r0;
      for (;;)
        { u64     round_max = 0;
          int64_t pos = beg;
          u32     carried = 0, vmask = 0, ntot = 0;
          bool    ovf = false;
          u32     v[NS];
          for (;;)
            { // ---- fill the register slots: slot j of thread tid is "lane position" j * 1024 + tid
              u32 isnew = 0;
              if (carried == 0)
                { const int64_t room = end - pos;
                  const u32 nnew = (room < (int64_t) cap_eff) ? (u32) room : (u32) cap_eff;
                  if (raw_beg != pos)
                    {
#pragma unroll
                      for (int j = 0; j < NS; j++)
                        { const u32 L = (u32) (j * AG_THREADS + tid);
                          if (L < nnew)
                            { const ag_rec<KW> rr = *(const ag_rec<KW> *) (recs + (pos + L) * KW);
#pragma unroll
                              for (int w = 0; w < KW; w++)
                                key[j][w] = rr.w[w];
                            }
                        }
                    }
#pragma unroll
                  for (int j = 0; j < NS; j++)
                    if ((u32) (j * AG_THREADS + tid) < nnew)
                      isnew |= (1u << j);
                  pos += nnew;
                }
              else
                { // the leaders carried over keep their slots; the free slots take the next records of the bin
                  u32 fr = 0;
#pragma unroll
                  for (int j = 0; j < NS; j++)
                    if ((u32) (j * AG_THREADS + tid) < (u32) cap_eff && !((vmask >> j) & 1u))
                      fr |= (1u << j);
                  u32 totfree;
                  u32 k = ag_block_exscan((u32) __popc(fr), sh_tmp, &totfree);
                  const int64_t room = end - pos;
                  const u32 nnew = (room < (int64_t) totfree) ? (u32) room : totfree;
#pragma unroll
                  for (int j = 0; j < NS; j++)
                    if ((fr >> j) & 1u)
                      { if (k < nnew)
                          { const ag_rec<KW> rr = *(const ag_rec<KW> *) (recs + (pos + k) * KW);
#pragma unroll
                            for (int w = 0; w < KW; w++)
                              key[j][w] = rr.w[w];
                            isnew |= (1u << j);
                          }
                        k += 1;
                      }
                  pos += nnew;
                }
              raw_beg = -1;
              AG_T(0);

              // ---- A: cell and rank of every record of the fill
              u32 sub[NS], rank[NS];
#pragma unroll
              for (int j = 0; j < NS; j++)
                { if ((isnew >> j) & 1u)
                    { wgt[j] = DEDUP ? 1u : (key[j][KW - 1] >> 16);
#pragma unroll
                      for (int w = 0; w < KW; w++)
                        key[j][w] &= kmask[w];
                    }
                  sub[j] = 0; rank[j] = 0;
                  if (((isnew | vmask) >> j) & 1u)
                    { u32 ha, hb;
                      fk_rec_hash<KW>(key[j], kbytes, ha, hb);
                      if (((isnew >> j) & 1u) && ((hb >> 16) & (R - 1)) != r)
                        continue;                              // not in this selection
                      vmask |= (1u << j);
                      sub[j]  = ha & (CAP - 1);
                      rank[j] = atomicAdd(&cell[sub[j]], 1u);
                    }
                }
              __syncthreads();
              AG_T(1);
              ntot = ag_scan_cells<NS>(cell, sh_tmp);
              __syncthreads();
              AG_T(2);

              // ---- B: positions; then keys and weights to their positions (the counters become the counts)
              u32 p[NS], off[NS];
#pragma unroll
              for (int j = 0; j < NS; j++)
                { off[j] = 0;
                  if ((vmask >> j) & 1u)
                    off[j] = cell[sub[j]];
                  p[j] = off[j] + rank[j];
                }
              __syncthreads();
#pragma unroll
              for (int j = 0; j < NS; j++)
                { if ((vmask >> j) & 1u)
                    {
#pragma unroll
                      for (int w = 0; w < KW; w++)
                        K[w * CAP + p[j]] = key[j][w];
                      cell[p[j]] = wgt[j];
                    }
                  const u32 L = (u32) (j * AG_THREADS + tid);
                  if (L >= ntot)
                    cell[L] = 0;
                }
              __syncthreads();
              AG_T(3);

              // ---- C: a record that is not the first of its cell looks for its k-mer in front of it
#pragma unroll
              for (int j = 0; j < NS; j++)
                if (((vmask >> j) & 1u) && rank[j] != 0)
                  { u32 q = off[j];
                    bool found = false;
                    while (q < p[j])
                      { u32 e = 0;
#pragma unroll
                        for (int w = 0; w < KW; w++)
                          e |= K[w * CAP + q] ^ key[j][w];
                        if (e == 0) { found = true; break; }
                        q += 1;
                      }
                    if (found)
                      { atomicAdd(&cell[q], wgt[j]);
                        cell[p[j]] = 0;
                      }
                  }
              __syncthreads();
              AG_T(4);

              // ---- harvest: counts by position; the cells are left zero for the next fill
#pragma unroll
              for (int j = 0; j < NS; j++)
                { const u32 L = (u32) (j * AG_THREADS + tid);
                  v[j] = 0;
                  if (L < ntot)
                    { v[j] = cell[L];
                      if (v[j] != 0) cell[L] = 0;
                    }
                }
              if (pos >= end)
                break;
              // more of the bin to come: the leaders stay, as records that carry their counts
              u32 nl = 0, D;
#pragma unroll
              for (int j = 0; j < NS; j++)
                nl += (v[j] != 0);
              ag_block_exscan(nl, sh_tmp, &D);
              if (D > (u32) limit)
                { ovf = true;
                  break;
                }
              vmask = 0;
#pragma unroll
              for (int j = 0; j < NS; j++)
                if (v[j] != 0)
                  { const u32 L = (u32) (j * AG_THREADS + tid);
                    vmask |= (1u << j);
#pragma unroll
                    for (int w = 0; w < KW; w++)
                      key[j][w] = K[w * CAP + L];
                    if (v[j] >= AG_HIGH)                       // stays far above 0x7fff: still saturated
                      { v[j] -= AG_CUT;
                        round_max += AG_CUT;
                      }
                    wgt[j] = v[j];
                  }
              carried = D;
              my_rounds += (tid == 0);
            }
          if (ovf)
            { // more distinct k-mers than the fill takes: halve the selection and start it again
              my_rounds += (tid == 0);
              bin_ovf = true;
              __syncthreads();
              if (R >= AG_MAXR)
                { if (tid == 0)
                    atomicAdd(&scal[3], 1ull);
                  failed = true;
                  break;
                }
              R <<= 1;
              continue;
            }

          // ---- what comes next: another selection of this bin, or the workgroup's next bin -- fetch its first
          // chunk now, the sweep below needs no registers of the fill
          { u32 Rn = R, rn = r;
            while (Rn > R0 && rn >= (Rn >> 1))
              { rn -= (Rn >> 1);
                Rn >>= 1;
              }
            const bool last = (Rn == R0) && (r0 + 1 >= R0);
            const int64_t fb = last ? nx_beg : beg, fe = last ? nx_end : end;
            if (fb >= 0 && fb < fe)
              { const int64_t room = fe - fb;
                const u32 nnew = (room < (int64_t) cap_eff) ? (u32) room : (u32) cap_eff;
#pragma unroll
                for (int j = 0; j < NS; j++)
                  { const u32 L = (u32) (j * AG_THREADS + tid);
                    if (L < nnew)
                      { const ag_rec<KW> rr = *(const ag_rec<KW> *) (recs + (fb + L) * KW);
#pragma unroll
                        for (int w = 0; w < KW; w++)
                          key[j][w] = rr.w[w];
                      }
                  }
                raw_beg = fb;
              }
          }

          // ---- emit: histogram, totals, table entries
          my_max += round_max;
          // (the k-mers seen once or twice -- four out of five on read sets with sequencing errors -- are counted
          // per wave with a ballot: one LDS atomic per lane on the same two histogram bins serialised the sweep)
          u32 c[NS];
          u32 nq = 0, n1 = 0, n2 = 0, nlead = 0;
#pragma unroll
          for (int j = 0; j < NS; j++)
            { const u32 vv = v[j];
              c[j] = 0;
              if (!DEDUP)
                { n1 += (u32) __popcll(__ballot(vv == 1u));
                  n2 += (u32) __popcll(__ballot(vv == 2u));
                }
              if (vv != 0)
                { nlead += 1;
                  u32 cc = vv;
                  if (DEDUP)
                    { nq += 1;
                      c[j] = vv;
                      continue;
                    }
                  if (vv >= sat)                               // sat = 0x7fff, MSDsort.c:498-506
                    my_max += vv;
                  if (vv >= 0x7fffu)
                    cc = 0x7fffu;
                  if ((variant & 2) || cc <= 2u) ;
                  else if (cc < AG_HB) atomicAdd(&lhist[cc], 1u);
                  else            atomicAdd(&hist_g[cc], 1ull);
                  if (cutoff > 0 && (int) cc >= cutoff)
                    { nq += 1;
                      c[j] = cc;
                    }
                }
            }
          my_distinct += nlead;
          if (!DEDUP && !(variant & 2) && fk_lane() == 0)
            { if (n1 != 0) atomicAdd(&lhist[1], n1);
              if (n2 != 0) atomicAdd(&lhist[2], n2);
            }
          if (R0 > 1)
            { u32 D;
              ag_block_exscan(nlead, sh_tmp, &D);
              bin_fill = max(bin_fill, D);
            }
          if ((DEDUP || cutoff > 0) && !(variant & 4))
            { u32 tot;
              const u32 off = ag_block_exscan(nq, sh_tmp, &tot);
              if (tid == 0 && tot > 0)
                *sh_base = atomicAdd(&scal[2], (u64) tot);
              __syncthreads();
              if (tot > 0 && *sh_base + tot > tcap)
                { if (tid == 0)                       // the table buffer is full (direct append to a union buffer)
                    atomicAdd(&scal[6], 1ull);
                }
              else if (tot > 0)
                { u64 o = *sh_base + off;
#pragma unroll
                  for (int j = 0; j < NS; j++)
                    if (c[j] != 0)
                      { const u32 L = (u32) (j * AG_THREADS + tid);
                        u32 kd[KW];
#pragma unroll
                        for (int w = 0; w < KW; w++)
                          kd[w] = K[w * CAP + L];
                        if (DEDUP)
                          { ag_rec<KW + 1> ro;
#pragma unroll
                            for (int w = 0; w < KW; w++)
                              ro.w[w] = kd[w];
                            ro.w[KW] = c[j];
                            *(ag_rec<KW + 1> *) (table + o * (KW + 1)) = ro;
                          }
                        else
                          { ag_rec<KW> ro;
#pragma unroll
                            for (int w = 0; w < KW - 1; w++)
                              ro.w[w] = kd[w];
                            ro.w[KW - 1] = kd[KW - 1] | (c[j] << 16);
                            *(ag_rec<KW> *) (table + o * KW) = ro;
                          }
                        o += 1;
                      }
                }
            }
          __syncthreads();
          AG_T(5);

          // ---- next selection below (R0, r0): sibling, or up
          while (R > R0 && r >= (R >> 1))
            { r -= (R >> 1);
              R >>= 1;
            }
          if (R == R0)
            break;
          r += (R >> 1);
        }
      }
      if (bin_ovf)
        R0 = min(R0 << 1, (u32) AG_MAXR);
      else if (R0 > 1 && bin_fill * 9 < (u32) limit * 4)
        R0 >>= 1;
    }

  // flush the private histogram and the per-thread totals
  __syncthreads();
  for (int i = tid; i < AG_HB; i += AG_THREADS)
    if (lhist[i] != 0)
      atomicAdd(&hist_g[i], (u64) lhist[i]);
  u64 d = my_distinct, rd = my_rounds;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
    { my_max += __shfl_down(my_max, o, 64);
      d      += __shfl_down(d, o, 64);
      rd     += __shfl_down(rd, o, 64);
    }
  if (fk_lane() == 0)
    { if (my_max) atomicAdd(&scal[0], my_max);
      if (d)      atomicAdd(&scal[1], d);
      if (rd)     atomicAdd(&scal[5], rd);
    }
#ifdef FK_ABLATION
  if (tid == 0)
    for (int k = 0; k < 8; k++)
      atomicAdd(&scal[8 + k], ph[k]);
#endif
}

// bins merged per fill: 2^gshift, the largest group whose expected size stays within 7/16 of a fill when doubled
template <int KW>
static int ag_gshift(int64_t n)
{ int gshift = 0;
  while (gshift < 16 && ((n >> (16 - gshift)) << 1) <= AgCfg<KW>::CAP * 7 / 8)
    gshift += 1;
  return (gshift);
}

template <int KW>
static int aggr_t(fk_ctx *ctx, const void *d_grouped, int64_t n, int cutoff, int64_t *hist,
                  int64_t *max_inst, int64_t *ndistinct, void *d_table, int64_t cap, int64_t *ntable)
{ hipStream_t s = ctx->stream;
  if (ntable) *ntable = 0;
  if (ndistinct) *ndistinct = 0;
  if (n == 0)
    return (FK_OK);
  if (cutoff > 0 && (d_table == NULL || cap <= 0))
    { fk_set_error(ctx, "aggregate: no table buffer");
      return (FK_EINVAL);
    }
  u64 *d_bounds = (u64 *) fk_slot(ctx, FK_SLOT_AG_BOUNDS, (AG_BINS + 1) * 8);
  u64 *d_hist   = (u64 *) fk_slot(ctx, FK_SLOT_CT_HIST, (FK_HIST_BINS + AG_NSCAL) * 8);
  if (d_bounds == NULL || d_hist == NULL)
    return (FK_ENOMEM);
  u64 *d_scal = d_hist + FK_HIST_BINS;
  static bool attr_set[8] = { false };
  const size_t lds = AgCfg<KW>::LDS;
  if (!attr_set[KW])
    { auto kern = k_ag_count<KW, false>;
      FK_HIP(ctx, hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
      attr_set[KW] = true;
    }
  FK_HIP(ctx, hipMemsetAsync(d_hist, 0, (FK_HIST_BINS + AG_NSCAL) * 8, s));
  hipLaunchKernelGGL(k_ag_bounds<KW>, dim3(AG_BINS / 256 + 1), dim3(256), 0, s, (const u32 *) d_grouped, n,
                     ctx->wid.kmer_bytes, d_bounds);
  const int cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
  // Neighbouring bins are merged while two of them still fit one fill with room to spare: small inputs get full
  // fills, and a fill is never expected to hold more records than it takes -- a bin beyond CAP costs a second
  // chunk with the first one's distinct k-mers carried along.
  int gshift = ag_gshift<KW>(n);
  if (ctx->dbg_aggr_limit > 0)
    gshift = 0;
  if (ctx->dbg_aggr_gshift > 0)
    gshift = ctx->dbg_aggr_gshift - 1;
  int cap_eff = AgCfg<KW>::CAP;                       // fk_debug_set("aggr_limit"): a fill takes only this many records
  if (ctx->dbg_aggr_limit > 0 && ctx->dbg_aggr_limit < cap_eff)
    cap_eff = ctx->dbg_aggr_limit < 4 ? 4 : ctx->dbg_aggr_limit;
  const int limit = cap_eff * 3 / 4;
  hipLaunchKernelGGL((k_ag_count<KW, false>), dim3((unsigned) cus), dim3(AG_THREADS), lds, s, (const u32 *) d_grouped,
                     (const u64 *) d_bounds, ctx->wid.kmer_bytes, cutoff, d_hist, d_scal, (u32 *) d_table, cap_eff, limit, ctx->dbg_aggr_variant, gshift,
                     (u32) (ctx->aggr_sat > 0 ? ctx->aggr_sat : 0x7fff), (u64) cap);
  FK_LAUNCH_CHECK(ctx);
  u64 *h = ctx->h_scratch;                       // pinned, 64 KB + 64 KB: the histogram needs 256 KB
  u64 *hh = (u64 *) malloc((FK_HIST_BINS + AG_NSCAL) * 8);
  if (hh == NULL) return (FK_ENOMEM);
  (void) h;
  if (hipMemcpyAsync(hh, d_hist, (FK_HIST_BINS + AG_NSCAL) * 8, hipMemcpyDeviceToHost, s) != hipSuccess
      || hipStreamSynchronize(s) != hipSuccess)
    { free(hh);
      fk_set_error(ctx, "aggregate: HIP failure: %s", hipGetErrorString(hipGetLastError()));
      return (FK_EHIP);
    }
  if (hh[FK_HIST_BINS + 3] != 0)
    { free(hh);
      return (FK_ESTATE);                          // a bin did not fit even in AG_MAXR rounds
    }
  if (hh[FK_HIST_BINS + 6] != 0)
    { free(hh);
      return (FKX_TABLE_FULL);                     // fewer than `cap` records of room: nothing was accumulated
    }
  for (int i = 1; i < FK_HIST_BINS; i++)
    hist[i] += (int64_t) hh[i];
  *max_inst += (int64_t) hh[FK_HIST_BINS + 0];
  if (ndistinct) *ndistinct = (int64_t) hh[FK_HIST_BINS + 1];
  if (ntable) *ntable = (int64_t) hh[FK_HIST_BINS + 2];
  ctx->aggr_extra_rounds = (int64_t) hh[FK_HIST_BINS + 5];
#ifdef FK_ABLATION
  if (getenv("FK_AG_TIMING") != NULL)
    { fprintf(stderr, "ag phases (cycles of thread 0, all workgroups) n=%lld:", (long long) n);
      for (int k = 0; k < 8; k++)
        fprintf(stderr, " %llu", (unsigned long long) hh[FK_HIST_BINS + 8 + k]);
      fprintf(stderr, "  (load, A, scan, B, C, harvest+emit)\n");
    }
#endif
  free(hh);
  return (FK_OK);
}

/* d_grouped: n weighted k-mer records ordered by the low 16 bits of fk_rec_hash's b word (the
   result of fkx_group(..., key_bytes = KMER_BYTES, npasses = 2)).  Adds the histogram of their
   counts into hist[1..0x7fff] and the instances of saturated k-mers into *max_inst; with cutoff > 0
   writes the (k-mer, count) records with count >= cutoff to d_table IN NO PARTICULAR ORDER.
   FK_ESTATE: some bin holds more distinct k-mers than AG_MAXR rounds of the LDS table take. */
int fkx_aggregate(fk_ctx *ctx, const void *d_grouped, int64_t n, int cutoff, int64_t *hist,
                  int64_t *max_inst, int64_t *ndistinct, void *d_table, int64_t cap, int64_t *ntable)
{ switch (ctx->wid.kmer_stride >> 2)
  { case 1: return aggr_t<1>(ctx, d_grouped, n, cutoff, hist, max_inst, ndistinct, d_table, cap, ntable);
    case 2: return aggr_t<2>(ctx, d_grouped, n, cutoff, hist, max_inst, ndistinct, d_table, cap, ntable);
    case 3: return aggr_t<3>(ctx, d_grouped, n, cutoff, hist, max_inst, ndistinct, d_table, cap, ntable);
    case 4: return aggr_t<4>(ctx, d_grouped, n, cutoff, hist, max_inst, ndistinct, d_table, cap, ntable);
    case 5: return aggr_t<5>(ctx, d_grouped, n, cutoff, hist, max_inst, ndistinct, d_table, cap, ntable);
    default:
      fk_set_error(ctx, "k-mer stride %d not built", ctx->wid.kmer_stride);
      return (FK_EUNSUPPORTED);
  }
}

// ---------------------------------------------------------------------------------------------
// Super-mer de-duplication with the same kernel: n records of KW dwords, grouped by 16 hash bits of
// the whole record -> every distinct record once, followed by its multiplicity (KW + 1 dwords each,
// in no particular order).  Replaces two of the four grouping passes and the run detection of the
// expansion (count.c:421-426).
template <int KW>
static int dedup_t(fk_ctx *ctx, const void *d_grouped, int64_t n, void *d_out, int64_t cap, int64_t *nout)
{ hipStream_t s = ctx->stream;
  *nout = 0;
  if (n == 0)
    return (FK_OK);
  if (d_out == NULL || cap < n)
    { fk_set_error(ctx, "dedup: the output buffer must take as many records as the input (%lld)", (long long) n);
      return (FK_EINVAL);
    }
  u64 *d_bounds = (u64 *) fk_slot(ctx, FK_SLOT_AG_BOUNDS, (AG_BINS + 1) * 8);
  u64 *d_hist   = (u64 *) fk_slot(ctx, FK_SLOT_CT_HIST, (FK_HIST_BINS + AG_NSCAL) * 8);
  if (d_bounds == NULL || d_hist == NULL)
    return (FK_ENOMEM);
  u64 *d_scal = d_hist + FK_HIST_BINS;
  static bool attr_set = false;
  const size_t lds = AgCfg<KW>::LDS;
  if (!attr_set)
    { auto kern = k_ag_count<KW, true>;
      FK_HIP(ctx, hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
      attr_set = true;
    }
  FK_HIP(ctx, hipMemsetAsync(d_scal, 0, 8 * 8, s));
  hipLaunchKernelGGL(k_ag_bounds<KW>, dim3(AG_BINS / 256 + 1), dim3(256), 0, s, (const u32 *) d_grouped, n,
                     KW * 4, d_bounds);
  const int cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
  const int gshift = ag_gshift<KW>(n);
  hipLaunchKernelGGL((k_ag_count<KW, true>), dim3((unsigned) cus), dim3(AG_THREADS), lds, s, (const u32 *) d_grouped,
                     (const u64 *) d_bounds, KW * 4, 1, d_hist, d_scal, (u32 *) d_out, AgCfg<KW>::CAP, AgCfg<KW>::CAP * 3 / 4, 0,
                     gshift, 0x7fffu, (u64) cap);
  FK_LAUNCH_CHECK(ctx);
  FK_HIP(ctx, hipMemcpyAsync(ctx->h_scratch + 4100, d_scal, 8 * 8, hipMemcpyDeviceToHost, s));
  FK_HIP(ctx, hipStreamSynchronize(s));
  if (ctx->h_scratch[4100 + 3] != 0)
    return (FK_ESTATE);
  *nout = (int64_t) ctx->h_scratch[4100 + 2];
  return (FK_OK);
}

int fkx_dedup_supermers(fk_ctx *ctx, const void *d_grouped, int64_t n, void *d_out, int64_t cap, int64_t *nout)
{ switch (ctx->wid.smer_stride >> 2)
  { case 2: return dedup_t<2>(ctx, d_grouped, n, d_out, cap, nout);
    case 3: return dedup_t<3>(ctx, d_grouped, n, d_out, cap, nout);
    case 4: return dedup_t<4>(ctx, d_grouped, n, d_out, cap, nout);
    case 5: return dedup_t<5>(ctx, d_grouped, n, d_out, cap, nout);
    case 6: return dedup_t<6>(ctx, d_grouped, n, d_out, cap, nout);
    case 7: return dedup_t<7>(ctx, d_grouped, n, d_out, cap, nout);
    default: return (FK_EUNSUPPORTED);              // the caller keeps the four-pass grouping
  }
}
