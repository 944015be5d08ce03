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
#define AG_HB      3072                 // LDS-private histogram bins
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
  static constexpr int SDW = (KW <= 3) ? 4 : 8;         // dwords of a slot: key, zero padding, count in the last one
  static constexpr size_t LDS = (size_t) CAP * SDW * 4 + CAP * 2 + AG_HB * 4 + 256;
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

// a value the optimiser cannot see through (keeps it from hoisting address arithmetic out of the bin loop and
// spilling the results)
__device__ __forceinline__ u32 ag_opaque(u32 x) { FK_OPAQUE(x); return (x); }

// Cell of the counting sort (low 13 bits) and 12 further bits for step C2: a hash of the key made of full-rate 24-bit
// multiplies (fk_rec_hash's 64-bit products are a quarter of the kernel's arithmetic).  It only spreads the k-mers of
// one bin over the cells -- a poor spread costs time, never correctness -- and is independent of the bin's own bits.
template <int KW>
__device__ __forceinline__ u32 ag_cellhash(const u32 (&key)[KW])
{ // a sum of 24-bit products of the key's pieces with odd constants (v_mad_u32_u24: three instructions per dword),
  // then a finish that brings the high bits -- which depend on everything -- down into the cell and the wave bits
  u32 h = 0x9E3779B9u;
#pragma unroll
  for (int w = 0; w < KW; w++)
    { h = __umul24(key[w], 0x5bd1e9u + 0x22a3c4u * (u32) w) + h;
      h = __umul24(key[w] >> 8, 0x3c6ef3u + 0x1b56c2u * (u32) w) + h;
    }
  h ^= h >> 15;
  h = __umul24(h, 0x2f0b4fu) ^ __umul24(h >> 8, 0x68e31du);
  return (h ^ (h >> 13));
}

// inclusive scan over the 64 lanes of a wave with data-parallel-primitive adds (no LDS crossbar round trips):
// row_shr 1, 2, 4, 8 inside rows of 16 lanes, then lane 15 of rows 0 / 2 into rows 1 / 3, then lane 31 into rows 2, 3
__device__ __forceinline__ u32 ag_wave_scan(u32 x)
{ x += (u32) __builtin_amdgcn_update_dpp(0, (int) x, 0x111, 0xf, 0xf, false);
  x += (u32) __builtin_amdgcn_update_dpp(0, (int) x, 0x112, 0xf, 0xf, false);
  x += (u32) __builtin_amdgcn_update_dpp(0, (int) x, 0x114, 0xf, 0xf, false);
  x += (u32) __builtin_amdgcn_update_dpp(0, (int) x, 0x118, 0xf, 0xf, false);
  x += (u32) __builtin_amdgcn_update_dpp(0, (int) x, 0x142, 0xa, 0xf, false);
  x += (u32) __builtin_amdgcn_update_dpp(0, (int) x, 0x143, 0xc, 0xf, false);
  return (x);
}

// the 16 wave totals in tmp[] -> (sum of the waves in front of this one, sum of all); every row of 16 lanes scans them
__device__ __forceinline__ u32 ag_wave_bases(const u32 *tmp, u32 *total)
{ const u32 wave = threadIdx.x >> 6;
  u32 y = tmp[fk_lane() & (AG_WAVES - 1)];
  y += (u32) __builtin_amdgcn_update_dpp(0, (int) y, 0x111, 0xf, 0xf, false);
  y += (u32) __builtin_amdgcn_update_dpp(0, (int) y, 0x112, 0xf, 0xf, false);
  y += (u32) __builtin_amdgcn_update_dpp(0, (int) y, 0x114, 0xf, 0xf, false);
  y += (u32) __builtin_amdgcn_update_dpp(0, (int) y, 0x118, 0xf, 0xf, false);
  *total = (u32) __builtin_amdgcn_readlane((int) y, AG_WAVES - 1);
  const u32 prev = (u32) __builtin_amdgcn_readlane((int) y, (int) ((wave + AG_WAVES - 1) & (AG_WAVES - 1)));
  return ((wave == 0) ? 0u : prev);
}

// exclusive scan over the 1024 threads of the block with ONE barrier.  tmp: AG_WAVES u32 of LDS that nobody has
// touched since the barrier before last (the callers alternate between two arrays).
__device__ __forceinline__ u32 ag_block_exscan(u32 v, u32 *tmp, u32 *total)
{ const u32 x = ag_wave_scan(v);
  if (fk_lane() == 63) tmp[threadIdx.x >> 6] = x;
  __syncthreads();
  return (ag_wave_bases(tmp, total) + x - v);
}

// In-place exclusive scan of the CAP = NS * 1024 cell counters (16 bits each, two to a dword): thread t owns cells
// NS*t .. NS*t + NS - 1.  Returns the total.  One barrier inside; the caller puts one behind it before a cell is read.
template <int NS>
__device__ __forceinline__ u32 ag_scan_cells(u32 *cell, u32 *tmp)
{ u32 *mine = cell + threadIdx.x * (NS / 2);
  u32 c[NS / 2];
  if (NS == 8)
    { const uint4 q = *(const uint4 *) mine;
      c[0] = q.x; c[1] = q.y; c[NS / 2 - 2] = q.z; c[NS / 2 - 1] = q.w;
    }
  else
    { const uint2 q = *(const uint2 *) mine;
      c[0] = q.x; c[1] = q.y;
    }
  u32 s = 0;
#pragma unroll
  for (int i = 0; i < NS / 2; i++)
    { const u32 lo = c[i] & 0xffffu, hi = c[i] >> 16;
      c[i] = s | ((s + lo) << 16);                       // exclusive prefixes of the pair, relative to the thread
      s += lo + hi;
    }
  const u32 x = ag_wave_scan(s);
  if (fk_lane() == 63) tmp[threadIdx.x >> 6] = x;
  __syncthreads();
  u32 total;
  const u32 base = ag_wave_bases(tmp, &total) + x - s;
  const u32 b2 = base * 0x10001u;                        // offsets stay below 2^16: no carry between the halves
  if (NS == 8)
    *(uint4 *) mine = make_uint4(c[0] + b2, c[1] + b2, c[NS / 2 - 2] + b2, c[NS / 2 - 1] + b2);
  else
    *(uint2 *) mine = make_uint2(c[0] + b2, c[1] + b2);
  return (total);
}

// scal: [0] max_inst  [1] distinct k-mers  [2] table entries  [3] failure flag  [4] bin ticket
//       [5] extra rounds taken  [6] table buffer too small (tcap records)
//
// One persistent workgroup per CU takes hash bins in turn.  A bin (<= CAP records, all copies of a k-mer among
// them) is summed by a COUNTING SORT IN LDS on 13 further hash bits followed by a leader search -- no hash-table
// protocol (compare-and-swap claims, locked slots, retries of the lanes that lost a race):
//   A   every thread holds NS records in registers; cell = hash & (CAP - 1); rank = atomic counter of the cell
//   S   exclusive scan of the CAP cell counters in place
//   B   position p = scanned counter + rank; slot[p] = (key, weight) with one 16-byte LDS write
//   C   a record with rank > 0 compares itself with the records in front of it in its cell (1.3 on average): the
//       first equal one is the k-mer's LEADER and takes the weight (one LDS add), the record's own count becomes 0
//   H   a slot with a count != 0 is a distinct k-mer: histogram, max_inst, table candidate
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
  constexpr int SDW = AgCfg<KW>::SDW;
  FK_DYN_LDS(uint4, ag_lds);
  u32 *slot    = (u32 *) ag_lds;                           // [CAP][SDW] key + count, by sorted position
  u32 *cell    = slot + CAP * SDW;                         // [CAP / 2] cell counters -> offsets, 16 bits each
  u32 *lhist   = cell + CAP / 2;                           // [AG_HB]
  u32 *sh_tmp  = lhist + AG_HB;                            // [2][AG_WAVES]
  u64 *sh_base = (u64 *) (sh_tmp + 2 * AG_WAVES);
  const int tid = threadIdx.x;
  const u32 lane = fk_lane();
  // thread index that the optimiser cannot see through: used where a phase derives per-record addresses from it,
  // so that the 8 x 3 address registers are computed where they are used instead of being kept (spilled) all along
#define AG_TID(t) u32 t = (u32) tid; FK_OPAQUE(t)

  for (int i = tid; i < CAP / 2; i += AG_THREADS)
    cell[i] = 0;
  for (int i = tid; i < AG_HB; i += AG_THREADS)
    lhist[i] = 0;
  u64 my_max = 0;
  u32 my_distinct = 0, my_rounds = 0;
  u32 R0 = 1, flip = 0;
#ifdef FK_ABLATION
  u64 ph[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, tlast = __builtin_readcyclecounter();
#endif
  u32 kmask[KW];                                // key bytes of each record dword (pad and weight off)
#pragma unroll
  for (int w = 0; w < KW; w++)
    kmask[w] = (4 * w + 4 <= kbytes) ? 0xffffffffu : (4 * w < kbytes) ? ((1u << (8 * (kbytes - 4 * w))) - 1u) : 0u;
  __syncthreads();

  // The next bin is brought into L2 while the current one is swept: one dword per 128-byte line and thread, into
  // a register that is looked at when the bin's own loads have arrived.  (Fetching the records themselves ahead, into
  // registers, did not survive the register allocator: 24 more live registers per thread ended in scratch, and a
  // reload from scratch waits for every load in flight.)
  u32 touch = 0;

  // Groups of 2^gshift neighbouring bins (one bin when the input is large) are dealt round-robin: hashing makes
  // them equally heavy.  The bounds of a group are fetched two groups ahead with a VECTOR load (lane 0: first
  // record, lane 1: end): a scalar load shares its counter with LDS, so the first LDS wait behind it would sit
  // out a trip to memory.
  const u32 nbins = (u32) (AG_BINS >> gshift);
#define AG_BOUNDS_OF(b) bounds[(size_t) min((b) + ag_opaque(lane & 1u), nbins) << gshift]
  u32 bin = blockIdx.x;
  int64_t beg = 0, end = 0;
  u64 nxv = 0;                                  // bounds of the next group: loaded a whole group ahead, made scalars when used
#define AG_SCALAR64(x, l) ((int64_t) (((u64) (u32) __builtin_amdgcn_readlane((int) ((x) >> 32), l) << 32) \
                                       | (u32) __builtin_amdgcn_readlane((int) (x), l)))
  { const u64 b0 = (bin < nbins) ? AG_BOUNDS_OF(bin) : 0ull;
    nxv = (bin + gridDim.x < nbins) ? AG_BOUNDS_OF(bin + gridDim.x) : 0ull;
    beg = AG_SCALAR64(b0, 0);
    end = AG_SCALAR64(b0, 1);
  }
  for (; bin < nbins; bin += gridDim.x)
    { // (the bounds of the next group become scalars -- which waits for every memory operation in flight -- in
      // the sweep, before the table entries are stored, not here behind the stores of the previous group)
      int64_t nx_beg = 0, nx_end = 0;
      bool    have_nx = false;
#define AG_ADVANCE() do { if (!have_nx) { nx_beg = AG_SCALAR64(nxv, 0); nx_end = AG_SCALAR64(nxv, 1); have_nx = true; \
                                           nxv = (bin + 2 * gridDim.x < nbins) ? AG_BOUNDS_OF(bin + 2 * gridDim.x) : 0ull; } } while (0)
      if (beg < end)
      {
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
          u32     v[NS], key[NS][KW], wgt[NS];
          for (;;)
            { // ---- fill the register slots: slot j of thread tid is "lane position" j * 1024 + tid
              u32 isnew = 0;
              if (carried == 0)
                { const int64_t room = end - pos;
                  const u32 nnew = (room < (int64_t) cap_eff) ? (u32) room : (u32) cap_eff;
                  { const u32 *bp = recs + pos * KW;
                      AG_TID(t);
#pragma unroll
                      for (int j = 0; j < NS; j++)
                        { const u32 L = min((u32) (j * AG_THREADS) + t, nnew - 1);    // (a slot beyond nnew is not used)
                          const ag_rec<KW> rr = *(const ag_rec<KW> *) (bp + L * KW);
#pragma unroll
                          for (int w = 0; w < KW; w++)
                            key[j][w] = rr.w[w];
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
                  u32 k = ag_block_exscan((u32) __popc(fr), sh_tmp + (flip ^= AG_WAVES), &totfree);
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
#ifdef FK_ABLATION
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // (timers: the loads' latency apart from step A)
#endif
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
                    { if (R > 1 && ((isnew >> j) & 1u))
                        { u32 ha, hb;                          // the selections of a bin go by the bits above the bin's
                          fk_rec_hash<KW>(key[j], kbytes, ha, hb);
                          if (((hb >> 16) & (R - 1)) != r)
                            continue;                          // not in this selection
                        }
                      vmask |= (1u << j);
                      const u32 hc = ag_cellhash<KW>(key[j]);
                      sub[j]  = (hc & (CAP - 1)) | ((hc >> 13) << 16);    // cell, and 12 more bits for step C
                      const u32 sh = (sub[j] & 1u) << 4;
                      rank[j] = (atomicAdd(&cell[(sub[j] & 0xffffu) >> 1], 1u << sh) >> sh) & 0xffffu;
                    }
                }
              __syncthreads();
              AG_T(1);
              ntot = ag_scan_cells<NS>(cell, sh_tmp + (flip ^= AG_WAVES));
              __syncthreads();
              AG_T(2);

              // ---- B: positions; key and weight to their slot
              u32 p[NS], q[NS];
#pragma unroll
              for (int j = 0; j < NS; j++)
                { q[j] = 0;
                  if ((vmask >> j) & 1u)
                    q[j] = (cell[(sub[j] & 0xffffu) >> 1] >> ((sub[j] & 1u) << 4)) & 0xffffu;
                }
#pragma unroll
              for (int j = 0; j < NS; j++)
                { p[j] = q[j] + rank[j];
                  if ((vmask >> j) & 1u)
                    { u32 *sp = slot + p[j] * SDW;
                      if (SDW == 4)
                        *(uint4 *) sp = make_uint4(key[j][0], KW > 1 ? key[j][KW > 1 ? 1 : 0] : 0u,
                                                   KW > 2 ? key[j][KW > 2 ? 2 : 0] : 0u, wgt[j]);
                      else
                        { *(uint4 *) sp = make_uint4(key[j][0], key[j][1], key[j][2], key[j][KW > 3 ? 3 : 0]);
                          *(uint4 *) (sp + 4) = make_uint4(KW > 4 ? key[j][KW > 4 ? 4 : 0] : 0u, KW > 5 ? key[j][KW > 5 ? 5 : 0] : 0u,
                                                           KW > 6 ? key[j][KW > 6 ? 6 : 0] : 0u, wgt[j]);
                        }
                    }
                }
              __syncthreads();
              AG_T(3);

              // ---- C: every k-mer of the fill gets a LEADER, the record that collects its count.
              // C1: a record that is not the first of its cell compares itself with the first: four out of five are
              //     the same k-mer.  (The cell counters are dead: they become the table of C2, all ones.)
              // C2: the others are records of a k-mer that shares its cell with another: the one at the lowest position
              //     per 12 further hash bits (atomic min) is the leader of all that are equal to it.
              // C3: what is left (two such k-mers with the same 12 bits) looks through its cell from the front.
              // A search through the cell alone takes as many steps as the longest wait of a new k-mer behind the
              // copies of another -- a dozen for every wave on 50x data.
              constexpr int T2 = CAP / 2;
              if (NS == 8) *(uint4 *) (cell + tid * (NS / 2)) = make_uint4(~0u, ~0u, ~0u, ~0u);
              else         *(uint2 *) (cell + tid * (NS / 2)) = make_uint2(~0u, ~0u);
              u32 need = 0;
#pragma unroll
              for (int g = 0; g < NS; g += 4)                   // four records at a time: their reads travel together
                { uint4 s0[4], s1[4];
#pragma unroll
                  for (int i = 0; i < 4; i++)
                    { s0[i] = *(const uint4 *) (slot + q[g + i] * SDW);
                      if (SDW == 8) s1[i] = *(const uint4 *) (slot + q[g + i] * SDW + 4);
                    }
#pragma unroll
                  for (int i = 0; i < 4; i++)
                    { const int j = g + i;
                      u32 e = s0[i].x ^ key[j][0];
                      if (KW > 1) e |= s0[i].y ^ key[j][KW > 1 ? 1 : 0];
                      if (KW > 2) e |= s0[i].z ^ key[j][KW > 2 ? 2 : 0];
                      if (KW > 3) e |= s0[i].w ^ key[j][KW > 3 ? 3 : 0];
                      if (KW > 4) e |= s1[i].x ^ key[j][KW > 4 ? 4 : 0];
                      if (KW > 5) e |= s1[i].y ^ key[j][KW > 5 ? 5 : 0];
                      if (KW > 6) e |= s1[i].z ^ key[j][KW > 6 ? 6 : 0];
                      if (((vmask >> j) & 1u) && p[j] != q[j])
                        { if (e == 0)
                            { atomicAdd(slot + q[j] * SDW + (SDW - 1), wgt[j]);
                              slot[p[j] * SDW + (SDW - 1)] = 0;
                            }
                          else
                            need |= (1u << j);
                        }
                    }
                }
              __syncthreads();
#pragma unroll
              for (int j = 0; j < NS; j++)
                if ((need >> j) & 1u)
                  atomicMin(&cell[(sub[j] >> 16) & (T2 - 1)], p[j]);
              __syncthreads();
#pragma unroll
              for (int j = 0; j < NS; j++)
                if ((need >> j) & 1u)
                  { const u32 m = cell[(sub[j] >> 16) & (T2 - 1)];
                    if (m == p[j])
                      need &= ~(1u << j);                       // the leader
                    else
                      { const u32 *sp = slot + m * SDW;
                        u32 e = 0;
#pragma unroll
                        for (int w = 0; w < KW; w++)
                          e |= sp[w] ^ key[j][w];
                        if (e == 0)
                          { atomicAdd(slot + m * SDW + (SDW - 1), wgt[j]);
                            slot[p[j] * SDW + (SDW - 1)] = 0;
                            need &= ~(1u << j);
                          }
                      }
                  }
#pragma unroll
              for (int j = 0; j < NS; j++)
                if ((need >> j) & 1u)
                  { u32 qq = q[j] + 1;                            // (the first of the cell is another k-mer)
                    bool found = false;
                    while (qq < p[j])
                      { const u32 *sp = slot + qq * SDW;
                        u32 e = 0;
#pragma unroll
                        for (int w = 0; w < KW; w++)
                          e |= sp[w] ^ key[j][w];
                        if (e == 0) { found = true; break; }
                        qq += 1;
                      }
                    if (found)
                      { atomicAdd(slot + qq * SDW + (SDW - 1), wgt[j]);
                        slot[p[j] * SDW + (SDW - 1)] = 0;
                      }
                  }
              __syncthreads();
              AG_T(4);

              // ---- harvest: counts by position; the cell counters are zeroed for the next fill
              if (NS == 8) *(uint4 *) (cell + tid * (NS / 2)) = make_uint4(0, 0, 0, 0);
              else         *(uint2 *) (cell + tid * (NS / 2)) = make_uint2(0, 0);
              { AG_TID(t);
#pragma unroll
                for (int j = 0; j < NS; j++)
                  { const u32 L = (u32) (j * AG_THREADS) + t;
                    v[j] = (L < ntot) ? slot[L * SDW + (SDW - 1)] : 0u;
                  }
              }
              if (pos >= end)
                break;
              // more of the bin to come: the leaders stay, as records that carry their counts
              u32 nl = 0, D;
#pragma unroll
              for (int j = 0; j < NS; j++)
                nl += (v[j] != 0);
              ag_block_exscan(nl, sh_tmp + (flip ^= AG_WAVES), &D);
              if (D > (u32) limit)
                { ovf = true;
                  break;
                }
              vmask = 0;
              AG_TID(tc);
#pragma unroll
              for (int j = 0; j < NS; j++)
                if (v[j] != 0)
                  { const u32 *sp = slot + ((u32) (j * AG_THREADS) + tc) * SDW;
                    vmask |= (1u << j);
#pragma unroll
                    for (int w = 0; w < KW; w++)
                      key[j][w] = sp[w];
                    if (v[j] >= AG_HIGH)                       // stays far above 0x7fff: still saturated
                      { v[j] -= AG_CUT;
                        round_max += AG_CUT;
                      }
                    wgt[j] = v[j];
                  }
              carried = D;
              my_rounds += (tid == 0);
              __syncthreads();                                 // the slots are read: the next fill may write them
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

          // ---- what comes next: another selection of this bin (in L2 already), or the workgroup's next bin -- one dword
          // per line of it is asked for now
          { u32 Rn = R, rn = r;
            while (Rn > R0 && rn >= (Rn >> 1))
              { rn -= (Rn >> 1);
                Rn >>= 1;
              }
            AG_ADVANCE();
            if ((Rn == R0) && (r0 + 1 >= R0) && nx_beg < nx_end)
              { const int64_t room = (nx_end - nx_beg) * KW;                   // dwords
                const u32 nd = (room < (int64_t) cap_eff * KW) ? (u32) room : (u32) cap_eff * KW;
                const u32 d = min(ag_opaque((u32) tid) * 32u, nd - 1);
                // (a plain load that the empty asm below keeps alive -- a volatile one becomes a FLAT load with
                // system scope, and a flat load counts as an LDS operation too: LDS waits behind it sit out its trip)
                FK_KEEP(touch);                                // (the previous one has long arrived)
                touch = recs[nx_beg * KW + d];
              }
          }

          // ---- emit.  First the number of table entries: their place in the table buffer comes from a global atomic,
          // a trip to memory that the histogram work below covers.
          my_max += round_max;
          u32 nq = 0, nlead = 0;
#pragma unroll
          for (int j = 0; j < NS; j++)
            { const u32 vv = v[j];
              nlead += (vv != 0);
              nq    += DEDUP ? (vv != 0) : (cutoff > 0 && vv >= (u32) cutoff);
            }
          const bool tab = (DEDUP || cutoff > 0) && !(variant & 4);
          u32 tot = 0, toff = 0;
          u64 tbase = 0;
          if (tab)
            { toff = ag_block_exscan(nq, sh_tmp + (flip ^= AG_WAVES), &tot);
              if (tid == 0 && tot > 0)
                tbase = atomicAdd(&scal[2], (u64) tot);
            }

          // histogram and totals (the k-mers seen once or twice -- four out of five on read sets with sequencing
          // errors -- are counted per wave with a ballot: one LDS atomic per lane on the same two histogram bins
          // serialised the sweep)
          my_distinct += nlead;
          if (!DEDUP)
            { u32 n1 = 0, n2 = 0;
#pragma unroll
              for (int j = 0; j < NS; j++)
                { const u32 vv = v[j];
                  n1 += (u32) __popcll(__ballot(vv == 1u));
                  n2 += (u32) __popcll(__ballot(vv == 2u));
                  if (vv > 2u && !(variant & 2))
                    { if (vv >= sat)                           // sat = 0x7fff, MSDsort.c:498-506
                        my_max += vv;
                      const u32 cc = min(vv, 0x7fffu);
                      if (cc < AG_HB) atomicAdd(&lhist[cc], 1u);
                      else            atomicAdd(&hist_g[cc], 1ull);
                    }
                }
              if (!(variant & 2) && lane == 0)
                { if (n1 != 0) atomicAdd(&lhist[1], n1);
                  if (n2 != 0) atomicAdd(&lhist[2], n2);
                }
            }
          if (R0 > 1)
            { u32 D;
              ag_block_exscan(nlead, sh_tmp + (flip ^= AG_WAVES), &D);
              bin_fill = max(bin_fill, D);
            }
          AG_T(5);
          if (tab)
            { if (tid == 0)
                *sh_base = tbase;
              __syncthreads();
              AG_T(6);
              if (tot > 0 && *sh_base + tot > tcap)
                { if (tid == 0)                       // the table buffer is full (direct append to a union buffer)
                    atomicAdd(&scal[6], 1ull);
                }
              else if (tot > 0)
                { u64 o = *sh_base + toff;
                  AG_TID(t);
#pragma unroll
                  for (int j = 0; j < NS; j++)
                    if (DEDUP ? (v[j] != 0) : (v[j] >= (u32) cutoff))
                      { const u32 *sp = slot + ((u32) (j * AG_THREADS) + t) * SDW;
                        u32 kd[KW];
#pragma unroll
                        for (int w = 0; w < KW; w++)
                          kd[w] = sp[w];
                        if (DEDUP)
                          { ag_rec<KW + 1> ro;
#pragma unroll
                            for (int w = 0; w < KW; w++)
                              ro.w[w] = kd[w];
                            ro.w[KW] = v[j];
                            *(ag_rec<KW + 1> *) (table + o * (KW + 1)) = ro;
                          }
                        else
                          { ag_rec<KW> ro;
#pragma unroll
                            for (int w = 0; w < KW - 1; w++)
                              ro.w[w] = kd[w];
                            ro.w[KW - 1] = kd[KW - 1] | (min(v[j], 0x7fffu) << 16);
                            *(ag_rec<KW> *) (table + o * KW) = ro;
                          }
                        o += 1;
                      }
                }
            }
          if (!tab)
            __syncthreads();                      // (the cells zeroed above meet the next fill's counters)
          AG_T(7);                                // (otherwise no barrier: the next fill meets two before it writes a slot)

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
      AG_ADVANCE();
      beg = nx_beg; end = nx_end;
    }
#undef AG_ADVANCE
  FK_KEEP(touch);
#undef AG_SCALAR64
#undef AG_BOUNDS_OF
#undef AG_TID

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
  if (lane == 0)
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

// ---------------------------------------------------------------------------------------------
// k_ag_count2 (round 4): the same result with no counting sort.  Every phase of k_ag_count is a chain of dependent
// LDS round trips that all sixteen waves walk at the same time (nothing hides a trip) -- this
// version has fewer links in the chain and lets the waves walk part of it on their own:
//   A   a record is stored at its OWN position (j * 1024 + tid: a linear 16-byte write, no bank conflicts) and
//       writes that position into head[cell] with a plain 16-bit store.  The stores of a cell race; whichever
//       stays is a complete position, and any record of the cell is as good a representative as any other -- no
//       returning atomic, no scan of the counters, no second pass that moves the records (steps A, S, B above)
//   C1  every record compares itself with its cell's representative: the same k-mer -> its weight goes there
//       (one LDS add) and it is dead; the representative itself is that k-mer's leader
//   C2  the records of the other k-mers of a shared cell (one in ten) are dealt to the WAVES by four further hash
//       bits (all records of a k-mer to the same wave; an LDS counter per wave hands out the places in its list).
//       One barrier, then every wave elects the leaders of its records on its own: racing 16-bit stores into the
//       wave's table of 128, the record that stays is the leader of all that are equal to it, the others of other
//       k-mers go on to the next round with another hash -- a wave's LDS operations execute in order and nobody
//       else touches its list and table, so the rounds need no workgroup barrier.  A full list sets a flag: the
//       fill then elects workgroup-wide (two tables of 2048 in turn, one barrier per round)
//   H   as before, except that the table candidates are compacted too: their positions go into a list and thread
//       i writes candidate i (coalesced 12-byte stores, one histogram update per candidate instead of eight
//       predicated steps per thread) when the cutoff is 3 or more (the default -t4); and the whole emit of a
//       selection is deferred to the top of the next fill, behind the loads of that fill's records
// Chunks of bins beyond a fill, selections, limits, saturation: exactly as in k_ag_count.
#define AG2_TSZ  2048                   // entries of one of the workgroup's two election tables (fallback)
#define AG2_LCAP 256                    // entries of a wave's list
#define AG2_TW   128                    // entries of a wave's election table

template <int KW> struct AgCfg2
{ static constexpr int HB = 768;                       // LDS-private histogram bins
  static constexpr size_t LDS = (size_t) AgCfg<KW>::CAP * AgCfg<KW>::SDW * 4 + AgCfg<KW>::CAP * 2
                                + AG_WAVES * (AG2_LCAP + AG2_TW) * 2 + HB * 4 + 256;
  static_assert(AG_WAVES * (AG2_LCAP + AG2_TW) >= 2 * AG2_TSZ, "the workgroup's tables lie over the waves' lists and tables");
};

template <int KW>
__global__ __launch_bounds__(AG_THREADS) void k_ag_count2(const u32 *__restrict__ recs,
                                                          const u64 *__restrict__ bounds, int kbytes,
                                                          int cutoff, u64 *__restrict__ hist_g,
                                                          u64 *__restrict__ scal, u32 *__restrict__ table,
                                                          int cap_eff, int limit, int gshift, u32 sat, u64 tcap, int lcap,
                                                          u32 nbins)       // bins = bounds[i << gshift] .. bounds[(i + 1) << gshift]
{ constexpr int CAP = AgCfg<KW>::CAP;
  constexpr int NS  = AgCfg<KW>::NS;
  constexpr int SDW = AgCfg<KW>::SDW;
  constexpr int HB  = AgCfg2<KW>::HB;
  FK_DYN_LDS(uint4, ag_lds);
  u32      *slot    = (u32 *) ag_lds;                      // [CAP][SDW] key + count, record j of thread t at j * 1024 + t
  uint16_t *head    = (uint16_t *) (slot + CAP * SDW);     // [CAP] cell -> a record of it; then lists of positions
  uint16_t *wl      = head + CAP;                          // [AG_WAVES][AG2_LCAP] the waves' lists of records to elect leaders for
  uint16_t *wt      = wl + AG_WAVES * AG2_LCAP;            // [AG_WAVES][AG2_TW] the waves' election tables
  uint16_t *etab    = wl;                                  // [2][AG2_TSZ] the workgroup's election tables (instead of both)
  u32      *lhist   = (u32 *) (wt + AG_WAVES * AG2_TW);    // [HB]
  u32      *sh_tmp  = lhist + HB;                          // [2][AG_WAVES]
  u64      *sh_base = (u64 *) (sh_tmp + 2 * AG_WAVES);
  u32      *sh_flag = (u32 *) (sh_base + 1);               // [3] "somebody is still looking for a leader", by round
  u32      *lcnt    = sh_flag + 3;                         // [AG_WAVES] entries of the waves' lists
  u32      *sh_ovf  = lcnt + AG_WAVES;                     // a list was full
  const int tid = threadIdx.x;
  const u32 lane = fk_lane();
#define AG_TID(t) u32 t = (u32) tid; FK_OPAQUE(t)

  for (int i = tid; i < HB; i += AG_THREADS)
    lhist[i] = 0;
  if (tid < 3 + AG_WAVES + 1)
    sh_flag[tid] = 0;
  u64 my_max = 0;
  u32 my_distinct = 0, my_rounds = 0;
  u32 R0 = 1, flip = 0;
  u32 e1 = 0, e3 = 0;                           // election round counter, mod 2 and mod 3 (never reset)
#ifdef FK_ABLATION
  u64 ph[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, tlast = __builtin_readcyclecounter();
#endif
  u32 kmask[KW];
#pragma unroll
  for (int w = 0; w < KW; w++)
    kmask[w] = (4 * w + 4 <= kbytes) ? 0xffffffffu : (4 * w < kbytes) ? ((1u << (8 * (kbytes - 4 * w))) - 1u) : 0u;
  __syncthreads();

  // The emit of a selection is DEFERRED: its table space is reserved at once, but the histogram and table work runs
  // at the top of the next fill, behind the loads of that fill's records -- the trip to memory (a sixth of a bin's time
  // when it is waited for) is covered by it.  One load site and one emit site: fetching the next bin "ahead" from a
  // second site made the register allocator copy the loaded registers on the spot, i.e. wait for them there.
  u32  v[NS];                                  // counts by position of the fill that is being / still to be emitted
  bool pending = false;
  u32  e_tot = 0, e_toff = 0;
  u64  e_tbase = 0;
  const bool tab   = (cutoff > 0);
  const bool dense = (cutoff >= 3);            // every table candidate has a count above 2

  // the deferred part of an emit: histogram, then the candidates' sweep (table entries of e_tot candidates at e_tbase)
  auto emit_rest = [&]()
    { { u32 n1 = 0, n2 = 0, n3 = 0;
        const bool b3 = dense && cutoff >= 4;           // count 3 by ballot too
#pragma unroll
        for (int j = 0; j < NS; j++)
          { const u32 vv = v[j];
            n1 += (u32) __popcll(__ballot(vv == 1u));
            n2 += (u32) __popcll(__ballot(vv == 2u));
            if (b3)
              n3 += (u32) __popcll(__ballot(vv == 3u));
          }
        if (lane == 0)
          { if (n1 != 0) atomicAdd(&lhist[1], n1);
            if (n2 != 0) atomicAdd(&lhist[2], n2);
            if (n3 != 0) atomicAdd(&lhist[3], n3);
          }
        if (!dense || cutoff > 4)                       // counts that neither a ballot nor the candidates' sweep sees
          { const u32 lo = b3 ? 4u : 3u;
#pragma unroll
            for (int j = 0; j < NS; j++)
              { const u32 vv = v[j];
                if (vv >= lo && !(dense && vv >= (u32) cutoff))
                  { if (vv >= sat)
                      my_max += vv;
                    const u32 cc = min(vv, 0x7fffu);
                    if (cc < HB) atomicAdd(&lhist[cc], 1u);
                    else         atomicAdd(&hist_g[cc], 1ull);
                  }
              }
          }
      }
      if (dense)
        { // the candidates' positions, in table order, over the head region
          u32 o = e_toff;
          AG_TID(t);
#pragma unroll
          for (int j = 0; j < NS; j++)
            if (v[j] >= (u32) cutoff)
              head[o++] = (uint16_t) ((u32) (j * AG_THREADS) + t);
        }
      AG_T(5);
      if (tab)
        { if (tid == 0)
            *sh_base = e_tbase;
          __syncthreads();
          AG_T(6);
          const u64  tb0  = *sh_base;
          const bool full = (e_tot > 0 && tb0 + e_tot > tcap);
          if (full)
            { if (tid == 0)
                atomicAdd(&scal[6], 1ull);
            }
          if (dense)
            { if (!full)
                for (u32 i = (u32) tid; i < e_tot; i += AG_THREADS)
                  { const u32 *sp = slot + (u32) head[i] * SDW;
                    u32 kd[KW], vv;
                    if (SDW == 4)
                      { const uint4 q = *(const uint4 *) sp;
                        kd[0] = q.x;
                        if (KW > 1) kd[KW > 1 ? 1 : 0] = q.y;
                        if (KW > 2) kd[KW > 2 ? 2 : 0] = q.z;
                        vv = q.w;
                      }
                    else
                      {
#pragma unroll
                        for (int w = 0; w < KW; w++)
                          kd[w] = sp[w];
                        vv = sp[SDW - 1];
                      }
                    if (vv >= sat)
                      my_max += vv;
                    const u32 cc = min(vv, 0x7fffu);
                    if (cc < HB) atomicAdd(&lhist[cc], 1u);
                    else         atomicAdd(&hist_g[cc], 1ull);
                    ag_rec<KW> ro;
#pragma unroll
                    for (int w = 0; w < KW - 1; w++)
                      ro.w[w] = kd[w];
                    ro.w[KW - 1] = kd[KW - 1] | (cc << 16);
                    *(ag_rec<KW> *) (table + (tb0 + i) * KW) = ro;
                  }
            }
          else if (!full && e_tot > 0)
            { u64 o = tb0 + e_toff;
              AG_TID(t);
#pragma unroll
              for (int j = 0; j < NS; j++)
                if (v[j] >= (u32) cutoff)
                  { const u32 *sp = slot + ((u32) (j * AG_THREADS) + t) * SDW;
                    ag_rec<KW> ro;
#pragma unroll
                    for (int w = 0; w < KW - 1; w++)
                      ro.w[w] = sp[w];
                    ro.w[KW - 1] = sp[KW - 1] | (min(v[j], 0x7fffu) << 16);
                    *(ag_rec<KW> *) (table + o * KW) = ro;
                    o += 1;
                  }
            }
        }
      AG_T(7);
    };
#define AG_BOUNDS_OF(b) bounds[(size_t) min((b) + ag_opaque(lane & 1u), nbins) << gshift]
  u32 bin = blockIdx.x;
  int64_t beg = 0, end = 0;
  u64 nxv = 0;
#define AG_SCALAR64(x, l) ((int64_t) (((u64) (u32) __builtin_amdgcn_readlane((int) ((x) >> 32), l) << 32) \
                                       | (u32) __builtin_amdgcn_readlane((int) (x), l)))
  { const u64 b0 = (bin < nbins) ? AG_BOUNDS_OF(bin) : 0ull;
    nxv = (bin + gridDim.x < nbins) ? AG_BOUNDS_OF(bin + gridDim.x) : 0ull;
    beg = AG_SCALAR64(b0, 0);
    end = AG_SCALAR64(b0, 1);
  }
  for (; bin < nbins; bin += gridDim.x)
    { int64_t nx_beg = 0, nx_end = 0;
      bool    have_nx = false;
#define AG_ADVANCE() do { if (!have_nx) { nx_beg = AG_SCALAR64(nxv, 0); nx_end = AG_SCALAR64(nxv, 1); have_nx = true; \
                                           nxv = (bin + 2 * gridDim.x < nbins) ? AG_BOUNDS_OF(bin + 2 * gridDim.x) : 0ull; } } while (0)
      if (beg < end)
      {
      bool bin_ovf = false, failed = false;
      u32  bin_fill = 0;
      for (u32 r0 = 0; r0 < R0 && !failed; r0++)
      { u32 R = R0, r = r0;
      for (;;)
        { u64     round_max = 0;
          int64_t pos = beg;
          u32     carried = 0, vmask = 0;
          bool    ovf = false;
          u32     key[NS][KW], wgt[NS];
          for (;;)
            { // ---- fill the register slots (as in k_ag_count)
              u32 isnew = 0;
              if (carried == 0)
                { const int64_t room = end - pos;
                  const u32 nnew = (room < (int64_t) cap_eff) ? (u32) room : (u32) cap_eff;
                  { const u32 *bp = recs + pos * KW;
                    AG_TID(t);
#pragma unroll
                    for (int j = 0; j < NS; j++)
                      { const u32 L = min((u32) (j * AG_THREADS) + t, nnew - 1);
                        const ag_rec<KW> rr = *(const ag_rec<KW> *) (bp + L * KW);
#pragma unroll
                        for (int w = 0; w < KW; w++)
                          key[j][w] = rr.w[w];
                      }
                  }
                  if (pending)                     // the previous selection's histogram and table work, while these travel
                    { emit_rest();
                      pending = false;
                    }
#pragma unroll
                  for (int j = 0; j < NS; j++)
                    if ((u32) (j * AG_THREADS + tid) < nnew)
                      isnew |= (1u << j);
                  pos += nnew;
                }
              else
                { u32 fr = 0;
#pragma unroll
                  for (int j = 0; j < NS; j++)
                    if ((u32) (j * AG_THREADS + tid) < (u32) cap_eff && !((vmask >> j) & 1u))
                      fr |= (1u << j);
                  u32 totfree;
                  u32 k = ag_block_exscan((u32) __popc(fr), sh_tmp + (flip ^= AG_WAVES), &totfree);
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
#ifdef FK_ABLATION
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
              __syncthreads();                   // (the candidates' sweep of the previous fill has read its slots and list)
              AG_T(0);

              // ---- A: every record to its own slot, its position into the head of its cell
              u32 cj[NS];
              { AG_TID(t);
#pragma unroll
                for (int j = 0; j < NS; j++)
                  { if ((isnew >> j) & 1u)
                      { wgt[j] = key[j][KW - 1] >> 16;
#pragma unroll
                        for (int w = 0; w < KW; w++)
                          key[j][w] &= kmask[w];
                      }
                    cj[j] = 0;
                    if (((isnew | vmask) >> j) & 1u)
                      { if (R > 1 && ((isnew >> j) & 1u))
                          { u32 ha, hb;
                            fk_rec_hash<KW>(key[j], kbytes, ha, hb);
                            if (((hb >> 16) & (R - 1)) != r)
                              continue;
                          }
                        vmask |= (1u << j);
                        cj[j] = ag_cellhash<KW>(key[j]);
                        const u32 P = (u32) (j * AG_THREADS) + t;
                        u32 *sp = slot + P * SDW;
                        if (SDW == 4)
                          *(uint4 *) sp = make_uint4(key[j][0], KW > 1 ? key[j][KW > 1 ? 1 : 0] : 0u,
                                                     KW > 2 ? key[j][KW > 2 ? 2 : 0] : 0u, wgt[j]);
                        else
                          { *(uint4 *) sp = make_uint4(key[j][0], key[j][1], key[j][2], key[j][KW > 3 ? 3 : 0]);
                            *(uint4 *) (sp + 4) = make_uint4(KW > 4 ? key[j][KW > 4 ? 4 : 0] : 0u, KW > 5 ? key[j][KW > 5 ? 5 : 0] : 0u,
                                                             KW > 6 ? key[j][KW > 6 ? 6 : 0] : 0u, wgt[j]);
                          }
                        head[cj[j] & (CAP - 1)] = (uint16_t) P;
                      }
                  }
              }
              __syncthreads();
              AG_T(1);

              // ---- C1: compare with the cell's representative
              u32 need = 0, dead = 0;
              { u32 L[NS];
#pragma unroll
                for (int j = 0; j < NS; j++)
                  L[j] = ((vmask >> j) & 1u) ? (u32) head[cj[j] & (CAP - 1)] : 0u;
                AG_TID(t);
#pragma unroll
                for (int g = 0; g < NS; g += 4)
                  { uint4 s0[4], s1[4];
#pragma unroll
                    for (int i = 0; i < 4; i++)
                      { s0[i] = *(const uint4 *) (slot + L[g + i] * SDW);
                        if (SDW == 8) s1[i] = *(const uint4 *) (slot + L[g + i] * SDW + 4);
                      }
#pragma unroll
                    for (int i = 0; i < 4; i++)
                      { const int j = g + i;
                        u32 e = s0[i].x ^ key[j][0];
                        if (KW > 1) e |= s0[i].y ^ key[j][KW > 1 ? 1 : 0];
                        if (KW > 2) e |= s0[i].z ^ key[j][KW > 2 ? 2 : 0];
                        if (KW > 3) e |= s0[i].w ^ key[j][KW > 3 ? 3 : 0];
                        if (KW > 4) e |= s1[i].x ^ key[j][KW > 4 ? 4 : 0];
                        if (KW > 5) e |= s1[i].y ^ key[j][KW > 5 ? 5 : 0];
                        if (KW > 6) e |= s1[i].z ^ key[j][KW > 6 ? 6 : 0];
                        if (((vmask >> j) & 1u) && L[j] != (u32) (j * AG_THREADS) + t)
                          { if (e == 0)
                              { atomicAdd(slot + L[j] * SDW + (SDW - 1), wgt[j]);
                                dead |= (1u << j);
                              }
                            else
                              need |= (1u << j);
                          }
                      }
                  }
              }
              // The records still without a leader (one in ten: other k-mers of a shared cell) are dealt to the WAVES by four
              // further hash bits -- all records of a k-mer to the same wave -- so that the elections need no workgroup
              // barriers: a wave's LDS operations execute in order, and nobody else touches its list and its table.
              { AG_TID(t);
#pragma unroll
                for (int j = 0; j < NS; j++)
                  if ((need >> j) & 1u)
                    { const u32 w   = (cj[j] >> 13) & (AG_WAVES - 1);
                      const u32 idx = atomicAdd(&lcnt[w], 1u);
                      if (idx < (u32) lcap) wl[w * AG2_LCAP + idx] = (uint16_t) ((u32) (j * AG_THREADS) + t);
                      else                  *sh_ovf = 1;
                    }
              }
              __syncthreads();
              AG_T(2);
              if (*sh_ovf == 0)
                { const u32 wave = (u32) tid >> 6;
                  const u32 n = (u32) __builtin_amdgcn_readfirstlane((int) lcnt[wave]);
                  if (lane == 0) lcnt[wave] = 0;
                  u32 act = 0, P2[AG2_LCAP / 64], hh[AG2_LCAP / 64];
#pragma unroll
                  for (int k = 0; k < AG2_LCAP / 64; k++)
                    { P2[k] = 0; hh[k] = 0;
                      if ((u32) (k * 64) < n)
                        { const u32 i = (u32) (k * 64) + lane;
                          if (i < n)
                            { P2[k] = wl[wave * AG2_LCAP + i];
                              const u32 *sp = slot + P2[k] * SDW;
                              if (SDW == 4)
                                { const uint4 q = *(const uint4 *) sp;
                                  key[k][0] = q.x;
                                  if (KW > 1) key[k][KW > 1 ? 1 : 0] = q.y;
                                  if (KW > 2) key[k][KW > 2 ? 2 : 0] = q.z;
                                  wgt[k] = q.w;
                                }
                              else
                                {
#pragma unroll
                                  for (int w = 0; w < KW; w++)
                                    key[k][w] = sp[w];
                                  wgt[k] = sp[SDW - 1];
                                }
                              hh[k] = ag_cellhash<KW>(key[k]) >> 17;
                              act |= (1u << k);
                            }
                        }
                    }
                  uint16_t *tw = wt + wave * AG2_TW;
                  u32 rd = 1;
                  while (__ballot(act != 0) != 0)
                    { u32 hx[AG2_LCAP / 64];
#pragma unroll
                      for (int k = 0; k < AG2_LCAP / 64; k++)
                        { hx[k] = 0;
                          if ((act >> k) & 1u)
                            { hx[k] = (((hh[k] + 0x632be5abu * rd) * 0x9E3779B1u) >> 16) & (AG2_TW - 1);
                              tw[hx[k]] = (uint16_t) P2[k];
                            }
                        }
                      FK_EMU_WAVE_SYNC();        // (the wave's stores above are all done before its loads below: LDS operations of a wave execute in order)
#pragma unroll
                      for (int k = 0; k < AG2_LCAP / 64; k++)
                        if ((act >> k) & 1u)
                          { const u32 m = tw[hx[k]];
                            if (m == P2[k])
                              act &= ~(1u << k);                   // the leader
                            else
                              { const u32 *sp = slot + m * SDW;
                                u32 e = 0;
                                if (SDW == 4)
                                  { const uint4 q = *(const uint4 *) sp;
                                    e = q.x ^ key[k][0];
                                    if (KW > 1) e |= q.y ^ key[k][KW > 1 ? 1 : 0];
                                    if (KW > 2) e |= q.z ^ key[k][KW > 2 ? 2 : 0];
                                  }
                                else
                                  {
#pragma unroll
                                    for (int w = 0; w < KW; w++)
                                      e |= sp[w] ^ key[k][w];
                                  }
                                if (e == 0)
                                  { atomicAdd(slot + m * SDW + (SDW - 1), wgt[k]);
                                    slot[P2[k] * SDW + (SDW - 1)] = 0;
                                    act &= ~(1u << k);
                                  }
                              }
                          }
                      rd += 1;
                    }
                }
              else
                { // some wave's list is full (lcap records): elections of the whole workgroup, two tables in turn and one
                  // barrier per round, until a round finds nobody left (the lists and tables of the waves are not in use)
                  if (lane == 0) lcnt[(u32) tid >> 6] = 0;
                  // the records elect among themselves where they are, eight sparse steps per thread and round
                  bool first = true;
                  for (;;)
                    { uint16_t *tb = etab + e1 * AG2_TSZ;
                      const u32 rmul = 0x632be5abu * (e3 + 3 * e1 + 1);
                      { AG_TID(t);
#pragma unroll
                        for (int j = 0; j < NS; j++)
                          if ((need >> j) & 1u)
                            tb[((((cj[j] >> 13) + rmul) * 0x9E3779B1u) >> 16) & (AG2_TSZ - 1)] = (uint16_t) ((u32) (j * AG_THREADS) + t);
                      }
                      if (need != 0) sh_flag[e3] = 1;
                      if (tid == 0) sh_flag[(e3 == 2) ? 0 : e3 + 1] = 0;
                      __syncthreads();
                      const bool any = (sh_flag[e3] != 0);
                      if (first && tid == 0) *sh_ovf = 0;          // (everybody has read it)
                      first = false;
                      e1 ^= 1;
                      e3 = (e3 == 2) ? 0 : e3 + 1;
                      if (!any)
                        break;
                      AG_TID(t);
#pragma unroll
                      for (int j = 0; j < NS; j++)
                        if ((need >> j) & 1u)
                          { const u32 m = tb[((((cj[j] >> 13) + rmul) * 0x9E3779B1u) >> 16) & (AG2_TSZ - 1)];
                            if (m == (u32) (j * AG_THREADS) + t)
                              need &= ~(1u << j);                  // the leader
                            else
                              { const u32 *sp = slot + m * SDW;
                                u32 e = 0;
#pragma unroll
                                for (int w = 0; w < KW; w++)
                                  e |= sp[w] ^ key[j][w];
                                if (e == 0)
                                  { atomicAdd(slot + m * SDW + (SDW - 1), wgt[j]);
                                    dead |= (1u << j);
                                    need &= ~(1u << j);
                                  }
                              }
                          }
                    }
                }
              __syncthreads();
              AG_T(3);
              AG_T(4);

              // ---- harvest: the counts of the leaders
              { AG_TID(t);
                const u32 live = vmask & ~dead;
#pragma unroll
                for (int j = 0; j < NS; j++)
                  v[j] = ((live >> j) & 1u) ? slot[((u32) (j * AG_THREADS) + t) * SDW + (SDW - 1)] : 0u;
              }
              if (pos >= end)
                break;
              // more of the bin to come: the leaders stay where they are and carry their counts
              u32 nl = 0, D;
#pragma unroll
              for (int j = 0; j < NS; j++)
                nl += (v[j] != 0);
              ag_block_exscan(nl, sh_tmp + (flip ^= AG_WAVES), &D);
              if (D > (u32) limit)
                { ovf = true;
                  break;
                }
              vmask = 0;
              AG_TID(tc);
#pragma unroll
              for (int j = 0; j < NS; j++)
                if (v[j] != 0)
                  { const u32 *sp = slot + ((u32) (j * AG_THREADS) + tc) * SDW;
                    vmask |= (1u << j);
#pragma unroll
                    for (int w = 0; w < KW; w++)
                      key[j][w] = sp[w];
                    if (v[j] >= AG_HIGH)
                      { v[j] -= AG_CUT;
                        round_max += AG_CUT;
                      }
                    wgt[j] = v[j];
                  }
              carried = D;
              my_rounds += (tid == 0);
            }
          if (ovf)
            { my_rounds += (tid == 0);
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

          AG_ADVANCE();
          // ---- emit, first part: the table space of the candidates (a global atomic whose result is looked at in the
          // deferred part); the harvest loop above has left the counts in v[]
          my_max += round_max;
          { u32 nq = 0, nlead = 0;
#pragma unroll
            for (int j = 0; j < NS; j++)
              { const u32 vv = v[j];
                nlead += (vv != 0);
                nq    += (tab && vv >= (u32) cutoff);
              }
            e_tot = 0; e_toff = 0; e_tbase = 0;
            if (tab)
              { e_toff = ag_block_exscan(nq, sh_tmp + (flip ^= AG_WAVES), &e_tot);
                if (tid == 0 && e_tot > 0)
                  e_tbase = atomicAdd(&scal[2], (u64) e_tot);
              }
            my_distinct += nlead;
            if (R0 > 1)
              { u32 D;
                ag_block_exscan(nlead, sh_tmp + (flip ^= AG_WAVES), &D);
                bin_fill = max(bin_fill, D);
              }
          }
          pending = true;

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
      AG_ADVANCE();
      beg = nx_beg; end = nx_end;
    }
#undef AG_ADVANCE
  if (pending)
    emit_rest();
#undef AG_SCALAR64
#undef AG_BOUNDS_OF
#undef AG_TID

  __syncthreads();
  for (int i = tid; i < HB; i += AG_THREADS)
    if (lhist[i] != 0)
      atomicAdd(&hist_g[i], (u64) lhist[i]);
  u64 d = my_distinct, rd = my_rounds;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
    { my_max += __shfl_down(my_max, o, 64);
      d      += __shfl_down(d, o, 64);
      rd     += __shfl_down(rd, o, 64);
    }
  if (lane == 0)
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

// pre_bounds != NULL: the fills are given (pre_nbins + 1 record positions, fkx_ref_bounds) -- no hash bins are looked for
template <int KW>
static int aggr_t(fk_ctx *ctx, const void *d_grouped, int64_t n, int cutoff, int64_t *hist,
                  int64_t *max_inst, int64_t *ndistinct, void *d_table, int64_t cap, int64_t *ntable,
                  const u64 *pre_bounds = NULL, int64_t pre_nbins = 0)
{ hipStream_t s = ctx->stream;
  if (ntable) *ntable = 0;
  if (ndistinct) *ndistinct = 0;
  if (n == 0)
    return (FK_OK);
  if (cutoff > 0 && (d_table == NULL || cap <= 0))
    { fk_set_error(ctx, "aggregate: no table buffer");
      return (FK_EINVAL);
    }
  u64 *d_bounds = (pre_bounds != NULL) ? (u64 *) pre_bounds : (u64 *) fk_slot(ctx, FK_SLOT_AG_BOUNDS, (AG_BINS + 1) * 8);
  u64 *d_hist   = (u64 *) fk_slot(ctx, FK_SLOT_CT_HIST, (FK_HIST_BINS + AG_NSCAL) * 8);
  if (d_bounds == NULL || d_hist == NULL)
    return (FK_ENOMEM);
  u64 *d_scal = d_hist + FK_HIST_BINS;
  static bool attr_set[8] = { false };
  const size_t lds = AgCfg<KW>::LDS, lds2 = AgCfg2<KW>::LDS;
  if (!attr_set[KW])
    { auto kern = k_ag_count<KW, false>;
      FK_HIP(ctx, hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
      auto kern2 = k_ag_count2<KW>;
      FK_HIP(ctx, hipFuncSetAttribute((const void *) kern2, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds2));
      attr_set[KW] = true;
    }
  FK_HIP(ctx, hipMemsetAsync(d_hist, 0, (FK_HIST_BINS + AG_NSCAL) * 8, s));
  if (pre_bounds == NULL)
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
  if (pre_bounds != NULL)
    gshift = 0;
  const u32 nbins = (pre_bounds != NULL) ? (u32) pre_nbins : (u32) (AG_BINS >> gshift);
  int cap_eff = AgCfg<KW>::CAP;                       // fk_debug_set("aggr_limit"): a fill takes only this many records
  if (ctx->dbg_aggr_limit > 0 && ctx->dbg_aggr_limit < cap_eff)
    cap_eff = ctx->dbg_aggr_limit < 4 ? 4 : ctx->dbg_aggr_limit;
  const int limit = cap_eff * 3 / 4;
  if (pre_bounds == NULL && (ctx->dbg_aggr_engine == 1 || ctx->dbg_aggr_variant != 0))     // the counting sort of round 3 (kept for comparison)
    hipLaunchKernelGGL((k_ag_count<KW, false>), dim3((unsigned) cus), dim3(AG_THREADS), lds, s, (const u32 *) d_grouped,
                       (const u64 *) d_bounds, ctx->wid.kmer_bytes, cutoff, d_hist, d_scal, (u32 *) d_table, cap_eff, limit, ctx->dbg_aggr_variant, gshift,
                       (u32) (ctx->aggr_sat > 0 ? ctx->aggr_sat : 0x7fff), (u64) cap);
  else
    hipLaunchKernelGGL((k_ag_count2<KW>), dim3((unsigned) cus), dim3(AG_THREADS), lds2, s, (const u32 *) d_grouped,
                       (const u64 *) d_bounds, ctx->wid.kmer_bytes, cutoff, d_hist, d_scal, (u32 *) d_table, cap_eff, limit, gshift,
                       (u32) (ctx->aggr_sat > 0 ? ctx->aggr_sat : 0x7fff), (u64) cap,
                       ctx->dbg_aggr_engine == 2 ? 2 : AG2_LCAP,       // (engine 2: tiny lists, the fallback elections run)
                       nbins);
  FK_LAUNCH_CHECK(ctx);
  u64 *h = ctx->h_scratch;                       // pinned, 64 KB + 64 KB: the histogram needs 256 KB
  u64 *hh = (u64 *) malloc((FK_HIST_BINS + AG_NSCAL) * 8);
  if (hh == NULL) return (FK_ENOMEM);
  (void) h;
  if (fkx_d2h_pageable(ctx, s, hh, d_hist, (FK_HIST_BINS + AG_NSCAL) * 8) != FK_OK)      // (hh is malloc'ed: no asynchronous copy)
    { free(hh);
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
      fprintf(stderr, "  (engine 1: load, A, scan, B, C, harvest+hist, table reserve, table write; engine 2: load, A, C1, C2, -, harvest+hist, reserve, write)\n");
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
/* The same over fills that are given: d_bounds[0 .. nfills] = record positions, every k-mer's copies inside one fill
   (or one oversized group, which is taken in chunks / selections like a bin that is too large). */
int fkx_aggregate_fills(fk_ctx *ctx, const void *d_grouped, int64_t n, int cutoff, int64_t *hist, int64_t *max_inst,
                        int64_t *ndistinct, void *d_table, int64_t cap, int64_t *ntable, const u64 *d_bounds, int64_t nfills)
{ if (d_bounds == NULL || nfills <= 0 || nfills > 0x7fffffffll)
    return (FK_EINVAL);
  switch (ctx->wid.kmer_stride >> 2)
  { case 2: return aggr_t<2>(ctx, d_grouped, n, cutoff, hist, max_inst, ndistinct, d_table, cap, ntable, d_bounds, nfills);
    case 3: return aggr_t<3>(ctx, d_grouped, n, cutoff, hist, max_inst, ndistinct, d_table, cap, ntable, d_bounds, nfills);
    case 4: return aggr_t<4>(ctx, d_grouped, n, cutoff, hist, max_inst, ndistinct, d_table, cap, ntable, d_bounds, nfills);
    case 5: return aggr_t<5>(ctx, d_grouped, n, cutoff, hist, max_inst, ndistinct, d_table, cap, ntable, d_bounds, nfills);
    default:
      fk_set_error(ctx, "k-mer stride %d not built", ctx->wid.kmer_stride);
      return (FK_EUNSUPPORTED);
  }
}

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
