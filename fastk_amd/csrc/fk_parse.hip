// fk_parse.hip -- FASTQ text -> 0-terminated reads, on the device.
//
// Replaces the per-byte state machine of the reference's reader threads for FASTQ
// (fast_output_thread, io.c:574-759; line rules io.c:678-734: records are strictly four lines --
// header, sequence, '+' line, qualities -- and every non-newline byte of the sequence line is a
// base).  A byte belongs to a sequence line iff the number of newlines before it is 1 (mod 4), so
// the classification is an exclusive scan of newline counts; the kept bytes (sequence bytes, and the
// sequence line's newline turned into the 0 terminator that DATA_BLOCK.bases uses, FastK.h:87-98)
// are compacted with a second scan.  Quality lines may contain the letters ACGT, which is why they
// have to be removed by position and cannot be left to the splitter's non-ACGT rule.
//
//   k_fq_count  per tile of 16 KB: newlines, and the number of bytes whose local newline count is
//               0,1,2,3 (mod 4) -- the kept bytes of the tile for each phase it can start in
//   k_fq_scan   one workgroup: phase and output offset of every tile, totals
//   k_fq_emit   classify again, compact through LDS, write
#include "fk_common.h"
#include <algorithm>

#define FQ_THREADS 256
#define FQ_PER     64                         // bytes per thread
#define FQ_TILE    (FQ_THREADS * FQ_PER)

// per-thread pass over its 64 bytes; c0 = newlines before the thread's first byte (mod 4 suffices)
// prev: the byte in front of the thread's first one; hoco: drop a base equal to the byte before it
// (homopolymer compression, io.c:284-294; the byte before the first base of a line is a newline)
template <bool EMIT>
__device__ __forceinline__ void fq_walk(const uint4 (&v)[4], int nvalid, u32 c0, u32 (&cnt)[4], u32 &nl,
                                        u32 &reads, unsigned char *stage, u32 phase, u32 &kept, u32 prev,
                                        bool hoco)
{ const u32 *w = (const u32 *) v;
#pragma unroll
  for (int i = 0; i < FQ_PER; i++)
    { if (i < nvalid)
        { const u32 ch = (w[i >> 2] >> (8 * (i & 3))) & 0xffu;
          const u32 r  = (c0 + nl) & 3u;
          const bool drop = hoco && ch == prev && ch != '\n';
          prev = ch;
          if (drop)
            ;
          else if (!EMIT)
            { cnt[0] += (r == 0u); cnt[1] += (r == 1u); cnt[2] += (r == 2u); cnt[3] += (r == 3u); }
          else if (((phase + r) & 3u) == 1u)
            { stage[kept] = (unsigned char) (ch == '\n' ? 0 : ch);
              kept += 1;
              reads += (ch == '\n');
            }
          nl += (ch == '\n');
        }
    }
}

__device__ __forceinline__ void fq_load(const unsigned char *raw, int64_t n, int64_t base, uint4 (&v)[4],
                                        int &nvalid)
{ nvalid = (n - base >= FQ_PER) ? FQ_PER : (n > base ? (int) (n - base) : 0);
  if (nvalid == FQ_PER && ((uintptr_t) (raw + base) & 15) == 0)
    {
#pragma unroll
      for (int k = 0; k < 4; k++)
        v[k] = *(const uint4 *) (raw + base + 16 * k);
    }
  else
    { unsigned char *b = (unsigned char *) v;
#pragma unroll
      for (int k = 0; k < 4; k++)
        v[k] = make_uint4(0, 0, 0, 0);
      for (int i = 0; i < nvalid; i++)
        b[i] = raw[base + i];
    }
}

// tile_info[t*8 + 0..3] = bytes of the tile with local newline count r (mod 4), [4] = newlines
__global__ __launch_bounds__(FQ_THREADS) void k_fq_count(const unsigned char *__restrict__ raw, int64_t n,
                                                         u32 *__restrict__ tile_info, int hoco)
{ __shared__ u32 tmp[8];
  __shared__ u32 red[5];
  const int64_t base = (int64_t) blockIdx.x * FQ_TILE + (int64_t) threadIdx.x * FQ_PER;
  uint4 v[4];
  int   nvalid;
  fq_load(raw, n, base, v, nvalid);
  // newlines of this thread first (cheap), block scan, then the residue census with the right start
  u32 mynl = 0;
  { const u32 *w = (const u32 *) v;
#pragma unroll
    for (int i = 0; i < 16; i++)
      { const u32 x = w[i] ^ 0x0a0a0a0au;                                  // zero byte <=> newline
        const u32 z = (x - 0x01010101u) & ~x & 0x80808080u;
        mynl += __popc(z);
      }
    if (nvalid < FQ_PER)
      { mynl = 0;
        const unsigned char *b = (const unsigned char *) v;
        for (int i = 0; i < nvalid; i++)
          mynl += (b[i] == '\n');
      }
  }
  u32 tot;
  const u32 c0 = fk_block_exscan_256<u32>(mynl, tmp, &tot);
  u32 cnt[4] = { 0, 0, 0, 0 }, nl = 0, reads = 0, kept = 0;
  fq_walk<false>(v, nvalid, c0, cnt, nl, reads, NULL, 0, kept, (nvalid > 0) ? (u32) raw[base - 1] : 0u, hoco != 0);
  if (threadIdx.x < 5) red[threadIdx.x] = 0;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; r++)
    { u32 x = cnt[r];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1)
        x += __shfl_down(x, o, 64);
      if (fk_lane() == 0 && x)
        atomicAdd(&red[r], x);
    }
  __syncthreads();
  if (threadIdx.x < 4)
    tile_info[(int64_t) blockIdx.x * 8 + threadIdx.x] = red[threadIdx.x];
  if (threadIdx.x == 4)
    tile_info[(int64_t) blockIdx.x * 8 + 4] = tot;
}

// tile_phase[t] = line phase at the start of tile t, tile_off[t] = kept bytes before it;
// out[0] = kept bytes, out[1] = newlines (the caller derives the phase after the chunk)
__global__ __launch_bounds__(256) void k_fq_scan(const u32 *__restrict__ tile_info, int64_t ntiles, u32 phase0,
                                                 u32 *__restrict__ tile_phase, u64 *__restrict__ tile_off,
                                                 u64 *__restrict__ out)
{ __shared__ u64 tmp64[8];
  __shared__ u32 tmp32[8];
  u64 carry_off = 0;
  u32 carry_nl = 0;
  for (int64_t b = 0; b < ntiles; b += 256)
    { const int64_t t = b + threadIdx.x;
      const u32 mynl = (t < ntiles) ? tile_info[t * 8 + 4] : 0u;
      u32 totnl;
      const u32 exnl = fk_block_exscan_256<u32>(mynl, tmp32, &totnl);
      const u32 ph = (phase0 + carry_nl + exnl) & 3u;
      const u64 mykeep = (t < ntiles) ? (u64) tile_info[t * 8 + ((1u - ph) & 3u)] : 0ull;
      u64 totk;
      const u64 exk = fk_block_exscan_256<u64>(mykeep, tmp64, &totk);
      if (t < ntiles)
        { tile_phase[t] = ph;
          tile_off[t] = carry_off + exk;
        }
      carry_off += totk;
      carry_nl = (carry_nl + totnl) & 3u;          // only the phase matters; the total is summed below
      if (threadIdx.x == 0)
        out[1] += totnl;
    }
  if (threadIdx.x == 0)
    out[0] = carry_off;
}

__global__ __launch_bounds__(FQ_THREADS) void k_fq_emit(const unsigned char *__restrict__ raw, int64_t n,
                                                        const u32 *__restrict__ tile_phase,
                                                        const u64 *__restrict__ tile_off,
                                                        unsigned char *__restrict__ dst, u64 *__restrict__ nreads,
                                                        int hoco)
{ __shared__ unsigned char stage[FQ_TILE];
  __shared__ u32 tmp[8];
  __shared__ u32 s_reads;
  const int64_t base = (int64_t) blockIdx.x * FQ_TILE + (int64_t) threadIdx.x * FQ_PER;
  uint4 v[4];
  int   nvalid;
  fq_load(raw, n, base, v, nvalid);
  if (threadIdx.x == 0) s_reads = 0;
  // newline prefix as in k_fq_count
  u32 mynl = 0;
  { const unsigned char *b = (const unsigned char *) v;
    if (nvalid == FQ_PER)
      { const u32 *w = (const u32 *) v;
#pragma unroll
        for (int i = 0; i < 16; i++)
          { const u32 x = w[i] ^ 0x0a0a0a0au;
            mynl += __popc((x - 0x01010101u) & ~x & 0x80808080u);
          }
      }
    else
      for (int i = 0; i < nvalid; i++)
        mynl += (b[i] == '\n');
  }
  u32 tot;
  const u32 c0 = fk_block_exscan_256<u32>(mynl, tmp, &tot);
  const u32 phase = tile_phase[blockIdx.x];
  // how many bytes this thread keeps, then where they go inside the tile's output
  u32 cnt[4] = { 0, 0, 0, 0 }, nl = 0, reads = 0, kept = 0;
  const u32 prev0 = (nvalid > 0) ? (u32) raw[base - 1] : 0u;
  fq_walk<false>(v, nvalid, c0, cnt, nl, reads, NULL, 0, kept, prev0, hoco != 0);
  const u32 want = (1u - phase) & 3u;               // local newline count (mod 4) of the kept bytes
  const u32 mine = (want == 0u) ? cnt[0] : (want == 1u) ? cnt[1] : (want == 2u) ? cnt[2] : cnt[3];
  u32 tkept;
  const u32 ex = fk_block_exscan_256<u32>(mine, tmp, &tkept);
  nl = 0; kept = 0;
  fq_walk<true>(v, nvalid, c0, cnt, nl, reads, stage + ex, phase, kept, prev0, hoco != 0);
  if (reads)
    atomicAdd(&s_reads, reads);
  __syncthreads();
  unsigned char *o = dst + tile_off[blockIdx.x];
  for (u32 i = threadIdx.x; i < tkept; i += FQ_THREADS)
    o[i] = stage[i];
  if (threadIdx.x == 0 && s_reads)
    atomicAdd(nreads, (u64) s_reads);
}

/* d_raw: nbytes of FASTQ text on the device, with one readable byte in front of it (the last byte
   of the previous piece, or 0); *phase: newlines seen so far in this file (mod 4), updated.
   The sequence lines, each ended by a 0, are written to d_dst (capacity >= nbytes); *nkept bytes. */
int fkx_parse_fastq(fk_ctx *ctx, const void *d_raw, int64_t nbytes, int flags, int *phase, void *d_dst,
                    int64_t *nkept, int64_t *nreads)
{ hipStream_t s = ctx->stream;
  *nkept = 0; *nreads = 0;
  if (nbytes <= 0)
    return (FK_OK);
  const int64_t ntiles = (nbytes + FQ_TILE - 1) / FQ_TILE;
  u32 *d_info  = (u32 *) fk_slot(ctx, FK_SLOT_FQ_INFO, ntiles * 8 * 4);
  u32 *d_phase = (u32 *) fk_slot(ctx, FK_SLOT_FQ_PHASE, ntiles * 4);
  u64 *d_off   = (u64 *) fk_slot(ctx, FK_SLOT_FQ_OFF, ntiles * 8);
  if (d_info == NULL || d_phase == NULL || d_off == NULL)
    return (FK_ENOMEM);
  u64 *d_out = ctx->d_scratch + 2048;              // [0] kept bytes [1] newlines [2] reads
  FK_HIP(ctx, hipMemsetAsync(d_out, 0, 3 * sizeof(u64), s));
  hipLaunchKernelGGL(k_fq_count, dim3((unsigned) ntiles), dim3(FQ_THREADS), 0, s,
                     (const unsigned char *) d_raw, nbytes, d_info, flags & FK_FASTQ_HOCO);
  hipLaunchKernelGGL(k_fq_scan, dim3(1), dim3(256), 0, s, (const u32 *) d_info, ntiles, (u32) (*phase & 3),
                     d_phase, d_off, d_out);
  hipLaunchKernelGGL(k_fq_emit, dim3((unsigned) ntiles), dim3(FQ_THREADS), 0, s,
                     (const unsigned char *) d_raw, nbytes, (const u32 *) d_phase, (const u64 *) d_off,
                     (unsigned char *) d_dst, d_out + 2, flags & FK_FASTQ_HOCO);
  FK_LAUNCH_CHECK(ctx);
  FK_HIP(ctx, hipMemcpyAsync(ctx->h_scratch, d_out, 3 * sizeof(u64), hipMemcpyDeviceToHost, s));
  FK_HIP(ctx, hipStreamSynchronize(s));
  *nkept  = (int64_t) ctx->h_scratch[0];
  *phase  = (*phase & ~3) | (int) (((*phase & 3) + ctx->h_scratch[1]) & 3);
  *nreads = (int64_t) ctx->h_scratch[2];
  return (FK_OK);
}

// ---------------------------------------------------------------------------------------------
// FASTA text -> 0-terminated reads (the FASTA branch of fast_output_thread, io.c:574-759): a line
// that starts with '>' is a header, every other line is sequence, and the sequence lines of a record
// are one read (newlines inside a record are dropped, io.c:700-734).  A byte's line is a header iff
// the byte after the last newline before it is '>', so the classification is an exclusive MAX-scan
// of newline positions; every header start emits the 0 that ends the previous record.
//
//   k_fa_count  per tile: position of its last newline (global), bytes before its first newline, and
//               the kept bytes after the first newline (those do not depend on other tiles)
//   k_fa_scan   one workgroup: last newline before every tile (max-scan) -> its state, output offsets
//   k_fa_emit   classify again, compact through LDS, write
#define FA_NONE 0xffffffffffffffffull        // "no newline so far" in the max-scan (wraps to -1 + 1 = 0)

// inclusive max-scan over the 256 threads of a block, returns the exclusive value (FA_NONE-aware:
// positions are stored +1, 0 = none)
__device__ __forceinline__ u64 fa_block_exmax(u64 v, u64 *tmp, u64 *total)
{ const u32 lane = fk_lane();
  const u32 wave = threadIdx.x >> 6;
  u64 x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1)
    { const u64 y = __shfl_up(x, o, 64);
      if ((int) lane >= o) x = max(x, y);
    }
  if (lane == 63) tmp[wave] = x;
  __syncthreads();
  u64 base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 4; w++)
    { const u64 t = tmp[w];
      if ((u32) w < wave) base = max(base, t);
      tot = max(tot, t);
    }
  __syncthreads();
  *total = tot;
  u64 ex = __shfl_up(x, 1, 64);
  if (lane == 0) ex = 0;
  return max(base, ex);
}

// walk over a thread's bytes.  fresh: the first byte starts a line; header: the line the first byte
// belongs to is a header (ignored when fresh)
// skip: bytes 0..skip are passed over as if they were newlines (the head of a tile, k_fa_count)
template <bool EMIT>
__device__ __forceinline__ void fa_walk(const uint4 (&v)[4], int nvalid, bool fresh, bool header, u32 &kept,
                                        u32 &recs, unsigned char *stage, int skip = -1)
{ const u32 *w = (const u32 *) v;
#pragma unroll
  for (int i = 0; i < FQ_PER; i++)
    { if (i < nvalid && i > skip)
        { const u32 ch = (w[i >> 2] >> (8 * (i & 3))) & 0xffu;
          if (fresh)
            { header = (ch == '>');
              if (header)
                { if (EMIT) stage[kept] = 0;
                  kept += 1;
                  recs += 1;
                }
            }
          fresh = (ch == '\n');
          if (!header && ch != '\n')
            { if (EMIT) stage[kept] = (unsigned char) ch;
              kept += 1;
            }
        }
    }
}

// last newline of the thread's bytes as position + 1 relative to `origin` (0 = none)
__device__ __forceinline__ u64 fa_last_nl(const uint4 (&v)[4], int nvalid, int64_t origin)
{ const unsigned char *b = (const unsigned char *) v;
  u64 last = 0;
#pragma unroll
  for (int i = 0; i < FQ_PER; i++)
    if (i < nvalid && b[i] == '\n')
      last = (u64) (origin + i) + 1;
  return last;
}

// tile_info[t*4 + 0] = global position + 1 of the tile's last newline (0 = none), [1] = bytes before
// its first newline, [2] = kept bytes after the first newline, [3] = records started after it
__global__ __launch_bounds__(FQ_THREADS) void k_fa_count(const unsigned char *__restrict__ raw, int64_t n,
                                                         u64 *__restrict__ tile_info)
{ __shared__ u64 tmp[8];
  __shared__ u32 red[2];
  __shared__ u64 s_first;
  const int64_t tbase = (int64_t) blockIdx.x * FQ_TILE;
  const int64_t base  = tbase + (int64_t) threadIdx.x * FQ_PER;
  uint4 v[4];
  int   nvalid;
  fq_load(raw, n, base, v, nvalid);
  const u64 mylast = fa_last_nl(v, nvalid, base);
  u64 tot;
  const u64 before = fa_block_exmax(mylast, tmp, &tot);       // last newline in front of this thread
  if (threadIdx.x == 0) { red[0] = 0; red[1] = 0; s_first = 0; }
  __syncthreads();
  // first newline of the tile: the thread that has one while nothing precedes it
  if (mylast != 0 && before == 0)
    { const unsigned char *b = (const unsigned char *) v;
      int f = 0;
      while (b[f] != '\n') f++;
      s_first = (u64) (base + f) + 1;
    }
  __syncthreads();
  const u64 first = s_first;
  u32 kept = 0, recs = 0;
  if (before != 0)                                             // everything here lies after a newline
    { const int64_t L = (int64_t) before - 1;
      const bool fresh = (L + 1 == base);
      fa_walk<false>(v, nvalid, fresh, !fresh && raw[L + 1] == '>', kept, recs, NULL);
    }
  else if (mylast != 0)                                        // holds the tile's first newline:
    { const int f = (int) ((int64_t) first - 1 - base);        //   what follows it starts a line
      fa_walk<false>(v, nvalid, true, false, kept, recs, NULL, f);
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
    { kept += __shfl_down(kept, o, 64);
      recs += __shfl_down(recs, o, 64);
    }
  if (fk_lane() == 0)
    { if (kept) atomicAdd(&red[0], kept);
      if (recs) atomicAdd(&red[1], recs);
    }
  __syncthreads();
  if (threadIdx.x == 0)
    { const int64_t tend = (n - tbase < FQ_TILE) ? n : tbase + FQ_TILE;
      tile_info[(int64_t) blockIdx.x * 4 + 0] = tot;
      tile_info[(int64_t) blockIdx.x * 4 + 1] = (first != 0) ? (u64) ((int64_t) first - 1 - tbase) : (u64) (tend - tbase);
      tile_info[(int64_t) blockIdx.x * 4 + 2] = red[0];
      tile_info[(int64_t) blockIdx.x * 4 + 3] = red[1];
    }
}

// tile_state[t] = position + 1 of the last newline before tile t (0 = none in this piece);
// tile_off[t] = kept bytes before tile t.  carry: bit 0 = the line open at the start of the piece is
// a header, bit 1 = the piece starts at a line start.  out[0] = kept bytes, out[1] = records
__global__ __launch_bounds__(256) void k_fa_scan(const unsigned char *__restrict__ raw,
                                                 const u64 *__restrict__ tile_info, int64_t ntiles, int carry,
                                                 u64 *__restrict__ tile_state, u64 *__restrict__ tile_off,
                                                 u64 *__restrict__ out)
{ __shared__ u64 tmp[8];
  u64 carry_last = 0, carry_off = 0, recs = 0;
  for (int64_t b = 0; b < ntiles; b += 256)
    { const int64_t t = b + threadIdx.x;
      const u64 mylast = (t < ntiles) ? tile_info[t * 4 + 0] : 0ull;
      u64 totlast;
      const u64 ex = max(carry_last, fa_block_exmax(mylast, tmp, &totlast));
      u64 mykeep = 0;
      if (t < ntiles)
        { const int64_t tstart = t * FQ_TILE;
          const u64 head = tile_info[t * 4 + 1];
          bool fresh, header;
          if (ex != 0)
            { fresh = ((int64_t) ex == tstart);            // (ex - 1) + 1 == tstart
              header = !fresh && raw[ex] == '>';
            }
          else
            { fresh = (carry & 2) && tstart == 0;
              header = !fresh && (((carry & 2) ? raw[0] == '>' : (carry & 1)) != 0);
            }
          u64 hk;
          if (fresh)
            hk = (head > 0 && raw[tstart] == '>') ? 1 : head;
          else
            hk = header ? 0 : head;
          if (fresh && head > 0 && raw[tstart] == '>')
            recs += 1;
          mykeep = hk + tile_info[t * 4 + 2];
          recs += tile_info[t * 4 + 3];
          tile_state[t] = ex;
        }
      u64 totk;
      const u64 exk = fk_block_exscan_256<u64>(mykeep, tmp, &totk);
      if (t < ntiles)
        tile_off[t] = carry_off + exk;
      carry_off += totk;
      carry_last = max(carry_last, totlast);
    }
  // records: sum over threads
  __shared__ u64 s_recs;
  if (threadIdx.x == 0) s_recs = 0;
  __syncthreads();
  if (recs) atomicAdd(&s_recs, recs);
  __syncthreads();
  if (threadIdx.x == 0)
    { out[0] = carry_off;
      out[1] = s_recs;
    }
}

__global__ __launch_bounds__(FQ_THREADS) void k_fa_emit(const unsigned char *__restrict__ raw, int64_t n,
                                                        const u64 *__restrict__ tile_state,
                                                        const u64 *__restrict__ tile_off, int carry,
                                                        unsigned char *__restrict__ dst)
{ __shared__ unsigned char stage[FQ_TILE + 256];
  __shared__ u64 tmp[8];
  __shared__ u32 tmp32[8];
  const int64_t tbase = (int64_t) blockIdx.x * FQ_TILE;
  const int64_t base  = tbase + (int64_t) threadIdx.x * FQ_PER;
  uint4 v[4];
  int   nvalid;
  fq_load(raw, n, base, v, nvalid);
  const u64 mylast = fa_last_nl(v, nvalid, base);
  u64 tot;
  const u64 before = max(tile_state[blockIdx.x], fa_block_exmax(mylast, tmp, &tot));
  bool fresh, header;
  if (before != 0)
    { fresh = ((int64_t) before == base);
      header = !fresh && raw[before] == '>';
    }
  else
    { fresh = (carry & 2) && base == 0;
      header = !fresh && (((carry & 2) ? raw[0] == '>' : (carry & 1)) != 0);
    }
  u32 kept = 0, recs = 0;
  fa_walk<false>(v, nvalid, fresh, header, kept, recs, NULL);
  u32 tkept;
  const u32 ex = fk_block_exscan_256<u32>(kept, tmp32, &tkept);
  kept = 0; recs = 0;
  fa_walk<true>(v, nvalid, fresh, header, kept, recs, stage + ex);
  __syncthreads();
  unsigned char *o = dst + tile_off[blockIdx.x];
  for (u32 i = threadIdx.x; i < tkept; i += FQ_THREADS)
    o[i] = stage[i];
}

/* d_raw: nbytes of FASTA text on the device.  *state: bit 0 = the open line is a header, bit 1 = at a
   line start (2 before the first byte of a file); updated by the caller from the host copy of the text.
   Every '>' at a line start writes a 0 (the end of the previous record), sequence bytes follow. */
int fkx_parse_fasta(fk_ctx *ctx, const void *d_raw, int64_t nbytes, int state, void *d_dst,
                    int64_t *nkept, int64_t *nrecs)
{ hipStream_t s = ctx->stream;
  *nkept = 0; *nrecs = 0;
  if (nbytes <= 0)
    return (FK_OK);
  const int64_t ntiles = (nbytes + FQ_TILE - 1) / FQ_TILE;
  u64 *d_info  = (u64 *) fk_slot(ctx, FK_SLOT_FQ_INFO, ntiles * 4 * 8);
  u64 *d_state = (u64 *) fk_slot(ctx, FK_SLOT_FQ_PHASE, ntiles * 8);
  u64 *d_off   = (u64 *) fk_slot(ctx, FK_SLOT_FQ_OFF, ntiles * 8);
  if (d_info == NULL || d_state == NULL || d_off == NULL)
    return (FK_ENOMEM);
  u64 *d_out = ctx->d_scratch + 2048;
  hipLaunchKernelGGL(k_fa_count, dim3((unsigned) ntiles), dim3(FQ_THREADS), 0, s,
                     (const unsigned char *) d_raw, nbytes, d_info);
  hipLaunchKernelGGL(k_fa_scan, dim3(1), dim3(256), 0, s, (const unsigned char *) d_raw, (const u64 *) d_info,
                     ntiles, state, d_state, d_off, d_out);
  hipLaunchKernelGGL(k_fa_emit, dim3((unsigned) ntiles), dim3(FQ_THREADS), 0, s,
                     (const unsigned char *) d_raw, nbytes, (const u64 *) d_state, (const u64 *) d_off, state,
                     (unsigned char *) d_dst);
  FK_LAUNCH_CHECK(ctx);
  FK_HIP(ctx, hipMemcpyAsync(ctx->h_scratch, d_out, 2 * sizeof(u64), hipMemcpyDeviceToHost, s));
  FK_HIP(ctx, hipStreamSynchronize(s));
  *nkept = (int64_t) ctx->h_scratch[0];
  *nrecs = (int64_t) ctx->h_scratch[1];
  return (FK_OK);
}


// ---------------------------------------------------------------------------------------------
// Reads in two bits per base -> 0-terminated ASCII reads, for the consumers that walk reads byte by byte (exact_parts,
// profiles) after the reads came through fk_push_packed -- the counting path splits the packed form directly
// (fk_split.hip).  One pushed block at a time: codes = the packed read buffer, the block's bases are the positions
// pos0 .. pos0 + nbases - 1 (four to a byte, first base in the two high bits; a c g t = 0 1 2 3, the .ktab encoding,
// README.md:977-984; pos0 a multiple of 16); roff[r] = first position of the block's read r, roff[nreads] >= pos0 +
// nbases (it includes the padding behind the block); read r's bases land at dst[(roff[r] - pos0) + r ...], its
// terminator behind them.
#define UP_THREADS 256
#define UP_BASES   16                      // bases per thread: one dword of codes

__device__ __forceinline__ int64_t up_read_of(const int64_t *__restrict__ roff, int64_t lo, int64_t hi, int64_t pos)
{ // largest r in [lo, hi] with roff[r] <= pos
  while (lo < hi)
    { const int64_t mid = (lo + hi + 1) >> 1;
      if (roff[mid] <= pos) lo = mid; else hi = mid - 1;
    }
  return (lo);
}

__global__ __launch_bounds__(UP_THREADS) void k_up_bases(const unsigned char *__restrict__ codes, int64_t pos0, int64_t nbases,
                                                         const int64_t *__restrict__ roff, int64_t nreads,
                                                         unsigned char *__restrict__ dst)
{ __shared__ int64_t sh_r[2];
  const int64_t end = pos0 + nbases;
  const int64_t b0 = pos0 + (int64_t) blockIdx.x * (UP_THREADS * UP_BASES);
  if (threadIdx.x < 2)
    { const int64_t pos = (threadIdx.x == 0) ? b0 : min(b0 + UP_THREADS * UP_BASES, end) - 1;
      sh_r[threadIdx.x] = up_read_of(roff, 0, nreads - 1, pos);
    }
  __syncthreads();
  const int64_t pos = b0 + (int64_t) threadIdx.x * UP_BASES;
  if (pos >= end)
    return;
  const u32 c4 = *(const u32 *) (codes + (pos >> 2));            // the buffer is padded to a dword multiple
  int64_t r = up_read_of(roff, sh_r[0], sh_r[1], pos);
  const int64_t last = min(pos + UP_BASES, end);
  if (roff[r + 1] >= last)
    { // the whole group lies in one read: four dwords, wherever the read's shift puts them
      unsigned char *o = dst + (pos - pos0) + r;
      const int n = (int) (last - pos);
      u32 w[4];
#pragma unroll
      for (int q = 0; q < 4; q++)
        { const u32 byte = (c4 >> (8 * q)) & 0xffu;
          u32 x = 0;
#pragma unroll
          for (int i = 0; i < 4; i++)
            { const u32 code = (byte >> (6 - 2 * i)) & 3u;
              x |= ((0x74676361u >> (8 * code)) & 0xffu) << (8 * i);      // "acgt"
            }
          w[q] = x;
        }
      if (n == UP_BASES)
        {
#pragma unroll
          for (int q = 0; q < 4; q++)
            __builtin_memcpy(o + 4 * q, &w[q], 4);
        }
      else
        for (int i = 0; i < n; i++)
          o[i] = (unsigned char) (w[i >> 2] >> (8 * (i & 3)));
      return;
    }
  for (int64_t i = pos; i < last; i++)
    { while (roff[r + 1] <= i)
        r += 1;
      const u32 byte = (c4 >> (8 * ((i - pos) >> 2))) & 0xffu;
      const u32 code = (byte >> (6 - 2 * ((i - pos) & 3))) & 3u;
      dst[(i - pos0) + r] = (unsigned char) ((0x74676361u >> (8 * code)) & 0xffu);
    }
}

__global__ __launch_bounds__(256) void k_up_ends(const int64_t *__restrict__ roff, int64_t nreads, int64_t pos0, int64_t nbases,
                                                 unsigned char *__restrict__ dst)
{ const int64_t r = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (r < nreads)
    dst[min(roff[r + 1] - pos0, nbases) + r] = 0;               // (the last read's entry lies behind the block's padding)
}

// inv: pairs (first position, length) of stretches that hold no acgt: 'n' there (one workgroup per stretch); the pairs
// outside [pos0, pos0 + nbases) -- other blocks', the padding -- do nothing
__global__ __launch_bounds__(256) void k_up_invalid(const int64_t *__restrict__ inv, const int64_t *__restrict__ roff,
                                                    int64_t nreads, int64_t pos0, int64_t nbases, unsigned char *__restrict__ dst)
{ const int64_t s = inv[2 * (int64_t) blockIdx.x], n = inv[2 * (int64_t) blockIdx.x + 1];
  for (int64_t i = s + threadIdx.x; i < s + n && i < pos0 + nbases; i += 256)
    if (i >= pos0)
      dst[(i - pos0) + up_read_of(roff, 0, nreads - 1, i)] = 'n';
}

/* One pushed block of the packed read buffer d_codes -> nbases + nreads bytes of ASCII at d_dst, on stream s; d_roff
   [nreads + 1] and d_inv [2 ninv] on the device (d_inv may hold any superset of the block's stretches). */
int fkx_unpack_reads(fk_ctx *ctx, hipStream_t s, const void *d_codes, int64_t pos0, int64_t nbases, const int64_t *d_roff,
                     int64_t nreads, const int64_t *d_inv, int64_t ninv, void *d_dst)
{ if (nreads <= 0)
    return (FK_OK);
  if (nbases > 0)
    { const int64_t nb = (nbases + UP_THREADS * UP_BASES - 1) / (UP_THREADS * UP_BASES);
      if (nb > 0x7fffffffll)
        { fk_set_error(ctx, "fk_push_packed: block too large");
          return (FK_EINVAL);
        }
      hipLaunchKernelGGL(k_up_bases, dim3((unsigned) nb), dim3(UP_THREADS), 0, s, (const unsigned char *) d_codes, pos0, nbases,
                         d_roff, nreads, (unsigned char *) d_dst);
    }
  hipLaunchKernelGGL(k_up_ends, dim3((unsigned) ((nreads + 255) / 256)), dim3(256), 0, s, d_roff, nreads, pos0, nbases,
                     (unsigned char *) d_dst);
  if (ninv > 0)
    hipLaunchKernelGGL(k_up_invalid, dim3((unsigned) ninv), dim3(256), 0, s, d_inv, d_roff, nreads, pos0, nbases,
                       (unsigned char *) d_dst);
  FK_LAUNCH_CHECK(ctx);
  return (FK_OK);
}


// The inverse for reads of one length (measurement helper beside fk_synth_reads): rows of read_len + 1 bytes -> codes.
__global__ __launch_bounds__(256) void k_pack_fixed(const unsigned char *__restrict__ src, int64_t nbases, u32 read_len,
                                                    unsigned char *__restrict__ codes, int64_t block0)
{ const int64_t o = (block0 + (int64_t) blockIdx.x) * 256 + threadIdx.x;          // output byte
  if (o * 4 >= nbases)
    return;
  u32 b = 0;
#pragma unroll
  for (int i = 0; i < 4; i++)
    { const int64_t pos = o * 4 + i;
      u32 code = 0;
      if (pos < nbases)
        { const unsigned char c = src[pos + pos / read_len] | 0x20;
          code = (c == 'c') ? 1u : (c == 'g') ? 2u : (c == 't') ? 3u : 0u;
        }
      b |= code << (6 - 2 * i);
    }
  codes[o] = (unsigned char) b;
}

int fkx_pack_fixed(fk_ctx *ctx, const void *d_bases, int64_t nreads, u32 read_len, void *d_codes)
{ const int64_t nbases = nreads * (int64_t) read_len;
  if (nbases <= 0)
    return (FK_OK);
  const int64_t nb = ((nbases + 3) / 4 + 255) / 256;
  for (int64_t b0 = 0; b0 < nb; b0 += (1 << 23))        // a launch holds fewer than 2^32 work-items (a larger grid wraps silently)
    hipLaunchKernelGGL(k_pack_fixed, dim3((unsigned) std::min<int64_t>(nb - b0, 1 << 23)), dim3(256), 0, ctx->stream,
                       (const unsigned char *) d_bases, nbases, read_len, (unsigned char *) d_codes, b0);
  FK_LAUNCH_CHECK(ctx);
  return (FK_OK);
}
