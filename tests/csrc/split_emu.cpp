/* Test harness (not shipped): the default splitter's kernel (fastk_amd/csrc/fk_split.hip: k_split, the position-parallel
   replacement of Distribute_Block + Stuff_Seq, split.c:1016-1393,864-989) compiled from its .hip source for the CPU
   (tests/csrc/hip_emu.h) and driven the way fkx_split drives it: a counting pass, the bucket regions from its counts, an
   emit pass with one cursor per bucket.  tests/test_split_emu.py checks that the super-mers hold every canonical k-mer
   of the reads exactly as often as it occurs.
     g++ -std=c++17 -O1 -pthread -DFK_HOST_EMU -shared -fPIC -I fastk_amd/csrc -I tests/csrc -o split_emu.so split_emu.cpp */
#define FK_EMU_DEFINE 1
#include "../../fastk_amd/csrc/fk_split.hip"

extern "C" {

/* bases: nbytes of 0-terminated reads (any byte that is not acgtACGT separates), readable 64 bytes beyond.  mbucket:
   FK_NRANKS bucket numbers.  out: cap records of sww dwords, grouped by bucket; counts[nb] records per bucket.
   Returns the records (-1: cap too small, -2: the kernel reported an overflow); *ninst = valid k-mer instances. */
int64_t emu_split(const unsigned char *bases, int64_t nbytes, int kmer, int smer_bytes, int sww, int nb,
                  const uint8_t *mbucket, u32 *out, int64_t cap, int64_t *counts, int64_t *ninst)
{ static u64 scratch[2048];
  memset(scratch, 0, sizeof(scratch));
  u64 *d_counts = scratch, *d_cursor = scratch + 512, *d_base = scratch + 768;
  u32 *d_ovf = (u32 *) (scratch + 1024);
  SplitArgs a = SplitArgs();
  a.bases = bases; a.nbytes = nbytes; a.kmer = kmer; a.smer_bytes = smer_bytes; a.sww = sww; a.nbuckets = nb;
  a.mbucket = mbucket; a.counts = d_counts; a.cursor = d_cursor; a.cstride = 1; a.out = out; a.cap = cap;
  a.overflowed = d_ovf; a.tile_stride = 1; a.limit = NULL; a.pos = NULL; a.skipb = 0x100u; a.ent = NULL;
  const int64_t nstarts = nbytes - kmer + 1;
  if (nstarts <= 0) { *ninst = 0; return (0); }
  const int64_t ntiles = (nstarts + SP_TILE - 1) / SP_TILE;
  a.tile0 = 0;
  emu_launch((unsigned) ntiles, SP_THREADS, (size_t) nb * 16, [&] { k_split<false, false, false>(a); });
  int64_t tot = 0;
  for (int b = 0; b < nb; b++)
    { d_base[b] = (u64) tot; counts[b] = (int64_t) d_counts[b]; tot += counts[b]; }
  int64_t t = 0;
  for (int x = 0; x < 64; x++) t += (int64_t) d_counts[256 + x];
  *ninst = t;
  if (tot > cap) return (-1);
  a.rbase = d_base; a.lstreams = 0;
  emu_launch((unsigned) ntiles, SP_THREADS, (size_t) nb * 16, [&] { k_split<true, false, false>(a); });
  return (*d_ovf ? -2 : tot);
}

}
