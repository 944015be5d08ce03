/* Test harness (not shipped): the exact splitter's kernels (fastk_amd/csrc/fk_split_exact.hip -- the replay of the
   reference's Distribute_Block, split.c:1016-1393) compiled from their .hip source for the CPU (tests/csrc/hip_emu.h)
   and driven the way fkx_split_exact drives them for one bucket: [segments:] k_xs_blocks, scan, k_xs_blkread,
   k_xs_find, scan, k_xs_segs, k_xs_ends; k_split_exact<count>, scan, k_split_exact<emit>, k_xs_pack.  The scans are
   the device's own (k_xs_tilesum / k_exscan_tiles / k_xs_tilescan above 32,768 counts).  tests/test_xs_emu.py compares
   the records with the oracle's.
     g++ -std=c++17 -O1 -pthread -DFK_HOST_EMU -shared -fPIC -I fastk_amd/csrc -I tests/csrc -o xs_emu.so xs_emu.cpp */
#define FK_EMU_DEFINE 1
#include "../../fastk_amd/csrc/fk_split_exact.hip"

static u64 g_ovf;

static void scan(const u32 *in, int64_t n, u64 *out, u64 *total)       /* xs_exscan, fk_split_exact.hip */
{ if (n <= 32768)
    { emu_launch(1, 256, 0, [&] { k_exscan_tiles(in, n, out, total); });
      return;
    }
  const int64_t nt = (n + 4095) / 4096;
  std::vector<u32> tsum((size_t) nt + 16);
  std::vector<u64> toff((size_t) nt + 16);
  emu_launch((unsigned) nt, 256, 0, [&] { k_xs_tilesum(in, n, tsum.data(), &g_ovf); });
  emu_launch(1, 256, 0, [&] { k_exscan_tiles(tsum.data(), nt, toff.data(), total); });
  emu_launch((unsigned) nt, 256, 0, [&] { k_xs_tilescan(in, n, toff.data(), out); });
}

extern "C" {

/* the device's scan of n counts; returns the overflow word */
int64_t emu_xs_scan(const u32 *in, int64_t n, u64 *out, u64 *total)
{ g_ovf = 0;
  scan(in, n, out, total);
  return ((int64_t) g_ovf);
}

/* bases: 0-terminated reads, 16-byte aligned, readable 16 bytes to either side; roff[nreads + 1].  out: room for cap
   records of sww dwords.  Returns the number of records (-1: more than cap); *ninst = valid k-mer instances,
   *nseg_out = threads of the split kernels. */
int64_t emu_split_exact(const unsigned char *bases, const int64_t *roff, int64_t nreads, int kmer, const int *tran,
                        int smer_bytes, int sww, int segments, int dq_cap, int noflip, u32 *out, int64_t cap,
                        int64_t *ninst, int64_t *nseg_out)
{ ExactArgs a;
  memset(&a, 0, sizeof(a));
  a.bases = bases; a.roff = roff; a.nreads = nreads; a.kmer = kmer; a.bc_prefix = 0;
  for (int i = 0; i < 4; i++) a.tran[i] = tran[i];
  a.smer_bytes = smer_bytes; a.sww = sww;
  a.nparts = 1; a.trie = NULL; a.pad_len = 5; a.pad2 = 0;
  a.defer = (sww >= 2) ? 1 : 0; a.noflip = noflip; a.dq_cap = dq_cap;
  int64_t nseg = nreads;
  a.nseg = nreads;
  std::vector<u32> nblk, blk_read, blk_j, blk_p0, flag, seg_read, seg_p0, seg_p1;
  std::vector<u64> boff, soff;
  g_ovf = 0;
  if (segments && kmer <= 64)
    { nblk.resize((size_t) nreads + 16); boff.resize((size_t) nreads + 16);
      const unsigned gr = (unsigned) ((nreads + 255) / 256);
      u64 tot = 0;
      emu_launch(gr, 256, 0, [&] { k_xs_blocks(roff, nreads, 0, kmer, nblk.data()); });
      scan(nblk.data(), nreads, boff.data(), &tot);
      const int64_t nblocks = (int64_t) tot;
      if (nblocks > nreads)
        { blk_read.resize((size_t) nblocks + 16); blk_j.resize((size_t) nblocks + 16); blk_p0.resize((size_t) nblocks + 16);
          flag.resize((size_t) nblocks + 16); soff.resize((size_t) nblocks + 16);
          emu_launch(gr, 256, 0, [&] { k_xs_blkread(nblk.data(), boff.data(), nreads, blk_read.data(), blk_j.data()); });
          emu_launch((unsigned) ((nblocks + 127) / 128), 128, 0,
                     [&] { k_xs_find(a, blk_read.data(), blk_j.data(), nblocks, blk_p0.data(), flag.data()); });
          u64 ns2 = 0;
          scan(flag.data(), nblocks, soff.data(), &ns2);
          nseg = (int64_t) ns2;
          seg_read.resize((size_t) nseg + 16); seg_p0.resize((size_t) nseg + 16); seg_p1.resize((size_t) nseg + 16);
          const unsigned gb = (unsigned) ((nblocks + 255) / 256);
          emu_launch(gb, 256, 0, [&] { k_xs_segs(blk_read.data(), blk_p0.data(), flag.data(), soff.data(), nblocks,
                                                 seg_read.data(), seg_p0.data()); });
          emu_launch((unsigned) ((nseg + 255) / 256), 256, 0, [&] { k_xs_ends(seg_read.data(), seg_p0.data(), nseg, seg_p1.data()); });
          a.nseg = nseg; a.seg_read = seg_read.data(); a.seg_p0 = seg_p0.data(); a.seg_p1 = seg_p1.data();
        }
    }
  *nseg_out = nseg;
  std::vector<u32> cnt((size_t) nseg + 16);
  std::vector<u64> off((size_t) nseg + 16);
  u64 inst[72];
  memset(inst, 0, sizeof(inst));
  a.cnt = cnt.data(); a.off = off.data(); a.out = NULL; a.inst = inst;
  const unsigned grid = (unsigned) ((nseg + XS_THREADS - 1) / XS_THREADS);
  emu_launch(grid, XS_THREADS, 0, [&] { k_split_exact<false, true>(a); });
  scan(cnt.data(), nseg, off.data(), &inst[64]);
  const int64_t ns = (int64_t) inst[64];
  int64_t ni = 0;
  for (int x = 0; x < 64; x++) ni += (int64_t) inst[x];
  *ninst = ni;
  if (ns > cap)
    return (-1);
  a.out = out;
  memset(inst, 0, sizeof(inst));
  if (ns > 0)
    { emu_launch(grid, XS_THREADS, 0, [&] { k_split_exact<true, true>(a); });
      if (a.defer)
        emu_launch((unsigned) ((ns + 255) / 256), 256, 0, [&] { k_xs_pack(a, ns); });
    }
  return (g_ovf ? -2 : ns);
}

}
