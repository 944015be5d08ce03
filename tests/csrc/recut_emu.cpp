/* Test harness (not shipped): the kernels of round 6's k-mer stage -- k_recut, k_ref_count, k_exscan_tiles,
   k_ex_expand<.., REF>, k_ref_bounds -- compiled FROM THEIR .hip SOURCES for the CPU (tests/csrc/hip_emu.h) and
   exported to tests/test_recut_emu.py.  Built by that test:
     g++ -std=c++17 -O1 -pthread -DFK_HOST_EMU -shared -fPIC -x c++ -I fastk_amd/csrc -I tests/csrc -o recut_emu.so recut_emu.cpp */
#define FK_EMU_DEFINE 1
#include "../../fastk_amd/csrc/fk_recut.hip"
#include "../../fastk_amd/csrc/fk_expand.hip"

template <int RW>
static int64_t run_recut(const u32 *dd, int64_t n, int kmer, int len_byte, u64 *out, u64 cap, u64 *scal)
{ const int64_t ntiles = (n + RC_THREADS - 1) / RC_THREADS;
  const size_t lds = (size_t) RcCfg<RW>::PMAX * RC_THREADS * sizeof(u32);
  scal[0] = scal[1] = scal[2] = scal[3] = 0;
  const unsigned grid = (unsigned) std::min<int64_t>(ntiles, 2);        // (persistent: two workgroups take the tiles in turn)
  emu_launch(grid, RC_THREADS, lds, [&] { k_recut<RW>(dd, n, kmer, len_byte, out, cap, scal, ntiles); });
  return ((int64_t) scal[0]);
}

template <int RW, int KN, int OW>
static void run_expand(const u32 *dd, const u64 *refs, int64_t nref, int kmer, int len_byte, const u64 *koff, u32 *out,
                       u64 *overflow, int kbytes)
{ const int64_t ntiles = (nref + EX_TILE - 1) / EX_TILE;
  const unsigned grid = (unsigned) std::min<int64_t>(ntiles, 2);
  emu_launch(grid, EX_THREADS, 0, [&]
    { k_ex_expand<RW, KN, OW, true, true>(dd, nref, kmer, len_byte, koff, out, overflow, ntiles, (uint8_t *) NULL, kbytes, refs); });
}

extern "C" {

/* dd: n records of RW + 1 dwords.  Returns the references made (-1: RW not built), *flags = the kernel's overflow word. */
int64_t emu_recut(int rw, const u32 *dd, int64_t n, int kmer, int len_byte, u64 *out, int64_t cap, int64_t *flags)
{ u64 scal[4];
  int64_t r = -1;
  switch (rw)
  { case 4: r = run_recut<4>(dd, n, kmer, len_byte, out, (u64) cap, scal); break;
    case 5: r = run_recut<5>(dd, n, kmer, len_byte, out, (u64) cap, scal); break;
    case 6: r = run_recut<6>(dd, n, kmer, len_byte, out, (u64) cap, scal); break;
    case 7: r = run_recut<7>(dd, n, kmer, len_byte, out, (u64) cap, scal); break;
    default: return (-1);
  }
  *flags = (int64_t) scal[1];
  return (r);
}

/* per-tile k-mers of the references and their exclusive offsets (koff: ntiles entries); returns the total */
int64_t emu_ref_offsets(const u64 *refs, int64_t nref, u64 *koff)
{ const int64_t ntiles = (nref + RF_TILE - 1) / RF_TILE;
  std::vector<u32> km((size_t) ntiles + 1);
  u64 total = 0;
  emu_launch((unsigned) ntiles, 256, 0, [&] { k_ref_count(refs, nref, km.data()); });
  emu_launch(1, 256, 0, [&] { k_exscan_tiles(km.data(), ntiles, koff, &total); });
  return ((int64_t) total);
}

/* W records (ow dwords each) of the sorted references; returns the overflow word, -1 when the widths are not built */
int64_t emu_expand_refs(int rw, int kn, int ow, const u32 *dd, const u64 *refs, int64_t nref, int kmer, int len_byte,
                        const u64 *koff, u32 *out, int kbytes)
{ u64 ovf = 0;
#define EMU_CASE(RW, KN, OW) if (rw == RW && kn == KN && ow == OW) { run_expand<RW, KN, OW>(dd, refs, nref, kmer, len_byte, koff, out, &ovf, kbytes); return ((int64_t) ovf); }
  EMU_CASE(4, 2, 3) EMU_CASE(5, 3, 3) EMU_CASE(6, 3, 4) EMU_CASE(7, 4, 4)      /* k = 32; 33..40; 41..48; 49..56 */
#undef EMU_CASE
  return (-1);
}

void emu_ref_bounds(const u64 *refs, int64_t nref, const u64 *koff, int64_t W, int target, u64 *bounds, int64_t nfills)
{ const int64_t ntiles = (nref + RF_TILE - 1) / RF_TILE;
  emu_launch((unsigned) ((nfills + 1 + 255) / 256), 256, 0,
             [&] { k_ref_bounds(refs, nref, koff, ntiles, (u64) W, (u32) target, nfills, bounds); });
}

}
