/* Test harness (not shipped): the radix engine, the LDS de-duplication and the LDS aggregation -- kernels AND the host code
   that drives them (fkx_lsd_sort, fkx_group, fkx_dedup_supermers, fkx_aggregate, fkx_aggregate_fills) -- compiled from
   their .hip sources for the CPU: tests/csrc/hip_emu.h stands in for the kernel language and, with -DFK_EMU_FULL, for the
   HIP runtime (device memory is host memory, a launch runs the kernel to its end); tests/csrc/emu_ctx.h for the context.
     g++ -std=c++17 -O1 -DFK_HOST_EMU -DFK_EMU_FULL -shared -fPIC -I fastk_amd/csrc -I tests/csrc -o full_emu.so full_emu.cpp */
#define FK_EMU_DEFINE 1
#include "../../fastk_amd/csrc/fk_radix.hip"
#include "../../fastk_amd/csrc/fk_aggr.hip"
#include "../../fastk_amd/csrc/fk_dedup.hip"
#include "emu_ctx.h"

extern "C" {

int emu_lsd_sort(fk_ctx *ctx, int64_t n, void *a, void *b, int rsize, const int *bytes, int nbytes, void **result)
{ return (fkx_lsd_sort(ctx, n, a, b, rsize, bytes, nbytes, result)); }
int emu_aggregate(fk_ctx *ctx, const void *grouped, int64_t n, int cutoff, int64_t *hist, int64_t *max_inst, int64_t *ndistinct,
                  void *table, int64_t cap, int64_t *ntable, const u64 *bounds, int64_t nfills)
{ return (bounds != NULL ? fkx_aggregate_fills(ctx, grouped, n, cutoff, hist, max_inst, ndistinct, table, cap, ntable, bounds, nfills)
                         : fkx_aggregate(ctx, grouped, n, cutoff, hist, max_inst, ndistinct, table, cap, ntable)); }
int emu_dedup(fk_ctx *ctx, const void *grouped, int64_t n, void *out, int64_t cap, int64_t *nout)
{ return (fkx_dedup_supermers(ctx, grouped, n, out, cap, nout)); }
int emu_group(fk_ctx *ctx, int64_t n, void *a, void *b, int rsize, int key_bytes, int npasses, void **result)
{ return (fkx_group(ctx, n, a, b, rsize, key_bytes, npasses, result)); }

}
