/* Test harness (not shipped): what the host halves of the .hip files need from a context when they run on the CPU under
   tests/csrc/hip_emu.h (-DFK_HOST_EMU -DFK_EMU_FULL): an fk_ctx with its scratch words, the arena (fk_slot), the error
   text, the event pool and the pageable copies -- the functions fk_api.hip provides in the product. */
#pragma once
#include <cstdarg>

void *fk_slot(fk_ctx *ctx, int slot, int64_t nbytes)
{ if (nbytes < 16) nbytes = 16;
  if (ctx->slot_cap[slot] < nbytes)
    { free(ctx->slot_ptr[slot]);
      ctx->slot_ptr[slot] = aligned_alloc(256, (size_t) ((nbytes + 255 + 256) & ~255ll));
      ctx->slot_cap[slot] = nbytes;
      memset(ctx->slot_ptr[slot], 0xA5, (size_t) nbytes);          /* (fresh device memory holds anything) */
    }
  return (ctx->slot_ptr[slot]);
}

void fk_set_error(fk_ctx *ctx, const char *fmt, ...)
{ va_list ap;
  va_start(ap, fmt);
  if (ctx != NULL) vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
  va_end(ap);
}

int  fkx_event_get(int, bool, hipEvent_t *e) { *e = malloc(8); return (FK_OK); }
void fkx_event_put(int, bool, hipEvent_t *e) { if (e != NULL && *e != NULL) { free(*e); *e = NULL; } }
int  fkx_d2h_pageable(fk_ctx *, hipStream_t, void *dst, const void *src, size_t n) { memcpy(dst, src, n); return (FK_OK); }
int  fkx_h2d_pageable(fk_ctx *, hipStream_t, void *dst, const void *src, size_t n) { memcpy(dst, src, n); return (FK_OK); }

extern "C" {

/* widths as fk_get_widths gives them (the caller passes what the oracle's parameters say) */
fk_ctx *emu_ctx_create(int kmer, int smer_bytes, int smer_stride, int kmer_bytes, int kmer_stride, int num_cus)
{ fk_ctx *ctx = (fk_ctx *) calloc(1, sizeof(fk_ctx));
  ctx->prm.kmer = kmer; ctx->prm.nbuckets = 1; ctx->prm.nthreads = 4;
  ctx->wid.kmer = kmer; ctx->wid.max_super = kmer - 4;
  ctx->wid.smer_bytes = smer_bytes; ctx->wid.slen_bytes = 1; ctx->wid.smer_word = smer_bytes + 1; ctx->wid.smer_stride = smer_stride;
  ctx->wid.kmer_bytes = kmer_bytes; ctx->wid.kmer_word = kmer_bytes + 2; ctx->wid.kmer_stride = kmer_stride;
  ctx->num_cus = num_cus;
  ctx->d_scratch = (u64 *) calloc(1, 65536);
  ctx->h_scratch = (u64 *) calloc(1, 65536 + 32 * 256 * 8);
  ctx->d_digit_hist = (u64 *) calloc(32 * 256, sizeof(u64));
  ctx->stream = malloc(8);
  ctx->ev0 = malloc(8); ctx->ev1 = malloc(8);
  return (ctx);
}

void emu_ctx_destroy(fk_ctx *ctx)
{ for (int i = 0; i < FK_NSLOTS; i++) free(ctx->slot_ptr[i]);
  free(ctx->d_scratch); free(ctx->h_scratch); free(ctx->d_digit_hist); free(ctx->stream); free(ctx->ev0); free(ctx->ev1);
  free(ctx);
}

const char *emu_ctx_error(fk_ctx *ctx) { return (ctx->err); }
void emu_ctx_debug(fk_ctx *ctx, const char *key, int value)
{ if (strcmp(key, "aggr_limit") == 0) ctx->dbg_aggr_limit = value;
  if (strcmp(key, "aggr_engine") == 0) ctx->dbg_aggr_engine = value;
  if (strcmp(key, "radix_engine") == 0) ctx->dbg_radix_engine = value;
}

}
