/* fk_synth.h -- deterministic synthetic read generator shared by host C and gfx950 device code.
 *
 * Every base of every read is a pure function of (seed, read index, base index), so the same
 * bytes are produced by the host (fixtures, CPU baseline sample) and by the device kernel
 * that fills HBM for bench.py -- no file or host->device copy of reads is needed at the
 * BASELINE.json config sizes.  The model follows SURVEY.md section 8(d): uniform ACGT genome,
 * reads sampled uniformly, random strand, iid substitution errors.
 *
 * This header is product code (the oracle includes it, never the other way round).
 */
#ifndef FK_SYNTH_H
#define FK_SYNTH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define FK_HD __host__ __device__ static inline
#else
#define FK_HD static inline
#endif

typedef struct
  { uint64_t seed;        /* PRNG seed                                             */
    uint64_t genome_len;  /* G: genome length in bases                             */
    uint32_t read_len;    /* L: every read has exactly L bases                     */
    uint32_t err_ppm;     /* substitution error rate in parts per million per base */
  } fk_synth_spec;

FK_HD uint64_t fk_mix64(uint64_t x)
{ x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return (x ^ (x >> 31));
}

/* 2-bit code (a,c,g,t = 0..3) of genome base i */
FK_HD uint32_t fk_synth_genome(uint64_t seed, uint64_t i)
{ return ((uint32_t) (fk_mix64(seed * 0x2545F4914F6CDD1Dull + i) >> 17) & 3u); }

/* start position and strand of read r */
FK_HD void fk_synth_place(const fk_synth_spec *sp, uint64_t r, uint64_t *start, uint32_t *strand)
{ uint64_t h = fk_mix64((sp->seed ^ 0xA5A5A5A55A5A5A5Aull) + 2*r);
  *start  = h % (sp->genome_len - sp->read_len + 1);
  *strand = (uint32_t) (fk_mix64((sp->seed ^ 0x3C3C3C3CC3C3C3C3ull) + 2*r + 1) >> 33) & 1u;
}

/* 2-bit code of base j of read r (read coordinates), given its placement */
FK_HD uint32_t fk_synth_base(const fk_synth_spec *sp, uint64_t r, uint32_t j,
                             uint64_t start, uint32_t strand)
{ uint32_t b;
  uint64_t e;

  if (strand == 0)
    b = fk_synth_genome(sp->seed,start+j);
  else
    b = 3u - fk_synth_genome(sp->seed,start+(sp->read_len-1-j));
  e = fk_mix64((sp->seed ^ 0x0F0F0F0FF0F0F0F0ull) + r*sp->read_len + j);
  if ((uint32_t) (e % 1000000u) < sp->err_ppm)
    b = (b + 1u + (uint32_t) ((e >> 40) % 3u)) & 3u;
  return (b);
}

#endif /* FK_SYNTH_H */
