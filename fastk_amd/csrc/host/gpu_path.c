/* gpu_path.c -- the shim of INTEGRATION.md, compilable: it replaces split.c, count.c, table.c,
 * merge.c and MSDsort.c in a build of the REFERENCE FastK (FastK.c, io.c, LSDsort.c, libfastk.c stay
 * as they are), so that the reference's own main(), option parsing and input layer drive the GPU
 * path through the C-ABI.  Built only where the reference sources are present (target ref_gpu of
 * the test-infrastructure Makefile); tests/test_gpu_parity.py runs the result as the end-to-end
 * drop-in check.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "libfastk.h"
#include "FastK.h"
#include "fastk_amd.h"

uint8   Comp[256];        /* defined in count.c in the reference; unused on this path */
int64  *NUM_RID;          /* defined in split.c in the reference                        */

static fk_ctx   *GPU;
static fk_result RES;

static void die(const char *what)
{ fprintf(stderr,"%s: %s: %s\n",Prog_Name,what,fk_last_error(GPU));
  Clean_Exit(1);
}

/* replaces Determine_Scheme (split.c:491): creates the device context; returns MAX_SUPER for the
   width globals of FastK.c:454-468 */
int Determine_Scheme(DATA_BLOCK *block)
{ fk_params p;
  fk_widths w;

  (void) block;
  /* -c: io.c compresses the homopolymer runs before it hands the blocks over (io.c:557-560) */
  fk_default_params(&p);
  p.kmer = KMER; p.nthreads = NTHREADS; p.bc_prefix = BC_PREFIX;
  p.table_cutoff = DO_PROFILE ? 1 : DO_TABLE;      /* profiles look every k-mer up in the table */
  /* Fast by default: the position-parallel splitter; .hist, the .ktab stub and the concatenated part payloads are
     the reference's byte for byte, only the hidden part files are cut at other first bytes.  FASTK_AMD_EXACT=1 in
     the environment (or -DFASTK_AMD_EXACT at build time) replays the reference's own super-mer cuts instead: every
     file byte-identical, the split stage ~2x the default one's. */
#ifdef FASTK_AMD_EXACT
  p.exact_parts = 1;
#else
  { const char *e = getenv("FASTK_AMD_EXACT");
    p.exact_parts = (e != NULL && e[0] != '\0' && e[0] != '0');
  }
#endif
  if (p.exact_parts && DO_PROFILE)  /* with -p the reference keeps its super-mers on the read's strand (split.c:1245) */
    p.exact_parts = 2;
  if (fk_create(&p,&GPU) != FK_OK)
    die("fk_create");
  if (p.exact_parts)                /* the reference's buckets (NPARTS of FastK.c:417-429) under its own scheme */
    fk_set_sort_memory(GPU,SORT_MEMORY,block->ratio);
  if (fk_train_block(GPU,block->bases,block->boff,block->nreads) != FK_OK)   /* split.c:529-575 */
    die("fk_train_block");
  fk_get_widths(KMER,&w);
  NUM_RID = (int64 *) calloc(ITHREADS > 0 ? ITHREADS : 1,sizeof(int64));
  return (w.max_super);
}

/* replaces Split_Kmers (split.c:1407) */
void Split_Kmers(Input_Partition *io, char *root)
{ (void) root;
  Scan_All_Input(io);
}

/* replaces Distribute_Block (split.c:1016), called by the io.c threads */
void Distribute_Block(DATA_BLOCK *block, int tid)
{ if (fk_push_block(GPU,block->bases,block->boff,block->nreads,block->rem,tid) != FK_OK)
    die("fk_push_block");
}

/* replaces Split_Table (split.c:1943): the relative table needs no redistribution here, it becomes
   the dictionary of the look-ups in Sorting */
void Split_Table(char *root)
{ (void) root; }

/* replaces Sorting (count.c:1202), including its profile half (count.c:639-1181) */
void Sorting(char *path, char *root)
{ char *name = malloc(strlen(path)+strlen(root)+10);

  if (PRO_TABLE == NULL)
    { if (fk_finish(GPU,&RES) != FK_OK)
        die("fk_finish");
      sprintf(name,"%s/%s.hist",path,root);
      if (fk_write_hist(&RES,KMER,name) != FK_OK)
        die("writing .hist");
    }
  else
    { /* -p:<table>: only profiles, counts from the given table (README.md:112-120) */
      int      tb = PRO_TABLE->tbyte;
      uint8   *recs = malloc((size_t) (PRO_TABLE->nels > 0 ? PRO_TABLE->nels : 1)*tb);
      int64    n = 0;
      if (recs == NULL)
        { fprintf(stderr,"%s: Out of memory\n",Prog_Name); Clean_Exit(1); }
      for (First_Kmer_Entry(PRO_TABLE); PRO_TABLE->csuf != NULL; Next_Kmer_Entry(PRO_TABLE))
        Current_Entry(PRO_TABLE,recs+(n++)*tb);
      if (fk_set_table(GPU,recs,n) != FK_OK)
        die("fk_set_table");
      free(recs);
    }
  if (DO_PROFILE)
    { fk_profiles pr;
      if (fk_make_profiles(GPU,NULL,0,&pr) != FK_OK)
        die("fk_make_profiles");
      if (fk_write_prof(&pr,KMER,ITHREADS,path,root) != FK_OK)     /* one part per INPUT thread (merge.c:880-930): fewer than
                                                                      -T for a small input */
        die("writing .prof");
    }
  free(name);
}

/* replaces Merge_Tables (table.c:346) */
void Merge_Tables(char *path, char *root)
{ if (DO_PROFILE && DO_TABLE > 1)       /* counted with cutoff 1 for the look-ups: keep count >= -t */
    { int      kw = ((2*KMER+7)>>3) + 2;
      const uint8 *t = RES.table;
      uint8   *keep = malloc((size_t) (RES.ntable > 0 ? RES.ntable : 1)*kw);
      int64    n = 0, x;
      if (keep == NULL)
        { fprintf(stderr,"%s: Out of memory\n",Prog_Name); Clean_Exit(1); }
      for (x = 0; x < RES.ntable; x++)
        if ((t[x*kw+kw-2] | (t[x*kw+kw-1] << 8)) >= DO_TABLE)
          memcpy(keep+(n++)*kw,t+x*kw,(size_t) kw);
      RES.table  = keep;
      RES.ntable = n;
    }
  if (fk_write_ktab(&RES,KMER,DO_TABLE,NTHREADS,path,root) != FK_OK)
    die("writing .ktab");
}

/* replaces Merge_Profiles (merge.c): the profiles were written in read order by Sorting */
void Merge_Profiles(char *path, char *root)
{ (void) path; (void) root; }
