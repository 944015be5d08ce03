/* gpu_path.c -- the shim of INTEGRATION.md, compilable: it replaces split.c, count.c, table.c and
 * MSDsort.c in a build of the REFERENCE FastK (FastK.c, io.c, merge.c, LSDsort.c, libfastk.c stay
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
  if (DO_PROFILE || PRO_TABLE != NULL || COMPRESS)
    { fprintf(stderr,"%s: -p and -c are not built on the GPU path\n",Prog_Name);
      Clean_Exit(1);
    }
  fk_default_params(&p);
  p.kmer = KMER; p.table_cutoff = DO_TABLE; p.nthreads = NTHREADS; p.bc_prefix = BC_PREFIX;
  p.exact_parts = 1;
  if (fk_create(&p,&GPU) != FK_OK)
    die("fk_create");
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

void Split_Table(char *root)
{ (void) root;
  fprintf(stderr,"%s: -p:table is not built on the GPU path\n",Prog_Name);
  Clean_Exit(1);
}

/* replaces Sorting (count.c:1202) */
void Sorting(char *path, char *root)
{ char *name = malloc(strlen(path)+strlen(root)+10);

  if (fk_finish(GPU,&RES) != FK_OK)
    die("fk_finish");
  sprintf(name,"%s/%s.hist",path,root);
  if (fk_write_hist(&RES,KMER,name) != FK_OK)
    die("writing .hist");
  free(name);
}

/* replaces Merge_Tables (table.c:346) */
void Merge_Tables(char *path, char *root)
{ if (fk_write_ktab(&RES,KMER,DO_TABLE,NTHREADS,path,root) != FK_OK)
    die("writing .ktab");
  fk_destroy(GPU);
  GPU = NULL;
}
