/* Fastmerge_amd.c -- Fastmerge's table / histogram merge over libfastk_amd.so.
 *
 *   Fastmerge_amd [-ht] [-T<int(4)>] [-#<int(1)>] [-P<dir>] <target> <source>[.hist|.ktab] ...
 *
 * Same options and outputs as the reference tool for tables and histograms (Fastmerge.c:26-29,
 * 521-540): -t writes <target>.ktab + hidden parts, -h writes <target>.hist; k-mers present in
 * several sources get the sum of their counts, saturated at 32767 with the reference's bookkeeping of
 * the histogram's high-count field (Fastmerge.c:313-329, 985-1030).  The k-mers are summed on the GPU
 * by the aggregation kernel of the counting path (fk_merge_tables).  The prefix-index width follows
 * Fastmerge's rule on the number of input entries (Fastmerge.c:742-756), the cutoff field is the
 * smallest of the sources' (Fastmerge.c:719-720); the part boundaries are our own.
 * -#<n>: n hidden part files per thread (README.md:190); -P is accepted and ignored (no caching is
 * needed); profiles cannot be merged by the reference either (README.md:176-179: use -p:<merged table>).
 * Not built: -S slices.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <fcntl.h>
#include <unistd.h>

#include "../../../include/fastk_amd.h"

static char *Prog_Name = "Fastmerge_amd";

static void die(fk_ctx *ctx, const char *what)
{ fprintf(stderr,"%s: %s: %s\n",Prog_Name,what,fk_last_error(ctx));
  exit (1);
}

#include "ktab_io.h"

/* high-count field of <dir>/<root>.hist (count.c:1893-1910: k, low, high, int64 ilowcnt, int64 ihighcnt, ...) */
static int64_t load_high_count(const char *dir, const char *root, int *found)
{ char name[4096];
  int  h[3], fd;
  int64_t v[2];

  snprintf(name,sizeof(name),"%s/%s.hist",dir,root);
  fd = open(name,O_RDONLY);
  *found = 0;
  if (fd < 0)
    return (0);
  if (read_all(fd,h,12) | read_all(fd,v,16))
    { close(fd); return (0); }
  close(fd);
  *found = 1;
  return (v[1]);
}

int main(int argc, char *argv[])
{ int  DO_HIST = 0, DO_TABLE = 0, NTHREADS = 4, NPARTS = 1;
  int  i, j, narg, kmer = 0, minval = 0x10000, ib;
  char *odir, *oroot, name[4096];
  uint8_t *recs = NULL;
  int64_t  n = 0, cap = 0, high = 0;
  fk_widths  w;
  fk_params  prm;
  fk_ctx    *ctx;
  fk_result *res;

  for (i = j = 1; i < argc; i++)
    if (argv[i][0] == '#' && argv[i][1] != '\0')      /* the usage line of the reference writes it without '-' */
      NPARTS = atoi(argv[i]+1);
    else if (argv[i][0] == '-' && argv[i][1] != '\0')
      { char *p;
        if (argv[i][1] == 'T')
          { NTHREADS = atoi(argv[i]+2); continue; }
        if (argv[i][1] == '#')                         /* parts per thread, Fastmerge.c:26, README.md:190 */
          { NPARTS = atoi(argv[i]+2); continue; }
        if (argv[i][1] == 'P')                         /* local cache directory: nothing is cached here */
          continue;
        for (p = argv[i]+1; *p; p++)
          if (*p == 'h') DO_HIST = 1;
          else if (*p == 't') DO_TABLE = 1;
          else
            { fprintf(stderr,"%s: -%c is not built in this tool (see the header of Fastmerge_amd.c)\n",Prog_Name,*p);
              exit (1);
            }
      }
    else
      argv[j++] = argv[i];
  argc = j;
  if (argc < 3 || NTHREADS < 1 || NPARTS < 1)
    { fprintf(stderr,"\nUsage: %s [-ht] [-T<int(4)>] [-#<int(1)>] [-P<dir>] <target> <source>[.hist|.ktab] ...\n",Prog_Name);
      exit (1);
    }
  if (DO_HIST + DO_TABLE == 0)
    { fprintf(stderr,"%s: At least one of -h or -t must be set\n",Prog_Name); exit (1); }
  split_path(argv[1],"",&odir,&oroot);
  narg = argc-2;

  /* k is in every stub: read it from the first one to size the records */
  { char *d, *r;
    int   fd, k;
    split_path(argv[2],".ktab",&d,&r);
    snprintf(name,sizeof(name),"%s/%s.ktab",d,r);
    fd = open(name,O_RDONLY);
    if (fd < 0 || read_all(fd,&k,4))
      { fprintf(stderr,"%s: Cannot open table %s\n",Prog_Name,name); exit (1); }
    close(fd);
    if (fk_get_widths(k,&w) != FK_OK)
      { fprintf(stderr,"%s: k = %d is not supported\n",Prog_Name,k); exit (1); }
    free(d); free(r);
  }
  for (i = 0; i < narg; i++)
    { char *d, *r;
      int   found;
      split_path(argv[i+2],".ktab",&d,&r);
      { size_t l = strlen(r);                          /* a source may also be named <root>.hist */
        if (l > 5 && strcmp(r+l-5,".hist") == 0) r[l-5] = '\0';
      }
      load_table(d,r,w.kmer_word,&kmer,&minval,&recs,&n,&cap,0);
      high += load_high_count(d,r,&found);
      if (!found && i == 0 && DO_HIST)
        fprintf(stderr,"%s: Warning: no input histograms => overflow count low\n",Prog_Name);
      free(d); free(r);
    }

  fk_default_params(&prm);
  prm.kmer = kmer; prm.table_cutoff = 1; prm.nthreads = NTHREADS;
  if (fk_create(&prm,&ctx) != FK_OK)
    die(NULL,"fk_create");
  res = malloc(sizeof(fk_result));
  if (fk_merge_tables(ctx,recs,n,high,res) != FK_OK)
    die(ctx,"fk_merge_tables");

  if (n >= 0x8000000ll) ib = 3;                        /* Fastmerge.c:742-756, on the INPUT entries */
  else if (n >= 0x80000ll) ib = 2;
  else ib = 1;
  if (DO_TABLE && fk_write_ktab_ex(res,kmer,minval,NTHREADS*NPARTS,ib,odir,oroot) != FK_OK)
    die(ctx,"writing .ktab");
  if (DO_HIST)
    { snprintf(name,sizeof(name),"%s/%s.hist",odir,oroot);
      if (fk_write_hist(res,kmer,name) != FK_OK)
        die(ctx,"writing .hist");
    }
  fk_destroy(ctx);
  free(recs); free(res);
  exit (0);
}
