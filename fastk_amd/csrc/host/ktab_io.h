/* ktab_io.h -- reading a FastK k-mer table (<root>.ktab stub + hidden parts, README.md:965-1006)
   into KMER_WORD-byte records; shared by the host drivers.  The including file defines Prog_Name. */
#ifndef FK_KTAB_IO_H
#define FK_KTAB_IO_H

#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static void split_path(const char *arg, const char *suffix, char **dir, char **root)
{ const char *slash = strrchr(arg,'/');
  const char *base  = slash ? slash+1 : arg;
  size_t bl = strlen(base), sl = strlen(suffix);
  if (bl > sl && strcmp(base+bl-sl,suffix) == 0)
    bl -= sl;
  *root = strndup(base,bl);
  *dir  = slash ? strndup(arg,(size_t) (slash-arg)) : strdup(".");
}

static int read_all(int fd, void *p, size_t n)
{ uint8_t *b = (uint8_t *) p;
  while (n > 0)
    { ssize_t r = read(fd,b,n);
      if (r <= 0) return (-1);
      b += r; n -= (size_t) r;
    }
  return (0);
}

/* appends the entries of table <dir>/<root>.ktab to *recs as KMER_WORD-byte records
   (README.md:965-1006: stub = k, nparts, minval, ibyte, cumulative prefix index; part t = k, n,
   n entries without their first ibyte bytes, in table order) */
static void load_table(const char *dir, const char *root, int kmer_word, int *kmer, int *minval,
                       uint8_t **recs, int64_t *n, int64_t *cap, int want_k)
{ char name[4096];
  int  k, nparts, mv, ib, fd, t;
  int64_t nidx, *idx, pre = 0, seen = 0;

  snprintf(name,sizeof(name),"%s/%s.ktab",dir,root);
  fd = open(name,O_RDONLY);
  if (fd < 0)
    { fprintf(stderr,"%s: Cannot open table %s\n",Prog_Name,name); exit (1); }
  if (read_all(fd,&k,4) | read_all(fd,&nparts,4) | read_all(fd,&mv,4) | read_all(fd,&ib,4))
    { fprintf(stderr,"%s: %s is not a k-mer table stub\n",Prog_Name,name); exit (1); }
  nidx = 1ll << (8*ib);
  idx  = malloc(sizeof(int64_t)*(size_t) nidx);
  if (idx == NULL || read_all(fd,idx,sizeof(int64_t)*(size_t) nidx))
    { fprintf(stderr,"%s: Cannot read the index of %s\n",Prog_Name,name); exit (1); }
  close(fd);
  if (want_k > 0 && k != want_k)      /* FastK.c:329-333 */
    { fprintf(stderr,"%s: K-mer size of table %s (%d) does not match -k value (%d)\n",Prog_Name,name,k,want_k);
      exit (1);
    }
  if (*kmer == 0)
    *kmer = k;
  else if (*kmer != k)
    { fprintf(stderr,"%s: K-mer tables do not involve the same K\n",Prog_Name); exit (1); }
  if (mv < *minval)
    *minval = mv;

  for (t = 1; t <= nparts; t++)
    { int     pk;
      int64_t pn, i;
      int     pw = kmer_word - ib;
      uint8_t *buf;

      snprintf(name,sizeof(name),"%s/.%s.ktab.%d",dir,root,t);
      fd = open(name,O_RDONLY);
      if (fd < 0 || read_all(fd,&pk,4) | read_all(fd,&pn,8))
        { fprintf(stderr,"%s: Cannot read table part %s\n",Prog_Name,name); exit (1); }
      buf = malloc((size_t) (pn > 0 ? pn : 1)*pw);
      if (buf == NULL || (pn > 0 && read_all(fd,buf,(size_t) pn*pw)))
        { fprintf(stderr,"%s: Cannot read table part %s\n",Prog_Name,name); exit (1); }
      close(fd);
      if (*n + pn > *cap)
        { *cap  = (*n + pn)*2 + 1024;
          *recs = realloc(*recs,(size_t) *cap*kmer_word);
          if (*recs == NULL)
            { fprintf(stderr,"%s: Out of memory\n",Prog_Name); exit (1); }
        }
      for (i = 0; i < pn; i++)
        { uint8_t *o = *recs + (*n + i)*kmer_word;
          int b;
          while (pre < nidx-1 && idx[pre] <= seen)        /* prefix of entry number `seen` */
            pre += 1;
          for (b = 0; b < ib; b++)
            o[b] = (uint8_t) (pre >> (8*(ib-1-b)));
          memcpy(o+ib,buf+i*pw,pw);
          seen += 1;
        }
      *n += pn;
      free(buf);
    }
  free(idx);
}

#endif
