// fk_files.hip -- the on-disk encodings: <root>.hist, <root>.ktab + hidden parts, <root>.prof + hidden
// .pidx/.prof parts (README.md:936-1069).  Host-only code (no device work), part of libfastk_amd.so.
#include "fk_common.h"
#include "../../include/fk_synth.h"
#include <pthread.h>
#include <stdarg.h>
#include <fcntl.h>
#include <unistd.h>
#include <algorithm>
#include <vector>
#include <thread>

// ---- encodings ----------------------------------------------------------------------------------
static int write_all(int fd, const void *p, size_t n)
{ const uint8_t *b = (const uint8_t *) p;
  while (n > 0)
    { ssize_t wr = write(fd, b, n);
      if (wr < 0) return (-1);
      b += wr; n -= (size_t) wr;
    }
  return (0);
}

// .hist: int k; int 1; int 0x7fff; int64 hist[1]; int64 max_inst; int64 hist[1..0x7fff]
// (count.c:1893-1910, README.md:936-961)
// <root>.prof stub + hidden .<root>.pidx.N / .<root>.prof.N (README "K-mer Profile Files"); part t
// holds reads [t nreads / nparts, (t+1) nreads / nparts)
// parts part0 .. part0 + nhere - 1 of an nparts-part profile set from the reads of p, whose first read is read
// `read_base` of the data set; the stub with stub != 0
extern "C" int fk_write_prof_range(const fk_profiles *p, int kmer, int nparts, int part0, int nhere, int64_t read_base,
                                   int stub, const char *dir, const char *root)
{ if (p == NULL || dir == NULL || root == NULL || nparts < 1 || part0 < 0 || nhere < 1 || part0 + nhere > nparts
      || (p->nreads > 0 && p->offsets == NULL))
    return (FK_EINVAL);
  char path[4096];
  int  bad = 0, fd;
  if (stub)
    { snprintf(path, sizeof(path), "%s/%s.prof", dir, root);
      fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
      if (fd < 0)
        { fk_set_error(NULL, "Cannot open %s for writing", path);
          return (FK_EINVAL);
        }
      int32_t st[2] = { kmer, nparts };
      bad = write_all(fd, st, 8);
      close(fd);
    }
  for (int t = 0; t < nhere && !bad; t++)
    { const bool by_thread = (p->nsplit == nhere && p->split != NULL && nhere == nparts);
      const int64_t r0 = by_thread ? p->split[t] : p->nreads * t / nhere;
      const int64_t r1 = by_thread ? p->split[t + 1] : p->nreads * (t + 1) / nhere;
      const int64_t n = r1 - r0, g0 = read_base + r0;
      const int64_t b0 = (n > 0) ? p->offsets[r0] : 0;
      snprintf(path, sizeof(path), "%s/.%s.pidx.%d", dir, root, part0 + t + 1);
      fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
      if (fd < 0) { bad = 1; break; }
      int32_t k32 = kmer;
      bad |= write_all(fd, &k32, 4) | write_all(fd, &g0, 8) | write_all(fd, &n, 8);
      int64_t buf[4096];
      for (int64_t i = 0; i < n && !bad; i += 4096)
        { const int64_t m = (n - i < 4096) ? n - i : 4096;
          for (int64_t j = 0; j < m; j++)
            buf[j] = p->offsets[r0 + i + j + 1] - b0;
          bad |= write_all(fd, buf, (size_t) m * 8);
        }
      close(fd);
      snprintf(path, sizeof(path), "%s/.%s.prof.%d", dir, root, part0 + t + 1);
      fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
      if (fd < 0) { bad = 1; break; }
      if (n > 0 && p->offsets[r1] > b0)
        bad |= write_all(fd, p->data + b0, (size_t) (p->offsets[r1] - b0));
      close(fd);
    }
  if (bad)
    { fk_set_error(NULL, "Cannot write profile files %s/%s.prof.  Enough disk space?", dir, root);
      return (FK_EINVAL);
    }
  return (FK_OK);
}

extern "C" int fk_write_prof(const fk_profiles *p, int kmer, int nparts, const char *dir, const char *root)
{ return fk_write_prof_range(p, kmer, nparts, 0, nparts, 0, 1, dir, root); }

extern "C" int fk_write_hist(const fk_result *res, int kmer, const char *path)
{ if (res == NULL || path == NULL) return (FK_EINVAL);
  int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
  if (fd < 0)
    { fk_set_error(NULL, "Cannot open %s for writing", path);
      return (FK_EINVAL);
    }
  int32_t h[3] = { kmer, 1, 0x7fff };
  int bad = write_all(fd, h, 12) | write_all(fd, &res->hist[1], 8) | write_all(fd, &res->max_inst, 8)
          | write_all(fd, &res->hist[1], 8 * 0x7fff);
  close(fd);
  if (bad)
    { fk_set_error(NULL, "Cannot write to %s.  Enough disk space?", path);
      return (FK_EINVAL);
    }
  return (FK_OK);
}

// .ktab stub + hidden parts (table.c:162-342, 485-498, README.md:965-1006).  Part t holds the
// first-byte range [split[t], split[t+1]) chosen by the reference's rule (MSDsort.c:330-352 over
// the weighted k-mer first-byte census, count.c:1560-1565).
extern "C" int fk_write_ktab_ex(const fk_result *res, int kmer, int table_cutoff, int nthreads,
                                int idx_bytes, const char *dir, const char *root);

extern "C" int fk_write_ktab(const fk_result *res, int kmer, int table_cutoff, int nthreads,
                             const char *dir, const char *root)
{ return fk_write_ktab_ex(res, kmer, table_cutoff, nthreads, 0, dir, root); }

// ---- .ktab writing, in pieces so that several ranks can each write the parts they hold ------------

extern "C" int fk_ktab_idx_bytes(int kmer, int64_t ntable)       // count.c:1620-1626
{ if (ntable > 0x4000000ll && kmer >= 12) return (3);
  if (ntable >= 0x40000ll && kmer >= 8) return (2);
  return (1);
}

// Table_Split (count.c:1560-1565 with MSDsort.c:330-352): first-byte boundaries of nparts parts that
// balance the weighted k-mer census; split[t] .. split[t+1] is part t's first-byte range
extern "C" int fk_ktab_split(const int64_t *wfirst, int kmer, int nparts, int *split)
{ fk_widths w;
  if (wfirst == NULL || split == NULL || nparts < 1 || fk_get_widths(kmer, &w) != FK_OK)
    return (FK_EINVAL);
  const int KW = w.kmer_word;
  int64_t asize = 0, sum = 0;
  for (int x = 0; x < 256; x++)
    asize += wfirst[x] * KW;
  int64_t thr = asize / nparts;
  int n = 0, beg = 0;
  for (int x = 0; x < 256; x++)
    { sum += wfirst[x] * KW;
      if (sum >= thr && n < nparts)
        { split[n++] = beg;
          thr = (asize * (n + 1)) / nparts;
          beg = x + 1;
        }
    }
  while (n < nparts)
    split[n++] = 256;
  split[nparts] = 256;
  return (FK_OK);
}

// Parts part0 .. part0+nhere-1 of a table from the sorted records that fall into their first-byte
// ranges (records outside are ignored); prefix_counts[p] is incremented for every record written
extern "C" int fk_write_ktab_range(const uint8_t *records, int64_t n, int kmer, int idx_bytes, const int *split,
                                   int part0, int nhere, const char *dir, const char *root,
                                   int64_t *prefix_counts)
{ fk_widths w;
  if (dir == NULL || root == NULL || split == NULL || prefix_counts == NULL || nhere < 0 || part0 < 0
      || idx_bytes < 1 || idx_bytes > 3 || n < 0 || (n > 0 && records == NULL)
      || fk_get_widths(kmer, &w) != FK_OK)
    return (FK_EINVAL);
  const int KW = w.kmer_word, ib = idx_bytes;
  // part boundaries by binary search on the first byte, then one writer thread per part: the parts
  // are disjoint first-byte ranges, so their prefix-index entries are disjoint as well
  std::vector<int64_t> bound((size_t) nhere + 1, 0);
  for (int t = 0; t <= nhere; t++)
    { int64_t lo = (t > 0) ? bound[t - 1] : 0, hi = n;
      while (lo < hi)
        { const int64_t mid = (lo + hi) >> 1;
          if (records[mid * KW] < split[part0 + t]) lo = mid + 1; else hi = mid;
        }
      bound[t] = lo;
    }
  // One writer thread per part, streaming through a buffer of a few MB.  (Building the whole part in fresh memory
  // first cost 27 GB of page faults at configs[2]; pieces of the parts dealt to a pool of 64 threads with pwrite
  // were twice as slow as this: writers of one file queue up behind its inode lock.)  The prefix counts are taken per
  // run of equal prefixes -- the records are sorted.
  std::vector<int> prc((size_t) (nhere > 0 ? nhere : 1), FK_OK);
  const int pw = KW - ib;
  auto write_part = [&](int t)
    { const int64_t lo = bound[t], hi = bound[t + 1], cnt = hi - lo;
      char pname[4096];
      snprintf(pname, sizeof(pname), "%s/.%s.ktab.%d", dir, root, part0 + t + 1);
      int fd = open(pname, O_WRONLY | O_CREAT | O_TRUNC, 0644);
      if (fd < 0) { prc[t] = FK_EINVAL; return; }
      const int64_t bufrecs = std::max<int64_t>((8ll << 20) / pw, 1);
      uint8_t *buf = (uint8_t *) malloc((size_t) bufrecs * pw);
      if (buf == NULL) { close(fd); prc[t] = FK_ENOMEM; return; }
      if (write_all(fd, &kmer, 4) | write_all(fd, &cnt, 8))
        prc[t] = FK_EINVAL;
      int64_t run_pre = -1, run_n = 0;
      for (int64_t x = lo; x < hi && prc[t] == FK_OK; x += bufrecs)
        { const int64_t e = std::min(hi, x + bufrecs);
          for (int64_t i = x; i < e; i++)
            { const uint8_t *rec = records + i * KW;
              int64_t pre = 0;
              for (int b = 0; b < ib; b++)
                pre = (pre << 8) | rec[b];
              if (pre != run_pre)
                { if (run_n > 0)
                    prefix_counts[run_pre] += run_n;
                  run_pre = pre;
                  run_n = 0;
                }
              run_n += 1;
              memcpy(buf + (i - x) * pw, rec + ib, pw);
            }
          if (write_all(fd, buf, (size_t) (e - x) * pw))
            prc[t] = FK_EINVAL;
        }
      if (run_n > 0)
        prefix_counts[run_pre] += run_n;
      free(buf);
      if (close(fd) != 0)
        prc[t] = FK_EINVAL;
    };
  if (nhere > 0)
    { std::vector<std::thread> th;
      for (int t = 1; t < nhere; t++)
        th.emplace_back(write_part, t);
      write_part(0);
      for (auto &x : th)
        x.join();
    }
  for (int t = 0; t < nhere; t++)
    if (prc[t] != FK_OK)
      { fk_set_error(NULL, "Cannot write to %s/.%s.ktab.%d.  Enough disk space?", dir, root, part0 + t + 1);
        return (prc[t]);
      }
  return (FK_OK);
}

// <root>.ktab: k, parts, cutoff, index width, cumulative prefix index (README.md:965-985) from the
// per-prefix counts summed over all parts
extern "C" int fk_write_ktab_stub(int kmer, int nparts, int table_cutoff, int idx_bytes,
                                  const int64_t *prefix_counts, const char *dir, const char *root)
{ if (dir == NULL || root == NULL || prefix_counts == NULL || idx_bytes < 1 || idx_bytes > 3 || nparts < 1)
    return (FK_EINVAL);
  const int64_t nidx = 1ll << (8 * idx_bytes);
  int64_t *idx = (int64_t *) malloc((size_t) nidx * sizeof(int64_t));
  if (idx == NULL) return (FK_ENOMEM);
  int64_t run = 0;
  for (int64_t i = 0; i < nidx; i++)
    { run += prefix_counts[i];
      idx[i] = run;
    }
  char name[4096];
  snprintf(name, sizeof(name), "%s/%s.ktab", dir, root);
  int rc = FK_OK;
  int fd = open(name, O_WRONLY | O_CREAT | O_TRUNC, 0644);
  if (fd < 0)
    rc = FK_EINVAL;
  else
    { int32_t h[4] = { kmer, nparts, table_cutoff, idx_bytes };
      if (write_all(fd, h, 16) | write_all(fd, idx, (size_t) nidx * 8))
        rc = FK_EINVAL;
      close(fd);
    }
  free(idx);
  if (rc != FK_OK)
    fk_set_error(NULL, "Cannot write to %s.  Enough disk space?", name);
  return (rc);
}

// idx_bytes 1..3 fixes the prefix-index width (Fastmerge chooses it from the number of INPUT entries,
// Fastmerge.c:742-756); 0 = FastK's rule on the table size (count.c:1620-1626)
extern "C" int fk_write_ktab_ex(const fk_result *res, int kmer, int table_cutoff, int nthreads,
                                int idx_bytes, const char *dir, const char *root)
{ if (res == NULL || dir == NULL || root == NULL || nthreads < 1 || table_cutoff < 1)
    return (FK_EINVAL);
  if (res->ntable > 0 && res->table == NULL) return (FK_EINVAL);
  const int ib = (idx_bytes >= 1 && idx_bytes <= 3) ? idx_bytes : fk_ktab_idx_bytes(kmer, res->ntable);
  std::vector<int> split((size_t) nthreads + 1);
  int rc = fk_ktab_split(res->wfirst, kmer, nthreads, split.data());
  if (rc != FK_OK) return (rc);
  int64_t *cnt = (int64_t *) calloc((size_t) 1 << (8 * ib), sizeof(int64_t));
  if (cnt == NULL) return (FK_ENOMEM);
  rc = fk_write_ktab_range(res->table, res->ntable, kmer, ib, split.data(), 0, nthreads, dir, root, cnt);
  if (rc == FK_OK)
    rc = fk_write_ktab_stub(kmer, nthreads, table_cutoff, ib, cnt, dir, root);
  free(cnt);
  return (rc);
}

