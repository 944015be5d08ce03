/* What FastK_amd -x -p does with its input files before any GPU is involved (TEST INFRASTRUCTURE: includes the driver's
   source, calls its static functions, and takes the place of fk_push_block; tests/test_host_input_threads.py compares the
   output with the .pidx parts the reference itself writes).
     input_threads_print starts <T> <fastq 0|1> <file> ...   "<file index> <offset>" per input thread (input_threads)
     input_threads_print deal   <T> <fastq 0|1> <file> ...   "<reads>" per input thread: the host scanner (scan_file) run over
                                                             the files, the blocks it would push counted by their tid */
#define main fastk_amd_driver_main
#define fk_push_block harness_push_block
#include "../../fastk_amd/csrc/host/FastK_amd.c"
#undef main

static int64_t H_reads[256];
static int     H_last = -1, H_order_ok = 1;

int harness_push_block(fk_ctx *ctx, const char *bases, const int32_t *boff, int nreads, int rem, int tid)
{ (void) ctx; (void) bases; (void) boff; (void) rem;
  if (tid < 0 || tid > 255 || tid < H_last) H_order_ok = 0;      /* the threads' blocks arrive in thread order */
  else { H_reads[tid] += nreads; H_last = tid; }
  return (FK_OK);
}

int main(int argc, char **argv)
{ Feeder f;
  int    t, i, fastq;
  if (argc < 5) return (2);
  memset(&f,0,sizeof(f));
  Prog_Name = "input_threads_print";
  NTHREADS  = atoi(argv[2]);
  fastq     = atoi(argv[3]);
  input_threads(&f,argv+4,argc-4,fastq);
  if (strcmp(argv[1],"starts") == 0)
    { for (t = 0; t < f.nstarts; t++)
        printf("%d %lld\n",f.st_file[t],(long long) f.starts[t]);
      return (0);
    }
  KMER = 12;
  f.cap_bytes = BLOCK_BYTES;
  f.cap_reads = BLOCK_READS;
  f.bases = malloc(BLOCK_BYTES+16);
  f.boff  = malloc(sizeof(int32_t)*(BLOCK_READS+2));
  f.boff[0] = 0;
  for (i = 4; i < argc; i++)
    { f.cur_file = i-4;
      scan_file(&f,argv[i],fastq);
    }
  flush_block(&f,0);
  if (!H_order_ok) return (3);
  for (t = 0; t <= H_last; t++)
    printf("%lld\n",(long long) H_reads[t]);
  return (0);
}
