/* Prints where FastK_amd's input_threads() lets the reference's input threads begin (TEST INFRASTRUCTURE: includes the
   driver's source and calls its static function; tests/test_host_input_threads.py compares the read ranges with the
   .pidx parts the reference itself writes).   input_threads_print <T> <fastq 0|1> <file> ...  ->  "<file index> <offset>" per thread */
#define main fastk_amd_driver_main
#include "../../fastk_amd/csrc/host/FastK_amd.c"
#undef main

int main(int argc, char **argv)
{ Feeder f;
  int    t;
  if (argc < 4) return (2);
  memset(&f,0,sizeof(f));
  NTHREADS = atoi(argv[1]);
  input_threads(&f,argv+3,argc-3,atoi(argv[2]));
  for (t = 0; t < f.nstarts; t++)
    printf("%d %lld\n",f.st_file[t],(long long) f.starts[t]);
  return (0);
}
