"""Time shard.profiles_exchanged (counting + profiles with owner-side look-ups) on one rank:
   python tools/profile_exchanged_bench.py [coverage=50] [genome=100000000] [k=40] [err_ppm=1000]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                         # noqa: E402
import torch.distributed as dist                                     # noqa: E402
import fastk_amd                                                     # noqa: E402
from tests import shard_model as shard                                          # noqa: E402

cov = int(sys.argv[1]) if len(sys.argv) > 1 else 50
G = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
k = int(sys.argv[3]) if len(sys.argv) > 3 else 40
err = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
L = 150
nreads = cov * G // L
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29597")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
with fastk_amd.Context(kmer=k, table_cutoff=1, nbuckets=1) as ctx:
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    nbytes = nreads * (L + 1)
    reads = torch.empty(nbytes + 64, dtype=torch.uint8, device=dev)
    ctx._ck(ctx.L.fk_synth_reads(ctx.h, 20240607, G, L, err, 0, nreads, reads.data_ptr()))
    torch.cuda.synchronize()
    eng = shard.HipEngine(ctx, dev)
    for _ in range(2):
        t0 = time.time()
        tot, data, offs = shard.profiles_exchanged(eng, reads[:nbytes])
        torch.cuda.synchronize()
        print("counting + profiles: %d reads, %d k-mer instances, %.2f GB encoded, %.3f s wall"
              % (len(offs) - 1, tot["ninst"], len(data) / 1e9, time.time() - t0))
dist.destroy_process_group()
