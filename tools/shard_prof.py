import os, sys, time
sys.path.insert(0, "/root/repo")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
import torch, torch.distributed as dist
import fastk_amd
from tests import shard_model as shard
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
L = 150; glen = 100_000_000; per = int(50 * glen / L); nbytes = per * (L + 1)
ctx = fastk_amd.Context(kmer=40, table_cutoff=1, nbuckets=1)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
reads = torch.empty(nbytes + 64, dtype=torch.uint8, device=dev)
ctx._ck(ctx.L.fk_synth_reads(ctx.h, 20251001, glen, L, 1000, 0, per, reads.data_ptr()))
torch.cuda.synchronize()
eng = shard.HipEngine(ctx, dev)
def T():
    torch.cuda.synchronize(); return time.perf_counter()
for it in range(3):
    t0 = T()
    recs, counts, s_off, ninst = eng.split(reads[:nbytes])
    t1 = T()
    send = torch.tensor(counts, dtype=torch.int64, device=dev); recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send)
    recv_n = [int(c) for c in recv.tolist()]
    t2 = T()
    nrecv = sum(recv_n)
    inbox = torch.empty(max(nrecv, 1) * eng.stride, dtype=torch.uint8, device=dev)[: nrecv * eng.stride]
    t3 = T()
    shard._exchange_records(recs, inbox, [int(c) for c in counts], recv_n, eng.stride, None, s_off)
    t4 = T()
    loc = eng.count_supermers(inbox, nrecv)
    t5 = T()
    tot = torch.zeros(40000, dtype=torch.int64, device=dev); dist.all_reduce(tot); tot.cpu()
    t6 = T()
    print("split %.2f counts %.2f alloc %.2f exchange %.2f count %.2f reduce %.2f total %.2f" % tuple(1e3 * x for x in (t1-t0, t2-t1, t3-t2, t4-t3, t5-t4, t6-t5, t6-t0)))
dist.destroy_process_group()
