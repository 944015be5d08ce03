"""Minimizer-bucket sharding of the counting path across the GPUs of one node.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  Every rank splits its
own stripe of reads into super-mers grouped by bucket; bucket b belongs to rank b (a bucket is a
function of the canonical minimizer, so equal k-mers always meet on one rank, FastK.h:3-7); one
all-to-all-v of SMER_WORD records replaces the reference's ".T" file shuffle (split.c:1263 <->
count.c:1347); each rank then sorts, expands, sorts and counts its records with no further
communication; the 0x8000-bin histogram and the scalar totals are all-reduced (count.c:1543-1553).

The compute stages sit behind a small engine interface so that the exchange logic can be
exercised on CPU (gloo) in tests with a checker engine; the product engine is HipEngine, which
calls libfastk_amd.so and has no CPU path.
"""
import numpy as np
import torch
import torch.distributed as dist

from .api import Context, HIST_BINS


class HipEngine:
    """Stages on torch CUDA(=HIP) tensors through the C-ABI."""

    def __init__(self, ctx: Context, device):
        self.ctx = ctx
        self.device = device
        self.stride = ctx.w.smer_stride

    def split(self, reads):
        """reads: uint8 tensor in HBM.  Returns (records uint8 tensor, per-bucket counts, ninst)."""
        n = reads.numel()
        ns, ni, counts = self.ctx.split(reads.data_ptr(), n)
        recs = torch.empty(max(ns, 1) * self.stride, dtype=torch.uint8, device=self.device)
        self.ctx.split(reads.data_ptr(), n, recs.data_ptr(), ns)
        return recs[: ns * self.stride], counts, ni

    def count_supermers(self, recs, nsuper):
        res = self.ctx.count_device_supermers(recs.data_ptr() if nsuper else None, nsuper)
        return dict(hist=res.hist, max_inst=res.max_inst, nweighted=res.nweighted,
                    ndistinct=res.ndistinct, ntable=res.ntable, result=res)


def count_sharded(engine, reads, group=None):
    """Run the sharded path; every rank returns the same global totals.

    Returns dict(hist int64[0x8000], max_inst, ninst, nsuper, nweighted, ndistinct, ntable,
    local=<this rank's engine result>)."""
    world = dist.get_world_size(group)
    stride = engine.stride
    recs, counts, ninst = engine.split(reads)
    assert len(counts) == world, "context must be created with nbuckets == world size"
    dev = recs.device

    send = torch.tensor(counts, dtype=torch.int64, device=dev)
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send, group=group)
    send_l = [int(c) * stride for c in counts]
    recv_l = [int(c) * stride for c in recv.tolist()]
    nrecv = sum(recv_l) // stride
    inbox = torch.empty(max(sum(recv_l), 1), dtype=torch.uint8, device=dev)[: sum(recv_l)]
    dist.all_to_all_single(inbox, recs, output_split_sizes=recv_l, input_split_sizes=send_l,
                           group=group)
    del recs

    loc = engine.count_supermers(inbox, nrecv)

    tot = torch.zeros(HIST_BINS + 8, dtype=torch.int64, device=dev)
    tot[:HIST_BINS] = torch.from_numpy(np.asarray(loc["hist"], dtype=np.int64)).to(dev)
    extra = [loc["max_inst"], ninst, nrecv, loc["nweighted"], loc["ndistinct"], loc["ntable"]]
    tot[HIST_BINS:HIST_BINS + 6] = torch.tensor(extra, dtype=torch.int64, device=dev)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=group)
    t = tot.cpu().numpy()
    return dict(hist=t[:HIST_BINS].copy(), max_inst=int(t[HIST_BINS]), ninst=int(t[HIST_BINS + 1]),
                nsuper=int(t[HIST_BINS + 2]), nweighted=int(t[HIST_BINS + 3]),
                ndistinct=int(t[HIST_BINS + 4]), ntable=int(t[HIST_BINS + 5]), local=loc)
