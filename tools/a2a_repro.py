#!/usr/bin/env python3
"""Minimal repro for the record loss seen in round 1 with one large torch.distributed.all_to_all_single
(uint8 payload of 4.7 GB on the RCCL backend): which sizes lose data, and is it the element count?

    python tools/a2a_repro.py                 # world 1 on device 0
    FK_RANKS=2 python tools/a2a_repro.py      # parent starts 2 ranks sharing device 0 (own NCCL_HOSTID each)

For every size the payload is a deterministic byte pattern; after the collective every rank checks the
bytes it received against the pattern of their sender (count of wrong bytes and the first wrong offset).
The same sizes are also sent as int64 elements (8x fewer elements for the same bytes) and in <= 1 GiB
slices (what shard.py does since round 1)."""
import os
import subprocess
import sys

SIZES = [(1 << 29) - 4096, (1 << 29) + (1 << 20), (1 << 30) + 12345, (1 << 31) + 4096, 4_700_000_000]


def pattern(torch, n, salt, dev):
    i = torch.arange(n, dtype=torch.int64, device=dev)
    return ((i * 2654435761 + salt * 40503) >> 7).to(torch.uint8)


def worker():
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    for nbytes in SIZES:
        per = nbytes // world // 8 * 8                   # bytes to each peer
        for mode in ("uint8", "int64", "sliced"):
            send = torch.cat([pattern(torch, per, rank * 16 + d, dev) for d in range(world)])
            recv = torch.zeros(per * world, dtype=torch.uint8, device=dev)
            if mode == "uint8":
                dist.all_to_all_single(recv, send, [per] * world, [per] * world)
            elif mode == "int64":
                dist.all_to_all_single(recv.view(torch.int64), send.view(torch.int64), [per // 8] * world, [per // 8] * world)
            else:
                step = 1 << 30
                for o in range(0, per, step):
                    n = min(step, per - o)
                    s = torch.cat([send[d * per + o: d * per + o + n] for d in range(world)])
                    r = torch.empty_like(s)
                    dist.all_to_all_single(r, s, [n] * world, [n] * world)
                    for d in range(world):
                        recv[d * per + o: d * per + o + n] = r[d * n:(d + 1) * n]
            torch.cuda.synchronize()
            bad, first = 0, -1
            for s_ in range(world):
                exp = pattern(torch, per, s_ * 16 + rank, dev)
                diff = recv[s_ * per:(s_ + 1) * per] != exp
                nb = int(diff.sum().item())
                if nb and first < 0:
                    first = s_ * per + int(torch.argmax(diff.to(torch.uint8)).item())
                bad += nb
                del exp, diff
            print("rank %d world %d: %d bytes per rank pair x %d, %-6s -> %d wrong bytes%s" % (
                rank, world, per, world, mode, bad, "" if not bad else " (first at byte %d = 2^29 + %d)" % (first, first - (1 << 29))),
                flush=True)
            del send, recv
    dist.destroy_process_group()


def main():
    if "RANK" in os.environ and os.environ.get("FK_A2A_WORKER") == "1":
        return worker()
    world = int(os.environ.get("FK_RANKS", "1"))
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT="29731",
                   FK_A2A_WORKER="1", NCCL_HOSTID="fk-rank-%d" % r)
        env.setdefault("NCCL_SOCKET_IFNAME", "lo")
        env.setdefault("NCCL_IB_DISABLE", "1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=env))
    sys.exit(max(p.wait() for p in procs))


if __name__ == "__main__":
    main()
