"""Soak of the one-shot all-reduce (GPU box): P processes on ONE GPU, a graph of 64 calls replayed N times with fresh inputs,
every result compared with the exactly rounded rank-order sum computed on the host side of each rank.  Exercises the tag
sequence far past the mailbox parity and any 16-bit boundary, back-to-back launches without host syncs, and P = 2, 3, 4."""
import os
import socket
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def rank_main(rank, world, port, n, replays, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from qeft_amd.oneshot import OneShotAllReduce
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        osr = OneShotAllReduce(n, dev)
        CALLS = 64
        big = torch.zeros(CALLS, n, device=dev)
        bufs = [big[c] for c in range(CALLS)]
        side = torch.cuda.Stream(dev)
        with torch.cuda.stream(side):
            for b in bufs[:2]:
                osr.all_reduce(b)
        torch.cuda.synchronize(dev)
        dist.barrier()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for b in bufs:
                osr.all_reduce(b)
        bad = 0
        t0 = time.time()
        for it in range(replays):
            # every rank can build every rank's inputs: the expected sum needs no communication
            gens = [torch.Generator().manual_seed(1000 * it + r) for r in range(world)]
            ins = [torch.randn(CALLS, n, generator=g) for g in gens]
            want = ins[0].clone()
            for r in range(1, world):
                want += ins[r]                  # rank order, fp32: what the kernel computes
            big.copy_(ins[rank])
            graph.replay()
            torch.cuda.synchronize(dev)
            osr.check_status()
            got = big.cpu()
            bad += int((got != want).any(dim=1).sum())
            if rank == 0 and it % 200 == 199:
                print(f"  world {world}: {it + 1} replays, {bad} wrong so far", flush=True)
        q.put((rank, bad, replays * CALLS, round(time.time() - t0, 1)))
        dist.barrier()
        osr.close()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    replays = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
    for world in [int(w) for w in (sys.argv[2].split(',') if len(sys.argv) > 2 else '2,3,4'.split(','))]:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=rank_main, args=(r, world, port, 4096 + 1024 * (world - 2), replays, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = sorted(q.get(timeout=900) for _ in range(world))
        for p in procs:
            p.join(timeout=120)
        print(f"world {world}: " + "; ".join(f"rank {r}: {bad} wrong of {tot} calls ({sec} s)" for r, bad, tot, sec in res),
              "exit codes", [p.exitcode for p in procs], flush=True)
