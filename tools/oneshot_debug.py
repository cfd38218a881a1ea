import os, sys, torch, torch.distributed as dist, torch.multiprocessing as mp
sys.path.insert(0, "/root/repo")
def w(rank, world, n, alloc_first):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29544")
    dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from qeft_amd.oneshot import OneShotAllReduce
    junk = [torch.randn(1000 + 37 * i, device=dev) for i in range(alloc_first)]
    ar = OneShotAllReduce(n, dev, dist.group.WORLD)
    t = torch.full((n,), float(rank + 1), device=dev)
    for it in range(6):
        t.fill_(float(rank + 1))
        for _ in range(4): ar.all_reduce(t)
        torch.cuda.synchronize()
        print(f"n={n} alloc_first={alloc_first} rank {rank} it {it}: t[0]={t[0].item()} (expect {3.0 * 2 ** 3}) status={ar.status.tolist()} seq={ar.seq.item()}", flush=True)
    dist.destroy_process_group()
if __name__ == "__main__":
    for n, af in ((512, 0), (512, 50), (4102, 50)):
        mp.spawn(w, args=(2, n, af), nprocs=2)
