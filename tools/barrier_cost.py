import time, torch, torch.distributed as dist
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29534", rank=0, world_size=1, device_id=dev)
x = torch.zeros(20_000_000, device=dev)
for i in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter(); dist.barrier(); torch.cuda.synchronize(); print("barrier+sync ms", (time.perf_counter() - t0) * 1e3)
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); dist.broadcast(x, 0); torch.cuda.synchronize(); print("bcast 80MB ms", (time.perf_counter() - t0) * 1e3)
dist.destroy_process_group()
