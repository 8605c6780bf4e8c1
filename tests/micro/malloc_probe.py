import time, torch
torch.cuda.init()
x = torch.empty(1, device='cuda'); torch.cuda.synchronize()
for gb in (0.5, 1, 2, 4, 8, 12, 12, 4):
    torch.cuda.empty_cache()
    t = time.perf_counter()
    a = torch.empty(int(gb * (1 << 30)), dtype=torch.uint8, device='cuda')
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    a.zero_(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    del a
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print('%5.1f GB: malloc %.3f s, first touch (memset) %.3f s, free %.3f s' % (gb, t1 - t, t2 - t1, t3 - t2), flush=True)
