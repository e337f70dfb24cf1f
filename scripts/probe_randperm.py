import time, torch
print("threads", torch.get_num_threads())
for n in (90941, 196615):
    for thr in (None, 1, 8):
        if thr: torch.set_num_threads(thr)
        torch.randperm(n)
        t0=time.perf_counter()
        for _ in range(20): torch.randperm(n)
        print(n, "threads", torch.get_num_threads(), f"{(time.perf_counter()-t0)/20*1e3:.2f} ms")
g=torch.Generator(); g.manual_seed(1)
torch.set_num_threads(1); a=torch.randperm(1000, generator=g)
g.manual_seed(1); torch.set_num_threads(64); b=torch.randperm(1000, generator=g)
print("same perm across thread counts:", torch.equal(a,b))
g.manual_seed(1); torch.set_num_threads(1); a=torch.randperm(200000, generator=g)
g.manual_seed(1); torch.set_num_threads(64); b=torch.randperm(200000, generator=g)
print("same perm (200k):", torch.equal(a,b))
