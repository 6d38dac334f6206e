import torch, time
dev=torch.device("cuda",0)
n=1<<30
a=torch.empty(n,dtype=torch.uint8,device=dev); b=torch.empty_like(a)
a.fill_(3)
def t(fn,reps=10):
    fn(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/reps
tc=t(lambda: b.copy_(a)); print("copy 1 GiB: %.3f ms -> %.2f TB/s (read+write)"%(tc*1e3, 2*n/tc/1e12))
a32=a.view(torch.int32)
tr=t(lambda: a32.sum()); print("sum  1 GiB: %.3f ms -> %.2f TB/s (read)"%(tr*1e3, n/tr/1e12))
tw=t(lambda: b.fill_(1)); print("fill 1 GiB: %.3f ms -> %.2f TB/s (write)"%(tw*1e3, n/tw/1e12))
# 256 MB working set (pyramid-sized): copy
m=243*(1<<20); a2=a[:m]; b2=b[:m]
tc=t(lambda: b2.copy_(a2),50); print("copy 243 MiB: %.3f ms -> %.2f TB/s (read+write)"%(tc*1e3, 2*m/tc/1e12))
