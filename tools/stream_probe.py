"""Dev tool (GPU box): which torch pool streams really run beside the current stream?  Two spin kernels (torch.cuda._sleep) on two streams take T when the
streams sit on different hardware queues and 2T when ROCm mapped them onto the same one (GPU_MAX_HW_QUEUES queues are shared by all HIP streams)."""
import os, sys, time
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import torch
dev = torch.device('cuda:0')
torch.cuda.set_device(dev)
main = torch.cuda.current_stream()
CYC = 2_000_000
def both(a, b):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(a): torch.cuda._sleep(CYC)
    with torch.cuda.stream(b): torch.cuda._sleep(CYC)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3
def one(a):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(a): torch.cuda._sleep(CYC)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3
one(main); base = min(one(main) for _ in range(3))
print('one spin kernel: %.3f ms' % base)
pool = [torch.cuda.Stream(dev) for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12)]
print('vs main :', ' '.join('%.2f' % (min(both(main, s) for _ in range(2)) / base) for s in pool))
for i, s in enumerate(pool[:8]):
    print('vs pool%d:' % i, ' '.join('%.2f' % (min(both(s, t) for _ in range(2)) / base) if t is not s else ' -- ' for t in pool))
