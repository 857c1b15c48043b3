import torch
dev='cuda:0'
x=torch.empty(1<<30, dtype=torch.float32, device=dev)   # 4 GB
y=torch.empty(1<<30, dtype=torch.float32, device=dev)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps
ms=t(lambda: x.fill_(1.0)); print('fill 4 GB: %.3f ms = %.2f TB/s write' % (ms, 4.295/ms))
ms=t(lambda: y.copy_(x)); print('copy 4 GB: %.3f ms = %.2f TB/s read+write' % (ms, 8.59/ms))
ms=t(lambda: x.sum()); print('sum 4 GB: %.3f ms = %.2f TB/s read' % (ms, 4.295/ms))
