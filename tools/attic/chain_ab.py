import sys, torch
sys.path[:0]=['/root/repo/retrieval-fuse_amd']
import model as rf_model
from rfuse import ops
from model.unet import Conv3dParams
dev=torch.device('cuda:0')
def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps
enc=rf_model.PCPatch48(12,64).to(dev).eval()
convs=[l for l in enc.layers if isinstance(l,Conv3dParams)]
with torch.no_grad():
    x=torch.randn(16,1,144,144,144,device=dev)
    for i,layer in enumerate(convs[:4]):
        fp=enc._conv(layer, x.data if isinstance(x,ops.SplitActs) else x) if not isinstance(x,ops.SplitActs) else None
        if i==0:
            t_fp=timed(lambda: enc._conv(layer,x)); t_sp=timed(lambda: enc._conv(layer,x,out_split=True))
            print('layer0 valu: fp32 out %.3f ms, split out %.3f ms'%(t_fp,t_sp))
            xf=enc._conv(layer,x); xs=enc._conv(layer,x,out_split=True)
        else:
            a=timed(lambda: enc._conv(layer,xf)); b=timed(lambda: enc._conv(layer,xf,out_split=True)); c=timed(lambda: enc._conv(layer,xs)); d=timed(lambda: enc._conv(layer,xs,out_split=True))
            print('layer%d split kernel: fp32->fp32 %.3f, fp32->split %.3f, split->fp32 %.3f, split->split %.3f ms'%(i,a,b,c,d))
            xf=enc._conv(layer,xf); xs=enc._conv(layer,xs,out_split=True)
    g=torch.randn(16,1,144,144,144,device=dev); w=torch.randn(1024,1,48,48,48,device=dev)
    for flag in (False, True, False, True):
        ops.USE_SPLIT_CHAIN=flag
        print('USE_SPLIT_CHAIN', flag, 'grid %.3f ms, windows %.3f ms'%(timed(lambda: enc.forward_grid(g,48,32)), timed(lambda: enc(w))))
    import types
    orig=enc._conv
    def spy(layer,x,out_split=False):
        print('   conv', layer.in_channels,'->',layer.out_channels,'in', 'split' if isinstance(x,ops.SplitActs) else 'fp32', 'out_split', out_split, tuple(x.shape[:3]))
        return orig(layer,x,out_split=out_split)
    enc._conv=spy
    enc.forward_grid(g,48,32)
