"""Dev tool (GPU box): the leading layers of a conv patch encoder on the whole padded chunk (fully-convolutional evaluation, model/retrieval.py
forward_grid) against the same layers on the 64 windows of the chunk -- per layer, HIP events."""
import sys
from pathlib import Path
import torch
REPO = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(REPO / 'retrieval-fuse_amd')]
import model as rf_model
from rfuse import ops
from model.unet import Conv3dParams

dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, nf, window, step in (('PCPatch48', 12, 48, 32), ('Patch32', 8, 32, 16)):
    enc = getattr(rf_model, name)(nf, 64).to(dev).eval()
    convs = [l for l in enc.layers if isinstance(l, Conv3dParams)]
    on_grid, _, _ = enc.grid_plan(window, step, 4)
    g = 3 * step + window
    xg = torch.randn(B, 1, g, g, g, device=dev)
    xw = torch.randn(B * 64, 1, window, window, window, device=dev)
    with torch.no_grad():
        print('%s, %d chunks: whole encoder on windows %.2f ms, on the grid %.2f ms' % (name, B, timed(lambda: enc(xw)), timed(lambda: enc.forward_grid(xg, window, step))))
        for i, layer in enumerate(convs[:on_grid + 1]):
            tw = timed(lambda: enc._conv(layer, xw))
            tg = timed(lambda: enc._conv(layer, xg)) if i < on_grid else float('nan')
            form = lambda x: 'split' if ops.conv_valid_split_supported(x, layer.out_channels, layer.kernel_size, layer.stride) else 'valu' if ops.conv_valid_valu_supported(x, layer.out_channels, layer.kernel_size, layer.stride) else 'lds/mfma'
            print('  layer %d  %3d -> %3d k%d s%d   windows %3d^3 x %5d: %7.3f ms (%s)   grid %3d^3 x %3d: %7.3f ms (%s)   output voxels %.2fx fewer' % (
                i, layer.in_channels, layer.out_channels, layer.kernel_size, layer.stride, xw.shape[2], xw.shape[0], tw, form(xw), xg.shape[2], xg.shape[0], tg, form(xg),
                xw.shape[0] * ((xw.shape[2] - layer.kernel_size) // layer.stride + 1) ** 3 / (xg.shape[0] * ((xg.shape[2] - layer.kernel_size) // layer.stride + 1) ** 3)))
            xw, xg = enc._conv(layer, xw), enc._conv(layer, xg)
