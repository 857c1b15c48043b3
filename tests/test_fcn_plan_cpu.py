"""CPU: the planner of the fully-convolutional patch-encoder evaluation (model/retrieval.py grid_plan) -- which leading layers run on the
whole padded chunk.  The rule: the window origins must stay on the layers' sampling lattice, and the grid must have fewer output voxels
than the windows together (ties go to the grid).  (The arithmetic itself is GPU-tested: tests/test_kernels_gpu.py.)"""
import sys
from pathlib import Path

import pytest

sys.path[:0] = [str(Path(__file__).resolve().parents[1] / 'retrieval-fuse_amd')]
import model as rf_model  # noqa: E402


def brute_force_plan(spec, window, step, npatch):
    """window / grid sizes layer by layer, straight from the definition of a valid strided convolution"""
    sw, sg, lat, n = window, (npatch - 1) * step + window, step, 0
    for _, _, k, st in spec:
        if lat % st:
            break
        sw2, sg2 = (sw - k) // st + 1, (sg - k) // st + 1
        origins = [a * (lat // st) for a in range(npatch)]
        if origins[-1] + sw2 > sg2 or sg2 ** 3 > npatch ** 3 * sw2 ** 3:
            break
        sw, sg, lat, n = sw2, sg2, lat // st, n + 1
    return n, sw, lat


@pytest.mark.parametrize('name,nf,window,step', [('PCPatch48', 12, 48, 32), ('Patch32', 8, 32, 16), ('Patch24V2', 8, 24, 16), ('PCPatch32', 12, 32, 32),
                                                 ('Patch16', 8, 16, 16), ('PCPatch64', 12, 64, 32), ('Patch24', 8, 24, 16), ('Patch12', 8, 12, 8)])
def test_grid_plan(name, nf, window, step):
    enc = getattr(rf_model, name)(nf, 64)
    got = enc.grid_plan(window, step, 4)
    assert got == brute_force_plan(enc.SPEC, window, step, 4)
    on_grid, sw, lat = got
    if window == step:
        assert on_grid == 0                                      # windows that do not overlap: nothing to share
    # the windows of the feature grid reduce to 1^3 through the remaining layers
    for _, _, k, st in enc.SPEC[on_grid:]:
        sw = (sw - k) // st + 1
    assert sw == 1


def test_known_plans():
    assert rf_model.PCPatch48(12, 64).grid_plan(48, 32, 4) == (5, 4, 4)       # 144 -> 140 -> 138 -> 68 -> 33 -> 16, windows of 4^3 every 4 (the last step is a tie: 16^3 = 64 x 4^3)
    assert rf_model.Patch32(8, 64).grid_plan(32, 16, 4) == (5, 4, 4)          # 80 -> 76 -> 74 -> 36 -> 34 -> 16, windows of 4^3 every 4


def test_persistent_grid_form_plan_on_the_host():
    """CPU: which layers rf_conv3d_valid_leaky_split_pg takes and how big its weight image is are host-side decisions of the C-ABI library (no GPU call):
    the 12 -> 17..24 couts (in fours) k = 3 stride-1 layer on even edges 64..254 -- PCPatch32 / 48 / 64's second layer on a padded chunk (csrc/conv_valid_split_pg.hip)"""
    from rfuse import _lib
    lib = _lib.load()
    ok = lib.rf_conv3d_valid_split_pg_supported
    assert ok(16, 12, 140, 24, 3, 1) and ok(1, 12, 64, 20, 3, 1) and ok(3, 12, 254, 24, 3, 1)
    for n, cin, s, cout, k, stride in [(16, 12, 141, 24, 3, 1), (16, 12, 62, 24, 3, 1), (16, 12, 256, 24, 3, 1), (16, 8, 140, 24, 3, 1), (16, 12, 140, 16, 3, 1),
                                       (16, 12, 140, 28, 3, 1), (16, 12, 140, 22, 3, 1), (16, 12, 140, 24, 5, 1), (16, 12, 140, 24, 3, 2), (0, 12, 140, 24, 3, 1)]:
        assert not ok(n, cin, s, cout, k, stride), (n, cin, s, cout, k, stride)
    # image = table of the 21 k-steps' 84 K slot offsets (ints) + 21 k-steps x (h, l) x 64 lanes x 16 bytes of 32 x 16 weight fragments; 0 when the form does not take the layer
    assert lib.rf_convv_split_pg_packed_bytes(24, 12, 3, 140, 1) == 84 * 4 + 21 * 2 * 64 * 16
    assert lib.rf_convv_split_pg_packed_bytes(24, 12, 3, 141, 1) == 0
    # the encoder's own plan sends exactly that layer of PCPatch48 on C5's 144^3 padded chunk there: layer 2 of the 4 on the grid reads a 140^3 volume of 12 channels
    enc = rf_model.PCPatch48(12, 64)
    assert enc.grid_plan(48, 32, 4)[0] == 5 and (enc.SPEC[1][0] * 12, enc.SPEC[1][1] * 12, enc.SPEC[1][2], enc.SPEC[1][3]) == (12, 24, 3, 1)
