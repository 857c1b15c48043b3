"""GPU: marching cubes on the device (csrc/mesh.hip through rfuse/mesh.py; reference util/visualization.py:34-37) against the CPU restatement
(oracle/mesh.py) -- same vertices in the same order, same triangles -- and, at scene size, through properties that need no restatement: every vertex on a
crossed grid edge at the interpolated position, every mesh edge used by exactly two triangles with opposite directions (closed, consistently oriented)."""
import numpy as np
import pytest
import torch

from oracle import mesh as omesh
from test_mesh_cpu import shell

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def gpu():
    if not torch.cuda.is_available():
        pytest.skip('no GPU visible')
    return torch.device('cuda:0')


@pytest.mark.parametrize('case', ['shell', 'smooth', 'ragged', 'empty', 'noise'])
def test_device_marching_cubes_equals_the_restatement(gpu, case):
    from rfuse import mesh
    rng = np.random.default_rng(5)
    level = 0.75
    if case == 'shell':
        vol = shell(18, 5.6, (8.3, 8.9, 8.1))
    elif case == 'smooth':
        g = np.stack(np.meshgrid(*[np.linspace(0, 3, 14)] * 3, indexing='ij'), -1)
        vol = (np.sin(g[..., 0] * 2.1) * np.cos(g[..., 1] * 1.7) + np.sin(g[..., 2] * 2.9) + 1.0).astype(np.float32)
    elif case == 'ragged':
        vol = shell(13, 4.2)[:9, :, :7].copy()                 # 9 x 13 x 7: the surface runs out of the volume (open boundary)
    elif case == 'empty':
        vol = np.full((6, 5, 4), 3.0, np.float32)
    else:
        vol = rng.random((10, 11, 9)).astype(np.float32) * 1.5     # white noise: every corner configuration, ambiguous faces included
    v, t = mesh.marching_cubes(torch.from_numpy(vol).to(gpu), level)
    rv, rt = omesh.marching_cubes_reference(vol, level)
    assert v.shape == (len(rv), 3) and t.shape == (len(rt), 3)
    if len(rv):
        assert np.abs(v.cpu().numpy() - rv).max() <= 1e-6
        np.testing.assert_array_equal(t.cpu().numpy(), rt)
    if case in ('shell', 'noise'):
        inv = omesh.mesh_invariants(vol, level, v.cpu().numpy(), t.cpu().numpy())
        assert inv['on_edges'] < 1e-5 and inv['oriented']
        if case == 'shell':
            assert set(inv['edge_uses']) == {2}


def test_scene_sized_volume_properties_and_obj(gpu, tmp_path):
    """a 128 x 128 x 64 scene (two refined chunks a side): properties only -- and the .obj the reference's loop would write"""
    from rfuse import mesh
    rng = np.random.default_rng(11)
    g = np.stack(np.meshgrid(np.arange(128), np.arange(128), np.arange(64), indexing='ij'), -1).astype(np.float32)
    d = np.full((128, 128, 64), 1e9, np.float32)
    for _ in range(6):                                          # union of spheres well inside the volume: a closed surface
        c = rng.uniform([25, 25, 20], [100, 100, 44])
        d = np.minimum(d, np.abs(np.linalg.norm(g - c.astype(np.float32), axis=-1) - rng.uniform(6, 14)))
    vol = np.minimum(d, 3.0).astype(np.float16).astype(np.float32)         # the scene driver's float16 values
    v, t = mesh.marching_cubes(torch.from_numpy(vol).to(gpu), 0.75)
    vn, tn = v.cpu().numpy().astype(np.float64), t.cpu().numpy().astype(np.int64)
    assert len(tn) > 10_000 and tn.min() == 0 and tn.max() == len(vn) - 1
    # every vertex on one grid edge, at the interpolated position, and the edge is crossed
    frac = vn - np.floor(vn)
    assert ((frac > 0).sum(1) <= 1).all()
    axis = np.argmax(frac > 0, axis=1)
    lo = np.floor(vn).astype(int)
    hi = lo.copy()
    hi[np.arange(len(hi)), axis] += 1
    v0, v1 = vol[lo[:, 0], lo[:, 1], lo[:, 2]].astype(np.float64), vol[hi[:, 0], hi[:, 1], hi[:, 2]].astype(np.float64)
    moving = (frac > 0).any(1)
    assert ((v0 < 0.75) != (v1 < 0.75))[moving].all()
    tt = (0.75 - v0) / np.where(v1 == v0, 1.0, v1 - v0)
    assert np.abs(frac[np.arange(len(frac)), axis] - tt)[moving].max() < 1e-5
    # closed and consistently oriented: every directed edge once, its reverse once
    e = np.concatenate([tn[:, [0, 1]], tn[:, [1, 2]], tn[:, [2, 0]]])
    key = e[:, 0] * (len(vn) + 1) + e[:, 1]
    rev = e[:, 1] * (len(vn) + 1) + e[:, 0]
    assert len(np.unique(key)) == len(key)
    assert np.array_equal(np.sort(key), np.sort(rev))
    mesh.visualize_sdf_as_mesh(torch.from_numpy(vol).to(gpu), tmp_path / 'scene.obj', level=0.75, scale_factor=2)
    lines = (tmp_path / 'scene.obj').read_text().splitlines()
    assert sum(l.startswith('v ') for l in lines) == len(vn) and sum(l.startswith('f ') for l in lines) == len(tn)
    first = [float(x) for x in lines[0].split()[1:]]
    assert np.allclose(first, vn[0] / 2, atol=1e-6)
