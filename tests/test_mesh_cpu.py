"""CPU: the marching-cubes restatement (oracle/mesh.py) satisfies the invariants of a correct extraction, and the PRODUCT's case table
(rfuse/mesh.py:build_tables, what csrc/mesh.hip walks) agrees with it on every one of the 256 corner configurations -- two independent codings of the
same face rule.  Parity with the reference's third-party `marching_cubes` package is unpinned by nature (not in the reference tree, not pinned)."""
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parents[1]
for p in (str(REPO), str(REPO / 'retrieval-fuse_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)
from oracle import mesh as omesh


def shell(n, radius, centre=None):
    """unsigned distance (in voxels) to a sphere surface: the kind of field the path produces (a truncated unsigned DF); level 0.75 gives two nested spheres"""
    c = np.asarray(centre if centre is not None else [(n - 1) / 2.0] * 3)
    g = np.stack(np.meshgrid(*[np.arange(n)] * 3, indexing='ij'), -1).astype(np.float64)
    return np.minimum(np.abs(np.linalg.norm(g - c, axis=-1) - radius), 3.0).astype(np.float32)


def test_oracle_extraction_of_a_shell_is_a_closed_oriented_manifold():
    vol = shell(20, 6.3, (9.2, 9.7, 9.4))
    v, t = omesh.marching_cubes_reference(vol, 0.75)
    inv = omesh.mesh_invariants(vol, 0.75, v, t)
    assert inv['on_edges'] < 1e-5
    assert set(inv['edge_uses']) == {2} and inv['oriented']
    edges = sum(inv['edge_uses'].values())
    assert len(v) - edges + len(t) == 4                      # Euler characteristic of two spheres
    # outward orientation: normals point away from the inside (value < level), i.e. along the gradient of the field
    p = v[t].astype(np.float64)
    normal = np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0])
    centroid = p.mean(axis=1)
    r = np.linalg.norm(centroid - np.array([9.2, 9.7, 9.4]), axis=1)
    radial = (centroid - np.array([9.2, 9.7, 9.4])) / r[:, None]
    sign = np.where(r > 6.3, 1.0, -1.0)                      # outside sphere: away from the centre; inner sphere: towards it
    assert ((normal * radial).sum(1) * sign > 0).all()


def test_product_case_table_equals_the_restatement_on_all_256_configurations():
    from rfuse import mesh as pmesh
    table, count = pmesh.build_tables()
    assert table.shape == (256, 16) and int(count.max()) == 5 and count[0] == 0 and count[255] == 0
    for cfg in range(256):
        vol = np.ones((2, 2, 2), np.float32)
        for c in range(8):
            if (cfg >> c) & 1:
                vol[c & 1, (c >> 1) & 1, (c >> 2) & 1] = 0.0
        v, t = omesh.marching_cubes_reference(vol, 0.5)
        # vertices come in grid-edge order (x, y, z, axis); map them back to cube-edge numbers axis * 4 + u + 2 v
        ids = []
        for p in v:
            axis = int(np.argmax(p - np.floor(p) > 0))
            o = [int(p[i]) for i in range(3) if i != axis]
            ids.append(axis * 4 + o[0] + 2 * o[1])
        got = [ids[i] for tri in t for i in tri]
        want = [int(e) for e in table[cfg] if e >= 0]
        assert got == want, (cfg, got, want)
        assert len(t) == count[cfg]


def test_obj_writer(tmp_path):
    from rfuse import mesh as pmesh
    v = np.array([[0, 0, 0], [1.5, 0, 0], [0, 2.25, 0]], np.float32)
    t = np.array([[0, 1, 2]], np.int32)
    pmesh.export_obj(v, t, tmp_path / 'm.obj')
    lines = (tmp_path / 'm.obj').read_text().splitlines()
    assert lines == ['v 0.000000 0.000000 0.000000', 'v 1.500000 0.000000 0.000000', 'v 0.000000 2.250000 0.000000', 'f 1 2 3']
