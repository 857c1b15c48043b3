"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the iso-surface extraction at the end of the reference's inference loop
(util/visualization.py:34-37: ``marching_cubes(sdf, 0.75)`` -> vertices, triangles -> .obj; called from trainer/train_refinement.py:170-173).

PARITY UNPINNED: the algorithm lives in the third-party ``marching_cubes`` package, which is neither under /root/reference nor pinned
(requirements.txt does not list it).  This file restates the published marching-cubes construction (Lorensen & Cline 1987: corners classified against
the level, one vertex per crossed cube edge by linear interpolation, polygons around the inside corners) with the face rule the product documents
(rfuse/mesh.py:build_tables): on every cube face, each maximal run of inside corners -- walking the face counter-clockwise seen from outside -- is cut
off by ONE segment from the run's exit crossing to its entry crossing.  It works cube by cube on the geometry, without a case table, so that it is an
independent coding of the same rule; tests compare the device result with it (same vertex and triangle order) and check invariants that do not depend
on it (tests/test_mesh_gpu.py).  Only tests/ may import this module.
"""
import numpy as np

# the six faces as corner coordinates (x, y, z), counter-clockwise seen from outside the cube
_FACES = (((0, 0, 0), (0, 0, 1), (0, 1, 1), (0, 1, 0)),   # x = 0
          ((1, 0, 0), (1, 1, 0), (1, 1, 1), (1, 0, 1)),   # x = 1
          ((0, 0, 0), (1, 0, 0), (1, 0, 1), (0, 0, 1)),   # y = 0
          ((0, 1, 0), (0, 1, 1), (1, 1, 1), (1, 1, 0)),   # y = 1
          ((0, 0, 0), (0, 1, 0), (1, 1, 0), (1, 0, 0)),   # z = 0
          ((0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)))   # z = 1


def _grid_edge(base, p, q):
    """the grid edge between cube corners p and q (unit apart) of the cube at ``base``: (x, y, z, axis) of its lower end"""
    axis = [i for i in range(3) if p[i] != q[i]][0]
    lo = p if p[axis] == 0 else q
    return (base[0] + lo[0], base[1] + lo[1], base[2] + lo[2], axis)


def _coplanar_in_a_face(base, e1, e2):
    """two grid edges of the cube at ``base`` lie in one of its six faces: all four end points agree in one coordinate"""
    pts = []
    for (x, y, z, axis) in (e1, e2):
        lo = [x - base[0], y - base[1], z - base[2]]
        hi = list(lo)
        hi[axis] += 1
        pts += [lo, hi]
    return any(len({p[i] for p in pts}) == 1 for i in range(3))


def marching_cubes_reference(volume, level=0.75):
    """volume [X, Y, Z] -> (vertices float32 [V, 3], triangles int32 [T, 3]); inside = value < level.  Vertices in the order of their grid edge
    (x, y, z, axis), triangles cube by cube (x, y, z), per cube loop by loop in the order of the loops' smallest cube edge (axis * 4 + u + 2 v), each
    loop a fan (apex: see below), wound so that the normal points away from the inside."""
    vol = np.asarray(volume, dtype=np.float32)
    X, Y, Z = vol.shape
    inside = vol < np.float32(level)
    lvl = np.float32(level)
    index, verts = {}, []
    for x in range(X):
        for y in range(Y):
            for z in range(Z):
                for axis, (dx, dy, dz) in enumerate(((1, 0, 0), (0, 1, 0), (0, 0, 1))):
                    x1, y1, z1 = x + dx, y + dy, z + dz
                    if x1 < X and y1 < Y and z1 < Z and inside[x, y, z] != inside[x1, y1, z1]:
                        v0, v1 = vol[x, y, z], vol[x1, y1, z1]
                        t = np.float32(lvl - v0) / np.float32(v1 - v0)
                        p = [np.float32(x), np.float32(y), np.float32(z)]
                        p[axis] = np.float32(p[axis] + t)
                        index[(x, y, z, axis)] = len(verts)
                        verts.append(p)
    tris = []

    def local_id(base, ge):                                   # cube-edge number of a grid edge: orders the loops and picks their first vertex
        ax = ge[3]
        o = [ge[i] - base[i] for i in range(3) if i != ax]
        return ax * 4 + o[0] + 2 * o[1]

    for x in range(X - 1):
        for y in range(Y - 1):
            for z in range(Z - 1):
                base = (x, y, z)
                nxt = {}
                for face in _FACES:
                    ins = [bool(inside[x + c[0], y + c[1], z + c[2]]) for c in face]
                    if all(ins) or not any(ins):
                        continue
                    for k in range(4):
                        if ins[k] and not ins[k - 1]:             # an inside run starts at corner k: it is entered across edge (k - 1, k) ...
                            j = k
                            while ins[(j + 1) % 4]:
                                j = (j + 1) % 4                   # ... and left across edge (j, j + 1)
                            nxt[_grid_edge(base, face[j], face[(j + 1) % 4])] = _grid_edge(base, face[k - 1], face[k])
                done = set()
                for start in sorted(nxt, key=lambda ge: local_id(base, ge)):
                    if start in done:
                        continue
                    loop, cur = [start], nxt[start]
                    done.add(start)
                    while cur != start:
                        loop.append(cur)
                        done.add(cur)
                        cur = nxt[cur]
                    # fan apex: the first vertex of the loop none of whose fan diagonals lies in a face of the cube (two crossed edges of one face are
                    # either neighbours in the loop or belong to a face with four crossings, where the neighbouring cube may draw the same diagonal)
                    for s in range(len(loop)):
                        fan = loop[s:] + loop[:s]
                        if all(not _coplanar_in_a_face(base, fan[0], fan[i]) for i in range(2, len(fan) - 1)):
                            break
                    else:
                        raise AssertionError('no admissible fan')
                    for i in range(1, len(fan) - 1):
                        tris.append((index[fan[0]], index[fan[i + 1]], index[fan[i]]))
    return np.asarray(verts, dtype=np.float32).reshape(-1, 3), np.asarray(tris, dtype=np.int32).reshape(-1, 3)


def mesh_invariants(volume, level, vertices, triangles):
    """facts every correct extraction satisfies, whatever the triangulation inside a cube: -> dict
       on_edges      every vertex lies on a grid edge whose end values straddle the level, at the linearly interpolated position (max deviation)
       edge_uses     histogram {number of triangles using an undirected mesh edge: count}  (closed 2-manifold <=> only 2)
       oriented      every directed edge is used once (consistent winding)"""
    vol = np.asarray(volume, dtype=np.float64)
    v = np.asarray(vertices, dtype=np.float64)
    t = np.asarray(triangles, dtype=np.int64)
    worst = 0.0
    for p in v:
        frac = p - np.floor(p)
        moving = [a for a in range(3) if frac[a] > 0]
        assert len(moving) <= 1, 'vertex %s is not on a grid edge' % (p,)
        a = moving[0] if moving else 0
        lo = np.floor(p).astype(int)
        hi = lo.copy()
        hi[a] += 1
        if not moving:                                       # exactly on a grid point: the level equals a corner value; any incident edge will do
            ok = False
            for a in range(3):
                for d in (1, -1):
                    q = lo.copy(); q[a] += d
                    if 0 <= q[a] < vol.shape[a] and (vol[tuple(lo)] < level) != (vol[tuple(q)] < level):
                        ok = True
            assert ok, 'vertex %s on a grid point that no crossed edge touches' % (p,)
            continue
        v0, v1 = vol[tuple(lo)], vol[tuple(hi)]
        assert (v0 < level) != (v1 < level), 'vertex %s on an edge the level does not cross' % (p,)
        worst = max(worst, abs(frac[a] - (level - v0) / (v1 - v0)))
    directed = {}
    for tri in t:
        for i in range(3):
            e = (int(tri[i]), int(tri[(i + 1) % 3]))
            directed[e] = directed.get(e, 0) + 1
    undirected = {}
    for (a, b), c in directed.items():
        undirected[(min(a, b), max(a, b))] = undirected.get((min(a, b), max(a, b)), 0) + c
    hist = {}
    for c in undirected.values():
        hist[c] = hist.get(c, 0) + 1
    return {'on_edges': worst, 'edge_uses': hist, 'oriented': all(c == 1 for c in directed.values())}
