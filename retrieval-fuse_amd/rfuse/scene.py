"""Scene-level recomposition of refined chunks.

Behaviour of the reference's ``PatchedSceneDataset.combine_chunks`` / ``combine_inputs`` / ``combine_targets``
(dataset/patched_scene_dataset.py:153-180), pinned by tests/golden/combine_chunks.npz which was produced by running
that method itself: 3DFront / Matterport3D scenes are cut into 64^3 chunks whose names end in ``__<x>_<y>_<z>`` (origin
of the chunk in target voxels) and share the superscene prefix ``<scene>__<room>``; every other dataset's chunk is a
whole scene at the origin.  The canvas is float64, filled with the truncation value, as large as the farthest chunk
reaches; chunks are pasted in list order (a later chunk overwrites an earlier one at the same origin).

``refine_scene`` is the scene-level driver around the hot path: what the reference's inference loop does per visualisation dataset
(trainer/train_refinement.py:158-169: batches of chunks -> forward_full -> network_pred_to_df -> .cpu().half() -> combine_retrievals(.., 0)).

The recomposition of refined chunks runs on the device (slice copies into a float64 canvas, one transfer per scene); ``combine_chunks`` is the
host-side form the goldens pin.  Mesh export (util/visualization.py:34-37: marching cubes at level 0.75 -> .obj) is rfuse/mesh.py.
"""
import numpy as np

_TILED_DATASETS = ('Matterport3D', '3DFront')


def superscene_and_position(chunk_name, dataset_name):
    """'<scene>__<room>__<x>_<y>_<z>' -> ('<scene>__<room>', int64[3]) for the tiled datasets, else (chunk_name, zeros)."""
    if not dataset_name.startswith(_TILED_DATASETS):
        return chunk_name, np.zeros(3, dtype=np.int64)
    fields = chunk_name.split('__')
    origin = np.array([int(t) for t in fields[-1].split('_')], dtype=np.int64)
    return '__'.join(fields[:2]), origin


def combine_chunks(chunk_names, chunk_volumes, dataset_name, scale_factor=1, chunk_size=64, trunc_val=0.0):
    """-> {superscene: float64 volume}.  ``chunk_volumes[i]`` is the cubic volume of ``chunk_names[i]`` (e.g. a refined df from
    RefinementEngine.refine squeezed to [64,64,64]); its origin is the name's position divided by ``scale_factor`` and
    truncated to int32 (so low-resolution inputs are combined with scale_factor = 64 / input_chunk_size, chunk_size =
    input_chunk_size)."""
    members = {}
    for i, name in enumerate(chunk_names):
        key, origin = superscene_and_position(name, dataset_name)
        members.setdefault(key, []).append((i, (origin / scale_factor).astype(np.int32)))
    scenes = {}
    for key, items in members.items():
        reach = np.max([o for _, o in items], axis=0) + chunk_size
        canvas = np.full(tuple(int(r) for r in reach), trunc_val, dtype=np.float64)
        for i, (x, y, z) in items:
            vol = np.asarray(chunk_volumes[i])
            e = vol.shape[0]
            canvas[x:x + e, y:y + e, z:z + e] = vol
        scenes[key] = canvas
    return scenes


def split_scene(scene_input, input_chunk_size, scene_name, room='room0', pad_value=0.0):
    """Cut a low-resolution scene volume [X, Y, Z] into the chunk grid the tiled datasets use: -> (names, chunks [n, s, s, s] float32).
    Names are '<scene>__<room>__<x>_<y>_<z>' with the chunk origin in TARGET voxels (64 per chunk), i.e. what ``superscene_and_position`` parses;
    a volume that is not a whole number of chunks is padded with ``pad_value`` (the input truncation) at its far faces."""
    vol = np.asarray(scene_input, dtype=np.float32)
    s = int(input_chunk_size)
    grid = [-(-d // s) for d in vol.shape]
    padded = np.full([g * s for g in grid], pad_value, dtype=np.float32)
    padded[:vol.shape[0], :vol.shape[1], :vol.shape[2]] = vol
    names, chunks = [], []
    for ix in range(grid[0]):
        for iy in range(grid[1]):
            for iz in range(grid[2]):
                names.append('%s__%s__%d_%d_%d' % (scene_name, room, ix * 64, iy * 64, iz * 64))
                chunks.append(padded[ix * s:(ix + 1) * s, iy * s:(iy + 1) * s, iz * s:(iz + 1) * s])
    return names, np.stack(chunks)


def combine_predictions(chunk_names, predictions, dataset_name, trunc_val):
    """The reference's ``combine_retrievals(predictions, 0)`` (dataset/patched_scene_dataset.py:182-186): ``predictions`` [n, K or 1, 64, 64, 64]
    in the order of ``chunk_names``; slot 0 of every chunk is pasted at the chunk's origin on a canvas filled with the target truncation."""
    preds = np.asarray(predictions)
    return combine_chunks(chunk_names, [preds[i, 0] for i in range(len(chunk_names))], dataset_name, 1, 64, trunc_val)


def _scene_layout(chunk_names, dataset_name, chunk_size=64):
    """-> {superscene: (canvas shape, [(chunk index, (x, y, z))])}: where ``combine_chunks`` would paste every chunk, in list order"""
    members = {}
    for i, name in enumerate(chunk_names):
        key, origin = superscene_and_position(name, dataset_name)
        members.setdefault(key, []).append((i, tuple(int(v) for v in origin.astype(np.int32))))
        if (origin < 0).any():
            # combine_chunks would follow numpy's slice semantics (an empty or wrapped destination); on the device a negative origin is an offset in front of
            # the canvas buffer.  No dataset of the reference produces one (positions are multiples of 64 from 0): refuse instead of guessing.
            raise ValueError('chunk %r has a negative origin %s: device recomposition needs origins >= 0' % (name, origin.tolist()))
    return {key: (tuple(int(r) for r in (np.max([o for _, o in items], axis=0) + chunk_size)), items) for key, items in members.items()}


def refine_scene(engine, chunk_names, chunk_inputs, batch=32, query_scene=None, patch_mask=None, half=True, assemble_on_device=True):
    """Scene-level inference: low-resolution chunks of one or several superscenes -> {superscene: refined TSDF volume (float64)}.

    chunk_inputs [n, s, s, s] raw (un-normalised) low-resolution chunks, ``chunk_names[i]`` the dataset's chunk name (it carries the position).
    Chunks run through ``engine.refine_stream`` in batches of ``batch`` (the last one ragged; the front end of batch i + 1 beside the back end of batch i);
    predictions are rounded to float16 like the reference's ``network_pred_to_df(pred_shape).cpu().half()`` (``half=False`` keeps fp32) and recomposed
    as ``combine_predictions`` does (reference dataset/patched_scene_dataset.py:160-186: a float64 canvas filled with the target truncation, chunks
    pasted in list order).

    ``assemble_on_device`` (default): the canvases live on the device -- every batch is pasted by slice copies as soon as it is refined, and each finished
    canvas crosses PCIe once, as float64, into pinned host memory (torch's caching host allocator hands the same pages out again once a previous
    result is released).  Round 3 copied float16 chunks to the host and pasted them into a float64 numpy canvas on one thread: 8.4 M conversions per 32
    chunks, four times the GPU time of the chunks themselves (bench.py `scene_driver`: 996 chunks/s against 4547 resident).  False: that host path,
    kept as the cross-check (same values: float16 -> float64 is exact either way)."""
    import torch
    cfg = engine.config
    n = len(chunk_names)
    x = torch.as_tensor(np.asarray(chunk_inputs, dtype=np.float32))
    assert x.shape[0] == n, 'one input chunk per name'
    dev = engine.device
    out_dtype = torch.float16 if half else torch.float32
    dataset_name, trunc = cfg['dataset_train']['dataset_name'], float(engine.target_trunc)
    spans = [(lo, min(lo + batch, n)) for lo in range(0, n, batch)]
    scenes_q = [query_scene[lo * 64:hi * 64] if query_scene is not None else None for lo, hi in spans]
    masks = [patch_mask[lo:hi] if patch_mask is not None else None for lo, hi in spans]
    x_pin = x.pin_memory() if dev.type == 'cuda' and not x.is_pinned() else x
    batches = (x_pin[lo:hi].to(dev, non_blocking=True) for lo, hi in spans)
    stream = engine.refine_stream(batches, scenes_q if query_scene is not None else None, masks if patch_mask is not None else None)
    if not assemble_on_device:
        host = torch.empty((n, 1, 64, 64, 64), dtype=out_dtype).pin_memory()
        copy = torch.cuda.Stream(dev)
        main = torch.cuda.current_stream(dev)
        for (lo, hi), df in zip(spans, stream):
            df = df.to(out_dtype)
            ready = torch.cuda.Event()
            ready.record(main)
            with torch.cuda.stream(copy):
                copy.wait_event(ready)
                host[lo:hi].copy_(df, non_blocking=True)
                df.record_stream(copy)
        copy.synchronize()
        return combine_predictions(chunk_names, host.numpy(), dataset_name, trunc)
    asm = _DeviceAssembly(engine, chunk_names, out_dtype)
    for (lo, hi), df in zip(spans, stream):
        asm.paste(lo, hi, df)
    return asm.result()


class _DeviceAssembly:
    """The canvases of one refine_scene call on the device: batches are pasted as they are refined, finished regions cross PCIe on a copy stream under the
    refinement of the next batch, ``result()`` waits for the last ones and hands out numpy views of one pinned host buffer."""

    def __init__(self, engine, chunk_names, out_dtype):
        import torch
        cfg = engine.config
        self.torch = torch
        self.dev = dev = engine.device
        self.out_dtype = out_dtype
        self.n = n = len(chunk_names)
        dataset_name, trunc = cfg['dataset_train']['dataset_name'], float(engine.target_trunc)
        self.layout = layout = _scene_layout(chunk_names, dataset_name)
        sizes = [int(np.prod(shape)) for shape, _ in layout.values()]
        self.starts = starts = np.concatenate([[0], np.cumsum(sizes)])
        with torch.cuda.device(dev):
            # ONE device buffer and ONE pinned host buffer hold every canvas of the call (a ShapeNet-style dataset has a canvas per chunk: 64 allocations and
            # 64 transfers per 64 chunks otherwise); the returned arrays are views of the host buffer
            self.flat = flat = torch.full((int(starts[-1]),), trunc, dtype=torch.float64, device=dev)
            self.canvases = canvases = {key: flat[starts[k]:starts[k + 1]].view(shape) for k, (key, (shape, _)) in enumerate(layout.items())}
            # paste plan: the chunks that survive (a later chunk at the same origin overwrites an earlier one) with the element offset of their origin in
            # `flat`, grouped by canvas strides -- a batch pastes its members of ALL canvases of one shape with ONE launch of rf_paste_chunks (float16
            # rounding, widening and the strided copy in one pass; round 4 first did this with .half() / .double() / gather / index_put: four passes over
            # 67 MB per 32 chunks, 2 ms per scene).  Canvases with origins off the 64-grid (chunks may overlap: order matters) take slice copies in list order.
            self.plans = plans = {}                                   # canvas -> (surviving chunk ids, their x cells) or None
            by_stride = {}
            for k, (key, (shape, items)) in enumerate(layout.items()):
                if any(d % 64 for d in shape) or any(c % 64 for _, o in items for c in o):
                    plans[key] = None
                    continue
                last = {}
                for i, o in items:
                    last[o] = i
                keep = sorted((i, o) for o, i in last.items())
                plans[key] = ([i for i, _ in keep], [o[0] // 64 for _, o in keep])
                grp = by_stride.setdefault((shape[1] * shape[2], shape[2]), [])
                grp += [(i, int(starts[k]) + (o[0] * shape[1] + o[1]) * shape[2] + o[2]) for i, o in keep]
            self.groups = []                                          # (sx, sy, ids ascending (numpy), offsets (device int64))
            for (sx, sy), grp in by_stride.items():
                grp.sort()
                # rf_paste_chunks writes 64 x 64 runs of 64 elements from each offset and checks nothing: the plan is validated here, on the host
                assert all(0 <= o and o + 63 * sx + 63 * sy + 64 <= flat.numel() for _, o in grp), 'paste plan reaches outside the canvas buffer'
                ids = np.array([i for i, _ in grp], dtype=np.int64)
                # (ids on the device too: a batch's `sel` is a device subtraction -- a host -> device copy per batch would be a synchronous one and stall the
                # software pipeline behind the whole previous batch)
                # ... and uploaded from PINNED memory without blocking: a pageable host -> device copy waits for everything queued on the stream, i.e. for
                # the previous scene, every time a scene starts)
                up = lambda a: torch.from_numpy(a).pin_memory().to(dev, non_blocking=True)
                self.groups.append((sx, sy, ids, up(ids.astype(np.int32)), up(np.array([o for _, o in grp], dtype=np.int64))))
            # transfer regions: x-slabs of 64 voxels of the planned canvases (contiguous in memory; split_scene lists chunks x-outermost, so slabs complete
            # in order), whole canvases otherwise.  A region goes to the pinned host buffer on a copy stream as soon as its last chunk has been pasted --
            # under the refinement of the next batch -- and only the regions the last batch completes are waited for.
            self.host = torch.empty(flat.shape, dtype=torch.float64, pin_memory=True)
            self.copy = torch.cuda.Stream(dev)
            self.main = torch.cuda.current_stream(dev)
            self.regions = regions = []                               # [start, end, chunks still to come]
            self.region_of = region_of = {}                           # chunk index -> region
            for k, (key, (shape, items)) in enumerate(layout.items()):
                slabs = plans[key] is not None and shape[0] > 64
                if not slabs:
                    members = items if plans[key] is None else [(int(i), None) for i in plans[key][0]]
                    regions.append([int(starts[k]), int(starts[k + 1]), len(members)])
                    for i, _ in members:
                        region_of[i] = len(regions) - 1
                    continue
                slab = 64 * shape[1] * shape[2]
                first = len(regions)
                regions += [[int(starts[k]) + ix * slab, int(starts[k]) + (ix + 1) * slab, 0] for ix in range(shape[0] // 64)]
                for i, cell in zip(plans[key][0], plans[key][1]):
                    region_of[int(i)] = first + cell
                    regions[first + cell][2] += 1
            self._ship([r for r, reg in enumerate(regions) if reg[2] == 0])    # regions no chunk lands in: the truncation fill

    def _ship(self, done):
        if not done:
            return
        torch = self.torch
        ready = torch.cuda.Event()
        ready.record(self.main)
        with torch.cuda.stream(self.copy):
            self.copy.wait_event(ready)
            for r in done:
                self.host[self.regions[r][0]:self.regions[r][1]].copy_(self.flat[self.regions[r][0]:self.regions[r][1]], non_blocking=True)

    def paste(self, lo, hi, df):
        """chunks lo .. hi - 1 of the call (a refined batch [hi - lo, 1, 64, 64, 64] on the device) into their canvases; ships the regions they complete"""
        torch, dev = self.torch, self.dev
        from . import ops
        with torch.cuda.device(dev):
            done = []
            for i in range(lo, hi):
                r = self.region_of.get(i)                            # (None: a chunk that a later one at the same origin overwrites)
                if r is not None:
                    self.regions[r][2] -= 1
                    if self.regions[r][2] == 0:
                        done.append(r)
            df = df.contiguous()
            half = self.out_dtype == torch.float16
            vals = None
            for key, (shape, items) in self.layout.items():
                if self.plans[key] is None:
                    if vals is None:
                        vals = df.to(self.out_dtype).to(torch.float64)   # the reference's float16 round trip; widening is exact
                    for i, (ox, oy, oz) in items:
                        if lo <= i < hi:
                            self.canvases[key][ox:ox + 64, oy:oy + 64, oz:oz + 64] = vals[i - lo, 0]
            for sx, sy, ids, ids_dev, offs in self.groups:
                a, b = int(np.searchsorted(ids, lo)), int(np.searchsorted(ids, hi))
                if b > a:
                    ops.paste_chunks(df, ids_dev[a:b] - lo, offs[a:b], sx, sy, self.flat, round_half=half)
            self._ship(done)

    def result(self):
        """waits for the last transfers -> {superscene: float64 volume} (numpy views of the pinned host buffer)"""
        self.copy.synchronize()
        self.flat.record_stream(self.copy)
        host_np = self.host.numpy()
        return {key: host_np[self.starts[k]:self.starts[k + 1]].reshape(shape) for k, (key, (shape, _)) in enumerate(self.layout.items())}


def refine_scenes(engine, scenes, batch=32, half=True):
    """Scene-level inference over MANY scenes: ``scenes`` is an iterable of ``(chunk_names, chunk_inputs)`` (optionally ``+ (query_scene, patch_mask)``) as
    ``refine_scene`` takes them; yields ``refine_scene``'s result for each, in order, and the same values bit for bit.

    One ``engine.refine_stream`` runs through all scenes, so the software pipeline does not drain at a scene's end, and a scene is handed out one batch
    late: its last regions cross PCIe under the back end of the next scene's first batch instead of in front of an idle GPU (a single ``refine_scene``
    call must return finished arrays, so it pays both: 19.0 ms for a 64-chunk scene whose chunks take 15.0 ms)."""
    import collections
    import torch
    dev = engine.device
    out_dtype = torch.float16 if half else torch.float32
    route = collections.deque()                                      # (assembly, lo, hi, last batch of its scene, empty scenes right before this scene) per batch in flight
    qs_of, pm_of = [], []                                            # per batch, indexed by refine_stream
    empties = [0]                                                    # scenes without chunks seen since the last scene with chunks: each yields {} in its place

    def batches():
        for sc in scenes:
            names, inputs = sc[0], sc[1]
            query_scene = sc[2] if len(sc) > 2 else None
            patch_mask = sc[3] if len(sc) > 3 else None
            n = len(names)
            if n == 0:                                               # nothing to refine: no batch carries it, the consumer yields {} at its position
                empties[0] += 1
                continue
            x = torch.as_tensor(np.asarray(inputs, dtype=np.float32))
            assert x.shape[0] == n, 'one input chunk per name'
            x_pin = x.pin_memory() if dev.type == 'cuda' and not x.is_pinned() else x
            asm = _DeviceAssembly(engine, names, out_dtype)
            spans = [(lo, min(lo + batch, n)) for lo in range(0, n, batch)]
            for lo, hi in spans:
                route.append((asm, lo, hi, hi == n, empties[0] if lo == 0 else 0))
                if lo == 0:
                    empties[0] = 0
                qs_of.append(query_scene[lo * 64:hi * 64] if query_scene is not None else None)
                pm_of.append(patch_mask[lo:hi] if patch_mask is not None else None)
                yield x_pin[lo:hi].to(dev, non_blocking=True)

    finished = collections.deque()
    for df in engine.refine_stream(batches(), qs_of, pm_of):
        asm, lo, hi, last, empty_before = route.popleft()
        while finished:                                              # the previous scene: its last transfers ran under the batch that has just been enqueued
            yield finished.popleft().result()
        for _ in range(empty_before):
            yield {}
        asm.paste(lo, hi, df)
        if last:
            finished.append(asm)
    while finished:
        yield finished.popleft().result()
    for _ in range(empties[0]):                                      # empty scenes after the last scene with chunks
        yield {}

