"""'Next' rows N1 (database build) and N2 (on-disk formats) of SURVEY.md section 8f."""
import numpy as np
import pytest
import torch

import helpers
from oracle import refpath
from rfuse import configs as rf_configs
from rfuse import formats, synthetic


def _target_sd(cfg, seed):
    import model
    _, fenc_target = model.get_retrieval_networks(cfg['retrieval_model'])
    shapes = {k: tuple(v.shape) for k, v in fenc_target.state_dict().items()}
    return fenc_target, helpers.seeded_sd(shapes, seed * 1000 + 22)


@pytest.mark.parametrize('cfg_name', ['C1', 'C5'])
def test_oracle_database_rows_match_reference(cfg_name):
    """oracle restatement of create_dictionary / get_zero_patch_entry vs embeddings captured from the reference's fenc_target"""
    fix = helpers.load_fixture('dbrow_' + cfg_name)
    cfg = rf_configs.get_config(cfg_name)
    _, trunc_t = rf_configs.truncations(cfg)
    _, sd = _target_sd(cfg, int(fix['seed']))
    raw = synthetic.make_chunk(int(fix['seed']) * 100, cfg)['target_raw']
    rows = refpath.database_rows(raw[None], sd, cfg, trunc_t)
    assert rows.shape == (65, 71)
    assert np.abs(rows[:64, 7:] - fix['emb']).max() <= 1e-6
    assert np.abs(rows[64] - fix['zero_row'][0]).max() <= 1e-6
    np.testing.assert_array_equal(rows[:64, 1:7].astype(np.int32), synthetic.patch_boxes_64())


@pytest.mark.gpu
@pytest.mark.parametrize('cfg_name', ['C1', 'C5'])
def test_device_database_build_matches_reference(cfg_name):
    if not torch.cuda.is_available():
        pytest.skip('no GPU visible')
    from rfuse.database import build_database_rows
    fix = helpers.load_fixture('dbrow_' + cfg_name)
    cfg = rf_configs.get_config(cfg_name)
    fenc_target, sd = _target_sd(cfg, int(fix['seed']))
    fenc_target.load_state_dict(sd)
    raws = np.stack([synthetic.make_chunk(int(fix['seed']) * 100 + i, cfg)['target_raw'] for i in range(3)])
    emb, meta = build_database_rows(cfg, fenc_target, raws, 'cuda:0', chunks_per_batch=2)
    emb, meta = emb.cpu().numpy(), meta.cpu().numpy()
    assert emb.shape == (3 * 64 + 1, 64) and meta.shape == (3 * 64 + 1, 7)
    assert np.abs(emb[:64] - fix['emb']).max() <= 2e-6
    assert np.abs(emb[-1] - fix['zero_row'][0, 7:]).max() <= 2e-6
    np.testing.assert_array_equal(meta[-1], fix['zero_row'][0, :7].astype(np.int32))
    np.testing.assert_array_equal(meta[64:128, 0], np.full(64, 1))
    np.testing.assert_array_equal(meta[:64, 1:], synthetic.patch_boxes_64())
    # subset build (occupancy filter stand-in)
    mask = np.zeros((3, 64), dtype=bool)
    mask[1, ::2] = True
    emb2, meta2 = build_database_rows(cfg, fenc_target, raws, 'cuda:0', patch_mask=mask)
    assert emb2.shape[0] == 33 and (meta2[:-1, 0].cpu().numpy() == 1).all()
    np.testing.assert_array_equal(emb2[:-1].cpu().numpy(), emb[64:128][::2])


def test_patch_names_and_database_files_roundtrip(tmp_path):
    fix = helpers.load_fixture('retrieval_map_compose')
    names = formats.chunk_patch_names('sceneA')
    assert len(names) == 64 and names[0] == 'sceneA--0000_0032_0000_0032_0000_0032' and names[-1] == 'sceneA--0048_0080_0048_0080_0048_0080'
    # the reference's own extent enumeration (captured from SceneHandler.get_extents_for_size) gives the same names
    ref_names = [formats.patch_name('sceneA', e) for e in fix['extents_64_16_8_16']]
    assert names == ref_names
    assert formats.parse_patch_name(names[5]) == ('sceneA', [0, 32, 16, 48, 16, 48])
    db = synthetic.make_database(3, rf_configs.get_config('C1'), 200, with_volumes=False)
    formats.save_database(tmp_path, db['meta'], db['emb'], ['s%d' % i for i in range(db['n_scenes'])])
    arr = np.load(tmp_path / 'database.npy')
    assert arr.shape == (201, 71) and arr.dtype == np.float32 and arr[-1, 0] == -1
    meta, emb, index = formats.load_database(tmp_path)
    np.testing.assert_array_equal(meta, db['meta'])
    np.testing.assert_array_equal(emb, db['emb'])
    assert index[2] == 's2'


def test_mapping_and_compose_files_match_reference_layout(tmp_path):
    fix = helpers.load_fixture('retrieval_map_compose')
    ref_map = fix['map_train']                                   # [64,K,8] as flann_knn_worker produced it
    names = formats.chunk_patch_names('scene003')
    mapping = formats.mapping_to_dict(names, ref_map[..., :7].astype(np.int32), ref_map[..., 7])
    assert mapping[names[7]].shape == (ref_map.shape[1], 8) and mapping[names[7]].dtype == np.float32
    np.testing.assert_array_equal(mapping[names[7]], ref_map[7])
    formats.save_mapping(tmp_path / 'map_train.npy', mapping)
    back = formats.load_mapping(tmp_path / 'map_train.npy')
    meta, dist = formats.dict_to_mapping(back, names)
    np.testing.assert_array_equal(meta, ref_map[..., :7].astype(np.int32))
    np.testing.assert_array_equal(dist, ref_map[..., 7])
    vols = np.random.default_rng(0).random((4, 64, 64, 64)).astype(np.float32)
    formats.save_compose(tmp_path, 'scene003', vols)
    np.testing.assert_array_equal(formats.load_compose(tmp_path, 'scene003'), vols)


def test_combine_chunks_semantics():
    """N3: chunk names carry the position (dataset/patched_scene_dataset.py:153-174); later chunks overwrite, gaps keep trunc."""
    from rfuse import scene
    rng = np.random.default_rng(1)
    names = ['houseA__room1__0_0_0', 'houseA__room1__64_0_0', 'houseA__room1__64_64_128', 'houseB__room2__0_0_0']
    vols = [rng.random((64, 64, 64)).astype(np.float32) for _ in names]
    out = scene.combine_chunks(names, vols, '3DFront', trunc_val=0.1625)
    assert set(out) == {'houseA__room1', 'houseB__room2'}
    a = out['houseA__room1']
    assert a.shape == (128, 128, 192)
    np.testing.assert_array_equal(a[:64, :64, :64], vols[0])
    np.testing.assert_array_equal(a[64:128, 64:128, 128:192], vols[2])
    assert (a[:64, 64:, :] == 0.1625).all()
    # low-res inputs: positions divided by the scale factor (combine_inputs, :176-177)
    lo = scene.combine_chunks(names[:2], [v[::8, ::8, ::8] for v in vols[:2]], 'Matterport3D16', scale_factor=8, chunk_size=8, trunc_val=1.0)
    assert lo['houseA__room1'].shape == (16, 8, 8)
    one = scene.combine_chunks(['02691156__abc'], [vols[0]], 'ShapeNetV2')
    np.testing.assert_array_equal(one['02691156__abc'], vols[0])


def test_combine_chunks_matches_reference_golden():
    """N3 pinned: tests/golden/combine_chunks.npz holds the outputs of the reference's own PatchedSceneDataset.combine_chunks
    (targets at scale 1, low-res inputs at scale 64/input_chunk_size) on seeded synthetic chunks."""
    import helpers
    from rfuse import configs, scene, synthetic
    fix = helpers.load_fixture('combine_chunks')
    seed = int(fix['seed'])
    for tag in ('front', 'shapenet'):
        cfg = configs.get_config(str(fix[tag + '_cfg']))
        trunc_i, trunc_t = configs.truncations(cfg)
        names = [str(n) for n in fix[tag + '_names']]
        chunks = [synthetic.make_chunk(seed * 100 + i, cfg) for i in range(len(names))]
        ds_name = cfg['dataset_train']['dataset_name']
        s_in = cfg['dataset_train']['input_chunk_size']
        tgt = scene.combine_chunks(names, [c['target_raw'] for c in chunks], ds_name, 1, 64, trunc_t)
        inp = scene.combine_chunks(names, [c['input_raw'] for c in chunks], ds_name, 64 / s_in, s_in, trunc_i)
        assert sorted(tgt) == [str(k) for k in fix[tag + '_keys']]
        for k in tgt:
            assert tuple(fix['%s_target_%s_shape' % (tag, k)]) == tgt[k].shape and str(tgt[k].dtype) == str(fix['%s_target_%s_dtype' % (tag, k)])
            assert helpers.sha(tgt[k]) == str(fix['%s_target_%s_sha' % (tag, k)])
            assert tuple(fix['%s_input_%s_shape' % (tag, k)]) == inp[k].shape
            assert helpers.sha(inp[k]) == str(fix['%s_input_%s_sha' % (tag, k)])


def test_scene_driver_recomposition_matches_reference_golden():
    """N3: the inference loop's recomposition (trainer/train_refinement.py:166 -> PatchedSceneDataset.combine_retrievals(pred_shapes, 0)) pinned by
    the reference's own method on float16 predictions; ``split_scene`` names parse back to the positions it cut."""
    from rfuse import scene
    fix = helpers.load_fixture('combine_chunks')
    seed = int(fix['seed'])
    for tag in ('front', 'shapenet'):
        cfg = rf_configs.get_config(str(fix[tag + '_cfg']))
        _, trunc_t = rf_configs.truncations(cfg)
        names = [str(n) for n in fix[tag + '_names']]
        preds = np.stack([synthetic.uniform_stress_volume(seed * 1000 + i, (64, 64, 64), trunc_t) for i in range(len(names))])[:, None].astype(np.float16)
        got = scene.combine_predictions(names, preds, cfg['dataset_train']['dataset_name'], trunc_t)
        assert sorted(got) == [str(k) for k in fix[tag + '_keys']]
        for k, vol in got.items():
            assert tuple(vol.shape) == tuple(fix['%s_pred_%s_shape' % (tag, k)])
            assert helpers.sha(vol) == str(fix['%s_pred_%s_sha' % (tag, k)])
    low = np.arange(16 * 8 * 12, dtype=np.float32).reshape(16, 8, 12)
    names, chunks = scene.split_scene(low, 8, 'sceneZ', pad_value=-1.0)
    assert chunks.shape == (4, 8, 8, 8) and names[0] == 'sceneZ__room0__0_0_0' and names[-1] == 'sceneZ__room0__64_0_64'
    back = scene.combine_chunks(names, list(chunks), '3DFront', scale_factor=8, chunk_size=8, trunc_val=-1.0)['sceneZ__room0']
    assert np.array_equal(back[:16, :8, :12], low) and (back[:, :, 12:] == -1.0).all()


# ------------------------------------------------------------------------------------------------ N2: the device path writes the reference's files
@pytest.mark.gpu
def test_device_written_map_and_compose_files_match_reference_golden(gpu, tmp_path):
    """`formats.retrieval_mapping` / `compose_scene` (device top-2K + demotion + patch gather) -> map_train.npy / map_val.npy / compose/<scene>.npz, read back with
    load_mapping / load_compose, against tests/golden/retrieval_map_compose.npz: the reference's own flann_knn_worker (exact stand-in for FLANN) with
    ignore_patches_from_source True / False and create_retrieval_from_mapping (util/retrieval.py:87-100,145-164,233-248)."""
    from rfuse.database import PatchDatabase
    fix = helpers.load_fixture('retrieval_map_compose')
    cfg = rf_configs.get_config('C1')
    _, trunc_t = rf_configs.truncations(cfg)
    K, q_scene = cfg['K'], int(fix['q_scene'])
    db = synthetic.make_database(int(fix['seed']), cfg, int(fix['n_patches']))
    index = ['scene%03d' % i for i in range(db['n_scenes'])]
    scene = index[q_scene]
    pdb = PatchDatabase(db['emb'], db['meta'], db['volumes'], gpu)
    q = torch.from_numpy(fix['queries']).to(gpu)
    names = formats.chunk_patch_names(scene)
    for split, ignore in (('train', True), ('val', False)):
        formats.save_mapping(tmp_path / ('map_%s.npy' % split), formats.retrieval_mapping(pdb, q, [scene], K, index, ignore))
        back = formats.load_mapping(tmp_path / ('map_%s.npy' % split))
        assert list(back) == names and back[names[0]].dtype == np.float32 and back[names[0]].shape == (K, 8)
        rows = np.stack([back[n] for n in names])
        np.testing.assert_array_equal(rows[..., :7], fix['map_' + split][..., :7])              # scene index + box: exact
        np.testing.assert_allclose(rows[..., 7], fix['map_' + split][..., 7], rtol=0, atol=1e-6)   # squared L2 distance: the stand-in's float64 brute force
    # a scene that is NOT in the database index is not demoted even with ignore_patches_from_source (util/retrieval.py:95)
    other = formats.retrieval_mapping(pdb, q, ['elsewhere'], K, index, True)
    np.testing.assert_array_equal(np.stack(list(other.values()))[..., :7], fix['map_val'][..., :7])
    # compose from the files: train from the device-written map; val from the fixture's map with its injected sentinel hit (the idx < 0 branch)
    train_map = formats.load_mapping(tmp_path / 'map_train.npy')
    formats.save_compose(tmp_path, scene, formats.compose_scene(pdb, train_map, scene, K, trunc_t))
    got = formats.load_compose(tmp_path, scene)
    assert got.shape == (K, 64, 64, 64) and got.dtype == np.float32 and helpers.sha(got) == str(fix['compose_train_sha'])
    np.testing.assert_array_equal(got[:, ::4, ::4, ::4], fix['compose_train_sub'])
    val_s = {n: fix['map_val_sentinel'][i] for i, n in enumerate(names)}
    assert helpers.sha(formats.compose_scene(pdb, val_s, scene, K, trunc_t)) == str(fix['compose_val_sha'])
    # occupancy filter: dropped patches are absent from the mapping file and keep the truncation fill in the composed volumes
    keep = fix['patch_keep']
    masked = formats.retrieval_mapping(pdb, q, [scene], K, index, False, patch_mask=keep[None])
    assert list(masked) == [n for n, f in zip(names, keep) if f]
    masked_s = {n: val_s[n] for n in masked}
    assert helpers.sha(formats.compose_scene(pdb, masked_s, scene, K, trunc_t)) == str(fix['compose_masked_sha'])
    # the same non-overlapping grid through the ordered, distance-gated branch (:156): every box mean is the initial 100 > any distance -- the same volumes
    assert helpers.sha(formats.compose_scene(pdb, val_s, scene, K, trunc_t, no_overlap=False)) == str(fix['compose_val_sha'])
    assert helpers.sha(formats.compose_scene(pdb, masked_s, scene, K, trunc_t, no_overlap=False)) == str(fix['compose_masked_sha'])
    with pytest.raises(NotImplementedError, match='compose_overlap'):
        from rfuse import ops
        ops.gather_patches(pdb.volumes, torch.zeros((64, K, 7), dtype=torch.int32, device=gpu), 1, K, trunc_t, 1.0, 0.0, 1.0, 0, no_overlap=False)


@pytest.mark.gpu
@pytest.mark.parametrize('half_store', [True, False])
def test_device_compose_of_an_overlapping_patch_grid_matches_reference_golden(gpu, half_store):
    """create_retrieval_from_mapping with dataset.no_overlap False (util/retrieval.py:156: a patch overwrites its box only while the mean stored distance of the
    box is above its own; patch stride 8 < patch size 16, 343 patches in order) -- `formats.compose_scene(no_overlap=False)` / `ops.compose_overlap` against the
    reference's own output on tests/golden/compose_overlap.npz (937 of 1372 visits copy; sentinel hits; a run of equal distances)."""
    from rfuse.database import PatchDatabase
    fix = helpers.load_fixture('compose_overlap')
    cfg = rf_configs.get_config('C1')
    _, trunc_t = rf_configs.truncations(cfg)
    K, q_scene = cfg['K'], int(fix['q_scene'])
    db = synthetic.make_database(int(fix['seed']), cfg, int(fix['n_db_patches']))
    assert helpers.sha(db['meta'], db['emb'], db['volumes']) == str(fix['db_sha'])
    pdb = PatchDatabase(db['emb'], db['meta'], db['volumes'], gpu, half_store=half_store)
    scene = 'scene%03d' % q_scene
    ext = formats.scene_patch_extents((64, 64, 64), 16, 8, 8)
    np.testing.assert_array_equal(ext, fix['extents'])                                  # the reference's get_extents_for_size (dataset/scene.py:152-160)
    names = [formats.patch_name(scene, e) for e in ext]
    mapping = {n: fix['mapping'][i] for i, n in enumerate(names)}
    got = formats.compose_scene(pdb, mapping, scene, K, trunc_t, no_overlap=False, stride=8)
    assert got.shape == (K, 64, 64, 64) and got.dtype == np.float32
    np.testing.assert_array_equal(got[:, ::4, ::4, ::4], fix['composed_sub'])
    assert helpers.sha(got) == str(fix['composed_sha'])
    assert 0 < int(fix['taken']) < int(fix['visited'])                                  # both outcomes of the comparison occur
    # patches the lookup does not hold are never visited: dropping the LAST patches leaves their boxes to the earlier ones
    fewer = {n: mapping[n] for n in names[:200]}
    part = formats.compose_scene(pdb, fewer, scene, K, trunc_t, no_overlap=False, stride=8)
    from oracle import refpath
    want = refpath.compose_retrieval_overlap(fix['mapping'][:200], fix['boxes'][:200], db['volumes'], K, trunc_t)
    np.testing.assert_array_equal(part, want)


@pytest.mark.gpu
def test_retrievals_to_disk_from_chunks_matches_oracle(gpu, tmp_path):
    """The whole offline pipeline from raw chunks on the device -- create_dictionary (fenc_target on the scene chunks -> database.npy / index.json) ->
    retrievals_to_disk('map') (fenc_input -> top-2K -> demotion -> map files) -> retrievals_to_disk('compose') -- against the oracle's restatement of the
    same stages on the database rows that were written."""
    from rfuse.engine import RefinementEngine
    cfg = rf_configs.get_config('C1')
    trunc_i, trunc_t = rf_configs.truncations(cfg)
    K = cfg['K']
    eng = RefinementEngine(cfg, gpu, None)
    sds = {n: helpers.seeded_sd({k: tuple(v.shape) for k, v in m.state_dict().items()}, 40 + i) for i, (n, m) in enumerate(eng.modules().items())}
    eng.load_state_dicts(sds)
    fenc_target, sd_t = _target_sd(cfg, 6)
    eng.fenc_target.load_state_dict(sd_t)
    scenes = [synthetic.make_chunk(7000 + i, cfg) for i in range(6)]
    index = ['shape%02d' % i for i in range(6)]
    volumes = np.stack([c['target_raw'] for c in scenes])
    eng.database = formats.create_dictionary(cfg, eng.fenc_target, volumes, index, tmp_path / 'tree', gpu)
    meta, emb, index_back = formats.load_database(tmp_path / 'tree')
    assert index_back == index and emb.shape == (6 * 64 + 1, 64) and meta[-1, 0] == -1
    # train = the database's own scenes (every query has same-scene rows to demote), val = two new scenes, one with an occupancy filter
    val_chunks = [synthetic.make_chunk(7100 + i, cfg) for i in range(2)]
    mask = np.ones((2, 64), dtype=bool)
    mask[1, ::3] = False
    splits = {'train': (index, np.stack([c['input_raw'] for c in scenes])),
              'val': (['new00', 'new01'], np.stack([c['input_raw'] for c in val_chunks]), mask)}
    out = tmp_path / 'retrievals'
    written = formats.retrievals_to_disk('map', eng, out, splits, index=index, batch=4)
    assert [p.name for p in written] == ['map_train.npy', 'map_val.npy']
    written = formats.retrievals_to_disk('compose', eng, out, splits)
    assert len(written) == 8 and all(p.exists() for p in written)
    for split, (names, chunks, *rest) in splits.items():
        mapping = formats.load_mapping(out / ('map_%s.npy' % split))
        keep = rest[0] if rest else np.ones((len(names), 64), dtype=bool)
        assert len(mapping) == int(keep.sum())
        for c, scene in enumerate(names):
            with torch.no_grad():
                q = refpath.embed_queries(refpath.extract_query_windows(chunks[c], cfg, trunc_i), sds['fenc_input'], cfg).numpy()
            idx, dist = refpath.knn_exact(q, emb, 2 * K)
            qs = index.index(scene) if (split == 'train' and scene in index) else -1
            want = refpath.demote_same_scene(refpath.mapping_rows(idx, dist, meta), np.full(64, qs), K)
            pn = formats.chunk_patch_names(scene)
            got = np.stack([mapping[n] if keep[c, p] else want[p] for p, n in enumerate(pn)])
            # neighbour lists: the device's fp32 query embeddings differ from the oracle's by ~1e-7, so a near-tie may swap -- compare by distance
            np.testing.assert_allclose(got[..., 7], want[..., 7], rtol=0, atol=5e-6)
            same = (got[..., :7] == want[..., :7]).all(axis=-1)
            assert same.mean() >= 0.98
            vols = formats.load_compose(out, scene)
            ref = refpath.compose_retrieval(got, volumes, K, trunc_t, patch_keep=keep[c])
            np.testing.assert_array_equal(vols, ref)


@pytest.mark.gpu
def test_retrievals_to_disk_target_features_and_split_truncations(gpu, tmp_path):
    """retrievals_to_disk with use_target_for_feats (reference util/retrieval.py:230-231: queries = fenc_target on the TARGET windows) and a val split whose
    truncation differs from the train split's (:148,159: fill with the split's truncation, scale database patches by split / train)."""
    from rfuse.engine import RefinementEngine
    cfg = rf_configs.get_config('C1')
    _, trunc_t = rf_configs.truncations(cfg)
    K = cfg['K']
    eng = RefinementEngine(cfg, gpu, None)
    sds = {n: helpers.seeded_sd({k: tuple(v.shape) for k, v in m.state_dict().items()}, 60 + i) for i, (n, m) in enumerate(eng.modules().items())}
    eng.load_state_dicts(sds)
    fenc_target, sd_t = _target_sd(cfg, 9)
    eng.fenc_target.load_state_dict(sd_t)
    scenes = [synthetic.make_chunk(7300 + i, cfg) for i in range(4)]
    index = ['shape%02d' % i for i in range(4)]
    volumes = np.stack([c['target_raw'] for c in scenes])
    eng.database = formats.create_dictionary(cfg, eng.fenc_target, volumes, index, tmp_path / 'tree', gpu)
    meta, emb, _ = formats.load_database(tmp_path / 'tree')
    val = [synthetic.make_chunk(7400 + i, cfg) for i in range(2)]
    mask = np.ones((2, 64), dtype=bool)
    mask[0, 1::4] = False
    splits = {'train': (index, np.stack([c['input_raw'] for c in scenes])), 'val': (['new00', 'new01'], np.stack([c['input_raw'] for c in val]), mask)}
    targets = {'train': volumes, 'val': np.stack([c['target_raw'] for c in val])}
    out = tmp_path / 'retrievals'
    with pytest.raises(ValueError):
        formats.retrievals_to_disk('map', eng, out, splits, index=index, use_target_for_feats=True)
    formats.retrievals_to_disk('map', eng, out, splits, index=index, batch=3, use_target_for_feats=True, fenc_target=eng.fenc_target, target_chunks=targets)
    # a train chunk queried with its own target features: its own row is THE nearest (distance ~0) and is demoted behind the other scenes' rows of the top-2K
    m_train = formats.load_mapping(out / 'map_train.npy')
    for c, scene in enumerate(index):
        for nm in formats.chunk_patch_names(scene):
            own = m_train[nm][:, 0] == c
            assert not own[0] or own.all(), 'a same-scene row ahead of another scene\'s'
            assert (np.diff(own.astype(np.int8)) >= 0).all(), 'same-scene rows are not at the end of the list'
    # a val chunk that IS a database scene finds itself at distance ~0 (no demotion on val)
    splits2 = {'val': (['again'], splits['train'][1][:1])}
    formats.retrievals_to_disk('map', eng, tmp_path / 'r2', splits2, index=index, use_target_for_feats=True, fenc_target=eng.fenc_target, target_chunks={'val': volumes[:1]})
    m2 = formats.load_mapping(tmp_path / 'r2' / 'map_val.npy')
    for nm in formats.chunk_patch_names('again'):
        assert m2[nm][0, 0] == 0 and m2[nm][0, 7] <= 1e-5
    # compose with a val truncation of half the train truncation: fill = the split's truncation, database patches and sentinel hits scaled by 1/2
    formats.retrievals_to_disk('compose', eng, out, splits, truncations_by_split={'train': trunc_t, 'val': trunc_t / 2})
    m_val = formats.load_mapping(out / 'map_val.npy')
    assert len(m_val) == int(mask.sum())
    for c, scene in enumerate(splits['val'][0]):
        pn = formats.chunk_patch_names(scene)
        got = np.stack([m_val[n] if mask[c, p] else np.zeros((K, 8), dtype=np.float32) for p, n in enumerate(pn)])
        want = refpath.compose_retrieval(got, volumes, K, trunc_t / 2, trunc_ratio=0.5, patch_keep=mask[c])
        np.testing.assert_array_equal(formats.load_compose(out, scene), want)
    np.testing.assert_array_equal(formats.load_compose(out, index[0]),
                                  refpath.compose_retrieval(np.stack([m_train[n] for n in formats.chunk_patch_names(index[0])]), volumes, K, trunc_t, patch_keep=np.ones(64, dtype=bool)))
