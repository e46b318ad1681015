"""Input side: sgg_amd.visual_genome against vectors produced by the reference's own dataloaders/visual_genome.py
(load_graphs with h5py.File replaced by an in-memory mapping, load_info, filter_dups, VG.__getitem__ on synthetic PIL images) and
dataloaders/image_transforms.py SquarePad -- tests/golden/vg_loader.npz."""
import os

import numpy as np
import pytest

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'vg_loader.npz'))
CASES = [
    dict(mode='train'), dict(mode='test'), dict(mode='val', num_val_im=6), dict(mode='train', num_val_im=6),
    dict(mode='train', filter_empty_rels=False), dict(mode='train', num_im=10),
    dict(mode='train', min_graph_size=3, max_graph_size=6), dict(mode='train', filter_non_overlap=True),
    dict(mode='test', training_triplets='set'), dict(mode='test', training_triplets='set', filter_zeroshots=False),
    dict(mode='test', training_triplets='counts', n_shots=10), dict(mode='test', training_triplets='counts', n_shots=100),
]


def tables():
    return {k[3:]: G[k] for k in G.files if k.startswith('h5_')}


@pytest.mark.parametrize('ci', range(len(CASES)))
def test_load_graphs_equals_reference(ci):
    from sgg_amd.visual_genome import load_graphs
    kw = dict(CASES[ci])
    counts = dict(zip(G['triplet_keys'].tolist(), G['triplet_counts'].tolist()))
    if kw.get('training_triplets') == 'set':
        kw['training_triplets'] = set(counts)
    elif kw.get('training_triplets') == 'counts':
        kw['training_triplets'] = counts
    mask, boxes, classes, rels = load_graphs(tables(), **kw)
    np.testing.assert_array_equal(mask, G['c%d_mask' % ci])
    assert len(boxes) == int(G['c%d_n' % ci]) and len(boxes) > 0
    for i in range(len(boxes)):
        np.testing.assert_array_equal(boxes[i], G['c%d_boxes_%d' % (ci, i)])
        np.testing.assert_array_equal(classes[i], G['c%d_classes_%d' % (ci, i)])
        np.testing.assert_array_equal(rels[i], G['c%d_rels_%d' % (ci, i)])
        assert boxes[i].dtype == G['c%d_boxes_%d' % (ci, i)].dtype


def test_load_graphs_rejects_unknown_mode_and_missing_files(tmp_path):
    """.h5 paths are opened by the package's own reader (sgg_amd.hdf5_lite, no h5py): a missing file and a file that is not HDF5
    are both errors, not empty splits."""
    from sgg_amd.visual_genome import load_graphs
    with pytest.raises(ValueError):
        load_graphs(tables(), mode='trainval')
    with pytest.raises(OSError):
        load_graphs('/nonexistent/VG-SGG.h5')
    bad = tmp_path / 'not_hdf5.h5'
    bad.write_bytes(b'plain text, no superblock' * 64)
    with pytest.raises((OSError, ValueError)):
        load_graphs(str(bad))


def test_info_dups_filenames(tmp_path):
    from sgg_amd import visual_genome as V
    c, p = V.load_info({'label_to_idx': {'dog': 2, 'cat': 1, 'tree': 3}, 'predicate_to_idx': {'on': 1, 'near': 3, 'has': 2}})
    assert c == G['info_classes'].tolist() and p == G['info_predicates'].tolist()
    np.testing.assert_array_equal(V.filter_dups(G['dups_in'], random_edge=False), G['dups_out'])
    np.random.seed(0)
    r = V.filter_dups(G['dups_in'])
    assert sorted(map(tuple, r[:, :2])) == sorted(set(map(tuple, G['dups_in'][:, :2])))
    for i in (7, 1592, 9):
        (tmp_path / ('%d.jpg' % i)).write_bytes(b'x')
    data = [{'image_id': 7}, {'image_id': 1592}, {'image_id': 8}, {'image_id': 9}]
    assert V.load_image_filenames(data, str(tmp_path), expected=2) == ['7.jpg', '9.jpg']          # corrupted / missing ones dropped
    with pytest.raises(AssertionError):
        V.load_image_filenames(data, str(tmp_path))                                                  # the reference's count check


def test_entry_geometry_and_square_pad_equal_reference():
    from sgg_amd.visual_genome import entry_geometry, square_pad_u8
    for k in range(int(G['n_entries'])):
        pre = 'g%d_' % k
        w, h = G[pre + 'wh']
        flip = bool(G[pre + 'flip'])
        boxes, im_size, scale = entry_geometry(int(w), int(h), G[pre + 'gt_in'], bool(G[pre + 'torch_detector']), flip)
        np.testing.assert_array_equal(boxes, G[pre + 'gt_out'])
        np.testing.assert_array_equal(np.array(im_size, dtype=np.float64), G[pre + 'im_size'])
        assert scale == float(G[pre + 'scale'])
        img = G[pre + 'img_in'][:, ::-1] if flip else G[pre + 'img_in']
        np.testing.assert_array_equal(square_pad_u8(img), G[pre + 'img_out'])


def test_dataset_items_feed_the_blob():
    """VG built from tables -> entries -> vg_collate: the tuple layout the model reads (dataloaders/blob.py:244-249)."""
    from sgg_amd.blob import vg_collate
    from sgg_amd.visual_genome import VG
    t = tables()
    n_img = len(t['split'])
    rng = np.random.RandomState(0)
    fake = {}

    def decode(path):
        return fake.setdefault(path, rng.randint(0, 255, size=(48, 64, 3)).astype(np.uint8))
    info = {'label_to_idx': {'c%d' % i: i for i in range(1, 12)}, 'predicate_to_idx': {'p%d' % i: i for i in range(1, 9)}}
    ds = VG('test', t, info, ['%d.jpg' % i for i in range(n_img)], num_val_im=0, decode=decode)
    assert len(ds) == int(G['c1_n']) and ds.num_classes == 12 and ds.num_predicates == 9
    assert ds.filenames == ['%d.jpg' % i for i in np.where(G['c1_mask'])[0]]
    assert sum(ds.triplet_counts.values()) == sum(len(r) for r in ds.relationships)
    e = ds[0]
    assert e['img'].dtype == np.uint8 and e['img'].shape == (48, 64, 3) and not e['flipped'] and e['scale'] == 1.0
    np.testing.assert_allclose(e['gt_boxes'], np.minimum(ds.gt_boxes[0] / (1024 / 64.), [[64, 48, 64, 48]]))
    blob = vg_collate([ds[0], ds[1]], mode='rel', is_train=False)
    imgs, im_sizes, _, gt_boxes, gt_classes, gt_rels, _, fns = blob[0]
    assert len(imgs) == 2 and im_sizes.shape == (2, 3) and gt_classes.shape[1] == 2 and gt_rels.shape[1] == 4
    assert gt_boxes.shape[0] == gt_classes.shape[0] == len(ds.gt_classes[0]) + len(ds.gt_classes[1])
    train = VG('train', t, info, ['%d.jpg' % i for i in range(n_img)], num_val_im=6, decode=decode)
    assert len(train) == int(np.sum(train.split_mask)) > 0 and train.filter_duplicate_rels
    # the statistics the scene-graph perturbations consult (visual_genome.py:211-227): every (subject, predicate) -> {object: count of
    # the whole triplet}, every (predicate, object) -> {subject: count}; together they hold every training triplet exactly once
    assert not hasattr(ds, 'subj_pred_pairs')                     # train mode only
    n_sp = sum(len(v) for v in train.subj_pred_pairs.values())
    assert n_sp == sum(len(v) for v in train.pred_obj_pairs.values()) == len(train.triplet_counts) > 0
    for pair, objs in train.subj_pred_pairs.items():
        for o, count in objs.items():
            s_, p_ = pair.split('_')
            assert train.triplet_counts['%s_%s_%s' % (s_, p_, o)] == count == train.pred_obj_pairs['%s_%s' % (p_, o)][int(s_)]
    np.random.seed(1)
    e = train[0]
    assert len(set(map(tuple, e['gt_relations'][:, :2]))) == len(e['gt_relations'])     # duplicates filtered in training
