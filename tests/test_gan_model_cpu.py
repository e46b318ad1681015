"""Host-side parts of the GAN feature-augmentation model against the reference's own outputs
(tests/golden/gan_model.npz, made by running augment/gan.py): the dummy-node bookkeeping and the parameter naming."""
import os

import numpy as np
import pytest
import torch

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'gan_model.npz'))
OBJ = ['__background__'] + ['obj%d' % i for i in range(1, 9)]
REL = ['__background__'] + ['rel%d' % i for i in range(1, 5)]


def small_gan(device='cpu'):
    from sgg_amd.feature_gan import GAN
    return GAN(OBJ, REL, embed_dim=12, hidden_dim=16, n_ch=32, pool_sz=7, fmap_sz=38, n_layers_G=3, device=device)


def test_dummy_nodes_equal_reference():
    from sgg_amd.feature_gan import dummy_nodes
    objs, boxes, rels = (torch.from_numpy(G[k]) for k in ('in_objs', 'in_boxes', 'in_rels'))
    d_objs, d_boxes, d_rels = dummy_nodes(objs, boxes, rels)
    assert torch.equal(d_objs, torch.from_numpy(G['dummy_objs']))
    assert torch.equal(d_boxes, torch.from_numpy(G['dummy_boxes']))
    assert torch.equal(d_rels, torch.from_numpy(G['dummy_rels']))
    # one image: the batch offsets vanish
    one = rels[rels[:, 0] == 0]
    o1, b1, r1 = dummy_nodes(objs[objs[:, 0] == 0], boxes[objs[:, 0] == 0], one)
    n = int((objs[:, 0] == 0).sum())
    assert o1.shape[0] == n + 1 and o1[-1].tolist() == [0, 0] and b1[-1].tolist() == [0, 0, 1, 1]
    assert r1.shape[0] == len(one) + 2 * n and torch.equal(r1[:len(one)], one)
    assert r1[len(one):len(one) + n, 1].eq(n).all() and r1[len(one) + n:, 2].eq(n).all() and r1[len(one):, 3].eq(0).all()
    # an image without relations cannot be indexed by the reference either
    with pytest.raises(IndexError):
        dummy_nodes(objs, boxes, rels[rels[:, 0] != 1])


def test_state_dict_names_and_shapes_are_the_references():
    gan = small_gan()
    mine = {k: tuple(v.shape) for k, v in gan.state_dict().items() if 'num_batches' not in k}
    ref = {k[2:]: G[k].shape for k in G.files if k.startswith('w_')}
    assert mine == ref
    gan.load_state_dict({k: torch.from_numpy(G['w_' + k]) for k in ref}, strict=False)
    # option switches keep the Sequential indices (Identity placeholders), so checkpoints of either form load by name
    from sgg_amd.feature_gan import GAN
    big = GAN(OBJ, REL, embed_dim=12, hidden_dim=16, n_ch=32, n_layers_G=3, largeD=True, SN=False, device='cpu')
    keys = set(big.state_dict())
    assert {'D_global.2.weight', 'D_global.7.weight', 'D_global.12.weight', 'D_global.15.weight'} <= keys
    assert not any('weight_orig' in k for k in keys)
    with pytest.raises(ValueError):
        GAN(OBJ, REL, embed_dim=12, hidden_dim=16, n_ch=32, n_layers_G=3, init_embed=True, device='cpu')


def test_loss_sides_can_be_switched_off():
    from sgg_amd.feature_gan import GAN
    gan = GAN(OBJ, REL, embed_dim=12, hidden_dim=16, n_ch=32, n_layers_G=3, losses=('rec',), device='cpu')
    assert gan.loss(updateD=True) == {} and gan.loss(updateD=False) == {}


def test_real_features_come_out_of_features_hdf5():
    """vis_cond (augment/gan.py:63-64,193-199): the file extract_features.py writes -- one dataset per class name, rows of n_ch x P x P
    features -- read by sgg_amd.hdf5_lite; one row per object, drawn with numpy's global RNG exactly as the reference draws it."""
    from sgg_amd.feature_gan import GAN
    h5 = os.path.join(os.path.dirname(__file__), 'golden', 'h5')
    exp = np.load(os.path.join(h5, 'expected.npz'))
    gan = GAN(OBJ, REL, embed_dim=12, hidden_dim=16, n_ch=32, n_layers_G=3, vis_cond=os.path.join(h5, 'features_small.hdf5'), device='cpu')
    assert gan.G_proj.in_channels == 16 + 32 and gan.h5_data is not None           # real features are concatenated in front of the fake ones
    gan.obj_classes = ['__background__', 'airplane', 'zebra', 'traffic light']
    gan.n_ch, gan.pool_sz = 6, 3                                                   # the fixture's feature shape
    classes = torch.tensor([3, 1, 2, 3, 3, 1])
    np.random.seed(11)
    got = gan.sample_real_features(classes)
    np.random.seed(11)
    for row, cls in zip(got, classes.tolist()):
        rows = exp['feat/' + gan.obj_classes[cls]]
        np.testing.assert_array_equal(row.numpy(), rows[np.random.permutation(rows.shape[0])[0]])
    with pytest.raises(AssertionError):
        gan.sample_real_features(torch.tensor([0]))
