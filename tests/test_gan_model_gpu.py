"""The GAN feature-augmentation model (BASELINE config 5) on the GPU against the reference's outputs
(tests/golden/gan_model.npz: augment/gan.py run with the same seeded weights): generator forward through the HIP gather / pool /
layout kernels, the six discriminator / generator losses in the reference's call order (spectral-norm state advances per call),
and gradients of the generator loss through the HIP adjoints."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'gan_model.npz'))
OBJ = ['__background__'] + ['obj%d' % i for i in range(1, 9)]
REL = ['__background__'] + ['rel%d' % i for i in range(1, 5)]
DEV = 'cuda:0'


def t(name):
    return torch.from_numpy(G[name]).to(DEV)


@pytest.fixture()
def gan():
    from sgg_amd.feature_gan import GAN
    m = GAN(OBJ, REL, embed_dim=12, hidden_dim=16, n_ch=32, pool_sz=7, fmap_sz=38, n_layers_G=3, device=DEV).to(DEV)
    missing, unexpected = m.load_state_dict({k[2:]: torch.from_numpy(G[k]) for k in G.files if k.startswith('w_')}, strict=False)
    assert not unexpected and all('num_batches' in k for k in missing)
    m.train()
    return m


def test_generator_forward_losses_and_gradients_match_reference(gan):
    fmap = gan(t('in_objs'), t('in_boxes'), t('in_rels'))
    assert fmap.shape == (3, 32, 38, 38) and fmap.min() >= 0
    torch.testing.assert_close(fmap.cpu(), torch.from_numpy(G['fmap_fake']), atol=3e-4, rtol=1e-3)
    out = {}
    out.update(gan.loss(t('f_real_o'), t('f_fake_o'), is_nodes=True, updateD=True, labels_fake=t('lab_o')))
    out.update(gan.loss(t('f_real_r'), t('f_fake_r'), is_nodes=False, updateD=True, labels_fake=t('lab_r_fake'), labels_real=t('lab_r_real')))
    out.update(gan.loss(t('fmap_real'), fmap, updateD=True, is_fmaps=True))
    out.update(gan.loss(features_fake=t('f_fake_o'), is_nodes=True, labels_fake=t('lab_o')))
    out.update(gan.loss(features_fake=t('f_fake_r'), is_nodes=False, labels_fake=t('lab_r_fake')))
    g_fmap = gan.loss(features_fake=fmap, is_fmaps=True)
    out.update(g_fmap)
    assert sorted(out) == ['D_fmap', 'D_obj', 'D_rel', 'G_fmap', 'G_obj', 'G_rel']
    for k, v in out.items():
        assert abs(float(v) - float(G['loss_' + k])) <= 2e-4 * max(1.0, abs(float(G['loss_' + k]))), (k, float(v), float(G['loss_' + k]))
    # the generator loss reaches the embeddings through refinement network, layout kernel, 1x1 / 3x3 convs and graph convolutions
    names = ['G_obj_embed.weight', 'G_proj.weight', 'G_refine.refinement_modules.0.net.0.weight', 'G_gcn.gconvs.0.net1.0.weight']
    params = dict(gan.named_parameters())
    grads = torch.autograd.grad(g_fmap['G_fmap'], [params[n] for n in names])
    for n, gr in zip(names, grads):
        ref = torch.from_numpy(G['grad_' + n])
        assert (gr.cpu() - ref).abs().max() <= 2e-3 * ref.abs().max() + 1e-7, (n, float((gr.cpu() - ref).abs().max()), float(ref.abs().max()))


def test_discriminator_update_does_not_reach_the_generator(gan):
    fmap = gan(t('in_objs'), t('in_boxes'), t('in_rels'))
    loss = gan.loss(t('fmap_real'), fmap, updateD=True, is_fmaps=True)['D_fmap']
    loss.backward()
    assert all(p.grad is None for n, p in gan.named_parameters() if n.startswith('G_'))
    assert all(p.grad is not None for n, p in gan.named_parameters() if n.startswith('D_global') and p.requires_grad)


def test_gan_training_iteration_with_the_sgg_model():
    """main.py:124-194 end to end at the real size (592x592 -> 38x38 x 512 feature maps, full-width GAN): the SGG model's training
    forward, then sgg_amd.feature_gan.gan_train_step -- generated feature maps -> the model's RoIAlign (with its HIP adjoint into
    the maps) -> the model's head -> G update (+ SGG update through the reconstruction losses) -> D update."""
    import sgg_amd
    from sgg_amd.feature_gan import GAN, gan_train_step
    from sgg_amd.synthetic import SyntheticData, init_weights, synthetic_batch
    torch.manual_seed(3)
    data = SyntheticData()
    model = init_weights(sgg_amd.RelModelStanford(data, mode='sgcls')).to(DEV)
    model.set_compute_dtype(torch.float32)
    for n, p in model.named_parameters():
        if n.startswith('detector.'):
            p.requires_grad = False
    gan = GAN(data.ind_to_classes, data.ind_to_predicates, n_ch=model.edge_dim, pool_sz=model.pool_sz, fmap_sz=model.fmap_sz,
              n_layers_G=2, device=DEV).to(DEV)
    assert model.fmap_sz == 38 and gan.obj_dim == model.edge_dim * model.pool_sz ** 2
    batch = synthetic_batch(B=2, n_boxes=5, n_fg=4, seed=8)
    dev_batch = tuple(t_.to(DEV) if isinstance(t_, torch.Tensor) else t_ for t_ in batch)
    model.train()
    res = model([dev_batch])
    gt_boxes, gt_objects, gt_rels = dev_batch[3].clone(), dev_batch[4].clone(), dev_batch[5].clone()
    sgg_params = [p for p in model.parameters() if p.requires_grad]
    optimizer = torch.optim.SGD(sgg_params, lr=1e-3, momentum=0.9)
    G_optimizer = torch.optim.Adam([p for n, p in gan.named_parameters() if n.startswith('G_')], lr=1e-4, betas=(0.5, 0.999))
    D_optimizer = torch.optim.Adam([p for n, p in gan.named_parameters() if n.startswith('D_')], lr=1e-4, betas=(0.5, 0.999))
    before = {n: p.detach().clone() for n, p in list(gan.named_parameters()) + [('sgg.' + n, p) for n, p in model.named_parameters() if p.requires_grad]}
    losses = gan_train_step(model, gan, res, gt_boxes, gt_objects, gt_rels, optimizer, G_optimizer, D_optimizer, ganw=1.0)
    assert sorted(losses) == ['D_fmap', 'D_obj', 'D_rel', 'G_fmap', 'G_obj', 'G_rel', 'rec']
    assert all(torch.isfinite(v) for v in losses.values())
    after = dict(list(gan.named_parameters()) + [('sgg.' + n, p) for n, p in model.named_parameters() if p.requires_grad])
    moved = {n for n in before if not torch.equal(before[n], after[n].detach())}
    for must in ('G_obj_embed.weight', 'G_gcn.gconvs.0.net1.0.weight', 'G_node.0.weight', 'G_proj.weight',
                 'G_refine.refinement_modules.2.net.3.weight', 'G_refine.output_conv.0.weight',       # generator: reached through RoIAlign's adjoint
                 'D_nodes.0.weight_orig', 'D_edges.6.weight_orig', 'D_global.15.weight_orig',         # discriminators
                 'sgg.rel_fc.weight', 'sgg.roi_fmap.1.0.weight', 'sgg.edge_gru.weight_ih'):          # SGG model: reconstruction losses
        assert must in moved, must
    assert all(torch.isfinite(p).all() for p in after.values())
    # a second iteration on the updated weights (momentum / Adam state, spectral-norm vectors advanced) stays finite
    res = model([dev_batch])
    losses2 = gan_train_step(model, gan, res, gt_boxes, gt_objects, gt_rels, optimizer, G_optimizer, D_optimizer, ganw=1.0,
                             ganlosses=('D', 'G'))
    assert sorted(losses2) == ['D_fmap', 'D_obj', 'D_rel', 'G_fmap', 'G_obj', 'G_rel'] and all(torch.isfinite(v) for v in losses2.values())


def test_gqa_configuration_iteration_resnet50_backbone_and_gan():
    """BASELINE configs[4] in one process: the GQA form of the model (backbone='resnet50': 1333-pixel images, 256 x 21 x 21 'pool'-level
    feature map, TwoMLPHead RoI heads) with the GAN sized from it (n_ch 256, fmap_sz 21: D_global then skips its first pooling), one
    SGG forward + one GAN iteration of main.py:124-194."""
    import sgg_amd
    from sgg_amd.feature_gan import GAN, gan_train_step
    from sgg_amd.synthetic import SyntheticData, init_weights, synthetic_batch
    torch.manual_seed(5)
    data = SyntheticData()
    model = init_weights(sgg_amd.RelModelStanford(data, mode='sgcls', backbone='resnet50')).to(DEV)
    for n, p in model.named_parameters():
        if n.startswith('detector.'):
            p.requires_grad = False
    gan = GAN(data.ind_to_classes, data.ind_to_predicates, n_ch=model.edge_dim, pool_sz=model.pool_sz, fmap_sz=model.fmap_sz,
              n_layers_G=2, device=DEV).to(DEV)
    assert (model.edge_dim, model.fmap_sz, model.obj_dim) == (256, 21, 1024) and isinstance(gan.D_global[4], torch.nn.Identity)
    batch = synthetic_batch(B=1, S=1333, n_boxes=5, n_fg=4, seed=12)
    dev_batch = tuple(t_.to(DEV) if isinstance(t_, torch.Tensor) else t_ for t_ in batch)
    model.train()
    res = model([dev_batch])
    assert tuple(res.fmap.shape) == (1, 256, 21, 21) and res.node_feat.shape[1:] == (256, 7, 7)
    opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=1e-3, momentum=0.9)
    G_opt = torch.optim.Adam([p for n, p in gan.named_parameters() if n.startswith('G_')], lr=1e-4)
    D_opt = torch.optim.Adam([p for n, p in gan.named_parameters() if n.startswith('D_')], lr=1e-4)
    g0 = gan.G_proj.weight.detach().clone()
    w0 = model.roi_fmap.fc6.weight.detach().clone()
    losses = gan_train_step(model, gan, res, dev_batch[3].clone(), dev_batch[4].clone(), dev_batch[5].clone(), opt, G_opt, D_opt)
    assert sorted(losses) == ['D_fmap', 'D_obj', 'D_rel', 'G_fmap', 'G_obj', 'G_rel', 'rec'] and all(torch.isfinite(v) for v in losses.values())
    assert not torch.equal(g0, gan.G_proj.weight.detach()) and not torch.equal(w0, model.roi_fmap.fc6.weight.detach())
