"""Generate the golden vectors under tests/golden/*.npz by RUNNING THE REFERENCE'S OWN CODE.

Run in the build container only (needs /root/reference):
    python oracle/build_ref.py && python tests/golden/make_golden.py

Each .npz holds inputs, the random weights used, and the reference's outputs.  Nothing of the
reference's source text is stored -- only data.  What is called, per file:

  raster.npz       lib/draw_rectangles/draw_rectangles.pyx:12  draw_union_boxes (re-cythonized)
  union_feats.npz  lib/get_union_boxes.py:17,63               UnionBoxesAndFeats(...).forward (eval)
  gru.npz          torch.nn.GRUCell as instantiated at sgg_models/rel_model_stanford.py:36-37
  message_pass.npz sgg_models/rel_model_stanford.py:48        RelModelStanford.message_pass
  predict.npz      sgg_models/rel_model_stanford.py:97        RelModelStanford.predict
  pairs.npz        sgg_models/rel_model_base.py:143 get_rel_inds ; lib/proposal_assignments_gtbox.py:7
  gan_ops.npz      augment/layout.py:33 boxes_to_layout (+ autograd gradient wrt the object features), augment/graphconv.py:17
                   GraphTripleConv.forward (seeded weights) -- the gather / scatter steps of the GAN generator
  rel_assign.npz   lib/rel_assignments.py:12 rel_assignments (sgdet training), numpy global RNG seeded per case
  eval_tail.npz    lib/surgery.py:17 filter_dets + the softmax/sort lines rel_model_stanford.py:187-204
  losses.npz       lib/losses.py:5,73
  recall.npz       lib/sgg_eval.py:14 BasicSceneGraphEvaluator (GC / no-GC / per-triplet, all modes), :420 mean recall, :481 eval_entry
  vg_loader.npz    dataloaders/visual_genome.py:516 load_graphs (h5py.File replaced by an in-memory mapping with the VG-SGG.h5 keys),
                   :662 load_info, :743 filter_dups, :377 VG.__getitem__ box / size arithmetic (synthetic PIL images),
                   dataloaders/image_transforms.py:8 SquarePad
  gan_model.npz    augment/gan.py:17 GAN (reduced widths, seeded weights; GloVe loader replaced by random vectors): dummy_nodes :262,
                   forward :174 (train-mode BatchNorm), loss :211 for the six D / G cases in a fixed order (spectral-norm power
                   iterations advance per call), gradients of the G_fmap loss wrt three generator parameters
  sg_perturb.npz   augment/sg_perturb.py:7 SceneGraphPerturb.perturb ('rand', 'neigh', 'graphn' with and without the semantic hop; numpy and
                   torch generators seeded per case), on synthetic graphs and synthetic training statistics
  freq_bias.npz    lib/get_dataset_counts.py:10 get_counts, lib/sparse_targets.py:7 FrequencyBias, and the use_bias block
                   sgg_models/rel_model_stanford.py:159-177 (executed on the reference's FrequencyBias module)
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

dr = ref_import.install()
from lib.get_union_boxes import UnionBoxesAndFeats  # noqa: E402
from lib.losses import edge_losses, node_losses  # noqa: E402
from lib.proposal_assignments_gtbox import proposal_assignments_gtbox  # noqa: E402
from lib.surgery import filter_dets  # noqa: E402
from sgg_models.rel_model_stanford import RelModelStanford  # noqa: E402


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print('%-18s %7.1f kB  %d arrays' % (name, os.path.getsize(path) / 1e3, len(out)))


def sd(module, prefix=''):
    return {prefix + k: v.detach().clone() for k, v in module.state_dict().items() if 'num_batches' not in k}


def rand_boxes(rng, n, lo=0, hi=400, wmin=12, wmax=192, clip=591):
    xy = rng.uniform(lo, hi, size=(n, 2))
    wh = rng.uniform(wmin, wmax, size=(n, 2))
    return np.concatenate((xy, np.minimum(xy + wh, clip)), 1).astype(np.float32)


# ---------------------------------------------------------------- raster
def gold_raster():
    rng = np.random.RandomState(111)
    a, b = rand_boxes(rng, 56), rand_boxes(rng, 56)
    pairs = np.concatenate((a, b), 1)
    special = np.array([
        [10, 10, 50, 50, 10, 10, 50, 50],            # identical
        [0, 0, 100, 100, 25, 25, 75, 75],            # nested
        [0, 0, 10, 10, 90, 90, 100, 100],            # disjoint
        [0, 0, 50, 50, 50, 0, 100, 50],              # touching edge
        [0, 0, 27, 27, 0, 0, 13.5, 13.5],            # grid-aligned
        [3.25, 7.5, 300.75, 20.125, 100, 1, 120, 500],   # thin / tall
        [0, 0, 591, 591, 1, 1, 2, 2],                # tiny inside huge
        [5, 5, 6, 6, 5.5, 5.5, 6.5, 6.5],            # sub-pixel
    ], dtype=np.float32)
    pairs = np.concatenate((pairs, special), 0).astype(np.float32)
    out27 = dr.draw_union_boxes(pairs, 27)
    out7 = dr.draw_union_boxes(pairs[:16], 7)
    save('raster', pairs=pairs, out27=out27, out7=out7)


# ---------------------------------------------------------------- UnionBoxesAndFeats
def randomize_bn(m, g):
    for mod in m.modules():
        if isinstance(mod, nn.BatchNorm2d):
            mod.running_mean.copy_(torch.randn(mod.running_mean.shape, generator=g) * 0.1)
            mod.running_var.copy_(torch.rand(mod.running_var.shape, generator=g) + 0.5)
            mod.weight.data.copy_(torch.rand(mod.weight.shape, generator=g) + 0.5)
            mod.bias.data.copy_(torch.randn(mod.bias.shape, generator=g) * 0.1)


def gold_union_feats():
    g = torch.Generator().manual_seed(111)
    rng = np.random.RandomState(5)
    arrs = {}
    for dim, n, tag in ((128, 6, 'd128'), (32, 9, 'd32')):
        torch.manual_seed(7 + dim)
        m = UnionBoxesAndFeats(pooling_size=7, stride=16, dim=dim, edge_model='motifs')
        randomize_bn(m, g)
        m.eval()
        boxes = rand_boxes(rng, n)
        rois = np.concatenate((np.zeros((n, 1), np.float32), boxes), 1)
        ui = np.array([(i, j) for i in range(n) for j in range(n) if i != j], dtype=np.int64)
        if dim == 128:
            ui = ui[::3]
        pools = torch.randn(len(ui), dim, 7, 7, generator=g)
        with torch.no_grad():
            out = m(pools, torch.from_numpy(rois), torch.from_numpy(ui), None)
            conv_out = out - pools
        assert conv_out.shape[-1] == 7 and torch.allclose(conv_out[..., :1, :1].expand_as(conv_out), conv_out, atol=1e-6)
        arrs.update({tag + '_rois': rois, tag + '_union_inds': ui, tag + '_pools': pools, tag + '_out': out,
                     tag + '_rect_feat': (out - pools)[:, :, 0, 0]})
        for k, v in sd(m).items():
            arrs[tag + '_w_' + k] = v
    # edge_model 'raw_boxes' (lib/get_union_boxes.py:69-80,105-116): boxes in image coordinates through grid_sample
    from lib.get_union_boxes import draw_union_boxes_grid
    n, dim = 9, 32
    torch.manual_seed(3)
    m = UnionBoxesAndFeats(pooling_size=7, stride=16, dim=dim, edge_model='raw_boxes')
    randomize_bn(m, g)
    m.eval()
    im = np.array([0] * 5 + [1] * 4)
    im_sizes = [(592, 400), (333, 592)]                      # (h, w) per image
    boxes = rand_boxes(rng, n, hi=200, wmax=120, clip=330)
    boxes[3] = [0, 0, 399, 591]                                # a box covering its whole image
    boxes[6] = [10.5, 20.25, 12.0, 300.0]                      # thin
    rois = np.concatenate((im[:, None].astype(np.float32), boxes), 1)
    ui = np.array([(i, j) for i in range(n) for j in range(n) if i != j and im[i] == im[j]], dtype=np.int64)
    pools = torch.randn(len(ui), dim, 7, 7, generator=g)
    with torch.no_grad():
        scale = torch.tensor([[im_sizes[i][1], im_sizes[i][0]] * 2 for i in im], dtype=torch.float32)
        grid = draw_union_boxes_grid(torch.from_numpy(boxes) / scale, torch.from_numpy(ui), 27)
        out = m(pools, torch.from_numpy(rois), torch.from_numpy(ui), im_sizes)
    arrs.update({'raw_rois': rois, 'raw_union_inds': ui, 'raw_pools': pools, 'raw_out': out, 'raw_rects': grid,
                 'raw_rect_feat': (out - pools)[:, :, 0, 0], 'raw_im_sizes': np.array(im_sizes)})
    for k, v in sd(m).items():
        arrs['raw_w_' + k] = v
    save('union_feats', **arrs)


# ---------------------------------------------------------------- model without torchvision
class _Data(object):
    ind_to_classes = ['c%d' % i for i in range(151)]
    ind_to_predicates = ['p%d' % i for i in range(51)]


def make_model(C, fc_dim, hidden, mp_iter=3, num_classes=151, num_rels=51, seed=0):
    """RelModelStanford built field by field (its constructor needs torchvision,
    sgg_models/rel_model_base.py:92-112); the attached sub-modules are the same torch.nn classes with the
    same names as rel_model_base.py:110-120 and rel_model_stanford.py:29-45."""
    torch.manual_seed(seed)
    m = RelModelStanford.__new__(RelModelStanford)
    nn.Module.__init__(m)
    m.classes = ['c%d' % i for i in range(num_classes)]
    m.rel_classes = ['p%d' % i for i in range(num_rels)]
    m.mode, m.backbone, m.RELS_PER_IMG, m.pool_sz, m.stride = 'sgcls', 'vgg16', 1024, 7, 16
    m.use_bias = m.test_bias = m.require_overlap = False
    m.obj_dim, m.hidden_dim, m.mp_iter, m.edge_dim = fc_dim, hidden, mp_iter, C
    cls = lambda: nn.Sequential(nn.Linear(C * 49, fc_dim), nn.ReLU(True), nn.Dropout(),
                                nn.Linear(fc_dim, fc_dim), nn.ReLU(True), nn.Dropout())
    vc = cls()
    del vc._modules['5']
    del vc._modules['4']
    m.roi_fmap = nn.Sequential(nn.Flatten(), vc)
    m.roi_fmap_obj = cls()
    m.union_boxes = UnionBoxesAndFeats(pooling_size=7, stride=16, dim=C, edge_model='motifs')
    m.rel_fc = nn.Linear(hidden, num_rels)
    m.obj_fc = nn.Linear(hidden, num_classes)
    m.obj_unary = nn.Linear(fc_dim, hidden)
    m.edge_unary = nn.Linear(fc_dim, hidden)
    m.edge_gru = nn.GRUCell(hidden, hidden)
    m.node_gru = nn.GRUCell(hidden, hidden)
    for nm in ('sub_vert_w_fc', 'obj_vert_w_fc', 'out_edge_w_fc', 'in_edge_w_fc'):
        setattr(m, nm, nn.Sequential(nn.Linear(hidden * 2, 1), nn.Sigmoid()))
    g = torch.Generator().manual_seed(seed + 1)
    randomize_bn(m, g)
    # default GRU/Linear init is tiny at these widths; widen a little so gates are not all ~0.5
    with torch.no_grad():
        for n_, p_ in m.named_parameters():
            if 'w_fc' in n_ or 'gru' in n_:
                p_.mul_(3.0)
    m.eval()
    return m


def graph(rng, sizes, drop=0.0):
    """ragged multi-image graph: returns im_inds, rel_inds[E,3] sorted like the reference."""
    im = np.concatenate([np.full(n, i, np.int64) for i, n in enumerate(sizes)])
    rel, off = [], 0
    for i, n in enumerate(sizes):
        for s in range(n):
            for o in range(n):
                if s != o and rng.rand() >= drop:
                    rel.append((i, off + s, off + o))
        off += n
    return im, np.array(rel, dtype=np.int64)


def gold_gru():
    torch.manual_seed(3)
    cell = nn.GRUCell(48, 48)
    x, h = torch.randn(21, 48), torch.randn(21, 48)
    with torch.no_grad():
        save('gru', x=x, h=h, out=cell(x, h), out_h0=cell(x, torch.zeros_like(h)),
             **{'w_' + k: v for k, v in sd(cell).items()})


def gold_message_pass():
    rng = np.random.RandomState(11)
    arrs = {}
    for tag, hidden, sizes, drop in (('h32_b1', 32, (7,), 0.0), ('h64_b3', 64, (5, 9, 3), 0.0),
                                     ('h32_sampled', 32, (6, 4), 0.4), ('h128_b2', 128, (4, 3), 0.0)):
        m = make_model(8, 16, hidden, seed=sum(map(ord, tag)))
        im, rel = graph(rng, sizes, drop)
        N, E = len(im), len(rel)
        g = torch.Generator().manual_seed(E)
        obj_rep, rel_rep = torch.randn(N, hidden, generator=g), torch.randn(E, hidden, generator=g).relu()
        with torch.no_grad():
            for it in range(4):
                m.mp_iter = it
                v, e = m.message_pass(rel_rep, obj_rep, torch.from_numpy(rel[:, 1:3]))
                arrs['%s_v%d' % (tag, it)], arrs['%s_e%d' % (tag, it)] = v, e
        arrs.update({tag + '_obj_rep': obj_rep, tag + '_rel_rep': rel_rep, tag + '_rel_inds': rel})
        for k, v in sd(m).items():
            if 'gru' in k or 'w_fc' in k:
                arrs[tag + '_w_' + k] = v
    save('message_pass', **arrs)


def gold_predict():
    rng = np.random.RandomState(21)
    arrs = {}
    for tag, C, fc, hidden, sizes in (('small', 8, 24, 32, (5, 4)), ('b3', 16, 32, 64, (3, 6, 2))):
        m = make_model(C, fc, hidden, seed=len(tag))
        im, rel = graph(rng, sizes)
        N, E = len(im), len(rel)
        boxes = rand_boxes(rng, N)
        rois = np.concatenate((im[:, None].astype(np.float32), boxes), 1)
        g = torch.Generator().manual_seed(N * E)
        nf, ef = torch.randn(N, C, 7, 7, generator=g), torch.randn(E, C, 7, 7, generator=g)
        with torch.no_grad():
            od, rd = m.predict(nf, ef, torch.from_numpy(rel), torch.from_numpy(rois), None)
        arrs.update({tag + '_node_feat': nf, tag + '_edge_feat': ef, tag + '_rel_inds': rel, tag + '_rois': rois,
                     tag + '_obj_dists': od, tag + '_rel_dists': rd})
        for k, v in sd(m).items():
            arrs[tag + '_w_' + k] = v
    save('predict', **arrs)


def gold_pairs():
    rng = np.random.RandomState(31)
    arrs = {}
    m = make_model(8, 16, 32)
    for tag, sizes in (('b1', (6,)), ('b3', (4, 7, 2)), ('b8x32', (32,) * 8)):
        im = np.concatenate([np.full(n, i, np.int64) for i, n in enumerate(sizes)])
        boxes = rand_boxes(rng, len(im))
        cls = rng.randint(1, 151, size=len(im))
        gt_classes = np.stack((im, cls), 1).astype(np.int64)
        rels = []
        for i, n in enumerate(sizes):
            seen = set()
            while len(seen) < min(6, n * (n - 1) // 2):
                s, o = rng.randint(n), rng.randint(n)
                if s != o and (s, o) not in seen:
                    seen.add((s, o))
                    rels.append((i, s, o, rng.randint(1, 51)))
        gt_rels = np.array(rels, dtype=np.int64)
        tim, tb = torch.from_numpy(im), torch.from_numpy(boxes)
        for ov in (False, True):
            m.require_overlap = ov
            m.eval()
            ri = m.get_rel_inds(None, tim, tb)
            arrs['%s_eval_ov%d' % (tag, int(ov))] = ri
        rois = torch.cat((tim[:, None].float(), tb), 1)
        r, lab, rl = proposal_assignments_gtbox(rois, tb, torch.from_numpy(gt_classes), torch.from_numpy(gt_rels),
                                                0, 1024)
        m.train()
        arrs.update({tag + '_im_inds': im, tag + '_boxes': boxes, tag + '_gt_classes': gt_classes,
                     tag + '_gt_rels': gt_rels, tag + '_train_labels': lab, tag + '_train_rel_labels': rl,
                     tag + '_train_rel_inds': m.get_rel_inds(rl, tim, tb)})
        m.eval()
    save('pairs', **arrs)


def gold_rel_assign():
    """lib/rel_assignments.py:12 on jittered / relabelled / background detections around seeded GT graphs.  The function ends with
    `.cuda(rpn_rois.get_device())` (:135): Tensor.cuda is made the identity for the duration of the call (no device here)."""
    from lib.rel_assignments import rel_assignments
    rng = np.random.RandomState(77)
    arrs, cases = {}, []
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        for tag, sizes, ndet in (('b1', (6,), (14,)), ('b3', (5, 9, 3), (20, 40, 2)), ('b2dense', (12, 12), (50, 50)), ('b2none', (4, 4), (6, 5))):
            gt_im = np.concatenate([np.full(n, i, np.int64) for i, n in enumerate(sizes)])
            gt_boxes = rand_boxes(rng, len(gt_im), wmin=40, wmax=220)
            gt_cls = rng.randint(1, 151, size=len(gt_im))
            gt_classes = np.stack((gt_im, gt_cls), 1).astype(np.int64)
            rels, first = [], np.concatenate(([0], np.cumsum(sizes)))
            for i, n in enumerate(sizes):
                seen = set()
                while len(seen) < min(2 * n, n * (n - 1)):
                    a, b = rng.randint(n), rng.randint(n)
                    if a != b:                                   # duplicates with different predicates are allowed (VG has them)
                        seen.add((a, b, rng.randint(1, 51)))
                rels += [(i, a, b, p) for a, b, p in sorted(seen)]
            gt_rels = np.array(rels, dtype=np.int64)
            det_im, det_boxes, det_lab = [], [], []
            for i, (n, nd) in enumerate(zip(sizes, ndet)):
                for _ in range(nd):
                    g = first[i] + rng.randint(n)
                    kind = rng.rand()
                    if tag == 'b2none':                         # nothing matches: wrong labels everywhere
                        box, lab = gt_boxes[g] + rng.uniform(-4, 4, 4), (gt_cls[g] % 150) + 1
                    elif kind < 0.55:                           # a good detection of a GT box
                        box, lab = gt_boxes[g] + rng.uniform(-8, 8, 4), gt_cls[g]
                    elif kind < 0.65:                           # the GT box itself (IoU exactly 1 with it)
                        box, lab = gt_boxes[g].copy(), gt_cls[g]
                    elif kind < 0.8:                            # right place, wrong label
                        box, lab = gt_boxes[g] + rng.uniform(-8, 8, 4), rng.randint(1, 151)
                    elif kind < 0.9:                            # background label
                        box, lab = gt_boxes[g] + rng.uniform(-30, 30, 4), 0
                    else:                                       # somewhere else
                        box, lab = rand_boxes(rng, 1)[0], rng.randint(1, 151)
                    box = np.clip(box, 0, 591).astype(np.float32)
                    box[2:] = np.maximum(box[2:], box[:2] + 2)
                    det_im.append(i); det_boxes.append(box); det_lab.append(lab)
            det_im = np.array(det_im, np.int64); det_boxes = np.stack(det_boxes).astype(np.float32); det_lab = np.array(det_lab, np.int64)
            arrs.update({tag + '_im_inds': det_im, tag + '_boxes': det_boxes, tag + '_labels': det_lab, tag + '_gt_boxes': gt_boxes,
                         tag + '_gt_classes': gt_classes, tag + '_gt_rels': gt_rels})
            for k, (per_gt, nonov, seed) in enumerate(((1, True, 5), (4, True, 6), (1, False, 7), (4, False, 8), (1, True, 9))):
                np.random.seed(seed)
                out = rel_assignments(torch.from_numpy(det_im), torch.from_numpy(det_boxes), torch.from_numpy(det_lab),
                                      torch.from_numpy(gt_boxes), torch.from_numpy(gt_classes), torch.from_numpy(gt_rels), 0,
                                      filter_non_overlap=nonov, num_sample_per_gt=per_gt)
                arrs['%s_out%d' % (tag, k)] = out
                arrs['%s_cfg%d' % (tag, k)] = np.array([per_gt, int(nonov), seed], np.int64)
            cases.append(tag)
    finally:
        torch.Tensor.cuda = real_cuda
    arrs['cases'] = np.array(cases)
    save('rel_assign', **arrs)


def gold_gan_model():
    import warnings
    import augment.gan as RG
    warnings.simplefilter('ignore')
    obj_classes = ['__background__'] + ['obj%d' % i for i in range(1, 9)]
    rel_classes = ['__background__'] + ['rel%d' % i for i in range(1, 5)]
    RG.obj_edge_vectors = lambda names, wv_dir='', wv_dim=300, word_vectors=None, avg_words=False: (torch.randn(len(names), wv_dim), None)
    torch.manual_seed(123)
    gan = RG.GAN(obj_classes, rel_classes, embed_dim=12, hidden_dim=16, n_ch=32, pool_sz=7, fmap_sz=38, n_layers_G=3, device='cpu')
    gan.train()
    arrs = {'w_' + k: v for k, v in sd(gan).items()}
    rng = np.random.RandomState(5)
    n_per = [4, 2, 5]
    objs, boxes, rels = [], [], []
    for im, n in enumerate(n_per):
        objs += [(im, int(c)) for c in rng.randint(1, len(obj_classes), size=n)]
        xy = rng.uniform(0, 0.55, size=(n, 2)); wh = rng.uniform(0.1, 0.45, size=(n, 2))
        boxes.append(np.concatenate((xy, np.minimum(xy + wh, 1.0)), 1))
        t = rng.randint(1, 2 * n)
        for _ in range(t):
            s_, o_ = rng.choice(n, 2, replace=False)
            rels.append((im, int(s_), int(o_), int(rng.randint(1, len(rel_classes)))))
    gt_objs = torch.tensor(objs, dtype=torch.int64)
    gt_boxes = torch.from_numpy(np.concatenate(boxes).astype(np.float32))
    gt_rels = torch.tensor(sorted(rels), dtype=torch.int64)
    arrs.update(in_objs=gt_objs, in_boxes=gt_boxes, in_rels=gt_rels)
    d_objs, d_boxes, d_rels = RG.dummy_nodes(gt_objs, gt_boxes, gt_rels)
    arrs.update(dummy_objs=d_objs, dummy_boxes=d_boxes, dummy_rels=d_rels)
    fmap = gan(gt_objs, gt_boxes, gt_rels)
    arrs['fmap_fake'] = fmap
    g = torch.Generator().manual_seed(6)
    n_obj, n_rel = len(gt_objs), 9
    f_real_o, f_fake_o = torch.randn(n_obj, 32 * 49, generator=g), torch.randn(n_obj, 32 * 49, generator=g)
    f_real_r, f_fake_r = torch.randn(n_rel + 2, 32, 7, 7, generator=g), torch.randn(n_rel, 32, 7, 7, generator=g)
    lab_o = gt_objs[:, 1].clone()
    lab_r_fake = torch.from_numpy(rng.randint(0, len(rel_classes), size=n_rel)).long()
    lab_r_real = torch.from_numpy(rng.randint(0, len(rel_classes), size=n_rel + 2)).long()
    fmap_real = torch.randn(3, 32, 38, 38, generator=g).relu()
    arrs.update(f_real_o=f_real_o, f_fake_o=f_fake_o, f_real_r=f_real_r, f_fake_r=f_fake_r, lab_o=lab_o, lab_r_fake=lab_r_fake,
                lab_r_real=lab_r_real, fmap_real=fmap_real)
    # fixed order of calls: every call advances the spectral-norm vectors of the discriminator it uses
    out = {}
    out.update(gan.loss(f_real_o, f_fake_o, is_nodes=True, updateD=True, labels_fake=lab_o))                       # D_obj (real labels = fake labels)
    out.update(gan.loss(f_real_r, f_fake_r, is_nodes=False, updateD=True, labels_fake=lab_r_fake, labels_real=lab_r_real))   # D_rel
    out.update(gan.loss(fmap_real, fmap, updateD=True, is_fmaps=True))                                             # D_fmap
    out.update(gan.loss(features_fake=f_fake_o, is_nodes=True, labels_fake=lab_o))                                  # G_obj
    out.update(gan.loss(features_fake=f_fake_r, is_nodes=False, labels_fake=lab_r_fake))                            # G_rel
    g_fmap = gan.loss(features_fake=fmap, is_fmaps=True)                                                            # G_fmap
    out.update(g_fmap)
    for k, v in out.items():
        arrs['loss_' + k] = v
    names = ['G_obj_embed.weight', 'G_proj.weight', 'G_refine.refinement_modules.0.net.0.weight', 'G_gcn.gconvs.0.net1.0.weight']
    params = dict(gan.named_parameters())
    grads = torch.autograd.grad(g_fmap['G_fmap'], [params[n] for n in names])
    for n, gr in zip(names, grads):
        arrs['grad_' + n] = gr
    save('gan_model', **arrs)


def gold_sg_perturb():
    if not hasattr(np, 'Inf'):
        np.Inf = np.inf                  # the reference spells it the NumPy-1 way (augment/sg_perturb.py:124,181); same value
    from augment.sg_perturb import SceneGraphPerturb
    rng = np.random.RandomState(17)
    C, P = 12, 6
    embed = torch.randn(C, 10, generator=torch.Generator().manual_seed(3))
    embed = embed / embed.norm(2, dim=1, keepdim=True)
    # a batch of 3 images
    objs, rels = [], []
    for im, n in enumerate((6, 3, 8)):
        cls = rng.randint(1, C, size=n)
        objs += [(im, int(c)) for c in cls]
        pairs = [(a, b) for a in range(n) for b in range(n) if a != b]
        for k in rng.choice(len(pairs), size=min(len(pairs), 2 * n), replace=False):
            rels.append((im, pairs[k][0], pairs[k][1], int(rng.randint(0, P))))          # predicate 0 = background rows too
    gt_obj = torch.tensor(objs, dtype=torch.int64)
    gt_rels = torch.tensor(sorted(rels), dtype=torch.int64)
    # synthetic training statistics in the dataset's form: {"s_p": {o: count}}, {"p_o": {s: count}}
    subj_pred, pred_obj, rows = {}, {}, []
    for _ in range(400):
        s_, p_, o_, cnt = int(rng.randint(1, C)), int(rng.randint(1, P)), int(rng.randint(1, C)), int(rng.randint(1, 40))
        subj_pred.setdefault('{}_{}'.format(s_, p_), {})[o_] = cnt
        pred_obj.setdefault('{}_{}'.format(p_, o_), {})[s_] = cnt
        rows.append((s_, p_, o_, cnt))
    arrs = dict(embed=embed, gt_obj=gt_obj, gt_rels=gt_rels, stats_rows=np.array(rows, dtype=np.int64))
    cases = [('rand', dict(L=0.5)), ('neigh', dict(L=0.3, topk=3)), ('graphn', dict(L=0.5, topk=0, alpha=2)),
             ('graphn', dict(L=1.0, topk=4, alpha=1)), ('rand', dict(L=0.2, uniform=True)), ('graphn', dict(L=0.4, topk=2, alpha=5, degree_smoothing=0.5))]
    for k, (method, kw) in enumerate(cases):
        sgp = SceneGraphPerturb(method=method, embed_objs=embed.clone(), subj_pred_obj_pairs=(subj_pred, pred_obj), **kw)
        for rep in range(3):
            np.random.seed(100 * k + rep)
            torch.manual_seed(100 * k + rep)
            arrs['case%d_rep%d' % (k, rep)] = sgp.perturb(gt_obj.clone(), gt_rels.clone())
    save('sg_perturb', **arrs)


def gold_gan_ops():
    import warnings
    from augment.graphconv import GraphTripleConv
    from augment.layout import boxes_to_layout
    warnings.simplefilter('ignore')                       # grid_sample's align_corners notice (default False since torch 1.3)
    g = torch.Generator().manual_seed(91)
    rng = np.random.RandomState(91)
    arrs = {}
    O_, D = 14, 8
    img = np.sort(rng.randint(0, 3, size=O_)).astype(np.int64)
    img[img == 1] = 2                                     # image 1 has no objects: its canvas stays zero
    xy = rng.uniform(0, 0.6, size=(O_, 2)); wh = rng.uniform(0.08, 0.6, size=(O_, 2))
    boxes = np.concatenate((xy, np.minimum(xy + wh, 1.0)), 1).astype(np.float32)
    boxes[0] = (0, 0, 1, 1)                                # the whole canvas
    boxes[1] = (0.5, 0.5, 0.52, 0.53)                      # smaller than one canvas cell
    arrs.update(lay_boxes=boxes, lay_img=img)
    for tag, shape in (('patch', (O_, D, 7, 7)), ('vec', (O_, D))):
        v = torch.randn(*shape, generator=g)
        arrs['lay_%s_in' % tag] = v
        for hw in ((38, 38), (10, 14)):
            for pool in ('sum', 'avg'):
                vv = v.clone().requires_grad_(True)
                out = boxes_to_layout(vv, torch.from_numpy(boxes), torch.from_numpy(img), hw[0], hw[1], pooling=pool)
                up = torch.randn(out.shape, generator=g)
                (out * up).sum().backward()
                key = 'lay_%s_%dx%d_%s' % (tag, hw[0], hw[1], pool)
                arrs.update({key + '_out': out, key + '_up': up, key + '_din': vv.grad})
    # one scene-graph convolution layer, both final_nonlinearity settings (GraphTripleConvNet uses False for the last layer)
    T, Din, De, Hd = 30, 12, 10, 16
    edges = np.stack((rng.randint(0, O_, T), rng.randint(0, O_, T)), 1).astype(np.int64)
    edges[edges[:, 0] == 5, 0] = 6                         # object 5 appears in no triple as a subject ...
    edges[edges[:, 1] == 5, 1] = 6                         # ... nor as an object: count 0 -> clamp(min=1)
    obj, pred = torch.randn(O_, Din, generator=g), torch.randn(T, De, generator=g)
    arrs.update(gc_edges=edges, gc_obj=obj, gc_pred=pred, gc_hidden=np.int64(Hd))
    for k, (final, pooling, dout) in enumerate(((True, 'avg', 16), (False, 'avg', 9), (True, 'sum', 16))):
        torch.manual_seed(92 + k)
        layer = GraphTripleConv(Din, input_edge_dim=De, output_dim=dout, hidden_dim=Hd, pooling=pooling,
                                mlp_normalization='none', final_nonlinearity=final)
        no, npred = layer(obj, pred, torch.from_numpy(edges))
        arrs.update({'gc%d_%s' % (k, n): t for n, t in sd(layer).items()})
        arrs.update({'gc%d_out_obj' % k: no, 'gc%d_out_pred' % k: npred, 'gc%d_cfg' % k: np.array([int(final), int(pooling == 'avg'), dout])})
    # the stacked network as augment/gan.py:109-115 builds it (reduced sizes), with and without BatchNorm1d (train-mode statistics)
    from augment.graphconv import GraphTripleConvNet
    for tag, norm in (('net', 'none'), ('netbn', 'batch')):
        torch.manual_seed(97)
        net = GraphTripleConvNet(Din, input_edge_dim=De, output_dim=20, num_layers=3, hidden_dim=Hd, pooling='avg', mlp_normalization=norm)
        no, npred = net(obj, pred, torch.from_numpy(edges))
        arrs.update({'%s_%s' % (tag, n): t for n, t in net.state_dict().items() if 'num_batches' not in n})
        arrs.update({tag + '_out_obj': no, tag + '_out_pred': npred})
    save('gan_ops', **arrs)


def gold_eval_tail():
    rng = np.random.RandomState(41)
    g = torch.Generator().manual_seed(41)
    N = 12
    im, rel = graph(rng, (N,))
    od, rd = torch.randn(N, 151, generator=g) * 2, torch.randn(len(rel), 51, generator=g) * 2
    boxes = torch.from_numpy(rand_boxes(rng, N))
    s = torch.softmax(od, 1)
    s[:, 0] = 0
    sc, order = s[:, 1:].sort(dim=1, descending=True)
    obj_preds, obj_scores = order[:, 0] + 1, sc[:, 0]
    rel_rep = torch.softmax(rd, 1)
    b, c, s_, r, ps = filter_dets(boxes, obj_scores, obj_preds, torch.from_numpy(rel[:, 1:]), rel_rep)
    gt_cls = torch.from_numpy(rng.randint(1, 151, size=N))
    b2, c2, s2, r2, ps2 = filter_dets(boxes, torch.ones(N), gt_cls, torch.from_numpy(rel[:, 1:]), rel_rep)
    save('eval_tail', obj_dists=od, rel_dists=rd, rel_inds=rel, boxes=boxes, gt_classes=gt_cls,
         sg_boxes=b, sg_classes=c, sg_scores=s_, sg_rels=r, sg_pred_scores=ps,
         pc_boxes=b2, pc_classes=c2, pc_scores=s2, pc_rels=r2, pc_pred_scores=ps2)


def gold_losses():
    g = torch.Generator().manual_seed(51)
    rd = torch.randn(40, 51, generator=g)
    lab = torch.zeros(40, dtype=torch.long)
    lab[[3, 7, 20]] = torch.tensor([5, 50, 1])
    od, ol = torch.randn(10, 151, generator=g), torch.randint(1, 151, (10,), generator=g)
    arrs = dict(rel_dists=rd, rel_labels=lab, obj_dists=od, obj_labels=ol,
                node=node_losses(od, ol)['obj_loss'])
    for lt in ('baseline', 'dnorm', 'dnorm-fgbg'):
        arrs['edge_' + lt] = edge_losses(rd, lab, loss_type=lt)['rel_loss']
        arrs['edge_nofg_' + lt] = edge_losses(rd, torch.zeros_like(lab), loss_type=lt)['rel_loss']
    save('losses', **arrs)


# ---------------------------------------------------------------- recall evaluator
def recall_cases(n_cases=6, seed=61, n_cls=12, n_pred=9):
    """Synthetic gt / pred entries in the format of lib/eval.py:150-164.  Small class spaces so that matches happen."""
    rng = np.random.RandomState(seed)
    cases = []
    for ci in range(n_cases):
        nb = int(rng.randint(6, 15))
        gt_boxes = rand_boxes(rng, nb)
        gt_classes = rng.randint(1, n_cls, size=nb).astype(np.int64)
        G = int(rng.randint(3, 11))
        pairs = np.array([(i, j) for i in range(nb) for j in range(nb) if i != j], dtype=np.int64)
        gi = rng.choice(len(pairs), size=G, replace=(ci == 0))       # case 0: duplicate pairs allowed
        gt_rels = np.column_stack((pairs[gi], rng.randint(1, n_pred, size=G))).astype(np.int64)
        pred_classes = gt_classes.copy()
        flip = rng.rand(nb) < 0.3
        pred_classes[flip] = rng.randint(1, n_cls, size=int(flip.sum()))
        obj_scores = rng.uniform(0.1, 1.0, size=nb).astype(np.float32)
        logits = rng.randn(len(pairs), n_pred).astype(np.float32) * 2
        logits[gi[: G // 2], gt_rels[: G // 2, 2]] += 4                  # some GT predicates rank high
        rel_scores = np.exp(logits) / np.exp(logits).sum(1, keepdims=True)
        order = rng.permutation(len(pairs))
        jitter = rng.uniform(-1, 1, size=(nb, 4)).astype(np.float32) * rng.choice([2.0, 8.0, 40.0], size=(nb, 1)).astype(np.float32)
        pred_boxes = gt_boxes + jitter
        pred_boxes[:, 2:] = np.maximum(pred_boxes[:, 2:], pred_boxes[:, :2] + 4)
        cases.append(dict(gt_boxes=gt_boxes, gt_classes=gt_classes, gt_relations=gt_rels, pred_boxes=pred_boxes.astype(np.float32),
                          pred_classes=pred_classes, obj_scores=obj_scores, pred_rel_inds=pairs[order],
                          rel_scores=rel_scores[order].astype(np.float32)))
    return cases


def gold_recall():
    import contextlib
    import io
    np.int = int                                                  # lib/sgg_eval.py:238 uses the removed alias
    from lib.sgg_eval import BasicSceneGraphEvaluator, calculate_mR_from_evaluator_list, eval_entry
    cases = recall_cases()
    arrs = {'n_cases': len(cases)}
    for i, c in enumerate(cases):
        for k, v in c.items():
            arrs['c%d_%s' % (i, k)] = v
    split = lambda c: ({k: c[k] for k in ('gt_boxes', 'gt_classes', 'gt_relations')},
                       {k: c[k].copy() for k in ('pred_boxes', 'pred_classes', 'obj_scores', 'pred_rel_inds', 'rel_scores')})
    quiet = contextlib.redirect_stdout(io.StringIO())
    for mode in ('predcls', 'sgcls', 'sgdet', 'phrdet', 'objcls', 'preddet'):
        for mp in (False, True):
            ev = BasicSceneGraphEvaluator(mode, multiple_preds=mp)
            firsts = []
            for c in cases:
                gt, pr = split(c)
                pred_to_gt, _, _ = ev.evaluate_scene_graph_entry(gt, pr)
                if pred_to_gt is not None:
                    first = np.full(len(c['gt_relations']), 2 ** 31 - 1, np.int64)
                    for p, lst in enumerate(pred_to_gt):
                        for g_ in lst:
                            first[g_] = min(first[g_], p)
                    firsts.append(first)
                    arrs.setdefault('nmatch_%s_%d' % (mode, mp), []).append(sum(len(x) for x in pred_to_gt))
            tag = '%s_%d' % (mode, int(mp))
            for k, v in ev.result_dict[mode + '_recall'].items():
                arrs['recall_%s_%d' % (tag, k)] = np.array(v, dtype=np.float64)
            if firsts:
                arrs['first_%s' % tag] = np.concatenate(firsts)
    # per-triplet statistics (needs the graph-unconstrained ranking)
    rng = np.random.RandomState(62)
    counts = {}
    for c in cases:
        for (o, s_, R) in c['gt_relations']:
            if rng.rand() < 0.7:
                counts['{}_{}_{}'.format(c['gt_classes'][o], R, c['gt_classes'][s_])] = int(rng.randint(0, 200))
    arrs['tc_keys'] = np.array(sorted(counts), dtype='U32')
    arrs['tc_vals'] = np.array([counts[k] for k in sorted(counts)], dtype=np.int64)
    ev = BasicSceneGraphEvaluator('sgcls', multiple_preds=True, per_triplet=True, triplet_counts=counts)
    for c in cases:
        ev.evaluate_scene_graph_entry(*split(c))
    with quiet:
        ev.print_stats()
    rd = ev.result_dict
    arrs['pt_rank'] = np.array(rd['sgcls_rank'], dtype=np.float64)
    arrs['pt_counts'] = np.array(rd['sgcls_counts'], dtype=np.float64)
    for k, v in rd['sgcls_recall_norm'].items():
        arrs['pt_recall_norm_%d' % k] = np.array(v, dtype=np.float64)
    for sfx in ('', '_norm'):
        arrs['pt_recall_triplet' + sfx] = np.array([rd['sgcls_recall_triplet' + sfx][k] for k in (5, 10, 15, 20, 50)], dtype=np.float64)
    arrs['pt_scalars'] = np.array([rd['sgcls_meanrank_triplet'], rd['sgcls_meanrank_triplet_norm'], rd['sgcls_medianrank_triplet'],
                                   rd['sgcls_medianrankclass_triplet'], rd['sgcls_medianrank_triplet_norm']], dtype=np.float64)
    # mean recall over predicates (lib/eval.py:46-53,76-80)
    for mp in (False, True):
        lst = [(pid, 'p%d' % pid, BasicSceneGraphEvaluator.all_modes(multiple_preds=mp)) for pid in range(1, 9)]
        other = [(pid, 'p%d' % pid, BasicSceneGraphEvaluator.all_modes(multiple_preds=not mp)) for pid in range(1, 9)]
        for c in cases:
            gt, pr = split(c)
            eval_entry('sgcls', gt, pr, lst if not mp else other, other if not mp else lst)
        with quiet:
            mr = calculate_mR_from_evaluator_list(lst, 'sgcls', multiple_preds=mp)
        arrs['mR_%d' % int(mp)] = np.array([mr['R@%d' % k] for k in (20, 50, 100, 200, 300)], dtype=np.float64)
    for k in list(arrs):
        if k.startswith('nmatch_'):
            arrs[k] = np.array(arrs[k], dtype=np.int64)
    save('recall', **arrs)


class _DS(object):
    """An HDF5 dataset: indexing reads a fresh copy (the reference edits what it reads in place)."""

    def __init__(self, a):
        self.a = a
        self.shape = a.shape

    def __getitem__(self, idx):
        return np.array(self.a[idx])


class _H5(dict):
    """Stand-in for h5py.File(path, 'r'): a mapping of datasets usable as a context manager."""

    def __getitem__(self, k):
        return _DS(dict.__getitem__(self, k))

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def synthetic_vg_h5(seed=17, n_img=40, n_cls=12, n_pred=9):
    """Arrays with the keys / dtypes / conventions of VG-SGG.h5 (boxes_1024 = xc, yc, w, h; relationships = GLOBAL box ids;
    -1 first-indices for images without boxes / relations)."""
    rng = np.random.RandomState(seed)
    split = (rng.rand(n_img) < 0.3).astype(np.int32) * 2
    boxes, labels, rels, preds = [], [], [], []
    first_box, last_box, first_rel, last_rel = [], [], [], []
    for i in range(n_img):
        nb = 0 if i in (3, 11) else int(rng.randint(2, 9))
        if nb == 0:
            first_box.append(-1); last_box.append(-1); first_rel.append(-1); last_rel.append(-1)
            continue
        fb = len(labels)
        xy = rng.uniform(40, 700, size=(nb, 2))
        wh = rng.uniform(20, 300, size=(nb, 2))
        boxes.append(np.concatenate((xy + wh / 2, wh), 1).astype(np.int32))
        labels.extend(rng.randint(1, n_cls, size=nb))
        first_box.append(fb); last_box.append(fb + nb - 1)
        nr = 0 if i % 7 == 5 else int(rng.randint(1, 7))
        if nr == 0:
            first_rel.append(-1); last_rel.append(-1)
            continue
        fr = len(preds)
        s = rng.randint(0, nb, size=nr)
        o = (s + rng.randint(1, nb, size=nr)) % nb
        if nr > 2:                                  # a duplicated pair with a different predicate (filter_dups)
            s[-1], o[-1] = s[0], o[0]
        rels.append(np.stack((s, o), 1) + fb)
        preds.extend(rng.randint(1, n_pred, size=nr))
        first_rel.append(fr); last_rel.append(fr + nr - 1)
    return _H5({'split': split, 'img_to_first_box': np.array(first_box, np.int32), 'img_to_last_box': np.array(last_box, np.int32),
                'img_to_first_rel': np.array(first_rel, np.int32), 'img_to_last_rel': np.array(last_rel, np.int32),
                'boxes_1024': np.concatenate(boxes).astype(np.int32), 'labels': np.array(labels, np.int32)[:, None],
                'relationships': np.concatenate(rels).astype(np.int32), 'predicates': np.array(preds, np.int32)[:, None]})


LOADER_CASES = [
    dict(mode='train'), dict(mode='test'), dict(mode='val', num_val_im=6), dict(mode='train', num_val_im=6),
    dict(mode='train', filter_empty_rels=False), dict(mode='train', num_im=10),
    dict(mode='train', min_graph_size=3, max_graph_size=6), dict(mode='train', filter_non_overlap=True),
    dict(mode='test', training_triplets='set'), dict(mode='test', training_triplets='set', filter_zeroshots=False),
    dict(mode='test', training_triplets='counts', n_shots=10), dict(mode='test', training_triplets='counts', n_shots=100),
]


def loader_triplets(h5):
    """training-triplet collections in the two forms load_graphs accepts: a set of 'cls_pred_cls' strings (zero-shot filter)
    and a {string: count} dict (few-shot filter), built from the train split of the same arrays."""
    lab, rel, pred = h5['labels'][:, 0], h5['relationships'], h5['predicates'][:, 0]
    counts = {}
    for i in np.where((h5['split'] == 0) & (h5['img_to_first_rel'] >= 0))[0][::2]:
        for r in range(h5['img_to_first_rel'][i], h5['img_to_last_rel'][i] + 1):
            k = '{}_{}_{}'.format(lab[rel[r, 0]], pred[r], lab[rel[r, 1]])
            counts[k] = counts.get(k, 0) + (1 if len(counts) % 3 else 12)
    return set(counts), counts


def gold_vg_loader():
    import json
    import tempfile
    from PIL import Image
    import torchvision.transforms as T

    class _Stub(object):
        def __init__(self, *a, **k):
            pass
    for n in ('Resize', 'Compose', 'ToTensor', 'Normalize'):
        setattr(T, n, _Stub)                                   # names only: dataloaders/visual_genome.py:14 imports them
    np.float, np.bool = float, bool
    import dataloaders.visual_genome as vg
    from dataloaders.image_transforms import SquarePad
    h5 = synthetic_vg_h5()
    arrs = {'h5_' + k: v for k, v in h5.items()}
    tset, tcounts = loader_triplets(dict(h5.items()))
    arrs['triplet_keys'] = np.array(sorted(tcounts))
    arrs['triplet_counts'] = np.array([tcounts[k] for k in sorted(tcounts)])
    vg.h5py.File = lambda path, mode='r': h5
    for ci, case in enumerate(LOADER_CASES):
        kw = dict(case)
        if kw.get('training_triplets') == 'set':
            kw['training_triplets'] = tset
        elif kw.get('training_triplets') == 'counts':
            kw['training_triplets'] = tcounts
        mask, boxes, classes, rels = vg.load_graphs('unused.h5', **kw)
        arrs['c%d_mask' % ci] = mask
        arrs['c%d_n' % ci] = len(boxes)
        for i in range(len(boxes)):
            arrs['c%d_boxes_%d' % (ci, i)], arrs['c%d_classes_%d' % (ci, i)], arrs['c%d_rels_%d' % (ci, i)] = boxes[i], classes[i], rels[i]
    # load_info
    info = {'label_to_idx': {'dog': 2, 'cat': 1, 'tree': 3}, 'predicate_to_idx': {'on': 1, 'near': 3, 'has': 2}}
    with tempfile.NamedTemporaryFile('w', suffix='.json', delete=False) as f:
        json.dump(info, f)
    c, p = vg.load_info(f.name)
    arrs['info_classes'], arrs['info_predicates'] = np.array(c), np.array(p)
    # filter_dups, deterministic form
    gr = np.array([[0, 1, 3], [2, 1, 4], [0, 1, 5], [2, 1, 4], [1, 0, 2]])
    arrs['dups_in'], arrs['dups_out'] = gr, vg.filter_dups(gr, random_edge=False)
    # VG.__getitem__ box / size arithmetic on synthetic images (torch_detector True and False, flipped or not)
    rng = np.random.RandomState(3)
    sizes = [(50, 37), (37, 50), (40, 40), (102, 68)]          # small: the arithmetic only depends on (w, h)
    n_entries = 0
    for torch_detector in (True, False):
        for (w, h) in sizes:
            for flip in (False, True):
                ds = vg.VG.__new__(vg.VG)
                img = Image.fromarray(rng.randint(0, 255, size=(h, w, 3)).astype(np.uint8))
                ds.images_dir, ds.filenames, ds.mode = '', ['x.png'], 'train' if flip else 'test'
                nb = 5
                xy = rng.uniform(0, 600, size=(nb, 2))
                gt = np.concatenate((xy, xy + rng.uniform(10, 420, size=(nb, 2))), 1).astype(np.float32)   # BOX_SCALE space
                ds.gt_boxes, ds.gt_classes = [gt], [rng.randint(1, 12, size=nb)]
                ds.relationships = [np.array([[0, 1, 2], [3, 4, 1]])]
                ds.torch_detector, ds.filter_duplicate_rels, ds.rpn_rois = torch_detector, False, None
                ds.transform_pipeline = lambda im: torch.from_numpy(np.asarray(SquarePad()(im)).copy()).permute(2, 0, 1)
                vg.Image.open = lambda path: img
                entry = None
                for seed in range(40):                        # np.random.random() > 0.5 decides the flip in train mode
                    np.random.seed(seed)
                    entry = ds[0]
                    if entry['flipped'] == flip:
                        break
                assert entry['flipped'] == flip, (flip, entry['flipped'])
                k = 'g%d_' % n_entries
                arrs[k + 'wh'], arrs[k + 'torch_detector'], arrs[k + 'flip'] = np.array([w, h]), torch_detector, flip
                arrs[k + 'img_in'], arrs[k + 'gt_in'] = np.asarray(img), gt
                arrs[k + 'img_out'] = entry['img'].permute(1, 2, 0).numpy()
                arrs[k + 'gt_out'], arrs[k + 'im_size'], arrs[k + 'scale'] = entry['gt_boxes'], np.array(entry['img_size'], dtype=np.float64), entry['scale']
                n_entries += 1
    arrs['n_entries'] = n_entries
    save('vg_loader', **arrs)


class _CountData(object):
    """The attributes lib/get_dataset_counts.py reads from a VG split (dataloaders/visual_genome.py:377-455)."""

    def __init__(self, seed=7, n_img=24, n_cls=12, n_pred=9):
        rng = np.random.RandomState(seed)
        self.num_classes, self.num_predicates = n_cls, n_pred
        self.gt_classes, self.relationships, self.gt_boxes = [], [], []
        for i in range(n_img):
            nb = int(rng.randint(2, 9))
            if i == 3:        # an image whose boxes do not overlap at all: box_filter falls back to every pair (:58-59)
                boxes = np.array([[0, 0, 10, 10], [100, 100, 130, 140], [300, 20, 350, 60]], dtype=np.float32)
                nb = 3
            else:
                boxes = rand_boxes(rng, nb, hi=300)
            self.gt_boxes.append(boxes)
            self.gt_classes.append(rng.randint(1, n_cls, size=nb).astype(np.int64))
            pairs = np.array([(a, b) for a in range(nb) for b in range(nb) if a != b], dtype=np.int64)
            G = int(rng.randint(1, min(len(pairs), 6) + 1))
            gi = rng.choice(len(pairs), size=G, replace=False)
            self.relationships.append(np.column_stack((pairs[gi], rng.randint(1, n_pred, size=G))).astype(np.int64))

    def __len__(self):
        return len(self.gt_classes)


def gold_freq_bias():
    import torch.nn.functional as F
    np.float, np.bool = float, bool                               # lib/get_dataset_counts.py:52,55 use the removed aliases
    from lib.get_dataset_counts import get_counts
    from lib.sparse_targets import FrequencyBias
    data = _CountData()
    arrs = {'n_img': len(data), 'n_cls': data.num_classes, 'n_pred': data.num_predicates}
    for i in range(len(data)):
        arrs['boxes_%d' % i], arrs['classes_%d' % i], arrs['rels_%d' % i] = data.gt_boxes[i], data.gt_classes[i], data.relationships[i]
    for ov in (True, False):
        fg, bg = get_counts(data, must_overlap=ov)
        arrs['fg_%d' % ov], arrs['bg_%d' % ov] = fg, bg
    fb = FrequencyBias(data)
    arrs['table'] = fb.obj_baseline.weight
    # the use_bias block of RelModelStanford.forward (rel_model_stanford.py:159-177) on one two-image graph
    rng = np.random.RandomState(5)
    N, C, P = 11, data.num_classes, data.num_predicates
    im = np.array([0] * 6 + [1] * 5)
    rel_inds = np.array([(im[a], a, b) for a in range(N) for b in range(N) if a != b and im[a] == im[b]], dtype=np.int64)
    obj_dists = torch.from_numpy(rng.randn(N, C).astype(np.float32) * 2)
    rel_dists = torch.from_numpy(rng.randn(len(rel_inds), P).astype(np.float32))
    gt_classes = torch.from_numpy(np.column_stack((im, rng.randint(1, C, size=N))).astype(np.int64))
    rel_t = torch.from_numpy(rel_inds)
    arrs.update(obj_dists=obj_dists, rel_dists=rel_dists, rel_inds=rel_inds, gt_classes=gt_classes)
    for mode in ('sgcls', 'predcls'):
        for test_bias in (False, True):
            scores_nz = F.softmax(obj_dists, dim=1).data
            scores_nz[:, 0] = 0.0
            _, score_ord = scores_nz[:, 1:].sort(dim=1, descending=True)
            obj_preds = score_ord[:, 0] + 1
            if mode == 'predcls':
                obj_preds = gt_classes.data[:, 1]
            freq_pred = fb.index_with_labels(torch.stack((obj_preds[rel_t[:, 1]], obj_preds[rel_t[:, 2]]), 1))
            out = freq_pred if test_bias else rel_dists + freq_pred
            arrs['preds_%s' % mode] = obj_preds
            arrs['out_%s_%d' % (mode, test_bias)] = out
    # gradient of the embedding under a seeded upstream gradient (train mode, sgcls)
    fb.zero_grad()
    g = torch.from_numpy(rng.randn(len(rel_inds), P).astype(np.float32))
    scores_nz = F.softmax(obj_dists, dim=1).data
    obj_preds = scores_nz[:, 1:].argmax(1) + 1
    (fb.index_with_labels(torch.stack((obj_preds[rel_t[:, 1]], obj_preds[rel_t[:, 2]]), 1)) * g).sum().backward()
    arrs.update(d_out=g, d_table=fb.obj_baseline.weight.grad)
    save('freq_bias', **arrs)


if __name__ == '__main__':
    torch.set_num_threads(1)
    if len(sys.argv) > 1:                     # e.g. `make_golden.py freq_bias`: regenerate one file
        for name in sys.argv[1:]:
            globals()['gold_' + name]()
        sys.exit(0)
    gold_raster()
    gold_union_feats()
    gold_gru()
    gold_message_pass()
    gold_predict()
    gold_pairs()
    gold_rel_assign()
    gold_gan_ops()
    gold_gan_model()
    gold_sg_perturb()
    gold_eval_tail()
    gold_losses()
    gold_recall()
    gold_freq_bias()
    gold_vg_loader()
