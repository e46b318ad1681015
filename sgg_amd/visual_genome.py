"""Visual Genome input side of the path: the on-disk formats and the per-image arithmetic of
dataloaders/visual_genome.py, host-side, feeding `sgg_amd.blob` (Blob / vg_collate / DeviceStager).

  load_graphs           VG-SGG.h5 semantics (:516-659): split / box / relation tables -> per-image boxes, classes, relations, with the
                        reference's filters (empty, graph size, non-overlap, zero- / few-shot triplets)
  load_info             VG-SGG-dicts.json (:662-678)
  load_image_filenames  image_data.json (:491-513)
  filter_dups           one predicate per (subject, object) pair (:743-750)
  entry_geometry        the box / size arithmetic of VG.__getitem__ (:377-455)
  VG                    the dataset object main.py and lib/eval.py read (gt_boxes, gt_classes, relationships, filenames, ...)

`graphs` may be a path (needs h5py, which this image does not have: ImportError says so) or any mapping holding the file's datasets as
arrays.  Images stay decoded uint8 [h, w, 3]: SquarePad + ToTensor happen inside `sgg_image_prep_u8` on the GPU; `square_pad_u8` is the
host restatement used to check that kernel against dataloaders/image_transforms.py:8-13."""
import json
import os
from collections import OrderedDict

import numpy as np

BOX_SCALE = 1024   # config.py:30
IM_SCALE = 592     # config.py:31
PAD_FILL = (int(0.485 * 256), int(0.456 * 256), int(0.406 * 256))    # image_transforms.py:12
CORRUPTED = ('1592.jpg', '1722.jpg', '4616.jpg', '4617.jpg')          # visual_genome.py:503


def _datasets(graphs, names):
    if isinstance(graphs, (str, bytes, os.PathLike)):
        # the reference reads the file with h5py (dataloaders/visual_genome.py:536); this image has none, so the path's own
        # read-only HDF5 reader is used -- same datasets, same dtypes (tests/test_hdf5_lite_cpu.py: files written by h5py)
        from .hdf5_lite import File
        with File(graphs) as f:
            return {k: f[k][:] for k in names if k in f}
    return {k: np.array(graphs[k][:]) for k in names}


def _iou_positive(boxes):
    b = np.asarray(boxes, dtype=np.float32)
    lt = np.maximum(b[:, None, :2], b[None, :, :2])
    rb = np.minimum(b[:, None, 2:], b[None, :, 2:])
    wh = np.clip(rb - lt, 0, None)
    return wh[..., 0] * wh[..., 1] > 0          # IoU > 0 <=> intersection area > 0 (areas are positive, :563)


def load_graphs(graphs, mode='train', num_im=-1, num_val_im=0, filter_empty_rels=True, min_graph_size=-1, max_graph_size=-1,
                filter_non_overlap=False, training_triplets=None, random_subset=False, filter_zeroshots=True, n_shots=-1):
    """-> (split_mask bool[num_images], boxes [list of i32[n,4] x1y1x2y2 at BOX_SCALE], gt_classes [list of [n]],
    relationships [list of [r,3] = (subj, obj, predicate), image-local box indices])"""
    if mode not in ('train', 'val', 'test'):
        raise ValueError('{} invalid'.format(mode))
    d = _datasets(graphs, ('split', 'img_to_first_box', 'img_to_last_box', 'img_to_first_rel', 'img_to_last_rel',
                           'boxes_{}'.format(BOX_SCALE), 'labels', 'relationships', 'predicates'))
    keep = (d['split'] == (2 if mode == 'test' else 0)) & (d['img_to_first_box'] >= 0)       # :541-546
    if filter_empty_rels:
        keep &= d['img_to_first_rel'] >= 0
    image_index = np.where(keep)[0]
    if num_im > -1:
        image_index = image_index[:num_im]
    if num_val_im > 0:                                                                       # :551-555
        if mode == 'val':
            image_index = image_index[:num_val_im]
        elif mode == 'train':
            image_index = image_index[num_val_im:]
    split_mask = np.zeros(d['split'].shape, dtype=bool)
    split_mask[image_index] = True
    labels = d['labels'][:, 0]
    boxes = d['boxes_{}'.format(BOX_SCALE)]
    assert np.all(boxes[:, :2] >= 0) and np.all(boxes[:, 2:] > 0)
    # (xc, yc, w, h) -> (x1, y1, x2, y2) in the table's own integer type: the half-size subtraction truncates (:566-567)
    boxes[:, :2] = boxes[:, :2] - boxes[:, 2:] / 2
    boxes[:, 2:] = boxes[:, :2] + boxes[:, 2:]
    first_box, last_box = d['img_to_first_box'][image_index], d['img_to_last_box'][image_index]
    first_rel, last_rel = d['img_to_first_rel'][image_index], d['img_to_last_rel'][image_index]
    rel_pairs, rel_preds = d['relationships'], d['predicates'][:, 0]
    assert rel_pairs.shape[0] == rel_preds.shape[0]
    few_shot = isinstance(training_triplets, dict)
    out_boxes, out_classes, out_rels = [], [], []
    for i, img in enumerate(image_index):
        b_i = boxes[first_box[i]:last_box[i] + 1, :]
        c_i = labels[first_box[i]:last_box[i] + 1]
        if (min_graph_size > -1 and len(c_i) <= min_graph_size) or (max_graph_size > -1 and len(c_i) > max_graph_size):
            split_mask[img] = False                                                          # :586-592
            continue
        picked = []
        if first_rel[i] >= 0:
            local = rel_pairs[first_rel[i]:last_rel[i] + 1] - first_box[i]
            assert np.all(local >= 0) and np.all(local < b_i.shape[0])
            rels = np.column_stack((local, rel_preds[first_rel[i]:last_rel[i] + 1]))
            if training_triplets and len(rels) > 0:                                         # :603-633
                if random_subset:
                    picked = np.random.permutation(len(rels))[:int(np.round(len(rels) / 15.))]
                else:
                    for r, (o1, o2, pred) in enumerate(rels):
                        key = '{}_{}_{}'.format(c_i[o1], pred, c_i[o2])
                        if few_shot:
                            assert n_shots > 0, n_shots
                            cnt = training_triplets.get(key)
                            if cnt is not None and ((n_shots == 10 and 1 <= cnt <= n_shots) or (n_shots == 100 and 11 <= cnt <= n_shots)):
                                picked.append(r)
                        elif key not in training_triplets:
                            assert n_shots == -1, n_shots
                            picked.append(r)
                    picked = np.array(picked)
                if filter_zeroshots:
                    rels = rels[picked] if len(picked) > 0 else np.zeros((0, 3), dtype=np.int32)
        else:
            assert not filter_empty_rels
            rels = np.zeros((0, 3), dtype=np.int32)
        if training_triplets and filter_zeroshots:
            assert len(rels) == len(picked), (len(rels), len(picked))
        if training_triplets and filter_empty_rels and len(picked) == 0:                     # :641-643
            split_mask[img] = False
            continue
        if filter_non_overlap:                                                               # :645-655
            assert mode == 'train'
            inc = np.where(_iou_positive(b_i)[rels[:, 0], rels[:, 1]])[0]
            if inc.size == 0:
                split_mask[img] = False
                continue
            rels = rels[inc]
        out_boxes.append(b_i)
        out_classes.append(c_i)
        out_rels.append(rels)
    return split_mask, out_boxes, out_classes, out_rels


def load_info(info):
    """VG-SGG-dicts.json (path or the parsed dict) -> (ind_to_classes, ind_to_predicates), background at index 0 (:662-678)"""
    if not isinstance(info, dict):
        with open(info, 'r') as f:
            info = json.load(f)
    c2i, p2i = dict(info['label_to_idx']), dict(info['predicate_to_idx'])
    c2i['__background__'] = 0
    p2i['__background__'] = 0
    return sorted(c2i, key=lambda k: c2i[k]), sorted(p2i, key=lambda k: p2i[k])


def load_image_filenames(image_file, image_dir, expected=108073, check_exists=True):
    """image_data.json (path or parsed list) -> basenames of the usable images, in file order (:491-513)"""
    if isinstance(image_file, (str, bytes, os.PathLike)):
        with open(image_file, 'r') as f:
            image_file = json.load(f)
    fns = []
    for img in image_file:
        base = '{}.jpg'.format(img['image_id'])
        if base in CORRUPTED:
            continue
        if not check_exists or os.path.exists(os.path.join(image_dir, base)):
            fns.append(base)
    if expected is not None and len(fns) != expected:
        raise AssertionError(len(fns))
    return fns


def filter_dups(gt_rels, random_edge=True):
    """one predicate per (subject, object) pair, pairs in order of first appearance (:743-750)"""
    by_pair = OrderedDict()
    for o0, o1, r in gt_rels:
        by_pair.setdefault((o0, o1), []).append(r)
    return np.array([(k[0], k[1], np.random.choice(v) if random_edge else v[0]) for k, v in by_pair.items()])


def square_pad_u8(img_hwc):
    """SquarePad (image_transforms.py:8-13): pad right / bottom to a square with the ImageNet mean colour."""
    h, w = img_hwc.shape[:2]
    s = max(h, w)
    out = np.empty((s, s, 3), dtype=np.uint8)
    out[...] = np.array(PAD_FILL, dtype=np.uint8)
    out[:h, :w] = img_hwc
    return out


def entry_geometry(w, h, gt_boxes, torch_detector, flipped, split='stanford'):
    """VG.__getitem__ (:383-430) for an image of size (w, h): boxes from the BOX_SCALE tables to the scale the model expects,
    clipped to the image, optionally mirrored.  -> (gt_boxes, img_size = (h', w', factor), scale)"""
    gt_boxes = np.array(gt_boxes, copy=True)
    if torch_detector:
        im_scale = box_scale = max(w, h)
    else:
        im_scale, box_scale = IM_SCALE, BOX_SCALE
    box_scale_factor = box_scale / max(w, h)
    if split in ('vte', 'gqa'):
        gt_boxes = gt_boxes * box_scale_factor
    elif torch_detector:
        gt_boxes = gt_boxes / (BOX_SCALE / max(w, h))           # tables are at BOX_SCALE: bring them to image scale
    gt_boxes[:, [1, 3]] = gt_boxes[:, [1, 3]].clip(None, box_scale / max(w, h) * h)
    gt_boxes[:, [0, 2]] = gt_boxes[:, [0, 2]].clip(None, box_scale / max(w, h) * w)
    if split in ('vte', 'gqa'):                                 # zero extent after clipping (:411-416, operator precedence kept)
        z = (gt_boxes[:, 2] - gt_boxes[:, 0]) == 0 & (gt_boxes[:, 0] > 0)
        gt_boxes[z, 0] -= 1
        z = (gt_boxes[:, 3] - gt_boxes[:, 1]) == 0 & (gt_boxes[:, 1] > 0)
        gt_boxes[z, 1] -= 1
    if flipped:
        gt_boxes[:, [0, 2]] = int(box_scale_factor * float(w)) - gt_boxes[:, [2, 0]]
    f = im_scale / max(w, h)
    if h > w:
        im_size = (im_scale, int(w * f), f)
    elif h < w:
        im_size = (int(h * f), im_scale, f)
    else:
        im_size = (im_scale, im_scale, f)
    return gt_boxes, im_size, im_scale / box_scale


def _decode(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert('RGB'))


class VG(object):
    """The attributes and items the training / evaluation loops read from dataloaders/visual_genome.py:VG, built from already
    opened tables.  `graphs`: path or mapping (load_graphs); `info`: path or dict (load_info); `filenames`: basenames of ALL images in
    table order (load_image_filenames) -- the split mask selects from them (:172-173).  Items keep decoded uint8 images
    (`decode(path) -> u8[h,w,3]`, PIL by default); torch_detector=True is the path's setting (main.py)."""

    split = 'stanford'

    def __init__(self, mode, graphs, info, filenames, images_dir='', filter_empty_rels=True, num_im=-1, num_val_im=5000,
                 filter_duplicate_rels=True, filter_non_overlap=True, min_graph_size=-1, max_graph_size=-1, torch_detector=True,
                 n_shots=-1, training_triplets=None, decode=None):
        assert mode in ('test', 'train', 'val'), '%s mode not recognized' % mode
        self.mode, self.images_dir, self.torch_detector = mode, images_dir, torch_detector
        self.filter_duplicate_rels = filter_duplicate_rels and mode == 'train'            # :63
        self.filter_non_overlap = filter_non_overlap and mode == 'train'
        self.split_mask, self.gt_boxes, self.gt_classes, self.relationships = load_graphs(
            graphs, mode, num_im, num_val_im=num_val_im, filter_empty_rels=filter_empty_rels,
            min_graph_size=min_graph_size if mode == 'train' else -1, max_graph_size=max_graph_size if mode == 'train' else -1,
            filter_non_overlap=self.filter_non_overlap, training_triplets=training_triplets, n_shots=n_shots)
        self.filenames = [fn for fn, keep in zip(filenames, self.split_mask) if keep]
        self.ind_to_classes, self.ind_to_predicates = load_info(info)
        self.decode = decode or _decode
        self.rpn_rois = None
        self.triplet_counts = {}
        for cls, rels in zip(self.gt_classes, self.relationships):                          # :214-226
            for s, o, p in rels:
                key = '{}_{}_{}'.format(cls[s], p, cls[o])
                self.triplet_counts[key] = self.triplet_counts.get(key, 0) + 1
        if mode == 'train':
            # what the scene-graph perturbations consult (visual_genome.py:211-227, augment/sg_perturb.py): for "subject predicate"
            # the objects seen with it, for "predicate object" the subjects, each with the count of the whole triplet; first-seen order
            self.subj_pred_pairs, self.pred_obj_pairs = {}, {}
            for cls, rels in zip(self.gt_classes, self.relationships):
                for s, o, p in rels:
                    count = self.triplet_counts['{}_{}_{}'.format(cls[s], p, cls[o])]
                    self.subj_pred_pairs.setdefault('{}_{}'.format(cls[s], p), {})[cls[o]] = count
                    self.pred_obj_pairs.setdefault('{}_{}'.format(p, cls[o]), {})[cls[s]] = count

    @property
    def is_train(self):
        return self.mode.startswith('train')

    @property
    def num_predicates(self):
        return len(self.ind_to_predicates)

    @property
    def num_classes(self):
        return len(self.ind_to_classes)

    def triplet2str(self, triplet):
        s, p, o = [int(t) for t in triplet.split('_')]
        return '{} {} {}'.format(self.ind_to_classes[s], self.ind_to_predicates[p], self.ind_to_classes[o])

    def __len__(self):
        return len(self.filenames)

    def __getitem__(self, index):
        img = self.decode(os.path.join(self.images_dir, self.filenames[index]))
        h, w = img.shape[:2]
        flipped = self.is_train and np.random.random() > 0.5
        gt_boxes, im_size, scale = entry_geometry(w, h, self.gt_boxes[index], self.torch_detector, flipped, self.split)
        if flipped:
            img = np.ascontiguousarray(img[:, ::-1])
        gt_rels = self.relationships[index].copy()
        if self.filter_duplicate_rels:
            gt_rels = filter_dups(gt_rels)
        entry = {'img': img, 'img_size': im_size, 'gt_boxes': gt_boxes, 'gt_classes': self.gt_classes[index].copy(),
                 'gt_relations': gt_rels, 'scale': scale, 'index': index, 'flipped': flipped, 'fn': self.filenames[index]}
        if self.rpn_rois is not None:
            entry['proposals'] = self.rpn_rois[index]
        if entry['gt_classes'].shape[0] != gt_boxes.shape[0]:                                # assertion_checks, :474-488
            raise ValueError('GT classes and GT boxes must have same number of examples')
        assert (gt_boxes[:, 2] >= gt_boxes[:, 0]).all() and (gt_boxes >= -1).all()
        return entry
