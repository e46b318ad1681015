"""Relation labels for SGDet training: detections -> sampled (img, subj, obj, predicate) rows.

Mirror of `lib/rel_assignments.py:12-137` (called at `sgg_models/rel_model_stanford.py:138-140` with
`filter_non_overlap=True, num_sample_per_gt=1`).  The IoU / match / candidate tables of all images come from ONE launch
(`sgg_rel_assign_tables`) and one D2H copy.  The sampling itself draws from numpy's global RandomState in the reference
(`npr.choice` :97,:103; `np.random.choice` :114) and a run seeded with `np.random.seed` must give the same rows, so the draws stay
on the host and are made with the reference's arguments in the reference's order:
  per GT relation with candidates   choice(n_cand, p=iou_s*iou_o / sum, size=min(n_cand, num_sample_per_gt), replace=False)
  per image, when > 16 FG rows      choice(n_fg, size=16, replace=False)
  per image, when BG candidates     choice(n_bg, size=min(64 - n_fg, n_bg), replace=False)      (also drawn when the size is 0)
"""
import numpy as np
import torch

from . import ops

REL_FG_FRACTION = 0.25      # config.py:33
RELS_PER_IMG_SGDET = 64     # the literal 64 of lib/rel_assignments.py:27,110


def rel_assignments(im_inds, rpn_rois, roi_gtlabels, gt_boxes, gt_classes, gt_rels, image_offset=0, fg_thresh=0.5,
                    num_sample_per_gt=4, filter_non_overlap=True):
    """im_inds i64[N], rpn_rois f32[N,4] (the reference passes boxes without the image column, :138), roi_gtlabels i64[N],
    gt_boxes f32[G,4], gt_classes i64[G,2]=(img,cls), gt_rels i64[R,4]=(img,subj_local,obj_local,pred)
    -> rel_labels i64[R',4] on the device of rpn_rois, rows sorted by (subj, obj) inside each image."""
    dev = rpn_rois.device
    fg_per_image = int(np.round(REL_FG_FRACTION * RELS_PER_IMG_SGDET))                       # :27
    gt_classes = gt_classes.to(dev).contiguous()
    if image_offset:
        gt_classes = gt_classes.clone()
        gt_classes[:, 0] -= image_offset                                                     # :36
    im_inds = im_inds.to(dev).long().contiguous()
    gt_iou, match, poss = ops.rel_assign_tables(rpn_rois.float().contiguous(), im_inds, roi_gtlabels.to(dev).long().contiguous(),
                                                gt_boxes.to(dev).float().contiguous(), gt_classes, fg_thresh, filter_non_overlap)
    N, G = gt_iou.shape
    # one D2H for everything the sampler reads (u8 tables travel as they are; the i64 columns are small)
    host = [t.cpu().numpy() for t in (gt_iou, match, poss, im_inds, gt_classes)]
    gt_iou, match, poss, det_img, gtc = host[0], host[1].astype(bool), host[2].astype(bool), host[3], host[4]
    rels = (gt_rels.cpu().numpy() if isinstance(gt_rels, torch.Tensor) else np.asarray(gt_rels)).astype(np.int64).copy()
    rels[:, 0] -= image_offset                                                               # :37
    num_im = int(gtc[:, 0].max()) + 1                                                        # :39
    out, seen = [], 0
    for im in range(num_im):
        det = np.flatnonzero(det_img == im)                                                  # :48
        gts = np.flatnonzero(gtc[:, 0] == im)                                                # :50
        iou_i = gt_iou[np.ix_(det, gts)]
        match_i = match[np.ix_(det, gts)]
        poss_i = poss[np.ix_(det, det)].copy()
        fg = []
        for s_gt, o_gt, pred in rels[rels[:, 0] == im, 1:]:                                  # :53,:81
            subs, objs = np.flatnonzero(match_i[:, s_gt]), np.flatnonzero(match_i[:, o_gt])
            a, b = np.repeat(subs, len(objs)), np.tile(objs, len(subs))                      # subject-major, as the nested loops :85-86
            keep = a != b
            a, b = a[keep], b[keep]
            if a.size == 0:
                continue
            poss_i[a, b] = False                                                             # :90
            p = iou_i[a, s_gt] * iou_i[b, o_gt]                                              # :89 (fp32 products)
            p = p / p.sum()                                                                  # :94-95
            for k in np.random.choice(p.shape[0], p=p, size=min(p.shape[0], num_sample_per_gt), replace=False):   # :97
                fg.append((a[k], b[k], pred))
        fg = np.array(fg, dtype=np.int64).reshape(-1, 3)
        if fg.shape[0] > fg_per_image:
            fg = fg[np.random.choice(fg.shape[0], size=fg_per_image, replace=False)]         # :102-103
        bs, bo = np.nonzero(poss_i)                                                          # :107 (row-major)
        n_bg = min(RELS_PER_IMG_SGDET - fg.shape[0], bs.shape[0])                            # :110
        if bs.shape[0] > 0:
            pick = np.random.choice(bs.shape[0], size=n_bg, replace=False)                   # :114-116
            bg = np.stack((bs[pick], bo[pick], np.zeros(n_bg, dtype=np.int64)), 1)
        else:
            bg = np.zeros((0, 3), dtype=np.int64)
        if fg.shape[0] == 0 and bg.shape[0] == 0:
            bg = np.zeros((1, 3), dtype=np.int64)                                            # :120-122 "just put something here"
        rows = np.concatenate((fg, bg), 0)
        rows[:, 0:2] += seen                                                                 # :126
        rows = rows[np.lexsort((rows[:, 1], rows[:, 0]))]                                    # :128 (stable)
        out.append(np.concatenate((np.full((rows.shape[0], 1), im, dtype=np.int64), rows), 1))
        seen += det.shape[0]                                                                 # :135
    return torch.from_numpy(np.concatenate(out, 0)).to(dev, non_blocking=True)               # :136-137
