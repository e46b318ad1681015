"""Scene-graph perturbations of the GAN training iteration (augment/sg_perturb.py:7-191, used at main.py:81-90,131-134): pick a share
L of every image's nodes (probability ~ node degree) and replace their class

    'rand'    by any other foreground class,
    'neigh'   by one of the top-k nearest classes in the word-embedding space,
    'graphn'  by a class that is plausible in the node's triplets according to the training statistics (`subj_pred_pairs` /
              `pred_obj_pairs` of the dataset: what else occurs as "<x> on surfboard", "wave near <x>"), weighted towards RARE
              compositions, optionally followed by a hop to one of its top-k embedding neighbours.

Host-side sampling logic.  The random draws are made with the reference's generators, arguments and order -- per image one
`np.random.choice(n, size, replace=False, p)` for the nodes, per node a `torch.randperm(len(candidates))[0]` (and, for 'graphn', one
`np.random.choice(candidates, p)` before it) -- so the same seeds give the same perturbed graphs as the reference
(tests/golden/sg_perturb.npz).  The word vectors themselves (GloVe, lib/word_vectors.py) are not carried: pass `embed_objs`
(rows = classes) for 'neigh' and for 'graphn' with topk > 0.
"""
import numpy as np
import torch


def pairwise_similarity(embed_objs):
    """augment/sg_perturb.py:179-184: cosine-style similarity table with background and self excluded (-inf)"""
    sim = torch.mm(embed_objs, embed_objs.t())
    sim[0, :] = -np.inf
    sim[:, 0] = -np.inf
    sim.fill_diagonal_(-np.inf)
    return sim


def random_choice(tensor):
    """augment/sg_perturb.py:187-188"""
    return tensor[torch.randperm(len(tensor))[0]].item()


def _segments(column):
    """[(start, end)] of the runs of equal values in a sorted 1-D tensor (the images of a batch)"""
    vals = column.tolist()
    cuts = [0] + [i for i in range(1, len(vals)) if vals[i] != vals[i - 1]] + [len(vals)]
    return [(cuts[i], cuts[i + 1]) for i in range(len(cuts) - 1)] if vals else []


class SceneGraphPerturb(object):
    def __init__(self, method, embed_objs, subj_pred_obj_pairs, L=0.2, topk=5, alpha=2, uniform=False, degree_smoothing=1,
                 data_dir=None, obj_classes=None, triplet2str=None):
        self.method = method
        needs_vectors = method == 'neigh' or (method == 'graphn' and topk > 0)
        if embed_objs is None and needs_vectors:
            raise ValueError("SceneGraphPerturb(%r, topk=%d) needs embed_objs (the GloVe tables of lib/word_vectors.py are not carried)" %
                             (method, topk))
        self.obj_pairwise = pairwise_similarity(embed_objs) if embed_objs is not None else None
        self.subj_pred_obj_pairs = subj_pred_obj_pairs          # (subj_pred_pairs, pred_obj_pairs) of the training set
        self.L, self.topk, self.alpha, self.uniform, self.degree_smoothing = L, topk, alpha, uniform, degree_smoothing
        self.obj_classes, self.triplet2str = obj_classes, triplet2str
        if self.obj_pairwise is not None:
            self.n_obj_classes = self.obj_pairwise.shape[0]
        elif obj_classes is not None:
            self.n_obj_classes = len(obj_classes)
        else:
            raise ValueError('SceneGraphPerturb: the number of classes comes from embed_objs or obj_classes')
        if method not in ('rand', 'neigh', 'graphn'):
            raise NotImplementedError(method)
        if method == 'neigh':
            assert topk > 0, topk

    # ------------------------------------------------------------------ which nodes
    def sample_nodes_(self, gt_obj_lst, gt_rels_lst):
        """{image: (node indices, [the foreground relations touching each])}; one np.random.choice per image (augment/sg_perturb.py:144-176)"""
        picked = {}
        for im, (objs, rels) in enumerate(zip(gt_obj_lst, gt_rels_lst)):
            n = len(objs)
            fg = rels[rels[:, -1] > 0]
            if self.L <= 0:
                picked[im] = (np.empty(0), [])
                continue
            if self.uniform:
                p = np.ones(n, dtype=np.float32)
            else:
                # degree = foreground relations with the node as subject or object (a self-loop counts once)
                touch = (fg[:, 1:3, None] == torch.arange(n)[None, None, :]).any(1)                  # [R, n]
                p = (touch.sum(0).numpy().astype(np.float32) ** self.degree_smoothing).clip(1e-2, None)
            p = p / np.sum(p)
            k = max(1, int(np.round(self.L * n)))
            nodes = np.random.choice(np.arange(n), size=k, replace=False, p=p)
            picked[im] = (nodes, [fg[(fg[:, 1] == int(i)) | (fg[:, 2] == int(i))] for i in nodes])
            if self.L >= 1:
                assert len(nodes) == n == k, (len(nodes), n, k, p)
        return picked

    # ------------------------------------------------------------------ which class
    def _graph_candidates(self, objs, rels, ind, cls):
        """classes seen in the node's place in at least min(#relations, 2) of its triplets (each at least alpha times), and the mean
        training frequency of those triplets -- in first-seen order, as the reference's dicts iterate"""
        seen = {}
        subj_pred, pred_obj = self.subj_pred_obj_pairs
        for _, o1, o2, R in rels.tolist():
            assert ind in (o1, o2), (ind, o1, o2, R)
            if ind == o1:
                table, key = pred_obj, '{}_{}'.format(R, int(objs[o2, 1]))       # what else is "<x> R o2"?
            else:
                table, key = subj_pred, '{}_{}'.format(int(objs[o1, 1]), R)       # what else is "o1 R <x>"?
            for obj, freq in table.get(key, {}).items():
                if obj != cls:
                    seen.setdefault(obj, []).append(freq)
        need = max(1, min(len(rels), 2))
        cands = [(obj, float(np.mean(f))) for obj, f in seen.items() if len(f) >= need and np.min(f) >= self.alpha]
        return [c for c, _ in cands], [m for _, m in cands]

    def perturb_object_(self, gt_objs, gt_rels, ind, verbose=False):
        cls = int(gt_objs[ind, 1].item())
        if self.method == 'rand':
            new = random_choice(torch.cat((torch.arange(1, cls), torch.arange(cls + 1, self.n_obj_classes))))
        elif self.method == 'neigh':
            new = random_choice(torch.argsort(self.obj_pairwise[cls])[-self.topk:])
        else:
            cands, freq = self._graph_candidates(gt_objs, gt_rels, ind, cls)
            if cands:
                p = 1.0 / np.array(freq)                         # rare compositions are preferred
                new = np.random.choice(cands, p=p / np.sum(p))
                assert new not in (0, cls), (new, cls)
            else:
                new = cls
            if self.topk > 0:                                    # hop to a semantic neighbour of the choice (or stay on it)
                row = self.obj_pairwise[new].clone()
                row[new] = np.inf
                row[cls] = -np.inf
                new = random_choice(torch.argsort(row)[-self.topk - 1:])
        if not (self.method == 'graphn' and self.topk == 0):
            assert new not in (0, cls), (new, cls)
        return new

    def perturb(self, gt_obj, gt_rels, verbose=False):
        """gt_obj i64[N,2] (image, class), gt_rels i64[R,4] (image, subject, object, predicate; image-local node indices), both sorted
        by image -> the objects with the sampled nodes' classes replaced (the input rows are modified in place, as in the reference:
        main.py passes clones)."""
        obj_lst = [gt_obj[s:e] for s, e in _segments(gt_obj[:, 0])]
        rel_lst = [gt_rels[s:e] for s, e in _segments(gt_rels[:, 0])]
        nodes = self.sample_nodes_(obj_lst, rel_lst)
        for im, objs in enumerate(obj_lst):
            for ind, rels in zip(*nodes[im]):
                objs[ind, 1] = self.perturb_object_(objs, rels, int(ind), verbose=verbose)
        return torch.cat(obj_lst)
