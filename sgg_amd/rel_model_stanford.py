"""RelModelStanford (IMP) on the HIP path -- mirror of sgg_models/rel_model_stanford.py."""
import os

import torch
import torch.nn as nn

from . import _lib, ops
from .imp import GATES, ImpWeights, message_pass, node_lane
from .pairing import PairedEdgeFeat, make_pairing, make_pairing_symmetric
from .rel_assignments import rel_assignments
from .rel_model_base import RelModelBase, to_device_with_mirror, to_rows


class RelModelStanford(RelModelBase):
    """Message Passing Model from "Scene Graph Generation by Iterative Message Passing" (rel_model_stanford.py:11-45).
    Same parameters / names as the reference, so `vgrel.pth` checkpoints load with load_state_dict."""

    def __init__(self, train_data, hidden_dim=512, mp_iter=3, **kwargs):
        super(RelModelStanford, self).__init__(train_data, **kwargs)
        self.hidden_dim = hidden_dim
        self.rel_fc = nn.Linear(hidden_dim, self.num_rels)
        self.obj_fc = nn.Linear(hidden_dim, self.num_classes)
        self.obj_unary = nn.Linear(self.obj_dim, hidden_dim)
        self.edge_unary = nn.Linear(self.obj_dim, hidden_dim)
        self.edge_gru = nn.GRUCell(input_size=hidden_dim, hidden_size=hidden_dim)
        self.node_gru = nn.GRUCell(input_size=hidden_dim, hidden_size=hidden_dim)
        self.mp_iter = mp_iter
        self.dropout_p = 0.5   # nn.Dropout() of the VGG classifier copies (rel_model_base.py:110-111)
        for g in GATES:
            setattr(self, g, nn.Sequential(nn.Linear(hidden_dim * 2, 1), nn.Sigmoid()))

    # ------------------------------------------------------------------ weights in kernel layout
    def head_named_parameters(self):
        """[(name, parameter)] of everything but the detector, cached: walking the module tree (named_parameters /
        state_dict) costs ~1 ms of Python per call, and the train step asks several times."""
        cache = self.__dict__.get('_head_named')
        if cache is None:
            cache = [(n, p) for n, p in self.named_parameters() if not n.startswith('detector.')]
            self.__dict__['_head_named'] = cache
        return cache

    def prepared(self):
        """Device operands derived from the fp32 master parameters, cached until a parameter changes: casts to the
        compute dtype, plus W6sum[n,c] = sum_p W6[n,c,p] -- the 512 extra K columns that fold `union_pools + conv(rects)`
        (lib/get_union_boxes.py:101) into fc6 by linearity: fc6(x + r (x) 1_49) = fc6(x) + W6sum r."""
        ev = getattr(self, '_operands_ready', None)     # a trainer may have queued an update / rebuild on its side stream
        if ev is not None:
            torch.cuda.current_stream(self.rel_fc.weight.device).wait_event(ev)
            self._operands_ready = None
        dtype = self.compute_dtype
        params = [p for _, p in self.head_named_parameters()]
        key = (dtype, getattr(self, 'weights_version', 0)) + tuple((p.data_ptr(), p._version) for p in params)
        if self._prep.get('key') == key:
            return self._prep['val']
        C, PP = self.edge_dim, self.pool_sz ** 2
        f = lambda t: t.detach().float().contiguous()
        w = {}
        fc, _ = self.fc_layers()
        w['fc6_obj'] = self._shadow_cast(fc['fc6_obj'][0] + '.weight', fc['fc6_obj'][1].weight)
        w6e = f(fc['fc6_edge'][1].weight)
        w['fc6_edge'] = self._shadow_cast(fc['fc6_edge'][0] + '.weight', w6e)   # [obj_dim, C*49], K order (c,ph,pw) as in the reference
        gs = getattr(self, '_sharded_group_sum', None)             # a trainer with a sharded optimiser: sums of this rank's rows, gathered
        if gs is not None:
            w['fc6_edge_sum'] = gs(w6e, C, PP, dtype)                    # [obj_dim, C]
        else:
            # rewritten in place (like the shadows and the transposed copies of train_weights): a derived operand keeps its address for as
            # long as the compute dtype stays, so that launch sequences replayed as hipGraphs keep reading the current values
            w['fc6_edge_sum'] = self._sum_buf = ops.group_sum(w6e, C, PP, dtype, out=self.__dict__.get('_sum_buf'))
        for name, pname, mod in (('fc7_obj',) + fc['fc7_obj'], ('fc7_edge',) + fc['fc7_edge'],
                                 ('obj_unary', 'obj_unary', self.obj_unary), ('edge_unary', 'edge_unary', self.edge_unary),
                                 ('obj_fc', 'obj_fc', self.obj_fc), ('rel_fc', 'rel_fc', self.rel_fc)):
            w[name] = self._shadow_cast(pname + '.weight', mod.weight)
            w[name + '_b'] = f(mod.bias)
        w['fc6_obj_b'] = f(fc['fc6_obj'][1].bias)
        w['fc6_edge_b'] = f(fc['fc6_edge'][1].bias)
        sd = {k: v for k, v in self.head_named_parameters() if 'gru' in k or 'w_fc' in k}
        w['imp'] = ImpWeights.from_state(sd, dtype, cast=self._shadow_cast)
        self._prep = dict(key=key, val=w)
        return w

    # Compute-dtype copies of the plain-cast weights live in stable buffers: the fused optimiser writes them in its
    # update pass (sgg_sgd_multi `shadow`), so a train step never re-reads 1 GB of fp32 masters just to cast them.

    def _shadow_tag(self, p):
        return (getattr(self, 'weights_version', 0), p.data_ptr(), p._version)

    def _shadow_cast(self, pname, p):
        dtype = self.compute_dtype
        src = p.detach().float().contiguous()
        if dtype == torch.float32:
            return src
        sh = self._shadow.get(pname)
        if sh is None or sh.shape != src.shape or sh.dtype != dtype or sh.device != src.device:
            sh = self._shadow[pname] = torch.empty(src.shape, dtype=dtype, device=src.device)
            self._shadow_tags.pop(pname, None)
        tag = self._shadow_tag(p)
        if self._shadow_tags.get(pname) != tag:
            ops.cast(src, dtype, out=sh)
            self._shadow_tags[pname] = tag
        return sh

    def shadow_buffers(self):
        """{param name: 16-bit buffer} the optimiser may refresh in place of a later cast (empty in fp32 mode)."""
        if not ops.is_half(self.compute_dtype):
            return {}
        return {n: t for n, t in self._shadow.items() if t.dtype == self.compute_dtype}

    def mark_shadow_fresh(self, names):
        """The optimiser just wrote these shadows from the updated masters (call after weights_version moved)."""
        params = dict(self.head_named_parameters())
        for n in names:
            if n in params and n in self._shadow:
                self._shadow_tags[n] = self._shadow_tag(params[n])

    # ------------------------------------------------------------------ reference API
    def message_pass(self, rel_rep, obj_rep, rel_inds):
        """rel_model_stanford.py:48-94.  rel_inds i64[E,2] = (subj, obj)."""
        w = self.prepared()
        dtype = self.compute_dtype
        ri3 = torch.cat((rel_inds.new_zeros((rel_inds.shape[0], 1)), rel_inds), 1).contiguous()
        csr = ops.edge_csr(ri3, obj_rep.shape[0])
        cast = lambda t: t.contiguous() if t.dtype == dtype else ops.cast(t.float() if t.dtype not in
                                                                          (torch.float32, torch.bfloat16, torch.float16) else t, dtype)
        return message_pass(cast(rel_rep), cast(obj_rep), ri3, csr, w['imp'], self.mp_iter, dtype)

    def predict(self, node_feat, edge_feat, rel_inds, rois, im_sizes, _im_inds=None, _graphs=None, _csr=None):
        """rel_model_stanford.py:97-107.  node_feat [N,C,7,7], edge_feat [E,C,7,7] (raw RoIAlign), rel_inds i64[E,3]
        -> (obj_dists f32[N,151], rel_dists f32[E,51])."""
        dtype = self.compute_dtype
        N, E = node_feat.shape[0], edge_feat.shape[0]
        rel_inds = rel_inds.contiguous()
        nf = to_rows(node_feat.view(N, -1, self.pool_sz, self.pool_sz), dtype)
        paired = edge_feat.pairing if isinstance(edge_feat, PairedEdgeFeat) else None
        if paired is not None:       # rows of the unordered pairs [U, C*P*P]; e2u maps the edges onto them
            ef = to_rows(edge_feat.rows.view(paired.U, -1, self.pool_sz, self.pool_sz), dtype)
        else:
            ef = to_rows(edge_feat.view(E, -1, self.pool_sz, self.pool_sz), dtype)
        if self.training:
            # Dropout, batch-statistic BatchNorm and the autograd node of the whole head (sgg_amd/train.py); the TwoMLPHead copies of
            # the resnet50 configuration have no Dropout layers (the VGG classifier's sit after fc6 and fc7)
            from .train import predict_train
            return predict_train(self, nf, ef, rel_inds, rois, _im_inds, dropout_p=self.dropout_p if self.backbone == 'vgg16' else 0.0,
                                 graphs=_graphs, im_sizes=im_sizes, pairing=paired, csr=_csr, seed=self.__dict__.get('_seed_dev'))
        w = self.prepared()
        # :100  union_boxes(edge_feat, rois, rel_inds[:,1:]) -- conv(rects)[E,512]; the broadcast add rides in fc6's K
        rect = self.union_boxes.rect_feat(rois, ops.pairs_of(rel_inds), dtype, im_sizes)
        # :103  obj_unary(roi_fmap_obj(node_feat)) -- three short-M GEMMs (256 rows), latency-bound: on the node lane's
        # stream they run under the edge MLP below instead of in front of it (message_pass keeps using that lane)
        def node_mlp():
            _lib.set_tag('fc6_obj')
            x = ops.gemm(nf, w['fc6_obj'], w['fc6_obj_b'], ops.ACT_RELU)
            _lib.set_tag('fc7_obj')
            x = ops.gemm(x, w['fc7_obj'], w['fc7_obj_b'], ops.ACT_RELU)
            _lib.set_tag('unary')
            return ops.gemm(x, w['obj_unary'], w['obj_unary_b'])
        lane = node_lane(nf.device)
        if lane is None:
            obj_rep = node_mlp()
        else:
            side, ev_main, _ = lane
            ev_main.record(torch.cuda.current_stream(nf.device))
            side.wait_event(ev_main)                                 # nf is ready
            with torch.cuda.stream(side):
                obj_rep = node_mlp()                                 # consumed on the same stream by message_pass
        # :104  relu(edge_unary(roi_fmap(edge_feat)))
        _lib.set_tag('fc6_edge')
        if paired is not None:       # the long contraction once per unordered pair (f32), then per edge: + rect term + bias, ReLU
            yu = ops.gemm(ef, w['fc6_edge'], out_dtype=torch.float32)
            _lib.set_tag('fc6_edge_rect')
            y = ops.gemm_addrows(rect, w['fc6_edge_sum'], w['fc6_edge_b'], yu, paired.e2u, ops.ACT_RELU)
        else:
            y = ops.gemm(ef, w['fc6_edge'], w['fc6_edge_b'], ops.ACT_RELU, A2=rect, W2=w['fc6_edge_sum'])
        _lib.set_tag('fc7_edge')
        y = ops.gemm(y, w['fc7_edge'], w['fc7_edge_b'], ops.ACT_RELU if self.fc_layers()[1] else ops.ACT_NONE)
        _lib.set_tag('unary')
        rel_rep = ops.gemm(y, w['edge_unary'], w['edge_unary_b'], ops.ACT_RELU)
        # :105
        _lib.set_tag('imp')
        # _im_inds / _graphs: only forward() passes them (its rel_inds are sorted by (image, subject, object))
        csr = _csr if _csr is not None else ops.edge_csr(rel_inds, N, _im_inds, graphs=_graphs)
        vert, edge = message_pass(rel_rep, obj_rep, rel_inds, csr, w['imp'], self.mp_iter, dtype)
        # :107
        _lib.set_tag('heads')
        out = (ops.gemm(vert, w['obj_fc'], w['obj_fc_b'], out_dtype=torch.float32),
               ops.gemm(edge, w['rel_fc'], w['rel_fc_b'], out_dtype=torch.float32))
        _lib.set_tag('')
        return out

    def _watch_pair_flag(self, flag):
        """Training: the pair tables' device-side flag (an edge list outside the promise the host made to sgg_amd/pairing.py: a
        relation of a box with itself, an image id out of range, more than two rows on one unordered pair -- any of which would make
        the pair path add or drop rows silently) leaves for pinned host memory behind the step's kernels and is looked at by the NEXT
        call of forward() (or by check_pair_flag()), when it has long arrived: no synchronisation, and no step goes unchecked."""
        host = torch.empty(1, dtype=torch.int32, pin_memory=True)
        host.copy_(flag, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(flag.device))
        self.__dict__.setdefault('_pair_flags', []).append((host, ev))
        if __debug__ and os.environ.get('SGG_CHECK_COUNTS'):
            self.check_pair_flag(wait=True)

    def check_pair_flag(self, wait=True):
        """Raise if a training forward's relation list did not fit the unordered-pair tables (see _watch_pair_flag).
        wait=True: block until every pending flag has arrived (flush(), tests); wait='older': block only for the flags of the steps BEFORE the
        last one -- the host runs a step or more ahead of the GPU, and waiting for the newest flag would stall it until the GPU has caught
        up with the previous forward (measured: ~1 ms of host time every other step, and GPU idle time wherever the host then lagged)."""
        pend = self.__dict__.get('_pair_flags') or []
        keep = []
        for k, (host, ev) in enumerate(pend):
            must = wait is True or (wait == 'older' and k < len(pend) - 1)
            if not must and not ev.query():
                keep.append((host, ev))
                continue
            ev.synchronize()
            if int(host[0]) != 0:
                self.__dict__['_pair_flags'] = []
                raise RuntimeError('sgg_amd: a training batch\'s relation rows do not fit the unordered-pair tables (flag %d: 1 = self / '
                                   'out-of-range relation, 2 = more than two rows on one box pair); the step that used them is wrong -- '
                                   'filter the relations (filter_dups) or set SGG_EDGE_PAIRS=0' % int(host[0]))
        self.__dict__['_pair_flags'] = keep

    def forward(self, batch):
        """rel_model_stanford.py:110-207.  batch[0] = Blob tuple (dataloaders/blob.py:244-249); only items 0,3,4,5
        (imgs, gt_boxes, gt_classes, gt_rels) are read."""
        assert len(batch) == 1, ('single GPU is only supported in this code', len(batch))
        if not self.training and not self.__dict__.get('_eval_capture') and self.__dict__.get('_eval_graphs'):
            # opt-in (enable_eval_graphs(); SGG_EVAL_GRAPH=1 in bench.py): evaluation with the inputs already on the device, the forward replayed
            # as one hipGraph per batch signature (graph_forward.py).  Off by default: measured (round 5) the evaluation forward is GPU-bound --
            # 1843 images/s replayed against 1861 launch by launch on the same box -- so the graph only buys independence from a loaded host.
            # None = this batch goes launch by launch (warm-up of a new signature, a shape the capture does not take, SGG_GRAPH=0)
            from .graph_forward import eval_graphs
            out = eval_graphs(self).run(batch[0])
            if out is not None:
                return out
        if self.__dict__.get('_pair_flags'):
            self.check_pair_flag(wait='older')       # flags of the steps before the previous one have long arrived; the previous one's is only polled
        x, gt_boxes, gt_classes, gt_rels = batch[0][0], batch[0][3], batch[0][4], batch[0][5]
        dev = self.rel_fc.weight.device
        # index tensors that arrive on the host keep a host mirror (no D2H sync later for data the host already has)
        mv = lambda t: to_device_with_mirror(t, dev) if not t.is_cuda else t
        gt_boxes, gt_classes = gt_boxes.to(dev), mv(gt_classes)
        gt_rels = mv(gt_rels) if gt_rels is not None else None
        with torch.no_grad():
            result = self.faster_rcnn(x, gt_boxes, gt_classes, gt_rels)                  # :125-129
            result.fmap = result.fmap.detach()                                           # :131
            im_inds, boxes = result.im_inds, result.rm_box_priors
            if self.training and not hasattr(result, 'rel_labels'):
                # :136-140.  (The reference tests `result.rel_labels is None` on a Result that has already dropped its None
                # fields, lib/pytorch_misc.py:696-700; the intent -- sample relation labels for the detections -- is kept.)
                assert self.mode == 'sgdet'
                result.rel_labels = rel_assignments(im_inds.data, boxes.data, result.rm_obj_labels.data, gt_boxes.data,
                                                    gt_classes.data, gt_rels.data, 0, filter_non_overlap=True,
                                                    num_sample_per_gt=1)
            elif not hasattr(result, 'rel_labels'):
                result.rel_labels = None
            # Given boxes (evaluation; training when every ordered pair has exactly one row): the candidate list (all ordered same-image
            # pairs, :143-165), its CSR lists and the unordered-pair
            # tables depend on the boxes-per-image counts ONLY -- built once per count signature (five index launches, ~85 us of a
            # 4.4 ms forward) and reused; the cached tensors are read-only inputs of the kernels below.
            segs = getattr(result, '_segs', None)
            plain = (not self.training) or getattr(result.rel_labels, '_sgg_plain', False)     # training: the host saw one row per ordered pair
            ckey = (tuple((i, e - s) for i, s, e in segs), str(dev), os.environ.get('SGG_EDGE_PAIRS', '1')) if (
                segs is not None and plain and self.mode != 'sgdet' and os.environ.get('SGG_GRAPH_CACHE', '1') != '0') else None
            cached = self.__dict__.setdefault('_graph_cache', {}).get(ckey) if ckey is not None else None
            if cached is not None:
                rel_inds = cached['rel_inds']        # (training, one row per ordered pair: the label rows' first three columns are this list)
            else:
                rel_inds = self.get_rel_inds(result.rel_labels if self.training else None, im_inds, boxes,
                                             _num=getattr(result, '_num_pairs', None))   # :144
            result.rel_inds = rel_inds
            rois = getattr(result, 'rois', None)                                         # :146 (both branches of faster_rcnn leave it)
            if rois is None:
                rois = torch.cat((im_inds[:, None].float(), boxes), 1)
            # every unordered box pair pooled (and, in predict, sent through fc6's long contraction) once: sgg_amd/pairing.py
            pairing = None
            if cached is not None:
                pairing = cached['pairing']
            elif getattr(result, '_segs', None) is not None and os.environ.get('SGG_EDGE_PAIRS', '1') != '0':
                # (training: the sampled rows may repeat an ordered pair -- only a host mirror of gt_rels can rule that out)
                pairing = make_pairing(rel_inds, result._segs, getattr(result.rel_labels, '_sgg_max_per_pair', 3)
                                       if self.training and result.rel_labels is not None else 2)
            elif not self.training and self.mode == 'sgdet' and os.environ.get('SGG_EDGE_PAIRS', '1') != '0':
                # detections: the overlap-filtered list of get_rel_inds is symmetric and sorted -- half of its edges are the pairs
                pairing = make_pairing_symmetric(rel_inds, rois.shape[0])
            if self.__dict__.get('_eval_capture') and os.environ.get('SGG_EVAL_CAPTURE_STOP') == 'vgg':
                return result.fmap.reshape(-1).view(torch.uint8)[:64].clone(), None
            result.node_feat, result.edge_feat = self.node_edge_features(
                result.fmap, rois, rel_inds[:, 1:], im_sizes=result.im_sizes, _pairing=pairing)   # :148
            if self.__dict__.get('_eval_capture') and os.environ.get('SGG_EVAL_CAPTURE_STOP') == 'roi':
                return result.node_feat.reshape(-1).view(torch.uint8)[:64].clone(), None
        csr = None
        fresh_tables = cached is None
        if ckey is not None:
            if cached is None:
                if len(self._graph_cache) > 32:
                    self._graph_cache.clear()
                rel_inds._sgg_pairs = rel_inds[:, 1:].contiguous()          # (subject, object) columns as the kernels take them: ops.pairs_of
                cached = self._graph_cache[ckey] = dict(rel_inds=rel_inds, pairing=pairing,
                                                        csr=ops.edge_csr(rel_inds, rois.shape[0], im_inds.contiguous(),
                                                                         graphs=getattr(result, '_graphs', None)))
            csr = cached['csr']
        result.rm_obj_dists, result.rel_dists = self.predict(result.node_feat, result.edge_feat, rel_inds,
                                                             rois=rois, im_sizes=result.im_sizes,
                                                             _im_inds=im_inds.contiguous(),
                                                             _graphs=getattr(result, '_graphs', None), _csr=csr)   # :153
        if self.__dict__.get('_eval_capture') and os.environ.get('SGG_EVAL_CAPTURE_STOP') == 'predict':
            return result.rel_dists.reshape(-1).view(torch.uint8)[:64].clone(), None
        if self.use_bias:                                                                # :159-177, one fused lookup
            result.rel_dists, result.obj_preds = self.freq_bias.apply_to(
                result.rel_dists, result.rm_obj_dists, rel_inds,
                gt_classes=gt_classes[:, 1].contiguous() if self.mode == 'predcls' else None, replace=self.test_bias)
        if self.training:
            result.rois = rois
            if pairing is not None and (ckey is None or fresh_tables):
                self._watch_pair_flag(pairing.flag)        # (a cache hit re-reads tables whose flag was looked at when they were built)
            if ckey is not None:
                # the cached index tensors serve every later batch of this box-count signature: hand out a copy, so that reference-style
                # consumer code which edits result.rel_inds in place (index offsets, ...) cannot corrupt them
                result.rel_inds = rel_inds.clone()
            return result                                                                # :179-181
        if self.mode == 'predcls':
            gt = gt_classes[:, 1].contiguous()                                           # :184-185
        elif self.mode in ['sgcls', 'sgdet']:
            gt = None
        else:
            raise NotImplementedError(self.mode)
        obj_scores, obj_preds, rels, pred_scores = ops.eval_tail(result.rm_obj_dists, result.rel_dists, rel_inds, gt)
        result.obj_scores, result.obj_preds = obj_scores, obj_preds
        if self.__dict__.get('_eval_capture') and os.environ.get('SGG_EVAL_CAPTURE_STOP') == 'tail':
            return pred_scores.reshape(-1).view(torch.uint8)[:64].clone(), None
        bboxes = result.rm_box_priors_org                                                # :199
        if bboxes.dim() != 2:
            raise ValueError('Boxes needs to be [num_box, 4] but its {}'.format(bboxes.size()))
        # lib/surgery.py:49-55: host numpy copies -- ONE device->host transfer (five blocking copies otherwise): the outputs are laid
        # end to end in a byte buffer, widest element type first (so that every piece is aligned for its dtype), and viewed back
        outs = [bboxes, obj_preds, obj_scores, rels, pred_scores] + ([pairing.flag] if pairing is not None else [])
        order = sorted(range(len(outs)), key=lambda i: -outs[i].element_size())
        packed = torch.cat([outs[i].contiguous().view(-1).view(torch.uint8) for i in order])
        meta = (order, [(tuple(t.shape), t.dtype) for t in outs], pairing is not None)
        if self.__dict__.get('_eval_capture'):
            return packed, meta             # (sgg_amd/graph_forward.py: the launches above are being captured; the host copy happens at replay)
        return self.unpack_eval(packed.cpu().numpy(), meta)

    def enable_eval_graphs(self, on=True):
        """evaluation forwards with device-resident inputs replayed as hipGraphs (sgg_amd/graph_forward.py); off by default"""
        self.__dict__['_eval_graphs'] = True if on else False
        return self

    @staticmethod
    def unpack_eval(host, meta):
        """the five host arrays of filter_dets (lib/surgery.py:49-55) out of the packed byte buffer of forward()"""
        order, specs, has_flag = meta
        pieces, off = [None] * len(specs), 0
        for i in order:
            shape, dtype = specs[i]
            nb = int(torch.zeros(0, dtype=dtype).element_size())
            for d_ in shape:
                nb *= int(d_)
            pieces[i] = host[off:off + nb].view(torch.zeros(0, dtype=dtype).numpy().dtype).reshape(shape)
            off += nb
        if has_flag and int(pieces[5][0]) != 0:
            raise RuntimeError('sgg_amd: the relation list does not fit the unordered-pair tables (flag %d)' % int(pieces[5][0]))
        return tuple(pieces[:5])
