"""The evaluation forward of RelModelStanford as one replayed hipGraph per batch signature (round 5; the train step's counterpart is
sgg_amd/graph_step.py, whose findings about the runtime apply here: one-stream graphs only, no empty graphs, a device-wide synchronisation
every few dozen replays; a fourth one was met here -- a captured hipMemsetAsync (sgg_eval_tail's) made the replay fault, csrc/common.h
sgg_fill_u32).

`model([batch])` in eval mode issues ~100 launches (VGG-16, RoIAlign, the head, message passing, the eval tail) and then copies its five
result arrays to the host -- a synchronisation per call, so the 1.6 ms the issuing thread needs add to the step's latency instead of hiding
behind the GPU.  Here the batch is copied into static inputs (one multi-tensor copy), ONE graph is launched (the lanes of the launch-by-launch
path are off during its capture: scheduling only, the same sums), and the packed result buffer is copied to the host as before.  Same kernels,
same arguments: the outputs are bit-equal to the plain forward's (tests/test_graph_gpu.py).

A signature = image shapes, boxes per image, mode, compute dtype.  Its first two calls run launch by launch (they fill the per-signature
caches a capture must not touch).  Taken: gt-box modes (sgcls / predcls) of the VGG-16 model in a 16-bit compute dtype, inputs on the device,
gt_classes with its host mirror (rel_model_base.to_device_with_mirror / DeviceStager), images the transform does not resize.  Everything
else, and any error during a capture, stays on the plain path.  SGG_GRAPH=0 turns it off."""
import os
import warnings

import torch

from . import _lib, ops
from .rel_model_base import image_hw

WARM_CALLS = 2


class _G(object):
    __slots__ = ('graph', 'static', 'packed', 'meta', 'pins')


class EvalGraphs(object):
    def __init__(self, model):
        self.model = model
        self.G, self.warm = {}, {}
        self.generation = None
        self.disabled = None
        self.pool = None
        self.since_sync = 0
        self.sync_every = int(os.environ.get('SGG_GRAPH_SYNC_EVERY', '0'))      # (round 6: off, see graph_step.py)
        self.stats = dict(replayed=0, plain=0, captures=0)

    def _signature(self, batch):
        m = self.model
        if self.disabled or os.environ.get('SGG_GRAPH', '1') == '0' or m.mode not in ('sgcls', 'predcls') or m.backbone != 'vgg16':
            return None
        if not ops.is_half(m.compute_dtype) or getattr(m, 'use_bias', False):
            return None
        imgs, boxes, classes = batch[0], batch[3], batch[4]
        if not (torch.is_tensor(boxes) and boxes.is_cuda and torch.is_tensor(classes) and classes.is_cuda):
            return None
        if not all(torch.is_tensor(im) and im.is_cuda for im in imgs):
            return None
        ch = getattr(classes, '_sgg_host', None)
        if ch is None or tuple(ch.shape) != tuple(classes.shape):
            return None
        counts, order = {}, []
        for i in ch[:, 0].tolist():
            if i not in counts:
                order.append(i)
                counts[i] = 0
            counts[i] += 1
        if order != list(range(len(imgs))) or min(counts.values()) < 2:
            return None
        for im in imgs:
            h, w = image_hw(im)
            if m.detector.transform.resized_hw(h, w) != (h, w):
                return None
        return (tuple((tuple(im.shape), str(im.dtype)) for im in imgs), tuple(counts[i] for i in order), tuple(boxes.shape), str(boxes.dtype),
                m.mode, str(m.compute_dtype), os.environ.get('SGG_EDGE_PAIRS', '1'), bool(m.require_overlap))

    def run(self, batch):
        """-> the forward's 5-tuple from a replayed graph, or None (the caller runs the plain forward)"""
        m = self.model
        # (compute dtype, a trainer's updates, anything that wrote a parameter -- load_state_dict, an optimiser of the caller's)
        gen = (getattr(m, '_operand_generation', 0), getattr(m, 'weights_version', 0), id(m),
               tuple(p._version for p in m.parameters()))
        sig = self._signature(batch)
        if sig is None:
            self.stats['plain'] += 1
            return None
        if self.generation != gen:
            # another compute dtype, or a trainer moved the weights: the derived operands a graph reads may have been rebuilt elsewhere
            # (prepared() allocates the fp32-mode ones anew) -- the graphs are made again
            self.G.clear()
            self.warm.clear()
            self.generation = gen
        g = self.G.get(sig)
        if g is None:
            n = self.warm.get(sig, 0)
            if n < WARM_CALLS or len(self.G) >= int(os.environ.get('SGG_GRAPH_SIGNATURES', '8')):      # (bounded: see graph_step.py)
                if len(self.warm) > 4096:
                    self.warm.clear()
                self.warm[sig] = n + 1
                self.stats['plain'] += 1
                return None
        try:
            if g is None:
                g = self._capture(sig, batch)
            else:
                self._feed(g, batch)
            self.since_sync += 1
            if self.sync_every and self.since_sync >= self.sync_every:
                torch.cuda.synchronize(batch[3].device)
                self.since_sync = 0
            g.graph.replay()
            host = g.packed.cpu().numpy()
            if g.meta is None:              # (debugging: a capture cut short by SGG_EVAL_CAPTURE_STOP)
                return None
        except Exception as e:
            if self.disabled is None:
                self.disabled = '%s: %s' % (type(e).__name__, e)
                warnings.warn('sgg_amd: hipGraph capture of the evaluation forward failed (%s); continuing launch by launch' % self.disabled, RuntimeWarning)
            m.__dict__['_eval_capture'] = False
            self.G.clear()
            if os.environ.get('SGG_GRAPH_STRICT') == '1':
                raise
            self.stats['plain'] += 1
            return None
        self.stats['replayed'] += 1
        return m.unpack_eval(host, g.meta)

    def _feed(self, g, batch):
        srcs = list(batch[0]) + [batch[3], batch[4]]
        dsts = list(g.static[0]) + [g.static[3], g.static[4]]
        torch._foreach_copy_(dsts, srcs)
        g.static[4]._sgg_host = getattr(batch[4], '_sgg_host', None)

    def _capture(self, sig, batch):
        from .graph_step import GraphStep
        m = self.model
        dev = batch[3].device
        g = _G()
        st = list(batch)
        st[0] = [torch.empty_like(im) for im in batch[0]]
        st[3], st[4] = torch.empty_like(batch[3]), torch.empty_like(batch[4])
        st[5] = None                        # (the evaluation forward does not read the relations)
        g.static = st
        self._feed(g, batch)
        m.prepared()                        # nothing of a rebuild may land in the capture
        if m.__dict__.get('_pair_flags'):
            m.check_pair_flag(wait=True)
        graph = torch.cuda.CUDAGraph()
        if self.pool is None:
            self.pool = torch.cuda.graph_pool_handle()
        saved = {k: os.environ.get(k) for k in GraphStep.LANE_SWITCHES}
        for k in GraphStep.LANE_SWITCHES:
            os.environ[k] = '0'
        m.__dict__['_eval_capture'] = True
        try:
            with _lib.CAPTURE_LOCK:
                torch.cuda.synchronize(dev)
                with torch.no_grad(), torch.cuda.graph(graph, pool=self.pool, capture_error_mode='relaxed'):
                    g.packed, g.meta = m([tuple(st)])
        finally:
            m.__dict__['_eval_capture'] = False
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        g.graph = graph
        from .graph_step import pin_caches
        g.pins = pin_caches(m)              # the per-signature index tensors the graph reads live as long as the graph
        self.G[sig] = g
        self.stats['captures'] += 1
        return g


def eval_graphs(model):
    eg = model.__dict__.get('_eval_graphs')
    if eg is None or eg is True:
        eg = model.__dict__['_eval_graphs'] = EvalGraphs(model)
    return eg
