"""Where do the two queues of a replayed train step (sgg_amd/graph_step.py) sit idle?  A kernel trace answers that under the profiler only
(its step takes 7.3 ms against 6.8); this stamps the constant-rate clock at stream positions instead -- before and after every graph replay
(U on the lane, V and B's segments on the calling stream) -- in an otherwise unprofiled run, and prints the median offsets within a step.
    python tools/step_stamps.py [steps]        (GPU; needs tools/native/libstamp.so: bash tools/native/build.sh)"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import sgg_amd
from sgg_amd.rel_model_base import to_device_with_mirror
from sgg_amd.synthetic import SyntheticData, init_weights, synthetic_batch
from sgg_amd.trainer import Trainer

DEV = 'cuda:0'
FIXED = 8
INNER = ('V first kernel', 'V last kernel', 'U first kernel', 'U last kernel')


class Stamps:
    def __init__(self, n):
        self.lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'native', 'libstamp.so'))
        self.buf = torch.zeros(n, dtype=torch.int64, device=DEV)
        self.tags = []

    def fixed(self, slot):
        """a stamp at a fixed address (the last FIXED words of the buffer): what a captured graph can carry -- every replay overwrites it"""
        rc = self.lib.stamp_launch(ctypes.c_void_p(self.buf.data_ptr() + 8 * (self.buf.numel() - FIXED + slot)),
                                   ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0

    def mark(self, tag):
        k = len(self.tags)
        if k >= self.buf.numel() - FIXED:
            return
        rc = self.lib.stamp_launch(ctypes.c_void_p(self.buf.data_ptr() + 8 * k), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
        self.tags.append(tag)


class Stamped:
    """a captured graph whose replay is bracketed by two stamps on the stream it is replayed on"""

    def __init__(self, g, tag, st):
        self.g, self.tag, self.st = g, tag, st

    def replay(self):
        self.st.mark(self.tag + ' begin')
        self.g.replay()
        self.st.mark(self.tag + ' end')

    def __getattr__(self, k):
        return getattr(self.g, k)


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls')).to(DEV).eval()
    model.set_compute_dtype(torch.float16)
    b = list(synthetic_batch(B=8, S=592, n_boxes=32, n_fg=6, seed=111))
    b[0] = [(im * 255).round().to(torch.uint8).permute(1, 2, 0).contiguous().to(DEV) for im in b[0]]
    b[3], b[4], b[5] = b[3].to(DEV), to_device_with_mirror(b[4], DEV), to_device_with_mirror(b[5], DEV)
    b = tuple(b)
    tr = Trainer(model, lr=1e-3, pipeline=True)
    st = Stamps(steps * 16 + 64)
    # stamps INSIDE the captured graphs U and V (fixed addresses): when does a replayed graph's first kernel really start?
    import sgg_amd.train as train_mod
    feat, opt_step, tw = model.detector.features, tr.opt.step, train_mod.train_weights

    def features(*a, **kw):
        if getattr(model.detector, '_features_override', None) is not None:      # (graph B's capture: the map is handed over, nothing runs)
            return feat(*a, **kw)
        st.fixed(0)
        r = feat(*a, **kw)
        st.fixed(1)
        return r

    def step_(*a, **kw):
        st.fixed(2)
        return opt_step(*a, **kw)

    def train_weights(m, *a, **kw):
        rebuilds = 'train' not in m.prepared()
        r = tw(m, *a, **kw)
        if rebuilds:
            st.fixed(3)
        return r
    model.detector.features, tr.opt.step, train_mod.train_weights = features, step_, train_weights
    for _ in range(12):
        tr.step(b)
    torch.cuda.synchronize()
    g = tr.graphs
    assert g is not None and g.disabled is None and g.B and g.U and g.V, 'graphs are not up'
    for k, (u1, u2) in list(g.U.items()):
        g.U[k] = (Stamped(u1, 'U', st), u2)
    for k, v in list(g.V.items()):       # 'V' = the first part (SGG_GRAPH_VSPLIT: the update is released after it), 'V2' = the rest
        v.parts = [Stamped(x, 'V' if i == 0 else 'V%d' % (i + 1), st) for i, x in enumerate(v.parts)]
    for bb in g.B.values():
        bb.segments = [(tag, Stamped(x, 'B.' + tag, st) if x is not None else None) for tag, x in bb.segments]
    feed = g._feed

    def stamped_feed(*a, **kw):
        st.mark('step')
        return feed(*a, **kw)
    g._feed = stamped_feed
    inner = {k: [] for k in INNER}
    for trial in range(max(1, steps // 8)):
        for _ in range(8):
            tr.step(b)
        torch.cuda.synchronize()
        # the fixed stamps now hold the LAST step's U and V; its 'V begin' is the last one on the ring
        ring = st.buf[:len(st.tags)].tolist()
        last_v = max(i for i, tag in enumerate(st.tags) if tag == 'V begin')
        fx = st.buf[-FIXED:].tolist()
        for j, k in enumerate(INNER):
            inner[k].append((fx[j] - ring[last_v]) / 100.0)
    tr.flush()
    torch.cuda.synchronize()
    t = st.buf[:len(st.tags)].tolist()
    # steps: from one 'step' mark to the next; offsets in us (the clock runs at 100 MHz) relative to the step's 'V begin'
    idx = [i for i, tag in enumerate(st.tags) if tag == 'step']
    rows = {}
    for a, z in zip(idx[3:-1], idx[4:]):
        tags = st.tags[a:z]
        if 'V begin' not in tags:
            continue
        t0 = t[a + tags.index('V begin')]
        seen = {}
        for j, tag in enumerate(tags):
            seen[tag] = seen.get(tag, 0) + 1
            rows.setdefault('%s #%d' % (tag, seen[tag]) if tag.startswith('B.') else tag, []).append((t[a + j] - t0) / 100.0)
        rows.setdefault('next step mark', []).append((t[z] - t0) / 100.0)
    med = lambda v: sorted(v)[len(v) // 2]
    print('median stream-position stamps of a replayed step, us after "V begin" (%d steps):' % len(rows.get('step', [])))
    for tag, v in sorted(rows.items(), key=lambda kv: med(kv[1])):
        print('  %-20s %9.1f   (min %.1f, max %.1f)' % (tag, med(v), min(v), max(v)))
    print('stamps captured inside the graphs (fixed addresses, the last step of each of %d runs of 8 steps), us after "V begin":' % len(inner[INNER[0]]))
    for k in INNER:
        v = inner[k]
        print('  %-16s %9.1f   (min %.1f, max %.1f)' % (k, med(v), min(v), max(v)))
    per = [(t[z] - t[a]) / 100.0 for a, z in zip(idx[3:-1], idx[4:])]
    print('step to step (main stream, stamp to stamp): median %.1f us' % med(per))


if __name__ == '__main__':
    main()
