"""RoIAlign at the bench configuration (8 x [37, 37, 512] maps, 256 object boxes + 3968 unordered pairs): time per call and the rate of
the rows it writes.  A/B between two builds:  python tools/roi_bench.py [lib.so ...]  (each library in its own process)."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import torch

    from sgg_amd import ops
    from sgg_amd.synthetic import synthetic_batch
    dev = 'cuda:0'
    b = synthetic_batch(B=8, S=592, n_boxes=32, n_fg=6, seed=111)
    boxes, im = b[3].float(), b[4][:, 0].float()
    rois = torch.cat((im[:, None], boxes), 1).to(dev)
    pr = [(i, j) for i in range(len(im)) for j in range(i + 1, len(im)) if im[i] == im[j]]
    pairs = torch.tensor(pr, dtype=torch.int64, device=dev)
    g = torch.Generator().manual_seed(3)
    res = {}
    for dtype in (torch.float16, torch.bfloat16, torch.float32):
        fm = torch.randn(8, 37, 37, 512, generator=g).to(dev).to(dtype)
        outs = {}
        for name, p in (('union', pairs), ('object', None)):
            out = ops.roi_align(fm, rois, p)
            for _ in range(5):
                ops.roi_align(fm, rois, p, out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                ops.roi_align(fm, rois, p, out=out)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 50
            outs[name] = out
            print('%-8s %-7s rows %5d  %.4f ms  %.0f GB/s written' % (str(dtype)[6:], name, out.shape[0], ms,
                                                                      out.numel() * out.element_size() / ms / 1e6))
        res[str(dtype)] = float(outs['union'].float().abs().sum()), float(outs['union'].float().sum())
    print('checksums', res)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] != '--child':
        for lib in sys.argv[1:]:
            print('==', lib, flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), '--child'], env=dict(os.environ, SGG_HIP_LIB=os.path.abspath(lib)))
    else:
        run()
