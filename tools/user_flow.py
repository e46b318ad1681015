"""A reference user's flow in a fresh process, first import = the package (main.py with one import changed): model, get_optim-style parameter
groups on a PLAIN torch SGD, MultiStepLR, a few train iterations with the reference's own loss code shape, checkpoint save / load, eval forward,
the x3 mode, Trainer.  Prints what it did; any exception is a finding.   python tools/user_flow.py"""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgg_amd import RelModelStanford           # noqa: E402  (FIRST import: no torch before it)
import numpy as np                              # noqa: E402
import torch                                    # noqa: E402
import torch.nn.functional as F                 # noqa: E402
from torch.optim.lr_scheduler import MultiStepLR  # noqa: E402

from sgg_amd.synthetic import SyntheticData, init_weights, synthetic_batch   # noqa: E402


def main():
    dev = 'cuda'
    S = 320
    model = init_weights(RelModelStanford(train_data=SyntheticData(), mode='sgcls', min_size=S, max_size=S))
    for n, p in model.detector.named_parameters():          # main.py:62-63
        p.requires_grad = False
    model.to(dev)
    model.set_compute_dtype(torch.bfloat16)                  # (no loss scale needed with a plain optimiser)
    fc = [p for n, p in model.named_parameters() if n.startswith('roi_fmap') and p.requires_grad]
    rest = [p for n, p in model.named_parameters() if not n.startswith('roi_fmap') and p.requires_grad]
    opt = torch.optim.SGD([{'params': fc, 'lr': 1e-4}, {'params': rest}], lr=1e-3, momentum=0.9, weight_decay=1e-4)     # pytorch_misc.py:130-157
    sched = MultiStepLR(opt, [3], gamma=0.1)
    batches = [synthetic_batch(B=2, S=S, n_boxes=6 + k, n_fg=3, seed=k) for k in range(3)]
    losses = []
    for it in range(6):
        b = batches[it % 3]
        model.train()
        res = model([b])                                      # host-resident Blob tuple, as the reference hands it over
        loss = F.cross_entropy(res.rm_obj_dists.float(), res.rm_obj_labels) + F.cross_entropy(res.rel_dists.float(), res.rel_labels[:, -1])
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_([p for p in model.parameters() if p.grad is not None], 5.0)
        opt.step()
        sched.step()
        losses.append(float(loss))
    print('plain SGD losses', ['%.3f' % v for v in losses])
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, 'vgrel.pth')
        torch.save({'state_dict': model.state_dict(), 'optimizer': opt.state_dict()}, path)
        ck = torch.load(path)
        m2 = RelModelStanford(train_data=SyntheticData(), mode='sgcls', min_size=S, max_size=S)
        missing, unexpected = m2.load_state_dict(ck['state_dict'], strict=False)
        assert not missing and not unexpected, (missing, unexpected)
        m2.to(dev).eval()
    model.eval()
    outs = {}
    for name, kw in (('bf16', dict(dtype=torch.bfloat16)), ('f16', dict(dtype=torch.float16)), ('x3', dict(dtype=torch.float32, split3=True)), ('f32', dict(dtype=torch.float32))):
        model.set_compute_dtype(**kw)
        m2.set_compute_dtype(**kw)
        with torch.no_grad():
            a = model([batches[0]])
            b2 = m2([batches[0]])
        assert np.array_equal(a[3], b2[3]) and np.allclose(a[4], b2[4], atol=1e-6), name       # the restored model is the model
        outs[name] = a
        print(name, 'eval ok:', a[4].shape, model.parity_clause['mode'])
    assert np.abs(outs['x3'][4] - outs['f32'][4]).max() < 1e-3
    from sgg_amd.trainer import Trainer
    model.set_compute_dtype(torch.float16)
    tr = Trainer(model, lr=1e-3)
    tl = [float(tr.step(batches[i % 3])) for i in range(6)]
    tr.flush()
    print('Trainer (f16) losses', ['%.3f' % v for v in tl])
    model.set_compute_dtype(torch.float32, split3=True, backward_f16=True)
    tr = Trainer(model, lr=1e-3)
    tl = [float(tr.step(batches[i % 3])) for i in range(4)]
    tr.flush()
    print('Trainer (x3, f16 backward) losses', ['%.3f' % v for v in tl])
    assert all(np.isfinite(tl))
    print('user flow OK')


if __name__ == '__main__':
    main()
