"""Race detector for the two-stream schedule: the same 600 bench-size train steps twice (rotating batches, dropout on, pipelined optimiser),
SHA-256 of all head weights and momenta after each run -- any unordered hand-over between the streams shows up as a different hash."""
import hashlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sgg_amd
from sgg_amd.synthetic import SyntheticData, init_weights, synthetic_batch
from sgg_amd.rel_model_base import to_device_with_mirror
from sgg_amd.trainer import Trainer
dev = 'cuda:0'
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
bs = []
for seed in (1, 2, 3):
    b = list(synthetic_batch(B=8, S=592, n_boxes=32, n_fg=6, seed=seed))
    b[0] = [im.to(dev) for im in b[0]]
    b[3], b[4], b[5] = b[3].to(dev), to_device_with_mirror(b[4], dev), to_device_with_mirror(b[5], dev)
    bs.append(tuple(b))


def run():
    torch.manual_seed(7)
    model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls')).to(dev)
    tr = Trainer(model, lr=1e-3, pipeline=True)
    t0 = time.time()
    for i in range(steps):
        loss = tr.step(bs[i % 3])
    tr.flush()
    torch.cuda.synchronize()
    h = hashlib.sha256()
    for n, t in sorted(model.state_dict().items()):
        if not n.startswith('detector.'):
            h.update(t.detach().float().cpu().numpy().tobytes())
    for p in tr.opt.params():
        if 'momentum_buffer' in tr.opt.state[p]:
            h.update(tr.opt.state[p]['momentum_buffer'].cpu().numpy().tobytes())
    return h.hexdigest(), float(loss), (time.time() - t0) / steps * 1e3


a = run()
b = run()
print('run 1: %s loss %.5f %.2f ms/step' % a)
print('run 2: %s loss %.5f %.2f ms/step' % b)
print('bit-equal' if a[0] == b[0] else 'DIFFERENT')
sys.exit(0 if a[0] == b[0] else 1)
