"""BASELINE configs[4] data-parallel: the GAN half of a training iteration (main.py:124-194, sgg_amd.feature_gan.gan_train_step) on TWO
ranks (gloo, one GPU) -- each with its own image -- against ONE process on the concatenated batch: the same generator, discriminator
and SGG-model weights afterwards.  Both forms of the SGG update: a plain optimiser (gradients all-reduced inside gan_train_step) and the
package's data-parallel Trainer (the backward's hooks, reduce-scatter, sharded fused clip + SGD)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
S = 592        # the real size: 38 x 38 x 512 feature maps (the global discriminator's layer table is built for 38 or 21)
WATCH = ['G_obj_embed.weight', 'G_proj.weight', 'G_refine.output_conv.0.weight', 'G_gcn.gconvs.0.net1.0.weight', 'D_nodes.0.weight_orig',
         'D_edges.6.weight_orig', 'D_global.15.weight_orig', 'sgg.rel_fc.weight', 'sgg.roi_fmap.1.0.weight', 'sgg.edge_gru.weight_ih',
         'sgg.union_boxes.conv.0.weight']


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _iteration(batch, use_trainer, dist_on):
    import sgg_amd
    from sgg_amd import dense
    from sgg_amd.feature_gan import GAN, gan_train_step
    from sgg_amd.synthetic import SyntheticData, init_weights
    from sgg_amd.trainer import Trainer
    dev = 'cuda:0'
    torch.manual_seed(3)
    data = SyntheticData()
    model = init_weights(sgg_amd.RelModelStanford(data, mode='sgcls', min_size=S, max_size=S)).to(dev)
    model.set_compute_dtype(torch.float32)
    model.dropout_p = 0.0
    for n, p in model.named_parameters():
        if n.startswith('detector.'):
            p.requires_grad = False
    gan = GAN(data.ind_to_classes, data.ind_to_predicates, n_ch=model.edge_dim, pool_sz=model.pool_sz, fmap_sz=model.fmap_sz, n_layers_G=2,
              BN=False, device=dev).to(dev)                       # (BatchNorm1d of the graph convolutions would be replica-local)
    if dist_on:      # the refinement network's BatchNorm2d layers (augment/crn.py: normalization='batch'): statistics over every rank's images
        gan = dense.sync_batchnorm_(gan)            # (the channels-last counterpart of SyncBatchNorm.convert_sync_batchnorm)
    dev_batch = tuple(t_.to(dev) if isinstance(t_, torch.Tensor) else t_ for t_ in batch)
    model.train()
    tr = Trainer(model, lr=1e-2, comm_dtype=None, sync_bn=True) if (use_trainer or dist_on) else None
    res = model([dev_batch])
    sgg_params = [p for p in model.parameters() if p.requires_grad]
    optimizer = None if use_trainer else torch.optim.SGD(sgg_params, lr=1e-2, momentum=0.9, weight_decay=1e-4)
    # (plain SGD for G and D here: Adam's first step is lr * sign(g), which turns rounding noise on near-zero gradients into full steps)
    G_opt = torch.optim.SGD([p for n, p in gan.named_parameters() if n.startswith('G_')], lr=1e-2)
    D_opt = torch.optim.SGD([p for n, p in gan.named_parameters() if n.startswith('D_')], lr=1e-2)
    named = dict(list(gan.named_parameters()) + [('sgg.' + n, p) for n, p in model.named_parameters() if p.requires_grad])
    before = {n: named[n].detach().float().cpu().numpy().copy() for n in WATCH}
    losses = gan_train_step(model, gan, res, dev_batch[3].clone(), dev_batch[4].clone(), dev_batch[5].clone(), optimizer, G_opt, D_opt,
                            ganw=1.0, trainer=tr if use_trainer else None)
    if tr is not None:
        tr.flush()
    torch.cuda.synchronize()
    after = {n: named[n].detach().float().cpu().numpy() for n in WATCH}
    return {k: float(v) for k, v in losses.items()}, before, after


def _global_batch():
    from sgg_amd.synthetic import synthetic_batch
    return synthetic_batch(B=2, S=S, n_boxes=5, n_fg=4, seed=8, ragged=True)       # 5 and 2 boxes: ragged shards


def _worker(rank, world, port, use_trainer, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK='0', WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch.distributed as dist
    from sgg_amd.synthetic import shard_batch
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    losses, _, after = _iteration(shard_batch(_global_batch(), rank, rank + 1), use_trainer, True)
    q.put((rank, losses, {k: v.tolist() for k, v in after.items()}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('use_trainer', [False, True], ids=['plain_optimizer', 'trainer_update'])
def test_gan_iteration_on_two_ranks_equals_one_process_on_the_concatenated_batch(use_trainer):
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, use_trainer, q)) for r in range(2)]
    for p in procs:
        p.start()
    import queue
    import time
    got, t0 = {}, time.time()
    while len(got) < 2:
        try:
            r, losses, after = q.get(timeout=5)
            got[r] = (losses, {k: np.asarray(v, dtype=np.float32) for k, v in after.items()})
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or time.time() - t0 > 600:
                for p in procs:
                    p.kill()
                pytest.fail('a rank exited with %s (or the iteration hung)' % dead)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ref_losses, before, ref_after = _iteration(_global_batch(), use_trainer, False)
    assert sorted(ref_losses) == ['D_fmap', 'D_obj', 'D_rel', 'G_fmap', 'G_obj', 'G_rel', 'rec']
    for k in ref_losses:                                         # weighted local means add up to the global mean
        tot = got[0][0][k] + got[1][0][k]
        assert abs(tot - ref_losses[k]) <= 2e-3 * max(1.0, abs(ref_losses[k])), (k, tot, ref_losses[k])
    for n in WATCH:
        step = np.abs(ref_after[n] - before[n]).max()
        assert step > 0, n
        for r in (0, 1):
            diff = np.abs(got[r][1][n] - ref_after[n]).max()
            assert diff <= 0.05 * step + 1e-7, (n, r, float(step), float(diff))
        np.testing.assert_allclose(got[0][1][n], got[1][1][n], rtol=0, atol=1e-6 * max(1.0, float(np.abs(ref_after[n]).max())))
