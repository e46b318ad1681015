"""f-2: the batch container and the host -> HBM hop (sgg_amd/blob.py, sgg_image_prep_u8)."""
import numpy as np
import pytest
import torch

from oracle import sgg_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def square_pad_to_tensor(u8):
    """dataloaders/image_transforms.py:8-13 (SquarePad: expand right/bottom, fill int(mean*256)) followed by ToTensor,
    restated with numpy for the check."""
    h, w, _ = u8.shape
    s = max(h, w)
    out = np.empty((s, s, 3), np.uint8)
    out[...] = np.array([int(0.485 * 256), int(0.456 * 256), int(0.406 * 256)], np.uint8)
    out[:h, :w] = u8
    return torch.from_numpy(out).permute(2, 0, 1).float().div(255)


@pytest.fixture(scope='module')
def env():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import sgg_amd
    from sgg_amd import ops
    return sgg_amd, ops


@pytest.mark.parametrize('hw,S', [((40, 40), 40), ((30, 52), 64), ((57, 33), 32)])
def test_image_prep_u8_equals_squarepad_totensor_transform(env, hw, S):
    _, ops = env
    rng = np.random.RandomState(hw[0])
    u8 = rng.randint(0, 256, size=hw + (3,)).astype(np.uint8)
    ref, sizes, _ = O.transform([square_pad_to_tensor(u8)], None, min_size=S, max_size=S)
    Hp = ref.shape[-1]
    buf = torch.zeros(1, Hp + 2, Hp + 2, 4, device=DEV)
    ops.image_prep_u8(torch.from_numpy(u8).to(DEV), sizes[0][0], sizes[0][1], buf, 0)
    torch.testing.assert_close(buf[:, 1:-1, 1:-1, :3].permute(0, 3, 1, 2).cpu(), ref, atol=2e-5, rtol=1e-5)
    assert float(buf[..., 3].abs().max()) == 0 and float(buf[:, 0].abs().max()) == 0


def test_forward_from_staged_u8_blob_equals_forward_from_f32_tensors(env):
    """The whole hop: datum dicts -> vg_collate -> DeviceStager (pinned, one async copy) -> forward on u8 images, against the
    reference-style call with SquarePad + ToTensor'ed f32 host tensors."""
    sgg_amd, _ = env
    from sgg_amd.blob import DeviceStager, vg_collate
    from sgg_amd.synthetic import SyntheticData, init_weights
    S = 96
    model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls', min_size=S, max_size=S)).to(DEV).eval()
    model.set_compute_dtype(torch.float32)
    rng = np.random.RandomState(4)

    def datum(i, h, w, nb):
        xy = rng.uniform(0, 50, size=(nb, 2))
        boxes = np.concatenate((xy, xy + rng.uniform(10, 40, size=(nb, 2))), 1)
        return {'img': rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8), 'img_size': (S, S, 1.0), 'gt_boxes': boxes,
                'gt_classes': rng.randint(1, 151, nb), 'scale': 1.0, 'fn': '%d.jpg' % i,
                'gt_relations': np.array([[0, 1, 3], [1, 2, 9]])}
    loader = [vg_collate([datum(2 * k, 96, 80, 4), datum(2 * k + 1, 64, 96, 5)], is_train=False, mode='rel') for k in range(3)]
    stager = DeviceStager()
    outs = []
    with torch.no_grad():
        for dev_batch in stager.prefetch(loader):
            assert all(im.is_cuda and im.dtype == torch.uint8 for im in dev_batch[0]) and dev_batch[3].is_cuda
            outs.append(model([dev_batch]))
        for blob, got in zip(loader, outs):
            t = list(blob[0])
            t[0] = [square_pad_to_tensor(im) for im in t[0]]          # what the reference's dataset hands over
            exp = model([tuple(t)])
            for a, b in zip(got, exp):
                np.testing.assert_allclose(a, b, rtol=1e-4, atol=1e-5)
    # a single staged batch (no prefetch) gives the same thing
    with torch.no_grad():
        again = model([stager.stage(loader[1][0])])
    for a, b in zip(again, outs[1]):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize('threaded', [True, False])
def test_prefetch_delivers_every_batch_intact_in_order(env, threaded):
    """DeviceStager.prefetch over more batches than slots, the packing on a worker thread (default) or in line: every batch arrives
    once, in order, with its own bytes -- while the consumer keeps the device busy between batches (so that copies really overlap)."""
    from sgg_amd.blob import DeviceStager
    rng = np.random.RandomState(7)
    batches = []
    for k in range(11):
        nb = 3 + k % 4
        imgs = [rng.randint(0, 256, size=(64 + 8 * (k % 3), 96, 3)).astype(np.uint8) for _ in range(2)]
        batches.append((imgs, np.array([[96, 96, 1.0]] * 2), 0, torch.from_numpy(rng.rand(nb, 4).astype(np.float32)),
                        torch.from_numpy(rng.randint(0, 9, size=(nb, 2))), torch.from_numpy(rng.randint(0, 9, size=(k + 1, 4))), None, ['a', 'b']))
    stager = DeviceStager(slots=3)
    busy = torch.randn(2048, 2048, device=DEV)
    got = []
    for b in stager.prefetch(iter(batches), threaded=threaded):
        for _ in range(4):
            busy = (busy @ busy).clamp_(-1, 1)                 # compute queued on the consumer's stream while the next copy flies
        got.append(([im.clone() for im in b[0]], b[3].clone(), b[4].clone(), b[5].clone(), b[5]._sgg_host))
    torch.cuda.synchronize()
    assert len(got) == len(batches)
    for (imgs, boxes, cls, rels, host_rels), ref in zip(got, batches):
        for a, r in zip(imgs, ref[0]):
            np.testing.assert_array_equal(a.cpu().numpy(), r)
        assert torch.equal(boxes.cpu(), ref[3]) and torch.equal(cls.cpu(), ref[4]) and torch.equal(rels.cpu(), ref[5])
        assert torch.equal(host_rels, ref[5])
    # a loader that raises surfaces in the consumer
    def bad():
        yield batches[0]
        raise KeyError('decode failed')
    with pytest.raises(KeyError):
        for _ in stager.prefetch(bad(), threaded=threaded):
            pass


@pytest.mark.parametrize('threaded', [True, False])
def test_prefetch_refuses_one_slot_and_survives_an_abandoned_consumer(env, threaded):
    """slots=1 used to deadlock (threaded) or overwrite the batch in use (in line); a consumer that walks away must not leave the
    worker blocked on its queue or racing a following prefetch() on the same slots."""
    import threading
    from sgg_amd.blob import DeviceStager
    rng = np.random.RandomState(3)
    mk = lambda: ([rng.randint(0, 256, size=(32, 48, 3)).astype(np.uint8)], np.array([[48, 48, 1.0]]), 0, torch.rand(2, 4),
                  torch.zeros((2, 2), dtype=torch.int64), torch.zeros((1, 4), dtype=torch.int64), None, ['a'])
    with pytest.raises(ValueError):
        next(DeviceStager(slots=1).prefetch(iter([mk(), mk()]), threaded=threaded))
    stager = DeviceStager(slots=2)
    before = threading.active_count()
    gen = stager.prefetch(iter([mk() for _ in range(9)]), threaded=threaded)
    next(gen)
    gen.close()                                   # the consumer goes away after one batch
    assert threading.active_count() <= before     # the worker has been joined
    assert sum(1 for _ in stager.prefetch(iter([mk() for _ in range(5)]), threaded=threaded)) == 5
    torch.cuda.synchronize()
