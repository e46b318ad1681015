"""The batch container in front of the path: dataloaders/blob.py + vg_collate, one process per GPU.

The reference's Blob chunks one batch over `num_gpus` devices of ONE process (blob.py:128-141,214-261) and moves
images with a synchronous per-image `x[i].to(device)` from pageable memory (sgg_models/rel_model_base.py:180).
Here a process owns one GPU, so a Blob holds exactly this rank's images, and the host->device hop is explicit:

  * `Blob.append(d)` / `reduce()` / `__getitem__(0)` -- the reference's datum dict and 8/9-tuple layout
    (blob.py:77-126,145-166,214-261), images kept as they were decoded: u8 [h,w,3] (4x fewer PCIe bytes; SquarePad +
    ToTensor then run inside `sgg_image_prep_u8`) or the reference's f32 [3,S,S] tensors;
  * `DeviceStager` -- ONE pinned staging buffer per batch slot, ONE async copy per batch on a dedicated HIP stream,
    an event the compute stream waits on; `prefetch(loader)` keeps the copy of batch i+1 in flight under the
    compute of batch i (two slots).
"""
import os

import numpy as np
import torch


class Blob(object):
    def __init__(self, mode='det', is_train=False, num_gpus=1, primary_gpu=0, batch_size_per_gpu=3, torch_detector=True,
                 is_cuda=True):
        assert mode in ('det', 'rel')
        if num_gpus != 1:
            raise ValueError('sgg_amd runs one process per GPU: build one Blob per rank (num_gpus=1), see sgg_amd/dist.py')
        self.mode, self.is_train, self.num_gpus = mode, is_train, 1
        self.batch_size_per_gpu, self.primary_gpu = batch_size_per_gpu, primary_gpu
        self.torch_detector, self.is_cuda = torch_detector, is_cuda
        self.fns, self.imgs, self.im_sizes = [], [], []
        self.gt_boxes, self.gt_classes, self.gt_rels, self.proposals = [], [], [], []
        self.gt_box_chunks = self.gt_rel_chunks = self.proposal_chunks = None

    @property
    def is_rel(self):
        return self.mode == 'rel'

    def append(self, d):
        """blob.py:77-126.  d: {'img', 'img_size' (h, w, scale), 'gt_boxes', 'gt_classes', 'scale', 'fn'[, 'gt_relations',
        'proposals']}"""
        self.fns.append(os.path.basename(d['fn']))
        i = len(self.imgs)
        self.imgs.append(d['img'])
        h, w, scale = d['img_size']
        self.im_sizes.append((h, w, scale))
        self.gt_boxes.append(np.asarray(d['gt_boxes']).astype(np.float32) * d['scale'])
        cls = np.asarray(d['gt_classes'])
        self.gt_classes.append(np.column_stack((i * np.ones(cls.shape[0], dtype=np.int64), cls)))
        if self.is_rel:
            rel = np.asarray(d['gt_relations'])
            self.gt_rels.append(np.column_stack((i * np.ones(rel.shape[0], dtype=np.int64), rel)))
        if 'proposals' in d:
            pr = np.asarray(d['proposals'])
            self.proposals.append(np.column_stack((i * np.ones(pr.shape[0], dtype=np.float32),
                                                   d['scale'] * pr.astype(np.float32))))

    @staticmethod
    def _cat(parts, dtype):
        t = np.concatenate(parts, 0) if parts else np.zeros((0,))
        if len(t) == 0:
            return 0                                   # blob.py:139-140 hands back the integer 0 for an empty list
        return torch.as_tensor(t, dtype=dtype)

    def reduce(self):
        """blob.py:145-166"""
        if len(self.imgs) != self.batch_size_per_gpu:
            raise ValueError('Wrong batch size? imgs len {} bsize/gpu {} numgpus {}'.format(
                len(self.imgs), self.batch_size_per_gpu, self.num_gpus))
        self.im_sizes = np.stack(self.im_sizes).reshape((1, self.batch_size_per_gpu, 3))
        if self.is_rel:
            self.gt_rel_chunks = [sum(r.shape[0] for r in self.gt_rels)]
            self.gt_rels = self._cat(self.gt_rels, torch.int64)
        self.gt_box_chunks = [sum(b.shape[0] for b in self.gt_boxes)]
        self.gt_boxes = self._cat(self.gt_boxes, torch.float32)
        self.gt_classes = self._cat(self.gt_classes, torch.int64)
        if len(self.proposals) != 0:
            self.proposal_chunks = [sum(p.shape[0] for p in self.proposals)]
            self.proposals = self._cat(self.proposals, torch.float32)

    def scatter(self):
        """blob.py:176-205 for one device: images stay where they are (the model / DeviceStager moves them)."""
        return self

    def __len__(self):
        return len(self.im_sizes)

    def __getitem__(self, index):
        """blob.py:214-261, the num_gpus == 1 branch."""
        if index != 0:
            raise ValueError('Out of bounds with index {} and {} gpus'.format(index, self.num_gpus))
        rels = self.gt_rels if self.is_rel else None
        proposals = self.proposals if self.proposal_chunks is not None else None
        if self.is_train:
            return (self.imgs, self.im_sizes[0], 0, self.gt_boxes, self.gt_classes, rels, proposals, None, self.fns)
        return self.imgs, self.im_sizes[0], 0, self.gt_boxes, self.gt_classes, rels, proposals, self.fns


def vg_collate(data, num_gpus=1, is_train=False, mode='det', torch_detector=True, is_cuda=True):
    """dataloaders/visual_genome.py:681-688"""
    assert mode in ('det', 'rel')
    blob = Blob(mode=mode, is_train=is_train, num_gpus=num_gpus, batch_size_per_gpu=len(data) // num_gpus,
                torch_detector=torch_detector, is_cuda=is_cuda)
    for d in data:
        blob.append(d)
    blob.reduce()
    return blob


class _Slot(object):
    def __init__(self):
        self.pinned = None
        self.pinned_np = None
        self.device = None
        self.event = torch.cuda.Event()
        self.consumed = None      # prefetch(): recorded on the compute stream once the step that read this slot's batch has been issued


class DeviceStager(object):
    """Host -> HBM for whole batches.  Every tensor of the batch tuple (images, gt_boxes, gt_classes, gt_rels) is packed
    into one pinned buffer (grown on demand, reused) and crosses PCIe as ONE async copy on `self.stream`; the returned
    tuple holds device views into the slot's device buffer.  Slots alternate, so batch i+1 can be staged while the
    compute stream still reads batch i."""

    ALIGN = 256

    def __init__(self, device=None, slots=3):
        if not torch.cuda.is_available():
            raise RuntimeError('DeviceStager needs the GPU')
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.slots = [_Slot() for _ in range(slots)]
        self._next = 0

    def _alloc(self, slot, cap):
        from ._lib import CAPTURE_LOCK
        with CAPTURE_LOCK:
            slot.pinned = torch.empty(cap, dtype=torch.uint8).pin_memory()
            slot.pinned_np = slot.pinned.numpy()
            slot.device = torch.empty(cap, dtype=torch.uint8, device=self.device)

    def reserve(self, nbytes):
        """Allocates every slot's pinned and device buffer for batches of up to `nbytes` packed bytes now (else: at the first batch)."""
        for slot in self.slots:
            if slot.pinned is None or slot.pinned.numel() < nbytes:
                slot.event.synchronize()
                self._alloc(slot, int(nbytes))
        return self

    @staticmethod
    def _as_tensor(x):
        if isinstance(x, np.ndarray):
            return torch.from_numpy(np.ascontiguousarray(x))
        return x.contiguous()

    def _pack(self, batch, slot):
        """Host half of staging: packs the batch's tensors into the slot's pinned buffer (plain memcpys; may run on a worker thread: numpy
        releases the GIL for them).  -> (items, offsets, total bytes, images, {tuple index: host tensor})"""
        imgs = [self._as_tensor(im) for im in batch[0]]
        rest = {i: self._as_tensor(batch[i]) for i in (3, 4, 5) if torch.is_tensor(batch[i]) or isinstance(batch[i], np.ndarray)}
        items = imgs + [rest[i] for i in sorted(rest)]
        offs, total = [], 0
        for t in items:
            offs.append(total)
            total += (t.numel() * t.element_size() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        total = max(total, self.ALIGN)
        slot.event.synchronize()       # host: the previous copy OUT of this pinned buffer has finished
        if slot.pinned is None or slot.pinned.numel() < total:
            self._alloc(slot, int(total * 1.25))
            for other in self.slots:   # the whole ring now, not one pinned allocation (milliseconds each) per step for the next len(slots) steps
                if other.pinned is None:
                    self._alloc(other, int(total * 1.25))
        # pack with plain single-threaded memcpys (numpy views): torch's CPU copy_ fans a 1 MB image out over every core,
        # and the OpenMP workers then spin-wait next to the HIP runtime's threads -- measured: 6.5 ms per staged forward
        # with this pack, 30-40 ms (random 60-100 ms stalls) with tensor.copy_ on a 128-thread host
        dst = slot.pinned_np
        flat = [t.numpy().reshape(-1).view(np.uint8) for t in items]
        k = 0
        # images of one size whose bytes are a multiple of ALIGN lie back to back: ONE concatenate into the pinned buffer (one stretch
        # without the interpreter lock instead of one per image -- on the worker thread every re-acquisition stalls the issuing thread)
        n0 = flat[0].size if flat else 0
        if len(imgs) > 1 and n0 and n0 % self.ALIGN == 0 and all(f.size == n0 for f in flat[:len(imgs)]):
            k = len(imgs)
            np.concatenate(flat[:k], out=dst[:k * n0])
        for f, o in zip(flat[k:], offs[k:]):
            if f.size:
                np.copyto(dst[o:o + f.size], f)
        return items, offs, total, len(imgs), rest

    def _launch(self, batch, slot, packed, ring=False, after=None):
        """Device half: ONE async copy of the packed bytes on the copy stream (issued from the consumer's thread, so that it is ordered
        behind the compute already queued there) + the device views.  -> (device tuple, slot)"""
        items, offs, total, n_img, rest = packed
        # device: whoever read this slot's previous contents was enqueued on the compute stream before this call.  prefetch() knows the
        # exact point (slot.consumed: the end of the step that read them, several steps back with a deep ring) -- the copy waits for THAT,
        # not for everything queued so far: waiting for the whole compute stream ties the copy to the previous step's last kernel, and the
        # HIP runtime then resolves the cross-queue dependency with the calling thread blocked (measured: 2.2-2.6 ms per step inside this
        # call, the issuing thread never more than one step ahead of the GPU)
        from ._lib import CAPTURE_LOCK
        with CAPTURE_LOCK:                          # (no GPU call of this thread while another one captures a hipGraph: sgg_amd/graph_step.py)
            if after is not None:                       # (prefetch()'s worker thread: an event of the consumer's stream, see there)
                self.stream.wait_event(after)
            elif ring and slot.consumed is not None:
                self.stream.wait_event(slot.consumed)
            else:
                self.stream.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.stream):
                slot.device[:total].copy_(slot.pinned[:total], non_blocking=True)
                slot.event.record(self.stream)
        views = []
        for t, o in zip(items, offs):
            n = t.numel() * t.element_size()
            views.append(slot.device[o:o + n].view(t.dtype).view(t.shape))
        out = list(batch)
        out[0] = views[:n_img]
        for k, i in enumerate(sorted(rest)):
            out[i] = views[n_img + k]
            if i in (4, 5):
                out[i]._sgg_host = rest[i]     # host mirror of gt_classes / gt_rels (rel_model_base.host_of): no D2H sync later
        return tuple(out), slot

    def _stage_async(self, batch, ring=False):
        """Packs `batch` into the next slot and launches its copy on the copy stream.  -> (device tuple, slot)"""
        slot = self.slots[self._next]
        self._next = (self._next + 1) % len(self.slots)
        if not ring:
            slot.consumed = None
        return self._launch(batch, slot, self._pack(batch, slot), ring)

    def stage(self, batch):
        """batch: the tuple of Blob.__getitem__(0).  -> same tuple with items 0,3,4,5 (imgs, gt_boxes, gt_classes, gt_rels)
        on the device.  The copy is asynchronous; the current stream waits for it (no host sync)."""
        out, slot = self._stage_async(batch)
        torch.cuda.current_stream(self.device).wait_event(slot.event)
        return out

    def prefetch(self, loader, threaded=True):
        """Generator over device-resident batches: the copy of the next batch is issued before the current one is handed to
        the caller, so it runs under the caller's compute.  `loader` yields Blobs or batch tuples.
        threaded (default): the loader itself, the packing into pinned memory (~1 ms of memcpys per 8-image batch) and the launch of the
        copy run on a worker thread, up to len(slots) - 1 batches ahead -- the consumer's thread, which is the one that launches the step's
        kernels, only makes its stream wait for the copy's event.  A batch's pinned buffer is reused once the copy out of it has finished
        (slot event), its device buffer once the step that read it has been ISSUED by the consumer (slot.consumed, an event on the consumer's
        stream that the next copy into the slot waits for).
        Contract (as with any loader that hands out recycled device buffers): everything that reads batch i must be ISSUED, on the stream
        that is current when the generator is resumed, before batch i + 1 is requested -- work issued on batch i after that request is not
        ordered against the copy that refills its slot len(slots) batches later.  Clone what has to live longer."""
        if len(self.slots) < 2:
            # batch i + 1 is copied while batch i is being read: with one slot that copy would overwrite the batch in use (in line), or the
            # worker would wait for a slot that is only released after the first yield (threaded: a deadlock)
            raise ValueError('DeviceStager.prefetch needs at least 2 slots (got %d); stage() works with one' % len(self.slots))
        for sl in self.slots:          # (events of an earlier generator say nothing about who read the slots last)
            sl.consumed = None
        th = None
        if not threaded:
            it = iter(loader)

            def grab():
                try:
                    b = next(it)
                except StopIteration:
                    return None
                return self._stage_async(b[0] if isinstance(b, Blob) else b, ring=True)
        else:
            import queue
            import threading
            S = len(self.slots)
            q = queue.Queue(maxsize=max(1, S - 1))
            # slot k: its DEVICE buffer may be overwritten (the consumer has recorded `consumed` for the batch that was in it, or the slot
            # has not been used by this generator).  The worker does everything up to and including the launch of the copy -- the consumer's
            # thread, which issues the step's kernels, only takes a finished item off the queue (round 4: 0.3 ms per batch on that thread
            # before, which a loop that synchronises every step -- inference, `filter_dets` on the host -- pays in full)
            free = [threading.Event() for _ in range(S)]
            for f in free:
                f.set()
            stop = threading.Event()
            start_ev = torch.cuda.Event()
            start_ev.record(torch.cuda.current_stream(self.device))     # whatever read the slots before this generator was queued before this point

            def work():
                try:
                    torch.cuda.set_device(self.device)
                    k = self._next
                    for b in loader:
                        b = b[0] if isinstance(b, Blob) else b
                        while not free[k].wait(0.05):
                            if stop.is_set():
                                return
                        free[k].clear()
                        slot = self.slots[k]
                        item = self._launch(b, slot, self._pack(b, slot), ring=True, after=slot.consumed if slot.consumed is not None else start_ev)
                        while True:
                            try:
                                q.put(item + (k,), timeout=0.05)
                                break
                            except queue.Full:
                                if stop.is_set():
                                    return
                        k = (k + 1) % S
                        if stop.is_set():
                            return
                    put_last(None)
                except BaseException as e:      # surfaces in the consumer
                    put_last(e)

            def put_last(item):
                """the end-of-stream mark / the worker's exception: never blocks for good on a consumer that has gone away"""
                while not stop.is_set():
                    try:
                        q.put(item, timeout=0.05)
                        return
                    except queue.Full:
                        pass
            th = threading.Thread(target=work, name='sgg-stager', daemon=True)
            th.start()

            def grab():
                item = q.get()
                if item is None:
                    return None
                if isinstance(item, BaseException):
                    raise item
                self._next = (item[2] + 1) % S
                return item
        try:
            cur = grab()
            while cur is not None:
                nxt = grab()                                                   # copy i+1 is in flight ...
                torch.cuda.current_stream(self.device).wait_event(cur[1].event)
                yield cur[0]                                                   # ... while the caller computes on batch i
                if cur[1].consumed is None:
                    cur[1].consumed = torch.cuda.Event()
                cur[1].consumed.record(torch.cuda.current_stream(self.device))  # the step on batch i has been issued: its slot may be refilled after this point
                if threaded:
                    free[cur[2]].set()
                cur = nxt
        finally:
            if threaded:
                stop.set()
                if th is not None and th is not threading.current_thread():
                    th.join(timeout=5.0)      # no late _launch of this generator's worker beside a following prefetch() / stage() on the same slots
