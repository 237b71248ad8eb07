"""Batches for the engines: CPU workers only decode JPEGs (the reference runs its whole transform chain in the loader
process with workers=0, configs/RGBNT_ieee_part_margin.yaml:13); the decoded bytes are resized / flipped / normalised on
the GPU per batch.  Yields the reference's batch dict: {'img': [RGB, NI, TI] float tensors [B,3,H,W] (on the device),
'pid', 'camid', 'impath', 'timeid'} (data/datasets/dataset.py:344-351)."""
import multiprocessing
import os
import queue
import sys
import threading

import numpy as np
import torch
from torch.utils.data import DataLoader

from .datasets import MultiModalImageDataset
from .sampler import build_train_sampler
from .transforms import build_transforms


def _worker_context(workers):
    """How the decode workers are started.  NOT by fork() from the training process when it can be avoided: forking a process
    that holds a live HIP context write-protects its page tables, the driver's MMU notifiers evict and re-validate the
    process's GPU-visible host memory, and its queues stand still meanwhile -- measured on the MI355X box
    (scripts/small_copy_probe.py): with a forked child alive the next 5 train steps take 364 ms each instead of 14.7, the
    next twenty 1 KB host->device copies 78 ms each; a loader lifecycle that costs 0.3 s in a fresh process costs 10-15 s after
    a train step has run in it.  A fork server (started through vfork + exec, preloaded with torch) hands out workers that
    share nothing with the training process but the shared-memory ring.  It needs what every `spawn` start needs -- a
    `__main__` that can be imported again without side effects (`if __name__ == "__main__":`, as scripts/mainMultiModal.py
    has) -- so an interactive / stdin `__main__` falls back to fork.  IEEE_LOADER_START = fork | forkserver | spawn overrides."""
    if workers <= 0:
        return None
    want = os.environ.get("IEEE_LOADER_START", "").lower()
    if want not in ("fork", "forkserver", "spawn"):
        main = sys.modules.get("__main__")
        path = getattr(main, "__file__", None)
        importable = getattr(main, "__spec__", None) is not None or (path is not None and os.path.isfile(path))
        want = "forkserver" if importable else "fork"
    if want == "forkserver":
        try:
            multiprocessing.set_forkserver_preload(["torch", "numpy", "PIL.Image", "ieee_amd.data.datasets"])
        except Exception:
            pass
    return multiprocessing.get_context(want)


def _collate(items):
    """Runs in the WORKER.  When the batch's images of a modality all have one size (every dataset of the 3-modal configs:
    RGBNT201 is 256x128 throughout) they leave the worker as ONE uint8 tensor [B, H, W, 3] per modality -- torch tensors
    cross the worker pipe through shared memory, a list of 192 numpy arrays is pickled and copied -- else as the list."""
    raw = [it['img'] for it in items]                                 # [sample][modality] uint8 arrays
    mods = len(raw[0]) if raw else 0
    if raw and all(len(set(r[m].shape for r in raw)) == 1 for m in range(mods)):
        raw = [torch.from_numpy(np.stack([r[m] for r in raw])) for m in range(mods)]       # [modality] -> [B, H, W, 3]
    return {'img': raw,
            'pid': torch.as_tensor([it['pid'] for it in items], dtype=torch.int64),
            'camid': torch.as_tensor([it['camid'] for it in items], dtype=torch.int64),
            'impath': [it['impath'] for it in items],
            'timeid': torch.as_tensor([it['timeid'] for it in items], dtype=torch.int64)}


class _SlotDataset(torch.utils.data.Dataset):
    """Ring path (DeviceLoader(prefetch > 0, workers > 0)): a worker decodes a WHOLE batch straight into slot `slot` of a
    shared uint8 ring [slots][modality][B][H][W][3] -- one mapping, inherited by the forked workers and registered with the
    HIP runtime as pinned host memory by the parent -- and sends back only the labels.  What the tensor path pays per batch
    in the parent (three file descriptors received and mapped, a 19 MB copy into pinned memory by the DataLoader's pin
    thread, all under the interpreter lock the train step also needs) disappears: measured on the GPU box with the B = 64
    step running beside it, scripts/loader_probe.py.  A batch whose images do not all have the ring's size travels as the
    list (`img` not None) and takes the general path."""

    def __init__(self, base, ring):
        self.base, self.ring = base, ring
        self.view = None                       # numpy view of the shared mapping, made in the worker

    def __getstate__(self):                    # (a numpy view would be pickled by value: the worker makes its own)
        return {"base": self.base, "ring": self.ring, "view": None}

    def __len__(self):
        return len(self.base)

    def __getitem__(self, key):
        slot, indices = key
        if slot < 0:                           # _SlotSampler's end-of-pass marker: nothing to decode, it only travels in order
            return {'end_of_pass': True}
        if self.view is None:
            self.view = self.ring.numpy()
        items = [self.base[i] for i in indices]
        mods, (H, W) = self.view.shape[1], self.view.shape[3:5]
        fits = len(items) <= self.view.shape[2] and all(len(it['img']) == mods and all(im.shape == (H, W, 3) for im in it['img'])
                                                         for it in items)
        if fits:
            for j, it in enumerate(items):
                for m in range(mods):
                    self.view[slot, m, j] = it['img'][m]
        return {'slot': slot, 'rows': len(items), 'img': None if fits else [it['img'] for it in items],
                'pid': torch.as_tensor([it['pid'] for it in items], dtype=torch.int64),
                'camid': torch.as_tensor([it['camid'] for it in items], dtype=torch.int64),
                'impath': [it['impath'] for it in items],
                'timeid': torch.as_tensor([it['timeid'] for it in items], dtype=torch.int64)}


class _SlotSampler(object):
    """(slot, dataset indices) per batch; the slot counter runs on across epochs, so a slot is rewritten only after
    `slots` further batches have been handed out.
    continuous: the index stream does not end with the epoch -- epoch k + 1's batches follow epoch k's at once, so the workers
    (each decodes a WHOLE batch: ~55 ms for 64 triples) are already busy with the next epoch while the consumer finishes this
    one.  With an iterator per epoch the pipeline drains and refills at every boundary: one 50-60 ms stall per epoch, measured
    (scripts/loader_probe.py: the single longest wait of every 40-step window, 3-4 steps of GPU time).  The next epoch's order
    is then drawn ~2 x workers batches EARLY, from the same generators in the same order -- the sequence of batches is the
    reference sampler's as long as nothing else consumes python's / numpy's global generators in between.
    A pass ends where the inner batch sampler ENDS, not after len() batches: RandomIdentitySampler's length is an upper bound
    (it stops once fewer than P identities have a group left, data/sampler.py), so every pass is followed by an END marker
    (slot -1) that travels through the workers in order and tells the consumer where the epoch stops (`epoch_items`) -- the
    epoch boundaries, the optimizer steps per epoch and which pass a batch comes from are the non-continuous loader's."""
    END = (-1, [])

    def __init__(self, batches, slots, continuous=False):
        self.batches, self.slots, self.k, self.continuous = batches, slots, 0, continuous

    def __iter__(self):
        while True:
            n = 0
            for idx in self.batches:
                yield (self.k % self.slots, list(idx))
                self.k += 1
                n += 1
            if not self.continuous:
                return
            yield self.END
            if n == 0:
                return

    def __len__(self):
        return len(self.batches)


def epoch_items(it):
    """the items of ONE pass out of a continuous index stream: everything up to the next end-of-pass marker (which is
    consumed); a stream that ends without one ends the epoch too"""
    for item in it:
        if isinstance(item, dict) and item.get('end_of_pass'):
            return
        yield item


def _identity(x):
    return x


class DeviceLoader(object):
    """rank / world (evaluation loaders): batch b belongs to rank b % world; only the owned batches
    are decoded and transformed -- the other ranks' rows never leave the disk here -- and every yielded batch says which
    global batch it is (`batch_index`).  `batch_labels()` lists (pids, camids) of ALL batches from the dataset records, so
    the evaluator can lay out the gathered descriptor matrix without touching an image.
    global_rows (training loader fed by a ShardedIdentitySampler): the size of the global batch this rank's batches are
    shards of, stamped on every batch so the engine needs no collective to scale the cross entropy."""

    def __init__(self, data, transform, batch_size, sampler=None, shuffle=False, workers=4, drop_last=False, rank=0, world=1,
                 global_rows=None, prefetch=0, persistent=False):
        self.dataset = MultiModalImageDataset(data)
        self.transform = transform
        # prefetch = k > 0: a background thread keeps up to k batches READY ON THE DEVICE -- it takes the decoded batch from
        # the workers (pinned), draws the flips, and runs the copy + resize / flip / normalise kernel on its own HIP stream
        # while the caller's stream runs the previous train step; the consumer only waits on an event.  Same batches, same
        # flips, same bits as prefetch = 0 (the thread is then the only drawer of flips, in batch order).
        self.prefetch = int(prefetch)
        self.rank, self.world = int(rank), int(world)
        self.global_rows = global_rows
        self.batch_size = int(batch_size)
        # persistent: keep the worker processes between epochs (the train loader: an epoch of RGBNT201 is a few hundred batches,
        # respawning W processes that import torch at every epoch costs seconds).  Off for the query / gallery loaders: their
        # workers would sit resident -- 2 x W processes preloaded with torch -- through the whole run for one pass per evaluation.
        persistent = bool(persistent) and workers > 0
        self._owned = None
        if self.world > 1 and sampler is None and not shuffle:
            n = len(self.dataset)
            self._all = [list(range(s, min(s + self.batch_size, n))) for s in range(0, n, self.batch_size)]
            if drop_last and self._all and len(self._all[-1]) < self.batch_size:
                self._all.pop()
            self._owned = list(range(self.rank, len(self._all), self.world))
            self.loader = DataLoader(self.dataset, batch_sampler=[self._all[b] for b in self._owned], num_workers=workers,
                                     collate_fn=_collate, pin_memory=self._pin(), persistent_workers=persistent,
                                     multiprocessing_context=_worker_context(workers))
        elif self.prefetch > 0 and workers > 0 and torch.cuda.is_available() and self._make_ring(sampler, shuffle, drop_last, workers):
            pass                                                         # self.loader / self.ring set by _make_ring
        else:
            self.loader = DataLoader(self.dataset, batch_size=batch_size, sampler=sampler, shuffle=shuffle and sampler is None,
                                     num_workers=workers, collate_fn=_collate, drop_last=drop_last, pin_memory=self._pin(),
                                     multiprocessing_context=_worker_context(workers), persistent_workers=persistent)

    ring = None

    def _make_ring(self, sampler, shuffle, drop_last, workers):
        """the shared pinned ring + a DataLoader whose workers fill it (see _SlotDataset); False: keep the tensor path"""
        if len(self.dataset) == 0:
            return False
        first = self.dataset[0]['img']
        shapes = set(np.asarray(im).shape for im in first)
        if len(shapes) != 1 or len(next(iter(shapes))) != 3 or next(iter(shapes))[2] != 3:
            return False
        H, W, _ = next(iter(shapes))
        slots = workers * 2 + self.prefetch + 3          # in flight: 2 per worker (prefetch_factor) + the ready queue + 1 in use
        ring = torch.empty((slots, len(first), self.batch_size, H, W, 3), dtype=torch.uint8).share_memory_()
        try:                                             # hipHostRegister: async H2D copies straight out of the shared mapping
            rc = torch.cuda.cudart().cudaHostRegister(ring.data_ptr(), ring.numel(), 0)
            if int(rc) != 0:
                return False
        except Exception:
            return False
        self.ring, self._ring_registered = ring, True
        base = torch.utils.data.RandomSampler(self.dataset) if (shuffle and sampler is None) else (
            sampler if sampler is not None else torch.utils.data.SequentialSampler(self.dataset))
        batches = torch.utils.data.BatchSampler(base, self.batch_size, drop_last)
        # (a rank-sharded sampler exchanges its epoch order with a collective in prepare(): that stays at the epoch boundary,
        # in the consumer's thread, in step with the other ranks)
        self._continuous = os.environ.get("IEEE_LOADER_CONTINUOUS", "1") != "0" and not hasattr(base, "prepare")
        self._it, self._epoch_pos = None, 0
        self._slot_sampler = _SlotSampler(batches, slots, continuous=self._continuous)
        self.loader = DataLoader(_SlotDataset(self.dataset, ring), batch_size=None, sampler=self._slot_sampler, num_workers=workers,
                                 collate_fn=_identity, pin_memory=False, persistent_workers=True,
                                 multiprocessing_context=_worker_context(workers))
        self.loader_base_sampler = base
        return True

    def __del__(self):
        try:
            if getattr(self, "_ring_registered", False):
                self.loader = None                       # stop the workers before the mapping goes away
                torch.cuda.cudart().cudaHostUnregister(self.ring.data_ptr())
                self._ring_registered = False
        except Exception:
            pass

    @property
    def sharded(self):
        return self._owned is not None

    def batch_labels(self):
        """[(pids, camids)] of every global batch, in loader order, from the records alone (no decode)"""
        assert self._owned is not None, "batch_labels() is for rank-sharded sequential loaders"
        recs = self.dataset.data
        return [([recs[i][1] for i in idx], [recs[i][2] for i in idx]) for idx in self._all]

    def __len__(self):
        return len(self.loader)

    def _pin(self):
        return self.prefetch > 0 and torch.cuda.is_available()

    def __iter__(self):
        if self.prefetch <= 0:
            return self._batches()
        return self._prefetched()

    def _prefetched(self):
        """the batches of _batches(), produced `prefetch` ahead by a thread on its own stream"""
        cuda = torch.cuda.is_available()
        dev = torch.cuda.current_device() if cuda else None
        # high priority: the copy + transform of a batch is ~0.1 ms of GPU work that must not queue behind a 15 ms step
        # (one stream per loader, kept across epochs: stream churn in the middle of training re-shuffles the runtime's
        # hardware-queue assignment, LABNOTES.md round-4 log)
        if cuda and getattr(self, "_side_stream", None) is None:
            self._side_stream = torch.cuda.Stream(device=dev, priority=-1)
        side = self._side_stream if cuda else None
        # one producer at a time: a prefetch thread of an earlier, abandoned iteration (the consumer broke out of the epoch) may
        # still be inside the DataLoader waiting for a worker -- it sees its stop flag at the next batch.  A second thread on the
        # same persistent loader would iterate it concurrently and interleave the flip draws: wait for the old one, and refuse
        # to go on if it does not end.
        old = getattr(self, "_prefetch_thread", None)
        if old is not None and old.is_alive():
            old.join(timeout=120.0)
            if old.is_alive():
                raise RuntimeError("DeviceLoader: the prefetch thread of the previous iteration is still running (a decode worker "
                                   "does not answer); not starting a second producer on the same loader")
        box = queue.Queue(maxsize=self.prefetch)
        stop = threading.Event()

        def put(item):
            while not stop.is_set():
                try:
                    box.put(item, timeout=0.1)
                    return True
                except queue.Full:
                    continue
            return False

        def work():
            try:
                if cuda:
                    torch.cuda.set_device(dev)
                    with torch.cuda.stream(side):
                        recent = []
                        for batch in self._batches():
                            ev = torch.cuda.Event()
                            ev.record(side)
                            # ring path: a slot is rewritten `slots` batches later; never leave more than prefetch + 2 copies
                            # unfinished, so the copy out of a slot is long complete when a worker writes to it again
                            recent.append(ev)
                            if len(recent) > self.prefetch + 2:
                                recent.pop(0).synchronize()
                            if not put((batch, ev)):
                                return
                else:
                    for batch in self._batches():
                        if not put((batch, None)):
                            return
                put(None)
            except BaseException as e:          # hand the failure to the consumer instead of dying silently
                put(e)

        th = self._prefetch_thread = threading.Thread(target=work, name="ieee-loader-prefetch", daemon=True)
        th.start()
        try:
            while True:
                item = box.get()
                if item is None:
                    return
                if isinstance(item, BaseException):
                    raise item
                batch, ev = item
                if ev is not None:
                    cur = torch.cuda.current_stream()
                    cur.wait_event(ev)                    # the consumer's stream, not the host, waits for the transform
                    for t in list(batch['img']) + [batch.get('pid_dev')]:
                        if torch.is_tensor(t) and t.is_cuda:
                            t.record_stream(cur)          # allocated on the side stream, used (and later freed) on this one
                yield batch
        finally:
            stop.set()
            th.join(timeout=5.0)          # (normally immediate; if it is still inside the DataLoader the next __iter__ waits for it)

    def _epoch_source(self):
        """the decoded batches of ONE epoch.  Continuous ring path: the items up to the sampler's end-of-pass marker out of the
        one iterator that lives across epochs (an epoch the consumer abandoned half way is dropped: fresh iterator, fresh draw
        of the order)"""
        if self.ring is None or not getattr(self, "_continuous", False):
            return iter(self.loader)

        def one_epoch():
            if self._it is None or self._epoch_pos != 0:
                self._it = iter(self.loader)
            self._epoch_pos = 0
            for k, item in enumerate(epoch_items(self._it)):      # up to the sampler's REAL end of pass (its marker)
                self._epoch_pos = k + 1          # (a consumer that stops here leaves a position != 0 behind)
                yield item
            self._epoch_pos = 0
        return one_epoch()

    def _batches(self):
        sampler = getattr(self, "loader_base_sampler", None) if self.ring is not None else getattr(self.loader, "sampler", None)
        if hasattr(sampler, "prepare"):          # rank-sharded identity sampler: draw / exchange the epoch's order here, in
            sampler.prepare()                    # the main process, not at the DataLoader's first prefetch
        lo, hi = getattr(sampler, "lo", None), getattr(sampler, "hi", None)
        for k, batch in enumerate(self._epoch_source()):
            raw = batch['img']
            if raw is None:                                          # ring path: the images sit in the pinned ring slot
                rows, slot = int(batch.pop('rows')), int(batch.pop('slot'))
                raw = [self.ring[slot, m, :rows] for m in range(self.ring.shape[1])]
            elif 'slot' in batch:                                    # ring path, odd-sized batch: the list travelled
                batch.pop('slot'), batch.pop('rows')
            stacked = bool(raw) and torch.is_tensor(raw[0])          # [modality] -> [B, H, W, 3] (see _collate)
            n, mods = (int(raw[0].shape[0]), len(raw)) if stacked else (len(raw), len(raw[0]))
            # the reference transforms sample by sample, modality by modality: draw the flips in that order
            if self.global_rows is not None and lo is not None and hi - lo == n:
                # a shard of a global batch: draw the flips of the WHOLE global batch -- every rank consumes the same
                # torch RNG stream, as the single-process loop would -- and keep this shard's rows; ranks seeded alike
                # (torch.manual_seed, as the reference's set_random_seed does) then augment the global batch exactly
                # like the single-process sequence instead of every shard repeating rank 0's pattern
                flips = self.transform.draw_flips(int(self.global_rows) * mods).reshape(int(self.global_rows), mods)[lo:hi]
            else:
                flips = self.transform.draw_flips(n * mods).reshape(n, mods)
            batch['img'] = [self.transform(raw[m] if stacked else [raw[i][m] for i in range(n)], flips=flips[:, m])
                            for m in range(mods)]
            if torch.cuda.is_available() and torch.is_tensor(batch['img'][0]) and batch['img'][0].is_cuda:
                # the identity labels follow the images to the device on the same stream (pinned, asynchronous): the engine's
                # own `pids.cuda()` is a blocking copy from pageable memory -- it drains the launch stream every step and the
                # host can no longer enqueue step k + 1 while step k runs (4 ms per step at B = 64, scripts/loader_probe.py)
                batch['pid_dev'] = batch['pid'].pin_memory().to(batch['img'][0].device, non_blocking=True)
            if self._owned is not None:
                batch['batch_index'] = self._owned[k]
            if self.global_rows is not None:
                batch['global_rows'] = int(self.global_rows)
            yield batch


def build_loaders(dataset, height=256, width=128, transforms='random_flip', batch_size_train=8, batch_size_test=100,
                  train_sampler='RandomIdentitySampler', num_instances=4, workers=4, norm_mean=None, norm_std=None,
                  rank=None, world=None, prefetch=2):
    """train / query / gallery loaders for an RGBNT201-style dataset object (reference data/datamanager.py:158-245).
    rank / world default to the process group's (ieee_amd.dist).  With several ranks batch_size_train is the GLOBAL batch:
    the train loader yields this rank's identity-aligned shard of every global batch (ShardedIdentitySampler; the engine
    sees `global_rows` in the batch and skips its own slicing), and the query / gallery loaders decode every world-th batch.
    prefetch: batches the train loader keeps ready on the device ahead of the step (DeviceLoader; 0 = synchronous)."""
    from .. import dist as ddp
    world = ddp.world_size() if world is None else int(world)
    rank = ddp.rank() if rank is None else int(rank)
    tr, te = build_transforms(height, width, transforms, norm_mean, norm_std)
    sampler = build_train_sampler(dataset.train, train_sampler, batch_size=batch_size_train, num_instances=num_instances,
                                  rank=rank, world=world)
    if world > 1:
        train = DeviceLoader(dataset.train, tr, sampler.local_batch, sampler=sampler, workers=workers, drop_last=True,
                             global_rows=sampler.global_batch, prefetch=prefetch, persistent=True)
    else:
        train = DeviceLoader(dataset.train, tr, batch_size_train, sampler=sampler, workers=workers, drop_last=True,
                             prefetch=prefetch, persistent=True)
    query = DeviceLoader(dataset.query, te, batch_size_test, workers=workers, rank=rank, world=world)
    gallery = DeviceLoader(dataset.gallery, te, batch_size_test, workers=workers, rank=rank, world=world)
    return train, query, gallery
