"""CPU: the input pipeline's host logic — Pillow resampling tables, the transform oracle against
the Pillow goldens, the identity sampler against the reference's, the RGBNT201 directory parser and the decode loader."""
import os
import random

import numpy as np
import pytest
import torch

from ieee_amd.data import datasets, sampler as smp, transforms as T
from oracle import transforms as ot

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "transform_golden.npz"))


def test_oracle_resize_and_normalize_match_pillow_goldens():
    for i, (h, w) in enumerate(GOLD["sizes"]):
        r = ot.pil_bilinear_resize_u8(GOLD["in%d" % i], 256, 128)
        assert np.array_equal(r, GOLD["resized%d" % i]), (h, w)
        if "tensor%d" % i in GOLD.files:
            t = ot.to_tensor_normalize(r, GOLD["mean"], GOLD["std"], bool(GOLD["flip%d" % i]))
            assert np.array_equal(t, GOLD["tensor%d" % i])
    same = ot.pil_bilinear_resize_u8(GOLD["in0"], 256, 128)            # same size: a copy, no filtering
    assert np.array_equal(same, GOLD["in0"]) and same is not GOLD["in0"]


@pytest.mark.parametrize("hs,ws", [(256, 128), (300, 150), (128, 64), (341, 97), (97, 33), (1000, 40), (257, 129), (256, 100)])
def test_host_tables_equal_the_oracle_plan(hs, ws):
    t, p = T.resample_tables(hs, ws, 256, 128), ot.resample_plan(hs, ws, 256, 128)
    assert (t["need_h"], t["need_v"], t["ybox_first"], t["tmp_rows"]) == (p["need_h"], p["need_v"], p["ybox_first"], p["tmp_rows"])
    assert t["ksize_h"] == p["ksize_h"] and t["ksize_v"] == p["ksize_v"]
    for a, b in (("bounds_h", "bh"), ("kk_h", "kh"), ("bounds_v", "bv"), ("kk_v", "kv")):
        assert np.array_equal(t[a], p[b]), a
    # each weight row sums to 1 in fixed point up to rounding, windows stay inside the image
    assert np.all(np.abs(t["kk_h"].sum(1) - (1 << T.PRECISION_BITS)) <= t["ksize_h"])
    assert np.all(t["bounds_v"][:, 0] >= 0) and np.all(t["bounds_v"].sum(1) <= hs)


def test_transform_argument_checks():
    with pytest.raises(ValueError):
        T.DeviceTransform(256, 128, transforms=("random_flip",))
    with pytest.raises(NotImplementedError):
        T.DeviceTransform(256, 128, transforms=["random_erase"])
    tr = T.DeviceTransform(256, 128, "Random_Flip")
    assert tr.flip and not T.DeviceTransform(256, 128, "random_flip", train=False).flip
    torch.manual_seed(4)
    ref = [1 if float(torch.rand(1)) < 0.5 else 0 for _ in range(12)]
    torch.manual_seed(4)
    assert tr.draw_flips(12).tolist() == ref                           # one torch.rand(1) per image, in order


def _source(n_pid=9, per=(3, 4, 5, 9)):
    data, idx = [], 0
    for pid in range(n_pid):
        for _ in range(per[pid % len(per)]):
            data.append((["a", "b", "c"], pid, idx % 4, 0))
            idx += 1
    return data


def test_identity_sampler_batches_hold_k_consecutive_samples_per_identity():
    data = _source()
    s = smp.RandomIdentitySampler(data, batch_size=12, num_instances=4)
    random.seed(1); np.random.seed(1)
    idx = list(iter(s))
    assert len(idx) % 12 == 0 and len(idx) <= len(s) + 12
    for b in range(0, len(idx), 12):
        pids = [data[i][1] for i in idx[b:b + 12]]
        assert all(len(set(pids[k:k + 4])) == 1 for k in range(0, 12, 4)) and len(set(pids)) == 3
    with pytest.raises(ValueError):
        smp.RandomIdentitySampler(data, batch_size=2, num_instances=4)


class _TinyBase(object):
    """dataset stand-in for the ring path: three 4 x 4 'images' per sample that carry the sample's index"""

    def __init__(self, data):
        self.data = data

    def __len__(self):
        return len(self.data)

    def __getitem__(self, i):
        img = np.full((4, 4, 3), i % 251, dtype=np.uint8)
        return {'img': [img, img, img], 'pid': self.data[i][1], 'camid': self.data[i][2], 'impath': str(i), 'timeid': 0}


@pytest.mark.parametrize("workers", [0, 2])
def test_continuous_index_stream_ends_every_epoch_where_the_sampler_ends(workers):
    """RandomIdentitySampler's len() is an UPPER bound: a pass stops once fewer than P identities have a group left.  The
    prefetching loader's one-index-stream-across-epochs mode must cut its epochs where the sampler's passes really end (an
    end-of-pass marker travels through the workers), not after len() batches -- else an epoch borrows its tail from the next
    pass and the boundary drifts.  Here: an identity distribution whose passes are SHORTER than len(); three epochs of the
    continuous stream (through a DataLoader, workers prefetching across the boundary) against the per-epoch iteration."""
    from torch.utils.data import BatchSampler, DataLoader
    from ieee_amd.data import loader as L
    data = _source(n_pid=13, per=(5, 5, 5, 5, 29))          # passes of 5, 4, 6 batches under a len() of 8
    s = smp.RandomIdentitySampler(data, batch_size=12, num_instances=4)
    batches = BatchSampler(s, 12, True)
    random.seed(4); np.random.seed(4)
    want = [[list(b) for b in batches] for _ in range(3)]
    assert all(0 < len(e) < len(batches) for e in want), ([len(e) for e in want], len(batches))   # the case the advisor describes
    assert len(set(len(e) for e in want)) > 1               # ... and the passes differ in length from epoch to epoch
    slots = 9
    ring = torch.zeros((slots, 3, 12, 4, 4, 3), dtype=torch.uint8).share_memory_()
    sampler = L._SlotSampler(batches, slots, continuous=True)
    dl = DataLoader(L._SlotDataset(_TinyBase(data), ring), batch_size=None, sampler=sampler, num_workers=workers,
                    collate_fn=L._identity, persistent_workers=workers > 0,
                    multiprocessing_context=L._worker_context(workers) if workers else None)
    random.seed(4); np.random.seed(4)
    it = iter(dl)
    got = []
    for epoch in range(3):
        got.append([[int(p) for p in item['impath']] for item in L.epoch_items(it)])
    assert got == want
    # the non-continuous form of the same sampler: one pass per iteration, no marker
    random.seed(4); np.random.seed(4)
    plain = [[idx for _, idx in L._SlotSampler(batches, slots, continuous=False)] for _ in range(3)]
    assert plain == want
    del it, dl


@pytest.mark.skipif(not os.path.isdir("/root/reference/torchreid"), reason="reference tree not present")
def test_identity_sampler_reproduces_the_reference_sequence():
    from oracle.ref_import import import_reference
    import_reference()
    from torchreid.data.sampler import RandomIdentitySampler as RefSampler
    data = _source(n_pid=17)
    for seed in (0, 7):
        random.seed(seed); np.random.seed(seed)
        want = list(iter(RefSampler(data, 16, 4)))
        random.seed(seed); np.random.seed(seed)
        got = list(iter(smp.RandomIdentitySampler(data, 16, 4)))
        assert got == want and len(smp.RandomIdentitySampler(data, 16, 4)) == len(RefSampler(data, 16, 4))


def _make_tree(root, names, size=(40, 24)):
    from PIL import Image
    rng = np.random.RandomState(0)
    for split in ("train_171", "test"):
        for mod in ("RGB", "NI", "TI"):
            d = os.path.join(root, "RGBNT201", split, mod)
            os.makedirs(d)
            for n in names:
                Image.fromarray(rng.randint(0, 256, size=size + (3,)).astype(np.uint8), "RGB").save(os.path.join(d, n), quality=95)


def test_rgbnt201_parser_and_decode_loader(tmp_path):
    names = ["000258_cam1_0_00.jpg", "000258_cam3_0_05.jpg", "000301_cam2_0_01.jpg", "000007_cam4_1_02.jpg"]
    _make_tree(str(tmp_path), names)
    ds = datasets.RGBNT201(root=str(tmp_path))
    assert len(ds.train) == 4 and ds.num_train_pids == 3
    by_name = {os.path.basename(d[0][0]): d for d in ds.train}
    assert by_name["000258_cam3_0_05.jpg"][2] == 2 and by_name["000007_cam4_1_02.jpg"][2] == 3
    assert by_name["000258_cam1_0_00.jpg"][1] == by_name["000258_cam3_0_05.jpg"][1]          # relabelled consistently
    assert sorted(set(d[1] for d in ds.train)) == [0, 1, 2]
    assert sorted(set(d[1] for d in ds.query)) == [7, 258, 301]                               # test split keeps raw pids
    for d in ds.train:
        assert [os.path.basename(os.path.dirname(p)) for p in d[0]] == ["RGB", "NI", "TI"]
        assert len(set(os.path.basename(p) for p in d[0])) == 1
    with pytest.raises(RuntimeError):
        datasets.RGBNT201(root=str(tmp_path / "missing"))
    # decode loader with a stand-in transform (the device kernel is exercised in tests/test_data_gpu.py)
    from ieee_amd.data.loader import DeviceLoader

    class Stub(object):
        def draw_flips(self, n):
            return np.zeros(n, dtype=np.uint8)

        def __call__(self, images, flips=None):
            return torch.stack([torch.from_numpy(np.asarray(im).copy()) for im in images])

    batches = list(DeviceLoader(ds.train, Stub(), batch_size=2, workers=0))
    assert len(batches) == 2 and len(batches[0]['img']) == 3
    assert batches[0]['img'][0].shape == (2, 40, 24, 3) and batches[0]['pid'].dtype == torch.int64
    from PIL import Image
    first = batches[0]
    want = np.asarray(Image.open(first['impath'][0][1]).convert('RGB'))
    assert np.array_equal(first['img'][1][0].numpy(), want)


def test_market1501_multimodal_parser(tmp_path):
    """reference data/datasets/image/market_to_RGBNT201.py:14-78: train / query / gallery folders, Market1501 file names,
    junk (-1) skipped, cameras 1..6 -> 0..5, training identities relabelled"""
    from PIL import Image
    split_names = {"train": ["0002_c1s1_000451_03.jpg", "0002_c3s1_000551_01.jpg", "0007_c2s3_071052_01.jpg", "-1_c1s1_000401_03.jpg"],
                   "query": ["0001_c6s1_001051_00.jpg", "0003_c4s6_015641_02.jpg"],
                   "gallery": ["0001_c1s1_001051_00.jpg", "0000_c2s1_000301_00.jpg", "-1_c5s1_000401_03.jpg", "1501_c6s4_001877_01.jpg"]}
    base = tmp_path / "mm"
    for split, names in split_names.items():
        for mod in ("RGB", "NI", "TI"):
            d = base / split / mod
            d.mkdir(parents=True)
            for n in names:
                Image.fromarray(np.zeros((8, 4, 3), dtype=np.uint8), "RGB").save(str(d / n))
    ds = datasets.Market1501MM(root=str(tmp_path), dataset_dir="mm")
    assert datasets.market_to_RGBNT201 is datasets.Market1501MM
    assert len(ds.train) == 3 and ds.num_train_pids == 2 and sorted({r[1] for r in ds.train}) == [0, 1]
    by = {os.path.basename(r[0][0]): r for r in ds.train + ds.query + ds.gallery}
    assert "-1_c1s1_000401_03.jpg" not in by and "-1_c5s1_000401_03.jpg" not in by
    assert by["0002_c3s1_000551_01.jpg"][2] == 2 and by["0001_c6s1_001051_00.jpg"][2] == 5
    assert by["0002_c1s1_000451_03.jpg"][1] == by["0002_c3s1_000551_01.jpg"][1] != by["0007_c2s3_071052_01.jpg"][1]
    assert sorted(r[1] for r in ds.gallery) == [0, 1, 1501] and sorted(r[1] for r in ds.query) == [1, 3]     # raw pids
    for r in ds.gallery:
        assert [os.path.basename(os.path.dirname(p)) for p in r[0]] == ["RGB", "NI", "TI"]
    item = datasets.MultiModalImageDataset(ds.query)[0]
    assert len(item["img"]) == 3 and item["img"][0].shape == (8, 4, 3)
    with pytest.raises(RuntimeError):
        datasets.Market1501MM(root=str(tmp_path))                     # the default folder name is not there
    (base / "train" / "RGB" / "0003_c7s1_000001_00.jpg").write_bytes(b"x")
    with pytest.raises(AssertionError):
        datasets.Market1501MM(root=str(tmp_path), dataset_dir="mm")   # camera 7 does not exist in Market1501


# ---- shard-aware sampler / loader
def test_sharded_sampler_slices_the_single_process_batches():
    # (3, 32): 8 identities = shards of 12, 12, 8 rows; (8, 512): BASELINE config 3 -- 8 ranks x 64 rows out of 171 identities
    for world, B, K in ((2, 16, 4), (3, 32, 4), (4, 16, 4), (8, 512, 4)):
        data = _source(n_pid=171 if B == 512 else 17)
        random.seed(5); np.random.seed(5)
        order = list(iter(smp.RandomIdentitySampler(data, B, K)))
        nb = len(order) // B
        parts = []
        for r in range(world):
            s = smp.build_train_sampler(data, 'RandomIdentitySampler', batch_size=B, num_instances=K, rank=r, world=world)
            assert isinstance(s, smp.ShardedIdentitySampler) and s.global_batch == B
            random.seed(5); np.random.seed(5)
            mine = list(iter(s))
            assert len(mine) == nb * s.local_batch == len(s) and s.local_batch % K == 0
            parts.append((s, mine))
        assert sum(s.local_batch for s, _ in parts) == B
        for g in range(nb):                                         # rank shards, in rank order, ARE the global batch
            got = []
            for s, mine in parts:
                got += mine[g * s.local_batch:(g + 1) * s.local_batch]
            assert got == order[g * B:(g + 1) * B]
    data = _source(n_pid=17)
    with pytest.raises(ValueError):
        smp.build_train_sampler(data, 'RandomSampler', rank=0, world=2)
    with pytest.raises(ValueError):
        smp.ShardedIdentitySampler(smp.RandomIdentitySampler(data, 8, 4), 2, 3)      # 2 identities cannot feed 3 ranks


@pytest.mark.skipif(not os.path.isdir("/root/reference/torchreid"), reason="reference tree not present")
def test_sharded_sampler_is_a_slice_of_the_reference_sequence():
    from oracle.ref_import import import_reference
    import_reference()
    from torchreid.data.sampler import RandomIdentitySampler as RefSampler
    data = _source(n_pid=17)
    random.seed(3); np.random.seed(3)
    want = list(iter(RefSampler(data, 16, 4)))
    for r in range(2):
        s = smp.ShardedIdentitySampler(smp.RandomIdentitySampler(data, 16, 4), r, 2)
        random.seed(3); np.random.seed(3)
        mine = list(iter(s))
        for g in range(len(want) // 16):
            assert mine[g * 8:(g + 1) * 8] == want[g * 16 + 8 * r:g * 16 + 8 * r + 8]


class _FlipStub(object):
    """stand-in device transform that draws flips like DeviceTransform (one torch.rand(1) per image) and records them"""

    def __init__(self):
        self.seen = []

    def draw_flips(self, n):
        return np.asarray([1 if float(torch.rand(1)) < 0.5 else 0 for _ in range(n)], dtype=np.uint8)

    def __call__(self, images, flips=None):
        self.seen.append(np.asarray(flips).copy())
        return torch.zeros(len(images), 1)


def test_rank_shards_flip_like_the_single_process_sequence(tmp_path):
    """The reference draws one flip per image, sample by sample over the GLOBAL batch (data/transforms.py:259-262 inside
    the dataset's __getitem__).  A rank-sharded loader draws the global batch's flips on every rank and keeps its rows, so
    ranks seeded alike reproduce that sequence -- not rank 0's pattern repeated in every shard."""
    from ieee_amd.data.loader import DeviceLoader
    names = ["%06d_cam%d_0_%02d.jpg" % (pid, 1 + (j % 4), j) for pid in (3, 8, 11, 20) for j in range(2)]
    _make_tree(str(tmp_path), names, size=(20, 12))
    ds = datasets.RGBNT201(root=str(tmp_path))
    B, K, world = 8, 2, 2
    random.seed(4); np.random.seed(4)
    single = _FlipStub()                                          # the single-process loader over the same global batches
    one = DeviceLoader(ds.train, single, B, sampler=smp.build_train_sampler(ds.train, 'RandomIdentitySampler', batch_size=B,
                                                                           num_instances=K), workers=0, drop_last=True)
    torch.manual_seed(9)
    whole = next(iter(one))
    want = np.stack(single.seen[:3], axis=1)                      # [sample, modality] of global batch 0
    got, pids = [], []
    for r in range(world):
        random.seed(4); np.random.seed(4)
        samp = smp.build_train_sampler(ds.train, 'RandomIdentitySampler', batch_size=B, num_instances=K, rank=r, world=world)
        stub = _FlipStub()
        loader = DeviceLoader(ds.train, stub, samp.local_batch, sampler=samp, workers=0, drop_last=True, global_rows=B)
        torch.manual_seed(9)
        first = next(iter(loader))
        assert len(first['pid']) == B // world and len(stub.seen) == 3
        got.append(np.stack(stub.seen[:3], axis=1))               # [rows, modality]
        pids.append(first['pid'])
    assert torch.equal(torch.cat(pids), whole['pid'])             # the shards are the single-process batch ...
    assert np.array_equal(np.concatenate(got, 0), want)           # ... and flip like it, row for row
    assert not np.array_equal(got[0], got[1])                     # (the shards do differ under this seed)


def test_sharded_sampler_prepares_outside_iter_and_reports_its_real_length():
    data = _source(n_pid=17)
    s = smp.ShardedIdentitySampler(smp.RandomIdentitySampler(data, 16, 4), 1, 2)
    upper = len(s)                                                # before any draw: the base sampler's upper bound
    random.seed(2); np.random.seed(2)
    assert s.set_epoch(0) is s and s._prepared is not None
    drawn = list(s._prepared)
    state = random.getstate()
    assert list(iter(s)) == drawn and random.getstate() == state  # __iter__ consumed the prepared order: no new draw
    assert len(s) == len(drawn) <= upper
    assert s._prepared is None and len(list(iter(s))) == len(s)   # the next epoch draws again


class _CountingStub(object):
    """stand-in device transform: counts the images it is handed (= decoded and transformed rows)"""

    def __init__(self):
        self.rows = 0

    def draw_flips(self, n):
        return np.zeros(n, dtype=np.uint8)

    def __call__(self, images, flips=None):
        self.rows += len(images)
        return torch.stack([torch.from_numpy(np.asarray(im).copy()).float().mean(dim=(0, 1)) for im in images])


def _shard_worker(rank, world, port, root, ret):
    import torch.distributed as dist
    from ieee_amd import dist as ddp
    from ieee_amd.data.loader import DeviceLoader
    from oracle import model as om
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank),
                      LOCAL_RANK=str(rank))
    ddp.init_from_env(backend="gloo")
    ds = datasets.RGBNT201(root=root)
    B, K, C = 8, 2, ds.num_train_pids
    # --- training loader: this rank's identity-aligned shard of every global batch, and ONLY that is decoded
    random.seed(100 + rank); np.random.seed(100 + rank)            # deliberately different seeds: rank 0's order is broadcast
    samp = smp.build_train_sampler(ds.train, 'RandomIdentitySampler', batch_size=B, num_instances=K, rank=rank, world=world)
    stub = _CountingStub()
    loader = DeviceLoader(ds.train, stub, samp.local_batch, sampler=samp, workers=0, drop_last=True, global_rows=samp.global_batch)
    decoded = []
    real_read = datasets.read_image
    datasets.read_image = lambda p: (decoded.append(p), real_read(p))[1]
    batches = list(loader)
    datasets.read_image = real_read
    assert len(batches) >= 1 and all(b['global_rows'] == B and len(b['pid']) == B // world for b in batches)
    assert len(decoded) == 3 * (B // world) * len(batches) and stub.rows == len(decoded)      # B/world triples per step
    # the summed rank gradients of a small model equal the gradient of the reference's loss on the global batch
    torch.manual_seed(0)
    W = torch.randn(9, C, requires_grad=True)
    Wf = torch.randn(9, 6, requires_grad=True)

    def loss_of(batch, scale):
        x = torch.cat(batch['img'], dim=1)                          # [rows, 9]: three modalities x RGB means
        f = torch.nn.functional.normalize(x @ Wf, dim=1)
        return scale * om.cross_entropy_ls(x @ W / 50.0, batch['pid'], C) + om.margin3m(f[:, 0:2], f[:, 2:4], f[:, 4:6], batch['pid'], 1.0)

    from ieee_amd.engine import Engine

    class _DM(object):
        train_loader, test_loader, num_train_pids, sources = [], {}, C, []
    eng = Engine(_DM(), use_gpu=False)
    flat = torch.zeros(W.numel() + Wf.numel())
    first = batches[0]
    data, total = eng._local_batch(first)                          # stamped batch: no slicing, no collective
    assert data is first and total == B
    loss_of(first, ddp.ce_grad_scale(len(first['pid']), total)).backward()
    flat[:W.numel()] = W.grad.flatten(); flat[W.numel():] = Wf.grad.flatten()
    ddp.allreduce_sum_(flat)
    # serial reference: all ranks' rows of global batch 0 (gathered), one process
    rows = [None] * world
    dist.all_gather_object(rows, {'img': first['img'], 'pid': first['pid']})
    whole = {'img': [torch.cat([r['img'][m] for r in rows]) for m in range(3)], 'pid': torch.cat([r['pid'] for r in rows])}
    assert all(len(set(whole['pid'][k:k + K].tolist())) == 1 for k in range(0, B, K))          # identity-contiguous
    W.grad = None; Wf.grad = None
    loss_of(whole, 1.0).backward()
    torch.testing.assert_close(flat[:W.numel()].view_as(W), W.grad, rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(flat[W.numel():].view_as(Wf), Wf.grad, rtol=1e-5, atol=1e-7)
    # a sampler built for another (rank, world) than the live group refuses to draw instead of disagreeing with its bounds
    wrong = smp.ShardedIdentitySampler(smp.RandomIdentitySampler(ds.train, B, K), (rank + 1) % world, world)
    with pytest.raises(RuntimeError):
        wrong.prepare()
    # --- evaluation loader: every world-th batch is decoded, the labels of ALL batches come from the records
    stub2 = _CountingStub()
    q = DeviceLoader(ds.query, stub2, 3, workers=0, rank=rank, world=world)
    labels = q.batch_labels()
    seen = [int(b['batch_index']) for b in q]
    assert seen == list(range(rank, len(labels), world)) and q.sharded
    assert stub2.rows == 3 * sum(len(labels[b][0]) for b in seen)
    assert sum(len(p) for p, _ in labels) == len(ds.query)
    dist.destroy_process_group()
    ret[rank] = (len(decoded), seen)


def test_two_ranks_decode_half_the_rows_and_sum_to_the_serial_gradient(tmp_path):
    import socket
    import torch.multiprocessing as mp
    names = ["%06d_cam%d_0_%02d.jpg" % (pid, 1 + (j % 4), j) for pid in (3, 8, 11, 20, 21, 30) for j in range(3)]
    _make_tree(str(tmp_path), names, size=(20, 12))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_shard_worker, args=(2, port, str(tmp_path), ret), nprocs=2, join=True)
        out = dict(ret)
        assert len(out) == 2 and out[0][0] == out[1][0]            # both ranks decoded the same (halved) number of files
        assert sorted(out[0][1] + out[1][1]) == list(range(len(out[0][1]) + len(out[1][1])))


def test_prefetch_thread_yields_the_synchronous_batches_and_propagates_errors(tmp_path):
    from ieee_amd.data.loader import DeviceLoader
    names = ["%06d_cam%d_0_%02d.jpg" % (pid, 1 + (j % 4), j) for pid in (3, 8, 11, 20) for j in range(4)]
    _make_tree(str(tmp_path), names, size=(20, 12))
    ds = datasets.RGBNT201(root=str(tmp_path))
    got = {}
    for prefetch in (0, 3):
        random.seed(4); np.random.seed(4); torch.manual_seed(4)
        stub = _FlipStub()
        samp = smp.build_train_sampler(ds.train, 'RandomIdentitySampler', batch_size=8, num_instances=2)
        loader = DeviceLoader(ds.train, stub, 8, sampler=samp, workers=0, drop_last=True, prefetch=prefetch)
        got[prefetch] = ([b['pid'].tolist() for b in loader], [f.tolist() for f in stub.seen])
    assert got[0] == got[3] and len(got[0][0]) >= 2

    class Boom(_FlipStub):
        def __call__(self, images, flips=None):
            raise RuntimeError("transform failed")
    with pytest.raises(RuntimeError, match="transform failed"):
        list(DeviceLoader(ds.train, Boom(), 4, workers=0, prefetch=2))
