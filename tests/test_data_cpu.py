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
