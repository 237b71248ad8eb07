"""GPU: the device-side Resize / flip / ToTensor / Normalize kernel (ieee_resize_flip_normalize) is BIT-EXACT against
the Pillow goldens and the oracle chain, for equal, up- and down-scaled sources, mixed sizes in one batch, and through
the JPEG decode loader."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "transform_golden.npz"))


def test_device_transform_is_bit_exact_against_pillow_goldens():
    from ieee_amd.data import DeviceTransform
    from oracle import transforms as ot
    tr = DeviceTransform(256, 128, "random_flip")
    n = len(GOLD["sizes"])
    for i in range(n):
        flip = bool(GOLD["flip%d" % i])
        got = tr([GOLD["in%d" % i]], flips=[1 if flip else 0]).cpu().numpy()[0]
        want = ot.to_tensor_normalize(GOLD["resized%d" % i], GOLD["mean"], GOLD["std"], flip)
        assert np.array_equal(got, want), tuple(GOLD["sizes"][i])
        if "tensor%d" % i in GOLD.files:
            assert np.array_equal(got, GOLD["tensor%d" % i])
    # every size in ONE call (grouped by source size inside), alternating flips
    flips = [i % 2 for i in range(n)]
    got = tr([GOLD["in%d" % i] for i in range(n)], flips=flips).cpu().numpy()
    for i in range(n):
        want = ot.to_tensor_normalize(GOLD["resized%d" % i], GOLD["mean"], GOLD["std"], bool(flips[i]))
        assert np.array_equal(got[i], want)
    assert tr([]).shape == (0, 3, 256, 128)
    with pytest.raises(ValueError):
        tr([np.zeros((4, 4), dtype=np.uint8)])


def test_random_sizes_against_the_oracle_and_custom_normalisation():
    from ieee_amd.data import DeviceTransform
    from oracle import transforms as ot
    rng = np.random.RandomState(8)
    mean, std = [0.5, 0.4, 0.3], [0.2, 0.25, 0.3]
    tr = DeviceTransform(64, 48, [], norm_mean=mean, norm_std=std, train=False)
    imgs = [rng.randint(0, 256, size=(int(h), int(w), 3)).astype(np.uint8)
            for h, w in ((64, 48), (65, 47), (31, 96), (200, 17), (64, 200), (130, 48), (1, 1), (2, 300))]
    got = tr(imgs).cpu().numpy()
    for g, im in zip(got, imgs):
        want = ot.to_tensor_normalize(ot.pil_bilinear_resize_u8(im, 64, 48), mean, std, False)
        assert np.array_equal(g, want), im.shape


def test_decode_loader_end_to_end(tmp_path):
    """JPEG tree -> RGBNT201 parser -> identity sampler -> worker decode -> device transform == the reference's chain
    (PIL decode, PIL resize, flip, to_tensor, normalize) restated by the oracle on the same files and flip draws"""
    from PIL import Image
    from ieee_amd import data as D
    from oracle import transforms as ot
    rng = np.random.RandomState(1)
    names = ["%06d_cam%d_0_%02d.jpg" % (pid, 1 + k % 4, k) for pid in (3, 9, 20) for k in range(4)]
    for split in ("train_171", "test"):
        for mod in ("RGB", "NI", "TI"):
            d = os.path.join(str(tmp_path), "RGBNT201", split, mod)
            os.makedirs(d)
            for nme in names:
                Image.fromarray(rng.randint(0, 256, size=(70, 30, 3)).astype(np.uint8), "RGB").save(os.path.join(d, nme), quality=92)
    ds = D.RGBNT201(root=str(tmp_path))
    train, query, _ = D.build_loaders(ds, 256, 128, "random_flip", batch_size_train=8, batch_size_test=5, workers=2)
    drawn = []
    orig = train.transform.draw_flips
    train.transform.draw_flips = lambda n: drawn.append(orig(n)) or drawn[-1]      # record the loader's flip draws
    batch = next(iter(train))
    assert len(batch['img']) == 3 and batch['img'][0].shape == (8, 3, 256, 128) and batch['img'][0].is_cuda
    pids = batch['pid'].tolist()
    assert pids[0:4] == [pids[0]] * 4 and pids[4:8] == [pids[4]] * 4
    flips = drawn[0].reshape(8, 3)               # sample-major, modality-minor: the reference's draw order
    assert flips.shape == (8, 3) and set(np.unique(flips)) <= {0, 1}
    for i in range(8):
        for m in range(3):
            im = np.asarray(Image.open(batch['impath'][i][m]).convert('RGB'))
            want = ot.to_tensor_normalize(ot.pil_bilinear_resize_u8(im, 256, 128), D.transforms.IMAGENET_MEAN,
                                          D.transforms.IMAGENET_STD, bool(flips[i, m]))
            assert np.array_equal(batch['img'][m][i].cpu().numpy(), want)
    q = next(iter(query))
    assert q['img'][2].shape == (5, 3, 256, 128)


def test_prefetched_loader_yields_the_same_batches_bit_for_bit(tmp_path):
    """DeviceLoader(prefetch = 2): a background thread copies + transforms on its own stream while the consumer works; the
    batches (rows, flips, pixels) are those of the synchronous loader under the same seeds"""
    import random
    from PIL import Image
    from ieee_amd import data as D
    rng = np.random.RandomState(2)
    names = ["%06d_cam%d_0_%02d.jpg" % (pid, 1 + k % 4, k) for pid in (3, 9, 20, 31) for k in range(4)]
    for split in ("train_171", "test"):
        for mod in ("RGB", "NI", "TI"):
            d = os.path.join(str(tmp_path), "RGBNT201", split, mod)
            os.makedirs(d)
            for nme in names:
                Image.fromarray(rng.randint(0, 256, size=(64, 32, 3)).astype(np.uint8), "RGB").save(os.path.join(d, nme), quality=92)
    ds = D.RGBNT201(root=str(tmp_path))
    seen = {}
    for prefetch in (0, 2):
        random.seed(3); np.random.seed(3); torch.manual_seed(3)
        train, _, _ = D.build_loaders(ds, 256, 128, "random_flip", batch_size_train=8, workers=2, prefetch=prefetch)
        assert getattr(train, "_continuous", False) == (prefetch > 0)
        out = []
        # THREE epochs: the prefetching loader keeps ONE index stream across epochs (the workers start on epoch k + 1 while
        # epoch k is still being consumed, its order drawn early) -- same batches, same flips, same pixels, epoch after epoch
        for epoch in range(3):
            n = 0
            for b in train:
                junk = torch.randn(512, 512, device="cuda") @ torch.randn(512, 512, device="cuda")    # the consumer's own work
                out.append((b['pid'].clone(), [x.clone() for x in b['img']], junk.sum()))
                n += 1
            assert n == len(train)
        torch.cuda.synchronize()
        seen[prefetch] = out
    assert len(seen[0]) == len(seen[2]) >= 6
    for (p0, x0, _), (p2, x2, _) in zip(seen[0], seen[2]):
        assert torch.equal(p0, p2) and all(torch.equal(a, b) for a, b in zip(x0, x2))
    # an epoch abandoned half way is dropped: the next one starts with a fresh draw, complete and in order
    it = iter(train)
    first = next(it)
    del it
    again = [b['pid'].clone() for b in train]
    assert len(again) == len(train) and sorted(torch.cat(again).tolist()) == sorted(torch.cat([p for p, _, _ in seen[2][:len(train)]]).tolist())


def test_continuous_loader_keeps_the_samplers_epoch_boundaries(tmp_path, monkeypatch):
    """identities with unequal image counts: a RandomIdentitySampler pass ends BEFORE len() batches.  The prefetching loader
    (one index stream across epochs) must end its epochs exactly there -- same number of steps per epoch and the same batches,
    epoch by epoch, as the per-epoch iterator (IEEE_LOADER_CONTINUOUS=0)."""
    import random
    from PIL import Image
    from ieee_amd import data as D
    rng = np.random.RandomState(5)
    counts = {3: 4, 9: 4, 20: 16, 31: 4, 40: 8, 41: 4, 57: 4}
    names = ["%06d_cam%d_0_%02d.jpg" % (pid, 1 + k % 4, k) for pid, c in counts.items() for k in range(c)]
    for split in ("train_171", "test"):
        for mod in ("RGB", "NI", "TI"):
            d = os.path.join(str(tmp_path), "RGBNT201", split, mod)
            os.makedirs(d)
            for nme in names:
                Image.fromarray(rng.randint(0, 256, size=(32, 16, 3)).astype(np.uint8), "RGB").save(os.path.join(d, nme), quality=90)
    ds = D.RGBNT201(root=str(tmp_path))
    seen = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("IEEE_LOADER_CONTINUOUS", mode)
        random.seed(6); np.random.seed(6); torch.manual_seed(6)
        train, _, _ = D.build_loaders(ds, 256, 128, "random_flip", batch_size_train=8, workers=2, prefetch=2)
        assert getattr(train, "_continuous", False) == (mode == "1")
        seen[mode] = [[b['pid'].tolist() for b in train] for _ in range(4)]
        upper = len(train)
        del train
    assert seen["0"] == seen["1"]
    lens = [len(e) for e in seen["0"]]
    print("batches per epoch:", lens, "len(loader):", upper)
    assert all(0 < n <= upper for n in lens) and any(n < upper for n in lens)      # the passes really are shorter than len()
