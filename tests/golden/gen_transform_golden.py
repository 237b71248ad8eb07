"""Generates tests/golden/transform_golden.npz from Pillow (the library behind the reference's
torchvision.transforms.Resize / ToTensor on PIL images; torchreid/data/transforms.py:233-326).  Run in this container:
    python tests/golden/gen_transform_golden.py
Inputs are seeded random uint8 images (incl. smooth ones); outputs are PIL's Image.resize((W, H), BILINEAR) bytes and
the normalised tensor computed with torch exactly as torchvision's functional to_tensor / normalize do
(img.permute(2,0,1).to(float32).div(255); tensor.sub_(mean).div_(std))."""
import numpy as np
import torch
from PIL import Image

H, W = 256, 128
MEAN, STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
CASES = [(256, 128), (300, 150), (128, 64), (341, 97), (200, 128), (256, 100), (97, 33), (400, 40)]
TENSOR_CASES = (0, 3)      # the float tensors are 393 KB each: keep two (one flipped, one not)


def main():
    rng = np.random.RandomState(2022)
    out = {"sizes": np.asarray(CASES, dtype=np.int32), "mean": np.asarray(MEAN), "std": np.asarray(STD)}
    for i, (h, w) in enumerate(CASES):
        img = rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
        if i % 2 == 1:      # a smooth image as well (photographs are not white noise)
            yy, xx = np.mgrid[0:h, 0:w]
            img = np.stack([(yy * 255 // max(h - 1, 1)), (xx * 255 // max(w - 1, 1)), ((yy + xx) % 256)], -1).astype(np.uint8)
        res = np.asarray(Image.fromarray(img, "RGB").resize((W, H), Image.BILINEAR))
        flip = bool(i % 3 == 0)
        pil = Image.fromarray(res, "RGB")
        if flip:
            pil = pil.transpose(Image.FLIP_LEFT_RIGHT)      # torchvision F.hflip on a PIL image
        t = torch.from_numpy(np.asarray(pil).copy()).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
        t = t.sub_(torch.tensor(MEAN).view(3, 1, 1)).div_(torch.tensor(STD).view(3, 1, 1))
        out["in%d" % i] = img
        out["resized%d" % i] = res
        out["flip%d" % i] = np.asarray(flip)
        if i in TENSOR_CASES:
            out["tensor%d" % i] = t.numpy()
    np.savez_compressed("tests/golden/transform_golden.npz", **out)
    print("wrote tests/golden/transform_golden.npz")


if __name__ == "__main__":
    main()
