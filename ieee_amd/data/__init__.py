"""Input pipeline for the 3-modal datasets: identity sampler, RGBNT201 directory parser, multi-worker
JPEG decode and the device-side Resize / flip / ToTensor / Normalize kernel (reference: torchreid/data/sampler.py,
data/datasets/image/RGBNT201.py, data/datasets/dataset.py:320-351, data/transforms.py:233-326)."""
from .datasets import RGBNT201, Market1501MM, market_to_RGBNT201, MultiModalImageDataset, read_image   # noqa: F401
from .loader import DeviceLoader, build_loaders                      # noqa: F401
from .sampler import RandomIdentitySampler, build_train_sampler      # noqa: F401
from .transforms import DeviceTransform, build_transforms, resample_tables   # noqa: F401
