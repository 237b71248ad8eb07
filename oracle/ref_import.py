"""Import harness for the *real* reference (THIS container only; /root/reference
does not exist on the GPU box).  Used by tests/golden/gen_*.py to produce golden
vectors and by CPU tests (skipped when the reference is absent) to validate the
oracle's restatements.  Recipe from SURVEY.md §8c: stub packages the image
lacks, never write bytecode into the read-only tree."""
import os
import sys
import warnings

REF_ROOT = "/root/reference"


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "torchreid"))


def import_reference():
    if not available():
        raise RuntimeError("reference tree not present")
    warnings.filterwarnings("ignore")
    from unittest.mock import MagicMock
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    sys.dont_write_bytecode = True
    for n in ["cv2", "torchvision", "torchvision.transforms", "torchvision.models", "yacs",
              "yacs.config", "tensorboard", "torch.utils.tensorboard", "h5py", "gdown", "imageio",
              "numpy.lib.function_base", "numpy.lib.twodim_base", "numpy.lib.type_check",
              "numpy.core.getlimits", "numpy.core.fromnumeric", "numpy.core.records"]:
        if n not in sys.modules:
            m = MagicMock(name=n)
            m.__name__ = n
            m.__path__ = []
            m.__spec__ = None
            sys.modules[n] = m
    import torchreid  # noqa: F401
    return torchreid
