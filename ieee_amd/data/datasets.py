"""RGBNT201 / Market1501-multimodal directory parsers and the 3-modal dataset (reference data/datasets/image/RGBNT201.py:13-79,
data/datasets/image/market_to_RGBNT201.py:14-78, data/datasets/dataset.py:320-351, utils/tools.py:98-119).  On-disk format: <root>/RGBNT201/{train_171,test}/{RGB,NI,TI}/
<pid6>_cam<X>_....jpg, the same file name in the three modality folders.  __getitem__ returns the DECODED uint8 images
(PIL 'RGB'); resize / flip / normalisation happen on the device for the whole batch (transforms.DeviceTransform)."""
import glob
import os.path as osp

import numpy as np
from PIL import Image
from torch.utils.data import Dataset


def read_image(path):
    """decoded 'RGB' PIL image; a missing file is an IOError, a failed read is retried like the reference does
    (utils/tools.py:98-119: network file systems hiccup)"""
    if not osp.exists(path):
        raise IOError('"{}" does not exist'.format(path))
    while True:
        try:
            return Image.open(path).convert('RGB')
        except IOError:
            print('IOError incurred when reading "{}". Will redo. Don\'t worry. Just chill.'.format(path))


def _file_name(path):
    # the reference splits on '\\' (RGBNT201.py:51,58), which only works on Windows; basename works everywhere and gives
    # the same name there
    return osp.basename(path.replace('\\', '/'))


class MultiModalImageDataset(Dataset):
    def __init__(self, data, mode='train'):
        self.data = data
        self.mode = mode

    def __len__(self):
        return len(self.data)

    def __getitem__(self, index):
        item = self.data[index]
        img_path, pid, camid = item[0], item[1], item[2]
        timeid = item[3] if len(item) > 3 else 0
        imgs = [np.asarray(read_image(p)) for p in img_path]          # uint8 HxWx3, RGB / NI / TI
        return {'img': imgs, 'pid': pid, 'camid': camid, 'impath': list(img_path), 'timeid': timeid}


def _parse_name(name):
    """<pid6>_cam<X>_...: (identity, zero-based camera)"""
    fields = name.split('_')
    return int(fields[0][:6]), int(fields[1][3]) - 1


class RGBNT201(object):
    """<root>/RGBNT201/train_171 for training, <root>/RGBNT201/test for BOTH query and gallery (the evaluator's
    same-identity-same-camera filter removes the self match): reference data/datasets/image/RGBNT201.py:13-79"""
    dataset_dir = 'RGBNT201'
    splits = (('train', 'train_171', True), ('query', 'test', False), ('gallery', 'test', False))

    def __init__(self, root='', **kwargs):
        self.root = osp.abspath(osp.expanduser(root))
        self.dataset_dir = self.data_dir = osp.join(self.root, self.dataset_dir)
        needed = [self.data_dir]
        for split, folder, _ in self.splits:
            setattr(self, split + '_dir', osp.join(self.data_dir, folder))
            needed.append(getattr(self, split + '_dir'))
        for path in needed:
            if not osp.exists(path):
                raise RuntimeError('"{}" is not found'.format(path))
        for split, _, relabel in self.splits:
            setattr(self, split, self.process_dir(getattr(self, split + '_dir'), relabel=relabel))
        self.num_train_pids = len({rec[1] for rec in self.train})
        self.num_train_cams = len({rec[2] for rec in self.train})

    def process_dir(self, dir_path, relabel=False):
        """one record per RGB file: ([RGB, NI, TI] paths with the same file name, pid, camid, 0).  relabel maps the
        identities to 0..n-1 in the iteration order of a set of ints, exactly as the reference's
        `{pid: label for label, pid in enumerate(pid_container)}` does (RGBNT201.py:49-54)."""
        names = [(_file_name(p), p) for p in glob.glob(osp.join(dir_path, 'RGB', '*.jpg'))]
        label_of = {pid: label for label, pid in enumerate({_parse_name(n)[0] for n, _ in names})}
        records = []
        for name, rgb_path in names:
            pid, cam = _parse_name(name)
            paths = [rgb_path] + [osp.join(dir_path, folder, name) for folder in ('NI', 'TI')]
            records.append((paths, label_of[pid] if relabel else pid, cam, 0))
        return records


class Market1501MM(RGBNT201):
    """The multi-modal version of Market1501 (BASELINE config 5; reference data/datasets/image/market_to_RGBNT201.py:14-78):
    <root>/<dataset_dir>/{train,query,gallery}/{RGB,NI,TI}/<pid>_c<cam>s<seq>_<frame>_<box>.jpg -- Market1501's own file
    names, the same name in the three modality folders.  pid -1 marks junk detections, which are skipped (:54-55, :70-71);
    0 <= pid <= 1501 and cameras 1..6 are asserted like the reference does (:72-73); the training identities are
    relabelled to 0..n-1 (750 of them in the published split).  The reference hard-codes its author's Windows path as
    `dataset_dir` (:15); here it is a folder name under `root`, overridable by keyword."""
    dataset_dir = 'market1501_to_RGBNT201_dark'
    splits = (('train', 'train', True), ('query', 'query', False), ('gallery', 'gallery', False))

    def __init__(self, root='', dataset_dir=None, **kwargs):
        if dataset_dir is not None:
            self.dataset_dir = dataset_dir
        super(Market1501MM, self).__init__(root=root, **kwargs)

    @staticmethod
    def _parse(name):
        fields = name.split('_')
        return int(fields[0]), int(fields[1][1])

    def process_dir(self, dir_path, relabel=False):
        names = [(_file_name(p), p) for p in glob.glob(osp.join(dir_path, 'RGB', '*.jpg'))]
        names = [(n, p) for n, p in names if self._parse(n)[0] != -1]           # junk images are just ignored
        label_of = {pid: label for label, pid in enumerate({self._parse(n)[0] for n, _ in names})}
        records = []
        for name, rgb_path in names:
            pid, cam = self._parse(name)
            assert 0 <= pid <= 1501      # pid == 0 means background
            assert 1 <= cam <= 6
            paths = [rgb_path] + [osp.join(dir_path, folder, name) for folder in ('NI', 'TI')]
            records.append((paths, label_of[pid] if relabel else pid, cam - 1, 0))
        return records


market_to_RGBNT201 = Market1501MM        # the reference's class name
