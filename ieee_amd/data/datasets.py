"""RGBNT201 directory parser and the 3-modal dataset (reference data/datasets/image/RGBNT201.py:13-79,
data/datasets/dataset.py:320-351, utils/tools.py:98-119).  On-disk format: <root>/RGBNT201/{train_171,test}/{RGB,NI,TI}/
<pid6>_cam<X>_....jpg, the same file name in the three modality folders.  __getitem__ returns the DECODED uint8 images
(PIL 'RGB'); resize / flip / normalisation happen on the device for the whole batch (transforms.DeviceTransform)."""
import glob
import os.path as osp
import warnings

import numpy as np
from PIL import Image
from torch.utils.data import Dataset


def read_image(path):
    """tools.py:98-119: PIL open + convert('RGB'), retrying on IOError"""
    got_img = False
    if not osp.exists(path):
        raise IOError('"{}" does not exist'.format(path))
    while not got_img:
        try:
            img = Image.open(path).convert('RGB')
            got_img = True
        except IOError:
            print('IOError incurred when reading "{}". Will redo. Don\'t worry. Just chill.'.format(path))
    return img


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


class RGBNT201(object):
    dataset_dir = 'RGBNT201'

    def __init__(self, root='', **kwargs):
        self.root = osp.abspath(osp.expanduser(root))
        self.dataset_dir = osp.join(self.root, self.dataset_dir)
        self.data_dir = self.dataset_dir
        if not osp.isdir(self.data_dir):
            warnings.warn('The current data structure is deprecated.')
        self.train_dir = osp.join(self.data_dir, 'train_171')
        self.query_dir = osp.join(self.data_dir, 'test')
        self.gallery_dir = osp.join(self.data_dir, 'test')
        for f in (self.data_dir, self.train_dir, self.query_dir, self.gallery_dir):
            if not osp.exists(f):
                raise RuntimeError('"{}" is not found'.format(f))
        self.train = self.process_dir(self.train_dir, relabel=True)
        self.query = self.process_dir(self.query_dir, relabel=False)
        self.gallery = self.process_dir(self.gallery_dir, relabel=False)
        self.num_train_pids = len(set(d[1] for d in self.train))
        self.num_train_cams = len(set(d[2] for d in self.train))

    def process_dir(self, dir_path, relabel=False):
        """RGBNT201.py:47-79 (pid = first 6 characters, camid = 4th character of the second '_' field, minus 1)"""
        img_paths_RGB = glob.glob(osp.join(dir_path, 'RGB', '*.jpg'))
        pid_container = set()
        for p in img_paths_RGB:
            pid_container.add(int(_file_name(p).split('_')[0][0:6]))
        pid2label = {pid: label for label, pid in enumerate(pid_container)}
        data = []
        for p in img_paths_RGB:
            name = _file_name(p)
            img = [p, osp.join(dir_path, 'NI', name), osp.join(dir_path, 'TI', name)]
            pid = int(name.split('_')[0][0:6])
            camid = int(name.split('_')[1][3]) - 1
            if relabel:
                pid = pid2label[pid]
            data.append((img, pid, camid, 0))
        return data
