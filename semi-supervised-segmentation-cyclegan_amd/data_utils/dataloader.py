"""VOC2012 / Cityscapes / ACDC datasets with the file layouts, split logic and return tuples of the reference
(data_utils/dataloader.py:15-393).  One base class holds what the three reference classes repeat: the seeded
labeled/unlabeled split (`np.random.seed(1)`, `np.random.choice(..., replace=False)`), the equalisation of the two
lists by repetition, and the (img, gt, name) / (img, name) item protocol.  Quirks kept on purpose, because they
decide WHICH files a run sees: `pd.read_table` without `header=None` swallows the first line of every VOC list;
`str.rstrip('.jpg')` / `.rstrip('.png')` strip characters, not a suffix; the Cityscapes sample name is cut at
fixed character offsets (34 / 38) of the path."""
import os
import re

import numpy as np
from PIL import Image
from torch.utils.data import Dataset


def recursive_glob(rootdir=".", suffix=""):
    """utils.py:302-312."""
    return [os.path.join(looproot, filename) for looproot, _, filenames in os.walk(rootdir) for filename in filenames
            if filename.endswith(suffix)]


def split_labeled_unlabeled(train_imgs, ratio):
    """The split every reference dataset performs right after `np.random.seed(1)` (dataloader.py:45-66,158-178,311-331);
    consumes the global numpy RNG exactly as the reference does."""
    labeled = list(np.random.choice(train_imgs, size=int(ratio * len(train_imgs)), replace=False))
    unlabeled = [x for x in train_imgs if x not in labeled]
    if ratio > 0.5:        # repeat the shorter list until both have the same length
        excess = round((ratio / (1 - ratio + 1e-6)), 1) - 1
        extra = list(np.random.choice(np.array(unlabeled), size=int((excess - int(excess)) * len(unlabeled)), replace=False))
        unlabeled += unlabeled * int(excess) + extra
    elif ratio < 0.5:
        excess = round(((1 - ratio) / (ratio + 1e-6)), 1) - 1
        extra = list(np.random.choice(np.array(labeled), size=int((excess - int(excess)) * len(labeled)), replace=False))
        labeled += labeled * int(excess) + extra
    return labeled, unlabeled


class _SegDataset(Dataset):
    SPLITS = ('label', 'unlabel', 'val', 'test')

    def __init__(self, root_path, name, ratio, transformation, augmentation):
        super().__init__()
        assert transformation is not None, 'transformation must be provided, give None'
        assert name in self.SPLITS, 'dataset name should be restricted in "label", "unlabel", "test" and "val", given %s' % name
        assert 0 <= ratio <= 1, 'the ratio between "labeled" and "unlabeled" should be between 0 and 1, given %.1f' % ratio
        self.root_path = self.root = root_path
        self.name, self.ratio = name, ratio
        self.transformation, self.augmentation = transformation, augmentation
        self.device_finish = bool(transformation.get('device_finish')) if isinstance(transformation, dict) else False
        np.random.seed(1)      # labeled and unlabeled instances draw the same split, so they never share an image
        self.items = self._select()
        if len(self.items) == 0:
            raise Exception("No files for name=[%s] found in %s" % (name, root_path))

    # -- per dataset
    def _select(self):
        raise NotImplementedError

    def _paths(self, item):
        """(image path, label path or None, sample name)"""
        raise NotImplementedError

    def _open_image(self, path):
        return Image.open(path).convert('RGB')

    def _finish_label(self, gt):
        return gt

    # -- shared
    def __len__(self):
        return len(self.items)

    def __getitem__(self, index):
        img_path, gt_path, sample = self._paths(self.items[index])
        img = self._open_image(img_path)
        if self.name == 'test':
            if self.augmentation is not None:
                img = self.augmentation(img)
            return self.transformation['img'](img), sample
        gt = Image.open(gt_path)
        if self.augmentation is not None:
            img, gt = self.augmentation(img, gt)
        img, gt = self.transformation['img'](img), self.transformation['gt'](gt)
        if not self.device_finish:        # on the device path the label table of DeviceLoader does this
            gt = self._finish_label(gt)
        return img, gt, sample


class VOCDataset(_SegDataset):
    """JPEGImages/<id>.jpg, SegmentationClassAug/<id>.png, id lists in ImageSets/Segmentation/{trainvalAug,val,test}.txt
    (dataloader.py:15-117).  Labels 0..20, 255 = boundary (relabelled to 0 by the transform)."""

    def __init__(self, root_path, name='label', ratio=0.5, transformation=None, augmentation=None):
        super().__init__(root_path, name, ratio, transformation, augmentation)
        self.imgs = self.gts = self.items

    def _list(self, fname):
        import pandas as pd
        return pd.read_table(os.path.join(self.root_path, 'ImageSets/Segmentation', fname)).values.reshape(-1)

    def _select(self):
        if self.name == 'test':
            return self._list('test.txt')
        train, val = self._list('trainvalAug.txt'), self._list('val.txt')
        labeled, unlabeled = split_labeled_unlabeled(train, self.ratio)
        return {'label': labeled, 'unlabel': unlabeled, 'val': val}[self.name]

    def _paths(self, item):
        return (os.path.join(self.root_path, 'JPEGImages', item + '.jpg'),
                os.path.join(self.root_path, 'SegmentationClassAug', item + '.png'), item)


class CityscapesDataset(_SegDataset):
    """leftImg8bit/{train,val}/<city>/*_leftImg8bit.png with gtFine/trainval/<city>/*_gtFine_labelIds.png, test images in
    leftImg8bit/test (dataloader.py:120-267).  34 label ids -> 19 classes + 19 = unlabelled."""

    n_classes = 20
    ignore_index = 250

    def __init__(self, root_path, name="train", ratio=0.5, transformation=False, augmentation=None):
        self.images_base = os.path.join(root_path, "leftImg8bit") if name != 'test' else os.path.join(root_path, "leftImg8bit", 'test')
        self.annotations_base = os.path.join(root_path, "gtFine", 'trainval')
        super().__init__(root_path, name, ratio, transformation, augmentation)
        self.files = {name: list(self.items)}
        print("Found %d %s images" % (len(self.items), name))

    def _select(self):
        if self.name == 'test':
            return recursive_glob(rootdir=self.images_base, suffix=".png")
        train = np.array(recursive_glob(rootdir=os.path.join(self.images_base, 'train'), suffix=".png"))
        val = recursive_glob(rootdir=os.path.join(self.images_base, 'val'), suffix=".png")
        labeled, unlabeled = split_labeled_unlabeled(train, self.ratio)
        return list({'label': labeled, 'unlabel': unlabeled, 'val': val}[self.name])

    def _paths(self, item):
        img_path = item.rstrip()
        cut = 34 if self.name == 'test' else 38
        sample = re.sub(r'.*/', '', img_path[cut:]).rstrip('.png')
        if self.name == 'test':
            return img_path, None, sample
        lbl = os.path.join(self.annotations_base, img_path.split(os.sep)[-2], os.path.basename(img_path)[:-15] + "gtFine_labelIds.png")
        return img_path, lbl, sample

    def _finish_label(self, gt):
        from . import cityscapes_encode
        return cityscapes_encode(gt)

    encode_segmap = _finish_label


class ACDCDataset(_SegDataset):
    """training/<id>.jpg with training_gt/<id>.png, testing/<id>.jpg; 85 % / 15 % train / val split of `training`
    (dataloader.py:270-393).  Single-channel images (opened as stored, no RGB conversion), labels 0..3."""

    split_ratio = [0.85, 0.15]

    def __init__(self, root_path, name='label', ratio=0.5, transformation=None, augmentation=None):
        self.images_base = os.path.join(root_path, 'training' if name != 'test' else 'testing')
        self.annotations_base = os.path.join(root_path, 'training_gt')
        super().__init__(root_path, name, ratio, transformation, augmentation)
        self.files = {name: list(self.items)}
        print("Found %d %s images" % (len(self.items), name))

    def _select(self):
        if self.name == 'test':
            return os.listdir(self.images_base)
        total = np.array(os.listdir(self.images_base))
        train = np.random.choice(total, size=int(self.split_ratio[0] * len(total)), replace=False)
        val = [x for x in total if x not in train]
        labeled, unlabeled = split_labeled_unlabeled(train, self.ratio)
        return list({'label': labeled, 'unlabel': unlabeled, 'val': val}[self.name])

    def _open_image(self, path):
        return Image.open(path)

    def _paths(self, item):
        sample = item.rstrip('.jpg')
        if self.name == 'test':
            return os.path.join(self.images_base, item), None, sample
        return os.path.join(self.images_base, item), os.path.join(self.annotations_base, sample + '.png'), sample
