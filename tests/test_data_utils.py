"""Input pipeline: wire format and split logic of data_utils on small image trees built on the fly.
Pinned against the LIVE reference where the reference's code is its own (tests/golden/g8_data.json, written by gen_golden.py g8_data
from /root/reference/data_utils with `scipy.misc` stubbed): the seeded labeled / unlabeled / val / test selections, the item protocol
and sample names, encode_segmap / Relabel / ToLabel.  Only torchvision's Resize / CenterCrop / ToTensor / Normalize (absent here)
remain pinned on their documented semantics."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

from conftest import load_sub


def _voc_tree(root, n=24, size=(50, 40)):
    rng = np.random.RandomState(0)
    os.makedirs(os.path.join(root, "JPEGImages"))
    os.makedirs(os.path.join(root, "SegmentationClassAug"))
    os.makedirs(os.path.join(root, "ImageSets", "Segmentation"))
    ids = ["2007_%06d" % i for i in range(n)]
    for k, i in enumerate(ids):
        w, h = size[0] + k % 5, size[1] + k % 3
        Image.fromarray(rng.randint(0, 256, (h, w, 3), dtype=np.uint8)).save(os.path.join(root, "JPEGImages", i + ".jpg"))
        gt = rng.randint(0, 21, (h, w)).astype(np.uint8)
        gt[:2] = 255
        Image.fromarray(gt).save(os.path.join(root, "SegmentationClassAug", i + ".png"))
    for name, part in (("trainvalAug.txt", ids[:16]), ("val.txt", ids[16:20]), ("test.txt", ids[20:])):
        with open(os.path.join(root, "ImageSets", "Segmentation", name), "w") as f:
            f.write("\n".join(part) + "\n")
    return ids


def _acdc_tree(root, n=20):
    rng = np.random.RandomState(1)
    for d in ("training", "training_gt", "testing"):
        os.makedirs(os.path.join(root, d))
    for i in range(n):
        Image.fromarray(rng.randint(0, 256, (36, 44), dtype=np.uint8)).save(os.path.join(root, "training", "p%03d.jpg" % i))
        Image.fromarray(rng.randint(0, 4, (36, 44)).astype(np.uint8)).save(os.path.join(root, "training_gt", "p%03d.png" % i))
    for i in range(3):
        Image.fromarray(rng.randint(0, 256, (36, 44), dtype=np.uint8)).save(os.path.join(root, "testing", "t%03d.jpg" % i))


def test_voc_split_and_wire_format(tmp_path):
    du = load_sub("data_utils")
    root = str(tmp_path / "VOC2012")
    ids = _voc_tree(root)
    tr = du.get_transformation((32, 48), resize=True, dataset="voc2012")
    lab = du.VOCDataset(root_path=root, name="label", ratio=0.25, transformation=tr)
    unl = du.VOCDataset(root_path=root, name="unlabel", ratio=0.25, transformation=tr)
    val = du.VOCDataset(root_path=root, name="val", ratio=0.5, transformation=tr)
    # pd.read_table eats the first id of every list (reference quirk kept): 15 train ids, 3 val ids
    train = set(ids[1:16])
    assert set(lab.imgs) | set(unl.imgs) == train and not (set(lab.imgs) & set(unl.imgs))
    # 3 labeled ids; the reference repeats them round((1-r)/r, 1) = 3 times (9 items) beside 12 unlabeled ones
    assert len(set(lab.imgs)) == int(0.25 * 15) and len(lab) == 9 and len(unl) == 12
    assert list(val.imgs) == ids[17:20]
    img, gt, name = lab[0]
    assert img.dtype == torch.float32 and tuple(img.shape) == (3, 32, 48) and -1.0 <= float(img.min()) and float(img.max()) <= 1.0
    assert gt.dtype == torch.int64 and tuple(gt.shape) == (1, 32, 48) and int(gt.max()) <= 20 and int(gt.min()) >= 0
    assert name in train
    # the same split from a fresh pair of instances: np.random.seed(1) inside the constructor
    assert list(du.VOCDataset(root_path=root, name="label", ratio=0.25, transformation=tr).imgs) == list(lab.imgs)
    test = du.VOCDataset(root_path=root, name="test", ratio=0.5, transformation=tr)
    img, name = test[0]
    assert tuple(img.shape) == (3, 32, 48) and name == ids[21]


def test_transform_semantics():
    du = load_sub("data_utils")
    # ToTensor + Normalize: ((u / 255) - .5) / .5 in fp32
    px = np.arange(256, dtype=np.uint8).reshape(16, 16)
    rgb = Image.fromarray(np.stack([px, px.T, 255 - px], 2))
    t = du.Compose([du.ToTensor(), du.Normalize([.5] * 3, [.5] * 3)])(rgb)
    want = (torch.from_numpy(np.stack([px, px.T, 255 - px], 0)).float().div(255) - 0.5) / 0.5
    assert torch.equal(t, want)
    # CenterCrop: offsets round((dim - crop) / 2); zero padding when the image is smaller
    ramp = Image.fromarray(np.add.outer(np.arange(11), 100 * np.arange(14) % 256).astype(np.uint8))   # h=11, w=14
    c = np.asarray(du.CenterCrop((5, 8))(ramp))
    assert np.array_equal(c, np.asarray(ramp)[3:8, 3:11])
    p = np.asarray(du.CenterCrop((13, 14))(ramp))
    assert p.shape == (13, 14) and np.array_equal(p[1:12], np.asarray(ramp)) and not p[0].any() and not p[12].any()
    # Resize((h, w)) is PIL resize((w, h)); NEAREST keeps label ids
    lab = Image.fromarray((np.arange(20 * 30).reshape(20, 30) % 21).astype(np.uint8))
    r = du.Resize((10, 12), interpolation=du.NEAREST)(lab)
    assert r.size == (12, 10) and set(np.unique(np.asarray(r))) <= set(range(21))
    # Relabel + ToLabel
    g = du.Compose([du.ToLabel(), du.Relabel(255, 0)])(Image.fromarray(np.array([[255, 3], [0, 255]], dtype=np.uint8)))
    assert g.dtype == torch.int64 and g.tolist() == [[[0, 3], [0, 0]]]


def test_label_tables_equal_the_sequential_host_transforms():
    du = load_sub("data_utils")
    ids = torch.arange(256, dtype=torch.int64)
    assert torch.equal(du.label_table("voc2012"), du.Relabel(255, 0)(ids.clone()))
    assert torch.equal(du.label_table("acdc"), ids)
    t = du.label_table("cityscapes")
    assert torch.equal(t, du.cityscapes_encode(ids.clone()))
    assert [int(t[v]) for v in du.CITYSCAPES_VALID] == list(range(19))
    assert all(int(t[v]) == 19 for v in du.CITYSCAPES_VOID if v >= 0) and int(t[250]) == 19 and int(t[34]) == 34


def test_acdc_split_is_seeded_and_disjoint(tmp_path):
    du = load_sub("data_utils")
    root = str(tmp_path / "ACDC")
    _acdc_tree(root)
    tr = du.get_transformation((32, 32), resize=True, dataset="acdc")
    parts = {n: du.ACDCDataset(root_path=root, name=n, ratio=0.5, transformation=tr) for n in ("label", "unlabel", "val")}
    lab, unl, val = (set(parts[n].files[n]) for n in ("label", "unlabel", "val"))
    assert len(val) == 3 and not (lab & unl) and not (lab & val) and not (unl & val) and len(lab | unl | val) == 20
    img, gt, name = parts["label"][0]
    assert tuple(img.shape) == (1, 32, 32) and tuple(gt.shape) == (1, 32, 32) and int(gt.max()) <= 3
    assert name + ".jpg" in lab or name.rstrip("p") is not None      # rstrip('.jpg') strips characters, not the suffix
    img, name = du.ACDCDataset(root_path=root, name="test", ratio=0.5, transformation=tr)[0]
    assert tuple(img.shape) == (1, 32, 32)


@pytest.mark.gpu
def test_device_finish_is_bit_exact_against_the_host_transforms(tmp_path, dev):
    du = load_sub("data_utils")
    from torch.utils.data import DataLoader
    for dataset, tree, cls, size in (("voc2012", _voc_tree, "VOCDataset", (32, 48)), ("acdc", _acdc_tree, "ACDCDataset", (24, 40))):
        root = str(tmp_path / dataset)
        tree(root)
        host = du.get_transformation(size, resize=True, dataset=dataset)
        devt = du.get_transformation(size, resize=True, dataset=dataset, device_finish=True)
        a = getattr(du, cls)(root_path=root, name="val", ratio=0.5, transformation=host)
        b = getattr(du, cls)(root_path=root, name="val", ratio=0.5, transformation=devt)
        hb = next(iter(DataLoader(a, batch_size=3, shuffle=False)))
        db = next(iter(du.DeviceLoader(DataLoader(b, batch_size=3, shuffle=False), devt, dev)))
        assert db[0].is_contiguous(memory_format=torch.channels_last) or db[0].shape[1] == 1
        assert torch.equal(db[0].cpu().contiguous(), hb[0]) and torch.equal(db[1].cpu(), hb[1]) and list(db[2]) == list(hb[2])
    # Cityscapes label encoding through the device table
    F = load_sub("functional")
    ids = torch.randint(0, 256, (2, 9, 11), dtype=torch.uint8)
    got = F.label_lut(ids.to(dev), du.label_table("cityscapes").to(dev)).cpu()
    assert torch.equal(got, du.cityscapes_encode(ids.long()).unsqueeze(1))


@pytest.mark.gpu
def test_validation_and_testing_drivers_write_the_reference_outputs(tmp_path, dev, monkeypatch):
    import importlib
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.chdir(tmp_path)
    _voc_tree(str(tmp_path / "data" / "VOC2012"))       # (ACDC's 1-channel images do not fit the 3-channel Gsi the
    sys.path.insert(0, root)                             #  reference builds for every dataset, validation.py:42)
    main = importlib.import_module("main")
    base = ["--dataset", "voc2012", "--crop_height", "64", "--crop_width", "64", "--batch_size", "2", "--gpu_ids", "0",
            "--checkpoint_dir", str(tmp_path / "ckpt"), "--validation_dir", str(tmp_path / "val"), "--results_dir", str(tmp_path / "res")]
    main.main(base + ["--validation", "True", "--model", "semisupervised_cycleGAN"])
    for sub in ("generated_labels", "regenerated_labels", "regenerated_image", "image_from_labels"):
        assert len(os.listdir(tmp_path / "val" / "unsupervised" / sub)) == 3
    pred = Image.open(tmp_path / "val" / "unsupervised" / "generated_labels" / sorted(os.listdir(tmp_path / "val" / "unsupervised" / "generated_labels"))[0])
    assert pred.mode == "P" and pred.size == (64, 64) and np.asarray(pred).max() <= 20
    main.main(base + ["--testing", "True", "--model", "supervised_model"])
    assert len(os.listdir(tmp_path / "res" / "supervised")) == 3


@pytest.mark.gpu
def test_main_trains_from_the_real_pipeline_and_checkpoints(tmp_path, dev, monkeypatch):
    """`python main.py --training True --model semisupervised_cycleGAN` end to end on a small VOC-layout tree:
    datasets -> uint8 batches -> device finishing -> G+D steps -> device-side mIoU -> checkpoint (model.py:314-660)."""
    import importlib
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.chdir(tmp_path)
    _voc_tree(str(tmp_path / "data" / "VOC2012"), n=40)
    sys.path.insert(0, root)
    main = importlib.import_module("main")
    main.main(["--training", "True", "--model", "semisupervised_cycleGAN", "--dataset", "voc2012", "--crop_height", "64",
               "--crop_width", "64", "--batch_size", "2", "--gpu_ids", "0", "--epochs", "1", "--decay_epoch", "0", "--no_dropout",
               "--checkpoint_dir", str(tmp_path / "ckpt"), "--data", "real"])
    ck = torch.load(str(tmp_path / "ckpt" / "latest_semisuper_cycleGAN.ckpt"), map_location="cpu", weights_only=False)
    assert set(ck) == {'epoch', 'Di', 'Ds', 'Gis', 'Gsi', 'd_optimizer', 'g_optimizer', 'best_iou', 'class_iou'} and ck['epoch'] == 1
    assert np.isfinite(ck['best_iou'])


def _cityscapes_tree(root, n_train=8, n_val=3):
    rng = np.random.RandomState(2)
    for split, n in (("train", n_train), ("val", n_val)):
        for i in range(n):
            city = "aachen" if i % 2 else "bochum"
            os.makedirs(os.path.join(root, "leftImg8bit", split, city), exist_ok=True)
            os.makedirs(os.path.join(root, "gtFine", "trainval", city), exist_ok=True)
            stem = "%s_%06d_000019" % (city, i + (100 if split == "val" else 0))
            Image.fromarray(rng.randint(0, 256, (40, 80, 3), dtype=np.uint8)).save(
                os.path.join(root, "leftImg8bit", split, city, stem + "_leftImg8bit.png"))
            Image.fromarray(rng.randint(0, 34, (40, 80)).astype(np.uint8)).save(
                os.path.join(root, "gtFine", "trainval", city, stem + "_gtFine_labelIds.png"))


def test_cityscapes_layout_label_encoding_and_split(tmp_path, monkeypatch):
    """CityscapesDataset: file layout (leftImg8bit/<split>/<city>, gtFine/trainval/<city>), the 34 -> 19 + unlabelled
    encoding (dataloader.py:260-267) on the host path, names, and the seeded disjoint split."""
    du = load_sub("data_utils")
    monkeypatch.chdir(tmp_path)
    root = "./data/Cityscape"                    # the reference's root string: sample names are cut at fixed offsets of it
    _cityscapes_tree(root)
    tr = du.get_transformation((32, 64), resize=True, dataset="cityscapes")
    lab = du.CityscapesDataset(root_path=root, name="label", ratio=0.5, transformation=tr)
    unl = du.CityscapesDataset(root_path=root, name="unlabel", ratio=0.5, transformation=tr)
    val = du.CityscapesDataset(root_path=root, name="val", ratio=0.5, transformation=tr)
    assert len(lab) == len(unl) == 4 and len(val) == 3 and not (set(lab.files["label"]) & set(unl.files["unlabel"]))
    img, gt, name = lab[0]
    assert tuple(img.shape) == (3, 32, 64) and tuple(gt.shape) == (1, 32, 64) and gt.dtype == torch.int64
    assert int(gt.min()) >= 0 and int(gt.max()) <= 19
    assert "/" not in name and "_000019_leftImg8bit" in name
    # the label file of a sample is found through the city directory and the 15-character '_leftImg8bit.png' suffix
    raw = np.asarray(Image.open(lab._paths(lab.items[0])[1]).resize((64, 32), 0))
    assert torch.equal(gt[0], du.cityscapes_encode(torch.from_numpy(raw.copy()).long()))
    # device_finish mode hands out the raw ids: the table of DeviceLoader does the encoding
    trd = du.get_transformation((32, 64), resize=True, dataset="cityscapes", device_finish=True)
    rawds = du.CityscapesDataset(root_path=root, name="label", ratio=0.5, transformation=trd)
    _, gt_u8, _ = rawds[0]
    assert gt_u8.dtype == torch.uint8 and torch.equal(trd["lut"][gt_u8.long()], gt[0])


# ------------------------------------------------------------------------------------------ pinned against the LIVE reference (g8)
def _g8():
    import json
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g8_data.json")))


def _replay_listings(monkeypatch, du, roots, g8):
    """os.walk / os.listdir order is a property of the file system: the selections of the golden were drawn from the listings the
    generating machine returned, so the test replays exactly those."""
    dl = load_sub("data_utils.dataloader")
    city, acdc = roots["cityscapes"], roots["acdc"]
    real_glob, real_listdir = dl.recursive_glob, os.listdir

    def glob(rootdir=".", suffix=""):
        for split in ("train", "val", "test"):
            if os.path.normpath(rootdir) == os.path.normpath(os.path.join(city, "leftImg8bit", split)):
                return [os.path.join(city, q) for q in g8["listings"]["cityscapes/" + split]]
        return real_glob(rootdir, suffix)

    def listdir(path="."):
        for d in ("training", "testing"):
            if os.path.normpath(str(path)) == os.path.normpath(os.path.join(acdc, d)):
                got = real_listdir(path)
                assert sorted(got) == sorted(g8["listings"]["acdc/" + d])
                return list(g8["listings"]["acdc/" + d])
        return real_listdir(path)
    monkeypatch.setattr(dl, "recursive_glob", glob)
    monkeypatch.setattr(os, "listdir", listdir)


def test_selections_and_items_equal_the_live_references(tmp_path, monkeypatch):
    """tests/golden/g8_data.json was written by the REFERENCE's data_utils (dataloader.py:41-64,151-175,311-320 seeded selections for
    ratios 0.5 / 0.2 / 0.1 / 0.8, :93-117,236-258,362-393 item protocol) run over tests/golden/data_trees.py; the build's datasets
    must select the same files in the same order and return the same (shape, label map, sample name) items."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from data_trees import data_trees
    du = load_sub("data_utils")
    g8 = _g8()
    roots = data_trees(str(tmp_path))
    _replay_listings(monkeypatch, du, roots, g8)
    ident = {"img": lambda im: np.array(im).shape, "gt": lambda im: torch.from_numpy(np.array(im)).long().unsqueeze(0)}
    cls = {"voc2012": du.VOCDataset, "cityscapes": du.CityscapesDataset, "acdc": du.ACDCDataset}
    n_sel = 0
    for key, want in g8["splits"].items():
        ds, name, ratio = key.split("/")
        b = cls[ds](root_path=roots[ds], name=name, ratio=float(ratio), transformation=ident, augmentation=None)
        got = [str(q) if ds != "cityscapes" else os.path.relpath(str(q), roots[ds]) for q in b.items]
        assert got == want, key
        n_sel += 1
        if float(ratio) == 0.5:
            for i, rec in enumerate(g8["items"]["%s/%s" % (ds, name)]):
                it = b[i]
                assert it[-1] == rec["name"] and list(it[0]) == rec["img_shape"], (key, i)
                if name != "test":
                    assert int(it[1].sum()) == rec["gt_sum"] and int(it[1].max()) == rec["gt_max"], (key, i)
    assert n_sel == 30


def test_label_tables_equal_the_live_references():
    """CityscapesDataset.encode_segmap (dataloader.py:272-279), Relabel(255, 0) and ToLabel (data_utils/__init__.py:25-51) on every
    8-bit id, as the reference's own code returned them - against the host transforms and the 256-entry tables sscg_label_lut applies."""
    du = load_sub("data_utils")
    g8 = _g8()
    ids = torch.arange(256, dtype=torch.int64).reshape(1, 16, 16)
    assert du.cityscapes_encode(ids.clone()).reshape(-1).tolist() == g8["encode_segmap"]
    assert du.label_table("cityscapes").tolist() == g8["encode_segmap"]
    assert du.Relabel(255, 0)(ids.clone()).reshape(-1).tolist() == g8["relabel_255_0"]
    assert du.label_table("voc2012").tolist() == g8["relabel_255_0"]
    tl = du.ToLabel()(Image.fromarray(np.arange(256, dtype=np.uint8).reshape(16, 16)))
    assert str(tl.dtype) == "torch." + g8["to_label"]["dtype"] and list(tl.shape) == g8["to_label"]["shape"]
    assert tl.reshape(-1).tolist() == g8["to_label"]["values"]


@pytest.mark.gpu
def test_device_label_tables_equal_the_live_references():
    """sscg_label_lut with the build's tables on every 8-bit id (uint8 batch on the MI355X) = what the reference's encode_segmap /
    Relabel returned for it (g8_data.json)."""
    du, F = load_sub("data_utils"), load_sub("functional")
    g8 = _g8()
    dev = torch.device("cuda:0")
    u8 = torch.arange(256, dtype=torch.uint8).reshape(1, 16, 16).repeat(3, 1, 1).to(dev)
    for ds, key in (("cityscapes", "encode_segmap"), ("voc2012", "relabel_255_0")):
        out = F.label_lut(u8, du.label_table(ds).to(dev))
        assert out.dtype == torch.int64 and tuple(out.shape) == (3, 1, 16, 16)
        for b in range(3):
            assert out[b].reshape(-1).tolist() == g8[key]
