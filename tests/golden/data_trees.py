"""Tiny VOC / Cityscapes / ACDC trees with the reference's file layouts (deterministic content), shared by the golden generator
(tests/golden/gen_golden.py g8_data: the reference's own data_utils runs over them) and tests/test_data_utils.py (the build's
data_utils must select the same files and return the same items)."""
import os

import numpy as np


def data_trees(root):
    """Tiny VOC / Cityscapes / ACDC trees with the reference's file layouts (deterministic content).  Returns the directory LISTINGS
    the split code consumes in the order this file system returned them (os.walk / os.listdir order is not portable: the tests
    replay the recorded order)."""
    from PIL import Image
    rng = np.random.RandomState(7)
    voc = os.path.join(root, "VOC2012")
    for d in ("JPEGImages", "SegmentationClassAug", "ImageSets/Segmentation"):
        os.makedirs(os.path.join(voc, d))
    ids = ["2008_%06d" % (3 * i + 1) for i in range(41)]
    for k, i in enumerate(ids):
        Image.fromarray(rng.randint(0, 256, (20 + k % 3, 24, 3), dtype=np.uint8)).save(os.path.join(voc, "JPEGImages", i + ".jpg"))
        gt = rng.randint(0, 21, (20 + k % 3, 24)).astype(np.uint8)
        gt[0] = 255
        Image.fromarray(gt).save(os.path.join(voc, "SegmentationClassAug", i + ".png"))
    for name, part in (("trainvalAug.txt", ids[:30]), ("val.txt", ids[30:37]), ("test.txt", ids[37:])):
        with open(os.path.join(voc, "ImageSets/Segmentation", name), "w") as f:
            f.write("\n".join(part) + "\n")
    city = os.path.join(root, "Cityscape")
    k = 0
    for split, towns in (("train", ("aachen", "bochum", "zurich")), ("val", ("lindau",)), ("test", ("berlin",))):
        for town in towns:
            os.makedirs(os.path.join(city, "leftImg8bit", split, town))
            if split != "test":
                os.makedirs(os.path.join(city, "gtFine", "trainval", town), exist_ok=True)
            for j in range(7 if split == "train" else 4):
                stem = "%s_%06d_%06d" % (town, j, 19 * j)
                Image.fromarray(rng.randint(0, 256, (16, 32, 3), dtype=np.uint8)).save(os.path.join(city, "leftImg8bit", split, town, stem + "_leftImg8bit.png"))
                if split != "test":
                    lab = ((np.arange(16 * 32).reshape(16, 32) + k) % 34).astype(np.uint8)
                    Image.fromarray(lab).save(os.path.join(city, "gtFine", "trainval", town, stem + "_gtFine_labelIds.png"))
                k += 1
    acdc = os.path.join(root, "ACDC")
    for d in ("training", "training_gt", "testing"):
        os.makedirs(os.path.join(acdc, d))
    for i in range(27):
        # ('pjg' characters at the end of a stem: rstrip('.jpg') strips characters, not a suffix)
        stem = ("patient%03d_frame%02d" % (i, i % 5)) + ("pg" if i % 6 == 0 else "")
        Image.fromarray(rng.randint(0, 256, (18, 22), dtype=np.uint8)).save(os.path.join(acdc, "training", stem + ".jpg"))
        Image.fromarray(rng.randint(0, 4, (18, 22)).astype(np.uint8)).save(os.path.join(acdc, "training_gt", stem.rstrip('.jpg') + ".png"))
    for i in range(3):
        Image.fromarray(rng.randint(0, 256, (18, 22), dtype=np.uint8)).save(os.path.join(acdc, "testing", "t%02d.jpg" % i))
    return {"voc2012": voc, "cityscapes": city, "acdc": acdc}
