"""`python main.py --validation ...`: the reference's validation.py (:20-157) on the MI355X backend.
Loads `latest_{supervised_model,semisuper_cycleGAN}.ckpt`, runs the DeepLab generators in eval mode over the val
split and writes the paletted predictions (and, for the semi-supervised model, the regenerated labels / images)
under `--validation_dir`, with the reference's directory names."""
import importlib
import os

import torch

PKG = "semi-supervised-segmentation-cyclegan_amd"


def _mk(*parts):
    d = os.path.join(*parts)
    os.makedirs(d, exist_ok=True)
    return d


def validation(args, val_loader=None):
    F = importlib.import_module(PKG + ".functional")
    arch = importlib.import_module(PKG + ".arch")
    utils = importlib.import_module(PKG + ".utils")
    n_channels = {'voc2012': 21, 'cityscapes': 20, 'acdc': 4}[args.dataset]
    dev = torch.device("cuda", args.gpu_ids[0])
    if val_loader is None:
        du = importlib.import_module(PKG + ".data_utils")
        from torch.utils.data import DataLoader
        tr = du.get_transformation((args.crop_height, args.crop_width), resize=True, dataset=args.dataset, device_finish=True)
        cls = {'voc2012': du.VOCDataset, 'cityscapes': du.CityscapesDataset, 'acdc': du.ACDCDataset}[args.dataset]
        root = {'voc2012': './data/VOC2012', 'cityscapes': './data/Cityscape', 'acdc': './data/ACDC'}[args.dataset]
        val_set = cls(root_path=root, name='val', ratio=0.5, transformation=tr, augmentation=None)
        val_loader = du.DeviceLoader(DataLoader(val_set, batch_size=args.batch_size, shuffle=False), tr, dev)

    mk = lambda i, o: arch.define_Gen(input_nc=i, output_nc=o, ngf=args.ngf, netG='deeplab', norm=args.norm,
                                      use_dropout=not args.no_dropout, gpu_ids=args.gpu_ids)
    Gsi, Gis = mk(3, n_channels), mk(n_channels, 3)            # validation.py:42-46
    size = (args.crop_height, args.crop_width)
    best_iou = 0
    semi = args.model == 'semisupervised_cycleGAN'
    try:
        ckpt = utils.load_checkpoint('%s/latest_%s.ckpt' % (args.checkpoint_dir, 'semisuper_cycleGAN' if semi else 'supervised_model'))
        Gsi.load_state_dict(ckpt['Gsi'])
        if semi:
            Gis.load_state_dict(ckpt['Gis'])
        best_iou = ckpt['best_iou']
    except Exception:
        print(' [*] No checkpoint!')

    seg = lambda x: F.softmax2d(F.upsample_bilinear(Gsi(x), size))                 # Gsi -> interp -> Softmax2d
    img = lambda x: F.act_fwd(F.to_nhwc(F.upsample_bilinear(Gis(x), size)), F.ACT_TANH)   # Gis -> interp -> Tanh
    Gsi.eval()
    with torch.no_grad():
        for i, (image_test, real_segmentation, image_name) in enumerate(val_loader):
            image_test, real_segmentation = utils.cuda([image_test, real_segmentation], args.gpu_ids)
            seg_map = seg(image_test)
            prediction = F.argmax_index(seg_map).cpu().numpy()
            if not semi:
                out = _mk(args.validation_dir, 'supervised')
                for j in range(prediction.shape[0]):
                    utils.colorize_mask(prediction[j], args.dataset).save(os.path.join(out, image_name[j] + '.png'))
            else:
                fake_img = img(seg_map)                                                                   # :108-110
                fake_img_from_labels = img(utils.make_one_hot(real_segmentation, args.dataset, args.gpu_ids))   # :112-114
                regenerated = F.argmax_index(seg(fake_img_from_labels)).cpu().numpy()                     # :115-120
                fake_img = F.to_nchw(fake_img).cpu() * 0.5 + 0.5                                          # undo Normalize(.5, .5)
                fake_img_from_labels = F.to_nchw(fake_img_from_labels).cpu() * 0.5 + 0.5
                base = os.path.join(args.validation_dir, 'unsupervised')
                for j in range(prediction.shape[0]):
                    utils.colorize_mask(prediction[j], args.dataset).save(os.path.join(_mk(base, 'generated_labels'), image_name[j] + '.png'))
                    utils.colorize_mask(regenerated[j], args.dataset).save(os.path.join(_mk(base, 'regenerated_labels'), image_name[j] + '.png'))
                    utils.save_image(fake_img[j], os.path.join(_mk(base, 'regenerated_image'), image_name[j] + '.jpg'))
                    utils.save_image(fake_img_from_labels[j], os.path.join(_mk(base, 'image_from_labels'), image_name[j] + '.jpg'))
            print('Epoch-', str(i + 1), ' Done!')
    print('The iou of the resulting segment maps: ', str(best_iou))
    return best_iou
