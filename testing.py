"""`python main.py --testing ...`: the reference's testing.py (:17-100) on the MI355X backend: paletted
predictions of Gsi over the test split under `--results_dir/{supervised,unsupervised}`.
As written there, Gsi is built as `resnet_9blocks_softmax` (testing.py:40) although training saves a DeepLab
`Gsi`: loading then fails, ' [*] No checkpoint!' is printed and the run continues on the initial weights.  The
behaviour is kept; `--testing_gen deeplab` builds the network the checkpoint actually holds."""
import importlib
import os

import torch

PKG = "semi-supervised-segmentation-cyclegan_amd"


def test(args, test_loader=None):
    F = importlib.import_module(PKG + ".functional")
    arch = importlib.import_module(PKG + ".arch")
    utils = importlib.import_module(PKG + ".utils")
    n_channels = {'voc2012': 21, 'cityscapes': 20, 'acdc': 4}[args.dataset]
    dev = torch.device("cuda", args.gpu_ids[0])
    if test_loader is None:
        du = importlib.import_module(PKG + ".data_utils")
        from torch.utils.data import DataLoader
        tr = du.get_transformation((args.crop_height, args.crop_width), resize=True, dataset=args.dataset, device_finish=True)
        cls = {'voc2012': du.VOCDataset, 'cityscapes': du.CityscapesDataset, 'acdc': du.ACDCDataset}[args.dataset]
        root = {'voc2012': './data/VOC2012', 'cityscapes': './data/Cityscape', 'acdc': './data/ACDC'}[args.dataset]
        test_set = cls(root_path=root, name='test', ratio=0.5, transformation=tr, augmentation=None)
        test_loader = du.DeviceLoader(DataLoader(test_set, batch_size=args.batch_size, shuffle=False), tr, dev)
    net = getattr(args, 'testing_gen', None) or 'resnet_9blocks_softmax'
    Gsi = arch.define_Gen(input_nc=3, output_nc=n_channels, ngf=args.ngf, netG=net, norm=args.norm,
                          use_dropout=not args.no_dropout, gpu_ids=args.gpu_ids)
    semi = args.model == 'semisupervised_cycleGAN'
    try:
        ckpt = utils.load_checkpoint('%s/latest_%s.ckpt' % (args.checkpoint_dir, 'semisuper_cycleGAN' if semi else 'supervised_model'))
        Gsi.load_state_dict(ckpt['Gsi'])
    except Exception:
        print(' [*] No checkpoint!')
    out = os.path.join(args.results_dir, 'unsupervised' if semi else 'supervised')
    os.makedirs(out, exist_ok=True)
    Gsi.eval()
    with torch.no_grad():
        for i, (image_test, image_name) in enumerate(test_loader):
            image_test = utils.cuda(image_test, args.gpu_ids)
            prediction = F.argmax_index(F.softmax2d(Gsi(image_test))).cpu().numpy()
            for j in range(prediction.shape[0]):
                utils.colorize_mask(prediction[j], args.dataset).save(os.path.join(out, image_name[j] + '.png'))
            print('Epoch-', str(i + 1), ' Done!')
