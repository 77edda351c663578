"""Flag surface of the training path (mirrors /root/reference/scene_generation/args.py:10-113: same
flag names, types and defaults, so reference command lines and checkpoint ``args`` dicts carry
over).  Dataset-path flags are kept for CLI compatibility; the MI355X bench feeds synthetic
COCO-shaped batches (scene_generation_amd.synthetic)."""
import argparse
import os
import socket
from datetime import datetime

from .utils import int_tuple, str_tuple, bool_flag

COCO_DIR = os.path.expanduser('datasets/coco')


def _build():
    p = argparse.ArgumentParser()
    A = p.add_argument
    # optimisation (args.py:13-16)
    A('--batch_size', default=12, type=int)
    A('--num_iterations', default=1000000, type=int)
    A('--learning_rate', default=1e-4, type=float)
    A('--mask_learning_rate', default=1e-5, type=float)
    # dataset (args.py:19-46)
    A('--image_size', default='128,128', type=int_tuple)
    A('--num_train_samples', default=None, type=int)
    A('--num_val_samples', default=1024, type=int)
    A('--shuffle_val', default=True, type=bool_flag)
    A('--loader_num_workers', default=4, type=int)
    for flag, rel in [('coco_train_image_dir', 'images/train2017'), ('coco_val_image_dir', 'images/val2017'),
                      ('coco_train_instances_json', 'annotations/instances_train2017.json'),
                      ('coco_train_stuff_json', 'annotations/stuff_train2017.json'),
                      ('coco_val_instances_json', 'annotations/instances_val2017.json'),
                      ('coco_val_stuff_json', 'annotations/stuff_val2017.json'),
                      ('coco_panoptic_train', 'annotations/panoptic_train2017.json'),
                      ('coco_panoptic_val', 'annotations/panoptic_val2017.json'),
                      ('coco_panoptic_segmentation_train', 'panoptic/annotations/panoptic_train2017'),
                      ('coco_panoptic_segmentation_val', 'panoptic/annotations/panoptic_val2017')]:
        A('--' + flag, default=os.path.join(COCO_DIR, rel))
    A('--instance_whitelist', default=None, type=str_tuple)
    A('--stuff_whitelist', default=None, type=str_tuple)
    A('--coco_include_other', default=False, type=bool_flag)
    A('--min_object_size', default=0.02, type=float)
    A('--min_objects_per_image', default=3, type=int)
    A('--max_objects_per_image', default=8, type=int)
    A('--coco_stuff_only', default=True, type=bool_flag)
    A('--is_panoptic', default=False, type=bool_flag)
    # generator (args.py:50-64)
    A('--mask_size', default=32, type=int)
    A('--embedding_dim', default=128, type=int)
    A('--gconv_dim', default=128, type=int)
    A('--gconv_hidden_dim', default=512, type=int)
    A('--gconv_num_layers', default=5, type=int)
    A('--mlp_normalization', default='none', type=str)
    A('--activation', default='leakyrelu-0.2')
    A('--pool_size', default=100, type=int)
    A('--output_nc', default=3, type=int)
    A('--n_downsample_global', default=4, type=int)
    A('--box_dim', default=128, type=int)
    A('--use_attributes', default=True, type=bool_flag)
    A('--beta1', default=0.5, type=float)
    A('--box_noise_dim', default=64, type=int)
    A('--mask_noise_dim', default=64, type=int)
    # appearance (args.py:67-68)
    A('--rep_size', default=32, type=int)
    A('--appearance_normalization', default='batch')
    # generator losses (args.py:71-79)
    A('--l1_pixel_loss_weight', default=.0, type=float)
    A('--bbox_pred_loss_weight', default=10, type=float)
    A('--vgg_features_weight', default=10.0, type=float)
    A('--d_img_weight', default=1.0, type=float)
    A('--d_img_features_weight', default=10.0, type=float)
    A('--d_mask_weight', default=1.0, type=float)
    A('--d_mask_features_weight', default=10.0, type=float)
    A('--d_obj_weight', default=0.1, type=float)
    A('--ac_loss_weight', default=0.1, type=float)
    # image D (args.py:82-86)
    A('--ndf', default=64, type=int)
    A('--num_D', default=2, type=int)
    A('--norm_D', default='instance', type=str)
    A('--n_layers_D', default=3, type=int)
    A('--no_lsgan', default=False, type=bool_flag)
    # mask D (args.py:89-92)
    A('--ndf_mask', default=64, type=int)
    A('--num_D_mask', default=1, type=int)
    A('--norm_D_mask', default='instance', type=str)
    A('--n_layers_D_mask', default=2, type=int)
    # object D (args.py:95-100)
    A('--gan_loss_type', default='gan')
    A('--d_normalization', default='batch')
    A('--d_padding', default='valid')
    A('--d_activation', default='leakyrelu-0.2')
    A('--d_obj_arch', default='C4-64-2,C4-128-2,C4-256-2')
    A('--crop_size', default=32, type=int)
    # output (args.py:103-109)
    stamp = datetime.now().strftime('%b%d_%H-%M-%S')
    A('--print_every', default=100, type=int)
    A('--checkpoint_every', default=10000, type=int)
    A('--output_dir', default=os.path.join(os.getcwd(), 'output', stamp + '_' + socket.gethostname()))
    A('--checkpoint_name', default='checkpoint')
    A('--restore_from_checkpoint', default=False, type=bool_flag)
    # extension (not a reference flag): torchvision vgg19 state_dict with the ImageNet weights the reference's VGGLoss
    # downloads (losses.py:183); without it a --vgg_features_weight > 0 run trains against RANDOM VGG19 features and says so
    A('--vgg_weights', default=None, type=str)
    return p


parser = _build()


def get_args(argv=None):
    return parser.parse_args(argv)
