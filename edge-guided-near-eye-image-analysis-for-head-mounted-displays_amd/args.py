"""Command-line flags of the entry scripts -- same names and defaults as the reference's args.py:30-65
(+ ``--synthetic`` because neither datasets nor checkpoints ship with the reference)."""
import argparse
from pprint import pprint

import torch


def parse_precision(prec):
    """args.py:17-28: 64 silently maps to float32."""
    if prec in (32, 64):
        return torch.float32
    if prec == 16:
        return torch.float16
    print('Invalid precision. Reverting to float32.')
    return torch.float32


def build_parser():
    p = argparse.ArgumentParser()
    a = p.add_argument
    a('--lr', type=float, default=5e-4, help='learning rate')
    a('--prec', type=int, default=32, help='precision. 16, 32, 64')
    a('--disp', type=int, default=0, help='display intermediate ouput')
    a('--model', type=str, default='ritnet_v2', help='select model')
    a('--curObj', type=str, default=None, help='select curriculum to train on')
    a('--epochs', type=int, default=40, help='total number of epochs')
    a('--resume', type=int, default=0, help='resume?')
    a('--workers', type=int, default=0, help='number of workers')
    a('--overfit', type=int, default=0, help='overfit to N batches?')
    a('--expname', type=str, default='dev', help='experiment number')
    a('--selfCorr', type=int, default=0, help='self regulation?')
    a('--loadfile', type=str, default='./weights/all.git_ok', help='load experiment')
    a('--path2data', type=str, default='/media/rakshit/Monster', help='path to dataset')
    a('--batchsize', type=int, default=12, help='select a batchsize')
    a('--test_mode', type=str, default='leaveoneout', help='testing strategy?')
    a('--disentangle', type=int, default=1, help='Explicit dataset bias removal?')
    a('--test_save_op_masks', type=int, default=0, help='save predicted output masks')
    a('--setting', type=str, default='error', help='where is setting ?')
    a('--id', type=int, default=0)
    a('--edge_thres', type=int, default=0, help='edge thres?')
    a('--test_normal', type=int, default=0)
    a('--record_iou', type=int, default=0)
    a('--record_img', type=int, default=0)
    a('--iou_filename', type=str, default='test.pkl')
    a('--visual_dir', type=str, default='iris')
    a('--method', type=str, default='baseline')
    a('--synthetic', type=int, default=0, help='N>0: run on N synthetic TEyeD-shaped frames (no dataset / checkpoint needed)')
    a('--device_prep', type=int, default=0, help='1: compute the distance maps from the labels on the GPU (egne_amd.dataprep) '
                                                 'instead of taking them from the Dataset (CurriculumLib.py:131-136)')
    return p


def parse_args(argv=None):
    args = build_parser().parse_args(argv)
    if args.curObj is None and not args.synthetic:
        raise SystemExit('--curObj is required (or use --synthetic N)')
    print('------')
    print('parsed arguments:')
    pprint(vars(args))
    args.prec = parse_precision(args.prec)
    return args
