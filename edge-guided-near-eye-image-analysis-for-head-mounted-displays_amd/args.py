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


# (flag, type, default, help).  Names and defaults are the drop-in surface (reference args.py:30-65); the table form, the
# grouping and the help texts are this repo's.
_TRAINING = (
    ("lr", float, 5e-4, "Adam step size (train.py)"),
    ("epochs", int, 40, "upper bound on training epochs"),
    ("batchsize", int, 12, "frames per optimiser step and per rank"),
    ("workers", int, 0, "DataLoader worker processes"),
    ("overfit", int, 0, "N > 0: stop every epoch after N batches"),
    ("resume", int, 0, "1: continue from --loadfile"),
    ("selfCorr", int, 0, "self-consistency loss term (off in the reference pipeline; not built here)"),
    ("disentangle", int, 1, "dataset-identity confusion loss on the latent"),
)
_MODEL = (
    ("model", str, "ritnet_v2", "key into modelSummary.model_dict"),
    ("setting", str, "error", "path of the yaml with the 10 model switches (configs/*.yaml)"),
    ("prec", int, 32, "16 | 32 | 64 (64 maps to float32, as in the reference)"),
    ("edge_thres", int, 0, "1: binarise the edge map at 0.1 before it enters the network"),
    ("loadfile", str, "./weights/all.git_ok", "checkpoint to load"),
)
_DATA = (
    ("curObj", str, None, "name of the pickled curriculum object (cond_<name>.pkl)"),
    ("path2data", str, "/media/rakshit/Monster", "dataset root"),
    ("test_mode", str, "leaveoneout", "evaluation split strategy"),
    ("synthetic", int, 0, "N > 0: run on N synthetic TEyeD-shaped frames (no dataset / checkpoint needed)"),
    ("pipeline", int, 1, "1 (default): the frozen edge network of a batch on a second HIP stream next to the previous batch's training "
                         "step (egne_amd.pipeline; the logged loss is then the previous batch's); 0: back to back, with the reference's "
                         "per-stage timers"),
    ("device_prep", int, 0, "1: distance maps computed from the labels on the GPU (egne_amd.dataprep) instead of taken "
                            "from the Dataset (CurriculumLib.py:131-136); 2: the boundary weights of CurriculumLib.py:128-129 too "
                            "(parity unpinned: restated from OpenCV's published Canny / dilate)"),
)
_OUTPUT = (
    ("expname", str, "dev", "sub-directory of logs/<model>/"),
    ("disp", int, 0, "1: show intermediate outputs"),
    ("test_save_op_masks", int, 0, "1: write predicted masks"),
    ("record_iou", int, 0, None),
    ("record_img", int, 0, None),
    ("iou_filename", str, "test.pkl", None),
    ("visual_dir", str, "iris", None),
    ("method", str, "baseline", None),
    ("test_normal", int, 0, None),
    ("id", int, 0, None),
)


def build_parser():
    p = argparse.ArgumentParser(description="entry scripts of the MI355X hot path (test.py / train.py / evaluate.py)")
    for title, table in (("training", _TRAINING), ("model", _MODEL), ("data", _DATA), ("output", _OUTPUT)):
        grp = p.add_argument_group(title)
        for name, typ, default, text in table:
            grp.add_argument("--" + name, type=typ, default=default, help=text)
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
