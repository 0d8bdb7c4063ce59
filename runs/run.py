#!/usr/bin/env python3
"""`python runs/run.py --cfg configs/projects/dist/ssv2/vit-b16-8+16f.yaml [KEY VAL ...]`
(reference runs/run.py:20-95): builds the run list from the config - train, single-view test, and (TEST.AUTOMATIC_MULTI_SCALE_TEST)
a second, multi-view test - and launches one process per GPU for each entry."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from dist_amd.utils.config import Config
from dist_amd.utils.launcher import launch_task
from runs.test import test
from runs.train import train


def multi_view_setting(cfg):
    """(views, crops) of the automatic multi-view test (reference runs/run.py:48-62): 10 x 1 by default, 10 x 3 for Kinetics / EPIC,
    3 x 1 for SSv2, 1 x 3 for ImageNet fine-tuning; TEST.OVERRIDE_MULTI_SCALE_TEST (every DiST yaml sets it: 3 x 1) wins."""
    ds = str(cfg.TEST.DATASET)
    views, crops = 10, 1
    if "kinetics" in ds or "epickitchen" in ds:
        crops = 3
    if "imagenet" in ds and not cfg.PRETRAIN.ENABLE:
        views, crops = 1, 3
    if "ssv2" in ds:
        views, crops = 3, 1
    ov = getattr(cfg.TEST, "OVERRIDE_MULTI_SCALE_TEST", None)
    if ov is not None and ov.ENABLE:
        views, crops = ov.NUM_ENSEMBLE_VIEWS, ov.NUM_SPATIAL_CROPS
    return views, crops


def _prepare_data(cfg):
    run_list = []
    if cfg.TRAIN.ENABLE:
        run_list.append([cfg.deep_copy(), train])
    if cfg.TEST.ENABLE:
        run_list.append([cfg.deep_copy(), test])                      # single view (TEST.NUM_ENSEMBLE_VIEWS x TEST.NUM_SPATIAL_CROPS as configured)
        if getattr(cfg.TEST, "AUTOMATIC_MULTI_SCALE_TEST", False):
            cfg.LOG_MODEL_INFO = False
            cfg.LOG_CONFIG_INFO = False
            cfg.TEST.NUM_ENSEMBLE_VIEWS, cfg.TEST.NUM_SPATIAL_CROPS = multi_view_setting(cfg)
            cfg.TEST.LOG_FILE = "val_{}clipsx{}crops.log".format(cfg.TEST.NUM_ENSEMBLE_VIEWS, cfg.TEST.NUM_SPATIAL_CROPS)
            run_list.append([cfg.deep_copy(), test])
    return run_list


def main():
    cfg = Config(load=True)
    for cfg_run, func in _prepare_data(cfg):
        launch_task(cfg=cfg_run, init_method=cfg_run.get_args().init_method, func=func)
    print("Finish running with config: {}".format(cfg.args.cfg_file))


if __name__ == "__main__":
    main()
