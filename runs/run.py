#!/usr/bin/env python3
"""`python runs/run.py --cfg configs/projects/dist/ssv2/vit-b16-8+16f.yaml [KEY VAL ...]`
(reference runs/run.py:20-95): builds the run list train -> test from the config and launches one process
per GPU."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from dist_amd.utils.config import Config
from dist_amd.utils.launcher import launch_task
from runs.test import test
from runs.train import train


def _prepare_data(cfg):
    run_list = []
    if cfg.TRAIN.ENABLE:
        run_list.append([cfg.deep_copy(), train])
    if cfg.TEST.ENABLE:
        run_list.append([cfg.deep_copy(), test])
    return run_list


def main():
    cfg = Config(load=True)
    for cfg_run, func in _prepare_data(cfg):
        launch_task(cfg=cfg_run, init_method=cfg_run.get_args().init_method, func=func)
    print("Finish running with config: {}".format(cfg.args.cfg_file))


if __name__ == "__main__":
    main()
