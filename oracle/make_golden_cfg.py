#!/usr/bin/env python3
"""Golden OUTPUTS of the reference's config loader.  TEST INFRASTRUCTURE, build container only (needs /root/reference, read-only).

Runs the reference's own `utils.config.Config` merge (base.yaml <- _BASE_RUN <- _BASE_MODEL <- _BASE chain <- the file <- KEY VAL
overrides; utils/config.py:30-38,95-152,177-246) over every DiST yaml of the reference (configs/projects/dist/{ssv2,k400}/*.yaml) and writes
the merged dictionaries to tests/golden/cfg_dist.json.  Nothing of the reference is copied: the fixture holds what its loader RETURNS.
`Config.__init__` itself is not called - it creates OUTPUT_DIR under the current directory, and /root/reference must not be written to -
its steps are (utils/config.py:30-38): _parse_args, _initialize_cfg, _load_yaml, _merge_cfg_from_base, _update_dict.

    python oracle/make_golden_cfg.py
"""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def merged(cfg_rel, opts=()):
    from utils.config import Config                       # the reference's class
    argv, cwd = sys.argv, os.getcwd()
    try:
        sys.argv = ["run.py", "--cfg", cfg_rel] + list(opts)
        os.chdir(REF)                                      # the loader opens ./configs/pool/base.yaml (read only)
        c = Config.__new__(Config)
        c._level = "cfg"
        c.args = c._parse_args()
        c.need_initialization = True
        base = c._initialize_cfg()
        d = c._load_yaml(c.args)
        d = c._merge_cfg_from_base(base, d)
        c._update_dict(d)
    finally:
        sys.argv = argv
        os.chdir(cwd)

    def plain(node):                                       # Config tree -> dict (values as the loader coerced them)
        return {k: (plain(v) if isinstance(v, Config) else v) for k, v in node.__dict__.items()
                if k not in ("args", "need_initialization", "cfg_dict") and not k.startswith("_")}
    return plain(c)


def main():
    assert os.path.isdir(REF), "the reference is only present in the build container"
    from make_golden import _stub_modules
    _stub_modules()
    sys.path.insert(0, REF)
    out = {}
    for path in sorted(glob.glob(os.path.join(REF, "configs", "projects", "dist", "*", "*.yaml"))):
        rel = os.path.relpath(path, REF)
        try:
            out[rel] = merged(rel)
        except FileNotFoundError as e:                     # the released L/14 yamls name a _BASE file the release does not contain
            out[rel] = {"__error__": f"FileNotFoundError: {os.path.relpath(e.filename, REF) if os.path.isabs(e.filename) else e.filename}"}
    # command-line overrides through the same loader (KEY VAL pairs, utils/config.py:177-232)
    rel = "configs/projects/dist/ssv2/vit-b16-8+16f.yaml"
    opts = ["TRAIN.BATCH_SIZE", "64", "OPTIMIZER.BASE_LR", "0.0001", "DATA.NUM_INPUT_FRAMES", "32"]
    out[rel + " :: " + " ".join(opts)] = merged(rel, opts)
    dst = os.path.join(ROOT, "tests", "golden", "cfg_dist.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", dst, len(out), "configs,", os.path.getsize(dst) // 1024, "KiB")


if __name__ == "__main__":
    main()
