"""Golden name map for published-checkpoint compatibility: runs the REFERENCE's own `rename_model_state`
(/root/reference/process_dist_cpkt.py:10-30; only that function is executed - the script's module level walks a directory of
the authors' machine) over every dist_net tensor name in its pre-release `ladder_net.*` spelling, for ViT-B/16 (383 tensors) and
the L/14 geometry, and stores {old name: new name} in tests/golden/ckpt_rename.json.  Runs only in the build container."""
import ast
import collections
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden", "ckpt_rename.json")

# the inverse of the reference's table, used ONLY to spell the inputs (old names) from today's names
OLD = [("dist_net.input_linears", "ladder_net.input_map_feat_nets"), ("dist_net.integration2temporal_nets", "ladder_net.s2t_fuse_nets"),
       ("dist_net.temporal2integration_nets", "ladder_net.t2s_fuse_nets"), ("dist_net.integration_nets", "ladder_net.spatial_nets"),
       ("dist_net.adapooling_nets", "ladder_net.final_temporal_nets"), ("dist_net.", "ladder_net.")]


def old_spelling(name):
    for new, old in OLD:
        if name.startswith(new):
            return old + name[len(new):]
    return name


def main():
    src = open("/root/reference/process_dist_cpkt.py").read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "rename_model_state"]
    assert len(fn) == 1
    ns = {"collections": collections}
    exec(compile(ast.Module(body=fn, type_ignores=[]), "process_dist_cpkt.py", "exec"), ns)
    from dist_amd import synth
    out = {}
    for gname in ("b16_8+16f", "l14_32+64f"):
        g = synth.geometry(gname)
        names = list(synth.dist_net_shapes(g)) + list(synth.visual_shapes(g)) + ["logit_scale"]
        pre = "backbone.base_encoder."
        old = collections.OrderedDict((pre + old_spelling(n), 0) for n in names)
        new = ns["rename_model_state"](old)
        assert len(new) == len(old)
        for o, n in zip(old.keys(), new.keys()):
            out[o] = n
    json.dump(out, open(OUT, "w"), indent=0, sort_keys=True)
    print("wrote", OUT, len(out), "names;", sum(1 for k in out if "ladder_net" in k), "in the old spelling")


if __name__ == "__main__":
    main()
