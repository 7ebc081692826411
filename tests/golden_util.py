"""Helpers shared by the tests: load committed golden fixtures (numpy, no pickle)
and rebuild oracle modules from the stored state dicts."""
import json
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return {k: z[k] for k in z.files}


def sub(d, prefix):
    return {k[len(prefix):]: torch.from_numpy(np.array(v)) for k, v in d.items() if k.startswith(prefix)}


def overfitting_json():
    """The reference's own 5-image annotation fixture (imSitu/overfitting.json),
    kept as data under tests/golden/."""
    return json.load(open(os.path.join(GOLDEN, "overfitting.json")))


def oracle_fcggnn(g3, steps=4, enc=None):
    from oracle.ref_encoder import RefEncoder
    from oracle.ref_model import RefBackbone, RefFCGGNN
    enc = enc or RefEncoder(overfitting_json())
    cfg = dict(depth=int(g3["cfg_depth"]), width=int(g3["cfg_width"]), blocks=tuple(int(b) for b in g3["cfg_blocks"]))
    net = RefFCGGNN(enc, int(g3["D"]), steps=steps, backbone_factory=lambda: RefBackbone(**cfg))
    missing, unexpected = net.load_state_dict(sub(g3, "state/"), strict=True)
    return net, enc, cfg
