"""Helpers shared by the tests: build our MILNet from the golden weight fixtures."""
import numpy as np
import torch

from conftest import load_weights

VARIANT = {  # tag -> (K, C, nonlinear, passing_v)
    "c16": (512, 1, True, False), "tcga": (512, 2, True, False), "musk": (166, 1, True, False),
    "tree": (1024, 2, True, False), "linq": (64, 3, False, False), "passv": (64, 2, True, True),
}


def state_dict_from_npz(p, nonlinear=True, passing_v=False):
    """Map the oracle parameter names back to the reference state_dict keys."""
    sd = {"i_classifier.fc.0.weight": p["fc_w"], "i_classifier.fc.0.bias": p["fc_b"],
          "b_classifier.fcc.weight": p["fcc_w"], "b_classifier.fcc.bias": p["fcc_b"]}
    if nonlinear:
        sd.update({"b_classifier.q.0.weight": p["q0_w"], "b_classifier.q.0.bias": p["q0_b"],
                   "b_classifier.q.2.weight": p["q2_w"], "b_classifier.q.2.bias": p["q2_b"]})
    else:
        sd.update({"b_classifier.q.weight": p["q0_w"], "b_classifier.q.bias": p["q0_b"]})
    if passing_v:
        sd.update({"b_classifier.v.1.weight": p["v_w"], "b_classifier.v.1.bias": p["v_b"]})
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}


def build_net(tag, device="cpu"):
    import dsmil
    K, C, nonlinear, passing_v = VARIANT[tag]
    net = dsmil.MILNet(dsmil.FCLayer(in_size=K, out_size=C),
                       dsmil.BClassifier(input_size=K, output_class=C, dropout_v=0.0,
                                         nonlinear=nonlinear, passing_v=passing_v))
    net.load_state_dict(state_dict_from_npz(load_weights(tag), nonlinear, passing_v), strict=True)
    return net.eval().to(device)
