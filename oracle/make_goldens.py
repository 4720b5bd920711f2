"""Generate tests/golden/*.npz by importing the reference.  BUILD-CONTAINER ONLY.

Run:  python oracle/make_goldens.py            (needs /root/reference; never runs on the GPU box)

For every case it
  1. builds the reference `EGNNDynamics(model=LEFTNet)` from /root/reference (through the
     stand-in modules in oracle/_stubs for torch_scatter / torch_geometric / pytorch_lightning),
     loads the hash-generated synthetic state dict with `strict=True` (pins names + shapes),
  2. runs the reference in float64 and float32,
  3. checks the oracle restatement (oracle/leftnet_oracle.py, literal node frame, float64)
     against the float64 reference to <= 1e-10 and records the deviations,
  4. writes inputs, reference outputs and a few oracle stage tensors as a compressed .npz.

A fixture is data only: arrays + a JSON string describing the case.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "_stubs"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from oa_reactdiff.dynamics import EGNNDynamics  # noqa: E402  (reference)
from oa_reactdiff.model import LEFTNet  # noqa: E402  (reference)
from oa_reactdiff.utils import get_edges_index, get_mask_for_frag, get_n_frag_switch  # noqa: E402

import leftnet_oracle as oracle  # noqa: E402
from oareactdiff_amd.spec import (  # noqa: E402
    PRODUCTION_LEFTNET_CONFIG, hash_normal, hash_uniform, state_spec, synthetic_state_dict)

OUT = os.path.join(ROOT, "tests", "golden")


def rel(a: torch.Tensor, b: torch.Tensor) -> float:
    """max|a-b| / max|b|"""
    d = float((a.double() - b.double()).abs().max()) if a.numel() else 0.0
    s = float(b.double().abs().max()) if b.numel() else 1.0
    return d / max(s, 1e-300)


def make_inputs(case: dict):
    frag = [torch.tensor(x, dtype=torch.long) for x in case["fragments_nodes"]]
    node_nfs = case["node_nfs"]
    B = frag[0].numel()
    masks = [get_mask_for_frag(n) for n in frag]
    combined_mask = torch.cat(masks)
    n_frag_switch = get_n_frag_switch(frag)
    edge_index = get_edges_index(combined_mask, remove_self_edge=True)
    xh = []
    for k, nf in enumerate(node_nfs):
        n = int(frag[k].sum())
        pos = torch.from_numpy(hash_normal(f"{case['name']}.pos{k}", n * 3, 1234).reshape(n, 3))
        # per-(object, sample) centre-of-mass free, as diffusion/_utils.py:22-31 produces
        if n:
            mean = oracle._scatter_mean(pos, masks[k], B)
            pos = pos - mean[masks[k]]
        pos = pos * case.get("pos_scale", 1.0)
        if case.get("onehot", False):
            # [one-hot(5) | atomic number]-like features (utils/sampling_tools.py:64-108)
            u = hash_uniform(f"{case['name']}.type{k}", n, 1234)
            typ = np.digitize(u, [0.5, 0.8, 0.9])  # H C N O with P = .5 .3 .1 .1
            z = np.array([1, 6, 7, 8])[typ]
            feat = np.zeros((n, nf - 3))
            feat[np.arange(n), typ] = 1.0
            feat[:, -1] = z
            h = torch.from_numpy(feat)
        else:
            h = torch.from_numpy(hash_uniform(f"{case['name']}.h{k}", n * (nf - 3), 1234).reshape(n, nf - 3))
        xh.append(torch.cat([pos, h], dim=1).float())       # inputs are float32 values
    cnf = case["condition_nf"]
    conditions = torch.from_numpy(hash_uniform(case["name"] + ".cond", B * max(cnf, 1), 1234)
                                  .reshape(B, max(cnf, 1))).float()
    if case["t_1d"]:
        t = torch.tensor([0.314])
    else:
        t = torch.from_numpy(hash_uniform(case["name"] + ".t", B, 1234).reshape(B, 1)).float()
    # general edge lists (round 6): what EGNNDynamics.forward accepts besides the complete graph (egnn_dynamics.py:63-72)
    mode = case.get("edges")
    if mode == "edge_cutoff":                                  # the reference's own builder, utils/_graph_tools.py:31-33
        pos_all = torch.cat([x[:, :3] for x in xh], dim=0)
        edge_index = get_edges_index(combined_mask, pos=pos_all, edge_cutoff=case["edge_cutoff"], remove_self_edge=True)
    elif mode == "random_subset":                              # an arbitrary DIRECTED subset: pins which end every aggregation uses
        keep = torch.from_numpy(hash_uniform(case["name"] + ".keep", edge_index.size(1), 99) < case["keep"])
        edge_index = edge_index[:, keep]
        perm = torch.from_numpy(np.argsort(hash_uniform(case["name"] + ".order", edge_index.size(1), 98)))
        edge_index = edge_index[:, perm]                       # ... in an arbitrary order
    elif mode == "arbitrary":                                  # whatever a caller may hand over: self loops, duplicates, edges across samples
        N = combined_mask.numel()
        E = case["n_edges"]
        a = (hash_uniform(case["name"] + ".src", E, 97) * (N - 1)).astype(np.int64)          # the last node keeps no edge at all
        b = (hash_uniform(case["name"] + ".dst", E, 96) * (N - 1)).astype(np.int64)
        a[5], b[5] = a[4], b[4]                                # a duplicate
        b[7] = a[7]                                            # a self loop
        edge_index = torch.from_numpy(np.stack([a, b]))
    elif mode == "components":                                 # two components per sample: nodes with even / odd row index never meet
        a, b = edge_index
        edge_index = edge_index[:, (a % 2) == (b % 2)]
    else:
        assert mode is None
    return xh, edge_index, t, conditions, n_frag_switch, combined_mask


def run_case(case: dict) -> None:
    name = case["name"]
    cfg = dict(case["model_config"])
    node_nfs, cnf = case["node_nfs"], case["condition_nf"]
    xh, edge_index, t, conditions, n_frag_switch, combined_mask = make_inputs(case)

    spec = state_spec(cfg, node_nfs, cnf)
    sd32 = synthetic_state_dict(spec, cfg, seed=42, dtype=torch.float32)

    torch.set_default_dtype(torch.float32)
    ref = EGNNDynamics(model_config=dict(cfg), fragment_names=[f"o{k}" for k in range(len(node_nfs))],
                       node_nfs=node_nfs, edge_nf=0, condition_nf=cnf, model=LEFTNet,
                       device=torch.device("cpu"))
    ref.load_state_dict(sd32, strict=True)                      # pins names and shapes
    ref.eval()
    with torch.no_grad():
        out32, _ = ref(xh, edge_index, t, conditions, n_frag_switch, combined_mask)
        # the reference's own fp32 self-noise: same call, edges permuted
        perm = torch.from_numpy(np.argsort(hash_uniform(name + ".perm", edge_index.size(1), 7)))
        out32p, _ = ref(xh, edge_index[:, perm], t, conditions, n_frag_switch, combined_mask)

    torch.set_default_dtype(torch.float64)
    ref64 = EGNNDynamics(model_config=dict(cfg), fragment_names=[f"o{k}" for k in range(len(node_nfs))],
                         node_nfs=node_nfs, edge_nf=0, condition_nf=cnf, model=LEFTNet,
                         device=torch.device("cpu"))
    sd64 = {k: v.double() for k, v in sd32.items()}
    ref64.load_state_dict(sd64, strict=True)
    ref64.eval()
    xh64 = [x.double() for x in xh]
    with torch.no_grad():
        out64, _ = ref64(xh64, edge_index, t.double(), conditions.double(), n_frag_switch, combined_mask)

    # oracle, float64
    st_lit, st_ex = {}, {}
    o_lit = oracle.dynamics_forward(sd64, cfg, xh64, edge_index, t.double(), conditions.double(),
                                    n_frag_switch, combined_mask, cnf, nodeframe="literal",
                                    stages=st_lit, direct_vel=False)
    # (the exact-arithmetic node frame is a property of the COMPLETE graph per sample: the general-edge-list cases keep the literal frame)
    nf_hip = "literal" if case.get("edges") else "exact"
    o_ex = oracle.dynamics_forward(sd64, cfg, xh64, edge_index, t.double(), conditions.double(),
                                   n_frag_switch, combined_mask, cnf, nodeframe=nf_hip, stages=st_ex)
    torch.set_default_dtype(torch.float32)
    # oracle, float32 network on float64 geometry, exact node frame: the arithmetic model of
    # the HIP path, in plain torch
    o_ex32 = oracle.dynamics_forward(sd32, cfg, xh, edge_index, t, conditions, n_frag_switch,
                                     combined_mask, cnf, nodeframe=nf_hip, geom64=True)

    nz = [k for k in range(len(xh)) if xh[k].size(0)]
    pd = 3
    # objects may have different feature widths: compare [all vel | all h flattened]
    cat = lambda L_: torch.cat([L_[k][:, :pd].double().reshape(-1) for k in nz]
                               + [L_[k][:, pd:].double().reshape(-1) for k in nz]).unsqueeze(0)
    nvel = sum(xh[k].size(0) for k in nz) * pd
    R64, R32, R32P = cat(out64), cat(out32), cat(out32p)
    OL, OE, OE32 = cat(o_lit), cat(o_ex), cat(o_ex32)
    pd = nvel
    report = {
        "oracle_literal_f64_vs_ref_f64": rel(OL, R64),
        "oracle_exact_f64_vs_ref_f64": rel(OE, R64),
        "oracle_exact_f64_vs_ref_f64_vel": rel(OE[:, :pd], R64[:, :pd]),
        "oracle_exact_f64_vs_ref_f64_h": rel(OE[:, pd:], R64[:, pd:]),
        "oracle_f32net_f64geom_vs_ref_f64_vel": rel(OE32[:, :pd], R64[:, :pd]),
        "oracle_f32net_f64geom_vs_ref_f64_h": rel(OE32[:, pd:], R64[:, pd:]),
        "ref_f32_vs_ref_f64_vel": rel(R32[:, :pd], R64[:, :pd]),
        "ref_f32_vs_ref_f64_h": rel(R32[:, pd:], R64[:, pd:]),
        "ref_f32_edgeperm_vs_ref_f32_vel": rel(R32P[:, :pd], R32[:, :pd]),
        "ref_f32_edgeperm_vs_ref_f32_h": rel(R32P[:, pd:], R32[:, pd:]),
    }
    print(name, json.dumps(report, indent=1))
    assert report["oracle_literal_f64_vs_ref_f64"] <= 1e-10, "oracle restatement disagrees with the reference"

    arrays = {}
    for k in range(len(xh)):
        arrays[f"xh{k}"] = xh[k].numpy()
        arrays[f"ref64_out{k}"] = out64[k].numpy()
        arrays[f"ref32_out{k}"] = out32[k].numpy()
        arrays[f"oracle_exact64_out{k}"] = o_ex[k].numpy()
    arrays["edge_index"] = edge_index.numpy()
    arrays["t"] = t.numpy()
    arrays["conditions"] = conditions.numpy()
    arrays["n_frag_switch"] = n_frag_switch.numpy()
    arrays["combined_mask"] = combined_mask.numpy()
    # oracle stage tensors (float64 literal run == reference to 1e-10) that pin the stages
    for key in case.get("stages", ["labels", "edge_mask", "pos_frame", "s0", "pos_prjt", "dpos", "h_out"]):
        arrays["stage." + key] = st_lit[key].numpy()
    meta = dict(case)
    meta["report"] = report
    meta["weights"] = {"generator": "oareactdiff_amd.spec.synthetic_state_dict", "seed": 42}
    arrays["meta"] = np.array(json.dumps(meta))
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)


TEST_CFG = dict(pos_require_grad=False, cutoff=5.0, num_layers=2, hidden_channels=32, num_radial=8,
                in_hidden_channels=8)            # tests/dynamics/test_egnn_dynamics.py:49-56
PROD = dict(PRODUCTION_LEFTNET_CONFIG)

CASES = [
    # G1: the reference's own wrapper test set-up (test_egnn_dynamics.py:58-119): heterogeneous
    # objects, condition_nf=3, an empty object in sample 1, 1-D t
    dict(name="g1_wrapper_small", model_config=TEST_CFG, node_nfs=[4, 5, 6], condition_nf=3,
         fragments_nodes=[[2, 0], [2, 3], [1, 2]], t_1d=True),
    # G2: production config, two 23-atom triples, per-sample t
    dict(name="g2_prod_b2_n23", model_config=PROD, node_nfs=[9, 9, 9], condition_nf=1,
         fragments_nodes=[[23, 23]] * 3, t_1d=False, onehot=True),
    # G2s: production config, one 5-atom triple, with the big stage tensors
    dict(name="g2s_prod_b1_n5", model_config=PROD, node_nfs=[9, 9, 9], condition_nf=1,
         fragments_nodes=[[5]] * 3, t_1d=False, onehot=True,
         stages=["labels", "edge_mask", "pos_frame", "dist", "coord_diff", "f", "s0", "NE1",
                 "edgeweight0", "pos_prjt", "l0.s_gcl", "l0.edgeweight", "l0.dx_msg", "l0.dvec_msg",
                 "l0.s", "l0.vec", "l5.s", "l5.vec", "l5.edgeweight", "dpos", "h_out"]),
    # G3: cutoff bites (pos x3, cutoff 5), ragged samples: pins labels / active set / pos_frame
    dict(name="g3_cutoff_ragged", model_config=dict(TEST_CFG, num_layers=3), node_nfs=[9, 9, 9],
         condition_nf=1, fragments_nodes=[[5, 7, 3]] * 3, t_1d=False, pos_scale=3.0, onehot=True),
    # G3p: production dims, cutoff bites
    dict(name="g3p_prod_cutoff", model_config=dict(PROD, num_layers=2), node_nfs=[9, 9, 9],
         condition_nf=1, fragments_nodes=[[9, 6]] * 3, t_1d=False, pos_scale=4.0, onehot=True),
    # G6: model-test dims (tests/model/utils.py:24-32: H=32, R=32, L=6, cutoff=20), ragged objects
    dict(name="g6_h32_r32", model_config=dict(pos_require_grad=False, cutoff=20.0, num_layers=6,
                                                hidden_channels=32, num_radial=32, in_hidden_channels=8),
         node_nfs=[9, 9, 9], condition_nf=1, fragments_nodes=[[4, 6], [5, 6], [4, 2]], t_1d=False),
    # G10: reflect_equiv = False (leftnet.py:268-272, 794-796; the setting tests/model/test_equiv.py:172-185 exercises): the Equi message
    # carries x (x) coord_cross, the edge scalarisation keeps its sign.  Test dims with a biting cutoff, and production dims.
    dict(name="g10_noreflect_h32", model_config=dict(TEST_CFG, num_layers=3, reflect_equiv=False), node_nfs=[9, 9, 9],
         condition_nf=1, fragments_nodes=[[5, 7, 3]] * 3, t_1d=False, pos_scale=3.0, onehot=True),
    dict(name="g10p_noreflect_prod", model_config=dict(PROD, num_layers=2, reflect_equiv=False), node_nfs=[9, 9, 9],
         condition_nf=1, fragments_nodes=[[9, 6]] * 3, t_1d=False, onehot=True),
    # G11: general edge lists - the only other thing EGNNDynamics.forward accepts (egnn_dynamics.py:63-72).  edge_cutoff graphs from the
    # reference's own builder (test dims with ragged samples; production dims), an arbitrary directed subset in arbitrary order (28 % of the
    # ordered pairs missing: in- and out-degrees differ, some nodes have no incoming edge), and two disconnected components per sample
    # (tests/model/test_equiv.py:177-230).  Run by the general path (csrc/oard_general.h), literal node frame.
    dict(name="g11_edge_cutoff_h32", model_config=dict(TEST_CFG, num_layers=3), node_nfs=[9, 9, 9], condition_nf=1,
         fragments_nodes=[[5, 7, 3]] * 3, t_1d=False, pos_scale=2.0, onehot=True, edges="edge_cutoff", edge_cutoff=3.5),
    dict(name="g11p_edge_cutoff_prod", model_config=dict(PROD, num_layers=2), node_nfs=[9, 9, 9], condition_nf=1,
         fragments_nodes=[[9, 6]] * 3, t_1d=False, pos_scale=2.0, onehot=True, edges="edge_cutoff", edge_cutoff=4.0),
    dict(name="g11_random_subset", model_config=dict(TEST_CFG, num_layers=2), node_nfs=[4, 5, 6], condition_nf=3,
         fragments_nodes=[[2, 3], [2, 3], [1, 2]], t_1d=True, edges="random_subset", keep=0.72),
    dict(name="g11_arbitrary", model_config=dict(TEST_CFG, num_layers=2), node_nfs=[9, 9, 9], condition_nf=1,
         fragments_nodes=[[3, 2, 4], [2, 0, 3], [1, 3, 2]], t_1d=False, pos_scale=1.5, onehot=True, edges="arbitrary", n_edges=80),
    dict(name="g11_components_noreflect", model_config=dict(TEST_CFG, num_layers=2, reflect_equiv=False), node_nfs=[9, 9, 9],
         condition_nf=1, fragments_nodes=[[6, 4]] * 3, t_1d=False, onehot=True, edges="components"),
]

if __name__ == "__main__":
    only = sys.argv[1:]
    for c in CASES:
        if only and c["name"] not in only:
            continue
        run_case(c)
