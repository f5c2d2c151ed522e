"""The plain-C oracle (oracle/mft_oracle_c.c) against the reference-generated
golden vectors and against the torch-based oracle.  CPU only."""
import subprocess
from pathlib import Path

import numpy as np
import pytest
import torch

import golden_inputs as gi
from oracle import mft_oracle as O

REPO = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def CO():
    subprocess.run(["make", "-C", str(REPO / "oracle")], check=True, capture_output=True)
    from oracle import c_oracle
    c_oracle.load()
    return c_oracle


@pytest.fixture(scope="module")
def ops(golden_dir):
    return dict(np.load(golden_dir / "ops.npz"))


def C_(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def test_c_corr_lookup_and_pyramid(CO, ops):
    d = gi.ops_inputs()
    h, w = gi.OPS_H, gi.OPS_W
    N = h * w
    vol = CO.corr_volume(C_(d["fmap1"].reshape(256, N)), C_(d["fmap2"].reshape(256, N)))
    levels, sizes = [vol], [(h, w)]
    for l in range(3):
        hl, wl = sizes[-1]
        levels.append(CO.avg_pool2(levels[-1], N, hl, wl).reshape(N, -1))
        sizes.append((hl // 2, wl // 2))
    rows = ops["pyr_rows_idx"]
    for l, lvl in enumerate(levels):
        assert np.abs(lvl[rows].reshape(ops[f"pyr{l}_rows"].shape) - ops[f"pyr{l}_rows"]).max() < 3e-5
    out = CO.corr_lookup([C_(l) for l in levels], sizes, C_(d["coords1"][0].reshape(2, N)))
    assert np.abs(out.reshape(1, 324, h, w) - ops["lookup"]).max() < 5e-5


def test_c_convex_upsample(CO, ops):
    d = gi.ops_inputs()
    up = CO.convex_upsample(C_(d["flow"][0]), C_(d["mask"][0]), 8.0)
    assert np.abs(up - ops["up_flow"][0]).max() < 2e-5
    up = CO.convex_upsample(C_(d["unc_lr"][0]), C_(d["mask"][0]), 1.0)
    assert np.abs(up - ops["up_unc"][0]).max() < 2e-5


def test_c_conv2d(CO, ops, weights_np):
    d = gi.ops_inputs()
    # flow head on a GRU-sized input: 3x3 128->256 (relu) then 3x3 256->2, vs torch
    x = d["net"][0]
    w1, b1 = weights_np["update_block.flow_head.conv1.weight"], weights_np["update_block.flow_head.conv1.bias"]
    y = np.maximum(CO.conv2d(C_(x), C_(w1), C_(b1)), 0)
    ref = torch.relu(torch.nn.functional.conv2d(torch.from_numpy(x)[None], torch.from_numpy(w1),
                                                 torch.from_numpy(b1), padding=1))[0].numpy()
    assert np.abs(y - ref).max() < 2e-5
    # 1x5 GRU gate conv
    wz, bz = weights_np["update_block.gru.convz1.weight"], weights_np["update_block.gru.convz1.bias"]
    hx = np.concatenate([d["net"][0], d["inp"][0], d["motion"][0]], 0)
    y = CO.conv2d(C_(hx), C_(wz), C_(bz))
    ref = torch.nn.functional.conv2d(torch.from_numpy(hx)[None], torch.from_numpy(wz), torch.from_numpy(bz),
                                     padding=(0, 2))[0].numpy()
    assert np.abs(y - ref).max() < 5e-5


def test_c_chain_select(CO, golden_dir):
    g = np.load(golden_dir / "sequence_stub.npz")
    L = tuple(C_(a) for a in gi.stub_flowou(0, 7))
    R = tuple(C_(a) for a in gi.stub_flowou(7, 9))
    flow, occ, sig = CO.chain(L, R)
    assert np.abs(flow - g["chain_flow"]).max() < 2e-5
    assert np.abs(occ - g["chain_occl"]).max() < 1e-5
    assert np.abs(sig - g["chain_sigma"]).max() < 1e-5
    # selection: C vs torch oracle on chained stub candidates
    cands = []
    for k, (a, b) in enumerate([(0, 9), (8, 9), (7, 9), (5, 9), (1, 9)]):
        Lk = tuple(C_(x) for x in gi.stub_flowou(0, a)) if a else (np.zeros((2, gi.SEQ_H, gi.SEQ_W), np.float32),
                                                                     np.zeros((1, gi.SEQ_H, gi.SEQ_W), np.float32),
                                                                     np.zeros((1, gi.SEQ_H, gi.SEQ_W), np.float32))
        cands.append(CO.chain(Lk, tuple(C_(x) for x in gi.stub_flowou(a, b))))
    f, o, s, idx = CO.select(cands, 0.02)
    tf, to, ts, tidx = O.select([tuple(torch.from_numpy(x) for x in c) for c in cands], 0.02)
    assert (idx == tidx.numpy()).all()
    assert np.abs(f - tf.numpy()).max() == 0 and np.abs(o - to.numpy()).max() == 0
    assert len(np.unique(idx)) >= 3
