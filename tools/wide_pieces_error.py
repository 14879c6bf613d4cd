"""Probe (not part of the product): error of the wide scans at two / three bf16 pieces per operand against the one-CU fp32 scan
(same inputs, same samples): max |deter|, |probs| differences, relative loss difference, largest gradient error relative to
the tensor's max.  usage: python tools/wide_pieces_error.py [mrssm_large|mmtrssm_cfg3dims] [batch] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle.cases import CASES, build_batch, build_model, build_noise, with_sizes
from tests.conftest import product_from_case
from multimodal_mtrssm_amd import scan

name = sys.argv[1] if len(sys.argv) > 1 else "mrssm_large"
batch, steps = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (8, 100)
DEV = "cuda:0"
case = with_sizes(CASES[name], batch, steps)
oracle = build_model(case)
batch_t = tuple(b.to(DEV) for b in build_batch(case))
noise = {k: v.to(DEV) for k, v in build_noise(case).items()}
mr = case.kind == "mrssm"


def run(wide: bool, pieces: int):
    scan.WIDE_SCAN, scan.WIDE_PIECES = wide, pieces
    model = product_from_case(case, oracle, DEV)
    with torch.no_grad():
        state0 = model.initial_state((batch_t[1][:, 0], batch_t[2][:, 0]), noise)
        post, prior = model.rollout_representation(actions=batch_t[0], observations=(batch_t[1], batch_t[2]), prev_state=state0, noise=noise)
    out = model.shared_step(batch_t, noise)
    out["loss"].backward()
    torch.cuda.synchronize()
    return post, prior, {k: float(v) for k, v in out.items()}, {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}


ref = run(False, 3)
for pieces in (3, 2):
    got = run(True, pieces)
    if mr:
        same = torch.equal(ref[0].stoch, got[0].stoch)
        dd = float((ref[0].deter - got[0].deter).abs().max())
        dp = max(float((ref[0].distribution.probs - got[0].distribution.probs).abs().max()), float((ref[1].distribution.probs - got[1].distribution.probs).abs().max()))
    else:
        same = torch.equal(ref[0].stoch_l, got[0].stoch_l) and torch.equal(ref[0].stoch_h, got[0].stoch_h)
        dd = max(float((ref[0].deter_l - got[0].deter_l).abs().max()), float((ref[0].deter_h - got[0].deter_h).abs().max()))
        dp = max(float((ref[0].distribution_l.probs - got[0].distribution_l.probs).abs().max()), float((ref[0].distribution_h.probs - got[0].distribution_h.probs).abs().max()),
                 float((ref[1].distribution_l.probs - got[1].distribution_l.probs).abs().max()), float((ref[1].distribution_h.probs - got[1].distribution_h.probs).abs().max()))
    dl = max(abs(got[2][k] - ref[2][k]) / (abs(ref[2][k]) + 1e-12) for k in ref[2])
    dg = max(float((got[3][k] - g).abs().max()) / (float(g.abs().max()) + 1e-12) for k, g in ref[3].items())
    print(f"{name} B={batch} T={steps} pieces={pieces}: same samples {same} | max |d deter| {dd:.2e} | max |d probs| {dp:.2e} | max rel loss diff {dl:.2e} | max grad err / tensor max {dg:.2e}")
