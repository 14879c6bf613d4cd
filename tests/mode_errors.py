"""Error of each conv MFMA mode against the golden fixtures (measurement script, run by hand on the MI355X: `python tests/mode_errors.py`;
it lives under tests/ because it uses the oracle): losses (relative), posterior
probabilities / deter (absolute), gradients (relative to the tensor's max)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multimodal_mtrssm_amd import conv
from oracle.cases import CASES, GOLDEN_CASES, build_model
from tests.conftest import golden_batch, golden_noise, load_golden, product_from_case

DEV = "cuda:0"
REF = {}
for mode in ("f32", "bf16x3", "bf16x2", "bf16"):
    conv.set_mfma_mode(mode)
    worst = {"loss_rel": 0.0, "post_abs": 0.0, "grad_rel": 0.0}
    for name in GOLDEN_CASES:
        case = CASES[name]
        fx = load_golden(name)
        oracle = build_model(case)
        batch, noise = golden_batch(fx), golden_noise(fx)
        ref = oracle.shared_step(batch, noise); ref["loss"].backward()
        model = product_from_case(case, oracle, DEV)
        out = model.shared_step(tuple(b.to(DEV) for b in batch), {k: v.to(DEV) for k, v in noise.items()})
        out["loss"].backward()
        for k in out:
            worst["loss_rel"] = max(worst["loss_rel"], abs(float(out[k]) - float(fx[f"loss/{k}"])) / abs(float(fx[f"loss/{k}"])))
        rg = {k: p.grad for k, p in oracle.named_parameters() if p.grad is not None}
        for k, p in model.named_parameters():
            if k in rg:
                worst["grad_rel"] = max(worst["grad_rel"], float((p.grad.cpu() - rg[k]).abs().max() / (rg[k].abs().max() + 1e-12)))
        with torch.no_grad():  # posterior rollout against the fp32-MFMA mode's (the first mode run)
            obs = (batch[1].to(DEV), batch[2].to(DEV))
            dn = {k: v.to(DEV) for k, v in noise.items()}
            s0 = model.initial_state((obs[0][:, 0], obs[1][:, 0]), dn)
            post, prior = model.rollout_representation(actions=batch[0].to(DEV), observations=obs, prev_state=s0, noise=dn)
        for attr in ("deter", "deter_l", "deter_h"):
            if hasattr(post, attr):
                v = getattr(post, attr).cpu()
                key = (name, attr)
                if mode == "f32":
                    REF[key] = v
                else:
                    worst["post_abs"] = max(worst["post_abs"], float((v - REF[key]).abs().max()))
    print(mode, {k: f"{v:.2e}" for k, v in worst.items()})
