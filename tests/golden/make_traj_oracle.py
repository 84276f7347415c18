"""Oracle curves of the 200-step loss-curve test, stored so that the GPU box does not spend ~2.5 minutes of CPU time
re-running the oracle loops (the GPU suite was dominated by the CPU oracle on the box).

    python tests/golden/make_traj_oracle.py        (CPU, ~3 minutes)

Writes tests/golden/traj_oracle.npz - per-step losses of tests.traj_stream.oracle_loop in the four settings
tests/test_gpu_train.py::test_loss_curve_200_steps_vs_oracle compares the engine with:
    ref_fp32 : fp32 master weights, forward / backward on their bf16 rounding (what the engine computes with), fp32 AdamW
    emu_fp32 : the same with the reference's bf16-autocast activation path emulated (the sensitivity envelope)
    ref_bf16 : the recipe's precision - bf16 parameters, bf16-rounded gradients, bf16 AdamW state
    emu_bf16 : the same with bf16 activations
The oracle is deterministic; tests/test_oracle_golden.py::test_stored_oracle_curves_are_the_oracles recomputes the first 30
steps of every curve in the CPU tier, so a change of the oracle or of the stream cannot leave a stale fixture behind.
(The REAL reference's curves are tests/golden/traj.npz, make_golden_traj.py.)"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests.traj_stream import oracle_loop  # noqa: E402

SETTINGS = {"ref_fp32": (False, False, True), "emu_fp32": (False, True, True), "ref_bf16": (True, False, False), "emu_bf16": (True, True, False)}

if __name__ == "__main__":
    out = {}
    for name, (bf16_state, bf16_acts, round_w) in SETTINGS.items():
        out[name] = np.array(oracle_loop(bf16_state, bf16_acts, round_weights=round_w)[0], np.float64)
        print(name, out[name][::40].round(4), flush=True)
    np.savez_compressed(os.path.join(HERE, "traj_oracle.npz"), **out)
