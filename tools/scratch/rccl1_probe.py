"""Probe: 1-process step vs the RCCL path with ONE rank (FB_FORCE_DIST=1) on the regularised one-group schedule: which switches make them bit-identical."""
import os, sys, tempfile
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from tests import test_gpu_sharded as T
from tests.helpers import spawn_bounded

def run(env, mode, steps):
    out = tempfile.mkdtemp()
    T.OVERRIDES[:] = [o for o in T.OVERRIDES if not o.startswith("hyp.steps")] + [f"hyp.steps={steps}"]
    for k in ("FB_FORCE_DIST", "FB_EXCHANGE_OVERLAP", "FB_F32_SPLIT", "FB_REPLAY", "FB_WGRAD_STREAM"):
        os.environ.pop(k, None)
    os.environ.update({k: v for k, v in env.items() if k != "FB_FORCE_DIST"})
    T._single(out, mode)
    os.environ.update(env)
    spawn_bounded(T._run, (1, T._free_port(), out, mode, "nccl", "rccl1"), 1, timeout=120)
    os.environ.pop("FB_FORCE_DIST", None)
    ref, got = torch.load(os.path.join(out, "w1_r0.pt")), torch.load(os.path.join(out, "rccl1_r0.pt"))
    worst = max(float(((a - b).abs() / (b.abs() + 1e-12)).max()) for a, b in zip(got["grads"], ref["grads"]))
    rel = float(torch.cat([(a - b).reshape(-1) for a, b in zip(got["grads"], ref["grads"])]).norm() / torch.cat([b.reshape(-1) for b in ref["grads"]]).norm())
    same_state = all(torch.equal(got["state"][k], ref["state"][k]) for k in ref["state"])
    print(f"{mode} steps={steps} {env}: p.grad rel L2 {rel:.2e}, worst elementwise {worst:.2e}; state identical: {same_state}; "
          f"grad_norm {got['stats']['grad_norm']} vs {ref['stats']['grad_norm']}", flush=True)

if __name__ == "__main__":
    for steps in (1, 2, 3):
        for split in ("bf16x6", "f16x2"):
            run({"FB_FORCE_DIST": "1", "FB_F32_SPLIT": split}, "onegroup_gradreg", steps)
    run({"FB_FORCE_DIST": "1", "FB_F32_SPLIT": "bf16x6", "FB_EXCHANGE_OVERLAP": "0"}, "onegroup_gradreg", 3)
    run({"FB_FORCE_DIST": "1", "FB_F32_SPLIT": "bf16x6", "FB_REPLAY": "0"}, "onegroup_gradreg", 3)
    run({"FB_FORCE_DIST": "1", "FB_F32_SPLIT": "bf16x6", "FB_WGRAD_STREAM": "0"}, "onegroup_gradreg", 3)
