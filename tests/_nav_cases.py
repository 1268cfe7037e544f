"""The planner and pose-filter cases of tests/golden/callers_nav.npz, rebuilt from the fixture on any device, so that the CPU test
(oracle field) and the GPU tests (NavQueries / NativeNavQueries) run the SAME restated callers (oracle/nav_oracle.py) on the same numbers."""
import os

import numpy as np
import torch

from oracle import nav_oracle as NO

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold():
    return np.load(os.path.join(GOLD, "callers_nav.npz"))


def planner_cfg(g, tag):
    """simulate.py:266-283 as the generator set it (tests/golden/make_callers_golden.py: nav_planner_cfg)"""
    return {"T_final": 2., "steps": int(g[f"{tag}_steps"]), "fade_out_epoch": int(g[f"{tag}_fade_out_epoch"]), "fade_out_sharpness": 10,
            "I": torch.eye(3), "g": 10., "mass": 1.}


def planner_case(g, tag, density_fn, dev="cpu"):
    """cost, per-state costs and the gradients w.r.t. (states, initial_accel) through `density_fn`"""
    t = lambda k: torch.from_numpy(g[f"{tag}_{k}"]).to(dev)                     # noqa: E731
    states, accel = t("states").requires_grad_(True), t("initial_accel").requires_grad_(True)
    res = NO.planner_costs(states, accel, t("start"), t("end"), planner_cfg(g, tag), t("robot_body"), density_fn, epoch=int(g[f"{tag}_epoch"]))
    res["total"].backward()
    return dict(res, grad_states=states.grad, grad_initial_accel=accel.grad)


def filter_case(g, get_rays_fn, render_fn, dev="cpu", hessian=True):
    """loss, gradient and the 12 x 12 Hessian of Estimator.measurement_fn through the injected queries"""
    t = lambda k: torch.from_numpy(g[f"mf_{k}"]).to(dev)                        # noqa: E731
    batch = g["mf_batch"]
    x = t("state").requires_grad_(True)
    loss = NO.measurement_loss(x, t("start"), t("sig"), t("target"), batch, get_rays_fn, render_fn)
    loss.backward()
    out = dict(loss=loss.detach(), grad=x.grad)
    if hessian:
        out["hessian"] = NO.measurement_hessian(t("state"), t("start"), t("sig"), t("target"), batch, get_rays_fn, render_fn)
    return out


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
