"""Evaluation metric of the reference's eval loop: PSNRMeter (nerf/utils.py:185-219), same interface (clear / update / measure /
report) so a Trainer-style caller can pass it in `metrics=[PSNRMeter()]` (main_nerf.py:113)."""
import numpy as np
import torch


class PSNRMeter:
    def __init__(self):
        self.V = 0
        self.N = 0

    def clear(self):
        self.V, self.N = 0, 0

    @staticmethod
    def _host(a):
        return a.detach().float().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)

    def update(self, preds, truths):
        """preds, truths [B, N, 3] or [B, H, W, 3] in [0, 1]: adds -10 log10(MSE) of this batch (peak value 1)."""
        p, t = self._host(preds), self._host(truths)
        # the reference keeps numpy's scalar types: float32 inputs give a float32 PSNR and a float32 running sum (nerf/utils.py:203-210);
        # pinned by tests/golden/callers_tier1.npz (three updates, executed reference code)
        self.V += -10 * np.log10(np.mean((p - t) ** 2))
        self.N += 1

    def measure(self):
        return self.V / self.N

    def write(self, writer, global_step, prefix=""):
        writer.add_scalar(f"{prefix}/PSNR" if prefix else "PSNR", self.measure(), global_step)

    def report(self):
        return f"PSNR = {self.measure():.6f}"
