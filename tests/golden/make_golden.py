#!/usr/bin/env python3
"""Generates tests/golden/*.npz.  Run in the authoring container only (it reads /root/reference, which does not exist
on the GPU box); the .npz files are committed, this script documents how they were made.

trunc_exp.npz : inputs, forward outputs and input gradients of the REFERENCE's own activation.trunc_exp
                (activation.py:5-18; pure torch, importable -- SURVEY.md 8c) on CPU float32.
The reference holds no other runnable code for the hot path (its kernels are CUDA) and no fixtures of its own.
"""
import importlib.util
import os
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
warnings.simplefilter("ignore")

spec = importlib.util.spec_from_file_location("ref_activation", "/root/reference/activation.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

rng = np.random.default_rng(0)
x = np.concatenate([rng.normal(scale=4.0, size=2000), np.linspace(-30, 30, 241), [-15.0, 15.0, -15.0001, 15.0001, 0.0, -0.0, 88.0, -104.0]]).astype(np.float32)
g = rng.normal(size=x.shape).astype(np.float32)
xt = torch.from_numpy(x).requires_grad_(True)
y = ref.trunc_exp(xt)
y.backward(torch.from_numpy(g))
np.savez_compressed(os.path.join(HERE, "trunc_exp.npz"), x=x, g=g, y=y.detach().numpy(), dx=xt.grad.numpy())
print("trunc_exp.npz:", x.shape, "finite y:", int(np.isfinite(y.detach().numpy()).sum()))
