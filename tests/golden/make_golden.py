#!/usr/bin/env python3
"""Generates tests/golden/*.npz.  Run in the authoring container only (it reads /root/reference, which does not exist
on the GPU box); the .npz files are committed, this script documents how they were made.

trunc_exp.npz : inputs, forward outputs and input gradients of the REFERENCE's own activation.trunc_exp
                (activation.py:5-18; pure torch, importable -- SURVEY.md 8c) on CPU float32.
freq_encoder.npz : inputs and outputs of the REFERENCE's pure-torch encoding.FreqEncoder (encoding.py:5-43; the class its
                CUDA freqencoder replaced, same layout [x | sin(2^f x), cos(2^f x)]: the commented-out line encoding.py:57
                shows the equivalence FreqEncoder(max_freq_log2=multires-1, N_freqs=multires) == freqencoder(degree=multires)),
                CPU float32, and the input gradients torch autograd gives for a fixed output gradient.
The reference holds no other runnable code for the hot path (its kernels are CUDA) and no fixtures of its own.
"""
import sys

sys.dont_write_bytecode = True      # nothing is written under /root/reference (no __pycache__ beside the files this script reads or imports)
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


spec = importlib.util.spec_from_file_location("ref_encoding", "/root/reference/encoding.py")
enc = importlib.util.module_from_spec(spec)
spec.loader.exec_module(enc)
deg = 6
fe = enc.FreqEncoder(input_dim=3, max_freq_log2=deg - 1, N_freqs=deg, log_sampling=True)
xf = np.concatenate([rng.uniform(-2, 2, size=(500, 3)), np.array([[0, 0, 0], [2, -2, 1], [-0.0, 1e-6, -1e-6], [1.5707964, 3.1415927, -3.1415927]])]).astype(np.float32)
gf = rng.normal(size=(xf.shape[0], fe.output_dim)).astype(np.float32)
xt = torch.from_numpy(xf).requires_grad_(True)
yf = fe(xt)
yf.backward(torch.from_numpy(gf))
np.savez_compressed(os.path.join(HERE, "freq_encoder.npz"), x=xf, g=gf, y=yf.detach().numpy(), dx=xt.grad.numpy(), degree=np.int32(deg))
print("freq_encoder.npz:", xf.shape, "->", tuple(yf.shape))
