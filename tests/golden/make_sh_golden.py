#!/usr/bin/env python3
"""Generates tests/golden/sh_deg8.npz from the TEXT of the reference's spherical-harmonics kernel.

shencoder/src/shencoder.cu:50-355 holds the 64 outputs and 3 x 64 partial derivatives of the degree-8 real SH basis as
literal arithmetic, one assignment per term (`outputs[k] = <expr> ;`, `dx[k] = ...`, `dy[k] = ...`, `dz[k] = ...`).  The kernel
is CUDA and cannot run here, but its arithmetic can be read: this script reads those 256 right-hand sides as text, parses them
with the little grammar below (decimal literals with an optional `f`, the monomial names of shencoder.cu:45-48, + - *, parentheses
and pow(e, n) with a literal integer n; nothing else is accepted and nothing is executed), evaluates them in float64 on fixed seeded inputs and stores
inputs, outputs [B,64] and dy_dx [B,3,64].  Run in the authoring container only: /root/reference does not exist on the GPU box.
The .npz is data (inputs and expected values); no reference source text is stored in it or anywhere else in the repository.
"""
import sys

sys.dont_write_bytecode = True      # nothing is written under /root/reference (no __pycache__ beside the files this script reads or imports)
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/shencoder/src/shencoder.cu"

_TOKEN = re.compile(r"\s*(?:(\d+\.\d*(?:[eE][-+]?\d+)?|\d+)f?|([A-Za-z_][A-Za-z_0-9]*)|(.))")


def tokenize(text):
    out, pos = [], 0
    text = text.strip()
    while pos < len(text):
        m = _TOKEN.match(text, pos)
        if not m:
            raise ValueError(f"cannot tokenize {text[pos:]!r}")
        pos = m.end()
        if m.group(1) is not None:
            out.append(("num", float(m.group(1))))
        elif m.group(2) is not None:
            out.append(("id", m.group(2)))
        elif m.group(3) in "+-*(),":
            out.append((m.group(3), None))
        else:
            raise ValueError(f"unexpected character {m.group(3)!r} in {text!r}")
    return out


class Parser:
    """expr := term (('+'|'-') term)* ; term := unary ('*' unary)* ; unary := '-' unary | atom ; atom := num | id | '(' expr ')'"""

    def __init__(self, tokens, env):
        self.t, self.i, self.env = tokens, 0, env

    def peek(self):
        return self.t[self.i][0] if self.i < len(self.t) else None

    def take(self):
        tok = self.t[self.i]
        self.i += 1
        return tok

    def expr(self):
        v = self.term()
        while self.peek() in ("+", "-"):
            op = self.take()[0]
            r = self.term()
            v = v + r if op == "+" else v - r
        return v

    def term(self):
        v = self.unary()
        while self.peek() == "*":
            self.take()
            v = v * self.unary()
        return v

    def unary(self):
        if self.peek() == "-":
            self.take()
            return -self.unary()
        return self.atom()

    def atom(self):
        kind, val = self.take()
        if kind == "num":
            return val
        if kind == "id" and val == "pow":             # pow(<expr>, <small integer>) appears in a few derivative terms
            if self.take()[0] != "(":
                raise ValueError("pow without (")
            base = self.expr()
            if self.take()[0] != ",":
                raise ValueError("pow without ,")
            kind, n = self.take()
            if kind != "num" or n != int(n) or not 0 <= n <= 8 or self.take()[0] != ")":
                raise ValueError("pow exponent must be a small integer literal")
            return base ** int(n)
        if kind == "id":
            return self.env[val]                      # KeyError on anything that is not a monomial of shencoder.cu:45-48
        if kind == "(":
            v = self.expr()
            if self.take()[0] != ")":
                raise ValueError("missing )")
            return v
        raise ValueError(f"unexpected token {kind}")


def evaluate(text, env):
    p = Parser(tokenize(text), env)
    v = p.expr()
    if p.i != len(p.t):
        raise ValueError(f"trailing tokens in {text!r}")
    return v


def read_terms():
    """{'outputs': [64 strings], 'dx': [...], 'dy': [...], 'dz': [...]} from lines 50-355."""
    pat = re.compile(r"^\s*(outputs|dx|dy|dz)\[(\d+)\]\s*=\s*(.*?);")
    terms = {k: {} for k in ("outputs", "dx", "dy", "dz")}
    with open(SRC) as f:
        lines = f.readlines()[49:355]
    for line in lines:
        m = pat.match(line)
        if m:
            terms[m.group(1)][int(m.group(2))] = m.group(3)
    for k, d in terms.items():
        assert sorted(d) == list(range(64)), (k, len(d))
    return {k: [d[i] for i in range(64)] for k, d in terms.items()}


def main():
    rng = np.random.default_rng(8)
    v = rng.normal(size=(200, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    # off-sphere points too: the reference's polynomials are defined for any (x, y, z) and the renderer feeds unnormalised dirs nowhere,
    # but the encoder itself does not normalise (shencoder.cu:43)
    extra = np.concatenate([rng.uniform(-1.2, 1.2, size=(48, 3)), np.eye(3), -np.eye(3), np.zeros((1, 3)),
                            np.array([[0.6, 0.0, 0.8], [0.0, -0.6, 0.8], [1.0, 1.0, 1.0]])])
    pts = np.concatenate([v, extra]).astype(np.float32).astype(np.float64)      # exactly representable in binary32
    x, y, z = pts[:, 0], pts[:, 1], pts[:, 2]
    env = dict(x=x, y=y, z=z)
    env.update(xy=x * y, xz=x * z, yz=y * z, x2=x * x, y2=y * y, z2=z * z)        # shencoder.cu:45
    env["xyz"] = env["xy"] * z
    env.update(x4=env["x2"] ** 2, y4=env["y2"] ** 2, z4=env["z2"] ** 2)           # :46
    env.update(x6=env["x4"] * env["x2"], y6=env["y4"] * env["y2"], z6=env["z4"] * env["z2"])   # :47
    terms = read_terms()
    B = pts.shape[0]
    out = np.zeros((B, 64))
    jac = np.zeros((B, 3, 64))
    for k in range(64):
        out[:, k] = evaluate(terms["outputs"][k], env)
        for a, name in enumerate(("dx", "dy", "dz")):
            jac[:, a, k] = evaluate(terms[name][k], env)
    np.savez_compressed(os.path.join(HERE, "sh_deg8.npz"), inputs=pts.astype(np.float32), outputs=out, dy_dx=jac)
    print("sh_deg8.npz:", pts.shape, "->", out.shape, jac.shape, "max |Y|", float(np.abs(out).max()))


if __name__ == "__main__":
    main()
