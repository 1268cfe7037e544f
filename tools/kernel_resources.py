#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel in an ISA listing produced by `hipcc -S` (the amdhsa.kernels metadata):
   python tools/kernel_resources.py file.s [name-substring ...]"""
import re
import sys


def kernels(text):
    meta = text[text.index("amdhsa.kernels:"):]
    for blk in re.split(r"\n  - \.agpr_count:", meta)[1:]:
        get = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", blk).group(1))          # noqa: E731
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        yield dict(name=name, vgpr=get("vgpr_count"), agpr=int(blk.split("\n")[0]), spill=get("vgpr_spill_count"), sgpr=get("sgpr_count"),
                   scratch=get("private_segment_fixed_size"), lds=get("group_segment_fixed_size"))


if __name__ == "__main__":
    text = open(sys.argv[1]).read()
    for k in kernels(text):
        if len(sys.argv) < 3 or any(s in k["name"] for s in sys.argv[2:]):
            print(k)
