"""The fused kernel's optional code paths must keep compiling for gfx950 (they are off by default, so nothing else builds them):
the one-sample-per-round kernel, the tile-order / pipeline options and the debug counters (tools/build_variant.sh uses them)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
FLAGS = ["--offload-arch=gfx950", "-O1", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-strict-aliasing", "-Wall",
         "-Wno-unused-function", "-Werror", "-DNGP_BUILD", "--cuda-device-only", "-c"]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("defs", [["-DRV_S=1"], ["-DRV_PIPELINE=1", "-DRV_TILE_ORDER=1", "-DRV_COUNTERS=1", "-DRV_BLOCK_SKIP=0"]],
                         ids=["single_sample_kernel", "pipeline_tileorder_counters"])
def test_render_fused_option_builds(tmp_path, defs):
    src = os.path.join(ROOT, "nerf-navigation_amd", "csrc", "render_fused.hip")
    out = subprocess.run([HIPCC] + FLAGS + defs + [src, "-o", str(tmp_path / "rf.o")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
