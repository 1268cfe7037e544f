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
@pytest.mark.parametrize("defs", [["-DRV_S=1"], ["-DRV_PIPELINE=1", "-DRV_TILE_ORDER=1", "-DRV_COUNTERS=1", "-DRV_BLOCK_SKIP=0"], ["-DRV_CU_CHUNKS=1"]],
                         ids=["single_sample_kernel", "pipeline_tileorder_counters", "cu_local_tile_chunks"])
def test_render_fused_option_builds(tmp_path, defs):
    src = os.path.join(ROOT, "nerf-navigation_amd", "csrc", "render_fused.hip")
    out = subprocess.run([HIPCC] + FLAGS + defs + [src, "-o", str(tmp_path / "rf.o")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_no_single_rounding_conversions_in_the_shipped_kernels(tmp_path):
    """`(_Float16)(a * b)` must round twice (binary32 product, then binary16) to follow the reference; hipcc likes to merge it
    into v_fma_mixlo/mixhi_f16, which rounds once on gfx950.  csrc/ngp_device.h: ngp_f2h prevents the merge;
    this checks the optimised ISA of every kernel file for the instruction."""
    from concurrent.futures import ThreadPoolExecutor
    csrc = os.path.join(ROOT, "nerf-navigation_amd", "csrc")
    files = ["raymarching", "gridencoder", "shencoder", "freqencoder", "ffmlp", "ffmlp_backward", "ffmlp_generic", "render_fused", "field_train"]
    flags = [f for f in FLAGS if f not in ("-O1", "-c", "-Werror")] + ["-O3", "-S"]      # (-S leaves hipcc's --hip-link unused: a warning)

    def isa(name):
        out = subprocess.run([HIPCC] + flags + [os.path.join(csrc, name + ".hip"), "-o", str(tmp_path / (name + ".s"))],
                             capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        text = open(tmp_path / (name + ".s")).read()
        # (v_fma_mix_f32 is fine: its result is a correctly rounded binary32)
        return name, sum(text.count(m) for m in ("v_fma_mixlo", "v_fma_mixhi", "v_mad_mixlo", "v_mad_mixhi"))

    with ThreadPoolExecutor(max_workers=4) as pool:
        counts = dict(pool.map(isa, files))
    assert all(v == 0 for v in counts.values()), counts


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_fused_kernels_do_not_spill(tmp_path):
    """The two kernels of the fused path must keep their working set in registers: a spill puts scratch loads and stores in the tile loop
    (k_field_forward_lds once spilled 81 VGPRs = 400 B of scratch per lane at 4 workgroups per CU)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from kernel_resources import kernels
    src = os.path.join(ROOT, "nerf-navigation_amd", "csrc", "render_fused.hip")
    flags = [f for f in FLAGS if f not in ("-O1", "-c", "-Werror")] + ["-O3", "-S"]
    out = subprocess.run([HIPCC] + flags + [src, "-o", str(tmp_path / "rf.s")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    res = {k["name"]: k for k in kernels(open(tmp_path / "rf.s").read())}
    field = next(v for n, v in res.items() if "k_field_forward_lds" in n)
    frame = next(v for n, v in res.items() if "k_render_frame_multi" in n)
    assert field["spill"] == 0 and frame["spill"] == 0, (field, frame)
    assert field["vgpr"] <= 256 and frame["vgpr"] <= 256
    # the training step: the backward runs one wave per SIMD so that the accumulator tiles of the weight gradients (44 / 28 x 4 registers)
    # fit beside two tiles of activations; all 72 in one kernel spilled 110 registers, hence its two parts
    src = os.path.join(ROOT, "nerf-navigation_amd", "csrc", "field_train.hip")
    out = subprocess.run([HIPCC] + flags + [src, "-o", str(tmp_path / "ft.s")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    train = [k for k in kernels(open(tmp_path / "ft.s").read()) if "k_field_train" in k["name"]]
    assert len(train) == 4 and all(k["spill"] == 0 for k in train), train
