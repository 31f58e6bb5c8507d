"""The index arithmetic of the 16x16x4 kernel family (csrc/mlp16.hip) on the host: tools/mlp16_layout_model.py mirrors the
packed-image formula, the chain order of the k-steps, the transposed image, the sample-major publish / operand reads of the
weight-gradient rounds and the block -> tensor store map, and checks each against plain matrix products.  No GPU: this is
what catches a layout slip before a GPU run does (the GPU parity tests are tests/test_gpu_mlp16.py)."""
import importlib.util
import os

import pytest

from conftest import ROOT


def _model():
    spec = importlib.util.spec_from_file_location("mlp16_layout_model", os.path.join(ROOT, "tools", "mlp16_layout_model.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("md,D", [(64, 3), (64, 30), (128, 4), (64, 17)])
def test_mlp16_layout_maps(md, D):
    assert _model().self_check(md, D)


def test_library_reports_width_256_support():
    """Size queries need no GPU: width 256 is served for every net kind by the 16-tile family (the modular actor since round 3)."""
    import pime_amd.native as nt
    L = nt.lib()
    assert L.pime_mlp_packed_floats(nt.MLP_CRITIC, 30, 0, 256) > 2 * 256 * 256
    assert L.pime_mlp_packed_floats(nt.MLP_PLAIN_ACTOR, 30, 0, 256) > 2 * 256 * 256
    assert L.pime_ppo_bwd_image_floats(nt.MLP_CRITIC, 30, 0, 256) == 2 * 256 * 256
    assert L.pime_ppo_workspace_floats(nt.MLP_CRITIC, 4096, 256) > 0
    assert L.pime_mlp_packed_floats(nt.MLP_MODULAR_ACTOR, 4, 1, 256) > 256 * 256 + 2 * 256 * 128
    assert L.pime_ppo_bwd_image_floats(nt.MLP_MODULAR_ACTOR, 4, 1, 256) == 2 * 256 * 256      # net.0^T + the two towers' 128 -> 256 transposes
    assert L.pime_ppo_workspace_floats(nt.MLP_MODULAR_ACTOR, 4096, 256) > 0
    assert L.pime_mlp_packed_floats(nt.MLP_CRITIC, 3, 0, 96) == 0


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_head_gradient_butterfly_model(seed):
    """tools/butterfly_model.py executes half_sums16's cross-lane steps (v_permlane16_swap, row_ror:8, row_half_mirror, quad_perm,
    bank-masked selects) on 64-lane arrays: every lane's two results are the 32-lane sums of the registers the kernel says it owns,
    and the writer lanes cover each of the 32 features of a tile exactly once (csrc/ppo_fused.hip: head weight gradient)."""
    spec = importlib.util.spec_from_file_location("butterfly_model", os.path.join(ROOT, "tools", "butterfly_model.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.self_check(seed)


@pytest.mark.parametrize("kt,ot", [(4, 4), (4, 2), (2, 2), (2, 1)])
def test_narrow_tile_forward_model(kt, ot):
    """tools/narrow_tile_model.py: the 16-lane-tile layer of the fused rollout kernels (v_mfma_f32_16x16x4_f32 re-addressing the
    32x32x2 packed image and the lane-half-major vectors, csrc/rollout_policy.hpp) reproduces W x + b for the widths 128 / 64 and
    the modular actor's 2:1 tower layers."""
    spec = importlib.util.spec_from_file_location("narrow_tile_model", os.path.join(ROOT, "tools", "narrow_tile_model.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.self_check(kt, ot)
