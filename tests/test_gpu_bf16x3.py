"""PIME_GRAD_BF16X3=1 (opt-in): the streamed layers of the 16-tile gradient kernels on bf16 matrix instructions, every f32 operand
split into three bf16 pieces (csrc/mlp16.hip: layer16r_b3, six v_mfma_f32_16x16x32_bf16 per 16 x 16 x 32 block).  A different rounding
of the same f32 products, so the contract is the f32 path's own: pime_ppo_minibatch_grad against PyTorch fp32 autograd of the
reference loss (agent.py:637-655) within 3e-4 of each tensor's largest entry, bitwise repeatable (tests/test_gpu_mlp16.py:
check_grads); the optimizer step fused into the slab reduction leaves the bf16 planes equal to a fresh split of the new parameters;
whole update_net calls against the weights the unmodified reference produced (tests/golden/ppo_update*.npz: width 256 plain and
modular, width 128 modular) at the f32 path's tolerances.
Child processes: both switches are read once per process.  The layer alone against float64: tools/layer16_b3_bench.hip
(profiles/r04_l_layer16_b3_bench.txt: 3.5e-7 of max |y|, the f32 chain 4.0e-7)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r'''
import os, sys
sys.path.insert(0, os.environ["PIME_ROOT"]); sys.path.insert(0, os.path.join(os.environ["PIME_ROOT"], "tests"))
import torch
import test_gpu_mlp16 as t
import test_gpu_ppo_fused as f
from pime_amd import native
lib = native.lib()
for kind, k in (("critic", native.MLP_CRITIC), ("actor", native.MLP_PLAIN_ACTOR), ("modular", native.MLP_MODULAR_ACTOR)):
    plain = lib.pime_ppo_bwd_image_floats(k, 4, 1, 128)
    assert plain > (3 if kind == "modular" else 2) * 128 * 128, (kind, plain)   # the planes sit behind the f32 transposed image
for kind, md, D, B in CASES:
    t.check_grads(kind, md, D, B)
for kind, md, D in (("modular", 128, 3), ("resid", 128, 3), ("resid", 256, 3)):
    f.test_fused_step_keeps_the_packed_images_current(kind, md, D)
    f.test_adam_fused_into_the_slab_reduction_equals_the_separate_step(kind, md, D)
import test_gpu_update_golden as u   # the UNMODIFIED reference's weights after a whole update_net, same tolerances as the f32 path
for tag, mode in (("wts10_256", "one_graph"), ("wtmod256", "two_graph"), ("ph128", "one_graph")):
    u.test_hip_update_net_matches_reference_weights(tag, mode)
print("BF16X3_OK")
'''


def _run(cases, extra_env):
    env = dict(os.environ, PIME_ROOT=ROOT, PIME_MLP16="1", **extra_env)
    r = subprocess.run([sys.executable, "-c", _CHILD.replace("CASES", repr(cases))], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "BF16X3_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_bf16x3_gradients_match_autograd_and_repeat_bitwise():
    _run([("resid", 128, 3, 4096), ("resid", 128, 3, 70000), ("resid", 128, 30, 1000), ("ppo", 128, 4, 777),
          ("modular", 128, 3, 4096), ("modular", 128, 4, 1000), ("modular", 128, 4, 40000),
          ("resid", 256, 30, 4096), ("resid", 256, 3, 777), ("modular", 256, 4, 4096), ("modular", 256, 3, 1000)],
         {"PIME_GRAD_BF16X3": "1"})
