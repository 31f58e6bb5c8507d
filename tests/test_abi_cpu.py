"""CPU-side checks of the C-ABI boundary: the library loads, exports every function include/pime_hip.h declares,
the binding covers them all, and the product path refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "pime_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pime_[a-z0-9_]+)\s*\(", src)))


def test_header_library_binding_agree():
    import pime_amd.native as nt
    declared = _declared_functions()
    assert len(declared) >= 20
    lib = C.CDLL(nt.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in pime_hip.h but not exported by libpime_hip.so"
    assert sorted(nt.EXPORTS) == declared, "ctypes binding and header disagree"
    header_version = int(re.search(r"#define PIME_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "pime_hip.h")).read()).group(1))
    assert nt.lib().pime_abi_version() == nt.ABI_VERSION == header_version


def test_cfg_struct_layout_matches_c():
    """pime_env_cfg_default round-trips through the ctypes struct: catches field-order/padding drift."""
    import pime_amd.native as nt
    cfg = nt.EnvCfg()
    nt.check(nt.lib().pime_env_cfg_default(nt.ENV_PH, C.byref(cfg)))
    assert (cfg.kind, cfg.max_steps, cfg.reward_type, cfg.integral_bound, cfg.resample_every) == (0, 50, 1, 1, 1)
    assert (cfg.range_lo[0], cfg.range_hi[0], cfg.range_lo[1], cfg.range_hi[1]) == (0.005, 0.015, 0.0015, 0.0025)
    assert (cfg.init_lo[0], cfg.init_hi[0], cfg.init_lo[1], cfg.init_hi[1]) == (0.0, 50.0, 3.0, 11.0)
    assert (cfg.ph_sample_t, cfg.ph_u_high, cfg.ph_table_scale, cfg.integral_max) == (20.0, 1.5, 1e5, 25.0)
    nt.check(nt.lib().pime_env_cfg_default(nt.ENV_WT, C.byref(cfg)))
    assert (cfg.max_steps, cfg.wt_n_discrete, cfg.wt_G, cfg.wt_dt, cfg.wt_noise_scale, cfg.wt_pmax) == \
        (200, 20, 980.0, 0.1, 0.01, 10.0)
    assert (cfg.range_lo[2], cfg.range_hi[2]) == (0.07, 0.17)
    assert nt.lib().pime_env_cfg_default(7, C.byref(cfg)) != 0 and "kind" in nt.last_error()


def test_struct_layouts_match_the_header(tmp_path):
    """sizeof / offsetof of every ABI struct as gcc sees include/pime_hip.h against the ctypes mirrors in native.py."""
    import subprocess
    import pime_amd.native as nt
    structs = {"pime_env_cfg": nt.EnvCfg, "pime_ph_chem": nt.PhChem, "pime_ppo_net": nt.PpoNet, "pime_ppo_batch": nt.PpoBatch,
               "pime_adam": nt.Adam, "pime_td3_net": nt.Td3Net, "pime_td3_batch": nt.Td3Batch}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "pime_hip.h"', 'int main(void) {']
    for cname, cls in structs.items():
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["gcc", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)])
    got = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for cname, cls in structs.items():
        assert int(got[cname]) == C.sizeof(cls), f"sizeof({cname})"
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, f"offsetof({cname}, {fname})"


def test_native_table_matches_oracle_bitwise(ph_table_oracle):
    import pime_amd.native as nt
    np.testing.assert_array_equal(nt.ph_table_build(), ph_table_oracle)


def test_mlp_pack_size_and_argument_errors():
    import pime_amd.native as nt
    L = nt.lib()
    # critic md 128, D 3: 128*4 + 2*(16384 + 128) + 128 + 4
    assert L.pime_mlp_packed_floats(nt.MLP_CRITIC, 3, 0, 128) == 128 * 4 + 2 * (128 * 128 + 128) + 128 + 4
    assert L.pime_mlp_packed_floats(nt.MLP_MODULAR_ACTOR, 3, 1, 128) * 4 < 160 * 1024
    # width 256: the streamed 16-tile family (csrc/mlp16.hip): first-layer image 1 k-step x 16 tiles x 64, three biases,
    # two 256 x 256 images, head weights, 4-float head-bias slot
    assert L.pime_mlp_packed_floats(nt.MLP_CRITIC, 3, 0, 256) == 16 * 64 + 3 * 256 + 2 * 256 * 256 + 256 + 4
    # the modular actor at width 256 (round 3: mlp16m kernels): two towers (first-layer image, bias, md -> md/2 image, bias) + net.0
    # image and bias + head weights + head-bias slot
    assert L.pime_mlp_packed_floats(nt.MLP_MODULAR_ACTOR, 3, 1, 256) == 2 * (16 * 64 + 256 + 256 * 128 + 128) + 256 * 256 + 256 + 256 + 4
    assert L.pime_mlp_packed_floats(nt.MLP_MODULAR_ACTOR, 3, 1, 512) == 0 and "512" in nt.last_error()
    assert L.pime_mlp_packed_floats(nt.MLP_CRITIC, 3, 0, 192) == 0 and "192" in nt.last_error()
    assert L.pime_mlp_packed_floats(nt.MLP_MODULAR_ACTOR, 3, 3, 128) == 0


def test_no_cpu_fallback():
    """Without a GPU the product path must fail loudly, not compute on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import pime_amd.native as nt
    from pime_amd.vec_env import VecPH
    assert nt.device_count() == 0
    with pytest.raises(nt.PimeError):
        VecPH(4, device="cpu")
    cfg = nt.EnvCfg()
    nt.check(nt.lib().pime_env_cfg_default(nt.ENV_WT, C.byref(cfg)))
    assert not nt.lib().pime_env_create(C.byref(cfg))
    assert "no HIP device" in nt.last_error() or "fallback" in nt.last_error()


def test_capture_flag_entry_points_work_without_a_gpu():
    """pime_capture_begin / _end / _leave only move a counter and drain an (empty) queue: callable on a host without a device."""
    import pime_amd.native as nt
    L = nt.lib()
    assert L.pime_deferred_releases() == 0
    L.pime_capture_begin(); L.pime_capture_begin(); L.pime_capture_leave(); L.pime_capture_end()
    with nt.capture_guard():
        assert L.pime_deferred_releases() == 0
    assert L.pime_deferred_releases() == 0


def test_bf16x3_switch_only_grows_the_transposed_image_of_the_16_tile_family():
    """PIME_GRAD_BF16X3 is read once per process by the library: child processes.  Without it pime_ppo_bwd_image_f32_floats ==
    pime_ppo_bwd_image_floats everywhere; with it the nets the 16-tile family serves at width 128 (PIME_MLP16=1) / 256 carry their
    chain layers' three bf16 planes behind the f32 image -- 1.5x the f32 floats per layer and direction (csrc/mlp16.hip:
    layout_b3_16 / layout_b3_16m) -- and nothing else changes size (host-side size functions: no GPU needed)."""
    import json
    import subprocess
    import sys
    child = r'''
import json, os, sys
sys.path.insert(0, os.environ["PIME_ROOT"])
from pime_amd import native as nt
L = nt.lib()
out = {}
for name, kind, D, Di, md in (("cri128", nt.MLP_CRITIC, 3, 0, 128), ("mod128", nt.MLP_MODULAR_ACTOR, 4, 1, 128), ("cri256", nt.MLP_CRITIC, 30, 0, 256),
                              ("mod256", nt.MLP_MODULAR_ACTOR, 4, 1, 256), ("cri64", nt.MLP_CRITIC, 3, 0, 64)):
    out[name] = [L.pime_ppo_bwd_image_f32_floats(kind, D, Di, md), L.pime_ppo_bwd_image_floats(kind, D, Di, md), L.pime_ppo_fwd_image_floats(kind, D, Di, md)]
print("SIZES " + json.dumps(out))
'''
    res = {}
    for tag, extra in (("off", {}), ("mlp16", {"PIME_MLP16": "1"}), ("on", {"PIME_MLP16": "1", "PIME_GRAD_BF16X3": "1"}),
                       ("on_default_family", {"PIME_GRAD_BF16X3": "1"})):
        env = {k: v for k, v in os.environ.items() if k not in ("PIME_MLP16", "PIME_GRAD_BF16X3")}
        env.update(PIME_ROOT=ROOT, **extra)
        r = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True, timeout=300)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("SIZES ")]
        assert r.returncode == 0 and line, r.stdout[-1000:] + r.stderr[-2000:]
        res[tag] = json.loads(line[-1][6:])
    for tag in ("off", "mlp16"):
        for name, (f32, total, _) in res[tag].items():
            assert f32 == total > 0, (tag, name)
    planes = lambda md, layers_sq, layers_half: (md // 32) * (md // 16) * 768 * layers_sq + (md // 32) * (md // 32) * 768 * layers_half
    on, base = res["on"], res["mlp16"]
    assert on["cri128"] == [base["cri128"][0], base["cri128"][1] + planes(128, 4, 0), base["cri128"][2]]
    assert on["mod128"] == [base["mod128"][0], base["mod128"][1] + planes(128, 2, 4), base["mod128"][2]]
    assert on["cri256"] == [base["cri256"][0], base["cri256"][1] + planes(256, 4, 0), base["cri256"][2]]
    assert on["mod256"] == [base["mod256"][0], base["mod256"][1] + planes(256, 2, 4), base["mod256"][2]]
    assert on["cri64"] == base["cri64"]                                  # width 64: no variant
    d = res["on_default_family"]                                          # without PIME_MLP16 width 128 stays on the LDS-resident kernels
    assert d["cri128"] == res["off"]["cri128"] and d["mod128"] == res["off"]["mod128"] and d["cri256"] == on["cri256"]
