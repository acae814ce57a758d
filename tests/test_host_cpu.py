"""CPU tests of the host side: config-file parser (reference format and defaults), the C-ABI library exporting
every declared symbol (no compute), workspace-size queries and error paths that need no GPU."""
import ctypes
import os
import re

import numpy as np

import pytest

from tgtc_style_amd import config as cfg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config_file_and_defaults(tmp_path):
    args = cfg.parse_args(["--config", os.path.join(ROOT, "configs", "fern.txt")])
    # values set by configs/fern.txt (reference configs/fern.txt:1-27)
    assert args.expname == "fern_style" and args.dataset_type == "llff" and args.factor == 4.0
    assert args.N_samples == 64 and args.N_samples_fine == 64 and args.batch_size == 2048 and args.style_D == 8
    assert args.use_viewdir is True and args.origin_step == 120001 and args.loss_coh_lambda == 100.0
    # untouched defaults (reference config.py:70-118,139)
    assert args.chunk == 32768 and args.netdepth == 8 and args.netwidth == 256 and args.embed_freq_coor == 10
    assert args.embed_freq_dir == 4 and args.vae_latent == 32 and args.sigma_scale == 1.0 and args.act_type == "relu"
    assert args.render_valid_style is False and args.no_reload is False
    # command line wins over the file; comments and bare flags parse
    p = tmp_path / "c.txt"
    p.write_text("expname = x  # trailing comment\n# full comment\nN_samples = 128\nno_ndc\nchunk = 1024\n")
    a = cfg.parse_args(["--config", str(p), "--N_samples", "32", "--render_valid_style", "--chunk", "2048"])
    assert a.expname == "x" and a.N_samples == 32 and a.no_ndc and a.render_valid_style and a.chunk == 2048


def test_library_exports_every_header_symbol():
    from tgtc_style_amd import fused_train, hip, style2d  # noqa: F401  (style2d / fused_train register their headers' symbols)
    assert os.path.exists(hip.LIB_PATH), "run __graft_entry__.build() first"
    assert hip.missing_symbols() == []
    lib = ctypes.CDLL(hip.LIB_PATH)
    declared = set()
    for header in ("tgtc_hip.h", "tgtc_style2d.h", "tgtc_train.h"):
        text = open(os.path.join(ROOT, "include", header)).read()
        declared |= set(re.findall(r"\b(tgtc_[a-z0-9_]+)\s*\(", text))
    declared -= {"tgtc_linear", "tgtc_named_tensor"}
    assert len(declared) >= 40
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert set(hip.header_symbols()) == declared


def test_no_gpu_calls_fail_loudly_or_are_pure():
    import torch
    from tgtc_style_amd import hip
    lib = hip.load()
    assert lib.tgtc_version() >= 100
    assert lib.tgtc_render_workspace_bytes(160000, 128, 64) >= 160000 * (128 * 6 + 192 * 5) * 4
    assert lib.tgtc_s2d_transformer_workspace_bytes(2500, 2500) > 8 * 2500 * 2500 * 4
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no GPU"):
            hip.nerf_create({}, "fp16x3")
        from tgtc_style_amd import utils
        with pytest.raises(RuntimeError):
            utils.alpha_composition(torch.zeros(2, 4, 3), torch.zeros(2, 4), torch.zeros(2, 4))


def test_llff_poses_golden(golden):
    """llff_poses (host numpy, like the reference) against the reference's own load_llff_data run on a synthetic scene
    directory (g11): recentred poses, rescaled bounds, the 120-view spiral, hold-out index, cps_valid."""
    from tgtc_style_amd import llff_poses
    g = golden("g11_llff_poses")
    out = llff_poses.scene_poses(g["poses_arr"], tuple(int(v) for v in g["image_hw"]), factor=int(g["factor"]))
    assert out["i_test"] == int(g["i_test"])
    for k in ("poses", "bds", "render_poses"):
        assert out[k].dtype == np.float32 and out[k].shape == g[k].shape, k
        assert np.abs(out[k] - g[k]).max() <= 2e-6 * max(1.0, float(np.abs(g[k]).max())), k
    cps = llff_poses.valid_camera_poses(out["render_poses"])
    assert cps.shape == (120, 4, 4) and np.abs(cps - g["cps_valid"]).max() <= 2e-6 * float(np.abs(g["cps_valid"]).max())
    # the spiral is closed and looks at one focus point: consecutive views differ smoothly, first != last
    d = np.linalg.norm(np.diff(out["render_poses"][:, :3, 3], axis=0), axis=1)
    assert d.max() < 4 * np.median(d) and np.linalg.norm(out["render_poses"][0, :3, 3] - out["render_poses"][-1, :3, 3]) > 0


def test_all_five_scene_configs_parse_and_name_their_outputs():
    """configs/*.txt (BASELINE config 5's five scenes) in the reference's format: only the per-scene keys differ."""
    import os
    from tgtc_style_amd import config
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs")
    seen = {}
    for scene in ("fern", "flower", "horns", "orchids", "trex"):
        a = config.parse_args(["--config", os.path.join(root, scene + ".txt")])
        assert a.expname == scene + "_style" and a.datadir == "./data/" + scene
        assert a.N_samples == 64 and a.N_samples_fine == 64 and a.use_viewdir and a.factor == 4
        seen[scene] = (a.loss_coh_lambda, a.valid_factor, a.total_step)
    assert seen == {"fern": (1e2, 3, 128001), "flower": (1e2, 3, 128000), "horns": (1e2, 2, 128001),
                    "orchids": (5e2, 3, 128001), "trex": (1e2, 2, 128001)}


@pytest.mark.parametrize("tool,args,inc", [("gen_mx_asm.py", ["nerf"], "mx_asm_nerf.inc"), ("gen_mx2_asm.py", [], "mx2_asm_nerf.inc"),
                                            ("gen_x3_asm.py", [], "x3_asm_nerf.inc")])
def test_generated_instruction_streams_are_current(tool, args, inc):
    """The committed csrc/*.inc files are exactly what tools/gen_*_asm.py emits today (the kernels' static_asserts pin the ring
    layout they were generated for; this pins the text: a generator edit without a regenerate, or the reverse, fails here)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", tool)] + args, capture_output=True, text=True, check=True).stdout
    with open(os.path.join(root, "tgtc-style_amd", "csrc", inc)) as f:
        committed = f.read()
    assert out == committed, "%s is stale: python tools/%s %s > tgtc-style_amd/csrc/%s" % (inc, tool, " ".join(args), inc)
    # every MFMA of a stream names an accumulator the stream owns, and the stream ends with the LDS queue drained
    assert out.count("v_mfma_") > 1000 and "s_waitcnt lgkmcnt(0)" in out
