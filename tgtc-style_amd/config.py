"""Command-line / config-file surface of the reference (config.py:5-148) without configargparse.

Format (configs/*.txt): one `key = value` per line, `#` starts a comment (also trailing), a bare `key` sets a
store-true flag; command-line options override the file.  The flag table below reproduces the reference's names
and defaults; flags that only matter to training are accepted and carried, not interpreted.
"""
import argparse

# name -> (type, default); type None = store_true flag.  Defaults: reference config.py:6-145.
FLAGS = {
    "expname": (str, None), "basedir": (str, "./logs/"), "datadir": (str, "./data/"), "styledir": (str, "./style/"),
    "dataset_type": (str, "llff"), "no_ndc": (None, False), "white_bkgd": (None, False), "half_res": (None, False),
    "spherify": (None, False), "decoder_pth_path": (str, "./pretrained/decoder.pth"),
    "vgg_pth_path": (str, "./pretrained/vgg_normalised.pth"), "vae_pth_path": (str, "./pretrained/vae.pth"),
    "factor": (float, 1.0), "gen_factor": (float, 0.2), "valid_factor": (float, 0.05), "num_workers": (int, 0),
    "store_rays": (int, 1), "use_viewdir": (None, False), "sample_type": (str, "uniform"), "act_type": (str, "relu"),
    "nerf_type": (str, "nerf"), "style_type": (str, "mlp"), "latent_type": (str, "variational"),
    "nerf_type_fine": (str, "nerf"), "sigma_noise_std": (float, 1.0), "siren_sigma_mul": (float, 20.0),
    "rgb_loss_lambda": (float, 1.0), "rgb_loss_lambda_2d": (float, 10.0), "style_loss_lambda": (float, 1.0),
    "content_loss_lambda": (float, 1.0), "loss_coh_lambda": (float, 5e3), "logp_loss_lambda": (float, 0.1),
    "logp_loss_decay": (float, 1.0), "lambda_u": (float, 0.01), "netdepth": (int, 8), "netwidth": (int, 256),
    "netdepth_fine": (int, 8), "netwidth_fine": (int, 256), "style_D": (int, 8), "style_feature_dim": (int, 1024),
    "vae_d": (int, 4), "vae_w": (int, 512), "vae_latent": (int, 32), "vae_kl_lambda": (float, 0.1),
    "embed_freq_coor": (int, 10), "embed_freq_dir": (int, 4), "batch_size": (int, 2048),
    "batch_size_style": (int, 1024), "lrate": (float, 5e-4), "lrate_decay": (int, 100000), "chunk": (int, 1024 * 32),
    "no_reload": (None, False), "total_step": (int, 50000001), "origin_step": (int, 250000),
    "decoder_step": (int, 170000), "steps_per_opt": (int, 1), "steps_patch": (int, -1), "N_samples": (int, 64),
    "N_samples_fine": (int, 64), "i_print": (int, 100), "i_weights": (int, 5000), "i_video": (int, 5000000),
    "ckp_num": (int, 3), "render_valid": (None, False), "render_train": (None, False),
    "render_valid_style": (None, False), "render_train_style": (None, False), "sigma_scale": (float, 1.0),
    "pixel_alignment": (None, False), "TT_far": (float, 8.0),
}
# additions of this build (not in the reference)
EXTRA = {
    "precision": (str, "fp16x3"),      # fp16x3 (fp32-equivalent, default) | fp16mx | fp16, or "coarse+fine" e.g. fp16x3+fp16mx
    "shard": (str, "frames"),          # under torchrun: frames = whole images round-robin over the ranks, rank-local files;
                                       # rays = contiguous ray ranges of every image + one all-gather, rank 0 writes
    "literal_batches": (None, False),  # render --batch_size rays per call like the reference; default: whole images per call
                                       # (rays come from the device and the stratified jitter is seeded per ray, so an image
                                       # does not depend on how its rays are batched)
    "synthetic": (None, False),        # no dataset / checkpoints: seeded weights + closed-form camera path
    "synthetic_hw": (int, 400),        # frame size of the synthetic scene
    "synthetic_frames": (int, 2),      # frames of the synthetic validation path
    "latent_seed": (int, -1),          # seed of the latent draw when the table is initialised from the VAE (-1: unseeded, like
                                       # the reference); under torchrun rank 0 draws and broadcasts either way
}


def read_config_file(path):
    """`key = value` lines -> list of argv tokens."""
    argv = []
    with open(path) as f:
        for raw in f:
            line = raw.split("#", 1)[0].strip()
            if not line:
                continue
            if "=" in line:
                k, v = (s.strip() for s in line.split("=", 1))
                argv += ["--" + k, v]
            else:
                argv.append("--" + line)
    return argv


def config_parser():
    p = argparse.ArgumentParser(description="TGTC-Style render CLI on MI355X")
    p.add_argument("--config", type=str, default=None, help="config file path")
    for table in (FLAGS, EXTRA):
        for name, (typ, default) in table.items():
            if typ is None:
                p.add_argument("--" + name, action="store_true", default=default)
            else:
                p.add_argument("--" + name, type=typ, default=default)
    return p


def parse_args(argv=None):
    """File first, command line second (so the command line wins), like configargparse."""
    import sys
    argv = list(sys.argv[1:] if argv is None else argv)
    p = config_parser()
    pre, _ = p.parse_known_args(argv)
    merged = (read_config_file(pre.config) if pre.config else []) + argv
    return p.parse_args(merged)
