#!/usr/bin/env python3
"""Development timing: the CLI's --render_valid_style on the synthetic scene (400x400, N frames), wall time per frame
including image files; TGTC_SYNC_IMAGES=1 writes the files synchronously like the reference."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tgtc_style_amd import train_tgtcs
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for rep in range(2):
    with tempfile.TemporaryDirectory() as d:
        t0 = time.time()
        train_tgtcs.main(["--config", os.path.join(root, "configs", "fern.txt"), "--basedir", d, "--synthetic", "--synthetic_hw", "400",
                          "--synthetic_frames", str(frames), "--render_valid_style"])
        torch.cuda.synchronize()
        dt = time.time() - t0
        print("rep %d: %d frames in %.2f s = %.1f ms/frame (sync images: %s)" % (rep, frames, dt, dt / frames * 1e3, os.environ.get("TGTC_SYNC_IMAGES", "0")), flush=True)
