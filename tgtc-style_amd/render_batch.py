"""Batch render of several scene configs in one job (BASELINE config 5: "all 5 LLFF configs (fern/flower/horns/orchids/
trex) batch render, 8 x MI355X"):

    torchrun --nproc-per-node 8 -m tgtc_style_amd.render_batch --configs configs/fern.txt configs/flower.txt \
        configs/horns.txt configs/orchids.txt configs/trex.txt -- --render_valid_style [--shard frames] [--precision fp16]

Everything after `--` is passed to every scene's `train_tgtcs` invocation.  One process per GPU for the whole batch:
the process group and the HIP context come up once, each scene's frames are dealt round-robin over the ranks
(`--shard frames`, no collective on the data path: SURVEY section 8e) and every rank writes its own files, so the job is
N independent streams of frames that only meet at the barriers between scenes.  The reference has no such driver (its
configs are run one by one, train_tgtcs.py:596-597); this is the scene loop around the same CLI.
"""
import argparse
import sys

from . import config as cfg
from . import train_tgtcs


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    rest = []
    if "--" in argv:
        i = argv.index("--")
        argv, rest = argv[:i], argv[i + 1:]
    ap = argparse.ArgumentParser(description="render several scene configs in one job")
    ap.add_argument("--configs", nargs="+", required=True)
    a = ap.parse_args(argv)
    outputs = []
    try:
        for path in a.configs:
            args = cfg.parse_args(["--config", path] + rest)
            if args.expname is None:
                raise SystemExit("render_batch: %s names no expname" % path)
            outputs.append(train_tgtcs.train(args))
    finally:
        train_tgtcs._finish_distributed()
    return outputs


if __name__ == "__main__":
    main()
