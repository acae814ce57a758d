"""Image side of the LLFF loader (reference load_llff.py:9-111 `_minify`, `_load_data`).

What the reference does before any ray exists, on the host:
  * `_load_data(basedir, factor)`: make sure `<basedir>/images_<factor>/` exists (`_minify`), list its jpg / png files
    in sorted order, check their number against `poses_bounds.npy`, overwrite the pose table's (H, W) with the
    downsized image's shape and divide its focal length by `factor` (load_llff.py:95-97), read every image as
    `imageio.imread(f)[..., :3] / 255.` and stack them on the LAST axis (:108-109); `load_llff_data` then moves that
    axis to the front (:242).
  * `_minify` shells out to ImageMagick (`mogrify -resize 25% -format png`, :39-58).  ImageMagick is not part of this
    build: the replacement resizes with PIL (Lanczos, the filter family ImageMagick uses when shrinking) and writes
    PNGs under the same directory name, so a scene prepared by the reference and a scene prepared here are
    interchangeable on disk; the pixels of a freshly minified scene differ in rounding, nothing downstream depends on
    which tool made the cache.

Pure host code (PIL + numpy), like the reference.
"""
import os

import numpy as np

_EXT = ("JPG", "jpg", "png", "jpeg", "PNG")


def _image_files(directory, exts=("JPG", "jpg", "png")):
    return [os.path.join(directory, f) for f in sorted(os.listdir(directory)) if f.endswith(exts)]


def minify(basedir, factor):
    """load_llff.py:9-58 for one integer / float factor: `<basedir>/images_<factor>/*.png`, made once."""
    from PIL import Image
    target = os.path.join(basedir, "images_{}".format(factor))
    if os.path.exists(target):
        return target
    src = os.path.join(basedir, "images")
    files = [os.path.join(src, f) for f in sorted(os.listdir(src)) if f.endswith(_EXT)]
    os.makedirs(target)
    for f in files:
        img = Image.open(f).convert("RGB")
        w, h = img.size
        # mogrify -resize P% rounds each dimension to the nearest pixel
        size = (max(1, int(round(w / factor))), max(1, int(round(h / factor))))
        img.resize(size, Image.LANCZOS).save(os.path.join(target, os.path.splitext(os.path.basename(f))[0] + ".png"))
    return target


def load_data(basedir, factor=None, load_imgs=True):
    """load_llff.py:63-111 with `factor` (the only variant the datasets use, dataset.py:69).  Returns
    (poses [3,5,N], bds [2,N], imgs [H,W,3,N] float64 in [0,1]) exactly like the reference -- or (poses, bds)."""
    from PIL import Image
    arr = np.load(os.path.join(basedir, "poses_bounds.npy"))
    poses = arr[:, :-2].reshape([-1, 3, 5]).transpose([1, 2, 0])
    bds = arr[:, -2:].transpose([1, 0])
    sfx = ""
    if factor is not None:
        sfx = "_{}".format(factor)
        minify(basedir, factor)
    else:
        factor = 1
    imgdir = os.path.join(basedir, "images" + sfx)
    if not os.path.exists(imgdir):
        raise FileNotFoundError(imgdir + " does not exist")
    files = _image_files(imgdir)
    if poses.shape[-1] != len(files):
        raise ValueError("Mismatch between imgs {} and poses {}".format(len(files), poses.shape[-1]))
    first = np.asarray(Image.open(files[0]))
    poses[:2, 4, :] = np.array(first.shape[:2]).reshape([2, 1])
    poses[2, 4, :] = poses[2, 4, :] * 1. / factor
    if not load_imgs:
        return poses, bds
    imgs = np.stack([np.asarray(Image.open(f))[..., :3] / 255. for f in files], -1)
    return poses, bds, imgs


def load_images(basedir, factor):
    """The `images` array of load_llff_data (load_llff.py:242,303): [N,H,W,3] float32."""
    _, _, imgs = load_data(basedir, factor)
    return np.moveaxis(imgs, -1, 0).astype(np.float32)
