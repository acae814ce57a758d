"""The disk-level driver of the 2-D style pass: reference trans_test.py:55-179 `transformer_render`.

    python -m tgtc_style_amd.trans_test --content_dir logs/<exp>/nerf_gen_data2 --style_dir style/ \
        --output data/<scene>/stylized_gen_4.0 --vgg pretrained/vgg_normalised.pth --save_dir pretrained \
        --decoder_path pretrained/decoder.pth

Loads the four checkpoint layouts the reference writes (SURVEY section 5 / trans_train.py:206-214):
  vgg_normalised.pth              bare state dict of the `vgg` nn.Sequential (only the [:31] prefix is used, :97)
  decoder.pth                     {'decoder': state dict, 'step': n} (:101-103)
  transformer_iter_N.pth          bare state dict, newest file in save_dir whose name contains 'transformer' (:123-129)
  embedding_iter_N.pth            bare state dict, newest file whose name contains 'embedding' (:131-137)
then stylises every rendered frame of `content_dir` (files with 'depth' or 'geometry' in their path are skipped, :82)
with every style image, writes `NNN<ext>` counted from 001 (:166-174) and `stylized_data.npz` with the keys the
reference's dataset reads back (:179, dataset.py:437-440; `read_stylized_data` below is that reader).

The forward pass runs on the HIP kernels (style2d.py); this module is host plumbing: PIL for image files (the
reference uses torchvision's PIL transforms: ToTensor, CenterCrop((h,w)), Resize((512,512))), numpy for the npz.
"""
import argparse
import os
from pathlib import Path

import numpy as np
import torch

from . import style2d


def _to_tensor(img):
    """torchvision.transforms.ToTensor on an RGB PIL image: uint8 HWC -> float32 CHW in [0,1]."""
    a = np.array(img.convert("RGB"), dtype=np.uint8)
    return torch.from_numpy(a).permute(2, 0, 1).to(torch.float32).div(255.0)


def _center_crop(img, h, w):
    """torchvision.transforms.CenterCrop((h, w)) on a PIL image (trans_test.py:30-38): an image smaller than the crop is
    first padded with black, left / top (crop - size) // 2 and right / bottom (crop - size + 1) // 2; the crop window of the
    (padded) image then sits at int(round(.)) offsets."""
    from PIL import Image
    W, H = img.size
    if w > W or h > H:
        pl, pt = ((w - W) // 2 if w > W else 0), ((h - H) // 2 if h > H else 0)
        pr, pb = ((w - W + 1) // 2 if w > W else 0), ((h - H + 1) // 2 if h > H else 0)
        padded = Image.new(img.mode, (W + pl + pr, H + pt + pb), 0)
        padded.paste(img, (pl, pt))
        img, (W, H) = padded, padded.size
    top, left = int(round((H - h) / 2.0)), int(round((W - w) / 2.0))
    return img.crop((left, top, left + w, top + h))


def style_image_array(path):
    """trans_test.py:47-53,144: Resize((512, 512)) (bilinear on PIL images) + ToTensor, as [1,512,512,3] float32."""
    from PIL import Image
    img = Image.open(str(path)).convert("RGB").resize((512, 512), Image.BILINEAR)
    return np.moveaxis(_to_tensor(img).unsqueeze(0).numpy(), 1, -1)


def _newest(save_dir, word):
    files = [os.path.join(save_dir, f) for f in sorted(os.listdir(save_dir)) if word in f]
    if not files:
        raise FileNotFoundError("no '%s' checkpoint in %s" % (word, save_dir))
    return files[-1]


def load_network(vgg, save_dir, decoder_path=None, precision="fp16x3", device="cuda"):
    """trans_test.py:95-139: the four modules from their checkpoint files, eval mode, on the device."""
    enc, dec, tr, emb = style2d.VGG(), style2d.Decoder(), style2d.Transformer(), style2d.PatchEmbed()
    sd = torch.load(vgg, map_location="cpu")
    enc.load_state_dict({k: v for k, v in sd.items() if k in enc.state_dict()})       # the [:31] prefix of `vgg`
    if decoder_path is not None:
        dec.load_state_dict(torch.load(decoder_path, map_location="cpu")["decoder"])
    tr.load_state_dict(torch.load(_newest(save_dir, "transformer"), map_location="cpu"))
    emb.load_state_dict(torch.load(_newest(save_dir, "embedding"), map_location="cpu"))
    for m in (enc, dec, tr, emb):
        m.precision = precision
        m.eval().to(device)
    return style2d.StyTrans(enc, dec, emb, tr)


def transformer_render(content_dir, style_dir, output, save_ext=".jpg", content=None, style=None,
                       vgg="./pretrained/vgg_normalised.pth", save_dir="./pretrained", position_embedding="sine",
                       hidden_dim=512, decoder_path=None, trans_path=None, embedding_path=None, content_size=512,
                       style_size=512, crop=True, preserve_color=True, precision="fp16x3", network=None):
    """Same signature as the reference (unused arguments are carried like there).  `network` lets a caller pass
    already-built modules (tests); otherwise they are loaded from the checkpoint files.  Returns the averaged
    style feature [1,1024]."""
    from PIL import Image
    if not torch.cuda.is_available():
        raise RuntimeError("trans_test: no GPU visible; the 2-D style pass has no CPU fallback")
    if content:
        content_paths = [Path(content)]
    else:
        content_paths = [f for f in sorted(Path(content_dir).glob("*")) if "depth" not in str(f) and "geometry" not in str(f)]
    style_paths = [Path(style)] if style else sorted(Path(style_dir).glob("*"))
    os.makedirs(output, exist_ok=True)
    net = network if network is not None else load_network(vgg, save_dir, decoder_path, precision)

    style_name = {os.path.splitext(os.path.basename(str(style_paths[0])))[0]: 0}
    style_path_str = style_dir if style_dir else os.path.dirname(str(style))
    style_path_for_list = [os.path.join(style_path_str, f) for f in sorted(os.listdir(style_path_str))][0]
    style_img = style_image_array(style_path_for_list)
    rows = np.zeros([1, 1024], dtype=np.float32)

    from .image_writer import writer
    cnt, feats = 0, []
    for content_path in content_paths:
        for style_path in style_paths:
            content_tensor = _to_tensor(Image.open(content_path))
            _, h, w = content_tensor.shape
            style_tensor = _to_tensor(_center_crop(Image.open(style_path).convert("RGB"), h, w))
            with torch.no_grad():
                image, feat, _ = style2d.stylize_frame(net, content_tensor.cuda().unsqueeze(0), style_tensor.cuda().unsqueeze(0))
            cnt += 1
            # torchvision.utils.save_image: x*255 + 0.5, clamped, uint8, HWC; encoded and written in the background
            img8 = image[0].detach().mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8).contiguous()
            writer().save("{:s}/{:03d}{:s}".format(output, cnt, save_ext), img8)
            feats.append(feat.detach().float().reshape(1024))
    writer().drain()
    if feats:
        rows = np.append(rows, torch.stack(feats).cpu().numpy(), axis=0)

    style_feature = np.sum(rows, axis=0, keepdims=True) / (rows.shape[0] - 1)
    np.savez(os.path.join(output, "stylized_data"), style_names=style_name, style_paths=style_path_for_list,
             style_images=style_img, style_features=style_feature)
    return style_feature


def read_stylized_data(data_path, factor):
    """dataset.py:437-440: what the datasets read back from `<data_path>/stylized_gen_<factor>/stylized_data.npz`.
    Returns None when the file does not exist (the reference then leaves the four names undefined)."""
    path = os.path.join(data_path, "stylized_gen_" + str(factor), "stylized_data.npz")
    if not os.path.exists(path):
        return None
    d = np.load(path, allow_pickle=True)
    return {"style_names": d["style_names"][()], "style_paths": d["style_paths"], "style_images": d["style_images"],
            "style_features": d["style_features"], "style_num": int(d["style_images"].shape[0])}


def main(argv=None):
    ap = argparse.ArgumentParser(description="2-D style pass over a directory of rendered frames (reference trans_test.py)")
    ap.add_argument("--content_dir", type=str, default=None)
    ap.add_argument("--content", type=str, default=None)
    ap.add_argument("--style_dir", type=str, default=None)
    ap.add_argument("--style", type=str, default=None)
    ap.add_argument("--output", type=str, required=True)
    ap.add_argument("--save_ext", type=str, default=".jpg")
    ap.add_argument("--vgg", type=str, default="./pretrained/vgg_normalised.pth")
    ap.add_argument("--save_dir", type=str, default="./pretrained")
    ap.add_argument("--decoder_path", type=str, default=None)
    ap.add_argument("--precision", type=str, default="fp16x3")
    a = ap.parse_args(argv)
    feat = transformer_render(a.content_dir, a.style_dir, a.output, save_ext=a.save_ext, content=a.content, style=a.style,
                              vgg=a.vgg, save_dir=a.save_dir, decoder_path=a.decoder_path, precision=a.precision)
    print("wrote", a.output, "style feature", feat.shape)


if __name__ == "__main__":
    main()
