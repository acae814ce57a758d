"""Reference-named entry point (tctrans.py): PatchEmbed, decoder, vgg, StyTrans on the HIP library."""
from .style2d import Decoder, PatchEmbed, StyTrans, VGG  # noqa: F401

decoder = Decoder       # the reference exposes module-level Sequential instances; here they are factories
vgg = VGG
