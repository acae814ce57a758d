"""The reference's checkpoint files, written and found the way it does.

  <ckpts>/NNNNNN.tar          {'global_step', 'model', 'model_fine', 'optimizer', 'style_optimizer'}   train_tgtcs.py:285-300
  <ckpts>/style_NNNNNN.tar    {'global_step', 'model' (style MLP), 'concat_model', 'optimizer'}        train_tgtcs.py:503-510
  <ckpts>/latent_NNNNNN.tar   {'global_step', 'train_set_1'}                                           train_tgtcs.py:512-517
  <save_dir>/transformer_iter_N.pth   bare state dict                                                  trans_train.py:206-207
  <save_dir>/decoder_iter_N.pth       {'decoder': state dict, 'step': N}                               trans_train.py:209-211
  <save_dir>/embedding_iter_N.pth     bare state dict                                                  trans_train.py:213-214

Selection on reload is the reference's: sorted file names, filtered by substring, the last one wins (train_tgtcs.py:60-82,
:139-146; trans_test.py:123-137); NeRF checkpoints are rotated so that at most `keep` .tar files of any kind remain
(train_tgtcs.py:303-305 removes the first of the sorted list).  Pure host code (torch.save / torch.load)."""
import os

import torch


def _cpu(sd):
    return {k: (v.detach().cpu() if isinstance(v, torch.Tensor) else v) for k, v in sd.items()}


def newest(path, want=(), reject=(), suffix="tar"):
    """The reference's pick: sorted listing, every `want` substring present, no `reject` substring, last entry."""
    if not os.path.isdir(path):
        return None
    files = [f for f in sorted(os.listdir(path)) if suffix in f and all(w in f for w in want) and not any(r in f for r in reject)]
    return os.path.join(path, files[-1]) if files else None


def save_nerf(ckpts_path, global_step, model, model_fine=None, optimizer=None, style_optimizer=None, keep=None):
    os.makedirs(ckpts_path, exist_ok=True)
    path = os.path.join(ckpts_path, '{:06d}.tar'.format(global_step))
    d = {'global_step': global_step, 'model': _cpu(model.state_dict())}
    if model_fine is not None:
        d['model_fine'] = _cpu(model_fine.state_dict())
    d['optimizer'] = optimizer.state_dict() if optimizer is not None else {}
    d['style_optimizer'] = style_optimizer.state_dict() if style_optimizer is not None else {}
    torch.save(d, path)
    if keep is not None:
        ckpts = [os.path.join(ckpts_path, f) for f in sorted(os.listdir(ckpts_path)) if 'tar' in f]
        if len(ckpts) > keep:
            os.remove(ckpts[0])
    return path


def save_style(ckpts_path, global_step, style_model, concat_model, optimizer=None):
    os.makedirs(ckpts_path, exist_ok=True)
    path = os.path.join(ckpts_path, 'style_{:06d}.tar'.format(global_step))
    torch.save({'global_step': global_step, 'model': _cpu(style_model.state_dict()), 'concat_model': _cpu(concat_model.state_dict()),
                'optimizer': optimizer.state_dict() if optimizer is not None else {}}, path)
    return path


def save_latents(ckpts_path, global_step, latents_model):
    os.makedirs(ckpts_path, exist_ok=True)
    path = os.path.join(ckpts_path, 'latent_{:06d}.tar'.format(global_step))
    torch.save({'global_step': global_step, 'train_set_1': _cpu(latents_model.state_dict())}, path)
    return path


def save_style2d(save_dir, step, transformer, decoder, embedding):
    """The three files of the 2-D module (the `new_ps` convolution of the transformer is part of its state dict, as in
    the reference)."""
    os.makedirs(save_dir, exist_ok=True)
    torch.save(_cpu(transformer.state_dict()), '{:s}/transformer_iter_{:d}.pth'.format(save_dir, step))
    torch.save({'decoder': _cpu(decoder.state_dict()), 'step': step}, '{:s}/decoder_iter_{:d}.pth'.format(save_dir, step))
    torch.save(_cpu(embedding.state_dict()), '{:s}/embedding_iter_{:d}.pth'.format(save_dir, step))


def load_nerf(ckpts_path, model, model_fine=None):
    """-> global_step or None (train_tgtcs.py:60-72)."""
    ck = newest(ckpts_path, ['tar'], ['style', 'latent'])
    if ck is None:
        return None
    sd = torch.load(ck, map_location='cpu')
    model.load_state_dict(sd['model'])
    if model_fine is not None:
        model_fine.load_state_dict(sd['model_fine'])
    return sd['global_step']


def load_style(ckpts_path, style_model, concat_model):
    """-> global_step or None (train_tgtcs.py:74-82)."""
    ck = newest(ckpts_path, ['tar', 'style'], ['latent'])
    if ck is None:
        return None
    sd = torch.load(ck, map_location='cpu')
    style_model.load_state_dict(sd['model'])
    concat_model.load_state_dict(sd['concat_model'])
    return sd['global_step']


def load_latents(ckpts_path, latents_model):
    """-> True if a latent checkpoint was found (train_tgtcs.py:139-146)."""
    ck = newest(ckpts_path, ['tar', 'latent'], ['style'])
    if ck is None:
        return False
    latents_model.load_state_dict(torch.load(ck, map_location='cpu')['train_set_1'])
    return True
