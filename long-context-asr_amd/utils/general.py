"""Checkpoint I/O of the training driver — mirror of lcasr/utils/general.py:97-172 (save_model, find_latest_checkpoint,
load_checkpoint): same arguments, same file naming (`step_<podcast_step>.pt`), same dictionary keys, so a checkpoint written
by either code base loads in the other (the model state_dict keys and the MADGRAD state layout are those of the reference).

One deliberate difference: files are read with `torch.load(weights_only=True)` first — a checkpoint is a pickle, and one that
came from somewhere else must not execute code on load.  Reference checkpoints whose 'config' entry is an OmegaConf object
need `trust_pickle=True` (the reference's own behaviour) after the user has decided to trust the file."""
from __future__ import annotations

import os
import warnings
from typing import Dict, List, Tuple

import torch


def save_model(model: torch.nn.Module, optimizer: torch.optim.Optimizer, scheduler, podcast_step: int, config: Dict,
               sequence_scheduler=None, seen_ids: List[int] = [], epoch: int = 0, other: Dict = {}):
    save_path = os.path.join(config['checkpointing']['dir'], f'step_{podcast_step}.pt')
    save_dict = {
        'model': model.state_dict(),
        'optimizer': optimizer.state_dict(),
        'scheduler': scheduler.state_dict() if scheduler is not None else None,
        'podcast_step': podcast_step,
        'config': config,
        'sequence_scheduler': sequence_scheduler.state_dict() if sequence_scheduler is not None else None,
        'seen_ids': seen_ids,
        'epoch': epoch,
        **other,
    }
    torch.save(save_dict, save_path)
    return save_path


def find_latest_checkpoint(path: str = './checkpoints'):
    names = [n for n in os.listdir(path) if n.endswith('.pt')]
    if not names:
        return None
    return max(names, key=lambda n: int(n.split('_')[1].split('.')[0]))


def load_checkpoint(args, model, optimizer=None, scheduler=None, sequence_scheduler=None, path='./checkpoints', device='cpu',
                    other: List[Tuple] = [], trust_pickle: bool = False):
    """Resume from the newest `step_*.pt` under `path`.  Returns (seen_ids, step, epoch); ([], 0, 0) without a checkpoint."""
    latest = find_latest_checkpoint(path)
    if latest is None:
        return [], 0, 0
    path = os.path.join(path, latest)
    try:
        checkpoint = torch.load(path, map_location=device, weights_only=True)
    except Exception as e:
        if not trust_pickle:
            raise RuntimeError(f'{path} holds objects torch.load(weights_only=True) refuses ({type(e).__name__}: {e}). '
                               f'If you trust the file, call load_checkpoint(..., trust_pickle=True).') from e
        checkpoint = torch.load(path, map_location=device, weights_only=False)
    if args and getattr(args, 'remove_scheduler', False):
        checkpoint['scheduler'] = None
        checkpoint['sequence_scheduler'] = None
    try:
        model.load_state_dict(checkpoint['model'])
    except Exception:
        warnings.warn('loading model with strict=False')
        model.load_state_dict(checkpoint['model'], strict=False)
        warnings.warn('SETTING OPTIMIZER TO NONE DUE TO NON-STRICT LOAD')
        optimizer = None
    if optimizer is not None and checkpoint.get('optimizer') is not None:
        optimizer.load_state_dict(checkpoint['optimizer'])
    if scheduler is not None and checkpoint.get('scheduler') is not None:
        scheduler.load_state_dict(checkpoint['scheduler'])
    if sequence_scheduler is not None and checkpoint.get('sequence_scheduler') is not None:
        sequence_scheduler.load_state_dict(checkpoint['sequence_scheduler'])
    for obj, key in other:
        if key in checkpoint:
            obj.load_state_dict(checkpoint[key])
        else:
            warnings.warn(f'Could not find {key} in checkpoint, skipping')
    return checkpoint.get('seen_ids', []), checkpoint.get('podcast_step', 0), checkpoint.get('epoch', 0)


# ---- construction seam (lcasr/utils/general.py:24-95): config -> model class -> model -> optimiser + LR schedule ---------------
def _get(cfg, key, default=None):
    """Configs are dicts here and OmegaConf nodes (attribute access) in the reference: read either."""
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return getattr(cfg, key, default) if not hasattr(cfg, 'get') else cfg.get(key, default)


def get_model_class(config: Dict = {}, args={}):
    """general.py:24-53.  Only the SConformerXL family is built on the MI355X path; the reference's other model classes
    (Mamba, EncDec*, SCConformerMeta) are out of scope and refused by name rather than silently replaced."""
    known = ['SCConformerXL', 'Mamba', 'EncDecSconformer', 'EncDecSconformerV2', 'SCConformerMeta']
    arg_cls = getattr(args, 'model_class', None) if not isinstance(args, dict) else args.get('model_class')
    if arg_cls is not None:
        name = arg_cls
    elif _get(config, 'model_class') is not None:
        name = _get(config, 'model_class')
    else:
        warnings.warn('No model_class specified in model config or args, defaulting to SCConformerXL')
        name = 'SCConformerXL'
    assert name in known, f'Unknown model class {name}, must be one of {known}'
    if name != 'SCConformerXL':
        raise NotImplementedError(f'model class {name} is not part of the MI355X hot path (only SCConformerXL is)')
    from ..models.sconformer_xl import SCConformerXL
    return SCConformerXL


def load_model(config: Dict, vocab_size: int, model_class=None):
    """general.py:57-59: model_class(**config.model, vocab_size=vocab_size)."""
    if model_class is None:
        from ..models.sconformer_xl import SCConformerXL as model_class
    return model_class(**dict(_get(config, 'model')), vocab_size=vocab_size)


def load_optimizer(config: Dict, model: torch.nn.Module):
    """general.py:61-95: optimiser from config['optimizer'] (name, args, weight_decay_groups) + the cosine LR schedule from
    config['scheduler']['warmup_steps'].  'madgrad' is the fused flat-buffer MADGRAD (the reference's default and the only one
    its paper configs use); 'adam' maps to torch.optim.Adam, which accepts the HIP path's gradients as they are.
    Call it AFTER model.cuda(): the fused optimiser flattens the parameters where they live."""
    from ..optim import MADGRAD
    from .scheduling import CosineLRScheduler
    opt = _get(config, 'optimizer')
    optim_type = _get(opt, 'name')
    allowed = ['adam', 'madgrad', 'mirrormadgrad']
    assert optim_type in allowed, f'Unknown optimizer {optim_type}, must be one of {allowed}'
    optim_args = dict(_get(opt, 'args'))
    groups_mode = _get(opt, 'weight_decay_groups', 'default')
    if groups_mode == 'default':
        param_groups = model.get_param_groups(optim_args) if hasattr(model, 'get_param_groups') else model.parameters()
    elif groups_mode == 'none':
        param_groups = model.parameters()
    else:
        raise NotImplementedError(f'Unknown weight_decay_groups {groups_mode}, must be one of [default, none]')
    if optim_type == 'madgrad':
        optimizer = MADGRAD(param_groups, **optim_args)
    elif optim_type == 'adam':
        optimizer = torch.optim.Adam(param_groups, **optim_args)
    else:
        raise NotImplementedError('MirrorMADGRAD is not used by any SConformerXL config and is not implemented')
    scheduler = CosineLRScheduler(optimizer=optimizer, warmup_steps=_get(_get(config, 'scheduler'), 'warmup_steps'),
                                  peak_value=optim_args['lr'], final_value=0.0)
    return optimizer, scheduler


def avg_all_models_in_dir(path: str, out_path: str, model_name: str = 'step_105360.pt', trust_pickle: bool = False):
    """general.py:175-194: uniform average of the 'model' state_dicts of <path>/*/<model_name>; the other entries of the first
    checkpoint (minus optimizer / scheduler) are carried over.  Files are read with weights_only=True unless trust_pickle."""
    folders = [el for el in sorted(os.listdir(path)) if os.path.exists(os.path.join(path, el, model_name))]
    total = len(folders)
    if total == 0:
        raise FileNotFoundError(f'no {model_name} under {path}/*/')
    avg, carried = None, None
    for folder in folders:
        ckpt = torch.load(os.path.join(path, folder, model_name), map_location='cpu', weights_only=not trust_pickle)
        sd = ckpt['model']
        if avg is None:
            avg = {k: sd[k] * (1 / total) for k in sd}
            carried = {k: ckpt[k] for k in ckpt if k not in ('model', 'optimizer', 'scheduler')}
        else:
            for k in sd:
                avg[k] += sd[k] * (1 / total)
    carried['model'] = avg
    torch.save(carried, out_path)
    return out_path
