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
