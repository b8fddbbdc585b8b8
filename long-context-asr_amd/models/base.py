"""BaseModel (lcasr/models/base.py:9-68): parameter counting and the decay / no-decay parameter grouping."""
import warnings

import torch


class BaseModel(torch.nn.Module):
    def print_total_params(self, only_trainable=False):
        total = sum(p.numel() for p in self.parameters() if p.requires_grad) if only_trainable else sum(p.numel() for p in self.parameters())
        print(('Total trainable params: ' if only_trainable else 'Total params: ') + ': ', total / 1e6, 'M')
        return total

    @staticmethod
    def create_custom_forward(module):
        def custom_forward(*args, **kwargs):
            return module(*args, **kwargs)
        return custom_forward

    def get_param_groups(self, optim_args):
        """base.py:26-68.  NOTE the reference's quirk is kept: modules in `blacklist_weight_decay_modules` go to
        no_decay and modules in `whitelist_weight_decay_modules` (the norms) go to decay; biases never decay."""
        wd = optim_args.get('weight_decay', 0.0)
        had_b, had_w = hasattr(self, 'blacklist_weight_decay_modules'), hasattr(self, 'whitelist_weight_decay_modules')
        if (not had_b or not had_w) and wd > 0.0:
            warnings.warn('Model does not specify blacklist/whitelist weight-decay modules; decaying all parameters')
            return self.parameters()
        if wd <= 0.0:
            return self.parameters()
        decay, no_decay = set(), set()
        for mn, m in self.named_modules():
            for pn, _p in m.named_parameters():
                fpn = f'{mn}.{pn}' if mn else pn
                if pn.endswith('bias'):
                    no_decay.add(fpn)
                elif isinstance(m, self.blacklist_weight_decay_modules):
                    no_decay.add(fpn)
                elif isinstance(m, self.whitelist_weight_decay_modules):
                    decay.add(fpn)
        param_dict = {pn: p for pn, p in self.named_parameters()}
        inter, union = decay & no_decay, decay | no_decay
        assert len(inter) == 0, f'parameters {inter} made it into both decay/no_decay sets!'
        assert len(param_dict.keys() - union) == 0, f'parameters {param_dict.keys() - union} were not separated into either set!'
        return [
            {'params': [param_dict[pn] for pn in sorted(decay)], 'weight_decay': wd},
            {'params': [param_dict[pn] for pn in sorted(no_decay)], 'weight_decay': 0.0},
        ]
