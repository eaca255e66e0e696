"""`DDPM`: the model wrapper the reference's drivers use (FastDiffSR/model/model.py, base_model.py,
model/__init__.py:create_model), over the HIP-backed netG: sampling (test / get_current_visuals), the
training step (optimize_parameters: forward, loss / (b*c*h*w), backward and Adam all on the device) and
the generator + optimiser checkpoints in the reference's own file formats."""
import logging
import os
from collections import OrderedDict

import torch
import torch.nn as nn

from . import networks

logger = logging.getLogger('base')


class DDPM:
    def __init__(self, opt):
        self.opt = opt
        if opt['gpu_ids'] is None:
            raise RuntimeError('fastdiffsr_amd runs on the GPU only (gpu_ids is null)')
        self.device = torch.device('cuda')                         # base_model.py:12-14
        self.begin_step = 0
        self.begin_epoch = 0
        self.netG = self.set_device(networks.define_G(opt))        # model.py:15
        self.schedule_phase = None
        self.set_loss()                                            # model.py:19
        self.set_new_noise_schedule(opt['model']['beta_schedule']['train'], schedule_phase='train')
        if opt['phase'] == 'train':
            self.netG.train()
            if opt['model'].get('finetune_norm'):                  # model.py:25-35
                # the reference freezes every parameter, then un-freezes (and zero-fills) those whose name contains
                # 'transformer', and hands only those to Adam.  No denoiser of this repository (nor of the reference's
                # four model dirs) has such a parameter, so the reference builds Adam over an EMPTY list, which torch
                # refuses ("optimizer got an empty parameter list"): same error, same moment.
                optim_params = []
                for k, v in self.netG.named_parameters():
                    v.requires_grad = False
                    if k.find('transformer') >= 0:
                        v.requires_grad = True
                        v.data.zero_()
                        optim_params.append(v)
                        logger.info('Params [{:s}] initialized to 0 and will optimize.'.format(k))
                if not optim_params:
                    raise ValueError('optimizer got an empty parameter list')
                raise NotImplementedError('finetune_norm with transformer parameters: the engine optimises the whole UNet')
            # torch.optim.Adam(netG.parameters(), lr) (model.py:37-38) lives in the engine: its state is the
            # (exp_avg, exp_avg_sq) pair kept beside the fp32 master copy of every executed tensor
            self.lr = float(opt['train']['optimizer']['lr'])
            self.betas, self.adam_eps = (0.9, 0.999), 1e-8
            self.log_dict = OrderedDict()
        self.load_network()                                        # model.py:41
        self.print_network()                                       # model.py:42

    def set_device(self, x):                                       # base_model.py:31-42
        if isinstance(x, dict):
            for k, v in x.items():
                if v is not None:
                    x[k] = v.to(self.device)
            return x
        return x.to(self.device)

    def feed_data(self, data):                                     # model.py:44-45
        self.data = self.set_device(data)

    def optimize_parameters(self):                                 # model.py:47-57
        """zero_grad; l_pix = netG(data); l_pix.sum() / (b*c*h*w); backward; optG.step() -- one call into the engine.
        Under torch.distributed every rank steps on its own shard and the gradient arena is all-reduced (the
        reference's nn.DataParallel, networks.py:116-118; then the divisor is the GLOBAL batch)."""
        from . import parallel
        world = parallel.world_size()
        hook = parallel.allreduce_grads if world > 1 else None
        b = int(self.data['HR'].shape[0])
        # the divisor is the GLOBAL sample count (ragged or empty shards included): one tiny all-reduce
        gb = parallel.sum_over_ranks(b) if world > 1 else b
        l_pix = self.netG.optimize_step(self.data, self.lr, self.betas, self.adam_eps, grad_hook=hook, global_batch=int(gb))
        self.log_dict['l_pix'] = parallel.sum_over_ranks(l_pix) if world > 1 else l_pix

    def get_current_log(self):                                     # model.py:94-95
        return self.log_dict

    def test(self, continous=False):                               # model.py:59-68
        self.netG.eval()
        with torch.no_grad():
            self.SR = self.netG.super_resolution(self.data['SR'], continous)
        self.netG.train()                                          # the reference flips back to train mode

    def set_loss(self):                                            # model.py:79-83
        self.netG.set_loss(self.device)

    def set_new_noise_schedule(self, schedule_opt, schedule_phase='train'):   # model.py:85-92
        if self.schedule_phase is None or self.schedule_phase != schedule_phase:
            self.schedule_phase = schedule_phase
            self.netG.set_new_noise_schedule(schedule_opt, self.device)

    def get_current_visuals(self, need_LR=True, sample=False):     # model.py:97-111
        out = OrderedDict()
        if sample:
            out['SAM'] = self.SR.detach().float().cpu()
        else:
            out['SR'] = self.SR.detach().float().cpu()
            out['INF'] = self.data['SR'].detach().float().cpu()
            out['HR'] = self.data['HR'].detach().float().cpu()
            if 'HR_Mask' in self.data:
                out['HR_Mask'] = self.data['HR_Mask'].detach().float().cpu()
            if need_LR and 'LR' in self.data:
                out['LR'] = self.data['LR'].detach().float().cpu()
            else:
                out['LR'] = out['INF']
        return out

    def get_network_description(self, network):                    # base_model.py:44-50
        if isinstance(network, nn.DataParallel):
            network = network.module
        return str(network), sum(x.numel() for x in network.parameters())

    def print_network(self):                                       # model.py:112-123
        s, n = self.get_network_description(self.netG)
        logger.info('n------Total params: %.3f M' % (n / (1024 * 1024)))
        logger.info('Network G structure: {}, with parameters: {:,d}'.format(self.netG.__class__.__name__, n))
        logger.info(s)

    def _optimizer_state_dict(self):
        """torch.optim.Adam.state_dict() of the reference's optG (model.py:143): parameter indices follow
        list(netG.parameters()); the 44 never-executed tensors have no state, as in torch."""
        unet = self.netG.denoise_fn
        eng = unet.engine
        live = {k for k, _, lv in eng.schema() if lv}
        state = {}
        names = [k for k, _ in unet.named_parameters()]
        for i, k in enumerate(names):
            if k in live:
                m, v, step = eng.optimizer_state(k)
                state[i] = {'step': torch.tensor(float(step)), 'exp_avg': torch.from_numpy(m), 'exp_avg_sq': torch.from_numpy(v)}
        group = {'lr': self.lr, 'betas': self.betas, 'eps': self.adam_eps, 'weight_decay': 0, 'amsgrad': False, 'maximize': False,
                 'foreach': None, 'capturable': False, 'differentiable': False, 'fused': None, 'params': list(range(len(names)))}
        return {'state': state, 'param_groups': [group]}

    def _load_optimizer_state_dict(self, osd):
        unet = self.netG.denoise_fn
        unet.sync_weights()
        eng = unet.engine
        names = [k for k, _ in unet.named_parameters()]
        for i, st in osd['state'].items():
            eng.set_optimizer_state(names[int(i)], st['exp_avg'].float().cpu().numpy(), st['exp_avg_sq'].float().cpu().numpy(),
                                    int(float(st['step'])))
        g = osd['param_groups'][0]
        self.lr, self.betas, self.adam_eps = float(g['lr']), tuple(g['betas']), float(g['eps'])

    def save_network(self, epoch, iter_step):                      # model.py:126-146
        gen_path = os.path.join(self.opt['path']['checkpoint'], 'I{}_E{}_gen.pth'.format(iter_step, epoch))
        opt_path = os.path.join(self.opt['path']['checkpoint'], 'I{}_E{}_opt.pth'.format(iter_step, epoch))
        os.makedirs(self.opt['path']['checkpoint'], exist_ok=True)   # the reference's parser made it (core/logger.py:37-43)
        sd = self.netG.state_dict()                                # pulls the engine's master copy after optimiser steps
        torch.save(OrderedDict((k, v.cpu()) for k, v in sd.items()), gen_path)
        if self.opt['phase'] == 'train' and self.netG.denoise_fn.engine.trained:
            torch.save({'epoch': epoch, 'iter': iter_step, 'scheduler': None, 'optimizer': self._optimizer_state_dict()}, opt_path)
        logger.info('Saved model in [{:s}] ...'.format(gen_path))
        return gen_path

    def load_network(self):                                        # model.py:148-166
        load_path = (self.opt.get('path') or {}).get('resume_state')
        if load_path is not None:
            logger.info('Loading pretrained model for G [{:s}] ...'.format(load_path))
            sd = torch.load('{}_gen.pth'.format(load_path), map_location='cpu')
            self.netG.load_state_dict(sd, strict=(not self.opt['model'].get('finetune_norm')))
            if self.opt['phase'] == 'train' and os.path.exists('{}_opt.pth'.format(load_path)):
                o = torch.load('{}_opt.pth'.format(load_path), map_location='cpu', weights_only=False)
                self._load_optimizer_state_dict(o['optimizer'])
                self.begin_step = o['iter']
                self.begin_epoch = o['epoch']


def create_model(opt):                                             # model/__init__.py:5-8
    m = DDPM(opt)
    logger.info('Model [{:s}] is created.'.format(m.__class__.__name__))
    return m
