"""`DDPM`: the model wrapper the reference's drivers use (FastDiffSR/model/model.py, base_model.py,
model/__init__.py:create_model), over the HIP-backed netG.  Sampling surface only (the optimizer /
`optimize_parameters` path is SURVEY 8f-3)."""
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

    def save_network(self, epoch, iter_step):                      # model.py:126-146 (generator part)
        gen_path = os.path.join(self.opt['path']['checkpoint'], 'I{}_E{}_gen.pth'.format(iter_step, epoch))
        sd = self.netG.state_dict()
        torch.save(OrderedDict((k, v.cpu()) for k, v in sd.items()), gen_path)
        return gen_path

    def load_network(self):                                        # model.py:148-160
        load_path = (self.opt.get('path') or {}).get('resume_state')
        if load_path is not None:
            logger.info('Loading pretrained model for G [{:s}] ...'.format(load_path))
            sd = torch.load('{}_gen.pth'.format(load_path), map_location='cpu')
            self.netG.load_state_dict(sd, strict=(not self.opt['model'].get('finetune_norm')))


def create_model(opt):                                             # model/__init__.py:5-8
    m = DDPM(opt)
    logger.info('Model [{:s}] is created.'.format(m.__class__.__name__))
    return m
