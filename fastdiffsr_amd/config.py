"""Minimal reader for the reference's JSON-with-`//`-comments configs
(FastDiffSR/core/logger.py:21-32, :97-112): strips per-line `//` comments and
returns a dict whose missing keys read as None."""
import json
from collections import OrderedDict


class NoneDict(dict):
    def __missing__(self, key):
        return None


def dict_to_nonedict(opt):
    if isinstance(opt, dict):
        return NoneDict(**{k: dict_to_nonedict(v) for k, v in opt.items()})
    if isinstance(opt, list):
        return [dict_to_nonedict(v) for v in opt]
    return opt


def parse_json_with_comments(text):
    lines = [ln.split('//')[0] + '\n' for ln in text.splitlines()]
    return json.loads(''.join(lines), object_pairs_hook=OrderedDict)


# what `-debug` rewrites (core/logger.py:59-68): validate every 2 iterations, checkpoint every 3, batch 2,
# T = 10 for both schedules, 6 training / 3 validation images
_DEBUG_OVERRIDES = (
    (('train', 'val_freq'), 2), (('train', 'print_freq'), 2), (('train', 'save_checkpoint_freq'), 3),
    (('datasets', 'train', 'batch_size'), 2),
    (('model', 'beta_schedule', 'train', 'n_timestep'), 10), (('model', 'beta_schedule', 'val', 'n_timestep'), 10),
    (('datasets', 'train', 'data_len'), 6), (('datasets', 'val', 'data_len'), 3),
)


def _assign(tree, path_, value):
    for key in path_[:-1]:
        tree = tree[key]
    tree[path_[-1]] = value


def load_config(path, phase='val', gpu_ids=None, debug=False, enable_wandb=False, log_wandb_ckpt=False,
                log_eval=False, log_infer=False, timestamp=None):
    """`core.logger.parse(args)` (logger.py:21-94) without its side effects: the experiment directories are
    named but not created and CUDA_VISIBLE_DEVICES is left alone.  Everything else follows the reference,
    including `distributed = len(gpu_list) > 1` on the comma-joined STRING (:52-56) and the `-debug`
    rewrites (:59-68).  Checked against the reference's own parser on its configs (tests/golden/configs.json)."""
    import os
    from datetime import datetime
    with open(path, 'r') as f:
        opt = parse_json_with_comments(f.read())
    if debug:
        opt['name'] = 'debug_{}'.format(opt['name'])
    stamp = timestamp or datetime.now().strftime('%y%m%d_%H%M%S')
    root = os.path.join('experiments', '{}_{}'.format(opt['name'], stamp))
    if 'path' in opt:
        opt['path']['experiments_root'] = root
        for key, sub in list(opt['path'].items()):
            if 'resume' not in key and 'experiments' not in key:
                opt['path'][key] = os.path.join(root, sub)
    opt['phase'] = phase
    if gpu_ids is not None:
        opt['gpu_ids'] = [int(i) for i in gpu_ids.split(',')]
        gpu_list = gpu_ids
    else:
        gpu_list = ','.join(str(x) for x in opt['gpu_ids'])
    opt['distributed'] = len(gpu_list) > 1
    if 'debug' in opt['name']:                       # logger.py:59-68: the -debug shrink, as a table
        for path_, value in _DEBUG_OVERRIDES:
            _assign(opt, path_, value)
    if phase == 'train':                             # logger.py:71-72
        _assign(opt, ('datasets', 'val', 'data_len'), 13)
    opt['log_wandb_ckpt'] = log_wandb_ckpt
    opt['log_eval'] = log_eval
    opt['log_infer'] = log_infer
    opt['enable_wandb'] = enable_wandb
    return dict_to_nonedict(opt)


def setup_logger(logger_name, root, phase, level=None, screen=False):
    """core/logger.py:128-141: `<root>/<phase>.log` (mode 'w'), optional stderr echo, the reference's line format."""
    import logging
    import os
    lg = logging.getLogger(logger_name)
    fmt = logging.Formatter('%(asctime)s.%(msecs)03d - %(levelname)s: %(message)s', datefmt='%y-%m-%d %H:%M:%S')
    os.makedirs(root, exist_ok=True)
    fh = logging.FileHandler(os.path.join(root, '{}.log'.format(phase)), mode='w')
    fh.setFormatter(fmt)
    lg.setLevel(logging.INFO if level is None else level)
    lg.addHandler(fh)
    if screen:
        sh = logging.StreamHandler()
        sh.setFormatter(fmt)
        lg.addHandler(sh)
    return lg
