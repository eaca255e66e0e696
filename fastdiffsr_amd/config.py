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


def load_config(path, phase='val'):
    with open(path, 'r') as f:
        opt = parse_json_with_comments(f.read())
    opt['phase'] = phase
    return dict_to_nonedict(opt)
