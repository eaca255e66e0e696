"""`python -m fastdiffsr_amd.train -c <config>` = the train phase of the reference's sr_mfe.py (:69-251): iterations with
log lines, the validation pass every val_freq (val schedule, then back), checkpoints every save_checkpoint_freq, resume -- for
which_model_G 'fastdiffsr' and for the siblings 'ddpm' (SR3) and 'gdp' (model/networks.py:82-119: the reference trains them all through the
same loop; define_G builds the GDP UNet at its default width of 128 channels whatever `inner_channel` says)."""
import json
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _config(root, exp, which='fastdiffsr'):
    sched = dict(schedule='linear_cosine', n_timestep=20, linear_start=1e-6, linear_end=1e-2)
    return {
        "name": "sr_fastdiffsr_train", "phase": "train", "gpu_ids": [0],
        "path": {"log": "logs", "tb_logger": "tb_logger", "results": "results", "checkpoint": "checkpoint", "resume_state": None},
        "datasets": {"train": {"name": "t", "mode": "HR", "dataroot": root, "datatype": "img", "l_resolution": 16,
                               "r_resolution": 64, "batch_size": 2, "num_workers": 0, "use_shuffle": True, "data_len": -1},
                     "val": {"name": "v", "mode": "LRHR", "dataroot": root, "datatype": "img", "l_resolution": 16,
                             "r_resolution": 64, "data_len": 2}},
        "model": {"which_model_G": which, "finetune_norm": False,
                  "unet": {"in_channel": 6, "out_channel": 3, "inner_channel": 32, "channel_multiplier": [1, 2, 2],
                           "attn_res": [16], "res_blocks": 1, "dropout": 0.2},
                  "beta_schedule": {"train": dict(sched), "val": dict(sched)},
                  "diffusion": {"image_size": 64, "channels": 3, "conditional": True}},
        "train": {"n_iter": 4, "val_freq": 2, "save_checkpoint_freq": 4, "print_freq": 1,
                  "optimizer": {"type": "adam", "lr": 1e-4},
                  "ema_scheduler": {"step_start_ema": 5000, "update_ema_every": 1, "ema_decay": 0.9999}},
        "wandb": {"project": "x"}}


@pytest.mark.parametrize('which', ['fastdiffsr', 'ddpm', 'gdp'])
def test_train_driver_iterations_val_checkpoint_resume(tmp_path, which):
    from fastdiffsr_amd import train
    from fastdiffsr_amd.config import load_config
    from test_val_host import make_dataset
    root = make_dataset(str(tmp_path / 'data'), n=5, l=16, r=64, seed=8)
    cwd = os.getcwd()
    os.chdir(tmp_path)                       # the parser creates experiments/<name>_<timestamp>/ under the cwd (core/logger.py:37-43)
    try:
        cpath = tmp_path / 'train.json'
        cpath.write_text(json.dumps(_config(root, 'a', which)))
        opt = load_config(str(cpath), phase='train')
        torch.manual_seed(3)
        np.random.seed(3)
        lines = []
        for prec in ('f16x3',):
            diffusion, hist = train.run(opt, precision=prec, log=lines.append)
        msgs = [m for m in lines if m.startswith('<epoch')]
        assert sum('l_pix' in m for m in msgs) == 4                    # print_freq 1, n_iter 4
        assert sum('sr_psnr' in m for m in msgs) == 2                  # validation at iter 2 and 4 (two summary lines each: bic, sr)
        losses = [v['l_pix'] for s, v in hist if 'l_pix' in v]
        assert len(losses) == 4 and all(np.isfinite(losses)) and all(0 < x < 10 for x in losses)
        ck = opt['path']['checkpoint']
        assert sorted(os.listdir(ck)) == ['I4_E2_gen.pth', 'I4_E2_opt.pth'], os.listdir(ck)      # 5 images / batch 2 = 3 iters per epoch
        # the parser forces val data_len = 13 in the train phase (core/logger.py:71-72): all 5 images, twice
        assert len([f for f in os.listdir(opt['path']['results']) if f.endswith('_sr.tif')]) == 10
        assert diffusion.schedule_phase == 'train'                     # switched back after the validation pass
        # resume: begin_step / begin_epoch restored, two more iterations run
        cfg2 = _config(root, 'b', which)
        cfg2['path']['resume_state'] = os.path.join(ck, 'I4_E2')
        cfg2['train'].update(n_iter=6, val_freq=100, save_checkpoint_freq=100)
        cpath2 = tmp_path / 'resume.json'
        cpath2.write_text(json.dumps(cfg2))
        opt2 = load_config(str(cpath2), phase='train')
        lines2 = []
        d2, hist2 = train.run(opt2, log=lines2.append)
        assert any('Resuming training from epoch: 2, iter: 4.' in m for m in lines2)
        assert [s for s, v in hist2 if 'l_pix' in v] == [5, 6]
        first_conv = 'input_blocks.0.0.weight' if which == 'gdp' else 'downs.0.weight'
        assert d2.netG.denoise_fn.engine.optimizer_state(first_conv)[2] == 6                # Adam's step count carried over
    finally:
        os.chdir(cwd)
