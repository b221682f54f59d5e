#!/usr/bin/env python
"""Train the 2.5D attention U-Net — the reference's VS_train.py (ref:VS_train.py) on the MI355X hot path.

    python VS_train.py --results_folder_name run1 [--dataset T2] [--no_attention] [--no_hardness] [--debug]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 VS_train.py ...   (data parallel)
"""
import argparse
import random

import numpy as np
import torch

from vs_seg_amd.params import VSparams

parser = argparse.ArgumentParser(description="Train the model")
p = VSparams(parser)
p.create_results_folders()
logger = p.set_up_logger("training_log.txt")
p.log_parameters()
train_files, val_files, test_files = p.load_T1_or_T2_data()
train_transforms, val_transforms, test_transforms = p.get_transforms()
# monai.utils.set_determinism(seed=0) (ref:VS_train.py:33): python / numpy / torch seeds; dropout masks are Philox(seed, step)
random.seed(0)
np.random.seed(0)
torch.manual_seed(0)
p.check_transforms_on_first_validation_image_and_label(val_files, val_transforms)
train_loader = p.cache_transformed_train_data(train_files, train_transforms)
val_loader = p.cache_transformed_val_data(val_files, val_transforms)
model = p.set_and_get_model()
loss_function = p.set_and_get_loss_function()
optimizer = p.set_and_get_optimizer(model)
epoch_loss_values, metric_values = p.run_training_algorithm(model, loss_function, optimizer, train_loader, val_loader)
p.plot_loss_curve_and_mean_dice(epoch_loss_values, metric_values)
