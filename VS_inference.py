#!/usr/bin/env python
"""Sliding-window inference on the test set — the reference's VS_inference.py (ref:VS_inference.py) on the MI355X hot path.

    python VS_inference.py --results_folder_name run1 [--dataset T2] [--no_attention] [--debug]
"""
import argparse
import random

import numpy as np
import torch

from vs_seg_amd.params import VSparams

parser = argparse.ArgumentParser(description="Run inference with the trained model")
p = VSparams(parser)
logger = p.set_up_logger("test_log.txt")
p.log_parameters()
train_files, val_files, test_files = p.load_T1_or_T2_data()
train_transforms, val_transforms, test_transforms = p.get_transforms()
random.seed(0)
np.random.seed(0)
torch.manual_seed(0)
test_loader = p.cache_transformed_test_data(test_files, test_transforms)
model = p.set_and_get_model()
model = p.load_trained_state_of_model(model)
p.run_inference(model, test_loader)
