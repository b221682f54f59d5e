"""`VSparams` — the reference's experiment driver (ref:params/VSparams.py) over the MI355X hot path.

Same command line, same public methods, same protocol as the reference class, so `VS_train.py` / `VS_inference.py` read
like the reference's scripts.  What is different underneath:

* data: no MONAI / nibabel / DataLoader workers — `vs_seg_amd.data` reads NIfTI, re-orients to RAS, normalises on the GPU,
  keeps every case cached in HBM and crops/flips whole batches with one HIP launch (the "loaders" returned by
  `cache_transformed_*_data` are light iterables over that cache yielding the reference's `{"image", "label"}` dicts);
* step: model / loss / optimizer are the HIP-backed objects; epoch sums stay on the device, the host reads them once per
  epoch (the reference calls `.item()` every step, ref:params/VSparams.py:463);
* validation (N1): eval forward + loss + hard Dice accumulated on the device, one read per validation pass.  The
  reference's loop adds every case twice (ref:params/VSparams.py:490-496): the mean Dice is unaffected, the logged
  validation loss is twice the mean — reproduced, because `best_metric`/logs are part of the observable behaviour;
* export (N3): argmax on the GPU, written as NIfTI in the label's original orientation and affine.

Not carried over (SURVEY §2 out-of-scope rows): TensorBoard writer, matplotlib figures.
"""
from __future__ import annotations

import csv
import logging
import os
from time import perf_counter, strftime
from typing import Dict, Iterator, List, Optional, Sequence

import numpy as np
import torch

from . import Adam, Dice_spvPA, UNet2d5_spvPA, compute_dice_score, sliding_window_inference
from .inferers import argmax_segmentation
from . import parallel as DP
from .data import nifti
from .data.transforms import PatchSampler, epoch_batches, load_case

HP = dict(
    channels=(16, 32, 48, 64, 80, 96),
    strides=((2, 2, 1), (2, 2, 1), (2, 2, 2), (2, 2, 2), (2, 2, 2)),
    kernel_sizes=((3, 3, 1), (3, 3, 1), (3, 3, 3), (3, 3, 3), (3, 3, 3), (3, 3, 3)),
    sample_kernel_sizes=((3, 3, 1), (3, 3, 1), (3, 3, 3), (3, 3, 3), (3, 3, 3)),
)


class CachedLoader:
    """Iterable over GPU-cached cases with the random tail of the transform chain applied per epoch.

    Yields the reference DataLoader's batch dicts: {"image": [B,1,X,Y,Z], "label": [B,1,X,Y,Z]} fp32 device tensors (+ the
    meta dicts of the cases for batch_size 1 loaders).  `len()` = number of cases, `batch_size` as in torch's DataLoader —
    the reference's first-epoch log line divides one by the other (ref:params/VSparams.py:466)."""

    def __init__(self, cases: List[Dict], roi: Optional[Sequence[int]], batch_size: int, shuffle: bool, flip_prob: Optional[float], seed: int = 0, pad: Optional[bool] = None):
        self.cases, self.batch_size, self.shuffle = cases, batch_size, shuffle
        # equal step counts on every rank (wrap-around padding) only where every step issues a collective: the shuffled training loader.
        # Validation / test loaders give rank r exactly shard_indices(n, r, world) — a padded case would be counted twice in their sums
        self.pad = shuffle if pad is None else pad
        self.rank, self.world = DP.get_rank(), DP.world_size()
        # every rank shuffles with the SAME stream (the shards must partition one permutation) but draws its own flips / crops
        self.sampler = PatchSampler(cases, roi, flip_prob, seed + 7919 * self.rank) if roi is not None else None
        self._order = np.random.RandomState(seed)

    def __len__(self):
        return len(self.cases)

    def __iter__(self) -> Iterator[Dict]:
        for idx in epoch_batches(len(self.cases), self.batch_size, self.shuffle, self._order, self.rank, self.world, pad=self.pad):
            if self.sampler is not None:
                img, lab = self.sampler.sample(idx)
            else:  # test chain: whole volumes, no crop (ref:params/VSparams.py:238-245)
                assert len(idx) == 1
                img, lab = self.cases[idx[0]]["image"][None, None], self.cases[idx[0]]["label"][None, None]
            batch = {"image": img, "label": lab}
            if len(idx) == 1:
                batch["image_meta_dict"], batch["label_meta_dict"] = self.cases[idx[0]]["image_meta"], self.cases[idx[0]]["label_meta"]
            yield batch


class VSparams:
    def __init__(self, parser, argv=None):
        parser.add_argument("--debug", dest="debug", action="store_true", help="activate debugging mode")
        parser.set_defaults(debug=False)
        parser.add_argument("--split", type=str, default="./params/split_TCIA.csv", help="path to CSV file that defines training, validation and test datasets")
        parser.add_argument("--dataset", type=str, default="T1", help='(string) use "T1" or "T2" to select dataset')
        parser.add_argument("--train_batch_size", type=int, default=1, help="batch size of the forward pass")
        parser.add_argument("--initial_learning_rate", type=float, default=1e-4, help="learning rate at first epoch")
        parser.add_argument("--no_attention", dest="attention", action="store_false", help="disables the attention module in the network and the attention map weighting in the loss function")
        parser.set_defaults(attention=True)
        parser.add_argument("--no_hardness", dest="hardness", action="store_false", help="disables the hardness weighting in the loss function")
        parser.set_defaults(hardness=True)
        parser.add_argument("--results_folder_name", type=str, default="temp" + strftime("%Y%m%d%H%M%S"), help="name of results folder")
        # additions of this implementation (defaults reproduce the reference)
        parser.add_argument("--data_root", type=str, default="./data/VS_defaced/", help="data set root (the reference hard-codes this path)")
        parser.add_argument("--compute_dtype", type=str, default="bf16", choices=["bf16", "fp32"], help="bf16 MFMA (benchmark) or exact-fp32 MFMA (parity)")
        parser.add_argument("--num_epochs", type=int, default=None)
        args = parser.parse_args(argv)

        self.debug, self.dataset, self.data_root = args.debug, args.dataset, args.data_root
        self.split_csv = "./params/split_debug.csv" if self.debug else args.split
        self.pad_crop_shape = [128, 128, 32] if self.debug else [384, 384, 64]
        self.pad_crop_shape_test = list(self.pad_crop_shape)
        self.num_workers = 4  # kept for the log; there are no loader workers (the cache lives in HBM)
        self.torch_device_arg = "cuda:0"
        self.train_batch_size = args.train_batch_size
        self.initial_learning_rate = args.initial_learning_rate
        self.epochs_with_const_lr = 3 if self.debug else 100
        self.lr_divisor = 2.0
        self.weight_decay = 1e-7
        self.num_epochs = args.num_epochs or (10 if self.debug else 300)
        self.val_interval = 2
        self.model = "UNet2d5_spvPA"
        self.sliding_window_inferer_roi_size = [128, 128, 32] if self.debug else [384, 384, 64]
        self.attention, self.hardness = args.attention, args.hardness
        self.export_inferred_segmentations = True
        self.compute_dtype = args.compute_dtype
        self.results_folder_path = os.path.join(self.data_root, "results", "debug" if self.debug else args.results_folder_name)
        self.logs_path = os.path.join(self.results_folder_path, "logs")
        self.model_path = os.path.join(self.results_folder_path, "model")
        self.figures_path = os.path.join(self.results_folder_path, "figures")
        rank, world, local = DP.init_distributed()
        self.rank, self.world = rank, world
        if not torch.cuda.is_available():
            raise RuntimeError("vs_seg_amd.VSparams: no GPU visible — this implementation has no CPU path (the reference falls back to whatever torch.device('cuda:0') does)")
        self.device = torch.device("cuda", local)
        self.logger = logging.getLogger()

    # ------------------------------------------------------------------ housekeeping
    def create_results_folders(self):
        for p in (self.logs_path, self.model_path, self.figures_path):
            os.makedirs(p, exist_ok=True)

    def set_up_logger(self, log_file_name):
        os.makedirs(self.logs_path, exist_ok=True)
        self.logger = logging.getLogger()
        fmt = logging.Formatter("%(asctime)s %(levelname)s        %(message)s")
        for h in (logging.FileHandler(os.path.join(self.logs_path, log_file_name), mode="w"), logging.StreamHandler()):
            h.setFormatter(fmt)
            self.logger.addHandler(h)
        self.logger.setLevel(logging.INFO if self.rank == 0 else logging.WARNING)
        self.logger.info("Created " + log_file_name)
        return self.logger

    def log_parameters(self):
        log = self.logger.info
        log("-" * 10)
        log("Parameters: ")
        for k in ("dataset", "data_root", "split_csv", "pad_crop_shape", "pad_crop_shape_test", "num_workers", "torch_device_arg", "train_batch_size", "initial_learning_rate",
                  "epochs_with_const_lr", "lr_divisor", "weight_decay", "num_epochs", "val_interval", "model", "sliding_window_inferer_roi_size", "attention", "hardness",
                  "results_folder_path", "export_inferred_segmentations", "compute_dtype"):
            log("{:<34s} {}".format(k + " =", getattr(self, k)))
        log("-" * 10)

    # ------------------------------------------------------------------ data
    def load_T1_or_T2_data(self):
        names = {"T1": ("vs_gk_t1_refT1.nii.gz", "vs_gk_seg_refT1.nii.gz"), "T2": ("vs_gk_t2_refT2.nii.gz", "vs_gk_seg_refT2.nii.gz")}[self.dataset]
        sets: Dict[str, list] = {"training": [], "validation": [], "test": []}
        with open(self.split_csv) as f:
            for row in csv.reader(f):
                if len(row) >= 2 and row[1] in sets:
                    d = os.path.join(self.data_root, "input_data", row[0])
                    sets[row[1]].append({"image": os.path.join(d, names[0]), "label": os.path.join(d, names[1])})
        for fd in sets["training"] + sets["validation"] + sets["test"]:
            for k in ("image", "label"):
                assert os.path.isfile(fd[k]), f" {fd[k]} is not a file"
        self.logger.info("Number of images in training set   = {}".format(len(sets["training"])))
        self.logger.info("Number of images in validation set = {}".format(len(sets["validation"])))
        self.logger.info("Number of images in test set       = {}".format(len(sets["test"])))
        return sets["training"], sets["validation"], sets["test"]

    def get_transforms(self):
        """The three chains as plain descriptions; `cache_transformed_*_data` executes them (deterministic head cached in
        HBM, random tail per batch)."""
        head = ["LoadNifti", "AddChannel", "Orientation(RAS)", "NormalizeIntensity(image)"]
        train = dict(chain=head + [f"SpatialPad({self.pad_crop_shape})", "RandFlip(p=0.5, axis=0)", f"RandSpatialCrop({self.pad_crop_shape})"], pad=self.pad_crop_shape, roi=self.pad_crop_shape, flip_prob=0.5)
        val = dict(chain=head + [f"SpatialPad({self.pad_crop_shape})", f"RandSpatialCrop({self.pad_crop_shape})"], pad=self.pad_crop_shape, roi=self.pad_crop_shape, flip_prob=None)
        test = dict(chain=head, pad=None, roi=None, flip_prob=None)
        return train, val, test

    @staticmethod
    def get_center_of_mass_slice(label):
        """Index of the z slice closest to the label's centre of mass (uniform weights for an empty label)."""
        lab = np.asarray(label.detach().cpu() if torch.is_tensor(label) else label, dtype=np.float64)
        n = lab.shape[2]
        masses = np.array([lab[:, :, z].sum() for z in range(n)])
        w = masses / sum(masses) if sum(masses) != 0 else np.ones(n) / n
        # left-to-right Python sum, as the reference accumulates it: for an empty 10-slice label that is 4.500000000000001 -> 5
        return int(sum(w * np.arange(n)).round())

    def check_transforms_on_first_validation_image_and_label(self, val_files, val_transforms):
        """ref:params/VSparams.py:266-297 (called at ref:VS_train.py:36): run the validation chain on the first validation case and
        log what came out.  The PNG of the centre-of-mass slice is out of scope; the slice index is still logged."""
        logger = self.logger
        case = load_case(val_files[0], val_transforms["pad"], self.device)
        sampler = PatchSampler([case], val_transforms["roi"], val_transforms["flip_prob"], seed=0)
        img, lab = sampler.sample([0])
        check_data = {"image": img, "label": lab, "image_meta_dict": case["image_meta"], "label_meta_dict": case["label_meta"]}
        image, label = check_data["image"][0][0], check_data["label"][0][0]
        logger.info("-" * 10)
        logger.info("Check the transforms on the first validation set image and label")
        logger.info("Length of check_data = {}".format(len(check_data)))
        logger.info("check_data['image'].shape = {}".format(check_data["image"].shape))
        logger.info("Validation image shape = {}".format(image.shape))
        logger.info("Validation label shape = {}".format(label.shape))
        slice_idx = self.get_center_of_mass_slice(label)
        logger.info("-" * 10)
        logger.info("image shape: {}, label shape: {}, slice = {}".format(image.shape, label.shape, slice_idx))
        return check_data

    def _cache(self, files, tf, batch_size, shuffle, what):
        self.logger.info(f"Caching {what} data set...")
        cases = [load_case(fd, tf["pad"], self.device) for fd in files]
        return CachedLoader(cases, tf["roi"], batch_size, shuffle, tf["flip_prob"], seed=0)  # the crop/flip stream is decorrelated per rank inside CachedLoader

    def cache_transformed_train_data(self, train_files, train_transforms):
        return self._cache(train_files, train_transforms, self.train_batch_size, True, "training")

    def cache_transformed_val_data(self, val_files, val_transforms):
        return self._cache(val_files, val_transforms, 1, False, "validation")

    def cache_transformed_test_data(self, test_files, test_transforms):
        return self._cache(test_files, test_transforms, 1, False, "test")

    # ------------------------------------------------------------------ model / loss / optimizer
    def set_and_get_model(self):
        self.logger.info("Setting up the model type...")
        if self.model != "UNet2d5_spvPA":
            raise Exception("Model not defined.")
        return UNet2d5_spvPA(dimensions=3, in_channels=1, out_channels=2, num_res_units=2, norm="batch", dropout=0.1, attention_module=self.attention,
                             compute_dtype=self.compute_dtype, **HP).to(self.device)

    def set_and_get_loss_function(self):
        self.logger.info("Setting up the loss function...")
        return Dice_spvPA(to_onehot_y=True, softmax=True, supervised_attention=self.attention, hardness_weighting=self.hardness)

    def set_and_get_optimizer(self, model):
        self.logger.info("Setting up the optimizer...")
        return Adam(model.parameters(), lr=self.initial_learning_rate, weight_decay=self.weight_decay)

    def compute_dice_score(self, predicted_probabilities, label):
        return compute_dice_score(predicted_probabilities, label)  # [1,1] tensor, as the reference's

    # ------------------------------------------------------------------ training (ref:params/VSparams.py:409-528)
    def run_training_algorithm(self, model, loss_function, optimizer, train_loader, val_loader):
        logger = self.logger
        logger.info("Running the training loop...")
        trainer = DP.DataParallelTrainer(model, loss_function, optimizer)
        best_metric, best_metric_epoch = -1, -1
        epoch_loss_values, metric_values = [], []
        start = perf_counter()
        for epoch in range(self.num_epochs):
            logger.info("-" * 10)
            logger.info("Epoch {}/{}".format(epoch + 1, self.num_epochs))
            if epoch == self.val_interval:
                el = perf_counter() - start
                logger.info("Average duration of first {0:.0f} epochs = {1:.2f} s. Expected total training time = {2:.2f} h".format(self.val_interval, el / self.val_interval, el * self.num_epochs / self.val_interval / 3600))
            model.train()
            step, losses = 0, []
            for batch in train_loader:
                step += 1
                losses.append(trainer.step(batch["image"], batch["label"], sync=False))  # device scalar, no host read
            lv = torch.stack(losses).double().cpu() if losses else torch.zeros(0)
            if epoch == 0:
                denom = len(train_loader) // train_loader.batch_size  # the reference's denominator (ref:params/VSparams.py:466)
                for i, v in enumerate(lv.tolist()):
                    logger.info("{}/{}, train_loss: {:.4f}".format(i + 1, denom, v))
            epoch_loss = float(lv.sum()) / max(step, 1)
            if not np.isfinite(epoch_loss):
                # the fixed-point accumulators (BatchNorm statistics, backward sums, Dice sums) flag an out-of-range or non-finite partial sum and every kernel that decodes
                # them returns NaN from then on (include/vsseg_hip.h): name that cause once and clear the flag, so that a later epoch is not poisoned by it
                from ._lib import fx_status

                if fx_status(reset=True):
                    logger.error("epoch {}: a fixed-point partial sum left its range or was NaN / Inf (diverging activations or gradients, or a loss scale beyond 256): "
                                 "the statistics / loss of the affected steps are NaN; the flag was reset".format(epoch + 1))
            epoch_loss_values.append(epoch_loss)
            logger.info("epoch {} average loss: {:.4f}".format(epoch + 1, epoch_loss))

            if (epoch + 1) % self.val_interval == 0:
                metric, epoch_loss_val = self.validate(model, loss_function, val_loader)
                metric_values.append(metric)
                logger.info("validation loss (reference accounting: twice the mean): {:.4f}".format(epoch_loss_val))
                if metric > best_metric:
                    best_metric, best_metric_epoch = metric, epoch + 1
                    if self.rank == 0:
                        torch.save(model.state_dict(), os.path.join(self.model_path, "best_metric_model.pth"))
                    logger.info("saved new best metric model")
                logger.info("current epoch {} current mean dice: {:.4f} best mean dice: {:.4f} at epoch {}".format(epoch + 1, metric, best_metric, best_metric_epoch))

            if (epoch + 1) % self.epochs_with_const_lr == 0:
                for group in optimizer.param_groups:
                    group["lr"] = group["lr"] / self.lr_divisor
                    logger.info("Dividing learning rate by {}. New learning rate is: lr = {}".format(self.lr_divisor, group["lr"]))

        logger.info("Train completed, best_metric: {:.4f}  at epoch: {}".format(best_metric, best_metric_epoch))
        if self.rank == 0:
            torch.save(model.state_dict(), os.path.join(self.model_path, "last_epoch_model.pth"))
        logger.info(f'Saved model of the last epoch at: {os.path.join(self.model_path, "last_epoch_model.pth")}')
        return epoch_loss_values, metric_values

    def validate(self, model, loss_function, val_loader):
        """N1: one validation pass without per-case host reads.  Returns (mean Dice, validation loss in the reference's
        accounting).  Multi-GPU: cases are sharded by the loader, sums are all-reduced."""
        DP.broadcast_buffers(model)  # data parallel: every rank validates (and rank 0 saves) rank 0's BatchNorm running statistics
        model.eval()
        dice_sum = torch.zeros((), dtype=torch.float64, device=self.device)
        loss_sum = torch.zeros((), dtype=torch.float64, device=self.device)
        count = steps = 0
        with torch.no_grad():
            for batch in val_loader:
                steps += 1
                outputs = model(batch["image"])
                dice = self.compute_dice_score(outputs[0], batch["label"])
                loss = loss_function(outputs, batch["label"])
                # the reference adds every case twice (ref:params/VSparams.py:490-496)
                count += 2 * len(dice)
                dice_sum += 2.0 * dice.sum().double()
                loss_sum += 2.0 * loss.double()
        t = torch.stack([dice_sum, loss_sum, torch.tensor(float(count), dtype=torch.float64, device=self.device), torch.tensor(float(steps), dtype=torch.float64, device=self.device)])
        t = DP.allreduce_sum(t)
        d, l, c, s = t.tolist()  # the only host read of the pass
        return d / max(c, 1.0), l / max(s, 1.0)

    def plot_loss_curve_and_mean_dice(self, epoch_loss_values, metric_values):
        """Figures are out of scope; the curves are written as CSV next to where the reference puts its PNG."""
        os.makedirs(self.figures_path, exist_ok=True)
        with open(os.path.join(self.figures_path, "epoch_average_loss_and_val_mean_dice.csv"), "w") as f:
            f.write("epoch,average_loss,val_mean_dice\n")
            for i, v in enumerate(epoch_loss_values):
                m = metric_values[(i + 1) // self.val_interval - 1] if (i + 1) % self.val_interval == 0 and (i + 1) // self.val_interval <= len(metric_values) else ""
                f.write(f"{i + 1},{v},{m}\n")

    # ------------------------------------------------------------------ inference (ref:params/VSparams.py:546-625)
    def load_trained_state_of_model(self, model):
        from .checkpoint import load_checkpoint

        load_checkpoint(model, os.path.join(self.model_path, "best_metric_model.pth"))
        return model

    def run_inference(self, model, data_loader):
        logger = self.logger
        logger.info("Running inference...")
        model.eval()
        n = len(data_loader)
        dice_dev = torch.zeros(n, dtype=torch.float32, device=self.device)
        predictor = model.segmentation_predictor() if hasattr(model, "segmentation_predictor") else (lambda *a, **k: model(*a, **k)[0])  # model_segmentation, ref:params/VSparams.py:560
        mine = DP.shard_indices(n)  # the unpadded, unshuffled test loader yields exactly these cases, in this order
        with torch.no_grad():
            for i, data in enumerate(data_loader):
                assert i < len(mine), "the test loader yielded more cases than this rank's shard (a padded loader would double-count them)"
                logger.info("starting image {}".format(mine[i]))
                outputs = sliding_window_inference(inputs=data["image"], roi_size=self.sliding_window_inferer_roi_size, sw_batch_size=1, predictor=predictor, mode="gaussian")
                gi = mine[i]
                dice_dev[gi] = self.compute_dice_score(outputs, data["label"]).reshape(())
                if self.export_inferred_segmentations:
                    self.export_segmentation(outputs, data["label_meta_dict"])
        dice_scores = DP.allreduce_sum(dice_dev).double().cpu().numpy() if self.world > 1 else dice_dev.double().cpu().numpy()
        for i, v in enumerate(dice_scores):
            logger.info(f"dice_score[{i}] = {v}")
        logger.info(f"all_dice_scores = {dice_scores}")
        logger.info(f"mean_dice_score = {dice_scores.mean()} +- {dice_scores.std()}")
        return dice_scores

    def export_segmentation(self, outputs, label_meta):
        """N3: argmax → uint8 NIfTI in the label's original orientation / affine, under
        results/inferred_segmentations_nifti/<case folder>/ like MONAI's NiftiSaver(output_postfix='')."""
        seg = argmax_segmentation(outputs)[0].cpu().numpy()  # vsseg_argmax2: ties -> class 0, like torch.argmax
        seg = nifti.from_ras(seg, label_meta["ornt"])
        src = label_meta["filename_or_obj"]
        folder = os.path.basename(os.path.dirname(src))
        out_dir = os.path.join(self.results_folder_path, "inferred_segmentations_nifti", folder, os.path.basename(src).split(".")[0])
        os.makedirs(out_dir, exist_ok=True)
        path = os.path.join(out_dir, os.path.basename(src).split(".")[0] + ".nii.gz")
        nifti.write_nifti(path, seg, label_meta["original_affine"], dtype=np.uint8)
        self.logger.info(f"export to nifti... {path}")
        return path
