"""Synthetic COCO-shaped data module producing the reference's collate wire format ON THE DEVICE:
`(x f32 [B,3,H,W] in [0,1] (BGR order, no mean/std: dataset/Pascal.py:98,179), y f32 [nT,6] = (img_idx_in_batch,
cls, x, y, w, h) normalised)` -- dataset/Pascal.py:176-186.  Datasets themselves are out of scope (absent data,
Windows paths); this module exists so that training/eval loops can run at full device speed without a CPU loader."""
import os

import torch


class SyntheticCocoModule:
    name = "synthetic"

    def __init__(self, batch_size, img_size, num_classes=80, targets_per_image=7, steps_per_epoch=8, device="cuda", seed=1,
                 raw_frames=False, channels_last=False, rank=None):
        self.batch_size, self.img_size, self.num_classes = batch_size, img_size, num_classes
        self.targets_per_image, self.steps_per_epoch = targets_per_image, steps_per_epoch
        # data-parallel ranks draw different batches (what PL's DistributedSampler does for the reference's datasets): rank r
        # seeds from `seed + r` (seed 1 + rank); rank=None reads RANK from the launcher's environment
        rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.device, self.seed, self.rank = torch.device(device), seed + rank, rank
        # raw_frames: emit COCO-like uint8 frames of varying size + pixel boxes and run the device collate
        # (resize / flips / normalise / row rewrite, dataset/collate.py) -- the train-time chain of Pascal.py:121-186
        self.raw_frames, self.channels_last = raw_frames, channels_last

    def setup(self, stage=None):
        return self

    def get_class(self):
        return [str(i) for i in range(self.num_classes)]

    def _batch(self, g):
        B, n = self.batch_size, self.batch_size * self.targets_per_image
        x = torch.rand(B, 3, self.img_size, self.img_size, generator=g, device=self.device)
        b = torch.arange(B, device=self.device).repeat_interleave(self.targets_per_image).float()
        cls = torch.randint(0, self.num_classes, (n,), generator=g, device=self.device).float()
        wh = torch.rand(n, 2, generator=g, device=self.device) * 0.48 + 0.02
        xy = torch.rand(n, 2, generator=g, device=self.device) * (1 - wh)
        return x, torch.cat([b[:, None], cls[:, None], xy, wh], 1)

    def _raw_batch(self, g, hg, train):
        from .collate import draw_flips, draw_shift_scale_rotate, gpu_collate, photometric_luts
        images, objects = [], []
        for _ in range(self.batch_size):
            h, w = [int(v) for v in torch.randint(320, 641, (2,), generator=hg)]            # COCO-sized frames
            images.append(torch.randint(0, 256, (h, w, 3), generator=g, device=self.device, dtype=torch.uint8))
            n = self.targets_per_image
            wh = (torch.rand(n, 2, generator=g, device=self.device) * 0.48 + 0.02)
            xy = torch.rand(n, 2, generator=g, device=self.device) * (1 - wh)
            scale = torch.tensor([w, h], device=self.device, dtype=torch.float32)
            cls = torch.randint(0, self.num_classes, (n, 1), generator=g, device=self.device).float()
            objects.append(torch.cat([xy * scale, wh * scale, cls], 1))
        flips = draw_flips(self.batch_size, hg) if train else None                         # val / test: Resize only
        ssr = draw_shift_scale_rotate(self.batch_size, hg) if train else None              # ShiftScaleRotate (p = 0.2)
        luts = photometric_luts(self.batch_size, hg) if train else None                    # RandomBrightnessContrast, RGBShift (p = 0.2)
        return gpu_collate(images, objects, self.img_size, flips, self.channels_last, luts, ssr)

    def _loader(self, offset):
        g = torch.Generator(device=self.device).manual_seed(self.seed + offset)
        hg = torch.Generator().manual_seed(self.seed + offset)
        for _ in range(self.steps_per_epoch):
            yield self._raw_batch(g, hg, offset == 0) if self.raw_frames else self._batch(g)

    def train_dataloader(self):
        return self._loader(0)

    def val_dataloader(self):
        return self._loader(1000)

    def test_dataloader(self):
        return self._loader(2000)
