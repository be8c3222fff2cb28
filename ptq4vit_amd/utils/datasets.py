"""ImageNet loaders for the accuracy side of the calibration path (BASELINE north_star: post-quantisation top-1).

Mirrors the contract of the reference's ``utils/datasets.py``: ``ViTImageNetLoaderGenerator(root, dataset_name,
train_batch_size, test_batch_size, num_workers, kwargs={"model": net})`` with ``train_loader()``, ``test_loader()``,
``val_loader()`` and -- the one the calibrator needs -- ``calib_loader(num, seed=3)`` (reference datasets.py:88-94: numpy's
legacy seeded permutation of the TRAIN set, first ``num`` indices, test-time transform, ONE batch of ``num`` images), the
``train`` / ``val`` sub-directories of the ImageNet root (datasets.py:224-233) and timm's evaluation transform of the model
(datasets.py:325-340: resize the short edge to ``floor(size / crop_pct)`` bicubic, centre crop, to-tensor, normalise).

torchvision and timm are not available offline, so the ImageFolder scan, the transform and the per-model data configuration
(what ``timm.data.resolve_data_config`` reads from the model's ``default_cfg``) are restated here on PIL + torch.  None of
this is on the timed hot path; it exists so that ``tools/eval_top1.py`` can produce the FP32 / quantised top-1 pair on a
machine that has ImageNet and a timm checkpoint.
"""
import math
import os

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset, Subset

IMAGENET_DEFAULT_MEAN, IMAGENET_DEFAULT_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
IMAGENET_INCEPTION_MEAN, IMAGENET_INCEPTION_STD = (0.5, 0.5, 0.5), (0.5, 0.5, 0.5)
IMG_EXTENSIONS = (".jpg", ".jpeg", ".png", ".ppm", ".bmp", ".pgm", ".tif", ".tiff", ".webp")


def data_config(model):
    """timm's evaluation data configuration of the named model (the `default_cfg` entries `resolve_data_config` reads):
    ViT checkpoints (augreg / original JAX weights) are normalised with mean = std = 0.5, DeiT and Swin with the ImageNet
    statistics; 224-pixel models crop 0.9 of the resized image, 384-pixel models the whole of it; all bicubic.
    `model`: a timm model name or a net built by utils.models.get_net (its `p4v_name`)."""
    name = model if isinstance(model, str) else getattr(model, "p4v_name", None)
    if name is None:
        raise ValueError("data_config: pass the model name (or a net built by ptq4vit_amd.utils.models.get_net)")
    from .models import input_size
    size = input_size(name)
    vit = name.startswith("vit_")
    return dict(input_size=(3, size, size), interpolation="bicubic", crop_pct=0.9 if size == 224 else 1.0,
                mean=IMAGENET_INCEPTION_MEAN if vit else IMAGENET_DEFAULT_MEAN,
                std=IMAGENET_INCEPTION_STD if vit else IMAGENET_DEFAULT_STD)


class EvalTransform:
    """Resize(short edge -> floor(size / crop_pct), bicubic) -> CenterCrop(size) -> ToTensor -> Normalize, on a PIL image:
    the arithmetic of torchvision's PIL backend (long edge = int(short' * long / short); crop offsets
    int(round((edge - size) / 2))), which is what timm's `create_transform(**config)` composes for evaluation."""

    def __init__(self, input_size, crop_pct, mean, std, interpolation="bicubic"):
        self.size = int(input_size[-1])
        self.scale_size = int(math.floor(self.size / crop_pct))
        self.mean = torch.tensor(mean, dtype=torch.float32).view(3, 1, 1)
        self.std = torch.tensor(std, dtype=torch.float32).view(3, 1, 1)
        self.interpolation = interpolation

    def __call__(self, img):
        from PIL import Image
        resample = {"bicubic": Image.BICUBIC, "bilinear": Image.BILINEAR, "nearest": Image.NEAREST}[self.interpolation]
        img = img.convert("RGB")
        w, h = img.size
        if (w <= h and w != self.scale_size) or (h < w and h != self.scale_size):
            if w <= h:
                nw, nh = self.scale_size, int(self.scale_size * h / w)
            else:
                nw, nh = int(self.scale_size * w / h), self.scale_size
            img = img.resize((nw, nh), resample)
            w, h = nw, nh
        left, top = int(round((w - self.size) / 2.0)), int(round((h - self.size) / 2.0))
        img = img.crop((left, top, left + self.size, top + self.size))
        x = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).to(torch.float32).div_(255.0)
        return (x - self.mean) / self.std


class ImageFolder(Dataset):
    """``root/<class>/<image>``: classes are the sorted sub-directory names, samples the sorted image files below each
    (recursively) -- torchvision's ``ImageFolder`` order, so that index i is the same image in both."""

    def __init__(self, root, transform=None):
        self.root, self.transform = root, transform
        if not os.path.isdir(root):
            raise FileNotFoundError(f"ImageFolder: {root} is not a directory")
        self.classes = sorted(d.name for d in os.scandir(root) if d.is_dir())
        if not self.classes:
            raise FileNotFoundError(f"ImageFolder: no class directories under {root}")
        self.class_to_idx = {c: i for i, c in enumerate(self.classes)}
        self.samples = []
        for c in self.classes:
            for dirpath, _, files in sorted(os.walk(os.path.join(root, c), followlinks=True)):
                for f in sorted(files):
                    if f.lower().endswith(IMG_EXTENSIONS):
                        self.samples.append((os.path.join(dirpath, f), self.class_to_idx[c]))
        if not self.samples:
            raise FileNotFoundError(f"ImageFolder: no images ({', '.join(IMG_EXTENSIONS)}) under {root}")
        self.targets = [t for _, t in self.samples]

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, i):
        from PIL import Image
        path, target = self.samples[i]
        with open(path, "rb") as fh:
            img = Image.open(fh)
            img = img.convert("RGB")
        return (self.transform(img) if self.transform is not None else img), target


class ViTImageNetLoaderGenerator:
    """Reference utils/datasets.py:325-340 (+ the LoaderGenerator base, :34-94).  The training-time augmentation of the
    reference's `train_loader` is not restated (nothing on the calibration path trains): `train_loader` serves the train set
    with the evaluation transform, shuffled."""

    def __init__(self, root, dataset_name="imagenet", train_batch_size=1, test_batch_size=1, num_workers=0, kwargs=None):
        kwargs = dict(kwargs or {})
        kwargs.update({"pin_memory": False})
        self.root, self.dataset_name = root, str.lower(dataset_name)
        self.train_batch_size, self.test_batch_size, self.num_workers, self.kwargs = train_batch_size, test_batch_size, num_workers, kwargs
        self._train_set = self._test_set = self._calib_set = None
        self.train_loader_kwargs = {"num_workers": num_workers, "pin_memory": kwargs.get("pin_memory", False),
                                    "drop_last": kwargs.get("drop_last", False)}
        self.test_loader_kwargs = dict(self.train_loader_kwargs)
        self.load()

    def load(self):
        model = self.kwargs.get("model", None)
        assert model is not None, "No model in ViTImageNetLoaderGenerator!"
        cfg = data_config(model)
        self.test_transform = EvalTransform(**cfg)
        self.train_transform = self.test_transform

    @property
    def train_set(self):
        if self._train_set is None:
            self._train_set = ImageFolder(os.path.join(self.root, "train"), self.train_transform)
        return self._train_set

    @property
    def test_set(self):
        if self._test_set is None:
            self._test_set = ImageFolder(os.path.join(self.root, "val"), self.test_transform)
        return self._test_set

    def train_loader(self):
        return DataLoader(self.train_set, batch_size=self.train_batch_size, shuffle=True, **self.train_loader_kwargs)

    def test_loader(self, shuffle=False, batch_size=None):
        return DataLoader(self.test_set, batch_size=self.test_batch_size if batch_size is None else batch_size,
                          shuffle=shuffle, **self.test_loader_kwargs)

    val_loader = test_loader

    def calib_indices(self, num=1024, seed=3):
        """The reference's choice of calibration images (datasets.py:89-91): legacy global numpy RNG seeded with `seed`,
        `permutation(len(train_set))[:num]`.  A local RandomState(seed) draws the same stream without touching the global one."""
        return np.random.RandomState(seed).permutation(len(self.train_set))[:num]

    def calib_loader(self, num=1024, seed=3):
        if self._calib_set is None:
            self._calib_set = Subset(ImageFolder(os.path.join(self.root, "train"), self.test_transform),
                                     [int(i) for i in self.calib_indices(num, seed)])
        return DataLoader(self._calib_set, batch_size=num, shuffle=False, **self.train_loader_kwargs)


def test_classification(net, test_loader, max_iteration=None, description=None, device=None):
    """Top-1 accuracy of `net` over `test_loader` (reference example/test_vit.py:26-45): argmax of the logits against the
    target, images and targets moved to the network's device.  Returns correct / total."""
    dev = device if device is not None else next(net.parameters()).device
    pos = tot = 0
    max_iteration = len(test_loader) if max_iteration is None else max_iteration
    with torch.no_grad():
        for i, (inp, target) in enumerate(test_loader, start=1):
            out = net(inp.to(dev))
            pos += int((out.argmax(1).cpu() == target.cpu()).sum())
            tot += inp.size(0)
            if i >= max_iteration:
                break
    acc = pos / max(tot, 1)
    if description:
        print(f"{description}: {acc:.5f} ({pos}/{tot})")
    return acc


test_classification.__test__ = False      # (not a pytest test, whatever module imports the name)
