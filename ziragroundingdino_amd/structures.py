"""Minimal stand-ins for the detectron2 structures the model surface mentions
(``Instances``, ``Boxes``, ``ImageList``): the reference passes ground truth as
``x["instances"]`` with ``.image_size``, ``.gt_classes`` and ``.gt_boxes.tensor`` and returns
predictions as ``Instances`` with ``pred_boxes / scores / pred_classes``
(groundingdino_dual_zero_rep_branch.py:614-675).  detectron2 is not installed here; objects of
the real classes work too because only these attributes are touched."""
import torch


class Boxes:
    def __init__(self, tensor):
        self.tensor = tensor

    def to(self, device):
        return Boxes(self.tensor.to(device))

    def scale(self, scale_x, scale_y):
        self.tensor[:, 0::2] *= scale_x
        self.tensor[:, 1::2] *= scale_y

    def __len__(self):
        return self.tensor.shape[0]


class Instances:
    def __init__(self, image_size, **fields):
        self.image_size = tuple(image_size)
        for k, v in fields.items():
            setattr(self, k, v)

    def to(self, device):
        out = Instances(self.image_size)
        for k, v in self.__dict__.items():
            if k != "image_size":
                setattr(out, k, v.to(device) if hasattr(v, "to") else v)
        return out


class ImageList:
    def __init__(self, tensor, image_sizes):
        self.tensor = tensor
        self.image_sizes = image_sizes

    @staticmethod
    def from_tensors(tensors, size_divisibility=0):
        sizes = [(t.shape[-2], t.shape[-1]) for t in tensors]
        H, W = max(s[0] for s in sizes), max(s[1] for s in sizes)
        batch = tensors[0].new_zeros((len(tensors), tensors[0].shape[0], H, W))
        for t, b in zip(tensors, batch):
            b[:, :t.shape[-2], :t.shape[-1]].copy_(t)
        return ImageList(batch, sizes)
