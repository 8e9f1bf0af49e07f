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

    def clip(self, box_size):
        """detectron2 ``Boxes.clip``: x into [0, w], y into [0, h], in place."""
        h, w = box_size
        self.tensor[:, 0::2] = self.tensor[:, 0::2].clamp(min=0, max=w)
        self.tensor[:, 1::2] = self.tensor[:, 1::2].clamp(min=0, max=h)

    def nonempty(self, threshold=0.0):
        """detectron2 ``Boxes.nonempty``: both sides longer than ``threshold``."""
        b = self.tensor
        return ((b[:, 2] - b[:, 0]) > threshold) & ((b[:, 3] - b[:, 1]) > threshold)

    def __getitem__(self, item):
        return Boxes(self.tensor[item].view(-1, 4) if isinstance(item, int) else self.tensor[item])

    def __len__(self):
        return self.tensor.shape[0]


def detector_postprocess(results, output_height, output_width):
    """What detectron2's ``detector_postprocess`` does to box predictions (the reference calls it on every
    image, groundingdino_dual_zero_rep_branch.py:599): boxes rescaled from the network's input size to the
    requested output size, clipped to it, empty boxes dropped (with their scores / classes)."""
    scale_x, scale_y = output_width / results.image_size[1], output_height / results.image_size[0]
    out = Instances((output_height, output_width), **{k: v for k, v in results.__dict__.items() if k != "image_size"})
    boxes = out.pred_boxes
    boxes.scale(scale_x, scale_y)
    boxes.clip(out.image_size)
    return out[boxes.nonempty()]


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

    def __getitem__(self, item):
        """Every field indexed the same way (detectron2 ``Instances.__getitem__``)."""
        out = Instances(self.image_size)
        for k, v in self.__dict__.items():
            if k != "image_size":
                setattr(out, k, v[item])
        return out

    def __len__(self):
        for k, v in self.__dict__.items():
            if k != "image_size":
                return len(v)
        return 0


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
