#!/usr/bin/env python3
"""Golden vectors for the evaluation branch of ``GroundingDINO.forward``
(groundingdino_dual_zero_rep_branch.py:589-602 and ``dt_inference`` :634-675): the shrunken ZiRa slice of
gen_step_golden.py -- the REFERENCE's modules -- in eval mode (side branches return their twins' output, zero
loss), then the reference's own tensor arithmetic of ``dt_inference`` (sigmoid, top-k over query x class, box
gather, cxcywh -> xyxy, scale to the image size).  detectron2 is not installed here, so the two things the
reference takes from it are re-enacted from its documented behaviour: ``Instances`` / ``Boxes`` as plain tensors
and ``detector_postprocess`` (rescale to the requested output size, clip, drop empty boxes).
      python tests/golden/gen_eval_golden.py
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402
from gen_step_golden import CFG, SALT, SCALES, Slice, build_inputs  # noqa: E402

TOPK = 12                                  # select_box_nums_for_evaluation of the slice
IMAGE_SIZES = [(64, 80), (64, 64)]         # network input sizes (image 1 is narrower, as in build_inputs)
OUTPUT_SIZES = [(128, 160), (96, 96)]      # "height" / "width" of the batched inputs


def main():
    torch.set_num_threads(1)
    ref = ref_import.load()
    S = Slice(ref)
    for m in S.parts.values():
        m.eval()
    inp = build_inputs(torch.Generator().manual_seed(17))
    box_ops = __import__("groundingdino.util.box_ops", fromlist=["box_cxcywh_to_xyxy"])
    with torch.no_grad():
        out, loss_conv, loss_lin = S.outputs(inp)
        box_cls, box_pred = out["pred_logits"], out["pred_boxes"]
        # dt_inference (:651-673), verbatim arithmetic
        prob = box_cls.sigmoid()
        topk_values, topk_indexes = torch.topk(prob.view(box_cls.shape[0], -1), TOPK, dim=1)
        scores = topk_values
        topk_boxes = torch.div(topk_indexes, box_cls.shape[2], rounding_mode="floor")
        labels = topk_indexes % box_cls.shape[2]
        boxes = torch.gather(box_pred, 1, topk_boxes.unsqueeze(-1).repeat(1, 1, 4))
        results = []
        for s, lab, b, size, osize in zip(scores, labels, boxes, IMAGE_SIZES, OUTPUT_SIZES):
            xyxy = box_ops.box_cxcywh_to_xyxy(b).clone()
            xyxy[:, 0::2] *= size[1]                      # Boxes.scale(scale_x=w, scale_y=h)
            xyxy[:, 1::2] *= size[0]
            # detector_postprocess(results, height, width)
            xyxy[:, 0::2] *= osize[1] / size[1]
            xyxy[:, 1::2] *= osize[0] / size[0]
            xyxy[:, 0::2] = xyxy[:, 0::2].clamp(min=0, max=osize[1])
            xyxy[:, 1::2] = xyxy[:, 1::2].clamp(min=0, max=osize[0])
            keep = ((xyxy[:, 2] - xyxy[:, 0]) > 0) & ((xyxy[:, 3] - xyxy[:, 1]) > 0)
            results.append(dict(pred_boxes=xyxy[keep], scores=s[keep], pred_classes=lab[keep]))
    path = os.path.join(HERE, "eval_zira_slice.pt")
    torch.save(dict(cfg=CFG, salt=SALT, scales=SCALES, inputs=inp, topk=TOPK, image_sizes=IMAGE_SIZES,
                    output_sizes=OUTPUT_SIZES, pred_logits=box_cls, pred_boxes=box_pred,
                    zero_losses=(float(loss_conv), float(loss_lin)), results=results), path)
    print("eval_zira_slice %.1f KiB; kept" % (os.path.getsize(path) / 1024), [len(r["scores"]) for r in results],
          "zero losses", float(loss_conv), float(loss_lin), "top score", float(scores.max()))


if __name__ == "__main__":
    main()
