import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_model_gpu import small_model
from ziragroundingdino_amd.train import ZiraTrainer, synthetic_batch
from ziragroundingdino_amd.structures import Boxes, Instances
model = small_model().train()
model.before_train(); model.use_frontend_graphs = False
a = synthetic_batch(1, 224, 320, n_categories=4, boxes_per_image=3, seed=1, device="cuda")[0]
b = synthetic_batch(1, 200, 272, n_categories=2, boxes_per_image=1, seed=2, device="cuda")[0]
mode = sys.argv[1] if len(sys.argv) > 1 else "both"
if mode in ("both", "nobox"):
    inst = b["instances"]
    b["instances"] = Instances(inst.image_size, gt_boxes=Boxes(inst.gt_boxes.tensor[:0]), gt_classes=inst.gt_classes[:0])
if mode == "samecap":
    b["captions"] = a["captions"]
data = [a, b]
print(a["captions"], "|", b["captions"])
def hook(name):
    def f(m, i, o):
        outs = o if isinstance(o, (tuple, list)) else [o]
        for j, t in enumerate(outs):
            if torch.is_tensor(t) and t.is_floating_point() and not torch.isfinite(t).all():
                print("non-finite output:", name, j, tuple(t.shape), "nan:", int(torch.isnan(t).sum()), "inf:", int(torch.isinf(t).sum()))
    return f
for n, m in model.named_modules():
    if n and n.count(".") <= 3:
        m.register_forward_hook(hook(n))
for flag in (True, False):
    for mod in model.modules():
        if hasattr(mod, "reassociate"):
            mod.reassociate = flag
    out = model(data)
    print("reassociate", flag, {k: float(v) for k, v in out.items() if not torch.isfinite(v)} or "all finite")
# --- inspect category masks and class logits
from ziragroundingdino_amd.utils import nested_tensor_from_tensor_list
caps = [d["captions"] for d in data]
text_dict, c2t, _ = model.encode_text(caps, torch.device("cuda"))
print("token mask:", text_dict["text_token_mask"].int().tolist())
for i, m in enumerate(c2t):
    print("image", i, "cate masks", tuple(m.shape), m.int().tolist())
