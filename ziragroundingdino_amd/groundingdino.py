"""ZiRa GroundingDINO model -- drop-in for the reference's
``groundingdino/models/GroundingDINO/groundingdino_dual_zero_rep_branch.py`` (class
``GroundingDINO`` :137-745, builder :748-827, registry name ``dualzerorepbranchgroundingdino``).

Same surface: ``model(batched_inputs)`` with ``[{"image": uint8/float [3,H,W], "captions":
"cat . dog .", "instances": Instances}]`` returns the weighted loss dict in training mode
(``loss_class/_bbox/_giou`` + ``_0.._4`` + ``_enc``, ``loss_conv_adapter``,
``loss_linear_adapter``) and ``[{"instances": Instances}]`` in eval mode; the hooks
``before_train`` / ``after_train`` / ``add_cls_prompt`` / ``load_state_dict`` and the
parameter names (``input_proj_conv_adapter.N.*``, ``rep_linear_adapter.*``, ``feat_map``,
``input_proj``, ``bbox_embed``, ``transformer.*``, ``backbone.0.*``, ``bert.*``) are the
reference's, so its checkpoints and its training driver work unchanged.

What runs where: both side branches go through the fused RSB epilogue kernels (rsb.py), all
12 deformable-attention calls through the gfx950 MSDA kernels (ms_deform_attn.py); the frozen
backbone and text encoder are stock PyTorch-ROCm and are evaluated without building an
autograd graph (nothing upstream of the side branches needs gradients).
"""
import copy
import os
import random
from typing import List

import torch
import torch.nn.functional as F
from torch import nn

from .backbone import build_backbone
from .bert import BertConfig, BertModel, SimpleTokenizer
from .box_ops import box_cxcywh_to_xyxy, box_xyxy_to_cxcywh
from .criterion import build_criterion
from .dense import conv_module_as_gemm
from .graphs import GraphedNoGrad, GraphedTransformer
from .rsb import RepZeroConv2d, RepZeroLinear
from .structures import Boxes, ImageList, Instances
from .text_masks import generate_masks_with_special_tokens_and_transfer_map
from .transformer import build_transformer
from .utils import (MLP, ContrastiveEmbed, NestedTensor, box_head, inverse_sigmoid,
                    nested_tensor_from_tensor_list, recover_to_cls_logits)


class _Registry:
    """name -> builder, as groundingdino/models/registry.py ``MODULE_BUILD_FUNCS``."""

    def __init__(self):
        self._funcs = {}

    def registe_with_name(self, module_name=None, force=False):
        def deco(fn):
            name = module_name or fn.__name__
            if name in self._funcs and not force:
                raise KeyError("%s is already registered" % name)
            self._funcs[name] = fn
            return fn
        return deco

    def get(self, name):
        return self._funcs[name]

    def __contains__(self, name):
        return name in self._funcs


MODULE_BUILD_FUNCS = _Registry()


class GroundingDINO(nn.Module):
    def __init__(self, backbone, transformer, num_queries, aux_loss=False, iter_update=False,
                 query_dim=2, num_feature_levels=1, nheads=8, two_stage_type="no",
                 dec_pred_bbox_embed_share=True, two_stage_class_embed_share=True,
                 two_stage_bbox_embed_share=True, num_patterns=0, dn_number=100,
                 dn_box_noise_scale=0.4, dn_label_noise_ratio=0.5, dn_labelbook_size=100,
                 text_encoder_type="bert-base-uncased", sub_sentence_present=True, max_text_len=256,
                 criterion=None, pixel_mean: List[float] = [123.675, 116.280, 103.530],
                 pixel_std: List[float] = [123.675, 116.280, 103.530], device="cuda",
                 select_box_nums_for_evaluation=200, freeze_all=False, loss_adapter_weight=0.1,
                 use_cet=False, use_prompt_memory=False, num_select_prompt=200,
                 use_zero_inter_loss=True, use_add_names=False, use_bert_tuning=False,
                 use_cls_linear=False, use_prompt_tuning=False, use_prompt_memory_output=True,
                 use_project_tuning=False, use_project_adapter=True,
                 use_zero_inter_loss_for_conv=True, use_learned_names=False,
                 bert=None, tokenizer=None, side_branch="rep", **_unused):
        super().__init__()
        assert query_dim == 4 and iter_update, "GroundingDINO uses 4-d queries with iterative update"
        assert not use_cls_linear, "use_cls_linear (linear probing ablation) is outside the ZiRa path"
        assert two_stage_type in ["no", "standard"]
        self.num_queries = num_queries
        self.transformer = transformer
        self.hidden_dim = hidden_dim = transformer.d_model
        self.num_feature_levels = num_feature_levels
        self.nheads = nheads
        self.max_text_len = max_text_len
        self.sub_sentence_present = sub_sentence_present
        self.query_dim = query_dim
        self.num_patterns = num_patterns
        self.dn_number, self.dn_box_noise_scale = dn_number, dn_box_noise_scale
        self.dn_label_noise_ratio, self.dn_labelbook_size = dn_label_noise_ratio, dn_labelbook_size

        # text encoder (frozen); tokenizer / pretrained weights can be injected by the caller
        self.tokenizer = tokenizer if tokenizer is not None else SimpleTokenizer()
        self.bert = bert if bert is not None else BertModel(BertConfig())
        self.bert.pooler.dense.weight.requires_grad_(False)
        self.bert.pooler.dense.bias.requires_grad_(False)
        self.feat_map = nn.Linear(self.bert.config.hidden_size, hidden_dim, bias=True)
        nn.init.constant_(self.feat_map.bias.data, 0)
        nn.init.xavier_uniform_(self.feat_map.weight.data)

        self.learned_classes = []
        self.use_cet = use_cet
        self.use_prompt_memory = use_prompt_memory
        self.use_zero_inter_loss = use_zero_inter_loss
        self.prompt_memory_pool = nn.ParameterDict()
        self.num_select_prompt = num_select_prompt
        self.use_prompt_tuning = use_prompt_tuning
        self.use_learned_names = use_learned_names
        self.use_prompt_memory_output = use_prompt_memory_output
        # "rep": the ZiRa model (groundingdino_dual_zero_rep_branch.py); "multilayer": its multilayer-branch
        # ablation (groundingdino_dual_zero_rep_multilayer_branch.py, rsb_multilayer.py)
        assert side_branch in ("rep", "multilayer")
        self.side_branch = side_branch
        if side_branch == "multilayer":  # unconditional language branch, its own name (reference :322-323)
            from . import rsb_multilayer

            self.rep_language_adapter = rsb_multilayer.RepZeroLinear(self.bert.config.hidden_size, hidden_dim)
            conv_branch = rsb_multilayer.RepZeroConv2dGN
        else:
            conv_branch = RepZeroConv2d
            if use_cet:  # RSB #1: language side branch beside feat_map
                self.rep_linear_adapter = RepZeroLinear(self.bert.config.hidden_size, hidden_dim)
        self.specical_tokens = self.tokenizer.convert_tokens_to_ids(["[CLS]", "[SEP]", ".", "?"])

        # input projections (frozen) and RSB #2: one zero-initialised conv branch beside each
        chans = list(backbone.num_channels)
        proj, adapters = [], []
        in_channels = chans[-1]
        if num_feature_levels > 1:
            for c in chans:
                proj.append(nn.Sequential(nn.Conv2d(c, hidden_dim, kernel_size=1), nn.GroupNorm(32, hidden_dim)))
                adapters.append(conv_branch(c, hidden_dim, kernel_size=1))
            for _ in range(num_feature_levels - len(chans)):
                proj.append(nn.Sequential(nn.Conv2d(in_channels, hidden_dim, kernel_size=3, stride=2, padding=1),
                                          nn.GroupNorm(32, hidden_dim)))
                adapters.append(conv_branch(in_channels, hidden_dim, kernel_size=3, stride=2, padding=1))
                in_channels = hidden_dim
        else:
            assert two_stage_type == "no"
            proj.append(nn.Sequential(nn.Conv2d(chans[-1], hidden_dim, kernel_size=1), nn.GroupNorm(32, hidden_dim)))
            adapters.append(conv_branch(chans[-1], hidden_dim, kernel_size=1))
        self.input_proj = nn.ModuleList(proj)
        self.use_project_adapter = use_project_adapter
        self.use_zero_inter_loss_for_conv = use_zero_inter_loss_for_conv
        if use_project_adapter:
            self.input_proj_conv_adapter = nn.ModuleList(adapters)

        self.backbone = backbone
        self.aux_loss = aux_loss
        self.box_pred_damping = None
        self.iter_update = iter_update

        # prediction heads, shared across decoder layers
        self.dec_pred_bbox_embed_share = dec_pred_bbox_embed_share
        _class_embed = ContrastiveEmbed(max_text_len=max_text_len)
        _bbox_embed = MLP(hidden_dim, hidden_dim, 4, 3)
        nn.init.constant_(_bbox_embed.layers[-1].weight.data, 0)
        nn.init.constant_(_bbox_embed.layers[-1].bias.data, 0)
        n_dec = transformer.num_decoder_layers
        boxes = [_bbox_embed if dec_pred_bbox_embed_share else copy.deepcopy(_bbox_embed) for _ in range(n_dec)]
        self.bbox_embed = nn.ModuleList(boxes)
        self.class_embed = nn.ModuleList([_class_embed for _ in range(n_dec)])
        self.transformer.decoder.bbox_embed = self.bbox_embed
        self.transformer.decoder.class_embed = self.class_embed
        self.two_stage_type = two_stage_type
        if two_stage_type != "no":
            if two_stage_bbox_embed_share:
                assert dec_pred_bbox_embed_share
                self.transformer.enc_out_bbox_embed = _bbox_embed
            else:
                self.transformer.enc_out_bbox_embed = copy.deepcopy(_bbox_embed)
            if two_stage_class_embed_share:
                assert dec_pred_bbox_embed_share
                self.transformer.enc_out_class_embed = _class_embed
            else:
                self.transformer.enc_out_class_embed = copy.deepcopy(_class_embed)
            self.refpoint_embed = None

        self.criterion = criterion
        self.pixel_mean, self.pixel_std = pixel_mean, pixel_std
        self._pixel_stats = {}
        self._text_cache = {}
        self.device = device
        self._reset_parameters()
        self.use_add_names = use_add_names
        self.select_box_nums_for_evaluation = select_box_nums_for_evaluation
        self.loss_adapter_weight = loss_adapter_weight
        self.freeze_all = freeze_all
        self.use_bert_tuning = use_bert_tuning
        self.use_cls_linear = use_cls_linear
        self.use_project_tuning = use_project_tuning
        # hipGraph replay of the frozen front end (GPU only; see graphs.py)
        self.use_frontend_graphs = True
        self._graphed_backbone = GraphedNoGrad(self._backbone_tensors, modules=(self.backbone,))
        self._graphed_bert = GraphedNoGrad(self._bert_hidden, modules=(self.bert,))
        # hipGraph replay of transformer forward + backward (training mode on the GPU with the transformer
        # frozen -- every ZiRa task; up to two input signatures, further ones run eagerly); see
        # graphs.GraphedTransformer.  On by default since round 3 (2000-step soak over rotating minibatches,
        # scripts/soak_graph.py; two ranks sharing a GPU, tests/test_model_gpu.py); False launches eagerly.
        # As with any torch.cuda.make_graphed_callables callable: ONE backward per forward (no retain_graph re-runs).
        self.use_transformer_graph = True
        self.overlap_text_and_image = True   # frozen BERT replayed on a side stream while the frozen Swin runs
        self._text_stream = None
        self._prefetch_stream = None
        self._graphed_transformer = GraphedTransformer(self.transformer)

    def _backbone_tensors(self, images, mask):
        """tensor-in / tensor-out view of the backbone for graph capture."""
        features, poss = self.backbone(NestedTensor(images, mask))
        return [f.tensors for f in features], [f.mask for f in features], poss

    def _bert_hidden(self, enc_in):
        return self.bert(**enc_in)["last_hidden_state"]

    def run_backbone(self, samples):
        """features (list of NestedTensor) and position encodings of the frozen backbone."""
        if self._frozen(self.backbone):
            if self.use_frontend_graphs and samples.tensors.is_cuda:
                ft, fm, poss = self._graphed_backbone(samples.tensors, samples.mask)
            else:
                with torch.no_grad():
                    ft, fm, poss = self._backbone_tensors(samples.tensors, samples.mask)
            return [NestedTensor(t, m) for t, m in zip(ft, fm)], list(poss)
        return self.backbone(samples)

    def _reset_parameters(self):
        for proj in self.input_proj:
            nn.init.xavier_uniform_(proj[0].weight, gain=1)
            nn.init.constant_(proj[0].bias, 0)

    def init_ref_points(self, use_num_queries):
        self.refpoint_embed = nn.Embedding(use_num_queries, self.query_dim)

    # ---- pieces of forward ----------------------------------------------------------------
    @staticmethod
    def _frozen(module):
        """No parameter of ``module`` requires grad.  Checked on every forward (the answer may change
        between calls), but on a remembered flat list: ``module.parameters()`` walks the module tree
        (0.7 ms for Swin / BERT), reading ``requires_grad`` of 370 tensors does not."""
        params = getattr(module, "_zira_param_list", None)
        if params is None:
            params = list(module.parameters())
            object.__setattr__(module, "_zira_param_list", params)
        return not any(p.requires_grad for p in params)

    def _loss_weights(self, name, suffixes, weight_dict, like):
        key = (name, tuple(suffixes), tuple(weight_dict[name + suf] for suf in suffixes), like.device, like.dtype)
        cache = self.__dict__.setdefault("_loss_weight_cache", {})
        w = cache.get(key)
        if w is None:
            w = cache[key] = torch.tensor(key[2], dtype=like.dtype).to(like.device)
        return w

    def _project_level(self, l, feat):
        """GroupNorm(input_proj conv + side branch); returns (src, zero-interference loss | None)."""
        from .dense import group_norm_frozen, group_norm_supported
        main = conv_module_as_gemm(self.input_proj[l][0], feat)  # GEMM library instead of MIOpen
        gn = self.input_proj[l][1]
        if not self.use_project_adapter:
            return (group_norm_frozen(main, gn) if group_norm_supported(main, gn) else gn(main)), None
        branch, zero_loss = self.input_proj_conv_adapter[l](feat)
        if self.side_branch == "multilayer":   # the branch carries its own GroupNorm (reference :575-576)
            return (group_norm_frozen(main, gn) if group_norm_supported(main, gn) else gn(main)) + branch, zero_loss
        if group_norm_supported(main, gn, branch):   # the sum is formed inside the kernel (csrc/groupnorm.hip)
            return group_norm_frozen(main, gn, branch), zero_loss
        return gn(main + branch), zero_loss

    def encode_text(self, captions, device, defer=False, hidden_only=False):
        # Tokenisation, the sub-sentence masks and their upload are a pure function of the caption
        # strings, and a task trains on one category list for thousands of steps: remembered per
        # (captions, device) instead of redone every step (2.5 ms of host time, five blocking copies).
        key = (tuple(captions), str(device))
        cached = self._text_cache.get(key)
        if cached is None:
            tokenized = self.tokenizer(captions, padding="longest", return_tensors="pt").to(device)
            masks, position_ids, cate_to_token_mask_list = generate_masks_with_special_tokens_and_transfer_map(
                tokenized, self.specical_tokens, self.tokenizer)
            L = self.max_text_len
            if masks.shape[1] > L:
                masks = masks[:, :L, :L]
                position_ids = position_ids[:, :L]
                for k in ("input_ids", "attention_mask", "token_type_ids"):
                    tokenized[k] = tokenized[k][:, :L]
            if self.sub_sentence_present:
                enc_in = {k: v for k, v in tokenized.items() if k != "attention_mask"}
                enc_in["attention_mask"] = masks
                enc_in["position_ids"] = position_ids
            else:
                enc_in = tokenized
            if len(self._text_cache) >= 64:
                self._text_cache.clear()
            cached = self._text_cache[key] = (tokenized, masks, position_ids, cate_to_token_mask_list, dict(enc_in))
        tokenized, masks, position_ids, cate_to_token_mask_list, enc_in = cached
        enc_in = dict(enc_in)
        side = None
        if self._frozen(self.bert):
            if self.use_frontend_graphs and enc_in["input_ids"].is_cuda:
                if defer and self.overlap_text_and_image:
                    # the frozen BERT (a graph of ~300 launch-bound kernels) runs beside the frozen Swin: its replay
                    # goes to a side stream here and is joined in finish(), after the caller has queued the backbone
                    if self._text_stream is None:
                        self._text_stream = torch.cuda.Stream(device=device)
                    side = self._text_stream
                    side.wait_stream(torch.cuda.current_stream(device))
                    with torch.cuda.stream(side):
                        hidden = self._graphed_bert(enc_in)
                else:
                    hidden = self._graphed_bert(enc_in)
            else:
                with torch.no_grad():
                    hidden = self._bert_hidden(enc_in)
        else:
            hidden = self._bert_hidden(enc_in)

        def finish():
            if side is not None:
                torch.cuda.current_stream(device).wait_stream(side)
                hidden.record_stream(torch.cuda.current_stream(device))
            elif hidden_only and hidden.is_cuda:   # computed on a prefetch stream; the caller has waited for it
                hidden.record_stream(torch.cuda.current_stream(device))
            text_dict, loss_linear_adapter = self.project_text(
                hidden, tokenized["attention_mask"].bool(), position_ids, masks)
            return text_dict, cate_to_token_mask_list, loss_linear_adapter

        return (finish, cate_to_token_mask_list) if defer else finish()

    def project_text(self, bert_hidden, text_token_mask, position_ids, text_self_attention_masks):
        """feat_map + language side branch (reference :459-476): BERT states -> text_dict."""
        L = self.max_text_len
        encoded_text = self.feat_map(bert_hidden)
        loss_linear_adapter = None
        if self.side_branch == "multilayer":
            rep_out, loss_linear_adapter = self.rep_language_adapter(bert_hidden)
            encoded_text = rep_out + encoded_text
        elif self.use_cet:
            rep_out, loss_linear_adapter = self.rep_linear_adapter(bert_hidden)
            encoded_text = rep_out + encoded_text
        if encoded_text.shape[1] > L:
            encoded_text, text_token_mask = encoded_text[:, :L, :], text_token_mask[:, :L]
            position_ids = position_ids[:, :L]
            text_self_attention_masks = text_self_attention_masks[:, :L, :L]
        text_dict = {"encoded_text": encoded_text, "text_token_mask": text_token_mask,
                     "position_ids": position_ids,
                     "text_self_attention_masks": text_self_attention_masks}
        return text_dict, loss_linear_adapter

    def _captions(self, batched_inputs):
        captions = [x["captions"] for x in batched_inputs]
        names_list = [x["captions"][:-1].split(".") for x in batched_inputs]
        if (self.use_add_names and not self.training) or (self.use_learned_names and self.training):
            extra = [c for c in self.learned_classes if c not in names_list[0]]
            if self.training and len(extra) >= self.num_select_prompt:
                extra = random.sample(extra, self.num_select_prompt)
            for i, (caption, names) in enumerate(zip(captions, names_list)):
                names_list[i] = names + extra
                captions[i] = caption + ".".join(extra)
                if not captions[i].endswith("."):
                    captions[i] += "."
        return captions, names_list

    def can_prefetch_frontend(self):
        return (self.training and self.use_frontend_graphs and self._frozen(self.backbone) and self._frozen(self.bert)
                and next(self.parameters()).is_cuda)

    def prefetch_frontend(self, batched_inputs):
        """Queue the FROZEN front end (image normalisation, Swin, position encodings, BERT) of a minibatch on a second
        stream and return a handle for ``forward(batched_inputs, frontend=handle)``.  Nothing in it depends on the
        trainable weights, so a trainer can queue the next minibatch's front end while the current step's launch-bound
        phases (fusion blocks, decoder, criterion, backward) leave most of the GPU idle.  The work per step is the
        same; only its place in time changes."""
        assert self.can_prefetch_frontend()
        dev = self.device if isinstance(self.device, torch.device) else torch.device(self.device)
        cur = torch.cuda.current_stream(dev)
        if self._prefetch_stream is None:
            # default (= lowest) priority.  Measured: a high-priority prefetch stream costs 20 % (36 vs 45 images/s), and
            # putting the step itself on a high-priority stream instead loses as well (39-40)
            self._prefetch_stream = torch.cuda.Stream(device=dev)
        side = self._prefetch_stream
        side.wait_stream(cur)          # (the inputs may have been produced on the current stream)
        with torch.cuda.stream(side), torch.no_grad():
            images = self.preprocess_image(batched_inputs)
            samples = nested_tensor_from_tensor_list(images)
            captions, names_list = self._captions(batched_inputs)
            overlap, self.overlap_text_and_image = self.overlap_text_and_image, False   # already off the main stream
            try:
                finish_text, cate_list = self.encode_text(captions, samples.device, defer=True, hidden_only=True)
            finally:
                self.overlap_text_and_image = overlap
            features, poss = self.run_backbone(samples)
            done = torch.cuda.Event()
            done.record(side)
        return {"inputs": batched_inputs, "images": images, "samples": samples, "names_list": names_list,
                "finish_text": finish_text, "cate_list": cate_list, "features": features, "poss": poss, "done": done,
                "stream": side}

    def forward(self, batched_inputs, frontend=None, **kw):
        if frontend is not None and frontend["inputs"] is batched_inputs:
            cur = torch.cuda.current_stream(frontend["samples"].tensors.device)
            cur.wait_event(frontend["done"])
            images, samples, names_list = frontend["images"], frontend["samples"], frontend["names_list"]
            features, poss, cate_to_token_mask_list = frontend["features"], frontend["poss"], frontend["cate_list"]
            for t in [samples.tensors, samples.mask] + [f.tensors for f in features] + [f.mask for f in features] + list(poss):
                if torch.is_tensor(t):
                    t.record_stream(cur)
            targets = None
            if self.training:
                gt_instances = [x["instances"].to(self.device) for x in batched_inputs]
                targets = self.prepare_targets(gt_instances, cate_to_token_mask_list, names_list)
            text_dict, cate_to_token_mask_list, loss_linear_adapter = frontend["finish_text"]()
            return self._forward_rest(batched_inputs, images, samples, features, poss, text_dict, cate_to_token_mask_list,
                                      loss_linear_adapter, targets)
        if self._prefetch_stream is not None:
            # A prefetch that this call does not consume (a handle for another minibatch) may still be running: it
            # replays the SAME front-end graphs, whose static input / output buffers this call is about to use.
            torch.cuda.current_stream(self._prefetch_stream.device).wait_stream(self._prefetch_stream)
        images = self.preprocess_image(batched_inputs)
        samples = nested_tensor_from_tensor_list(images)
        captions, names_list = self._captions(batched_inputs)
        finish_text, cate_to_token_mask_list = self.encode_text(captions, samples.device, defer=True)

        targets = None
        if self.training:
            gt_instances = [x["instances"].to(self.device) for x in batched_inputs]
            targets = self.prepare_targets(gt_instances, cate_to_token_mask_list, names_list)

        features, poss = self.run_backbone(samples)
        text_dict, cate_to_token_mask_list, loss_linear_adapter = finish_text()
        return self._forward_rest(batched_inputs, images, samples, features, poss, text_dict, cate_to_token_mask_list,
                                  loss_linear_adapter, targets)

    def _forward_rest(self, batched_inputs, images, samples, features, poss, text_dict, cate_to_token_mask_list,
                      loss_linear_adapter, targets):
        out_or_loss = self.forward_features(features, poss, samples.mask, text_dict,
                                            cate_to_token_mask_list, loss_linear_adapter, targets,
                                            no_padding=getattr(samples, "no_padding", False))
        if self.training:
            return out_or_loss
        out = out_or_loss
        return self.postprocess(out["pred_logits"], out["pred_boxes"], batched_inputs, images.image_sizes)

    def postprocess(self, box_cls, box_pred, batched_inputs, image_sizes):
        """The evaluation tail of ``forward`` (reference :589-602): top-k detections per image, rescaled to the
        requested output size, clipped, empty boxes dropped."""
        from .structures import detector_postprocess

        results = self.dt_inference(box_cls, box_pred, image_sizes)
        processed = []
        for r, inp, image_size in zip(results, batched_inputs, image_sizes):
            height, width = inp.get("height", image_size[0]), inp.get("width", image_size[1])
            processed.append({"instances": detector_postprocess(r, height, width)})
        return processed

    def forward_features(self, features, poss, samples_mask, text_dict, cate_to_token_mask_list,
                         loss_linear_adapter=None, targets=None, no_padding=False):
        """Everything downstream of the frozen backbone / text encoder: input projections with
        the vision side branches, transformer, heads, and in training mode the criterion
        (reference :483-587).  ``features``: list of NestedTensor, ``poss``: their position
        encodings, ``samples_mask``: [B,H,W] padding mask of the input images."""
        poss = list(poss)
        srcs, masks, loss_conv_adapter = [], [], None

        def add_zero_loss(z):
            nonlocal loss_conv_adapter
            if z is not None:
                loss_conv_adapter = z if loss_conv_adapter is None else loss_conv_adapter + z

        for l, feat in enumerate(features):
            src, mask = feat.decompose()
            src, z = self._project_level(l, src)
            add_zero_loss(z)
            srcs.append(src)
            masks.append(mask)
        for l in range(len(srcs), self.num_feature_levels):
            inp = features[-1].tensors if l == len(features) else srcs[-1]
            src, z = self._project_level(l, inp)
            add_zero_loss(z)
            mask = F.interpolate(samples_mask[None].float(), size=src.shape[-2:]).to(torch.bool)[0]
            if len(poss) <= l:  # (callers that start at feature level may pass it themselves)
                poss.append(self.backbone[1](NestedTensor(src, mask)).to(src.dtype))
            srcs.append(src)
            masks.append(mask)

        # (capture under autocast needs the weight-cast cache off, as ZiraTrainer sets it: with the cache on, eager launches)
        autocast_cache = torch.is_autocast_enabled("cuda") and torch.is_autocast_cache_enabled()
        if (self.use_transformer_graph and self.training and srcs[0].is_cuda and not autocast_cache
                and self._frozen(self.transformer) and torch.is_grad_enabled()):
            hs, reference, hs_enc, ref_enc, init_box_proposal = self._graphed_transformer(
                srcs, masks, poss, text_dict, no_padding=no_padding)
        else:
            hs, reference, hs_enc, ref_enc, init_box_proposal, _ = self.transformer(
                srcs, masks, None, poss, None, None, text_dict, no_padding=no_padding)

        # Heads (reference groundingdino_dual_zero_rep_branch.py:559-583).  The box MLP is shared by
        # the decoder layers and the classifier is parameter-free, so all layers -- and in training
        # the encoder proposals -- go through them as ONE stacked tensor instead of layer by layer.
        n_dec = len(hs)
        shared_heads = (all(m is self.bbox_embed[0] for m in self.bbox_embed)
                        and all(m is self.class_embed[0] for m in self.class_embed))
        enc_cls = getattr(self.transformer, "enc_out_class_embed", None)
        with_enc = (self.training and hs_enc is not None and shared_heads
                    and isinstance(enc_cls, ContrastiveEmbed)          # parameter-free: a deep copy
                    and isinstance(self.class_embed[0], ContrastiveEmbed)  # computes the same thing
                    and enc_cls.max_text_len == self.class_embed[0].max_text_len
                    and hs_enc[-1].shape == hs[0].shape)
        if shared_heads:
            hs_all = torch.stack(list(hs))                                   # [L, B, Q, d]
            ref_all = torch.stack(list(reference[:-1]))
            outputs_coord_list = box_head(self.bbox_embed[0](hs_all), ref_all)
            cls_in = torch.cat([hs_all, hs_enc[-1][None]]) if with_enc else hs_all
            cls_all = recover_to_cls_logits(self.class_embed[0](cls_in, text_dict),
                                            cate_to_token_mask_list, for_fill=-100.0)
            outputs_class = cls_all[:n_dec]
        else:
            outputs_coord_list = []
            for layer_ref_sig, layer_bbox_embed, layer_hs in zip(reference[:-1], self.bbox_embed, hs):
                unsig = layer_bbox_embed(layer_hs) + inverse_sigmoid(layer_ref_sig)
                outputs_coord_list.append(unsig.sigmoid())
            outputs_coord_list = torch.stack(outputs_coord_list)
            outputs_class = torch.stack([
                recover_to_cls_logits(layer_cls_embed(layer_hs, text_dict), cate_to_token_mask_list, for_fill=-100.0)
                for layer_cls_embed, layer_hs in zip(self.class_embed, hs)])
        out = {"pred_logits": outputs_class[-1], "pred_boxes": outputs_coord_list[-1],
               "cate_to_token_mask_list": cate_to_token_mask_list}

        if self.training:
            if self.aux_loss:
                out["aux_outputs"] = [{"pred_logits": a, "pred_boxes": b}
                                      for a, b in zip(outputs_class[:-1], outputs_coord_list[:-1])]
            if hs_enc is not None:
                if with_enc:
                    interm_class = cls_all[n_dec]
                else:
                    interm_class = self.transformer.enc_out_class_embed(hs_enc[-1], text_dict)
                    interm_class = recover_to_cls_logits(interm_class, cate_to_token_mask_list, for_fill=-100.0)
                out["enc_outputs"] = {"pred_logits": interm_class, "pred_boxes": ref_enc[-1]}
            if with_enc and self.aux_loss:     # the criterion takes all 7 sets as stacked tensors
                out["stacked"] = (cls_all, torch.cat([outputs_coord_list, ref_enc[-1][None]]),
                                  ["_%d" % i for i in range(n_dec - 1)] + ["", "_enc"])
            assert targets is not None and self.criterion is not None
            loss_dict = self.criterion(out, targets)
            weight_dict = self.criterion.weight_dict
            total = None
            stacked = getattr(loss_dict, "stacked", None)
            if stacked and all(name + suf in weight_dict for name, (_, sufs) in stacked.items() for suf in sufs):
                # the criterion made its entries from one vector per loss type: weight the vectors (one multiply per
                # type), hand out their elements under the reference's keys, and keep the sum for the trainer
                for name, (vec, sufs) in stacked.items():
                    w = self._loss_weights(name, sufs, weight_dict, vec)
                    weighted = vec * w
                    for s_, suf in enumerate(sufs):
                        loss_dict[name + suf] = weighted[s_]
                    total = weighted.sum() if total is None else total + weighted.sum()
            else:
                for k in loss_dict.keys():
                    if k in weight_dict:
                        loss_dict[k] = loss_dict[k] * weight_dict[k]
            if self.use_project_adapter and self.use_zero_inter_loss_for_conv:
                loss_dict["loss_conv_adapter"] = loss_conv_adapter * self.loss_adapter_weight
                total = None if total is None else total + loss_dict["loss_conv_adapter"].reshape(())
            if self.use_cet and self.use_zero_inter_loss:
                key = "loss_language_adapter" if self.side_branch == "multilayer" else "loss_linear_adapter"
                loss_dict[key] = loss_linear_adapter * self.loss_adapter_weight
                total = None if total is None else total + loss_dict[key].reshape(())
            if total is not None:
                loss_dict.total = total        # == sum(loss_dict.values()); ZiraTrainer back-propagates this one
            return loss_dict
        return out

    # ---- helpers with the reference's names ---------------------------------------------------
    def prepare_targets(self, targets, cate_to_token_mask_list, names_list):
        new_targets = []
        for t in targets:
            h, w = t.image_size
            scale = self._box_scale(w, h)
            new_targets.append({"labels": t.gt_classes,
                                "boxes": box_xyxy_to_cxcywh(t.gt_boxes.tensor / scale)})
        return new_targets

    def _box_scale(self, w, h):
        key = (int(w), int(h), str(self.device))     # (w, h, w, h) per image size, uploaded once
        t = self._pixel_stats.get(key)
        if t is None:
            t = self._pixel_stats[key] = torch.as_tensor([w, h, w, h], dtype=torch.float, device=self.device)
        return t

    def preprocess_image(self, batched_inputs):
        images = [self.normalizer(x["image"].to(self.device)) for x in batched_inputs]
        return ImageList.from_tensors(images)

    def normalizer(self, x):
        # the two constants are uploaded once per device (a host list -> device tensor is a
        # blocking copy: 4 of them per step cost 5 ms of host time here)
        key = (x.device, x.dtype)
        cached = self._pixel_stats.get(key)
        if cached is None:
            cached = self._pixel_stats[key] = (
                torch.tensor(self.pixel_mean, device=x.device, dtype=x.dtype).view(3, 1, 1),
                torch.tensor(self.pixel_std, device=x.device, dtype=x.dtype).view(3, 1, 1))
        return (x - cached[0]) / cached[1]

    def dt_inference(self, box_cls, box_pred, image_sizes):
        """sigmoid -> top-k over (query x class) -> boxes in absolute xyxy (reference :634-675)."""
        assert len(box_cls) == len(image_sizes)
        prob = box_cls.sigmoid()
        scores, topk_indexes = torch.topk(prob.view(box_cls.shape[0], -1),
                                          self.select_box_nums_for_evaluation, dim=1)
        topk_boxes = torch.div(topk_indexes, box_cls.shape[2], rounding_mode="floor")
        labels = topk_indexes % box_cls.shape[2]
        boxes = torch.gather(box_pred, 1, topk_boxes.unsqueeze(-1).repeat(1, 1, 4))
        results = []
        for s, lab, b, image_size in zip(scores, labels, boxes, image_sizes):
            r = Instances(image_size)
            r.pred_boxes = Boxes(box_cxcywh_to_xyxy(b))
            r.pred_boxes.scale(scale_x=image_size[1], scale_y=image_size[0])
            r.scores = s
            r.pred_classes = lab
            results.append(r)
        return results

    def unfreeze_module_(self, pat_names, verbose=False):
        for name, param in self.named_parameters():
            if any(p in name for p in pat_names):
                param.requires_grad = True
                if verbose:
                    print("unfreeze:", name)

    def load_state_dict(self, state_dict, strict=True):
        for k, v in state_dict.items():
            if "prompt_memory_pool" in k:
                class_name = k.split(".")[-1]
                if class_name == "prompt_memory_pool":
                    continue
                self.prompt_memory_pool[class_name] = nn.Parameter(v)
                self.learned_classes.append(class_name[1:-1])
        return super().load_state_dict(state_dict=state_dict, strict=strict)

    def add_cls_prompt(self, class_names, device="cpu", fixed_name=False):
        for class_name in class_names:
            self.learned_classes.append(class_name)
            if not fixed_name:
                class_name = "-{}-".format(class_name)
            if class_name not in self.prompt_memory_pool:
                self.prompt_memory_pool[class_name] = nn.Parameter(torch.randn(self.hidden_dim).to(device))

    def before_train(self):
        """Freeze everything, then unfreeze what the method trains: every parameter whose name
        contains "adapter" -- the two side branches and their twins (reference :722-734)."""
        if self.freeze_all:
            for param in self.parameters():
                param.requires_grad = False
        if self.use_bert_tuning:
            self.unfreeze_module_(["bert", "feat_map"])
        if self.use_prompt_tuning:
            self.unfreeze_module_(["prompt_memory_pool"])
        if self.use_project_tuning:
            self.unfreeze_module_(["input_proj"])
        self.unfreeze_module_(["adapter"])

    def after_train(self):
        """End of a task: merge every side branch into its twin (reference :739-745)."""
        for module in self.modules():
            if hasattr(module, "__rep__"):
                module.__rep__()

    def side_branch_parameters(self):
        """The only tensors that ever receive gradients in ZiRa (4 622 853 values for Swin-T)."""
        return [p for n, p in self.named_parameters() if "adapter" in n]


@MODULE_BUILD_FUNCS.registe_with_name(module_name="dualzerorepbranchgroundingdino")
def build_dual_zero_rep_branch_groundingdino(args, bert=None, tokenizer=None, side_branch="rep"):
    backbone = build_backbone(args)
    transformer = build_transformer(args)
    criterion = build_criterion(args)
    return GroundingDINO(
        backbone, transformer, num_queries=args.num_queries, aux_loss=True, iter_update=True,
        query_dim=4, num_feature_levels=args.num_feature_levels, nheads=args.nheads,
        dec_pred_bbox_embed_share=args.dec_pred_bbox_embed_share, two_stage_type=args.two_stage_type,
        two_stage_bbox_embed_share=args.two_stage_bbox_embed_share,
        two_stage_class_embed_share=args.two_stage_class_embed_share, num_patterns=args.num_patterns,
        dn_number=0, dn_box_noise_scale=args.dn_box_noise_scale,
        dn_label_noise_ratio=args.dn_label_noise_ratio, dn_labelbook_size=args.dn_labelbook_size,
        text_encoder_type=args.text_encoder_type, sub_sentence_present=args.sub_sentence_present,
        max_text_len=args.max_text_len, criterion=criterion, freeze_all=args.freeze_all,
        select_box_nums_for_evaluation=args.select_box_nums_for_evaluation,
        loss_adapter_weight=args.loss_adapter_weight, use_cet=args.use_cet,
        use_prompt_memory=args.use_prompt_memory, use_zero_inter_loss=args.use_zero_inter_loss,
        use_add_names=args.use_add_names, use_bert_tuning=args.use_bert_tuning,
        use_cls_linear=args.use_cls_linear, use_prompt_tuning=args.use_prompt_tuning,
        use_prompt_memory_output=args.use_prompt_memory_output,
        use_project_tuning=getattr(args, "use_project_tuning", False),
        use_project_adapter=args.use_project_adapter,
        use_zero_inter_loss_for_conv=args.use_zero_inter_loss_for_conv,
        use_learned_names=args.use_learned_names, device=getattr(args, "device", "cuda"),
        bert=bert, tokenizer=tokenizer, side_branch=side_branch)


@MODULE_BUILD_FUNCS.registe_with_name(module_name="dualzerorepmultilayerbranchgroundingdino")
def build_dual_zero_rep_multi_layer_branch_groundingdino(args, bert=None, tokenizer=None):
    """The multilayer-branch ablation (reference groundingdino_dual_zero_rep_multilayer_branch.py:971-1020):
    same constructor arguments, other side-branch modules (rsb_multilayer.py)."""
    model = build_dual_zero_rep_branch_groundingdino(args, bert=bert, tokenizer=tokenizer, side_branch="multilayer")
    return model


def build_model(args, **kw):
    """groundingdino/models/__init__.py:11-18: dispatch on ``args.modelname``."""
    assert args.modelname in MODULE_BUILD_FUNCS, "unknown model %s" % args.modelname
    return MODULE_BUILD_FUNCS.get(args.modelname)(args, **kw)
