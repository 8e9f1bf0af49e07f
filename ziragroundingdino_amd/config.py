"""Model configuration: the reference keeps it as flat Python variables in
groundingdino/config/GroundingDINO_SwinT_OGC_rep.py (read there through SLConfig, which needs
addict + yapf); here the same file format is read with a plain ``exec`` and the ZiRa defaults
are built in, so ``--model-config-file`` of the reference's CLI keeps working."""
from types import SimpleNamespace

# values of config/GroundingDINO_SwinT_OGC_rep.py:1-93 (the ZiRa model config)
ZIRA_SWINT_DEFAULTS = dict(
    batch_size=1, modelname="dualzerorepbranchgroundingdino", backbone="swin_T_224_1k",
    position_embedding="sine", pe_temperatureH=20, pe_temperatureW=20, return_interm_indices=[1, 2, 3],
    backbone_freeze_keywords=None, enc_layers=6, dec_layers=6, pre_norm=False, dim_feedforward=2048,
    hidden_dim=256, dropout=0.0, nheads=8, num_queries=900, query_dim=4, num_patterns=0,
    num_feature_levels=4, enc_n_points=4, dec_n_points=4, two_stage_type="standard",
    two_stage_bbox_embed_share=False, two_stage_class_embed_share=False, transformer_activation="relu",
    dec_pred_bbox_embed_share=True, dn_box_noise_scale=1.0, dn_label_noise_ratio=0.5, dn_label_coef=1.0,
    dn_bbox_coef=1.0, embed_init_tgt=True, dn_labelbook_size=2000, max_text_len=256,
    text_encoder_type="bert-base-uncased", use_text_enhancer=True, use_fusion_layer=True,
    use_checkpoint=False, use_transformer_ckpt=False, use_text_cross_attention=True, text_dropout=0.0,
    fusion_dropout=0.0, fusion_droppath=0.1, sub_sentence_present=True, aux_loss=True, freeze_all=True,
    select_box_nums_for_evaluation=200, use_adapter=False, use_self_kd=False, use_add_names=False,
    use_learned_names=False, use_cet=True, cet_middle_dim=1024, use_prompt_memory=False,
    use_prompt_memory_output=True, use_zero_inter_loss=True, num_experts=1, num_topk_experts=1,
    use_bert_tuning=False, use_cls_linear=False, use_prompt_tuning=False, use_project_adapter=True,
    use_zero_inter_loss_for_conv=True, loss_adapter_weight=0.1)


def zira_swint_config(**overrides):
    cfg = dict(ZIRA_SWINT_DEFAULTS)
    cfg.update(overrides)
    return SimpleNamespace(**cfg)


def load_config_file(path, **overrides):
    """Read a flat-python-variables model config (the SLConfig file format)."""
    scope = {}
    with open(path) as f:
        exec(compile(f.read(), path, "exec"), {}, scope)
    cfg = dict(ZIRA_SWINT_DEFAULTS)
    cfg.update({k: v for k, v in scope.items() if not k.startswith("_")})
    cfg.update(overrides)
    return SimpleNamespace(**cfg)
