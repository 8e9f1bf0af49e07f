"""Sub-sentence attention masks for the text encoder (reference
groundingdino/models/GroundingDINO/bertwarper.py:224-273).

Pure integer / boolean index logic: for a caption "cat . dog . person ." every phrase between
two special tokens ([CLS], [SEP], ".", "?") only attends to itself, position ids restart per
phrase, and each phrase yields one boolean token mask (category -> its tokens).
"""
import torch


def generate_masks_with_special_tokens_and_transfer_map(tokenized, special_tokens_list, tokenizer=None):
    """-> (attention_mask [bs,T,T] bool, position_ids [bs,T] int64, list of [n_cat,T] bool)."""
    input_ids = tokenized["input_ids"]
    bs, num_token = input_ids.shape
    device = input_ids.device
    special = torch.zeros((bs, num_token), device=device, dtype=torch.bool)
    for tok in special_tokens_list:
        special |= input_ids == tok

    attention_mask = torch.eye(num_token, device=device, dtype=torch.bool).unsqueeze(0).repeat(bs, 1, 1)
    position_ids = torch.zeros((bs, num_token), device=device, dtype=torch.long)
    cate_masks = [[] for _ in range(bs)]
    # one host copy of the special-token positions instead of a sync per token
    rows, cols = torch.nonzero(special, as_tuple=True)
    previous_col = 0
    for row, col in zip(rows.tolist(), cols.tolist()):
        if col == 0 or col == num_token - 1:
            attention_mask[row, col, col] = True
            position_ids[row, col] = 0
        else:
            attention_mask[row, previous_col + 1:col + 1, previous_col + 1:col + 1] = True
            position_ids[row, previous_col + 1:col + 1] = torch.arange(0, col - previous_col, device=device)
            m = torch.zeros(num_token, device=device, dtype=torch.bool)
            m[previous_col + 1:col] = True
            cate_masks[row].append(m)
        previous_col = col  # NB: carried across rows exactly as the reference does
    cate_masks = [torch.stack(m, dim=0) if len(m) else torch.zeros((0, num_token), dtype=torch.bool, device=device)
                  for m in cate_masks]
    return attention_mask, position_ids, cate_masks
