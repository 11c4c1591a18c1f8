"""ORACLE (test infrastructure — never imported by the product path).

The epilogue of the reference's ``get_results`` (utils/utils_manet.py:59-163) restated with the SAME torch calls the
reference makes, run on the CPU:

    pred_label = nn.functional.interpolate(pred_label, size=(h, w), mode='bilinear', align_corners=True)   :78-79
    probs.append(pred_label); pred_label = torch.argmax(pred_label, dim=1); pred_masks.append(pred_label.float())   :80-82
    final_masks = torch.cat(pred_masks_reverse, 0); all_P = torch.softmax(torch.cat(probs_reverse, 0), 1)   :158-161

``get_results`` below is the reference control flow (interaction head on ``next_frame``, forward propagation,
backward propagation, list reversal + concatenation) with ``.cuda()`` dropped, so a deterministic stand-in model
can drive both it and the product's ``ivos_w_amd.utils.utils_manet.get_results``.  The reference has no tests for
this function; the pin is that these are its own library calls (PyTorch 2.10 CPU kernels here vs 1.3 there; the
ops used have unchanged semantics with ``align_corners=True`` passed explicitly).
"""
import torch
from torch import nn


def epilogue(logits, h, w):
    up = nn.functional.interpolate(logits, size=(h, w), mode='bilinear', align_corners=True)
    label = torch.argmax(up, dim=1)
    return up, label


def get_results(model, ref_frame_embedding, scribble_label, prev_label, eval_global_map_tmp_dic, local_map_dics,
                n_interaction, sequence, obj_nums, next_frame, first_scribble, h, w, prev_label_storage, total_frame_num,
                embedding_memory, knns):
    pred_masks, pred_masks_reverse, probs, probs_reverse = [], [], [], []
    tmp_dic, local_map_dics = model.int_seghead(ref_frame_embedding=ref_frame_embedding, ref_scribble_label=scribble_label,
                                                prev_round_label=prev_label, global_map_tmp_dic=eval_global_map_tmp_dic,
                                                local_map_dics=local_map_dics, interaction_num=n_interaction,
                                                seq_names=[sequence], gt_ids=torch.Tensor([obj_nums]),
                                                frame_num=[next_frame], first_inter=first_scribble)
    pred_label, lab = epilogue(tmp_dic[sequence], h, w)
    probs.append(pred_label)
    pred_label = lab
    pred_masks.append(pred_label.float())
    prev_label_storage[next_frame] = pred_label
    ref_prev_label = pred_label.unsqueeze(0)
    prev_label = pred_label.unsqueeze(0)
    prev_embedding = ref_frame_embedding
    for ii in range(next_frame + 1, total_frame_num):
        current_embedding = embedding_memory[ii].unsqueeze(0)
        tmp_dic, eval_global_map_tmp_dic, local_map_dics = model.prop_seghead(
            ref_frame_embedding, prev_embedding, current_embedding, scribble_label, prev_label,
            normalize_nearest_neighbor_distances=True, use_local_map=True, seq_names=[sequence],
            gt_ids=torch.Tensor([obj_nums]), k_nearest_neighbors=knns, global_map_tmp_dic=eval_global_map_tmp_dic,
            local_map_dics=local_map_dics, interaction_num=n_interaction, start_annotated_frame=next_frame,
            frame_num=[ii], dynamic_seghead=model.dynamic_seghead)
        pred_label, lab = epilogue(tmp_dic[sequence], h, w)
        probs.append(pred_label)
        pred_label = lab
        pred_masks.append(pred_label.float())
        prev_label = pred_label.unsqueeze(0)
        prev_embedding = current_embedding
        prev_label_storage[ii] = pred_label
    prev_label = ref_prev_label
    prev_embedding = ref_frame_embedding
    for ii in range(next_frame):
        current_frame_num = next_frame - 1 - ii
        current_embedding = embedding_memory[current_frame_num].unsqueeze(0)
        tmp_dic, eval_global_map_tmp_dic, local_map_dics = model.prop_seghead(
            ref_frame_embedding, prev_embedding, current_embedding, scribble_label, prev_label,
            normalize_nearest_neighbor_distances=True, use_local_map=True, seq_names=[sequence],
            gt_ids=torch.Tensor([obj_nums]), k_nearest_neighbors=knns, global_map_tmp_dic=eval_global_map_tmp_dic,
            local_map_dics=local_map_dics, interaction_num=n_interaction, start_annotated_frame=next_frame,
            frame_num=[current_frame_num], dynamic_seghead=model.dynamic_seghead)
        pred_label, lab = epilogue(tmp_dic[sequence], h, w)
        probs_reverse.append(pred_label)
        pred_label = lab
        pred_masks_reverse.append(pred_label.float())
        prev_label = pred_label.unsqueeze(0)
        prev_embedding = current_embedding
        prev_label_storage[current_frame_num] = pred_label
    pred_masks_reverse.reverse()
    pred_masks_reverse.extend(pred_masks)
    probs_reverse.reverse()
    probs_reverse.extend(probs)
    final_masks = torch.cat(pred_masks_reverse, 0)
    all_P = torch.softmax(torch.cat(probs_reverse, 0), 1)
    return final_masks, all_P
