"""MANet adapter glue on the hot-path side: the per-frame epilogue of ``get_results`` (reference
``utils/utils_manet.py:59-163``) — SURVEY §8(f) row 4.

The MANet network itself (``model.int_seghead`` / ``model.prop_seghead``) is an external clone and stays whatever the
caller passes in; what is rebuilt here is everything between its low-resolution logits and the tensors the rest of
the loop consumes: bilinear upsampling, argmax (the label that drives the next propagation step and the J/F metrics),
and the softmax that becomes ``all_P`` — fused into one HIP kernel per frame (``csrc/seg_epilogue.hip``), writing
``all_P`` once, object-major, so ``assess_all_objects`` slices per-object soft masks without a transpose copy.
There is no CPU fallback.
"""
import torch

from .. import _lib as L


class ProbStore:
    """``all_P`` for one sequence, stored object-major ([C, n, H, W] contiguous) and exposed in the reference's
    [n, C, H, W] indexing through ``all_P`` (a permuted view: same values, different strides)."""

    def __init__(self, n_frames, n_channels, h, w, device):
        self.buf = torch.empty((n_channels, n_frames, h, w), dtype=torch.float32, device=device)
        self.labels_u8 = torch.zeros((n_frames, h, w), dtype=torch.uint8, device=device)     # for the J/F kernels
        self.final_masks = torch.empty((n_frames, h, w), dtype=torch.float32, device=device)

    @property
    def all_P(self):
        return self.buf.permute(1, 0, 2, 3)


def seg_epilogue(pred_label, h, w, store=None, frame=None):
    """logits [k, C, hs, ws] (device fp32) -> (label int64 [k, h, w], probs [k, C, h, w]).

    With ``store``/``frame`` the probabilities, the float label and the uint8 label are written straight into the
    sequence-level buffers (frames ``frame .. frame+k-1``) and the returned probs are a view of that slot."""
    if not isinstance(pred_label, torch.Tensor) or not pred_label.is_cuda:
        raise RuntimeError("seg_epilogue: logits must be a CUDA tensor (no CPU fallback)")
    x = pred_label.contiguous()
    if x.dtype != torch.float32:
        x = x.float()
    k, C, hs, ws = x.shape
    dev = x.device
    label = torch.empty((k, h, w), dtype=torch.int64, device=dev)
    lib = L.lib()
    if store is None:
        probs = torch.empty((k, C, h, w), dtype=torch.float32, device=dev)
        L.check(lib.ivosw_seg_epilogue(L.dptr(x), k, C, hs, ws, h, w, L.dptr(probs), C * h * w, h * w, L.dptr(label), None, None,
                                       L.stream_ptr(dev)), "seg_epilogue")
        return label, probs
    n_total = store.buf.shape[1]
    if store.buf.shape[0] != C or tuple(store.buf.shape[2:]) != (h, w) or frame < 0 or frame + k > n_total:
        raise ValueError("ProbStore shape does not match the logits")
    hw = h * w
    slot = store.buf[:, frame:frame + k]
    L.check(lib.ivosw_seg_epilogue(L.dptr(x), k, C, hs, ws, h, w,
                                   torch_ptr(store.buf, frame * hw), hw, n_total * hw, L.dptr(label),
                                   torch_ptr(store.labels_u8, frame * hw), torch_ptr(store.final_masks, frame * hw),
                                   L.stream_ptr(dev)), "seg_epilogue")
    return label, slot.permute(1, 0, 2, 3)


def torch_ptr(t, elem_offset):
    import ctypes
    L.dptr(t)          # CUDA + contiguity check
    return ctypes.c_void_p(t.data_ptr() + elem_offset * t.element_size())


def get_results(model, ref_frame_embedding, scribble_label, prev_label, eval_global_map_tmp_dic, local_map_dics,
                n_interaction, sequence, obj_nums, next_frame, first_scribble, h, w, prev_label_storage, total_frame_num,
                embedding_memory, knns=None, store=None):
    """Same signature and return value as the reference's ``get_results`` (``final_masks`` float [n, h, w], ``all_P``
    [n, O+1, h, w] fp32) plus two optional keywords: ``knns`` (the reference reads MANet's global ``cfg.KNNS``) and
    ``store`` (a ProbStore to reuse across interactions).  Control flow follows utils/utils_manet.py:59-163 line by
    line: interaction head on ``next_frame``, propagation forwards, propagation backwards."""
    if knns is None:
        from config import cfg as manet_cfg          # MANet's own config module, as in the reference
        knns = manet_cfg.KNNS
    dev = ref_frame_embedding.device
    tmp_dic, local_map_dics = model.int_seghead(ref_frame_embedding=ref_frame_embedding,
                                                ref_scribble_label=scribble_label,
                                                prev_round_label=prev_label,
                                                global_map_tmp_dic=eval_global_map_tmp_dic,
                                                local_map_dics=local_map_dics,
                                                interaction_num=n_interaction,
                                                seq_names=[sequence],
                                                gt_ids=torch.Tensor([obj_nums]),
                                                frame_num=[next_frame],
                                                first_inter=first_scribble)
    logits = tmp_dic[sequence]
    if store is None:
        store = ProbStore(total_frame_num, logits.shape[1], h, w, dev)
    pred_label, _ = seg_epilogue(logits, h, w, store, next_frame)
    prev_label_storage[next_frame] = pred_label

    ref_prev_label = pred_label.unsqueeze(0)

    def propagate(frames):
        nonlocal eval_global_map_tmp_dic, local_map_dics
        prev_label = ref_prev_label
        prev_embedding = ref_frame_embedding
        for ii in frames:
            current_embedding = embedding_memory[ii].unsqueeze(0)
            prev_label = prev_label.to(dev)
            tmp_dic, eval_global_map_tmp_dic, local_map_dics = model.prop_seghead(
                ref_frame_embedding, prev_embedding, current_embedding, scribble_label, prev_label,
                normalize_nearest_neighbor_distances=True, use_local_map=True, seq_names=[sequence],
                gt_ids=torch.Tensor([obj_nums]), k_nearest_neighbors=knns,
                global_map_tmp_dic=eval_global_map_tmp_dic, local_map_dics=local_map_dics,
                interaction_num=n_interaction, start_annotated_frame=next_frame, frame_num=[ii],
                dynamic_seghead=model.dynamic_seghead)
            label, _ = seg_epilogue(tmp_dic[sequence], h, w, store, ii)
            prev_label = label.unsqueeze(0)
            prev_embedding = current_embedding
            prev_label_storage[ii] = label

    propagate(range(next_frame + 1, total_frame_num))          # propagation ->
    propagate(range(next_frame - 1, -1, -1))                   # propagation <-
    return store.final_masks, store.all_P
