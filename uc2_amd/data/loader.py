"""Device-side input pipeline: the reference's collate functions + PrefetchLoader, re-cut for the GPU.

Reference: `xlmr_itm_collate` (data/itm.py:205-232), `xlmr_mlm_collate` (data/mlm.py:761-801), `xlmr_mrfr_collate` /
`xlmr_mrc_collate` (data/mrm.py:73-119,253-288) pad every sample in Python loops on the host (`pad_tensors`,
`get_gather_index`, `_mask_img_feat`, data/data.py:360-384, data/mrm.py:36-39) and `PrefetchLoader`
(data/loader.py:85-140) then moves the dict to the GPU tensor by tensor on a side stream.

Here a collate call only CONCATENATES the ragged per-sample arrays into a handful of flat pinned host buffers
(`RaggedBatch`); `DevicePrefetcher` copies those with one async H2D copy each on a side stream and two kernels
(`uc2_collate_regions`, `uc2_collate_index`) write the padded batch -- same dict keys, same values as the reference's
collates.  `feat_dtype=torch.bfloat16` emits the region features directly in the compute dtype (the model's cast pass
disappears); the default float32 is value-identical to the reference.
"""
import torch

from .. import _lib
from .._lib import call, ptr

TASK_FIELDS = {          # the per-sample tuples of the reference's datasets
    "itm": ("input_ids", "img_feat", "img_pos_feat", "attn_masks", "target"),                                  # data/itm.py:186-202
    "mlm": ("input_ids", "img_feat", "img_pos_feat", "attn_masks", "txt_labels"),                              # data/mlm.py:375-394
    "mrfr": ("input_ids", "img_feat", "img_pos_feat", "attn_masks", "img_mask", "img_mask_tgt"),               # data/mrm.py:54-71
    "mrc": ("input_ids", "img_feat", "img_pos_feat", "img_soft_labels", "attn_masks", "img_mask", "img_mask_tgt"),   # data/mrm.py:233-250
}


class RaggedBatch:
    """flat pinned host buffers of one batch"""

    def __init__(self, task, tensors, sizes):
        self.task, self.t, self.sizes = task, tensors, sizes


def _pin(t):
    return t.pin_memory() if torch.cuda.is_available() else t


def ragged_collate(task):
    """collate_fn for torch.utils.data.DataLoader: a list of per-sample tuples (the reference datasets' __getitem__
    output) -> RaggedBatch.  No padding, no per-sample Python tensor writes: one torch.cat per field."""
    if task not in TASK_FIELDS:
        raise ValueError("unsupported task %r" % (task,))

    def collate(inputs):
        cols = list(zip(*inputs))
        input_ids, img_feats, img_pos_feats = cols[0], cols[1], cols[2]
        txt_lens = [int(t.numel()) for t in input_ids]
        num_bbs = [int(f.shape[0]) for f in img_feats]
        t = {"ids": _pin(torch.cat(list(input_ids)).long()),
             "feat": _pin(torch.cat(list(img_feats), 0).float().contiguous()),
             "pos": _pin(torch.cat(list(img_pos_feats), 0).float().contiguous()),
             "txt_off": _pin(torch.tensor([0] + txt_lens, dtype=torch.long).cumsum(0)),
             "row_off": _pin(torch.tensor([0] + num_bbs, dtype=torch.long).cumsum(0))}
        if task == "itm":
            t["targets"] = _pin(torch.cat(list(cols[4]), 0).long())
        elif task == "mlm":
            t["labels"] = _pin(torch.cat(list(cols[4])).long())
        elif task == "mrfr":                 # (ids, feat, pos, attn, img_mask, img_mask_tgt), data/mrm.py:60-71
            t["mask"] = _pin(torch.cat(list(cols[4])).to(torch.uint8))
        else:                                # mrc: (ids, feat, pos, soft_labels, attn, img_mask, img_mask_tgt), data/mrm.py:233-250
            t["mask"] = _pin(torch.cat(list(cols[5])).to(torch.uint8))
            t["soft"] = _pin(torch.cat(list(cols[3]), 0).float().contiguous())
        sizes = dict(B=len(inputs), maxT=max(txt_lens), maxR=max(num_bbs),
                     Lout=max(a + b for a, b in zip(txt_lens, num_bbs)))
        return RaggedBatch(task, t, sizes)
    return collate


def assemble(rb, device, feat_dtype=torch.float32, pad_id=1):
    """RaggedBatch -> the reference's batch dict on `device` (kernels on the current stream)"""
    s, task = rb.sizes, rb.task
    d = {k: v.to(device, non_blocking=True) for k, v in rb.t.items()}
    B, maxT, maxR, Lout = s["B"], s["maxT"], s["maxR"], s["Lout"]
    D = d["feat"].shape[1]
    st = torch.cuda.current_stream(device).cuda_stream
    masked = task in ("mrfr", "mrc")
    mask = d.get("mask")
    batch = {"position_ids": torch.arange(0, maxT, dtype=torch.long, device=device).unsqueeze(0)}
    img_feat = torch.empty((B, maxR, D), dtype=feat_dtype, device=device)
    img_pos = torch.empty((B, maxR, d["pos"].shape[1]), dtype=torch.float32, device=device)
    call("uc2_collate_regions", _lib.dt(feat_dtype), B, maxR, D, ptr(d["feat"]), ptr(d["row_off"]), ptr(mask), ptr(img_feat), st)
    call("uc2_collate_regions", 0, B, maxR, d["pos"].shape[1], ptr(d["pos"]), ptr(d["row_off"]), None, ptr(img_pos), st)
    input_ids = torch.empty((B, maxT), dtype=torch.long, device=device)
    attn = torch.empty((B, Lout), dtype=torch.long, device=device)
    gather = torch.empty((B, Lout), dtype=torch.long, device=device)
    img_masks = torch.empty((B, maxR), dtype=torch.uint8, device=device) if masked else None
    img_mask_tgt = torch.empty((B, Lout), dtype=torch.uint8, device=device) if masked else None
    txt_labels = torch.empty((B, maxT), dtype=torch.long, device=device) if task == "mlm" else None
    call("uc2_collate_index", B, maxT, maxR, Lout, ptr(d["ids"]), ptr(d["txt_off"]), ptr(d["row_off"]), pad_id, ptr(mask),
         ptr(d.get("labels")), ptr(input_ids), ptr(attn), ptr(gather), ptr(img_masks), ptr(img_mask_tgt), ptr(txt_labels), st)
    batch.update(input_ids=input_ids, img_feat=img_feat, img_pos_feat=img_pos, attn_masks=attn, gather_index=gather)
    if task == "itm":
        batch["targets"] = d["targets"]
    elif task == "mlm":
        batch["txt_labels"] = txt_labels
        # the number of masked tokens is known here, on the host: with it the model compacts the masked rows without a
        # device -> host sync (uc2_amd/model/model.py::_masked_rows; the reference's boolean indexing syncs, model/model.py:653-657)
        batch["n_txt_labels"] = int((rb.t["labels"] != -1).sum())
    else:
        batch["img_masks"] = img_masks.bool()
        batch["img_mask_tgt"] = img_mask_tgt.bool()
        batch["n_img_mask_tgt"] = int(rb.t["mask"].sum())
        # targets of the masked regions in batch-major, region-minor order (= boolean indexing, data/mrm.py:28-33):
        # rows of the flat buffers selected by the flat mask itself
        rows = torch.nonzero(rb.t["mask"], as_tuple=False).view(-1).to(device, non_blocking=True)     # host-side index list
        if task == "mrfr":
            ft = torch.empty((rows.numel(), D), dtype=torch.float32, device=device)
            call("uc2_select_rows", 0, rows.numel(), D, ptr(d["feat"]), D, ptr(rows), ptr(ft), D, 0, st)
            batch["feat_targets"] = ft
        else:
            C = d["soft"].shape[1]
            lt = torch.empty((rows.numel(), C), dtype=torch.float32, device=device)
            if C % 4 == 0:
                call("uc2_select_rows", 0, rows.numel(), C, ptr(d["soft"]), C, ptr(rows), ptr(lt), C, 0, st)
            else:
                lt = d["soft"].index_select(0, rows)
            batch["label_targets"] = lt
    for v in d.values():                      # the flat buffers were allocated on this stream: nothing to record
        pass
    return batch


class DevicePrefetcher:
    """PrefetchLoader (data/loader.py:85-140) for RaggedBatch streams: the next batch's H2D copies and assembly kernels
    run on a side stream while the model works on the current one; tensors are handed over with an event + record_stream
    exactly like the reference does."""

    def __init__(self, loader, device, feat_dtype=torch.float32):
        self.loader, self.device, self.feat_dtype = loader, torch.device(device), feat_dtype
        self.stream = torch.cuda.Stream(device=self.device)

    def __len__(self):
        return len(self.loader)

    def _preload(self, it):
        try:
            rb = next(it)
        except StopIteration:
            return None
        name = None
        if isinstance(rb, tuple):            # MetaLoader yields (task name, batch)
            name, rb = rb
        with torch.cuda.stream(self.stream):
            batch = assemble(rb, self.device, self.feat_dtype)
        return (name, batch) if name is not None else batch

    def __iter__(self):
        it = iter(self.loader)
        nxt = self._preload(it)
        while nxt is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.stream)
            cur = nxt
            b = cur[1] if isinstance(cur, tuple) else cur
            for v in b.values():
                if torch.is_tensor(v):
                    v.record_stream(torch.cuda.current_stream(self.device))
            nxt = self._preload(it)
            yield cur
