"""Image-text retrieval evaluation on the device (reference itm.py:492-538 `evaluate` / `inference`, itm.py:448-489
`validate`, eval/itm.py:6-53 `itm_eval`).

The encoder runs forward-only (no activations kept, single-stream GELU GEMM, no LayerNorm statistics: see
ops.BertLayerFn); every mini-batch's scores are written straight into its slice of the fp16 score matrix on the
device, and the recall numbers come from a rank-counting kernel (uc2_rank_of_target: the position of the ground truth
in a stable descending sort) instead of top-k + nonzero round trips.  Ties are broken by index (torch.topk leaves the
order of equal scores unspecified)."""
import torch

from .. import _lib
from ..utils.distributed import _rank, _world, all_gather_list


def _dtype_code(t):
    return {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[t.dtype]


def rank_of_target(scores, off, stride, target, nc):
    """rank[q] of candidate target[q] among scores[off[q] + c*stride], c < nc (int32, on the device)"""
    nq = off.numel()
    out = torch.empty(nq, dtype=torch.int32, device=scores.device)
    _lib.call("uc2_rank_of_target", _dtype_code(scores), nq, nc, _lib.ptr(scores), _lib.ptr(off.contiguous()), int(stride),
              _lib.ptr(target.contiguous()), _lib.ptr(out), _lib.stream())
    return out


@torch.no_grad()
def inference(model, eval_loader, n_txt=None, n_img=None):
    """score_matrix [n_txt_local, n_img] fp16 (itm.py:516-538).  `eval_loader` yields, per text, a list of mini-batches
    covering all images in order (ItmEvalDataset, inf_minibatch_size images each)."""
    was_training = model.training
    model.eval()
    dev = next(model.parameters()).device
    if n_txt is None:
        n_txt, n_img = len(eval_loader.dataset), len(eval_loader.dataset.all_img_ids)
    score_matrix = torch.zeros(n_txt, n_img, device=dev, dtype=torch.float16)
    for i, mini_batches in enumerate(eval_loader):
        j = 0
        for batch in mini_batches:
            scores = model(batch, compute_loss=False)
            bs = scores.size(0)
            score_matrix[i, j:j + bs] = scores.reshape(-1).to(torch.float16)
            j += bs
        assert j == n_img
    if was_training:
        model.train()
    return score_matrix


@torch.no_grad()
def itm_eval(score_matrix, txt_ids, img_ids, txt2img, img2txts, reference_row_term=True):
    """recall@{1,5,10} both ways + means, same keys as eval/itm.py:6-53.
    reference_row_term: the reference counts `(rank < k).sum()` over the [n_found, 2] output of nonzero(), i.e. it adds
    the number of found texts whose ROW INDEX is below k to the number whose rank is below k (eval/itm.py:16-20); at
    real evaluation sizes that is at most k / n_txt.  True reproduces its numbers exactly, False gives plain recall."""
    dev = score_matrix.device
    n_txt, n_img = score_matrix.shape
    sm = score_matrix.contiguous()
    img2j = {i: j for j, i in enumerate(img_ids)}
    txt2i = {t: i for i, t in enumerate(txt_ids)}
    # image retrieval: for text i rank its image among the n_img columns of row i
    gt = torch.tensor([img2j[txt2img[t]] for t in txt_ids], dtype=torch.long, device=dev)
    off = torch.arange(n_txt, dtype=torch.long, device=dev) * n_img
    r = rank_of_target(sm, off, 1, gt, n_img)
    found = r < 10                                       # the reference looks at the top 10 only
    rows = torch.arange(n_txt, device=dev)
    ir = [float(((r < k) & found).sum().item() + (((rows < k) & found).sum().item() if reference_row_term else 0))
          / len(txt_ids) for k in (1, 5, 10)]
    # text retrieval: for image j the best-ranked of its ground-truth texts along column j
    pairs_j, pairs_i = [], []
    for j, img_id in enumerate(img_ids):
        for t in img2txts[img_id]:
            pairs_j.append(j)
            pairs_i.append(txt2i[t])
    pj = torch.tensor(pairs_j, dtype=torch.long, device=dev)
    pi = torch.tensor(pairs_i, dtype=torch.long, device=dev)
    rr = rank_of_target(sm, pj, n_img, pi, n_txt).long()
    best = torch.full((n_img,), 1 << 30, dtype=torch.long, device=dev)
    best.scatter_reduce_(0, pj, rr, reduce="amin")
    tr = [float((best < k).sum().item()) / len(img_ids) for k in (1, 5, 10)]
    tr_mean, ir_mean = sum(tr) / 3, sum(ir) / 3
    return {'txt_r1': tr[0], 'txt_r5': tr[1], 'txt_r10': tr[2], 'txt_r_mean': tr_mean,
            'img_r1': ir[0], 'img_r5': ir[1], 'img_r10': ir[2], 'img_r_mean': ir_mean,
            'r_mean': (tr_mean + ir_mean) / 2}


def allgather_rows(x):
    """hvd.allgather(x) (itm.py:496): concatenation over ranks along dim 0 where every rank may hold a DIFFERENT number
    of rows -- the text ids are dealt out as ids[rank::size] (data/data.py:201-203), so n_txt % world != 0 gives ragged
    shards (COCO: 25 010 texts on 8 ranks).  Row counts are exchanged first; shards are padded to the longest for the
    fixed-shape all_gather and trimmed again."""
    import torch.distributed as dist
    world = _world()
    n = torch.tensor([x.shape[0]], dtype=torch.long, device=x.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    nmax = max(counts)
    if x.shape[0] < nmax:
        x = torch.cat([x, x.new_zeros((nmax - x.shape[0],) + tuple(x.shape[1:]))], 0)
    parts = [torch.empty_like(x) for _ in range(world)]
    dist.all_gather(parts, x.contiguous())
    return torch.cat([p[:c] for p, c in zip(parts, counts)], 0)


@torch.no_grad()
def evaluate(model, eval_loader):
    """itm.py:492-513: local score rows -> gathered over ranks -> metrics on rank 0"""
    score_matrix = inference(model, eval_loader)
    dset = eval_loader.dataset
    if _world() > 1:
        score_matrix = allgather_rows(score_matrix)
    all_txt_ids = [i for ids in all_gather_list(dset.ids) for i in ids]
    if _rank() != 0:
        return {}
    return itm_eval(score_matrix, all_txt_ids, dset.all_img_ids, dset.txt2img, dset.img2txts)


@torch.no_grad()
def validate(model, val_loader):
    """itm.py:448-489: each batch = one text with its positive image at index 0 followed by negatives"""
    was_training = model.training
    model.eval()
    ranks = []
    for batch in val_loader:
        scores = model(batch, compute_loss=False).reshape(-1).float()
        z = torch.zeros(1, dtype=torch.long, device=scores.device)
        ranks.append(rank_of_target(scores, z, 1, z, scores.numel()))
    r = torch.cat(ranks) if ranks else torch.zeros(0, dtype=torch.int32)
    n_ex = sum(all_gather_list(int(r.numel())))
    out = {'valid/recall_%d' % k: sum(all_gather_list(int((r < k).sum().item()))) / max(n_ex, 1) for k in (1, 5, 10)}
    if was_training:
        model.train()
    return out
