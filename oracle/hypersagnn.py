"""Oracle: functional PyTorch-CPU restatement of the reference's Hyper-SAGNN classifier.

TEST INFRASTRUCTURE (see oracle/__init__.py) -- never imported by matcha_amd/.

Restates, op for op, the *live* path of ``/root/reference/Code/Modules.py``
(``Classifier.forward`` :278-318 and everything it reaches) on a plain dict of
tensors keyed by the reference's own ``state_dict`` names, so reference weights
drop straight in.  Dead branches of the reference (encode2, fc2, pff_n2, the
autoencoder decoder, node_embedding.next_w -- SURVEY.md headline fact 3) are not
computed; their parameters simply receive no gradient, exactly as in the
reference.  Pinned against the real reference by tests/golden/ (see
tests/test_oracle_golden.py).

Randomness is *injected*: dropout masks are multiplier tensors (0 or 1/(1-p)),
``random_chrom`` (Modules.py:192) is an argument.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

N_HEAD = 8            # main.py:616
LN_EPS = 1e-5         # nn.LayerNorm default (Modules.py:240-241, :343, :498-500)
P_DROP_ADJ = 0.2      # Modules.py:174
P_DROP_FC1 = 0.3      # Modules.py:226 (dropout_mul)
P_DROP_PFF = 0.4      # Modules.py:227 (dropout_pff)


# --------------------------------------------------------------------------------------
# preprocessing restated from main.py / Modules.__init__
# --------------------------------------------------------------------------------------
def corrcoef_features(intra_adj: np.ndarray, chrom_range: np.ndarray) -> List[np.ndarray]:
    """main.py:571-577 -- per-chromosome np.corrcoef of the intra block, NaN -> 0, float32."""
    out = []
    for lo, hi in chrom_range:
        blk = np.asarray(intra_adj, dtype=np.float32)[lo - 1:hi - 1, lo - 1:hi - 1]
        with np.errstate(invalid="ignore", divide="ignore"):
            c = np.corrcoef(blk).astype(np.float32)
        c[np.isnan(c)] = 0.0
        out.append(c)
    return out


def zscore_inter(inter: np.ndarray) -> np.ndarray:
    """Modules.py:146-152 -- per row, z-score (ddof=0) of the strictly positive entries, NaN -> 0."""
    inter = np.array(inter, dtype=np.float32, copy=True)
    for i in range(len(inter)):
        row = inter[i, :]
        pos = row > 0
        v = row[pos]
        if v.size:
            with np.errstate(invalid="ignore", divide="ignore"):
                z = (v - v.mean()) / v.std()
            inter[i, pos] = z.astype(np.float32)
    inter[np.isnan(inter)] = 0.0
    return inter


def attribute_table(num: List[int]) -> np.ndarray:
    """main.py:497-512 -- [N+1, C+1]: one-hot chromosome || (bin index in chrom)/num[0]; row 0 zeros."""
    C = len(num)
    rows = []
    for i, n in enumerate(num):
        onehot = np.zeros((n, C))
        onehot[:, i] = 1
        coor = np.arange(n).reshape(-1, 1).astype("float32")
        coor /= num[0]
        rows.append(np.concatenate([onehot, coor], axis=-1))
    allr = np.concatenate(rows, axis=0)
    return np.concatenate([np.zeros((1, allr.shape[-1])), allr], axis=0).astype("float32")


@dataclass
class FrontEnd:
    """Non-parameter state of the node-embedding front end.

    mode 'adj'   -> MultipleEmbedding (Modules.py:125-201): feats[i] is the [n_i,n_i] feature
                    matrix of chromosome i, ``inter`` the z-scored [N,N] inter-chrom matrix.
    mode 'table' -> Wrap_Embedding (Modules.py:29-34): a trainable [N+1,d] table, row 0 = pad.
    ``bounds`` is Modules.py:138's num_list: [0, n_0, n_0+n_1, ...].
    """
    mode: str
    bounds: List[int]
    feats: List[torch.Tensor] = field(default_factory=list)
    inter: Optional[torch.Tensor] = None

    @property
    def n_chrom(self) -> int:
        return len(self.bounds) - 1

    @property
    def n_nodes(self) -> int:
        return int(self.bounds[-1])


# --------------------------------------------------------------------------------------
# forward
# --------------------------------------------------------------------------------------
def _ln(x, P, prefix):
    return F.layer_norm(x, (x.shape[-1],), P[prefix + ".weight"], P[prefix + ".bias"], LN_EPS)


def node_embeddings(P: Dict[str, torch.Tensor], fe: FrontEnd, xf: torch.Tensor,
                    random_chrom: Optional[int] = None,
                    adj_mask: Optional[torch.Tensor] = None):
    """``node_embedding(x.view(-1))`` -> (rows [T,d], recon_loss [1]).

    adj  : MultipleEmbedding.forward Modules.py:176-201 (+ SparseEmbedding :67, TiedAutoEncoder :104-113)
    table: Wrap_Embedding.forward Modules.py:33-34
    ``adj_mask`` [T, max n_i]: dropout(0.2) multiplier on the gathered feature rows (:186), or None.
    """
    if fe.mode == "table":
        W = P["node_embedding.weight"]
        return F.embedding(xf, W, padding_idx=0), torch.zeros(1, dtype=W.dtype)

    d = P["node_embedding.Embedding_Linear0.tied weight_1"].shape[0]
    T = xf.shape[0]
    final = torch.zeros((T, d), dtype=torch.float32)
    for i in range(fe.n_chrom):
        lo, hi = fe.bounds[i] + 1, fe.bounds[i + 1] + 1
        sel = (xf >= lo) & (xf < hi)                                  # :181
        if int(sel.sum()) == 0:                                       # :182-183
            continue
        rows = fe.feats[i][xf[sel] - lo]                              # :184, :67  <- THE GATHER
        if adj_mask is not None:
            rows = rows * adj_mask[sel][:, : rows.shape[1]]           # :186
        w0 = P[f"node_embedding.Embedding_Linear{i}.tied weight_0"]
        w1 = P[f"node_embedding.Embedding_Linear{i}.tied weight_1"]
        h = torch.tanh(rows @ w0.t()) @ w1.t()                        # :109-113 (use_bias=False :163)
        final = final.index_put((sel.nonzero(as_tuple=True)[0],), h)  # :188
    recon_loss = torch.zeros(1, dtype=torch.float32)
    if random_chrom is None:
        raise ValueError("adj mode needs the drawn random_chrom (Modules.py:192)")
    r = int(random_chrom)
    other = ((xf < fe.bounds[r] + 1) | (xf >= fe.bounds[r + 1] + 1)) & (xf != 0)   # :194
    if int(other.sum()) != 0:                                         # :195
        target = fe.inter[xf[other] - 1][:, fe.bounds[r]:fe.bounds[r + 1]]         # :196-197
        wr = P[f"node_embedding.Embedding_recon{r}.FF_Linear0.weight"]
        br = P[f"node_embedding.Embedding_recon{r}.FF_Linear0.bias"]
        rec = torch.tanh(final[other]) @ wr.t() + br                  # :198
        recon_loss = recon_loss + (target - rec).pow(2).mean(dim=-1).mean() * 100  # :199
    return final, recon_loss


def classifier_forward(P: Dict[str, torch.Tensor], fe: FrontEnd, x: torch.Tensor, *,
                       random_chrom: Optional[int] = None,
                       masks: Optional[Dict[str, torch.Tensor]] = None,
                       return_intermediates: bool = False):
    """Classifier.forward(x, return_recon=True) (Modules.py:278-318) -> (logits [B,1], recon_loss [1]).

    x: LongTensor [B,L], 0 = padding.  ``masks`` (training only) may hold multiplier tensors
    'adj' [T,max n_i], 'fc1' [T,d], 'pff' [T,d] (token-major; the reference draws pff's mask
    in [B,d,L] layout, Modules.py:354-360 -- same distribution).
    NOTE the reference never applies its key-pad mask (SURVEY.md headline fact 7): pad slots
    are ordinary keys/values; only the diagonal is masked (-1e32).
    """
    masks = masks or {}
    x = x.long()
    B, L = x.shape
    xf = x.reshape(-1)
    non_pad = x.ne(0).to(torch.float32).unsqueeze(-1)                     # :12-14  [B,L,1]

    # get_embedding :261-276
    attr = P["attribute_dict_embedding.weight"][xf] @ P["attribute_nn.weight"].t() + P["attribute_nn.bias"]   # :263-264
    node, recon_loss = node_embeddings(P, fe, xf, random_chrom, masks.get("adj"))
    d = node.shape[-1]
    x0 = node + attr                                                       # :269
    X = torch.tanh(x0 @ P["next_w.FF_Linear0.weight"].t() + P["next_w.FF_Linear0.bias"])   # :270
    X = X.view(B, L, d)

    # encode1.mul_head_attn  :513-575 (called as (dynamic, dynamic, static, key_pad) :612-613)
    pre = "encode1.mul_head_attn."
    H = N_HEAD
    q = (_ln(X, P, pre + "layer_norm1") @ P[pre + "w_qs.weight"].t()).view(B, L, H, d)   # :519, :527
    k = (_ln(X, P, pre + "layer_norm2") @ P[pre + "w_ks.weight"].t()).view(B, L, H, d)   # :520, :528
    v = (_ln(X, P, pre + "layer_norm3") @ P[pre + "w_vs.weight"].t()).view(B, L, H, d)   # :521, :529
    q = q.permute(2, 0, 1, 3).reshape(H * B, L, d)                        # :531-536
    k = k.permute(2, 0, 1, 3).reshape(H * B, L, d)
    v = v.permute(2, 0, 1, 3).reshape(H * B, L, d)
    attn = torch.bmm(q, k.transpose(1, 2)) / math.sqrt(d)                 # :449-450 (temperature = d_k**0.5 :493)
    eye = torch.eye(L, dtype=torch.bool).unsqueeze(0)
    attn = attn.masked_fill(eye, -1e32)                                   # :443-445 with diag_mask = 1 - eye :543-546
    attn = torch.softmax(attn, dim=-1)
    o = torch.bmm(attn, v)                                                # :458
    o = o.view(H, B, L, d).permute(1, 2, 0, 3).reshape(B, L, H * d)       # :563-566
    dyn = o @ P[pre + "fc1.weight"].t() + P[pre + "fc1.bias"]             # :572
    if "fc1" in masks:
        dyn = dyn * masks["fc1"].view(B, L, d)

    # encode1.pff_n1 ([d,d,d], residual, layer_norm)  :614, :353-376 ; Conv1d(k=1) == per-token Linear
    pp = "encode1.pff_n1."
    y = dyn * non_pad
    h = torch.tanh(y @ P[pp + "PWF_Conv0.weight"][:, :, 0].t() + P[pp + "PWF_Conv0.bias"])   # :357-358
    if "pff" in masks:
        h = h * masks["pff"].view(B, L, d)                                 # :359-360
    h = h @ P[pp + "PWF_Conv1.weight"][:, :, 0].t() + P[pp + "PWF_Conv1.bias"]               # :362
    h = h + y                                                              # :370-371
    dynamic = _ln(h, P, pp + "layer_norm") * non_pad                      # :373-374, :614

    # Classifier.forward tail :290-311
    dynamic_n = _ln(dynamic, P, "layer_norm1")
    static_n = _ln(X, P, "layer_norm2")
    diff2 = (dynamic_n - static_n) ** 2                                    # :295
    out = diff2 @ P["pff_classifier.PWF_Conv0.weight"][:, :, 0].t() + P["pff_classifier.PWF_Conv0.bias"]  # :299
    logits = (out * non_pad).sum(dim=-2) / (non_pad.sum(dim=-2) + 1e-15)  # :309-311
    if return_intermediates:
        return logits, recon_loss, dict(x0=x0, X=X, q=q, k=k, v=v, attn=attn, o=o, dyn=dyn, y=y,
                                        dynamic=dynamic, out=out, node=node)
    return logits, recon_loss


def bce_with_logits(logits, y, w):
    """main.py:529, :56 -- F.binary_cross_entropy_with_logits(pred, y, weight=w), mean reduction."""
    return F.binary_cross_entropy_with_logits(logits, y, weight=w)


def total_loss(P, fe, x, y, w, alpha, beta, **kw):
    """main.py:54-56, :166 -- loss = bce*alpha + recon*beta."""
    logits, recon = classifier_forward(P, fe, x, **kw)
    bce = bce_with_logits(logits, y, w)
    return bce * alpha + recon * beta, bce, recon, logits


def loss_and_grads(P, fe, x, y, w, alpha, beta, **kw):
    """One forward + autograd backward; returns (loss, bce, recon, logits, grads) where grads[name]
    is None for parameters the step does not reach (they are skipped by AdamW)."""
    names = [n for n, t in P.items() if t.requires_grad]
    loss, bce, recon, logits = total_loss(P, fe, x, y, w, alpha, beta, **kw)
    gs = torch.autograd.grad(loss, [P[n] for n in names], allow_unused=True)
    return loss.detach(), bce.detach(), recon.detach(), logits.detach(), dict(zip(names, gs))


# --------------------------------------------------------------------------------------
# optimizer
# --------------------------------------------------------------------------------------
class AdamWRef:
    """torch.optim.AdamW(params, lr=1e-3) as main.py:630 / :671 builds it, restated.

    Defaults (torch): betas (0.9, 0.999), eps 1e-8, weight_decay 1e-2, amsgrad False.
    Per tensor: ``step`` counts only the steps in which the tensor had a gradient; tensors
    whose grad is None are skipped entirely (no decay).  Checked against torch.optim.AdamW in
    tests/test_oracle_unit.py.
    """

    def __init__(self, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=1e-2):
        self.lr, self.b1, self.b2, self.eps, self.wd = lr, beta1, beta2, eps, weight_decay
        self.state: Dict[str, dict] = {}

    @torch.no_grad()
    def step(self, P: Dict[str, torch.Tensor], grads: Dict[str, Optional[torch.Tensor]]):
        for name, g in grads.items():
            if g is None:
                continue
            p = P[name]
            st = self.state.setdefault(name, dict(step=0, m=torch.zeros_like(p), v=torch.zeros_like(p)))
            st["step"] += 1
            t = st["step"]
            p.mul_(1.0 - self.lr * self.wd)
            st["m"].lerp_(g, 1.0 - self.b1)
            st["v"].mul_(self.b2).addcmul_(g, g, value=1.0 - self.b2)
            bc1 = 1.0 - self.b1 ** t
            bc2 = 1.0 - self.b2 ** t
            denom = (st["v"].sqrt() / math.sqrt(bc2)).add_(self.eps)
            p.addcdiv_(st["m"], denom, value=-(self.lr / bc1))


def train_step(P, fe, opt: AdamWRef, x, y, w, alpha, beta, **kw):
    """main.py:164-183 -- forward, backward, AdamW.  Returns (bce, recon, logits)."""
    loss, bce, recon, logits, grads = loss_and_grads(P, fe, x, y, w, alpha, beta, **kw)
    opt.step(P, grads)
    return bce, recon, logits


@torch.no_grad()
def save_embeddings(P, fe: FrontEnd, batch_size: int = 96) -> np.ndarray:
    """main.py:462-479 -- eval-mode get_node_embeddings(ids[:,None]) for ids 1..N -> float32 [N,d]."""
    ids = torch.arange(1, fe.n_nodes + 1, dtype=torch.long).view(-1, 1)
    chunks = []
    for j in range(math.ceil(len(ids) / batch_size)):
        xb = ids[j * batch_size:(j + 1) * batch_size]
        rows, _ = node_embeddings(P, fe, xb.reshape(-1), random_chrom=0)
        chunks.append(rows.view(xb.shape[0], 1, -1).numpy())
    return np.concatenate(chunks, axis=0)[:, 0, :]
