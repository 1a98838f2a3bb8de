"""fp64 torch.autograd restatement of the four reference graphs, used to cross-check the C oracle's
hand-written backward.  Mirrors Model.py:55-74 (batch views), TransE.py:11-51, TransH.py:12-69,
TransR.py:16-75, TransD.py:23-84 op for op (l2_normalize = x * rsqrt(max(sum x^2, 1e-12)))."""
import torch


def l2n(x):
    return x * torch.rsqrt(torch.clamp((x * x).sum(-1, keepdim=True), min=1e-12))


def calc(h, t, r):
    return (l2n(h) + l2n(r) - l2n(t)).abs()


def loss_fn(model, P, bh, bt, br, B, N, margin, De, Dr, negative_rel=0):
    bh, bt, br = [torch.as_tensor(x, dtype=torch.long) for x in (bh, bt, br)]
    ph, pt, pr = bh[:B].view(B, 1), bt[:B].view(B, 1), br[:B].view(B, 1)
    nh, nt, nr = [x[B:].view(N, B).t() for x in (bh, bt, br)]
    ent, rel = P["ent_embeddings"], P["rel_embeddings"]
    if model == "transe":
        p = calc(ent[ph], ent[pt], rel[pr])
        n = calc(ent[nh], ent[nt], rel[nr])
    elif model == "transh":
        def tr(e, w):
            w = l2n(w)
            return e - (e * w).sum(-1, keepdim=True) * w
        W = P["normal_vectors"]
        p = calc(tr(ent[ph], W[pr]), tr(ent[pt], W[pr]), rel[pr])
        n = calc(tr(ent[nh], W[nr]), tr(ent[nt], W[nr]), rel[nr])
    elif model == "transr":
        M = P["transfer_matrix"]
        pm = M[pr].view(B, De, Dr)
        p = calc(ent[ph] @ pm, ent[pt] @ pm, rel[pr])
        if negative_rel == 0:
            n = calc(ent[nh] @ pm, ent[nt] @ pm, rel[nr])
        else:
            # reference reshapes lookup(transfer_matrix, neg_r) to [-1,De,Dr] (TransR.py:62), which
            # only broadcasts for N == 1; the per-negative meaning is a matrix per (b,k)
            nm = M[nr].view(B, N, De, Dr)
            n = calc((ent[nh].unsqueeze(2) @ nm).squeeze(2), (ent[nt].unsqueeze(2) @ nm).squeeze(2), rel[nr])
    elif model == "transd":
        et, rt = P["ent_transfer"], P["rel_transfer"]
        def tr(e, ep, rp):
            return e + (e * ep).sum(-1, keepdim=True) * rp
        p = calc(tr(ent[ph], et[ph], rt[pr]), tr(ent[pt], et[pt], rt[pr]), rel[pr])
        n = calc(tr(ent[nh], et[nh], rt[nr]), tr(ent[nt], et[nt], rt[nr]), rel[nr])
    p_score = p.sum(-1, keepdim=True)
    n_score = n.sum(-1, keepdim=True)
    return torch.clamp(p_score - n_score + margin, min=0).mean()


def loss_and_grads(model, params, bh, bt, br, B, N, margin, De, Dr, negative_rel=0, dtype=torch.float64):
    P = {k: torch.tensor(v, dtype=dtype, requires_grad=True) for k, v in params.items()}
    loss = loss_fn(model, P, bh, bt, br, B, N, margin, De, Dr, negative_rel)
    loss.backward()
    return float(loss.detach()), {k: (v.grad.numpy() if v.grad is not None else None) for k, v in P.items()}


def near_kink_rows(model, params, bh, bt, br, B, N, De, Dr, tol=1e-6, negative_rel=0):
    """Rows of every table whose gradient a sign flip of d|e|/de could change: the score is sum |e| with
    e = h^ + r^ - t^ (TransE.py:15), whose derivative jumps at e = 0, so an element of e within fp32 rounding
    of zero gets sign +1 from one evaluation order and -1 (or 0) from another.  Evaluated here in fp64 from the
    given parameters.  Returns ({table: set(rows)}, number of elements with |e| < tol).  Only rows of triples
    whose hinge could be active matter, but every near-zero element is reported (a superset)."""
    import numpy as np
    P = {k: torch.tensor(np.asarray(v), dtype=torch.float64) for k, v in params.items()}
    bh, bt, br = [torch.as_tensor(np.asarray(x), dtype=torch.long) for x in (bh, bt, br)]
    ent, rel = P["ent_embeddings"], P["rel_embeddings"]
    n_tr = B * (1 + N)
    pos_of = torch.arange(n_tr) % B                      # triple j belongs to positive j % B (Base.cpp:109-139)
    if model == "transe":
        e = l2n(ent[bh]) + l2n(rel[br]) - l2n(ent[bt])
    elif model == "transh":
        w = l2n(P["normal_vectors"][br])
        tr = lambda x: x - (x * w).sum(-1, keepdim=True) * w
        e = l2n(tr(ent[bh])) + l2n(rel[br]) - l2n(tr(ent[bt]))
    elif model == "transd":
        et, rt = P["ent_transfer"], P["rel_transfer"]
        tr = lambda ids: ent[ids] + (ent[ids] * et[ids]).sum(-1, keepdim=True) * rt[br]
        e = l2n(tr(bh)) + l2n(rel[br]) - l2n(tr(bt))
    else:   # transr: the matrix is the POSITIVE's when negative_rel == 0 (TransR.py:57-60)
        M = P["transfer_matrix"]
        mr = br if negative_rel else br[pos_of]
        e = torch.empty((n_tr, Dr), dtype=torch.float64)
        for r in torch.unique(mr).tolist():
            idx = torch.nonzero(mr == r).squeeze(1)
            m = M[r].view(De, Dr)
            e[idx] = l2n(ent[bh[idx]] @ m) + l2n(rel[br[idx]]) - l2n(ent[bt[idx]] @ m)
    near = (e.abs() < tol).any(-1)
    groups = torch.unique(pos_of[near])                  # a flip in any triple of a group reaches the group's shared rows
    member = torch.isin(pos_of, groups)
    rows = {k: set() for k in params}
    hs, ts, rs = bh[member].tolist(), bt[member].tolist(), br[member].tolist()
    rows["ent_embeddings"].update(hs); rows["ent_embeddings"].update(ts)
    rows["rel_embeddings"].update(rs)
    for k in ("normal_vectors", "rel_transfer", "transfer_matrix"):
        if k in rows:
            rows[k].update(rs)
    if "ent_transfer" in rows:
        rows["ent_transfer"].update(hs); rows["ent_transfer"].update(ts)
    return rows, int((e.abs() < tol).sum())
