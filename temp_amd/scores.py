"""Decoders with the reference's call signatures (utils/scores.py:4-55).

`mode` is 'single' (one candidate per triple), 'tail' (candidate axis on the object: o is
(P,K,D)) or 'head' (candidate axis on the subject: s is (P,K,D)).

Formulated as "fold the two known arguments into one query vector, then reduce against the
candidates": every scorer is  score[p,k] = reduce_d f(query[p,d], cand[p,k,d]).  That is the shape
the fused all-entity pass + scorer + cross-entropy kernel consumes; today the
reduction is plain tensor algebra on the embeddings' device.
"""
import torch


def _split(x):
    half = x.shape[-1] // 2
    return x[..., :half], x[..., half:]


def _dot_candidates(query, cand, mode):
    if mode in ('tail', 'head'):
        return (query.unsqueeze(1) * cand).sum(dim=-1)
    return (query * cand).sum(dim=-1)


def distmult(s, r, o, mode='single'):
    """<s, r, o> trilinear product."""
    if mode == 'head':
        return _dot_candidates(r * o, s, mode)
    return _dot_candidates(s * r, o, mode)


def complex(head, relation, tail, mode='single'):  # noqa: A001 - the reference's name
    """Re(<h, r, conj(t)>) with the embedding split into real / imaginary halves."""
    re_r, im_r = _split(relation)
    if mode == 'head':
        re_t, im_t = _split(tail)
        query = torch.cat([re_r * re_t + im_r * im_t, re_r * im_t - im_r * re_t], dim=-1)
        return _dot_candidates(query, head, mode)
    re_h, im_h = _split(head)
    query = torch.cat([re_h * re_r - im_h * im_r, re_h * im_r + im_h * re_r], dim=-1)
    return _dot_candidates(query, tail, mode)


def transE(head, relation, tail, mode='single'):
    """-|| h + r - t ||_1."""
    if mode == 'head':
        diff = head + (relation - tail).unsqueeze(1)
    elif mode == 'tail':
        diff = (head + relation).unsqueeze(1) - tail
    else:
        diff = head + relation - tail
    return -diff.abs().sum(dim=-1)


def simple(head, head_inv, rel, rel_inv, tail, tail_inv, mode='tail'):
    """SimplE: mean of the two DistMult directions (utils/scores.py:14-24)."""
    if mode == 'head':
        a = _dot_candidates(rel * tail_inv, head, mode)
        b = _dot_candidates(rel_inv * tail, head_inv, mode)
    else:
        a = _dot_candidates(head * rel, tail_inv, mode)
        b = _dot_candidates(head_inv * rel_inv, tail, mode)
    return (a + b) / 2


def bilinear_query(name, known, relation, mode):
    """Fold the known entity and the relation into ONE query vector q such that
    score(candidate) = <q, candidate>.  `known` is the subject for mode='tail' (candidates are
    objects) and the object for mode='head'.  Returns None for scorers that are not bilinear (transE)."""
    if name == "distmult":
        return known * relation
    if name == "complex":
        re_r, im_r = _split(relation)
        re_k, im_k = _split(known)
        if mode == "head":
            return torch.cat([re_r * re_k + im_r * im_k, re_r * im_k - im_r * re_k], dim=-1)
        return torch.cat([re_k * re_r - im_k * im_r, re_k * im_r + im_k * re_r], dim=-1)
    return None
