"""Synthetic DBoW2-style vocabulary trees for the ComputeBoW tests (ORBvoc.txt is not available offline): k-ary tree of
depth L (optionally ragged), random 256-bit node descriptors, idf-like positive weights with a few stopped (zero) words."""
import numpy as np


def make_vocab(k=10, L=3, seed=0, ragged=False, stopped=0.05):
    rng = np.random.default_rng(seed)
    children = [[]]
    depth = [0]
    frontier = [0]
    for lvl in range(1, L + 1):
        nxt = []
        for nd in frontier:
            kk = k if not ragged else int(rng.integers(1, k + 1))
            if ragged and lvl > 1 and rng.random() < 0.15:
                continue  # an early leaf
            for _ in range(kk):
                children.append([])
                depth.append(lvl)
                children[nd].append(len(children) - 1)
                nxt.append(len(children) - 1)
        frontier = nxt
    nn = len(children)
    node_desc = rng.integers(0, 256, (nn, 32), dtype=np.uint8)
    # children of one node: make some of them equidistant twins to exercise the first-minimum rule
    for nd in range(nn):
        ch = children[nd]
        if len(ch) >= 2 and rng.random() < 0.2:
            node_desc[ch[1]] = node_desc[ch[0]]
    node_word = np.full(nn, 0, np.int32)
    node_weight = np.zeros(nn, np.float64)
    w = 0
    for nd in range(nn):
        if not children[nd]:
            node_word[nd] = w
            w += 1
            node_weight[nd] = 0.0 if rng.random() < stopped else float(rng.uniform(0.1, 9.0))
    return children, node_desc, node_weight, node_word, L
