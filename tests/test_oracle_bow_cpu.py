"""Oracle of Frame::ComputeBoW (oracle/bow_oracle.cpp) vs a plain-Python restatement of DBoW2's transform."""
import numpy as np

import bow_vocab
import oracle_lib


def _ham(a, b):
    return int(np.unpackbits(np.bitwise_xor(a, b)).sum())


def _py_transform(vocab, desc, levelsup):
    children, nd, nw, nwd, L = vocab
    bow, fv = {}, {}
    per = []
    for i, f in enumerate(desc):
        node, lvl, nid = 0, 0, 0
        while children[node]:
            lvl += 1
            best, bd = children[node][0], _ham(f, nd[children[node][0]])
            for c in children[node][1:]:
                d = _ham(f, nd[c])
                if d < bd:
                    best, bd = c, d
            node = best
            if lvl == L - levelsup:
                nid = node
        per.append((int(nwd[node]), float(nw[node]), nid))
        if nw[node] > 0:
            bow[int(nwd[node])] = bow.get(int(nwd[node]), 0.0) + float(nw[node]) if int(nwd[node]) in bow else float(nw[node])
            fv.setdefault(nid, []).append(i)
    norm = 0.0
    for k in sorted(bow):
        norm += abs(bow[k])
    if norm > 0:
        for k in bow:
            bow[k] /= norm
    return per, bow, fv


def test_compute_bow_against_plain_python():
    rng = np.random.default_rng(1)
    for k, L, ragged, levelsup in ((10, 3, False, 2), (4, 5, True, 4), (3, 2, False, 4)):
        vocab = bow_vocab.make_vocab(k, L, seed=k + L, ragged=ragged)
        desc = rng.integers(0, 256, (300, 32), dtype=np.uint8)
        o = oracle_lib.compute_bow(*vocab, desc, levelsup)
        per, bow, fv = _py_transform(vocab, desc, levelsup)
        assert [(int(a), float(b), int(c)) for a, b, c in zip(o["word"], o["weight"], o["nid"])] == per
        assert list(o["bow_id"]) == sorted(bow)
        np.testing.assert_array_equal(o["bow_val"], [bow[k] for k in sorted(bow)])
        assert abs(o["bow_val"].sum() - 1.0) < 1e-12
        assert list(o["fv_node"]) == sorted(fv)
        for g, nd in enumerate(o["fv_node"]):
            assert list(o["fv_idx"][o["fv_start"][g]:o["fv_start"][g + 1]]) == fv[int(nd)]
        if levelsup >= L:
            assert list(o["fv_node"]) == [0]  # nid_level <= 0: everything under the root


def test_compute_bow_feeds_search_by_bow():
    """the FeatureVector layout is exactly what SearchByBoW takes (fidx + node runs)"""
    vocab = bow_vocab.make_vocab(6, 3, seed=2, stopped=0.0)
    rng = np.random.default_rng(3)
    dF = rng.integers(0, 256, (200, 32), dtype=np.uint8)
    dK = dF[rng.permutation(200)[:150]].copy()
    dK[:, 7] ^= 1
    oF, oK = oracle_lib.compute_bow(*vocab, dF, 2), oracle_lib.compute_bow(*vocab, dK, 2)
    startF = {int(nd): (int(oF["fv_start"][g]), int(oF["fv_start"][g + 1] - oF["fv_start"][g])) for g, nd in enumerate(oF["fv_node"])}
    runs, qd = [], []
    for g, nd in enumerate(oK["fv_node"]):
        if int(nd) in startF:
            for i in oK["fv_idx"][oK["fv_start"][g]:oK["fv_start"][g + 1]]:
                runs.append(startF[int(nd)]); qd.append(dK[i])
    nm, match, _ = oracle_lib.search_by_bow(dF, np.zeros(200, np.float32), oF["fv_idx"], np.array(runs, np.int32), np.array(qd, np.uint8),
                                            np.zeros(len(runs), np.float32), 0.9, False)
    assert nm > 100  # near-duplicates end up under the same node and match
