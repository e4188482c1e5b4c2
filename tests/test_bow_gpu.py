"""GPU parity of Frame::ComputeBoW (DBoW2 transform on a flat vocabulary) vs the CPU oracle: exact, including the f64
BowVector values (same additions in the same order)."""
import numpy as np
import pytest

import bow_vocab

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k,L,ragged,levelsup,n", [(10, 3, False, 2, 1000), (4, 5, True, 4, 700), (3, 2, False, 4, 50), (70, 2, False, 1, 300)])
def test_compute_bow_exact(k, L, ragged, levelsup, n):
    import psl_slam_amd as P
    import oracle_lib
    vocab = bow_vocab.make_vocab(k, L, seed=10 + k, ragged=ragged)
    rng = np.random.default_rng(5)
    desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    # half of the features are exact copies of leaf descriptors: zero distances and ties
    leaves = [i for i, c in enumerate(vocab[0]) if not c]
    desc[::2] = vocab[1][rng.choice(leaves, len(desc[::2]))]
    got = P.ORBVocabulary(*vocab).transform(desc, levelsup)
    ref = oracle_lib.compute_bow(*vocab, desc, levelsup)
    for key in ("word", "weight", "nid", "bow_id", "bow_val", "fv_node", "fv_start", "fv_idx"):
        assert got[key].tobytes() == ref[key].tobytes(), key
    assert len(ref["bow_id"]) > 0


def test_compute_bow_on_real_orb_descriptors_and_empty():
    import psl_slam_amd as P
    import oracle_lib
    import synth_frames as sf
    kps, desc = oracle_lib.OracleORB()(sf.Scene(640, 480, "desk", seed=11).gray(0))
    vocab = bow_vocab.make_vocab(10, 4, seed=4)
    V = P.ORBVocabulary(*vocab)
    got, ref = V.transform(desc, 4), oracle_lib.compute_bow(*vocab, desc, 4)
    for key in ref:
        assert got[key].tobytes() == ref[key].tobytes(), key
    e = V.transform(np.zeros((0, 32), np.uint8))
    assert len(e["bow_id"]) == 0 and len(e["fv_node"]) == 0


def test_vocab_validation():
    import psl_slam_amd as P
    children, nd, nw, nwd, L = bow_vocab.make_vocab(3, 2, seed=1)
    bad = [list(c) for c in children]
    bad[1].append(2)  # node 2 gets a second parent
    with pytest.raises(P.PslfeError):
        P.ORBVocabulary(bad, nd, nw, nwd, L)
    with pytest.raises(P.PslfeError):
        P.ORBVocabulary(children, nd, nw, nwd, 1)  # leaves deeper than L
