"""Bag-of-words transform oracle on a hand-made vocabulary with known answers (DBoW2 rules: first minimum on ties, word
weights added per feature, stopped words dropped, L1 normalisation in word order)."""
import numpy as np

import oracle_lib as O


def tiny_tree():
    # root 0 -> {1, 2}; 1 -> {3 (word 0), 4 (word 1)}; 2 -> {5 (word 2), 6 (word 3, stopped)}
    child_offset = np.array([0, 2, 4, 6, 6, 6, 6, 6], np.int32)
    children = np.array([1, 2, 3, 4, 5, 6], np.int32)
    d = np.zeros((7, 32), np.uint8)
    d[1] = 0x00; d[2] = 0xFF
    d[3] = 0x00; d[4, :4] = 0xFF            # below node 1: all-zero vs 32 bits set
    d[5] = 0xFF; d[6, :16] = 0xFF           # below node 2
    w = np.array([0, 0, 0, 2.0, 0.5, 1.25, 0.0])
    word = np.array([0, 0, 0, 0, 1, 2, 3], np.uint32)
    return child_offset, children, d, w, word


def test_tiny_vocabulary_known_answers():
    co, ch, d, w, word = tiny_tree()
    f = np.zeros((6, 32), np.uint8)
    f[1, :4] = 0xFF                 # -> node 1 -> word 1 (distance 0 to node 4)
    f[2] = 0xFF                     # -> node 2 -> word 2
    f[3, :16] = 0xFF                # tie at the root (128 vs 128) -> first child (node 1); then 128 vs 96 -> node 4 / word 1
    f[4, :24] = 0xFF                # -> node 2; 64 vs 64 tie -> first child node 5 / word 2
    f[5, :20] = 0xFF; f[5, 31] = 0  # -> node 2 (96 vs 160), then 96 vs 32 -> node 6: stopped word
    wid, nid, bw, bv, fn, ff = O.bow_transform(co, ch, d, w, word, 2, f, 1, 1, 1)     # nid level = L - levelsup = 1
    assert wid.tolist() == [0, 1, 2, 1, 2, 0xFFFFFFFF]
    assert nid.tolist() == [1, 1, 2, 1, 2, 0xFFFFFFFF]
    assert bw.tolist() == [0, 1, 2]
    tot = 2.0 + (0.5 + 0.5) + (1.25 + 1.25)
    assert bv.tolist() == [2.0 / tot, 1.0 / tot, 2.5 / tot]
    assert fn.tolist() == [1, 1, 1, 2, 2] and ff.tolist() == [0, 1, 3, 2, 4]
    # IDF / BINARY keep one copy; no normalisation with DOT_PRODUCT
    _, _, bw, bv, _, _ = O.bow_transform(co, ch, d, w, word, 2, f, 1, 0, 0)
    assert bw.tolist() == [0, 1, 2] and bv.tolist() == [2.0, 0.5, 1.25]
    # TF_IDF without normalisation divides by the number of distinct words; L2
    _, _, _, bv, _, _ = O.bow_transform(co, ch, d, w, word, 2, f, 1, 1, 0)
    assert bv.tolist() == [2.0 / 3, 1.0 / 3, 2.5 / 3]
    _, _, _, bv, _, _ = O.bow_transform(co, ch, d, w, word, 2, f, 1, 1, 2)
    nrm = np.sqrt(2.0 * 2.0 + 1.0 * 1.0 + 2.5 * 2.5)
    assert bv.tolist() == [2.0 / nrm, 1.0 / nrm, 2.5 / nrm]
    # levelsup >= L files everything under the root; levelsup = 0 under the leaf itself
    _, nid, _, _, fn, _ = O.bow_transform(co, ch, d, w, word, 2, f, 4, 1, 1)
    assert nid[:5].tolist() == [0] * 5 and fn.tolist() == [0] * 5
    _, nid, _, _, _, _ = O.bow_transform(co, ch, d, w, word, 2, f, 0, 1, 1)
    assert nid[:5].tolist() == [3, 4, 5, 4, 5]
    # no features
    wid, nid, bw, bv, fn, ff = O.bow_transform(co, ch, d, w, word, 2, np.zeros((0, 32), np.uint8), 1, 1, 1)
    assert len(wid) == len(bw) == len(fn) == 0


def test_random_vocabulary_invariants():
    rng = np.random.default_rng(2)
    parents, is_leaf, descs, weights = O.random_vocab(rng, 6, 4)
    n = len(parents)
    order = np.argsort(parents[1:], kind="stable") + 1
    co = np.concatenate([[0], np.cumsum(np.bincount(parents[1:], minlength=n))]).astype(np.int32)
    word = np.zeros(n, np.uint32); word[is_leaf] = np.arange(is_leaf.sum())
    f = rng.integers(0, 256, (500, 32), dtype=np.uint8)
    f[:50] = descs[rng.choice(np.flatnonzero(is_leaf), 50)]            # exact leaf descriptors
    wid, nid, bw, bv, fn, ff = O.bow_transform(co, order, descs, weights, word, 4, f, 2, 1, 1)
    kept = wid != 0xFFFFFFFF
    assert (np.diff(bw.astype(np.int64)) > 0).all() and set(bw.tolist()) == set(wid[kept].tolist())
    assert abs(bv.sum() - 1.0) < 1e-12 and (bv > 0).all()
    assert sorted(ff.tolist()) == np.flatnonzero(kept).tolist()
    key = fn.astype(np.int64) * 4096 + ff
    assert (np.diff(key) > 0).all()
    assert np.array_equal(fn, nid[ff])
    # a node id is an ancestor (or self) of the word's leaf
    leaf_of_word = np.flatnonzero(is_leaf)
    for i in np.flatnonzero(kept)[:100]:
        a = leaf_of_word[wid[i]]
        while a != nid[i] and a != 0:
            a = parents[a]
        assert a == nid[i]
