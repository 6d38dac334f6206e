"""GPU parity of the keyframe-database query (SURVEY 8f.4 / place recognition): orbgpu_bow_score through the C ABI
against the oracle (bit-exact doubles, exact counts) for every scoring DBoW2 offers that the kernel implements, the
host mirror DBoW2Database against a literal transcription of DetectNBestCandidates, and the error paths.  The
reference has no test or fixture for this path and ORBvoc.txt is absent: parity is unpinned beyond the oracle."""
import ctypes as C
import io

import numpy as np
import pytest

from db_literal import LiteralDatabase, flatten, random_database

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fe():
    from orb_slam3_amd.features import ORBFeatureExtractor
    e = ORBFeatureExtractor(320, 240, 1.2, 4, 20, 7, max_features=300)
    yield e
    e.close()


def gpu_score(fe, qw, qv, offs, words, vals, scoring, qc=None, kc=None):
    from orb_slam3_amd import _lib
    n_kf = len(offs) - 1
    shared = np.full(max(n_kf, 1), 12345, np.uint64)
    score = np.full(max(n_kf, 1), -7.0, np.float64)
    a = _lib.BowScoreArgs(_lib.ptr(qw), _lib.ptr(qv), _lib.ptr(qc), len(qw), n_kf, _lib.ptr(offs), _lib.ptr(words),
                          _lib.ptr(vals), _lib.ptr(kc), scoring, 0)
    _lib.check(fe._lib.orbgpu_bow_score(fe.handle, C.byref(a), _lib.ptr(shared), _lib.ptr(score)), fe.handle)
    return shared[:n_kf], score[:n_kf]


def query_arrays(bow, wf=None):
    qw = np.asarray(sorted(bow), np.uint32)
    qv = np.asarray([bow[w] for w in sorted(bow)], np.float64)
    qc = None if wf is None else np.asarray([wf[w] for w in sorted(bow)], np.uint32)
    return qw, qv, qc


@pytest.mark.parametrize("scoring", [0, 1, 2, 4, 5])
@pytest.mark.parametrize("shape", [(300, 2000, 150, False), (64, 400, 300, True), (1000, 100000, 900, False)])
def test_bow_score_parity(fe, oracle, scoring, shape):
    n_kf, vocab, wpk, dense = shape
    bows, freqs = random_database(7 * scoring + n_kf, n_kf, vocab, wpk, weighted_counts=True, dense_overlap=dense)
    ids, offs, words, vals, counts = flatten(bows, freqs)
    for qk in (ids[1], ids[n_kf // 2 + 1]):          # neither is one of the possibly empty keyframes (k % 7 == 3)
        qw, qv, qc = query_arrays(bows[qk], {w: c + 2 for w, c in freqs[qk].items()})
        for use_counts in (True, False):
            a = (qc, counts) if use_counts else (None, None)
            sh_o, sc_o = oracle.bow_score(qw, qv, offs, words, vals, scoring, *a)
            sh_g, sc_g = gpu_score(fe, qw, qv, offs, words, vals, scoring, *a)
            assert np.array_equal(sh_g, sh_o)
            assert np.array_equal(sc_g.view(np.uint64), sc_o.view(np.uint64))   # bit-exact doubles
        assert sh_g.max() > 0


def test_bow_score_edges(fe, oracle):
    bows, _ = random_database(3, 40, 300, 60)
    ids, offs, words, vals, _ = flatten(bows)
    # empty query: nothing shared, score of two disjoint vectors
    sh, sc = gpu_score(fe, np.zeros(0, np.uint32), np.zeros(0), offs, words, vals, 0)
    assert not sh.any() and not sc.any()
    # a database whose keyframes are all empty
    e_offs = np.zeros(6, np.int64)
    qw, qv, _ = query_arrays(bows[ids[0]])
    sh, sc = gpu_score(fe, qw, qv, e_offs, np.zeros(0, np.uint32), np.zeros(0), 0)
    assert len(sh) == 5 and not sh.any() and not sc.any()
    # zero keyframes
    sh, sc = gpu_score(fe, qw, qv, np.zeros(1, np.int64), np.zeros(0, np.uint32), np.zeros(0), 0)
    assert len(sh) == 0
    # a query with more words than the LDS copy holds (binary search in global memory): 20000 words
    rng = np.random.default_rng(5)
    big_words = np.sort(rng.choice(200000, 20000, replace=False)).astype(np.uint32)
    big_vals = rng.uniform(0.1, 1.0, 20000)
    k_words = [np.sort(rng.choice(big_words, 500, replace=False)) for _ in range(30)]
    k_offs = np.concatenate([[0], np.cumsum([len(w) for w in k_words])]).astype(np.int64)
    k_w = np.concatenate(k_words).astype(np.uint32)
    k_v = rng.uniform(0.1, 1.0, len(k_w))
    sh_o, sc_o = oracle.bow_score(big_words, big_vals, k_offs, k_w, k_v, 0)
    sh_g, sc_g = gpu_score(fe, big_words, big_vals, k_offs, k_w, k_v, 0)
    assert np.array_equal(sh_g, sh_o) and np.array_equal(sc_g.view(np.uint64), sc_o.view(np.uint64))
    assert (sh_g == 500).all()
    # self-score of an L1-normalised vector is 1 up to rounding (a size-independent property)
    sh, sc = gpu_score(fe, qw, qv, offs, words, vals, 0)
    assert abs(sc[0] - 1.0) < 1e-12 and sh[0] == len(qw)


def test_bow_score_errors(fe):
    from orb_slam3_amd import _lib
    bows, _ = random_database(4, 10, 100, 20)
    ids, offs, words, vals, _ = flatten(bows)
    qw, qv, _ = query_arrays(bows[ids[0]])
    with pytest.raises(_lib.OrbGpuError, match="scoring"):
        gpu_score(fe, qw, qv, offs, words, vals, 3)            # KL is not offered
    with pytest.raises(_lib.OrbGpuError, match="ascending"):
        gpu_score(fe, qw[::-1].copy(), qv, offs, words, vals, 0)
    bad = words.copy()
    bad[[0, 1]] = bad[[1, 0]]
    with pytest.raises(_lib.OrbGpuError, match="ascending"):
        gpu_score(fe, qw, qv, offs, bad, vals, 0)
    bad_offs = offs.copy()
    bad_offs[2] = bad_offs[1] - 1
    with pytest.raises(_lib.OrbGpuError, match="non-decreasing"):
        gpu_score(fe, qw, qv, bad_offs, words, vals, 0)


def test_database_mirror_vs_literal(fe):
    """DBoW2Database (host mirror over the GPU query) against the literal transcription: Append / Erase, the stream
    round trip, SearchWordSharingKeyFrames and DetectNBestCandidates with a covisibility graph and two maps."""
    from orb_slam3_amd.database import DBoW2Database
    vocab = 800
    bows, freqs = random_database(11, 120, vocab, 90, dense_overlap=True)
    ids = sorted(bows)
    rng = np.random.default_rng(12)
    cov = {k: [int(x) for x in rng.choice(ids, 10, replace=False) if x != k] for k in ids}
    map_id = {k: int(rng.integers(0, 2)) for k in ids}
    db, lit = DBoW2Database(fe, vocab), LiteralDatabase(vocab)
    for k in ids:
        db.Append(k, bows[k])
        lit.Append(k, bows[k], freqs[k])
    for k in ids[5:15]:
        db.Erase(k)
        lit.Erase(k)
    assert db.inverted_file == lit.inverted_file
    buf = io.BytesIO()
    db.Serialize(buf)
    buf.seek(0)
    db2 = DBoW2Database.Deserialize(buf, fe, {k: bows[k] for k in lit.bow})
    assert db2.inverted_file == lit.inverted_file
    n_loop = n_merge = 0
    for d in (db, db2):
        for qk in ids[:40:3]:
            q_bow, q_wf = bows[qk], freqs[qk]
            assert d.SearchWordSharingKeyFrames(q_bow) == lit.SearchWordSharingKeyFrames(q_wf)
            for count in (2, (1 << 64) - 1):
                got = d.DetectNBestCandidates(q_bow, cov.__getitem__, map_id.__getitem__, 0, count)
                want = lit.DetectNBestCandidates(q_bow, q_wf, 0, cov.__getitem__, map_id.__getitem__, 0, count)
                assert got == want
                n_loop += len(got[0])
                n_merge += len(got[1])
            assert d.DetectRelocCandidates(q_bow, cov.__getitem__, map_id.__getitem__, 0) == want[0]
    assert n_loop > 0 and n_merge > 0


def test_resident_database_churn(fe, oracle):
    """orbgpu_bowdb_*: appends, replacements and erasures -- enough of them that the arena grows and is squeezed
    several times -- then every query answer against the oracle on the surviving set."""
    from orb_slam3_amd.database import DBoW2Database
    vocab = 5000
    bows, freqs = random_database(21, 700, vocab, 400, weighted_counts=True)
    ids = sorted(bows)
    rng = np.random.default_rng(22)
    db = DBoW2Database(fe, vocab, scoring=0, with_counts=True)
    live = {}
    for step in range(3):
        for k in ids:
            db.Append(k, bows[k], freqs[k])
            live[k] = (bows[k], freqs[k])
        for k in rng.choice(ids, 450, replace=False):       # most of the data becomes holes
            db.Erase(int(k))
            live.pop(int(k))
        for k in rng.choice(sorted(live), 40, replace=False):   # replacement: same id, another keyframe's vector
            src = ids[int(rng.integers(0, len(ids)))]
            db.Append(int(k), bows[src], freqs[src])
            live[int(k)] = (bows[src], freqs[src])
        lids, offs, words, vals, counts = flatten({k: v[0] for k, v in live.items()}, {k: v[1] for k, v in live.items()})
        for qk in (ids[0], ids[123]):
            qw, qv, qc = query_arrays(bows[qk], freqs[qk])
            got_ids, sh_g, sc_g = db.ScoreAll(bows[qk], freqs[qk])
            sh_o, sc_o = oracle.bow_score(qw, qv, offs, words, vals, 0, qc, counts)
            assert got_ids == lids
            assert np.array_equal(sh_g, sh_o) and np.array_equal(sc_g.view(np.uint64), sc_o.view(np.uint64))
    # an empty database answers with nothing
    for k in list(live):
        db.Erase(k)
    assert db.ScoreAll(bows[ids[0]]) [0] == []
    db.close()
