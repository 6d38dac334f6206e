"""The one datum the reference's own tests hold for this path -- the 11 keypoints and three window queries of
/root/reference/test/features/feature_tests.cpp:21-50 (FeatureTests.ListFeaturesInAreaReturnsCorrectIndices) -- put
through the HIP path: orbgpu_match_window (plain scan), orbgpu_match_window_grid_segments_device (search grid) and the
cell membership of orbgpu_grid_build_device.  Expected values are the reference's ASSERT_EQs (1 / 0 / 2 hits) and the
cells PosInGrid's body (features.cpp:150-158) gives by hand: 640 / 64 = 480 / 48 = 10 px per cell.

How a candidate LIST is read out of a matcher that returns one 'from' per 'to': every keypoint gets a descriptor of its
own (pairwise distance >= 16) and the query is issued once per keypoint j with that keypoint's descriptor, TH 0: 'to' row
j then matches 'from' j iff j is among the query's candidates (a single candidate within TH passes the ratio test
against UINT_MAX, second_nearest_neighbor_matcher.hpp:41-55), and nothing else can match."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# feature_tests.cpp:21-31: (x, y, level)
REF_KEYPOINTS = [(120, 62, 0), (20, 42, 1), (57, 259, 0), (185, 336, 2), (198, 78, 0), (1, 1, 0), (17, 451, 5),
                 (301, 42, 0), (510, 79, 1), (604, 21, 0), (632, 262, 0)]
# feature_tests.cpp:42-50: (point, window, minLevel, maxLevel) -> number of hits the reference asserts; which rows they
# are follows from the coordinates (row 4 = (198, 78) is 8 / 2 px away, row 0 = (120, 62) is 70 / 18 px away)
REF_QUERIES = [((190.0, 80.0), 10, 0, 2, [4]), ((190.0, 80.0), 10, 1, 2, []), ((190.0, 80.0), 100, 0, 2, [0, 4])]


def _descriptors(n):
    d = np.zeros((n, 32), np.uint8)
    for j in range(n):
        d[j, 2 * j:2 * j + 2] = 0xff          # 16 private bits per row: pairwise distance 32
    return d


@pytest.fixture(scope="module")
def fe():
    from orb_slam3_amd.features import ORBFeatureExtractor
    e = ORBFeatureExtractor(640, 480, 1.2, 8, 20, 7, max_features=1000)
    yield e
    e.close()


@pytest.mark.parametrize("query", REF_QUERIES)
def test_reference_window_queries_plain_scan(fe, query):
    """orbgpu_match_window, host pointers: the call behind SNNMatcher<ProjectionSearchIterator>."""
    from orb_slam3_amd import _lib
    lib = _lib.load()
    (px, py), win, lmin, lmax, want = query
    n = len(REF_KEYPOINTS)
    desc = _descriptors(n)
    from_xy = np.array([(x, y) for x, y, _ in REF_KEYPOINTS], np.float64)
    from_lv = np.array([l for _, _, l in REF_KEYPOINTS], np.int32)
    to_xy = np.tile(np.array([[px, py]], np.float64), (n, 1))
    to_win = np.full(n, win, np.uint64)
    to_lmin, to_lmax = np.full(n, lmin, np.int32), np.full(n, lmax, np.int32)
    a = _lib.WindowArgs(_lib.ptr(desc), _lib.ptr(to_xy), _lib.ptr(to_win), _lib.ptr(to_lmin), _lib.ptr(to_lmax), None, n,
                        _lib.ptr(desc), _lib.ptr(from_xy), _lib.ptr(from_lv), n, 0, 2.0)
    idx, dist = np.full(n, -7, np.int32), np.zeros(n, np.uint32)
    assert lib.orbgpu_match_window(fe.handle, C.byref(a), _lib.ptr(idx), _lib.ptr(dist)) == 0, lib.orbgpu_last_error(fe.handle)
    hits = [j for j in range(n) if idx[j] == j]
    assert hits == want
    assert all(idx[j] == -1 for j in range(n) if j not in want)
    assert len(hits) == len(want)                         # the reference's ASSERT_EQ(1 / 0 / 2, result.size())


def test_reference_grid_cells_and_queries(fe):
    """orbgpu_grid_build_device: cell membership of the 11 keypoints; then the three queries over that grid
    (orbgpu_match_window_grid_segments_device), one segment per query."""
    import torch
    from orb_slam3_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    n = len(REF_KEYPOINTS)
    xy = np.array([(x, y) for x, y, _ in REF_KEYPOINTS], np.float64)
    lv = np.array([l for _, _, l in REF_KEYPOINTS], np.int32)
    d_xy = torch.from_numpy(xy).to(dev)
    d_rng = torch.tensor([[0, n]], dtype=torch.int32, device=dev)
    d_cs = torch.full((1, 3073), -5, dtype=torch.int32, device=dev)
    d_it = torch.full((1, n), -5, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    assert lib.orbgpu_grid_build_device(fe.handle, d_xy.data_ptr(), d_rng.data_ptr(), 1, n, 640.0, 480.0, d_cs.data_ptr(),
                                        d_it.data_ptr()) == 0, lib.orbgpu_last_error(fe.handle)
    assert lib.orbgpu_synchronize(fe.handle) == 0
    cs, it = d_cs.cpu().numpy()[0], d_it.cpu().numpy()[0]
    # PosInGrid (features.cpp:150-158) by hand: posX = x / 10, posY = y / 10, flat cell = posX * 48 + posY (grid[64][48])
    want_cell = [(x // 10) * 48 + (y // 10) for x, y, _ in REF_KEYPOINTS]
    assert want_cell == [582, 100, 265, 897, 919, 0, 93, 1444, 2455, 2882, 3050]
    got_cell = {}
    for c in range(3072):
        for k in range(cs[c], cs[c + 1]):
            got_cell[int(it[k])] = c
    assert [got_cell[j] for j in range(n)] == want_cell
    assert cs[0] == 0 and cs[3072] == n

    # the three queries as three segments that share the 'from' rows (the 11 keypoints)
    S = len(REF_QUERIES)
    desc = _descriptors(n)
    to_desc = np.tile(desc, (S, 1))
    to_xy = np.concatenate([np.tile(np.array([q[0]], np.float64), (n, 1)) for q in REF_QUERIES])
    to_win = np.concatenate([np.full(n, q[1], np.int64) for q in REF_QUERIES])
    to_lmin = np.concatenate([np.full(n, q[2], np.int32) for q in REF_QUERIES])
    to_lmax = np.concatenate([np.full(n, q[3], np.int32) for q in REF_QUERIES])
    # every segment gets a copy of the 'from' rows: segments that share 'from' rows would share their uniqueness claims
    from_desc, from_xy, from_lv = np.tile(desc, (S, 1)), np.tile(xy, (S, 1)), np.tile(lv, S)
    segs = np.array([[s * n, n, s * n, n] for s in range(S)], np.int32)
    t = {k: torch.from_numpy(v).to(dev) for k, v in dict(to_desc=to_desc, to_xy=to_xy, to_win=to_win, to_lmin=to_lmin,
                                                         to_lmax=to_lmax, from_desc=from_desc, from_xy=from_xy,
                                                         from_lv=from_lv, segs=segs).items()}
    a = _lib.WindowArgs(t["to_desc"].data_ptr(), t["to_xy"].data_ptr(), t["to_win"].data_ptr(), t["to_lmin"].data_ptr(),
                        t["to_lmax"].data_ptr(), None, S * n, t["from_desc"].data_ptr(), t["from_xy"].data_ptr(),
                        t["from_lv"].data_ptr(), S * n, 0, 2.0)
    outs = []
    for grid in (False, True):
        oi = torch.full((S * n,), -7, dtype=torch.int32, device=dev)
        od = torch.zeros(S * n, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        if grid:
            rc = lib.orbgpu_match_window_grid_segments_device(fe.handle, C.byref(a), t["segs"].data_ptr(), S, n, n, 640.0,
                                                              480.0, oi.data_ptr(), od.data_ptr())
        else:
            rc = lib.orbgpu_match_window_segments_device(fe.handle, C.byref(a), t["segs"].data_ptr(), S, n, oi.data_ptr(),
                                                         od.data_ptr())
        assert rc == 0, lib.orbgpu_last_error(fe.handle)
        assert lib.orbgpu_synchronize(fe.handle) == 0
        outs.append(oi.cpu().numpy().reshape(S, n))
    for got in outs:
        for s, q in enumerate(REF_QUERIES):
            hits = [j for j in range(n) if got[s, j] == j]
            assert hits == q[4], (s, got[s])
            assert all(got[s, j] == -1 for j in range(n) if j not in q[4])
