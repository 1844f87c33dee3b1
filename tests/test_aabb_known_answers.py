"""The five check groups of the reference's (never called) _test_aabb_intersection
(rt_raytracer.cpp:373-419) as known-answer tests of the restated Woo ray/box test.  They are the
only expectations the reference's authors wrote down for this path."""
import numpy as np


def test_aabb_groups(oracle_cls):
    orc = oracle_cls()
    mn, mx = np.array([-1, -1, -1], np.float32), np.array([1, 1, 1], np.float32)
    # 1. six axis rays from the centre
    for i in range(6):
        d = np.zeros(3, np.float32)
        d[i % 3] = 1.0 if i < 3 else -1.0
        assert orc.ray_aabb(np.zeros(3, np.float32), d, mn, mx), "origin test %d" % i
    # 2. six axis rays from +-5 towards the box
    for i in range(6):
        o = np.zeros(3, np.float32)
        o[i % 3] = -5.0 if i < 3 else 5.0
        assert orc.ray_aabb(o, -o, mn, mx), "axis test %d" % i
    rng = np.random.RandomState(0)
    # 3. "random to centre": integer division makes every origin (0,0,0) in the reference and the
    #    direction -normalize(0)*10 = NaN; the origin is inside, so the test returns true
    for _ in range(100):
        assert orc.ray_aabb(np.zeros(3, np.float32), np.full(3, np.nan, np.float32), mn, mx)
    # 4. random origins in [5,10]^3 shooting away must miss
    for _ in range(1000):
        o = ((rng.rand(3) + 1.0) * 5.0).astype(np.float32)
        assert not orc.ray_aabb(o, (o / np.linalg.norm(o)).astype(np.float32), mn, mx)
    # 5. random blast from (0,0,2) towards -z within the unit square must hit
    for _ in range(1000):
        d = np.array([rng.rand(), rng.rand(), -1.0], np.float32)
        d /= np.linalg.norm(d)
        assert orc.ray_aabb(np.array([0, 0, 2], np.float32), d, mn, mx)
