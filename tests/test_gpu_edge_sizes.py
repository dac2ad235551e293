"""Ragged and tiny sizes of the joint path against the oracle: fewer observations than a tile, process
sizes straddling the 64-row tile and 512-column panel boundaries, one-point processes, no prediction
points, prediction sets around the 256-point threshold of the library's own Hilbert ordering -- in the
caller's site order and in the library's."""
import numpy as np
import pytest

from oracle import cokrige_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("site_order", [0, 1])
@pytest.mark.parametrize("n0,n1,m", [(5, 3, 4), (1, 1, 1), (63, 65, 300), (64, 64, 257), (511, 1, 10), (700, 900, 0),
                                      (1, 600, 1000), (513, 511, 255)])
def test_ragged_sizes(n0, n1, m, site_order):
    from sif_xco2_cokriging_amd import native, synth
    pv = np.array(synth.SET_A, dtype=float)
    rng = np.random.default_rng(1000 * n0 + n1)
    c0 = np.column_stack([rng.uniform(25, 50, n0), rng.uniform(-120, -70, n0)])
    c1 = np.column_stack([rng.uniform(25, 50, n1), rng.uniform(-120, -70, n1)])
    c1[: min(n0, n1) // 2] = c0[: min(n0, n1) // 2]          # co-located sites: cross-covariance at h == 0
    z0, z1 = rng.standard_normal(n0), rng.standard_normal(n1)
    pc = np.column_stack([rng.uniform(25, 50, m), rng.uniform(-120, -70, m)])
    h = native.Handle(0)
    try:
        h.set_model(2, pv[0:2], pv[2:5], pv[5:8], pv[8:10], pv[10])
        h.set_metric(0)
        h.set_option("site_order", site_order)
        h.set_data(0, c0, z0)
        h.set_data(1, c1, z1)
        h.assemble_joint()
        assert h.factor() == 0
        p = orc.Params.from_flat(pv)
        for i in (0, 1):
            pred, err = h.predict(i, pc)
            assert pred.shape == (m,) and err.shape == (m,)
            if m == 0:
                continue
            rp, re = orc.joint_predict(p, [c0, c1], [z0, z1], pc, i, 0)
            assert np.max(np.abs(pred - rp)) / np.max(np.abs(rp)) < 1e-9
            assert np.max(np.abs(err ** 2 - re ** 2)) < 1e-10
    finally:
        h.close()
