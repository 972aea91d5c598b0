"""CPU: the AUSE / sparsification helper (HLP:382-438) against curves captured from the reference, and the
row-sharding helper of the tiled evaluation."""
import numpy as np
import pytest
import torch

from cfnerf_amd import evaluate as E


@pytest.mark.parametrize("ut", ["c", "v"])
@pytest.mark.parametrize("et", ["rmse", "mae"])
def test_sparsification_plot_vs_reference_golden(golden, ut, et):
    g = golden("g9_sparsification")
    a, b = E.sparsification_plot(torch.tensor(g["var_vec"]), torch.tensor(g["err_vec"]), uncert_type=ut, err_type=et)
    assert a.shape == (100,) and b.shape == (100,)
    np.testing.assert_array_equal(a, g[f"oracle_{ut}_{et}"])        # same torch ops: bit-exact
    np.testing.assert_array_equal(b, g[f"byvar_{ut}_{et}"])


def test_ause_is_small_for_informative_uncertainty_and_row_shards_cover_the_image():
    rng = np.random.default_rng(0)
    err = torch.tensor(rng.uniform(0, 1, 2000) ** 2, dtype=torch.float32)
    assert E.ause(err.clone(), err) < 1e-6                          # perfect uncertainty = the oracle ordering
    assert E.ause(torch.tensor(rng.uniform(0, 1, 2000), dtype=torch.float32), err) > 0.01
    for H, world in ((800, 8), (378, 8), (7, 3), (5, 8)):
        rows = [E.row_shard(H, r, world) for r in range(world)]
        assert rows[0][0] == 0 and rows[-1][1] == H
        assert all(rows[i][1] == rows[i + 1][0] for i in range(world - 1))
        assert max(b - a for a, b in rows) - min(b - a for a, b in rows) <= 1
