"""The CPU checker (oracle/) against the golden vectors generated from the reference.

This is what pins the oracle: every fixture under tests/golden/ was produced by the
reference's own code (tests/golden/make_golden.py).  Integer results are compared
bit for bit; float32 features/duals within the tolerance written at each assert.
"""
import numpy as np
import pytest
import torch

from oracle import features_np, jv, one_gnn_ref


def test_seeded_solver_bit_exact(seeded_cases):
    branches = {}
    for k in range(len(seeded_cases)):
        c = seeded_cases.case(k)
        ret, x, y, st = jv.seeded_raw(c["C"], c["u"], c["v"], c["eps"])
        assert ret == c["ret"], c["label"]
        if ret == 0:
            assert np.array_equal(x, c["x"]), c["label"]
            assert np.array_equal(y, c["y"]), c["label"]
            branches[st["branch"]] = branches.get(st["branch"], 0) + 1
    # every branch of the seeded solve is represented in the fixtures
    assert set(branches) == {1, 2, 3}, branches


def test_seeded_fixtures_cover_projection_and_infeasible(seeded_cases):
    rets = [seeded_cases.case(k)["ret"] for k in range(len(seeded_cases))]
    assert -3 in rets
    fired = 0
    for k in range(len(seeded_cases)):
        c = seeded_cases.case(k)
        fired += jv.seeded_raw(c["C"], c["u"], c["v"], c["eps"])[3]["proj_events"] > 0
    assert fired > 20


def test_seeded_fixtures_cover_micro_arr_firing(seeded_cases):
    """lapjv_seeded.cpp:136-159 fires in >= 10 fixtures (x, y there come from the reference build)."""
    want = seeded_cases.z["arr_fired"]
    ks = [k for k in range(len(seeded_cases)) if want[k] > 0]
    assert len(ks) >= 10 and max(want) >= 2
    for k in ks:
        c = seeded_cases.case(k)
        ret, x, y, st = jv.seeded_raw(c["C"], c["u"], c["v"], c["eps"])
        assert ret == 0 and st["arr_fired"] == want[k] and st["branch"] == 1, c["label"]
        assert np.array_equal(x, c["x"]) and np.array_equal(y, c["y"]), c["label"]


def test_cold_solver_bit_exact(cold_cases):
    for k in range(len(cold_cases)):
        c = cold_cases.case(k)
        ret, x, y, _ = jv.dense_raw(c["C"])
        assert ret == c["ret"] == 0
        assert np.array_equal(x, c["x"]), c["label"]
        assert np.array_equal(y, c["y"]), c["label"]


def test_demo_cases_of_the_reference(seeded_cases):
    """LAP/test_seeded.py and LAP/demo_seeded.py print these; SURVEY 8(c) records them."""
    c = seeded_cases.case(seeded_cases.labels.index("demo3x3/zeros"))
    x, y, cost = jv.lapjv_seeded(c["C"], c["u"], c["v"])
    assert cost == 5.0 and list(x) == [1, 0, 2]
    c = seeded_cases.case(seeded_cases.labels.index("demo4x4/feasible"))
    x, y, cost = jv.lapjv_seeded(c["C"], c["u"], c["v"])
    assert cost == 8.0 and list(x) == [1, 2, 3, 0] and list(y) == [3, 0, 1, 2]


def test_row_features_match_reference(features_cases):
    z = features_cases
    assert tuple(z["empty_shape"]) == features_np.compute_row_features(np.zeros((0, 0))).shape
    assert int(z["row_feature_dim"]) == features_np.ROW_FEATURE_DIM == 21
    for key in [str(s) for s in z["labels"]]:
        got = features_np.compute_row_features(z[f"C__{key}"])
        want = z[f"feat__{key}"]
        assert got.dtype == np.float32 and got.shape == want.shape
        # float32 outputs of fp64 statistics: identical up to 1 ulp of summation order
        np.testing.assert_allclose(got, want, rtol=2e-7, atol=1e-12, err_msg=key)
        # counting features are exact
        assert np.array_equal(got[:, 11], want[:, 11]) and np.array_equal(got[:, 12], want[:, 12])


def test_dual_utilities_match_reference(features_cases):
    z = features_cases
    for key in [str(s) for s in z["labels"]]:
        C, u0, v0 = z[f"C__{key}"], z[f"u0__{key}"], z[f"v0__{key}"]
        pu, pv = features_np.project_feasible(C, u0, v0)
        assert np.array_equal(pu, z[f"proj_u__{key}"]) and np.array_equal(pv, z[f"proj_v__{key}"])
        if f"red_shift__{key}" in z.files:
            assert np.array_equal(features_np.reduce_costs(C, u0, v0, True), z[f"red_shift__{key}"])
            assert np.array_equal(features_np.reduce_costs(C, pu, pv, False), z[f"red_noshift__{key}"])


@pytest.mark.parametrize("tag", ["h64l2", "h192l4"])
def test_onegnn_forward_matches_reference(onegnn_cases, tag):
    z = onegnn_cases
    sd = {k.split("__", 2)[2]: torch.from_numpy(z[k]) for k in z.files if k.startswith(f"sd__{tag}__")}
    keys = [k[len("u__"):] for k in z.files if k.startswith(f"u__{tag}__") and not k.endswith("batch")]
    assert keys
    for key in keys:
        C = z[f"C__{key}"]
        u, v = one_gnn_ref.predict(sd, C)
        # same torch CPU kernels, same op order: tolerance is the north-star 1e-5
        np.testing.assert_allclose(u, z[f"u__{key}"], rtol=0, atol=1e-5, err_msg=key)
        np.testing.assert_allclose(v, z[f"v__{key}"], rtol=0, atol=1e-5, err_msg=key)
        feat = torch.from_numpy(features_np.compute_row_features(C)).float().unsqueeze(0)
        mask = torch.ones((1, C.shape[0]), dtype=torch.bool)
        u_nc = one_gnn_ref.forward(sd, feat, None, mask).squeeze(0).numpy()
        np.testing.assert_allclose(u_nc, z[f"u_nocost__{key}"], rtol=0, atol=1e-5, err_msg=key)
    if f"u__{tag}__batch" in z.files:
        Cb = z[f"C__{tag}__batch"]
        feat = torch.from_numpy(np.stack([features_np.compute_row_features(c) for c in Cb])).float()
        ub = one_gnn_ref.forward(sd, feat, torch.from_numpy(Cb).float(),
                                 torch.from_numpy(z[f"mask__{tag}__batch"])).numpy()
        np.testing.assert_allclose(ub, z[f"u__{tag}__batch"], rtol=0, atol=1e-5)
