"""GPU parity tests (-m gpu): the HIP path, called through the C ABI / the reference-shaped
Python API, against (a) the golden vectors generated from the reference and (b) the CPU oracle
on the same seeded inputs.  Integer results bit-exact; float32 model outputs within the
north-star tolerance 1e-5."""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch


def _seeded_ret(fn, C, u, v, eps):
    try:
        x, y, cost = fn(C, u, v, eps)
        return 0, x, y, cost
    except ValueError as e:
        assert "Infeasible seed potentials" in str(e)
        return -3, None, None, None


# --------------------------------------------------------------------------- golden vectors
def test_seeded_golden_bit_exact(seeded_cases):
    import lap
    n_ok = n_bad3 = 0
    for k in range(len(seeded_cases)):
        c = seeded_cases.case(k)
        ret, x, y, cost = _seeded_ret(lap.lapjv_seeded, c["C"], c["u"], c["v"], c["eps"])
        assert ret == c["ret"], c["label"]
        if ret == 0:
            assert x.dtype == np.int64 and y.dtype == np.int64
            assert np.array_equal(x, c["x"]), c["label"]
            assert np.array_equal(y, c["y"]), c["label"]
            assert cost == float(np.sum(c["C"][np.arange(c["n"]), c["x"]]))
            n_ok += 1
        else:
            n_bad3 += 1
    assert n_ok > 200 and n_bad3 >= 5


def test_cold_golden_bit_exact(cold_cases):
    import lap
    for k in range(len(cold_cases)):
        c = cold_cases.case(k)
        opt, x, y = lap.lapjv(c["C"])
        assert x.dtype == np.int32 and y.dtype == np.int32
        assert np.array_equal(x, c["x"]), c["label"]
        assert np.array_equal(y, c["y"]), c["label"]
        assert opt == c["C"][np.arange(c["n"]), c["x"]].sum()


def test_reference_known_answers():
    """The exact-answer cases of LAP/lap/tests/test_lapjv.py:60-129 (test_square), as data."""
    import lap
    from test_host_logic import KNOWN_SQUARE
    for cost, (opt, ex, ey) in KNOWN_SQUARE:
        ret = lap.lapjv(cost)
        assert ret[0] == opt
        assert list(ret[1]) == ex and list(ret[2]) == ey
        assert cost[range(cost.shape[0]), ret[1]].sum() == ret[0]


def test_reference_known_answer_with_inf_entries():
    """LAP/lap/tests/test_lapjv.py:132-148 (test_sparse_square): inf entries go through the cold
    ARR path's inf/NaN handling on the device."""
    import lap
    from test_host_logic import KNOWN_INF
    cost, (opt, ex, ey) = KNOWN_INF
    ret = lap.lapjv(cost)
    assert ret[0] == opt
    assert list(ret[1]) == ex and list(ret[2]) == ey


def test_micro_arr_firing_cases(seeded_cases, torch_cuda):
    """Instances on which the micro-ARR step fires (lapjv_seeded.cpp:136-159; found on CPU with
    the oracle's counter, confirmed against the reference build, frozen in make_golden.py):
    assignments bit-exact vs the reference AND the kernel reports the same number of firings."""
    torch = torch_cuda
    from gnn import OneGNN, WarmStartPipeline
    fired = seeded_cases.z["arr_fired"]
    ks = [k for k in range(len(seeded_cases)) if fired[k] > 0]
    assert len(ks) >= 10
    pipe = WarmStartPipeline(OneGNN(21), "cuda:0")
    for k in ks:
        c = seeded_cases.case(k)
        x, y, ret, stats = pipe.seeded_batch(torch.from_numpy(c["C"][None]).cuda(),
                                             torch.from_numpy(c["u"][None]).cuda(),
                                             torch.from_numpy(c["v"][None]).cuda(), c["eps"])
        torch.cuda.synchronize()
        assert int(ret[0]) == c["ret"] == 0, c["label"]
        assert int(stats[0, 3]) == int(fired[k]), (c["label"], int(stats[0, 3]), int(fired[k]))
        assert np.array_equal(x[0].cpu().numpy(), c["x"]), c["label"]
        assert np.array_equal(y[0].cpu().numpy(), c["y"]), c["label"]


def test_compute_row_features_torch_matches_host_entry(features_cases, torch_cuda):
    """gnn/features.py:246-351's device entry: CUDA tensors in fp64 and fp32 give
    exactly what compute_row_features gives on the same (widened) values."""
    torch = torch_cuda
    from gnn import compute_row_features, compute_row_features_torch
    z = features_cases
    for key in ("uniform_n64", "sparse_n200", "tie_n64", "metric_n9"):
        C = z[f"C__{key}"]
        want = compute_row_features(C)
        got64 = compute_row_features_torch(torch.from_numpy(C).cuda())
        assert got64.is_cuda and got64.dtype == torch.float32
        assert np.array_equal(got64.cpu().numpy(), want), key
        C32 = C.astype(np.float32)
        got32 = compute_row_features_torch(torch.from_numpy(C32).cuda())
        assert np.array_equal(got32.cpu().numpy(), compute_row_features(C32.astype(np.float64))), key
    assert tuple(compute_row_features_torch(torch.zeros((0, 0)).cuda()).shape) == (0, 0)


def test_row_feature_order_statistics_are_exact_on_adversarial_rows():
    """The feature kernel selects order statistics (16 smallest, median, MAD) instead of sorting
    (gnn/features.py:190,199 sort).  Rows built to stress the selection: heavy ties, one huge
    outlier (every other value in one bucket), all-equal rows, geometric spacing (radix narrowing),
    -0.0/+0.0, odd and tiny n.  Exact against NumPy's sort-based statistics."""
    from gnn import compute_row_features
    from oracle import features_np
    rs = np.random.RandomState(11)

    def make_row(kind, n):
        if kind == 0:
            return rs.uniform(0, 1, n)
        if kind == 1:
            return np.round(rs.uniform(0, 1, n) * 4) / 4                          # 5 distinct values
        if kind == 2:
            return np.full(n, 0.375)                                              # all equal
        if kind == 3:
            return np.where(rs.uniform(size=n) < 0.7, 1e6, rs.uniform(0, 1, n))   # sparse-family shape
        if kind == 4:
            r = rs.uniform(0, 1e-3, n)                                            # one huge outlier
            r[rs.randint(n)] = 1e9
            return r
        if kind == 5:
            return 2.0 ** -rs.randint(0, 900, n).astype(np.float64)               # geometric spacing
        if kind == 6:
            return np.where(rs.uniform(size=n) < 0.5, -0.0, 0.0)                  # signed zeros
        if kind == 7:
            return -rs.uniform(0, 1, n) * 1e3                                     # negative values
        return np.sort(rs.uniform(0, 1, n))[::-1].copy()                          # descending

    for n in (1, 2, 3, 15, 16, 17, 63, 64, 65, 200, 257, 1000, 2048):
        C = np.ascontiguousarray(np.stack([make_row(i % 9, n) for i in range(n)]))
        got, topk = compute_row_features(C, return_topk=True)
        want = features_np.row_statistics(C).astype(np.float32)
        for col in (0, 1, 4, 6, 11, 12):  # min, max, MAD, gap, counting features
            assert np.array_equal(got[:, col], want[:, col]), (n, col, np.flatnonzero(got[:, col] != want[:, col])[:5])
        srt = np.sort(C, axis=1)[:, :16].astype(np.float32)
        assert np.array_equal(topk[:, :min(n, 16)], srt), n
        assert np.all(np.isposinf(topk[:, min(n, 16):])), n
        np.testing.assert_allclose(got[:, :13], want, rtol=2e-6, atol=1e-6, err_msg=str(n))


def test_row_features_golden(features_cases):
    from gnn import compute_row_features
    z = features_cases
    assert compute_row_features(np.zeros((0, 0))).shape == tuple(z["empty_shape"])
    for key in [str(s) for s in z["labels"]]:
        got = compute_row_features(z[f"C__{key}"])
        want = z[f"feat__{key}"]
        assert got.dtype == np.float32 and got.shape == want.shape
        # fp64 statistics rounded to float32: differences come from summation order only
        np.testing.assert_allclose(got, want, rtol=3e-6, atol=1e-9, err_msg=key)
        for col in (0, 1, 4, 6, 11, 12):  # min, max, MAD, gap, and the two counting features: exact
            assert np.array_equal(got[:, col], want[:, col]), (key, col)
        assert np.array_equal(got[:, 13:], want[:, 13:]), key  # positional encodings


def test_dual_utilities_golden(features_cases):
    from solvers import check_dual_feasible, project_feasible, reduce_costs
    z = features_cases
    for key in [str(s) for s in z["labels"]]:
        C, u0, v0 = z[f"C__{key}"], z[f"u0__{key}"], z[f"v0__{key}"]
        pu, pv = project_feasible(C, u0, v0)
        assert np.array_equal(pu, z[f"proj_u__{key}"]), key
        assert np.array_equal(pv, z[f"proj_v__{key}"]), key
        assert check_dual_feasible(C, pu, pv, tol=1e-8)
        if f"red_shift__{key}" in z.files:
            assert np.array_equal(reduce_costs(C, u0, v0, True), z[f"red_shift__{key}"]), key
            assert np.array_equal(reduce_costs(C, pu, pv, False), z[f"red_noshift__{key}"]), key
    with pytest.raises(AssertionError):
        check_dual_feasible(np.zeros((4, 4)), np.ones(4), np.ones(4))


def test_classical_seeds(features_cases):
    """seed_row_col_minima is the reference's sweeps one for one: bit-identical to the NumPy
    restatement; seed_noisy_optimal must return feasible duals whose seeded solve is optimal."""
    import lap
    from oracle import features_np
    from solvers import check_dual_feasible, seed_noisy_optimal, seed_row_col_minima
    z = features_cases
    for key in [str(s) for s in z["labels"]]:
        C = z[f"C__{key}"]
        u, v = seed_row_col_minima(C)
        uo, vo = features_np.seed_row_col_minima(C)
        assert np.array_equal(u, uo) and np.array_equal(v, vo), key
        assert check_dual_feasible(C, u, v, tol=1e-9)
    C = z["C__uniform_n64"]
    u, v = seed_noisy_optimal(C, noise_std=0.05, rng=np.random.default_rng(3))
    assert check_dual_feasible(C, u, v, tol=1e-9)
    x, y, cost = lap.lapjv_seeded(C, u, v)
    assert abs(cost - lap.lapjv(C)[0]) < 1e-9


@pytest.mark.parametrize("tag,H,L", [("h64l2", 64, 2), ("h192l4", 192, 4)])
def test_onegnn_device_forward_golden(onegnn_cases, torch_cuda, tag, H, L):
    torch = torch_cuda
    from gnn import GNNPredictor, OneGNN
    z = onegnn_cases
    sd = {k.split("__", 2)[2]: torch.from_numpy(z[k]) for k in z.files if k.startswith(f"sd__{tag}__")}
    model = OneGNN(21, hidden=H, layers=L)
    model.load_state_dict(sd)  # the reference's state-dict keys load unchanged
    pred = GNNPredictor(model=model, device="cuda:0")
    keys = [k[len("u__"):] for k in z.files if k.startswith(f"u__{tag}__") and not k.endswith("batch")]
    assert keys
    for key in keys:
        C = z[f"C__{key}"]
        u, v = pred.predict(C)
        assert u.dtype == np.float64 and v.dtype == np.float64
        # north-star tolerance for u/v against the PyTorch-CPU forward
        assert np.abs(u - z[f"u__{key}"]).max() <= 1e-5, (key, np.abs(u - z[f"u__{key}"]).max())
        assert np.abs(v - z[f"v__{key}"]).max() <= 1e-5, (key, np.abs(v - z[f"v__{key}"]).max())
    # drop-in call convention of the harness: model(row_feat, cost=..., mask=...)
    key = keys[0]
    C = z[f"C__{key}"]
    from gnn import compute_row_features
    n = C.shape[0]
    row = torch.from_numpy(compute_row_features(C)).float().unsqueeze(0).cuda()
    cost = torch.from_numpy(C).float().unsqueeze(0).cuda()
    mask = torch.ones((1, n), dtype=torch.bool, device="cuda")
    model = model.cuda().eval()
    with torch.inference_mode():
        u2 = model(row, cost=cost, mask=mask)["u"].squeeze(0).cpu().numpy()
        u3 = model(row, mask=mask)["u"].squeeze(0).cpu().numpy()
    assert np.abs(u2 - z[f"u__{key}"]).max() <= 1e-5
    assert np.abs(u3 - z[f"u_nocost__{key}"]).max() <= 1e-5
    if f"u__{tag}__batch" in z.files:
        Cb = z[f"C__{tag}__batch"]
        feats = np.stack([compute_row_features(c) for c in Cb])
        with torch.inference_mode():
            ub = model(torch.from_numpy(feats).cuda(), cost=torch.from_numpy(Cb).float().cuda(),
                       mask=torch.from_numpy(z[f"mask__{tag}__batch"]).cuda())["u"].cpu().numpy()
        assert np.abs(ub - z[f"u__{tag}__batch"]).max() <= 1e-5


# --------------------------------------------------------------------------- oracle, larger sizes
def test_native_parity_driver_up_to_256():
    """Thousands of seeded cases across families / seed kinds / every branch, C ABI vs oracle."""
    exe = ROOT / "tests" / "native" / "_build" / "parity_driver"
    assert exe.exists(), "build it with __graft_entry__.build()"
    proc = subprocess.run([str(exe), "256", "2"], capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stdout[-3000:]
    assert "bad=0" in proc.stdout.splitlines()[-1]


def test_batched_k2_counters_match_oracle(torch_cuda):
    """K2-shaped batch (n=512, uniform seeds 42+i): assignments AND the kernel's control-flow
    counters (paths, minima collections, relax steps/elements, ARR iterations) equal the oracle's."""
    torch = torch_cuda
    from gnn import OneGNN, WarmStartPipeline
    from oracle import jv
    B, n = 8, 512
    Cs = np.stack([np.random.RandomState(42 + i).uniform(0, 1, (n, n)) for i in range(B)])
    # even instances: row-min seeds + fp64 min-trick (SSP branch); odd: zero seeds (fallback branch)
    us = np.stack([C.min(1) if i % 2 == 0 else np.zeros(n) for i, C in enumerate(Cs)])
    vs = np.stack([(C - u[:, None]).min(0) if i % 2 == 0 else np.zeros(n) for i, (C, u) in enumerate(zip(Cs, us))])
    pipe = WarmStartPipeline(OneGNN(21), "cuda:0")
    x, y, ret, stats = pipe.seeded_batch(torch.from_numpy(Cs).cuda(), torch.from_numpy(us).cuda(),
                                         torch.from_numpy(vs).cuda())
    torch.cuda.synchronize()
    x, y, ret, stats = x.cpu().numpy(), y.cpu().numpy(), ret.cpu().numpy(), stats.cpu().numpy()
    names = ["branch", "tight_edges", "free_rows", "arr_fired", "paths", "finds", "scan_steps", "scan_elems",
             "init_elems", "colred_elems", "transfer_rows", "arr_iters"]
    seen = set()
    for b in range(B):
        r, xo, yo, st = jv.seeded_raw(Cs[b], us[b], vs[b])
        assert r == ret[b] == 0
        assert np.array_equal(xo, x[b]) and np.array_equal(yo, y[b])
        for q, name in enumerate(names):
            assert stats[b, q] == st[name], (b, name, stats[b, q], st[name])
        seen.add(st["branch"])
    assert seen == {1, 3}


def test_k2_config_oracle_duals(torch_cuda):
    """BASELINE configs[1] (K2): batch=64, n=512, uniform RandomState(42+i); u = optimal duals
    derived from the cold JV, v = fp64 min-trick; HIP column minima + row features + seeded JV.
    Every instance bit-exact against the oracle, duals feasible and tight on the optimum."""
    torch = torch_cuda
    from gnn import OneGNN, WarmStartPipeline
    from gnn.features import min_trick_device, row_features_device
    from oracle import features_np, jv
    B, n = 64, 512
    Cs = np.stack([np.random.RandomState(42 + i).uniform(0, 1, (n, n)) for i in range(B)])
    pipe = WarmStartPipeline(OneGNN(21), "cuda:0")
    C = torch.from_numpy(Cs).cuda()
    xc, u, v_opt, ret = pipe.optimal_duals_batch(C)
    v = min_trick_device(C, u)
    feat, top16 = row_features_device(C)
    x, y, ret2, stats = pipe.seeded_batch(C, u, v)
    torch.cuda.synchronize()
    assert int(ret.abs().sum()) == 0 and int(ret2.abs().sum()) == 0
    un, vn, von = u.cpu().numpy(), v.cpu().numpy(), v_opt.cpu().numpy()
    xs, ys, xcn = x.cpu().numpy(), y.cpu().numpy(), xc.cpu().numpy()
    fn = feat.cpu().numpy()
    for b in range(B):
        red = Cs[b] - un[b][:, None] - von[b][None, :]
        assert red.min() > -1e-9 and np.abs(red[np.arange(n), xcn[b]]).max() < 1e-9  # optimal dual pair
        assert np.array_equal(vn[b], features_np.min_trick(Cs[b], un[b]))
        r, xo, yo, _ = jv.seeded_raw(Cs[b], un[b], vn[b])
        assert r == 0 and np.array_equal(xo, xs[b]) and np.array_equal(yo, ys[b]), b
        assert abs(Cs[b][np.arange(n), xs[b]].sum() - Cs[b][np.arange(n), xcn[b]].sum()) < 1e-9
        if b < 4:
            np.testing.assert_allclose(fn[b], features_np.compute_row_features(Cs[b]), rtol=3e-6, atol=1e-9)


@pytest.mark.parametrize("n", [2048, 4096, 5000])
def test_full_size_single_instances(n):
    """BASELINE sizes through the drop-in API: LDS-resident state (2048, 4096) and the
    global-workspace variant (5000)."""
    import lap
    from oracle import jv
    C = np.random.RandomState(42).uniform(0, 1, (n, n))
    u = C.min(1)
    v = (C - u[:, None]).min(0)
    x, y, cost = lap.lapjv_seeded(C, u, v)
    assert sorted(x.tolist()) == list(range(n)) and np.array_equal(y[x], np.arange(n))
    r, xo, yo, st = jv.seeded_raw(C, u, v)
    assert r == 0 and st["branch"] == 1
    assert np.array_equal(x, xo) and np.array_equal(y, yo)
    if n == 2048:
        _, xc, yc = lap.lapjv(C)
        rc, xco, yco, _ = jv.dense_raw(C)
        assert np.array_equal(xc, xco) and np.array_equal(yc, yco)
        assert abs(C[np.arange(n), xc].sum() - cost) < 1e-9  # seeded and cold agree on the optimum


@pytest.mark.parametrize("n", [8192, 16384])
def test_large_n_pipeline_exact(torch_cuda, n):
    """K5-sized instance (n=16384) and n=8192: solver state no longer fits LDS (global-workspace
    variant), features run with the 128-KiB LDS row.  Whole pipeline, one instance, bit-exact
    assignment and identical relax-step count against the oracle fed the GPU's own (u, v)."""
    torch = torch_cuda
    from gnn import OneGNN, WarmStartPipeline
    from oracle import jv
    C = np.random.RandomState(42).uniform(0, 1, (n, n))
    torch.manual_seed(0)
    pipe = WarmStartPipeline(OneGNN(21, hidden=64, layers=2).eval(), "cuda:0")
    out = pipe.solve_batch(torch.from_numpy(C).cuda().unsqueeze(0))
    torch.cuda.synchronize()
    assert int(out["ret"][0]) == 0
    x = out["x"][0].cpu().numpy()
    assert np.array_equal(np.sort(x), np.arange(n))
    u = out["u"][0].cpu().numpy().astype(np.float64)
    v = out["v"][0].cpu().numpy()
    r, xo, yo, so = jv.seeded_raw(C, u, v)
    assert r == 0 and np.array_equal(xo, x) and np.array_equal(yo, out["y"][0].cpu().numpy())
    st = out["stats"][0].cpu().numpy()
    assert st[6] == so["scan_steps"] and st[4] == so["paths"] and st[7] == so["scan_elems"]


def test_cooperative_kernel_forced_on_small_sizes_native_sweep():
    """The cooperative shortest-path kernel (coop_ssp.hip: one instance over several single-wave
    workgroups, normally n >= 4428) forced on for EVERY size (LAPWARM_COOP_MIN_N=1): the native sweep
    to n = 512 -- 10 families x 7 seed kinds, cold solves included -- must stay bit-exact.  Tie-heavy
    families make the kernel stop at a path boundary and hand the rest to jv_instance_kernel's
    resume phase, so the sweep covers the hand-over in both directions; sizes 1 .. 63 run with a
    single, partly empty member."""
    exe = ROOT / "tests" / "native" / "_build" / "parity_driver"
    env = dict(os.environ, LAPWARM_COOP_MIN_N="1")
    proc = subprocess.run([str(exe), "512", "1"], capture_output=True, text=True, timeout=900, env=env)
    assert proc.returncode == 0, proc.stdout[-3000:]
    assert "bad=0" in proc.stdout.splitlines()[-1]


@pytest.mark.parametrize("n", [4611, 9216])
def test_cooperative_kernel_member_counts(torch_cuda, n):
    """n = 4611: an odd n above the threshold (rows only 8-byte aligned: the 16-byte row prefetch is off,
    the last member is partly empty; 19 members of 256 positions).  n = 9216: 18 members of 512 positions
    -- between 17 and 26 members a collection round's records no longer fit one 64-lane register while
    the poll still takes two loads per lane (the case a round-3 build got wrong: error -137 for every
    size from 8193 to 13312).  Bit-exact with equal counters."""
    torch = torch_cuda
    from gnn import OneGNN, WarmStartPipeline
    from gnn.features import min_trick_device
    from oracle import jv
    B = 2
    Cs = np.stack([np.random.RandomState(7 + i).uniform(0, 1, (n, n)) for i in range(B)])
    pipe = WarmStartPipeline(OneGNN(21, hidden=64, layers=2).eval(), "cuda:0")
    C = torch.from_numpy(Cs).cuda()
    u = C.min(dim=2).values.contiguous()
    v = min_trick_device(C, u)
    x, y, ret, stats = pipe.seeded_batch(C, u, v)
    torch.cuda.synchronize()
    st = stats.cpu().numpy()
    assert (st[:, 15] >= 0).all()
    un, vn = u.cpu().numpy(), v.cpu().numpy()
    for b in range(B):
        r, xo, yo, so = jv.seeded_raw(Cs[b], un[b], vn[b])
        assert r == int(ret[b]) == 0 and np.array_equal(xo, x[b].cpu().numpy()) and np.array_equal(yo, y[b].cpu().numpy())
        assert st[b, 6] == so["scan_steps"] and st[b, 5] == so["finds"] and st[b, 4] == so["paths"]


_FORCED_COOP = r"""
import sys, numpy as np, torch
sys.path[:0] = [%r, %r]
from gnn import OneGNN, WarmStartPipeline
from gnn.features import min_trick_device
from oracle import jv
from lap import _hip
n = int(sys.argv[1])
assert _hip.load().lapwarm_coop_members(n) == int(sys.argv[2]), _hip.load().lapwarm_coop_members(n)
Cs = np.stack([np.random.RandomState(11 + i).uniform(0, 1, (n, n)) for i in range(2)])
pipe = WarmStartPipeline(OneGNN(21, hidden=64, layers=2).eval(), "cuda:0")
C = torch.from_numpy(Cs).cuda()
u = C.min(dim=2).values.contiguous()
v = min_trick_device(C, u)
x, y, ret, stats = pipe.seeded_batch(C, u, v)
torch.cuda.synchronize()
st = stats.cpu().numpy()
assert (st[:, 15] >= 0).all()
for b in range(2):
    r, xo, yo, so = jv.seeded_raw(Cs[b], u[b].cpu().numpy(), v[b].cpu().numpy())
    assert r == int(ret[b]) == 0 and np.array_equal(xo, x[b].cpu().numpy()) and np.array_equal(yo, y[b].cpu().numpy())
    assert st[b, 6] == so["scan_steps"] and st[b, 5] == so["finds"] and st[b, 4] == so["paths"]
print("forced-coop ok")
"""


@pytest.mark.parametrize("n,members", [(1024, 8), (3072, 12), (4096, 16)])
def test_cooperative_kernel_forced_member_counts(n, members):
    """Member counts the default plan never produces above its threshold -- 8 members of 128 positions
    (one granule load per lane), 12 and 16 members of 256 (two loads, records within one register) --
    forced on below the threshold in a child process (the switch is read once per process)."""
    env = dict(os.environ, LAPWARM_COOP_MIN_N="1")
    script = _FORCED_COOP % (str(ROOT), str(ROOT / "gnn-accelerated-lap-warm-start-pipeline_amd"))
    proc = subprocess.run([sys.executable, "-c", script, str(n), str(members)], capture_output=True, text=True,
                          timeout=600, env=env, cwd=str(ROOT))
    assert proc.returncode == 0 and "forced-coop ok" in proc.stdout, (proc.stdout[-2000:], proc.stderr[-3000:])


@pytest.mark.parametrize("fams", [("uniform", "sparse"), ("tie", "clustered")])
def test_cooperative_kernel_at_its_threshold_size(torch_cuda, fams):
    """n = 4608 (the first sizes whose solver state leaves LDS run the cooperative kernel, 18 members
    of 256 positions): continuous costs stay in it for every path; sparse / tie / clustered costs
    produce tie collections, where it stops and jv_instance_kernel resumes.  Assignments bit-exact
    and the path / collection / relax-step / element counters equal to the oracle's either way --
    the counters of the two kernels add up to the serial algorithm's."""
    torch = torch_cuda
    from gnn import OneGNN, WarmStartPipeline
    from gnn.features import min_trick_device
    from oracle import jv
    from solvers.generators import mixed_batch
    B, n = 4, 4608
    Cs, names = mixed_batch(B, n, families=fams, seed=99)
    pipe = WarmStartPipeline(OneGNN(21, hidden=64, layers=2).eval(), "cuda:0")
    C = torch.from_numpy(Cs).cuda()
    u = C.min(dim=2).values.contiguous()
    v = min_trick_device(C, u)
    x, y, ret, stats = pipe.seeded_batch(C, u, v)
    torch.cuda.synchronize()
    st = stats.cpu().numpy()
    assert (st[:, 15] >= 0).all(), "the cooperative kernel did not run at n = 4608"
    coop_paths = st[:, 15] & 0xffffffff
    un, vn = u.cpu().numpy(), v.cpu().numpy()
    for b in range(B):
        r, xo, yo, so = jv.seeded_raw(Cs[b], un[b], vn[b])
        assert r == int(ret[b]) == 0, (names[b], r, int(ret[b]), st[b, 12])
        assert np.array_equal(xo, x[b].cpu().numpy()) and np.array_equal(yo, y[b].cpu().numpy()), names[b]
        for q, k in ((4, "paths"), (5, "finds"), (6, "scan_steps"), (7, "scan_elems"), (8, "init_elems")):
            assert st[b, q] == so[k], (names[b], k, st[b, q], so[k])
        if names[b] == "uniform" and so["branch"] == 1:
            assert coop_paths[b] == so["paths"], (names[b], coop_paths[b], so["paths"])  # never left the kernel


def test_k5_dense_stages_n16384(torch_cuda):
    """BASELINE config K5 (n=16384, top-16 refinement path): the dense stages on one instance.
    Row features (128-KiB LDS row) against the NumPy oracle on sampled rows, u against the
    PyTorch-CPU forward (fp32 cost tensor, top-k over 16384 columns) within the north-star 1e-5,
    v-hat exact."""
    torch = torch_cuda
    from gnn import OneGNN
    from gnn.features import min_trick_device, row_features_device
    from oracle import features_np, one_gnn_ref
    n = 16384
    C = np.random.RandomState(42).uniform(0, 1, (n, n))
    Cd = torch.from_numpy(C).cuda().unsqueeze(0)
    feat, topk = row_features_device(Cd)
    torch.cuda.synchronize()
    got = feat[0].cpu().numpy()
    rows = np.random.RandomState(7).choice(n, size=48, replace=False)
    want = features_np.row_statistics(C[rows], col_min=C.min(axis=0)).astype(np.float32)
    np.testing.assert_allclose(got[rows, :13], want, rtol=3e-6, atol=1e-9)
    for col in (0, 1, 4, 6, 11, 12):
        assert np.array_equal(got[rows, col], want[:, col]), col
    assert np.array_equal(got[:, 13:], features_np.positional_encodings(n))
    # the 16 smallest entries per row that the refinement consumes
    assert np.array_equal(topk[0].cpu().numpy()[rows], features_np.topk_smallest(C[rows], 16).astype(np.float32))

    sd = one_gnn_ref.init_state_dict(hidden=64, layers=2, seed=0)
    model = OneGNN(21, hidden=64, layers=2)
    model.load_state_dict(sd)
    model = model.cuda().eval()
    mask = torch.ones((1, n), dtype=torch.bool, device="cuda")
    with torch.inference_mode():
        u_gpu = model(feat, mask=mask, topk_values=topk)["u"][0]
        u_ref = one_gnn_ref.forward(sd, feat.cpu(), torch.from_numpy(C).float().unsqueeze(0), mask.cpu())[0]
    assert np.abs(u_gpu.cpu().numpy() - u_ref.numpy()).max() <= 1e-5
    v = min_trick_device(Cd, u_gpu.unsqueeze(0))[0].cpu().numpy()
    u64 = u_gpu.cpu().numpy().astype(np.float64)
    want_v = np.full(n, np.inf)
    for lo in range(0, n, 2048):  # (row blocks: the n x n temporary would be 2 GiB)
        want_v = np.minimum(want_v, (C[lo:lo + 2048] - u64[lo:lo + 2048, None]).min(axis=0))
    assert np.array_equal(v, want_v)


def test_pipeline_mixed_families_end_to_end(torch_cuda):
    """K3-shaped (reduced batch): features + OneGNN + min-trick + seeded solve, all families.
    u/v within 1e-5 of the CPU forward; assignments bit-exact given the GPU's own (u, v)."""
    torch = torch_cuda
    from gnn import OneGNN, WarmStartPipeline
    from oracle import jv, one_gnn_ref
    from solvers.generators import mixed_batch
    B, n = 4, 768
    Cs, fams = mixed_batch(B, n, seed=99)
    torch.manual_seed(0)
    model = OneGNN(21, hidden=192, layers=4).eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    out = WarmStartPipeline(model, "cuda:0").solve_batch(torch.from_numpy(Cs).cuda())
    torch.cuda.synchronize()
    u, v = out["u"].cpu().numpy(), out["v"].cpu().numpy()
    x, y, ret = out["x"].cpu().numpy(), out["y"].cpu().numpy(), out["ret"].cpu().numpy()
    for b in range(B):
        ur, vr = one_gnn_ref.predict(sd, Cs[b])
        assert np.abs(u[b] - ur).max() <= 1e-5, (fams[b], np.abs(u[b] - ur).max())
        scale = max(1.0, float(np.abs(vr).max()))
        assert np.abs(v[b] - vr).max() <= 1e-5 * scale, (fams[b], np.abs(v[b] - vr).max())
        r, xo, yo, _ = jv.seeded_raw(Cs[b], u[b].astype(np.float64), v[b])
        assert r == ret[b], fams[b]
        if r == 0:
            assert np.array_equal(xo, x[b]) and np.array_equal(yo, y[b]), fams[b]


def test_k3_config_full_size(torch_cuda):
    """BASELINE configs[2] (K3) at its stated size: batch=32, n=2048, 8 each uniform / sparse /
    metric / clustered, OneGNN H=192 L=4.  u/v within 1e-5 of the CPU forward on one instance per
    family; x/y bit-exact and control-flow counters equal to the oracle's on all 32."""
    torch = torch_cuda
    from gnn import OneGNN, WarmStartPipeline
    from oracle import jv, one_gnn_ref
    from solvers.generators import mixed_batch
    B, n = 32, 2048
    Cs, fams = mixed_batch(B, n, seed=1234)
    assert sorted(set(fams)) == ["clustered", "metric", "sparse", "uniform"]
    torch.manual_seed(0)
    model = OneGNN(21, hidden=192, layers=4).eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    out = WarmStartPipeline(model, "cuda:0").solve_batch(torch.from_numpy(Cs).cuda())
    torch.cuda.synchronize()
    u, v = out["u"].cpu().numpy(), out["v"].cpu().numpy()
    x, y, ret = out["x"].cpu().numpy(), out["y"].cpu().numpy(), out["ret"].cpu().numpy()
    stats = out["stats"].cpu().numpy()
    for f in sorted(set(fams)):
        b = fams.index(f)
        ur, vr = one_gnn_ref.predict(sd, Cs[b])
        assert np.abs(u[b] - ur).max() <= 1e-5, (f, np.abs(u[b] - ur).max())
        scale = max(1.0, float(np.abs(vr).max()))
        assert np.abs(v[b] - vr).max() <= 1e-5 * scale, (f, np.abs(v[b] - vr).max())
    names = ["branch", "tight_edges", "free_rows", "arr_fired", "paths", "finds", "scan_steps", "scan_elems",
             "init_elems", "colred_elems", "transfer_rows", "arr_iters"]
    for b in range(B):
        r, xo, yo, st = jv.seeded_raw(Cs[b], u[b].astype(np.float64), v[b])
        assert r == ret[b] == 0, (b, fams[b], r, ret[b])
        assert np.array_equal(xo, x[b]) and np.array_equal(yo, y[b]), (b, fams[b])
        for q, name in enumerate(names):
            assert stats[b, q] == st[name], (b, fams[b], name, stats[b, q], st[name])


@pytest.mark.parametrize("source", ["k4", "mixed"])
def test_k4_config_slice(torch_cuda, source):
    """BASELINE configs[3] (K4: batch 256 over 8 GPUs, n=4096): one GPU's slice shape at a batch
    the oracle finishes in under a minute -- 8 of the 32 instances, with K4's OWN input
    (uniform RandomState(42 + i), scripts/gnn_large_scale_benchmark.py:243-251, exactly what
    `bench.py --config K4` feeds rank 0) and with two instances per family.  n=4096 runs
    the solver with x / free-row list in global memory (LDS level 1) and 4 positions per thread.
    x/y bit-exact and counters equal to the oracle's, fed the GPU's own (u, v)."""
    torch = torch_cuda
    from gnn import OneGNN, WarmStartPipeline
    from oracle import jv
    from solvers.generators import mixed_batch
    B, n = 8, 4096
    if source == "k4":
        Cs = np.stack([np.random.RandomState(42 + i).uniform(0, 1, (n, n)) for i in range(B)])
        fams = ["uniform"] * B
    else:
        Cs, fams = mixed_batch(B, n, seed=77)
        assert sorted(set(fams)) == ["clustered", "metric", "sparse", "uniform"]
    torch.manual_seed(0)
    model = OneGNN(21, hidden=192, layers=4).eval()
    out = WarmStartPipeline(model, "cuda:0").solve_batch(torch.from_numpy(Cs).cuda())
    torch.cuda.synchronize()
    u, v = out["u"].cpu().numpy(), out["v"].cpu().numpy()
    x, y, ret = out["x"].cpu().numpy(), out["y"].cpu().numpy(), out["ret"].cpu().numpy()
    stats = out["stats"].cpu().numpy()
    names = ["branch", "tight_edges", "free_rows", "arr_fired", "paths", "finds", "scan_steps", "scan_elems"]
    for b in range(B):
        r, xo, yo, st = jv.seeded_raw(Cs[b], u[b].astype(np.float64), v[b])
        assert r == ret[b] == 0, (b, fams[b], r, ret[b])
        assert np.array_equal(xo, x[b]) and np.array_equal(yo, y[b]), (b, fams[b])
        for q, name in enumerate(names):
            assert stats[b, q] == st[name], (b, fams[b], name, stats[b, q], st[name])


def test_two_stream_pipeline_and_graph_capture_match_eager(torch_cuda):
    """(a) The two-stream software pipeline of bench.py (dense sweeps + OneGNN of batch k+1 beside
    the solver of batch k) returns exactly what solve_batch returns; (b) the whole per-batch launch
    chain is capturable in a HIP graph (no allocation, no host sync in the C ABI) and the replayed
    graph reproduces the eager result."""
    torch = torch_cuda
    from gnn import OneGNN, WarmStartPipeline
    from solvers.generators import mixed_batch
    B, n = 8, 512
    Cs, _ = mixed_batch(B, n, seed=5)
    Cs2, _ = mixed_batch(B, n, seed=6)
    torch.manual_seed(0)
    model = OneGNN(21, hidden=64, layers=2).eval()
    pipe = WarmStartPipeline(model, "cuda:0")
    C1, C2 = torch.from_numpy(Cs).cuda(), torch.from_numpy(Cs2).cuda()
    ref1, ref2 = pipe.solve_batch(C1), pipe.solve_batch(C2)
    torch.cuda.synchronize()
    pipe.pipeline_submit(C1)
    o1 = pipe.pipeline_step(C_next=C2)
    o2 = pipe.pipeline_step(C_next=None)
    pipe.pipeline_drain()
    for o, r in ((o1, ref1), (o2, ref2)):
        for key in ("x", "y", "ret", "u", "v"):
            assert torch.equal(o[key], r[key]), key
    # HIP graph: warm up on a side stream (workspace + allocator pools), capture, replay twice
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        pipe.solve_batch(C1)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    static_C = C1.clone()
    with torch.cuda.graph(g):
        out = pipe.solve_batch(static_C)
    ws = pipe._workspace(B, n)[0]
    for rep, (src, ref) in enumerate(((C1, ref1), (C2, ref2), (C1, ref1))):
        static_C.copy_(src)
        # Poison the whole solver workspace before the replay: every word the captured kernels read
        # (seeds, per-instance flags, helper ring, tight bitmaps, global solver state) must be
        # written inside the captured chain itself.  A stale word -- the flake of round 2 -- would
        # now be 0xFF.. instead of "whatever the previous replay left", and names itself below.
        with torch.inference_mode():  # (the workspace was allocated inside solve_batch's inference mode)
            ws.fill_(0xFF)
        g.replay()
        torch.cuda.synchronize()
        st = out["stats"].cpu().numpy()
        assert out["ret"].tolist() == ref["ret"].tolist(), (rep, out["ret"].tolist(), st[:, 12].tolist(), st[:, 0].tolist())
        assert not st[:, 12].any(), (rep, st[:, 12].tolist())
        for key in ("x", "y", "ret", "u", "v"):
            assert torch.equal(out[key], ref[key]), (rep, key)


@pytest.mark.parametrize("n,hint", [(1024, 1024), (1024, 512), (1024, 256), (2048, 1024), (2048, 512)])
def test_batched_solver_across_workgroup_geometries(torch_cuda, n, hint):
    """Same inputs through different workgroup geometries (1, 2, 4 or 8 positions per thread:
    different code paths and timing): assignments and control-flow counters must equal the
    oracle's for every instance, GNN-quality seeds, all four K3 families."""
    torch = torch_cuda
    from gnn import OneGNN, WarmStartPipeline
    from oracle import jv
    from solvers.generators import mixed_batch
    B = 8
    Cs, fams = mixed_batch(B, n, seed=4242 + n)
    torch.manual_seed(1)
    pipe = WarmStartPipeline(OneGNN(21, hidden=64, layers=2).eval(), "cuda:0", threads_hint=hint)
    C = torch.from_numpy(Cs).cuda()
    u, v = pipe.predict_batch(C)
    for rep in range(2):  # twice: no state may leak between launches
        x, y, ret, stats = pipe.seeded_batch(C, u, v)
        torch.cuda.synchronize()
        xs, ys, rets, st = x.cpu().numpy(), y.cpu().numpy(), ret.cpu().numpy(), stats.cpu().numpy()
        un, vn = u.cpu().numpy().astype(np.float64), v.cpu().numpy()
        for b in range(B):
            r, xo, yo, so = jv.seeded_raw(Cs[b], un[b], vn[b])
            assert r == rets[b] == 0, (fams[b], r, rets[b], st[b, 12])
            assert np.array_equal(xo, xs[b]) and np.array_equal(yo, ys[b]), (fams[b], rep)
            for q, name in ((4, "paths"), (5, "finds"), (6, "scan_steps"), (7, "scan_elems"), (11, "arr_iters")):
                assert st[b, q] == so[name], (fams[b], name, st[b, q], so[name])


def test_solver_without_helper_workgroups_is_bit_exact():
    """The helper workgroups only pull rows into the shared L2; the solver must give the same results
    without them (LAPWARM_HELPER=0; the rest of the suite runs with them): native sweep to n = 2048
    in a process of its own."""
    exe = ROOT / "tests" / "native" / "_build" / "parity_driver"
    env = dict(os.environ, LAPWARM_HELPER="0")
    proc = subprocess.run([str(exe), "2048", "1"], capture_output=True, text=True, timeout=900, env=env)
    assert proc.returncode == 0, proc.stdout[-3000:]
    assert "bad=0" in proc.stdout.splitlines()[-1]


def test_run_to_run_determinism_under_concurrent_traffic(torch_cuda):
    """The solver synchronises 16 waves through LDS slots and parity-buffered bitmaps; a race
    would show as run-to-run differences.  Repeat the same batch while another stream streams
    through HBM (different wave arrival times): assignments and counters must not move."""
    torch = torch_cuda
    from gnn import OneGNN, WarmStartPipeline
    from solvers.generators import mixed_batch
    B, n = 16, 1024
    Cs, _ = mixed_batch(B, n, seed=77)
    torch.manual_seed(2)
    pipe = WarmStartPipeline(OneGNN(21, hidden=64, layers=2).eval(), "cuda:0")
    C = torch.from_numpy(Cs).cuda()
    u, v = pipe.predict_batch(C)
    side = torch.cuda.Stream()
    junk = torch.empty((32, 1024, 1024), device="cuda")
    first = None
    for rep in range(60):  # (a path-end race once needed ~100-300 repetitions to show: tools/stress_determinism.py)
        if rep % 2:
            with torch.cuda.stream(side):
                for _ in range(10):
                    junk.mul_(1.0001)
        x, y, ret, st = pipe.seeded_batch(C, u, v)
        torch.cuda.synchronize()
        cur = (x.cpu().numpy(), ret.cpu().numpy(), st[:, :13].cpu().numpy())
        assert (cur[1] == 0).all()
        if first is None:
            first = cur
        else:
            assert np.array_equal(cur[0], first[0]) and np.array_equal(cur[2], first[2]), rep


# --------------------------------------------------------------------------- wrappers / errors
def test_solver_wrappers_and_error_behaviour():
    import lap
    from solvers import LAPSolver, SciPySolver, SeededLAPSolver, WarmStartLAPSolver, time_solver_rigorous
    C = np.random.RandomState(3).uniform(0, 1, (64, 64))
    u = C.min(1)
    v = (C - u[:, None]).min(0)
    rows, cols, cost = SeededLAPSolver().solve(C, u=u, v=v)  # keyword call, as analyze_all_types_pipeline.py:242
    assert rows.dtype == np.int64 and cols.dtype == np.int64 and isinstance(cost, float)
    assert np.array_equal(cols[rows], np.arange(64))  # "rows" = x, "cols" = y
    r2, c2, cost2 = LAPSolver().solve(C)
    assert np.array_equal(r2, np.arange(64)) and abs(cost2 - cost) < 1e-9
    r3, c3, cost3 = WarmStartLAPSolver().solve(C, u, v)
    assert abs(cost3 - cost) < 1e-9 and abs(SciPySolver().solve(C)[2] - cost) < 1e-9
    from oracle import jv
    # device-resident reduce+solve == the oracle's cold JV on the host-formed reduced matrix
    # (reference solvers/warmstart_solver.py:49-63), bit-exact, with and without the shift
    for shift in (True, False):
        for uu, vv in ((u, v), (u + 0.25, v + 0.125)):  # the second pair makes min(C') negative
            Cp = C - uu[:, None] - vv[None, :]
            if shift and Cp.min() < 0:
                Cp = Cp - Cp.min()
            if not shift and Cp.min() < 0:
                continue  # negative costs: the cold JV's answer is still defined, but keep to the reference's use
            want = jv.lapjv(Cp)
            got = WarmStartLAPSolver().solve(C, uu, vv, shift_nonneg=shift)
            assert np.array_equal(got[1], want[1]) and got[2] == float(C[np.arange(64), want[1]].sum())
    r4, c4, cost4 = WarmStartLAPSolver(use_lap=False).solve(C, u, v)
    assert abs(cost4 - cost) < 1e-9
    stats = time_solver_rigorous(lambda: SeededLAPSolver().solve(C, u, v), num_warmups=1, num_repeats=3)
    assert stats["success"] and stats["num_samples"] == 3
    with pytest.raises(ValueError, match="u/v sizes must match C"):
        lap.lapjv_seeded(C, u[:-1].copy(), v)
    with pytest.raises(ValueError):
        lap.lapjv_seeded(np.asfortranarray(C), u, v)  # not C-contiguous
    with pytest.raises(ValueError):
        lap.lapjv_seeded(C.astype(np.float32), u, v)
    with pytest.raises(RuntimeError, match="code -4"):
        lap.lapjv_seeded(np.zeros((3, 2)), np.zeros(3), np.zeros(2))
    with pytest.raises(ValueError):
        lap.lapjv(np.zeros((3, 2)))
    with pytest.raises(ValueError):
        lap.lapjv(np.zeros(3))
    ret = lap.lapjv(C[:3, :3][:, ::1][::1])  # non-contiguous view is copied (test_lapjv_non_contigous)
    assert len(ret) == 3


# --------------------------------------------------------------------------- cold solve: candidate lists
def test_cold_row_reduction_candidate_lists_families(torch_cuda):
    """Cold `lapjv` at a size where the augmenting row reduction answers most iterations from per-row
    candidate lists (jv_solver.hip, cold_arr_sweep; n >= 512): seven cost families, assignments
    bit-exact and the ARR-iteration / path / relax-step counters equal to the oracle's (the lists change
    how the two minima of a row are found, never which iterations happen).  Uniform, sparse and tie
    costs must actually take the list path; low-rank and clustered ones go stale and take the full scan."""
    torch = torch_cuda
    from gnn import OneGNN, WarmStartPipeline
    from oracle import jv
    from solvers.generators import mixed_batch
    fams = ("uniform", "sparse", "metric", "clustered", "low_rank", "noisy_linear", "tie")
    n = 704  # 11 x 64: the last class block of a list build is partial for n % 256 != 0
    Cs, names = mixed_batch(len(fams), n, families=fams, seed=77)
    pipe = WarmStartPipeline(OneGNN(21), "cuda:0")
    x, y, ret, st = pipe.lapjv_batch(torch.from_numpy(Cs).cuda())
    torch.cuda.synchronize()
    st = st.cpu().numpy()
    for b in range(len(fams)):
        r, xo, yo, so = jv.dense_raw(Cs[b])
        assert r == int(ret[b]) == 0, (names[b], st[b, 12])
        assert np.array_equal(xo, x[b].cpu().numpy()) and np.array_equal(yo, y[b].cpu().numpy()), names[b]
        for q, k in ((11, "arr_iters"), (4, "paths"), (5, "finds"), (6, "scan_steps"), (10, "transfer_rows")):
            assert st[b, q] == so[k], (names[b], k, st[b, q], so[k])
        if names[b] in ("uniform", "sparse", "tie"):
            assert st[b, 27] > 0.9 * st[b, 11], (names[b], st[b, 27], st[b, 11])


def test_quality_gate_fallback_beside_seeded_instances(torch_cuda):
    """Seeds that fail the quality gate (lapjv_seeded.cpp:116; here all-zero seeds on positive costs: no
    tight edge at all) take the cold branch inside the seeded kernel, in the same batch as instances
    whose seeds pass -- at a size where the cold ENTRY points use the candidate lists, the seeded
    launch must keep solving its fallbacks with plain row scans.  Both kinds bit-exact."""
    torch = torch_cuda
    from gnn import OneGNN, WarmStartPipeline
    from gnn.features import min_trick_device
    from oracle import jv
    B, n = 4, 640
    Cs = np.stack([np.random.RandomState(3 + i).uniform(0.5, 1.5, (n, n)) for i in range(B)])
    C = torch.from_numpy(Cs).cuda()
    u = C.min(dim=2).values.contiguous()
    v = min_trick_device(C, u)
    u[1::2] = 0.0
    v[1::2] = 0.0
    pipe = WarmStartPipeline(OneGNN(21), "cuda:0")
    x, y, ret, stats = pipe.seeded_batch(C, u, v)
    torch.cuda.synchronize()
    st = stats.cpu().numpy()
    un, vn = u.cpu().numpy(), v.cpu().numpy()
    branches = set()
    for b in range(B):
        r, xo, yo, so = jv.seeded_raw(Cs[b], un[b], vn[b])
        assert r == int(ret[b]) == 0
        assert np.array_equal(xo, x[b].cpu().numpy()) and np.array_equal(yo, y[b].cpu().numpy())
        assert st[b, 0] == so["branch"] and st[b, 11] == so["arr_iters"] and st[b, 4] == so["paths"]
        branches.add(int(so["branch"]))
    assert len(branches) == 2, branches  # fallback and shortest-path instances side by side


def test_cold_integer_costs_with_lists_repeatable(torch_cuda):
    """Integer costs 1..100 at n = 1536 (512 threads x 4 columns per thread, a fifth of the row-reduction
    iterations falling back to the full scan and rebuilding their list): the geometry that failed --
    differently from run to run -- while the instantiation with the candidate lists also ran the shortest
    paths.  Three solves of the same batch, each bit-exact."""
    torch = torch_cuda
    from gnn import OneGNN, WarmStartPipeline
    from oracle import jv
    n, B = 1536, 2
    Cs = np.stack([np.random.RandomState(5 + i).randint(1, 101, (n, n)).astype(np.float64) for i in range(B)])
    C = torch.from_numpy(Cs).cuda()
    pipe = WarmStartPipeline(OneGNN(21), "cuda:0")
    ref = [jv.dense_raw(Cs[b]) for b in range(B)]
    for rep in range(3):
        x, y, ret, st = pipe.lapjv_batch(C)
        torch.cuda.synchronize()
        st = st.cpu().numpy()
        for b in range(B):
            r, xo, yo, so = ref[b]
            assert r == int(ret[b]) == 0, (rep, b, int(ret[b]), st[b, 12])
            assert np.array_equal(xo, x[b].cpu().numpy()) and np.array_equal(yo, y[b].cpu().numpy()), (rep, b)
            assert st[b, 11] == so["arr_iters"] and st[b, 4] == so["paths"] and st[b, 27] > 0


@pytest.mark.parametrize("n,fam", [(513, "int100"), (600, "tie"), (4608, "uniform"), (8192, "uniform")])
def test_cold_lists_at_boundary_and_cooperative_sizes(torch_cuda, n, fam):
    """Cold `lapjv` with candidate lists at the first sizes that use them (n = 513: the last column class
    holds one element; n = 600: not a multiple of 64) and above the cooperative threshold (n = 4608:
    preparation with lists, cooperative shortest paths, final phase -- three kinds of launches).  Bit-exact."""
    torch = torch_cuda
    from gnn import OneGNN, WarmStartPipeline
    from oracle import jv
    from solvers.generators import mixed_batch
    if fam == "int100":
        C0 = np.random.RandomState(9).randint(1, 101, (n, n)).astype(np.float64)
    elif fam == "uniform":
        C0 = np.random.RandomState(9).uniform(0, 1, (n, n))
    else:
        C0 = mixed_batch(1, n, families=(fam,), seed=9)[0][0]
    pipe = WarmStartPipeline(OneGNN(21), "cuda:0")
    x, y, ret, st = pipe.lapjv_batch(torch.from_numpy(C0).cuda().unsqueeze(0))
    torch.cuda.synchronize()
    st = st.cpu().numpy()
    r, xo, yo, so = jv.dense_raw(C0)
    assert r == int(ret[0]) == 0, st[0, 12]
    assert np.array_equal(xo, x[0].cpu().numpy()) and np.array_equal(yo, y[0].cpu().numpy())
    assert st[0, 11] == so["arr_iters"] and st[0, 4] == so["paths"] and st[0, 6] == so["scan_steps"]
    assert st[0, 27] > 0  # the lists were used


def test_quality_gate_fallback_at_a_cooperative_size(torch_cuda):
    """Seeds without a single tight edge at n = 4608: the preparation launch takes the cold branch (column
    reduction + row reduction with plain scans -- the seeded launches carry no candidate lists), the
    cooperative kernel searches the paths, the final launch writes the outputs.  Bit-exact, branch 3."""
    torch = torch_cuda
    from gnn import OneGNN, WarmStartPipeline
    from oracle import jv
    n = 4608
    C0 = np.random.RandomState(21).uniform(0.5, 1.5, (n, n))
    C = torch.from_numpy(C0).cuda().unsqueeze(0)
    u = torch.zeros((1, n), dtype=torch.float64, device="cuda")
    v = torch.zeros((1, n), dtype=torch.float64, device="cuda")
    pipe = WarmStartPipeline(OneGNN(21), "cuda:0")
    x, y, ret, stats = pipe.seeded_batch(C, u, v)
    torch.cuda.synchronize()
    st = stats.cpu().numpy()
    r, xo, yo, so = jv.seeded_raw(C0, np.zeros(n), np.zeros(n))
    assert r == int(ret[0]) == 0, st[0, 12]
    assert np.array_equal(xo, x[0].cpu().numpy()) and np.array_equal(yo, y[0].cpu().numpy())
    assert st[0, 0] == so["branch"] == 3 and st[0, 11] == so["arr_iters"] and st[0, 4] == so["paths"]
