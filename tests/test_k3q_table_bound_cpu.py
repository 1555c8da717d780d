"""The error bound behind K3q's fp32 table (csrc/mmidx_scan_q.h), checked numerically on the CPU.

K3q decides pass A on integer sums a(c) = sum_s min(4095, floor(x_s)) where x_s is the scaled table entry
LUT[s][code_s] * 4000 / qr (IVFPQ.java:525-538 computes LUT in fp64) -- computed by the kernel in PACKED FP32 from fp32 copies of the
residual and the codebook.  The proof that `a(c2) >= a(c1) + 18` implies d(c2) > d(c1) needs the summed error of a code's sixteen
computed entries to stay under 0.5 table units for every code whose scaled distance Y is at most 4128; the kernel guarantees it by
sending a query to the exact kernel unless

    G = 2 u sqrt(4128 scale) (||r|| + max ||x||) <= 0.25,   u = 2^-24, scale = 4000 / qr,

and DESIGN.md claims  sum_s |x_s - Y_s| <= G + 13 u 4128 + O(u^2)  for those codes.  This file replays the kernel's arithmetic in
numpy (float32 roundings in the kernel's order; the fused multiply-add as an exact float64 product and sum rounded once -- both
operands have 24-bit significands, so the float64 intermediate is exact up to one rounding far below fp32's) on random and on
adversarial data (codebooks far from the origin, tiny and huge magnitudes) and checks the claim, and with it the two-sided bracket
Y - 16 - eps < a <= Y + eps of an unsaturated code."""
import numpy as np
import pytest

U = 2.0 ** -24
SCALE = 4000.0


def _table_entries(r, pq, qr):
    """x[s][j] as the kernel computes them (calc_rows): fp32 inputs, df = r - p rounded, acc = fma(df, df, acc), x = acc * inv"""
    m, ks, dsub = pq.shape
    r32 = r.astype(np.float32).reshape(m, 1, dsub)
    p32 = pq.astype(np.float32)
    inv32 = np.float32(SCALE / qr)
    df = (r32 - p32).astype(np.float32)  # one rounding (float32 - float32 in float32)
    acc = np.zeros((m, ks), np.float32)
    for t in range(dsub):
        d64 = df[:, :, t].astype(np.float64)
        acc = (d64 * d64 + acc.astype(np.float64)).astype(np.float32)  # v_pk_fma_f32: one rounding
    return (acc * inv32).astype(np.float32)


def _exact_scaled(r, pq, qr):
    """Y[s][j] = LUT[s][j] * 4000 / qr in extended precision (the real-number value the proof argues about)"""
    m, ks, dsub = pq.shape
    d = r.astype(np.longdouble).reshape(m, 1, dsub) - pq.astype(np.longdouble)
    return (d * d).sum(-1) * (np.longdouble(SCALE) / np.longdouble(qr))


def _case(rng, m, dsub, ks, offset, spread, mag):
    pq = (offset + spread * rng.standard_normal((m, ks, dsub))) * mag
    r = (offset + spread * rng.standard_normal(m * dsub)) * mag
    # the kernel's scale: the mean distance to a code with independent uniform entries (closed form from the codebook's statistics)
    mu = pq.mean(axis=1).reshape(-1)
    nu = (pq * pq).sum(-1).mean(axis=1)
    qr = float((r * r - 2.0 * r * mu).sum() + nu.sum())
    pmax = float(np.sqrt((pq * pq).sum(-1).max(axis=1).sum()))
    return r, pq, qr, pmax


@pytest.mark.parametrize("offset,spread,mag", [(0.0, 1.0, 1.0), (0.0, 1.0, 1e-9), (0.0, 1.0, 1e9), (3.0, 1.0, 1.0), (100.0, 1.0, 1.0),
                                               (30.0, 0.3, 1.0), (1000.0, 1.0, 1.0), (0.0, 1.0, 1e-13)])
def test_fp32_table_error_is_under_the_bound_the_kernel_checks(offset, spread, mag):
    m, dsub, ks = 16, 8, 256
    rng = np.random.default_rng(int(offset) + 7)
    worst = 0.0
    checked = 0
    for rep in range(6):
        r, pq, qr, pmax = _case(rng, m, dsub, ks, offset, spread, mag)
        if not (1e-24 < qr < 1e24):
            continue
        scale = SCALE / qr
        G = 2.0 * U * np.sqrt(4128.0 * scale) * (np.linalg.norm(r) + pmax)
        x = _table_entries(r, pq, qr).astype(np.longdouble)
        Y = _exact_scaled(r, pq, qr)
        codes = rng.integers(0, ks, (20000, m))
        # a few codes near the query as well: the nearest entry of every row and its neighbours in rank
        order = np.argsort(np.asarray(Y, np.float64), axis=1)
        near = order[:, rng.integers(0, 6, (4000, 1))[:, 0]].T if ks >= 6 else codes[:0]
        codes = np.concatenate([codes, near])
        rows = np.arange(m)
        Ys = Y[rows, codes].sum(-1)
        err = np.abs(x[rows, codes] - Y[rows, codes]).sum(-1)
        rel = Ys <= 4128.0  # the codes the argument needs
        if not rel.any():
            continue
        bound = G + 13.0 * U * 4128.0 + 1e-3  # (O(u^2) and the double rounding of this emulation: far below 1e-3)
        worst = max(worst, float((err[rel] - bound).max()))
        checked += int(rel.sum())
        assert float(err[rel].max()) <= bound, (offset, spread, mag, float(err[rel].max()), bound)
        if G <= 0.25:  # the kernel lets this query use the table: the bracket of an unsaturated code
            xs = x[rows, codes]
            unsat = rel & (xs < 4095.0).all(-1)
            a = np.floor(np.minimum(xs, 4095.0)).sum(-1)
            eps = 0.5
            assert np.all(a[unsat] <= Ys[unsat] + eps) and np.all(a[unsat] > Ys[unsat] - 16.0 - eps)
            # any code: never above Y + eps (a saturated entry is clamped DOWN to 4095)
            assert np.all(a[rel] <= Ys[rel] + eps)
    assert checked > 0 or mag != 1.0  # (extreme magnitudes may leave no code under 4128 units: nothing to check there)


def test_the_guard_trips_where_fp32_cannot_hold_the_table():
    """offset 1e5 spreads: G is far above 0.25 -- the kernel hands the query to the exact kernel (tests/test_gpu_parity.py runs it)"""
    rng = np.random.default_rng(3)
    r, pq, qr, pmax = _case(rng, 16, 8, 256, 1e5, 1.0, 1.0)
    G = 2.0 * U * np.sqrt(4128.0 * SCALE / qr) * (np.linalg.norm(r) + pmax)
    assert G > 0.25
    r, pq, qr, pmax = _case(rng, 16, 8, 256, 0.0, 1.0, 1.0)
    G = 2.0 * U * np.sqrt(4128.0 * SCALE / qr) * (np.linalg.norm(r) + pmax)
    assert G < 5e-3  # (the generated benchmarks: three orders of magnitude of room)


@pytest.mark.parametrize("offset,K1,n", [(0.0, 101, 12000), (0.0, 21, 3000), (50.0, 101, 12000), (0.0, 152, 6000)])
def test_integer_selection_keeps_every_code_at_or_under_its_threshold(offset, K1, n):
    """K3q's rule, replayed whole: a* = the K1-th smallest integer sum, candidates = codes with a <= a* + 17, T = the largest exact
    distance among the candidates with a <= a* (at least K1 of them).  Claim: no code outside the candidates has an
    exact distance <= T -- so the pool K3q publishes (candidates with d <= T) is exactly {codes with d <= T}, what the exact kernel
    K3h publishes for the same T.  Exact distances: the fp64 table summed in sub-quantizer order (IVFPQ.java:435-438)."""
    m, dsub, ks = 16, 8, 256
    rng = np.random.default_rng(K1 + n)
    for rep in range(4):
        r, pq, qr, pmax = _case(rng, m, dsub, ks, offset, 1.0, 1.0)
        G = 2.0 * U * np.sqrt(4128.0 * SCALE / qr) * (np.linalg.norm(r) + pmax)
        assert G <= 0.25
        x = _table_entries(r, pq, qr)
        q = np.minimum(np.floor(x), 4095.0).astype(np.int64)  # the u16 entries
        lut = ((r.reshape(m, 1, dsub) - pq) ** 2).sum(-1)  # fp64 (the order of the inner sum does not matter for this claim)
        codes = rng.integers(0, ks, (n, m))
        # a cluster of near codes, as a list around the query has: perturb the best code of every row
        best = np.argsort(lut, axis=1)[:, :24]
        codes[: n // 8] = best[np.arange(m), rng.integers(0, 24, (n // 8, m))]
        rows = np.arange(m)
        a = q[rows, codes].sum(-1)
        d = np.zeros(n)
        for s in range(m):  # sub-quantizer order
            d = d + lut[s, codes[:, s]]
        astar = np.sort(a)[K1 - 1]
        assert astar < 4095  # (else the kernel hands the query back)
        cand = a <= astar + 17
        evid = a <= astar
        assert evid.sum() >= K1
        T = d[evid].max()
        outside = ~cand
        assert not np.any(d[outside] <= T), (float(d[outside].min()), float(T))
        # and the pool holds at least K1 entries, at most the candidates
        assert K1 <= int((d[cand] <= T).sum()) <= int(cand.sum())
