"""np_compat.h claims about NumPy/BLAS arithmetic, checked against values captured from NumPy
on the fixture machine (tests/golden/numpy_semantics.npz)."""
import math

import numpy as np


def _fma(a, b, c):
    from fractions import Fraction

    return float(Fraction(a) * Fraction(b) + Fraction(c))


def test_blas_inner_products_are_fma_chains(golden_dir):
    z = np.load(golden_dir / "numpy_semantics.npz")
    v, m2, dot = z["v"], z["m2"], z["dot"]
    for i in range(len(v)):
        for j in range(2):
            assert dot[i, j] == _fma(v[i, 1], m2[1, j], v[i, 0] * m2[0, j])
        assert z["norm1d"][i] == math.sqrt(_fma(v[i, 1], v[i, 1], v[i, 0] * v[i, 0]))
    cd = z["cdist"]
    for i in range(40):
        for j in range(40):
            a, b = v[i], v[j]
            acc = 1.0 * (b[0] * b[0])
            acc = _fma(1.0, b[1] * b[1], acc)
            acc = _fma(a[0], -2 * b[0], acc)
            acc = _fma(a[1], -2 * b[1], acc)
            acc = _fma(a[0] * a[0], 1.0, acc)
            acc = _fma(a[1] * a[1], 1.0, acc)
            assert cd[i, j] == acc


def test_pairwise_sum(golden_dir):
    z = np.load(golden_dir / "numpy_semantics.npz")
    v = z["v"][:, 0]

    def pw(a):
        n = len(a)
        if n < 8:
            r = 0.0
            for x in a:
                r += x
            return r
        if n <= 128:
            r = list(a[:8])
            i = 8
            while i < n - (n % 8):
                for j in range(8):
                    r[j] += a[i + j]
                i += 8
            res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
            while i < n:
                res += a[i]
                i += 1
            return res
        n2 = n // 2
        n2 -= n2 % 8
        return pw(a[:n2]) + pw(a[n2:])

    for k in range(1, 400):
        assert z["sums"][k - 1] == pw(v[:k])


def test_det3_sign_is_numpy_linalg_det_sign():
    """path_parameterization.py:86-92 takes the sign of the curvature from np.linalg.det of three homogeneous points; on an
    exactly straight stretch the determinant is zero up to rounding and its sign is the rounding of OpenBLAS' LU.  The
    restatement (np_compat.h det3_lu: left-looking getf2, reciprocal scaling, fused one-row gemv) gives NumPy's sign on
    collinear, nearly collinear, permuted and general triples at scales 1e-2 .. 1e4 m."""
    import ctypes

    import oracle_lib

    L = oracle_lib.lib()
    L.fsdo_det3.restype = ctypes.c_double
    rng = np.random.default_rng(3)
    bad = zeros = 0
    n = 20000
    for k in range(n):
        sc = 10.0 ** rng.uniform(-2, 4)
        p0, d = rng.normal(0, sc, 2), rng.normal(0, 1, 2)
        t1, t2 = rng.uniform(0.01, 5, 2)
        p = np.array([p0, p0 + t1 * d, p0 + (t1 + t2) * d])
        if k % 4 == 1:
            p += rng.normal(0, 1e-13 * sc, p.shape)
        elif k % 4 == 2:
            p = p[rng.permutation(3)]
        elif k % 4 == 3:
            p = rng.normal(0, sc, (3, 2))
        ref = np.sign(np.linalg.det(np.column_stack((np.ones(3), p))))
        mine = np.sign(L.fsdo_det3(p.ravel().ctypes.data_as(ctypes.POINTER(ctypes.c_double))))
        bad += ref != mine
        zeros += ref == 0
    assert bad == 0, (bad, n)
    assert zeros > n // 10  # exact zeros are part of the contract (the reference then returns curvature 0)


def test_dot_of_a_single_point_is_gemv_order(golden_dir):
    """rotate() of the reference (utils/math_utils.py:103-117) is np.dot(points, R).  For ONE point — a 1-D vector or a (1, 2)
    array — NumPy calls gemv and OpenBLAS forms fma(x, R0j, y * R1j); from two rows on it is gemm: fma(y, R1j, x * R0j)
    (np_compat.h blas_dot2_single_row / blas_dot2).  The skidpad mission transforms the car's position as a single point
    (skidpad_relocalizer.py:140-153): with the gemm order there, 2 of the 341 frames of the replay flipped their sample count
    (rounds 1-3: attributed to libm; tests/test_skidpad_cpu.py now asserts none)."""
    z = np.load(golden_dir / "numpy_semantics_gemv.npz")
    v, c, s = z["v"].tolist(), z["cos"].tolist(), z["sin"].tolist()
    c07, s07 = float(z["cos07"]), float(z["sin07"])
    differ = 0
    for i in range(len(v)):
        R = ((c[i], s[i]), (-s[i], c[i]))  # rotation_matrix = [[c, -s], [s, c]].T
        R07 = ((c07, s07), (-s07, c07))
        for j in range(2):
            gemv = _fma(v[i][0], R[0][j], v[i][1] * R[1][j])
            gemm = _fma(v[i][1], R[1][j], v[i][0] * R[0][j])
            differ += gemv != gemm
            assert z["single"][i, j] == gemv and z["one_row"][i, j] == gemv
            if i < len(z["two_rows"]):
                assert z["two_rows"][i, j] == gemm
            assert z["many"][i, j] == _fma(v[i][1], R07[1][j], v[i][0] * R07[0][j])
    assert differ > 60  # the two orders do differ on these inputs (the fixture can tell them apart)
