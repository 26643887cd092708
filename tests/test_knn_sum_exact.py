"""The two-lanes-per-query k-NN kernel of the outlier filter (cwipc_util_amd/csrc/kernels_sor.hip, knn_pair_kernel) sums a query's
sixteen neighbour distances as a SET, in whatever order the pair's two lists leave them, where the reference (and the oracle:
oracle/cwipc_oracle.c, sor_filter; PCL StatisticalOutlierRemoval::applyFilterIndices) sums them in ascending order in f64.  That is
the same number bit for bit as long as the fp32 values lie within 2^23 of each other: every partial sum is then a multiple of the
smallest term's unit in the last place and stays below 2^53 of it, so no f64 addition rounds.  The kernel checks exactly this bound
(sqrtf(largest) < sqrtf(least) * 2^23) and sorts otherwise.  This test pins the arithmetic fact the kernel relies on."""
import numpy as np


def sum_in_order(values, order):
    s = np.float64(0.0)
    for i in order:
        s = s + np.float64(values[i])
    return s


def test_f64_sum_of_seventeen_fp32_values_within_2_pow_23_is_exact_in_any_order():
    rng = np.random.default_rng(20261005)
    for trial in range(300):
        n = int(rng.integers(2, 18))
        lo = np.float32(10.0 ** rng.uniform(-12, 6))
        # values between lo and lo * 2^23 (the kernel's bound, exclusive), any mantissas
        v = (lo * np.float32(2.0) ** rng.uniform(0, 23, n).astype(np.float32) * (1 + rng.random(n).astype(np.float32))).astype(np.float32)
        v = np.minimum(v, np.nextafter(np.float32(lo * np.float32(2.0 ** 23)), np.float32(0)))
        v[0] = lo
        exact = sum(int(np.float64(x) / np.float64(np.spacing(np.float32(lo)) / 2 ** 24)) for x in v)   # integers: multiples of a unit far below the smallest term's
        ascending = sum_in_order(v, np.argsort(v))
        for _ in range(5):
            assert sum_in_order(v, rng.permutation(n)) == ascending
        assert np.float64(exact) * np.float64(np.spacing(np.float32(lo)) / 2 ** 24) == ascending


def test_the_order_does_matter_beyond_that_bound():
    # one large term and sixteen small ones that only count together: ascending adds the small ones up first and keeps them
    w = np.array([2.0 ** 53] + [1.0] * 16, dtype=np.float32)
    assert sum_in_order(w, range(17)) == np.float64(2.0 ** 53)                  # each 1.0 is half a unit of the sum: rounded away, sixteen times
    assert sum_in_order(w, np.argsort(w, kind="stable")) == np.float64(2.0 ** 53 + 16)
